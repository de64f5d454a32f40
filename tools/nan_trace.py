#!/usr/bin/env python3
"""Replays the camera paths of given pixels of BASELINE configs[1] (1920 x 1080, 16 passes) one by one through the oracle (strict math),
the compiled reference's -O2 build where it travelled, and the STRICT kernels (kajo_hip_kat_shade), and prints the paths whose
radiance differs in its bits or is not finite, with the oracle's event log. usage: nan_trace.py x y [x y ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib, available, debug_path, camera_ray

W, H, S, PASSES, SEED = 1920, 1080, 32, 16, 0o715517
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
scene = Scene.from_npz(z, "spheres_a169/", "spheres")
orc = OracleLib("oracle").create(scene, 1)
ref = OracleLib("ref_strict").create(scene) if available("ref_strict") else None
args = [int(a) for a in sys.argv[1:]]
for x, y in zip(args[0::2], args[1::2]):
    rays = np.zeros((PASSES * 25, 6), np.float32)
    states = np.zeros((PASSES * 25, 2), np.uint64)
    k = 0
    for p in range(1, PASSES + 1):
        for s in range(25):
            o, d, st = camera_ray(orc, W, H, S, x, y, s, npass=p, seed=SEED)
            rays[k, :3], rays[k, 3:], states[k] = o, d, st
            k += 1
    o_rgb, o_fin = orc.shade(rays[:, :3], rays[:, 3:], states, depth_limit=8)
    with HipRenderer(scene, W, H, spp=S, depth_limit=8, seed=SEED, strict=True) as r:
        g_rgb, g_fin = r.kat_shade(rays[:, :3], rays[:, 3:], states)
    r_rgb = ref.shade(rays[:, :3], rays[:, 3:], states, depth_limit=8)[0] if ref else None
    bad = np.nonzero(((o_rgb.view(np.uint32) != g_rgb.view(np.uint32)) & ~(np.isnan(o_rgb) & np.isnan(g_rgb))).any(-1) | ~np.isfinite(o_rgb).all(-1) | ~np.isfinite(g_rgb).all(-1))[0]
    print("pixel (%d, %d): %d of %d paths differ or are not finite; final generator states equal on %d" % (x, y, bad.size, len(rays), int((o_fin == g_fin).all(-1).sum())))
    for k in bad[:4]:
        p, s = divmod(int(k), 25)
        print("  pass %d sample %d: oracle %s  kernel %s  reference -O2 %s" % (p + 1, s, o_rgb[k], g_rgb[k], r_rgb[k] if r_rgb is not None else "-"))
        ev, rgb = debug_path(orc, W, H, S, x, y, s, npass=p + 1, seed=SEED, depth_limit=8)
        for e in ev:
            print("     ", e)
