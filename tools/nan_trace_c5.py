#!/usr/bin/env python3
"""As tools/nan_trace.py, for a pixel of the 1000-sphere / 16-light scene at 1920 x 1080, 2 passes: the pixel's paths one by one through the
oracle (strict math) and the STRICT large-scene kernels (a whole small render restricted to the pixel is not possible; the frame is rendered
and the pixel compared). usage: nan_trace_c5.py x y"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, stress_scene
from oraclelib import OracleLib
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
sc = stress_scene(Scene.from_npz(z, "spheres_a169/", "s"), 1000, 16)
x, y = int(sys.argv[1]), int(sys.argv[2])
W, H, P = 1920, 1080, 2
x0, y0 = max(0, x - 8), max(0, y - 4)
want = OracleLib("oracle").create(sc, 1).render(W, H, S=32, passes=P, seed=0o715517, depth_limit=8, rect=(x0, y0, 16, 8), threads=16)[y0:y0 + 8, x0:x0 + 16, :3]
for flags in (0, 128):
    with HipRenderer(sc, W, H, spp=32, depth_limit=8, seed=0o715517, strict=True, passes_per_launch=P, flags=flags) as r:
        got = r.render(P).radiance()[y0:y0 + 8, x0:x0 + 16, :3]
    d = ((got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))).any(-1)
    print("flags %d: %d of 128 pixels around (%d, %d) differ; the pixel: kernel %s oracle %s" % (flags, int(d.sum()), x, y, got[y - y0, x - x0], want[y - y0, x - x0]))
