#!/usr/bin/env python3
"""Where does the FAST-vs-oracle RMSE of bench.py's parity leg come from? Lists the pixels that carry it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w, h = 256, 144
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
scene = Scene.from_npz(z, "spheres_a169/", "spheres")
want = OracleLib("oracle").create(scene, 0).render(w, h, S=32, passes=passes, seed=0o715517, depth_limit=8, threads=64)[..., :3] / passes
for strict in (False, True):
    r = HipRenderer(scene, w, h, spp=32, depth_limit=8, seed=0o715517, strict=strict)
    got = r.render(passes).radiance()[..., :3] / passes
    r.close()
    cl = np.clip(got, 0, 1) - np.clip(want, 0, 1)
    sq = (cl ** 2).sum(-1)
    tot = sq.sum()
    print("strict" if strict else "fast", "rmse_clamped", np.sqrt(tot / cl.size), "rmse_unclamped", np.sqrt(np.mean((got - want) ** 2)))
    order = np.argsort(sq.ravel())[::-1]
    cum = np.cumsum(sq.ravel()[order]) / max(tot, 1e-30)
    for k in (1, 5, 20, 100, 1000):
        print("  top %4d pixels carry %.1f %% of the squared error" % (k, 100 * cum[k - 1]))
    rest = np.sqrt((tot - sq.ravel()[order[:100]].sum()) / cl.size)
    print("  rmse without the top 100 pixels: %.3g" % rest)
    for i in order[:8]:
        y, x = divmod(int(i), w)
        print("   px (%3d,%3d) got %s want %s" % (x, y, got[y, x], want[y, x]))
