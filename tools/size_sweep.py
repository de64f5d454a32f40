#!/usr/bin/env python3
"""Throughput of the FAST kernels against the frame size (16 passes x S=32 per launch, spheres.json 16:9):
how many pixels it takes to fill the chip (256 CUs x 4 SIMDs x 4 waves = 4096 resident waves = 262 144 px)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch  # noqa: F401
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene

z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
sc = Scene.from_npz(z, "spheres_a169/", "spheres")
from kajo_amd import capi
NO_SPLIT = capi.KAJO_FLAG_NO_SPLIT if "--no-split" in sys.argv else 0
SIZES = ((64, 36), (128, 72), (256, 144), (512, 288), (640, 360), (640, 480), (800, 600), (960, 540), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320))
if "--small" in sys.argv:
    SIZES = ((256, 144), (512, 288), (640, 360), (640, 480), (800, 600), (960, 540), (1280, 720), (1600, 900), (1920, 1080))
MODE = "fast" if "fast" in sys.argv else ("strict" if "strict" in sys.argv else "exact")  # default: the build bench.py times
print("numerics", MODE, flush=True)
for (w, h) in SIZES:
    with HipRenderer(sc, w, h, counters=True, flags=NO_SPLIT, strict=(MODE == "strict"), exact=(MODE == "exact")) as r:
        r.render(16).wait()  # records the trip counts the launch order uses
        c0 = r.counters()
        n = 3
        for _ in range(n):
            r.render(16)
        r.wait()
        c1 = r.counters()
    ms = (c1["kernelMs"] - c0["kernelMs"]) / n
    paths = (c1["paths"] - c0["paths"]) / n
    print("%5dx%-5d %9d px %6d waves  kernel %8.3f ms  %8.1f M paths/s" % (w, h, w * h, -(-w // 8) * -(-h // 8), ms, paths / ms / 1e3))
