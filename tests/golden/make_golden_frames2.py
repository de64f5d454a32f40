#!/usr/bin/env python3
"""Round-2 additions to the golden frames (fixture 7), again from the COMPILED REFERENCE
(oracle/_ref through oracle/ref_harness.cpp, both builds), written to tests/golden/frames2.npz:

  c1_256      BASELINE configs[0] at its full size: spheres.json aspect 1, 256 x 256, S = 16, one pass, depth limit 1.
              ARGB8 of the whole frame + the float sums of six 64 x 32 crops (a whole float frame is 786 KB).
  conv_64     a converged frame: 64 x 64, S = 32, 64 passes, depth 8. Float sums + ARGB8.
  c2_1080p    BASELINE configs[1] at its full size and pass count: spheres.json 16:9, 1920 x 1080, 16 passes x S = 32,
              depth 8 -- the float sums of eight 64 x 32 crops (glass sphere, its silhouette, Phong sphere, a diffuse
              sphere, the emitter, floor shadow, two corners), which the reference renders in seconds.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_frames2.py
The file holds data only: the crop rectangles and the reference's outputs.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from kajo_amd.scene import Scene  # noqa: E402
from oraclelib import OracleLib  # noqa: E402
from workload_crops import feature_crops  # noqa: E402

SEED = 0o715517


def main():
    fast, strict = OracleLib("ref"), OracleLib("ref_strict")
    z = np.load(os.path.join(HERE, "scenes.npz"))
    a1 = Scene.from_npz(z, "spheres_a1/", "spheres_a1")
    a169 = Scene.from_npz(z, "spheres_a169/", "spheres_a169")
    out = {"seed": np.uint64(SEED)}

    W = H = 256
    crops = feature_crops(a1, W, H)[:6]
    out["c1_256/crops"] = np.array([(x, y, w, h) for _, x, y, w, h in crops], np.int32)
    out["c1_256/crop_names"] = json.dumps([c[0] for c in crops])
    for tag, L in (("fast", fast), ("strict", strict)):
        h = L.create(a1)
        acc = h.render(W, H, S=16, passes=1, seed=SEED, depth_limit=1)
        out["c1_256/argb8_" + tag] = L.resolve(acc, 1).reshape(H, W)
        out["c1_256/rgb_crops_" + tag] = np.stack([acc[y:y + hh, x:x + ww, :3] for _, x, y, ww, hh in crops])
        h.close()

    for tag, L in (("fast", fast), ("strict", strict)):
        h = L.create(a1)
        acc = h.render(64, 64, S=32, passes=64, seed=SEED, depth_limit=8)
        out["conv_64/rgb_" + tag] = acc[..., :3].copy()
        out["conv_64/argb8_" + tag] = L.resolve(acc, 64).reshape(64, 64)
        h.close()

    W, H, P = 1920, 1080, 16
    crops = feature_crops(a169, W, H)[:8]
    out["c2_1080p/crops"] = np.array([(x, y, w, h) for _, x, y, w, h in crops], np.int32)
    out["c2_1080p/crop_names"] = json.dumps([c[0] for c in crops])
    for tag, L in (("fast", fast), ("strict", strict)):
        h = L.create(a169)
        got = []
        for _, x, y, ww, hh in crops:
            acc = h.render(W, H, S=32, passes=P, seed=SEED, depth_limit=8, rect=(x, y, ww, hh))
            got.append(acc[y:y + hh, x:x + ww, :3].copy())
        out["c2_1080p/rgb_crops_" + tag] = np.stack(got)
        h.close()
    np.savez_compressed(os.path.join(HERE, "frames2.npz"), **out)
    print("frames2.npz %d bytes" % os.path.getsize(os.path.join(HERE, "frames2.npz")))


if __name__ == "__main__":
    main()
