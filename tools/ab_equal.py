#!/usr/bin/env python3
"""Do two builds of the library render the same FAST (or STRICT) buffers, bit for bit? Each library in a child process of its own (it is
chosen at import, KAJO_HIP_LIB), frames of spheres.json, the caustics scene and test.json kept as .npy under /tmp and compared.
usage: ab_equal.py libA.so libB.so [fast|strict]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("spheres_a169", 1280, 720, 32, 4, 8), ("caustics_a169", 960, 540, 32, 4, 8), ("test_a1", 512, 512, 16, 3, 8), ("spheres_a1", 256, 256, 16, 1, 1),
         ("dialect_a1", 400, 300, 9, 2, 5)]


def child(tag, mode):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import numpy as np
    from kajo_amd.renderer import HipRenderer
    from kajo_amd.scene import Scene
    z = np.load(os.path.join(ROOT, "tests/golden/scenes.npz"))
    for key, W, H, S, passes, depth in CASES:
        with HipRenderer(Scene.from_npz(z, key + "/", key), W, H, spp=S, depth_limit=depth, strict=(mode == "strict"), passes_per_launch=2) as r:
            np.save("/tmp/ab_%s_%s.npy" % (tag, key), r.render(passes).radiance())


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    import numpy as np
    a, b = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else "fast"
    for tag, lib in (("a", a), ("b", b)):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag, mode], env=dict(os.environ, KAJO_HIP_LIB=os.path.abspath(lib)), check=True)
    for key, W, H, S, passes, depth in CASES:
        x, y = np.load("/tmp/ab_a_%s.npy" % key)[..., :3], np.load("/tmp/ab_b_%s.npy" % key)[..., :3]
        same = (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))
        d = np.abs(x - y)
        print("%s %-14s %4dx%-4d S=%d x%d: %d of %d px differ%s" % (mode, key, W, H, S, passes, int((~same).any(-1).sum()), W * H,
              "" if same.all() else "; max |d| %.3g, median of the differing %.3g" % (float(np.nanmax(d)), float(np.nanmedian(d[~same])))), flush=True)
