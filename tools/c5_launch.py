"""One warm and two measured launches of BASELINE configs[4] (1000 spheres / 16 lights, 3840 x 2160, 32 passes in one launch) -- the
command tools/pmc_c5.sh profiles. usage: c5_launch.py [strict|exact] [nolists]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, warnings
warnings.filterwarnings("ignore")
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, stress_scene
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
sc = stress_scene(Scene.from_npz(z, "spheres_a169/", "s"), 1000, 16)
strict, exact, flags = "strict" in sys.argv, "exact" in sys.argv, (128 if "nolists" in sys.argv else 0)
with HipRenderer(sc, 3840, 2160, spp=32, depth_limit=8, strict=strict, exact=exact, passes_per_launch=32, flags=flags) as r:
    r.render(32).wait()
    t = time.perf_counter()
    r.render(32).render(32).wait()
    dt = (time.perf_counter() - t) / 2
print("C5 4K x 32 passes %s%s: %.1f ms per launch, %.1f M paths/s" % ("STRICT" if strict else "EXACT" if exact else "FAST", " (no lists)" if flags else "", dt * 1e3, 3840 * 2160 * 25 * 32 / dt / 1e6))
