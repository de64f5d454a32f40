"""One workload, one numerics build, a few launches in steady state -- the command tools/pmc_workload.sh profiles (the first launch of a
handle runs in image order and measures the blocks: the counters' medians and the steady-state statistics leave it out).
usage: launch_workload.py c2|c4|c5 [fast|strict|exact] [launches=N] [nolists]
  c2  BASELINE configs[1]: spheres.json 1920 x 1080, 16 passes per launch      (kajo_render_<mode>)
  c4  BASELINE configs[3]: caustics scene (3 lights) 1920 x 1080, 16 passes    (kajo_render_<mode>_lights)
  c5  BASELINE configs[4]: 1000 spheres / 16 lights 3840 x 2160, 32 passes     (kajo_render_<mode>_biglist)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, warnings
warnings.filterwarnings("ignore")
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, stress_scene
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
a169 = Scene.from_npz(z, "spheres_a169/", "s")
key = next((a for a in sys.argv[1:] if a in ("c2", "c4", "c5")), "c2")
mode = next((a for a in sys.argv[1:] if a in ("fast", "strict", "exact")), "exact")
launches = next((int(a[9:]) for a in sys.argv[1:] if a.startswith("launches=")), 4)
sc, W, H, P = {"c2": (a169, 1920, 1080, 16), "c4": (Scene.from_npz(z, "caustics_a169/", "c"), 1920, 1080, 16),
               "c5": (stress_scene(a169, 1000, 16), 3840, 2160, 32)}[key]
flags = 128 if "nolists" in sys.argv else 0
with HipRenderer(sc, W, H, spp=32, depth_limit=8, strict=(mode == "strict"), exact=(mode == "exact"), passes_per_launch=P, flags=flags) as r:
    r.render(P).wait()
    r.render(P).wait()
    c0 = r.counters()
    t = time.perf_counter()
    for _ in range(launches):
        r.render(P)
    r.wait()
    dt = (time.perf_counter() - t) / launches
    c1 = r.counters()
print("%s %dx%d x %d passes %s%s: %.2f ms per launch (%.2f in-kernel), %.1f M paths/s" % (key, W, H, P, mode.upper(), " (no lists)" if flags else "", dt * 1e3,
      (c1["kernelMs"] - c0["kernelMs"]) / launches, W * H * 25 * P / dt / 1e6))
