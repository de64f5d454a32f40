"""The three numerics builds side by side on one workload: in-kernel rate (HIP events of the handle), device counters, and -- with
`parity` -- the whole frame against the oracle (clamped RMSE, pixels off by more than 1e-3, not-a-number pixels).
usage: modes.py [c2|c4|c5|c1] [parity] [modes=fast,strict,exact] [reps=N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, warnings
warnings.filterwarnings("ignore")
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, stress_scene

z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
a169 = Scene.from_npz(z, "spheres_a169/", "spheres.json 16:9")
cases = {
    "c2": ("configs[1] spheres.json 1920x1080 x 16", a169, 1920, 1080, 32, 16, 8),
    "c4": ("configs[3] caustics 1920x1080 x 16 (of 128)", Scene.from_npz(z, "caustics_a169/", "caustics"), 1920, 1080, 32, 16, 8),
    "c5": ("configs[4] 1000 spheres / 16 lights 3840x2160 x 32", stress_scene(a169, 1000, 16), 3840, 2160, 32, 32, 8),
    "c5s": ("configs[4] 1000 spheres / 16 lights 1920x1080 x 2", stress_scene(a169, 1000, 16), 1920, 1080, 32, 2, 8),
    "c1": ("configs[0] 256x256 S=16 1 pass depth 1", Scene.from_npz(z, "spheres_a1/", "spheres 1:1"), 256, 256, 16, 1, 1),
}
args = sys.argv[1:]
modes = ["fast", "strict", "exact"]
reps = 5
keys = []
parity = False
for a in args:
    if a.startswith("modes="):
        modes = a[6:].split(",")
    elif a.startswith("reps="):
        reps = int(a[5:])
    elif a == "parity":
        parity = True
    else:
        keys.append(a)
SEED = 0o715517
for key in keys or ["c2"]:
    name, sc, W, H, S, P, depth = cases[key]
    print(name, flush=True)
    want = None
    if parity:
        from oraclelib import OracleLib
        from bench import host_cores
        t0 = time.time()
        want = OracleLib("oracle").create(sc, 1).render(W, H, S=S, passes=P, seed=SEED, depth_limit=depth, threads=max(1, min(host_cores(), 64)))
        print("  oracle(strict) %.0f s" % (time.time() - t0), flush=True)
    for m in modes:
        kw = dict(strict=(m == "strict"), exact=(m == "exact"))
        with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=SEED, passes_per_launch=P, **kw) as r:
            r.render(P).wait(); r.render(P).wait()
            c0 = r.counters()
            for _ in range(reps):
                r.render(P)
            r.wait()
            c1 = r.counters()
            ms = (c1["kernelMs"] - c0["kernelMs"]) / reps
            paths = (c1["paths"] - c0["paths"]) / reps
        line = "  %-7s %8.3f ms in-kernel  %9.1f M paths/s" % (m, ms, paths / ms / 1e3)
        with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=SEED, passes_per_launch=P, counters=True, **kw) as r:
            got = r.render(P).radiance()
            c = r.counters()
        line += "  trav/path %.3f vert/path %.3f lane eff %.3f" % (c["traversals"] / c["paths"], c["vertices"] / c["paths"], c["traversals"] / max(1, c["laneSlots"]))
        if want is not None:
            g, w = got[..., :3].astype(np.float64) / P, want[..., :3].astype(np.float64) / P
            ng, nw = ~np.isfinite(g).all(-1), ~np.isfinite(w).all(-1)
            ok = ~(ng | nw)
            d = np.abs(g - w)[ok]
            rmse = np.sqrt(np.mean((np.clip(g[ok], 0, 1) - np.clip(w[ok], 0, 1)) ** 2))
            line += "\n          vs oracle(strict): clamped RMSE %.3e, max |d| %.3e, px off > 1e-3: %d, bit-identical px %.4f, NaN px here %d / oracle %d / both %d" % (
                rmse, d.max(), int((d.max(-1) > 1e-3).sum()), float(np.mean((got[..., :3].view(np.uint32) == want[..., :3].view(np.uint32)).all(-1))),
                int(ng.sum()), int(nw.sum()), int((ng & nw).sum()))
        print(line, flush=True)
