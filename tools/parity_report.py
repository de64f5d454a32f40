"""Parity numbers for DESIGN.md / the judge: the HIP kernels against the golden frames captured from the
compiled reference (both builds), against the oracle, and the reference's own two-build floor.
Writes profiles/r01_parity.json (run on the GPU box)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene
from oraclelib import OracleLib

z = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))
zs = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
frames = json.loads(str(z["frames"])); seed = int(z["seed"])
O = OracleLib("oracle")

def stats(a, b):
    m = np.isfinite(a) & np.isfinite(b)
    d = np.abs(a - b)[m]
    cl = np.sqrt(np.mean(((np.clip(a, 0, 1) - np.clip(b, 0, 1)) ** 2)[m]))
    return {"median_abs": float(np.median(d)), "p99_abs": float(np.percentile(d, 99)), "max_abs": float(d.max()),
            "rmse_clamped01": float(cl), "identical_frac": float(np.mean(d == 0)), "nonfinite": int((~m).sum())}

out = {"note": "radiance estimate = accumulation / passes, RGB; all frames at seed 0715517", "frames": {}}
for name, key, W, H, S, passes, depth in frames:
    sc = Scene.from_npz(zs, key + "/", key)
    with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=seed) as r:
        fast = r.render(passes).radiance()[..., :3] / passes
    with HipRenderer(sc, W, H, spp=S, depth_limit=depth, seed=seed, strict=True) as r:
        strict = r.render(passes).radiance()[..., :3] / passes
    ref_s, ref_f = z[name + "/rgb_strict"] / passes, z[name + "/rgb_fast"] / passes
    ora = O.create(sc, 0).render(W, H, S=S, passes=passes, seed=seed, depth_limit=depth)[..., :3] / passes
    ora_s = O.create(sc, 1).render(W, H, S=S, passes=passes, seed=seed, depth_limit=depth)[..., :3] / passes
    out["frames"][name] = {
        "config": {"scene": key, "W": W, "H": H, "S": S, "passes": passes, "depth": depth},
        "reference_O2_vs_reference_fastmath (floor)": stats(ref_s, ref_f),
        "hip_fast_vs_reference_O2": stats(fast, ref_s), "hip_fast_vs_reference_fastmath": stats(fast, ref_f),
        "hip_fast_vs_oracle_libm": stats(fast, ora),
        "hip_strict_vs_oracle_strict": stats(strict, ora_s), "hip_strict_vs_reference_O2": stats(strict, ref_s),
        "oracle_libm_vs_reference_O2": stats(ora, ref_s),
    }
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r01_parity.json"), "w"), indent=1)
for n, f in out["frames"].items():
    print(n)
    for k, v in f.items():
        if k != "config":
            print("   %-46s median %.2e p99 %.2e clamped-rmse %.2e identical %.3f" % (k, v["median_abs"], v["p99_abs"], v["rmse_clamped01"], v["identical_frac"]))
