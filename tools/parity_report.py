"""Parity numbers for DESIGN.md / the judge: the HIP kernels against the golden frames captured from the
compiled reference (both builds), against the oracle, and the reference's own two-build floor.
Writes gpurun_out/r02_parity.json (run on the GPU box; copied to profiles/). Round 2 adds the frames of
tests/golden/frames2.npz: the 64-pass converged frame and the 1080p x 16-pass crops of BASELINE configs[1]."""
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
# ---- round 2: frames2.npz
z2 = np.load(os.path.join(ROOT, "tests", "golden", "frames2.npz"))
sc = Scene.from_npz(zs, "spheres_a1/", "spheres_a1")
with HipRenderer(sc, 64, 64, spp=32, depth_limit=8, seed=seed) as r:
    fast = r.render(64).radiance()[..., :3] / 64
with HipRenderer(sc, 64, 64, spp=32, depth_limit=8, seed=seed, strict=True) as r:
    strict = r.render(64).radiance()[..., :3] / 64
ref_s, ref_f = z2["conv_64/rgb_strict"] / 64, z2["conv_64/rgb_fast"] / 64
out["frames"]["conv_64 (64 passes)"] = {"config": {"scene": "spheres_a1", "W": 64, "H": 64, "S": 32, "passes": 64, "depth": 8},
    "reference_O2_vs_reference_fastmath (floor)": stats(ref_s, ref_f), "hip_fast_vs_reference_O2": stats(fast, ref_s),
    "hip_strict_vs_reference_O2": stats(strict, ref_s)}
sc = Scene.from_npz(zs, "spheres_a169/", "spheres_a169")
with HipRenderer(sc, 1920, 1080, spp=32, depth_limit=8, seed=seed) as r:
    fast = r.render(16).radiance()[..., :3] / 16
with HipRenderer(sc, 1920, 1080, spp=32, depth_limit=8, seed=seed, strict=True) as r:
    strict = r.render(16).radiance()[..., :3] / 16
crop = lambda a: np.stack([a[y:y + h, x:x + w] for x, y, w, h in z2["c2_1080p/crops"]])
ref_s, ref_f = z2["c2_1080p/rgb_crops_strict"] / 16, z2["c2_1080p/rgb_crops_fast"] / 16
out["frames"]["c2_1080p x16 passes, 8 crops of 64x32"] = {"config": {"scene": "spheres_a169", "W": 1920, "H": 1080, "S": 32, "passes": 16, "depth": 8},
    "reference_O2_vs_reference_fastmath (floor)": stats(ref_s, ref_f), "hip_fast_vs_reference_O2": stats(crop(fast), ref_s),
    "hip_strict_vs_reference_O2": stats(crop(strict), ref_s)}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r02_parity.json"), "w"), indent=1)
for n, f in out["frames"].items():
    print(n)
    for k, v in f.items():
        if k != "config":
            print("   %-46s median %.2e p99 %.2e clamped-rmse %.2e identical %.3f" % (k, v["median_abs"], v["p99_abs"], v["rmse_clamped01"], v["identical_frac"]))
