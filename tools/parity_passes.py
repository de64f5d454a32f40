"""Where does FAST meet BASELINE.json's per-pixel RMSE < 1e-4? The bench's parity leg (whole 256 x 144 frame, clamped [0, 1]
radiance estimate, FAST kernels vs oracle(libm), same streams) at the pass counts of the BASELINE configs: a path whose
hit / miss decision flips moves its pixel by one path's radiance / (25 passes), so the RMSE of a given flip rate falls roughly as
1 / sqrt(passes). Writes gpurun_out/r04_parity_passes.json (copied to profiles/r04_parity.json by hand).
usage: python tools/parity_passes.py   (GPU box; the oracle runs on the host cores)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from kajo_amd.renderer import HipRenderer  # noqa: E402
from kajo_amd.scene import Scene, stress_scene  # noqa: E402
from oraclelib import OracleLib  # noqa: E402
from bench import host_cores  # noqa: E402

SEED = 0o715517
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
a169 = Scene.from_npz(z, "spheres_a169/", "spheres.json 16:9")
caustics = Scene.from_npz(z, "caustics_a169/", "caustics (3 lights)")
cases = [("configs[1] spheres.json", a169, 256, 144, (4, 16, 64, 128)),
         ("configs[2] spheres.json (64 passes)", a169, 256, 144, ()),  # same scene: covered by the row above
         ("configs[3] caustics", caustics, 256, 144, (16, 128)),
         ("configs[4] 1000 spheres / 16 lights", stress_scene(a169, 1000, 16), 128, 72, (2, 32))]
threads = max(1, min(host_cores(), 64))
rows = []
O = OracleLib("oracle")
for name, sc, W, H, pass_counts in cases:
    for passes in pass_counts:
        t0 = time.time()
        res = {"scene": name, "frame": "%dx%d" % (W, H), "passes": passes}
        for strict in (False, True):
            want = O.create(sc, 1 if strict else 0).render(W, H, S=32, passes=passes, seed=SEED, depth_limit=8, threads=threads)[..., :3] / passes
            with HipRenderer(sc, W, H, spp=32, depth_limit=8, seed=SEED, strict=strict, passes_per_launch=min(passes, 64)) as r:
                got = r.render(passes).radiance()[..., :3] / passes
            m = np.isfinite(got) & np.isfinite(want)
            cl = np.where(m, np.clip(got, 0, 1) - np.clip(want, 0, 1), 0.0)
            sq = np.sort((cl ** 2).sum(-1).ravel())[::-1]
            tag = "strict" if strict else "fast"
            res[tag + "_rmse_clamped01"] = float(np.sqrt(sq.sum() / cl.size))
            res[tag + "_px_off_by_more_than_1e-3"] = int((np.abs(cl).max(-1) > 1e-3).sum())
            res[tag + "_meets_1e-4"] = bool(res[tag + "_rmse_clamped01"] < 1e-4)
            if not strict:
                res["fast_rmse_without_worst_20_px"] = float(np.sqrt(sq[20:].sum() / cl.size))
                res["fast_median_abs"] = float(np.median(np.abs(got - want)[m]))
        res["seconds"] = round(time.time() - t0, 1)
        rows.append(res)
        print(json.dumps(res), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r04_parity_passes.json"), "w"), indent=1)
