"""A roofline line for every BASELINE.json config that fits one GPU, per numerics build: in-kernel rate (HIP events of the handle, launches back
to back), work per path from the device counters, the algorithmic FLOP of SURVEY.md section 8d, fraction of the FP32 vector peak.
  small scenes (every object walked):  FLOP/path = T (14 nPlanes + 28 nSpheres) + 150 V        T = closest-hit walks, V = vertices, per path
  grid scenes (configs[4]):            FLOP/path = W 14 nPlanes + 28 S + 150 V                 W = grid walks + list queries per path,
      S = sphere tests the RAYS' OWN walks run per path: the spheres registered in the cells a ray crosses up to its hit, the items of its
      query's list up to its reach, the light itself -- counted per lane by the diagnostic twin (libkajo_hip_count.so, KAJO_COUNT_TESTS) on a
      1920 x 1080 x 4-pass frame of the same scene and camera. The brute-force count of Raytracer.cpp:126-138 (1016 tests per walk) is not
      what a culled walk has to do; this is.
usage: configs_roofline.py [out.json]   (needs kajo_amd/libkajo_hip_count.so for the grid scene's S: make -C kajo_amd/csrc count)"""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, warnings
warnings.filterwarnings("ignore")
PEAK = 157.3

if len(sys.argv) > 1 and sys.argv[1] == "--count-tests":  # child process on the diagnostic twin: sphere tests per path of the grid scene
    from kajo_amd import capi
    from kajo_amd.renderer import HipRenderer
    from kajo_amd.scene import Scene, stress_scene
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    sc = stress_scene(Scene.from_npz(z, "spheres_a169/", "s"), 1000, 16)
    res = {}
    for m in ("fast", "strict"):
        with HipRenderer(sc, 1920, 1080, spp=32, depth_limit=8, strict=(m == "strict"), counters=True, passes_per_launch=4) as r:
            r.render(4).wait()
            c = r.counters()
            out = (C.c_ulonglong * 28)()
            capi.check(capi.lib().kajo_hip_debug_profile(r._h, out))
        res[m] = {"paths": c["paths"], "grid_walks": c["traversals"], "list_queries": c["shadowQueries"], "vertices": c["vertices"],
                  "grid_sphere_tests": out[25], "list_sphere_tests": out[26], "light_sphere_tests": out[27]}
    print(json.dumps(res))
    raise SystemExit(0)

from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import Scene, stress_scene
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
a1 = Scene.from_npz(z, "spheres_a1/", "spheres.json 1:1")
a169 = Scene.from_npz(z, "spheres_a169/", "spheres.json 16:9")
caus = Scene.from_npz(z, "caustics_a169/", "caustics")
stress = stress_scene(a169, 1000, 16)
cases = [  # key, label, scene, W, H, S, passes per frame (= per launch), depth
    ("configs[0]", "data/spheres.json 256x256, 16 spp, 1 bounce", a1, 256, 256, 16, 1, 1),
    ("configs[1]", "data/spheres.json 1920x1080, 512 spp = 16 x S32", a169, 1920, 1080, 32, 16, 8),
    ("configs[2] on one GPU", "data/spheres.json 3840x2160, 2048 spp = 64 x S32 (the 8-GPU config's whole frame on one)", a169, 3840, 2160, 32, 64, 8),
    ("configs[3]", "caustics (ideal reflector, glass, Phong, 3 lights) 1920x1080, 4096 spp = 128 x S32 in launches of 16, 8 bounces", caus, 1920, 1080, 32, 16, 8),
    ("configs[4]", "1000 spheres / 16 lights 3840x2160, 1024 spp = 32 x S32", stress, 3840, 2160, 32, 32, 8),
]
tests = None
prof = os.path.join(ROOT, "kajo_amd", "libkajo_hip_count.so")
if os.path.exists(prof):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--count-tests"], env=dict(os.environ, KAJO_HIP_LIB=prof), capture_output=True, text=True)
    if p.returncode == 0:
        tests = json.loads(p.stdout.strip().splitlines()[-1])
    else:
        print("diagnostic twin failed:", p.stderr[-500:], file=sys.stderr)
rows = []
for key, label, sc, W, H, S, P, depth in cases:
    for m in ("exact", "fast", "strict"):
        kw = dict(strict=(m == "strict"), exact=(m == "exact"))
        with HipRenderer(sc, W, H, spp=S, depth_limit=depth, passes_per_launch=P, **kw) as r:
            r.render(P).wait(); r.render(P).wait()
            reps = 20 if W * H * P < 4e6 else (4 if W * H * P < 2e8 else 2)
            c0 = r.counters()
            for _ in range(reps):
                r.render(P)
            r.wait()
            c1 = r.counters()
        ms = (c1["kernelMs"] - c0["kernelMs"]) / (c1["launches"] - c0["launches"])
        paths = (c1["paths"] - c0["paths"]) / (c1["launches"] - c0["launches"])
        with HipRenderer(sc, W, H, spp=S, depth_limit=depth, passes_per_launch=P, counters=True, **kw) as r:
            c = r.render(min(P, 16)).counters()
        T = (c["traversals"] + c["shadowQueries"]) / c["paths"]
        V = c["vertices"] / c["paths"]
        row = {"config": key, "workload": label, "numerics": m, "kernel_ms_per_launch": ms, "value": paths / ms / 1e3, "unit": "Msamples/s (in-kernel)",
               "walks_per_path": T, "vertices_per_path": V, "lane_efficiency": c["traversals"] / max(c["laneSlots"], 1)}
        big = sc.n_spheres >= 48
        if not big:
            fpp = T * (14 * sc.n_planes + 28 * sc.n_spheres) + 150 * V
            row["model"] = "T (14 nPlanes + 28 nSpheres) + 150 V"
        elif tests:
            t = tests["strict" if m != "fast" else "fast"]
            Sp = (t["grid_sphere_tests"] + t["list_sphere_tests"] + t["light_sphere_tests"]) / t["paths"]
            fpp = T * 14 * sc.n_planes + 28 * Sp + 150 * V
            row["model"] = "W 14 nPlanes + 28 S + 150 V, S = sphere tests of the rays' own culled walks (diagnostic twin)"
            row["sphere_tests_per_path"] = {"grid_walks": t["grid_sphere_tests"] / t["paths"], "list_walks": t["list_sphere_tests"] / t["paths"],
                                            "lights": t["light_sphere_tests"] / t["paths"], "counted_on": "1920x1080 x 4 passes, libkajo_hip_count.so"}
            row["brute_force_model_flops_per_path"] = T * (14 * sc.n_planes + 28 * sc.n_spheres) + 150 * V
        else:
            fpp = None
        if fpp:
            ach = fpp * paths / (ms * 1e-3) / 1e12
            row.update({"flops_per_path": fpp, "roofline": {"bound": "valu", "achieved": ach, "peak": PEAK, "unit": "TFLOP/s", "frac": ach / PEAK}})
        rows.append(row)
        print("%-22s %-6s %9.1f Msamples/s  %8.3f ms  %6.1f FLOP/path  frac %s" % (key, m, row["value"], ms, fpp or 0, ("%.4f" % row["roofline"]["frac"]) if fpp else "-"), flush=True)
out = {"collected": "MI355X, tools/configs_roofline.py", "peak": "157.3 TFLOP/s FP32 vector (MI355X_MICROARCH.md)", "rows": rows}
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
