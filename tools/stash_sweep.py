"""Sweep of the launch-shaping knobs (read by libkajo_hip_tune.so when a handle is created, kajo_amd/csrc/tuning.h; the deferred
experiment's stash / ring knobs by libkajo_hip_exp.so):
KAJO_STASH_DEPTH, KAJO_RING_SLOTS, KAJO_THR_L, KAJO_THR_STALL, KAJO_STEAL_WINDOW, KAJO_WAVES_PER_BLOCK.

  python tools/stash_sweep.py [c2|c4|c5|c1] [fast|strict] [--lib PATH ...] "D=2,R=4,L=40,S=12,W=2,B=1" ...

Every configuration renders the workload twice warm and reports the best in-kernel rate of three further launches
(HIP events of the library), with traversals per path and lane efficiency from the device counters. Libraries given with
--lib are timed beside the shipped one in the same process order (e.g. the round-2 kernels: kajo_amd/libkajo_hip_r02.so),
each in a child process because the library is chosen at import.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"D": "KAJO_STASH_DEPTH", "R": "KAJO_RING_SLOTS", "L": "KAJO_THR_L", "S": "KAJO_THR_STALL", "W": "KAJO_STEAL_WINDOW",
        "B": "KAJO_WAVES_PER_BLOCK", "X": "KAJO_LDS_EXTRA", "N": "KAJO_SHADOW_BINS", "G": "KAJO_GRID_LDS_LIMIT", "H": "KAJO_HOLD_TRIPS"}


def child(workload, mode, configs):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import warnings
    warnings.filterwarnings("ignore")
    import numpy as np
    from kajo_amd import capi
    from kajo_amd.renderer import HipRenderer
    from kajo_amd.scene import Scene, stress_scene
    z = np.load(os.path.join(ROOT, "tests/golden/scenes.npz"))
    a169 = Scene.from_npz(z, "spheres_a169/", "spheres 16:9")
    cases = {
        "c1": (Scene.from_npz(z, "spheres_a1/", "spheres a1"), 256, 256, 16, 1, 1),
        "c2": (a169, 1920, 1080, 32, 16, 8),
        "c4": (Scene.from_npz(z, "caustics_a169/", "caustics"), 1920, 1080, 32, 16, 8),
        "c5": (stress_scene(a169, 1000, 16), 3840, 2160, 32, 4, 8),
        "c5full": (stress_scene(a169, 1000, 16), 3840, 2160, 32, 32, 8),
    }
    sc, W, H, S, passes, depth = cases[workload]
    for cfg in configs:
        for k in KEYS.values():
            os.environ.pop(k, None)
        for item in filter(None, cfg.split(",")):
            k, v = item.split("=")
            os.environ[KEYS[k]] = v
        try:
            if os.environ.get("KAJO_SWEEP_WALL"):  # no device counters (their per-wave atomics weigh on launches of many short waves): wall clock over 50 launches
                import time
                with HipRenderer(sc, W, H, spp=S, depth_limit=depth, strict=(mode == "strict"), passes_per_launch=passes) as r:
                    r.render(passes).wait()
                    r.render(passes).wait()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        r.render(passes)
                    r.wait()
                    dt = (time.perf_counter() - t0) / 50
                print("%-14s %-6s %-36s %9.1f Mpaths/s  wall %8.3f ms per launch (50 launches back to back, no counters)" % (
                    os.path.basename(capi.LIB_PATH).replace("libkajo_hip", "lib").replace(".so", ""), mode, cfg or "(defaults)",
                    W * H * int(S ** .5) ** 2 * passes / dt / 1e6, dt * 1e3), flush=True)
                continue
            with HipRenderer(sc, W, H, spp=S, depth_limit=depth, strict=(mode == "strict"), counters=True, passes_per_launch=passes) as r:
                r.render(passes).wait()
                r.render(passes).wait()
                best = None
                for _ in range(3):
                    c0 = r.counters()
                    r.render(passes).wait()
                    c1 = r.counters()
                    ms = c1["kernelMs"] - c0["kernelMs"]
                    if best is None or ms < best[0]:
                        best = (ms, c0, c1)
            ms, c0, c1 = best
            paths = c1["paths"] - c0["paths"]
            trav = c1["traversals"] - c0["traversals"]
            print("%-14s %-6s %-36s %9.1f Mpaths/s  kernel %8.3f ms  trav/path %.3f  lane eff %.3f" % (
                os.path.basename(capi.LIB_PATH).replace("libkajo_hip", "lib").replace(".so", ""), mode, cfg or "(defaults)",
                paths / ms / 1e3, ms, trav / paths, trav / max(1, c1["laneSlots"] - c0["laneSlots"])), flush=True)
        except Exception as e:  # a configuration the library refuses (LDS budget)
            print("%-14s %-6s %-36s refused: %s" % (os.path.basename(capi.LIB_PATH), mode, cfg, e), flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(args[1], args[2], json.loads(args[3]))
        return
    workload = args.pop(0) if args and args[0] in ("c1", "c2", "c4", "c5", "c5full") else "c2"
    mode = args.pop(0) if args and args[0] in ("fast", "strict") else "fast"
    libs = [None]
    configs = []
    while args:
        a = args.pop(0)
        if a == "--lib":
            libs.append(args.pop(0))
        else:
            configs.append(a)
    configs = configs or [""]
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["KAJO_HIP_LIB"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
        cfgs = configs  # (a --lib built without -DKAJO_TUNING ignores the knobs)
        if lib is None and any(cfgs):  # knobs are read by the tools' twin only (kajo_amd/csrc/tuning.h); the product library ignores them
            env["KAJO_HIP_LIB"] = os.path.join(ROOT, "kajo_amd", "libkajo_hip_tune.so")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", workload, mode, json.dumps(cfgs)], env=env, check=False)


if __name__ == "__main__":
    main()
