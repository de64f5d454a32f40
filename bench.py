#!/usr/bin/env python3
"""Benchmark of the HIP backend on the reference's headline workloads (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself, as a child process
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, rendezvous on 127.0.0.1) -- decided before
anything touches the GPU; the parent only waits and passes the child's exit code on. Launched under
torch.distributed.run (the driver's multi-GPU form) the ranks read RANK / LOCAL_RANK / WORLD_SIZE as usual.

One STEP = one complete frame of the hot path, rendered into HBM, gathered to rank 0 (ONE RCCL gather of the tile
buffers when N > 1), composed and resolved to ARGB8 on the device. The staged scene is resident in HBM before the
timed region; nothing is read back to the host inside it.

  N = 1   BASELINE configs[1]: data/spheres.json, 1920 x 1080, 512 spp = 16 passes x S=32
  N > 1   BASELINE configs[2]: data/spheres.json, 3840 x 2160, 2048 spp = 64 passes x S=32, the SAME frame at every
          N (strong scaling): 64 x 16-pixel tiles dealt round-robin over the ranks, no collective on the data path
          other than the per-frame gather. (`--workload c3 --gpus 1` renders that frame on one GPU.)

(25 camera paths are traced per pixel per pass, renderer/cpu/Renderer.cpp:21,38; depth limit 8, MIS on.)

value = Msamples/s = camera paths of all ranks / max-over-ranks wall time. Rank 0 prints ONE
JSON line. The oracle / compiled reference are used ONLY for the cpu_baseline and parity legs, never inside the timed
region.

Which numerics build is timed (round 5): EXACT -- the one of the three (include/kajo_hip.h: fast, strict, exact) that is the
fastest to meet BASELINE.json's "per-pixel RMSE < 1e-4 vs reference" on the frame being timed. `value`, `roofline` and
`config.numerics` are that build's; the parity leg renders the WHOLE timed frame (1920 x 1080 x 16 passes) with the CPU
oracle (~25 s of the host's cores, outside the timed region) and compares every build with it; the other two builds are
reported beside it (`fast_mode`, `strict_mode`). `--fast` / `--strict` time those instead.
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SPP = 32
DEPTH = 8
SEED = 0o715517
PEAK_FP32_TFLOPS = 157.3  # MI355X vector FP32, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
WORKLOADS = {
    # name: (W, H, passes per frame, passes fused per kernel launch, label)
    "c2": (1920, 1080, 16, 16, "BASELINE configs[1]"),
    "c3": (3840, 2160, 64, 64, "BASELINE configs[2]"),
}
NUMERICS = ("fast", "strict", "exact")
KERNEL_SOURCES = ["kajo_amd/csrc/integrator.inc.hip", "kajo_amd/csrc/kernel_fast.hip", "kajo_amd/csrc/kernel_strict.hip", "kajo_amd/csrc/kernel_exact.hip",
                  "kajo_amd/csrc/launch.inc.hip", "kajo_amd/csrc/render_args.h", "kajo_amd/csrc/device_scene.h",
                  "kajo_amd/csrc/Makefile", "include/kajo_stream.h", "include/kajo_strictmath.h",
                  # launch shaping (hold thresholds, steal window, split / sample-chunk selection, waves per block; grid and
                  # material records): they change instruction counts and lane utilisation as the kernel text does
                  "kajo_amd/csrc/capi.cpp", "kajo_amd/csrc/stage.cpp", "kajo_amd/csrc/tuning.h"]


def kernel_source_hash():
    """Identifies the kernels a counters file was collected on: SHA-256 over the sources the render kernels are built from."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


class DevicePtr:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 3}


def host_cores():
    """CPU cores this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show
    all hardware threads of the host but grant a quota of a few cores; oversubscribing that only adds throttling)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(math.ceil(q / per))))
        except (OSError, ValueError):
            pass
    return n


def flops_per_path(n_planes, n_spheres, traversals_per_path, vertices_per_path):
    """T * (14 nPlanes + 28 nSpheres) + V * 150."""
    return traversals_per_path * (14 * n_planes + 28 * n_spheres) + vertices_per_path * 150


def counters_files():
    """profiles/r<NN>_counters.json, newest round first (written by tools/pmc_summary.py from rocprofv3 --pmc runs; ONE place
    decides which file is current: the newest one collected on the kernels being timed)."""
    import glob
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_counters.json")), reverse=True)


def profile_counters(kernel, workload):
    """PMC figures of the render kernel per launch (HBM bytes; FP32 work actually executed), taken with rocprofv3 --pmc
    in separate passes over THIS command and committed under profiles/ (bench.py cannot collect counters itself).
    Returned only for the kernel and workload they were taken on; the source file travels in the line."""
    want = kernel_source_hash()
    for path in counters_files():
        e = json.load(open(path)).get(kernel)
        if not e or e.get("workload") != workload:
            continue
        if e.get("kernel_source_hash") != want:
            continue  # collected on other kernels than the ones being timed: stale figures are not reported
        e = dict(e)
        e["source"] = os.path.relpath(path, ROOT)
        return e
    return None


def cpu_baseline(scene, W, H):
    """The reference's own hot loop (cpu::Renderer::render on row slices, one std::async per core,
    renderer/cpu/Scheduler.cpp:32-42) timed on the host cores this box grants, on a bounded sample of the same frame:
    from the compiled reference (oracle/_ref, kind "reference") when it travelled with the snapshot, and from this
    repo's scalar port built with the reference's flag set (kind "port") beside it."""
    from oraclelib import OracleLib, available

    cores = host_cores()
    legs = {}
    if available("ref"):
        legs["reference"] = OracleLib("ref").create(scene)
    if available("oracle_fast"):
        legs["port"] = OracleLib("oracle_fast").create(scene, 0)
    elif available("oracle"):
        legs["port"] = OracleLib("oracle").create(scene, 0)
    out = {}
    budget = 14.0 / max(len(legs), 1)  # seconds of CPU wall time per leg
    for kind, h in legs.items():
        t1, _ = h.render_native(W, H, 1, cores)
        passes = max(1, min(32, int(round(budget / max(t1, 1e-3)))))
        t, _ = h.render_native(W, H, passes, cores)
        t1c, _ = h.render_native(W, H // 16, 1, 1)  # one thread, a sixteenth of the rows
        out[kind] = {"value": W * H * 25 * passes / t / 1e6, "one_thread_value": W * (H // 16) * 25 / t1c / 1e6,
                     "sample": "%dx%d, %d pass(es) x 25 paths/px, reference slicing (1 slice per core), %.1f s" % (W, H, passes, t)}
    kind = "reference" if "reference" in out else "port"
    res = {"value": out[kind]["value"], "unit": "Msamples/s", "cores": cores, "kind": kind,
           "sample": out[kind]["sample"] + "; cores = affinity mask capped by the cgroup CPU quota (%d hardware threads visible)" % (os.cpu_count() or 0),
           "one_thread_value": out[kind]["one_thread_value"]}
    for k, v in out.items():
        res[k] = v  # both kinds side by side: {"reference": {...}, "port": {...}}
    return res


def oracle_frame(scene, W, H, passes, math=1):
    """The timed frame rendered by the CPU oracle (strict math: the arithmetic the STRICT kernels reproduce bit for bit and the
    EXACT kernels decision for decision), on the host cores this box grants; ~25 s for 1920 x 1080 x 16 passes on 16 cores."""
    from oraclelib import OracleLib

    t0 = time.perf_counter()
    want = OracleLib("oracle").create(scene, math).render(W, H, S=SPP, passes=passes, seed=SEED, depth_limit=DEPTH, threads=max(1, min(host_cores(), 64)))
    return want, time.perf_counter() - t0


def parity_leg(want, oracle_s, got, W, H, passes):
    """Per-pixel figures of the radiance estimate of one numerics build against the CPU oracle over the WHOLE timed frame."""
    # bit for bit; a NaN channel (the reference emits them: inf * 0 at grazing glass hits) matches a NaN, whatever its payload
    same = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)))[..., :3].all(-1)
    nan_g, nan_w = ~np.isfinite(got[..., :3]).all(-1), ~np.isfinite(want[..., :3]).all(-1)
    g, w = got[..., :3] / np.float32(passes), want[..., :3] / np.float32(passes)
    m = np.isfinite(g) & np.isfinite(w)
    d = np.abs(g - w)[m]
    cl = np.where(m, np.clip(g, 0, 1).astype(np.float64) - np.clip(w, 0, 1), 0.0)
    sq = np.sort((cl ** 2).sum(-1).ravel())[::-1]  # per-pixel squared error, largest first
    rmse = float(np.sqrt(sq.sum() / cl.size))
    # same streams, different roundings: in the FAST build a 1e-7 difference flips a hit/miss decision in a few paths per
    # million, each moving its pixel by one path's worth of radiance; those few pixels carry its RMSE. EXACT flips none.
    return {"frame": "%dx%d, %d passes (the timed frame, every pixel), vs the CPU oracle (strict math), rendered in %.0f s on %d host cores" % (W, H, passes, oracle_s, host_cores()),
            "bit_identical_px": int(same.sum()), "px": int(same.size),
            "median_abs": float(np.median(d)), "p99_abs": float(np.percentile(d, 99)), "max_abs": float(d.max()),
            "rmse_clamped01": rmse, "meets_north_star_rmse": bool(rmse < 1e-4),
            "rmse_clamped01_without_worst_100_px": float(np.sqrt(sq[100:].sum() / cl.size)),
            "share_of_sq_error_in_worst_20_px": float(sq[:20].sum() / sq.sum()) if sq.sum() > 0 else None,
            "px_off_by_more_than_1e-3": int((np.abs(cl).max(-1) > 1e-3).sum()),
            # the same over the LINEAR (unclamped) estimate -- the clamp hides every pixel brighter than 1, the lights above all -- and
            # relative to the oracle's own magnitude (include/kajo_hip.h KAJO_EXACT_REL_TOL is a bound on the last figure for EXACT)
            "rmse_linear": float(np.sqrt((np.where(m, g.astype(np.float64) - w, 0.0) ** 2).sum() / m.size)),
            "max_rel_to_max_oracle_1e-3": float((np.abs(g - w) / np.maximum(np.abs(w), 1e-3))[m].max()),
            "nan_px": int(nan_g.sum()), "nan_px_oracle": int(nan_w.sum()), "nan_px_in_both": int((nan_g & nan_w).sum())}


def mode_leg(scene, W, H, passes, ppl, local_rank, numerics, fpp):
    """One numerics build on the timed frame, outside the timed region: rate, roofline fraction (the same algorithmic FLOP per
    path: every build traces the same paths) and the frame itself for the parity leg."""
    from kajo_amd.renderer import HipRenderer

    r = HipRenderer(scene, W, H, spp=SPP, depth_limit=DEPTH, seed=SEED, strict=(numerics == "strict"), exact=(numerics == "exact"),
                    device=local_rank, passes_per_launch=ppl)
    # the first launch measures the blocks; from the second on the launches run in cost order with the cheapest blocks in parts
    # (FAST / EXACT; capi.cpp partTheTail) -- the parity leg gets the frame AS THE TIMED LAUNCHES RENDER IT
    r.render(passes).wait()
    r.reset()
    got = r.render(passes).radiance()
    r.render(passes).wait()
    c0 = r.counters()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        r.render(passes)
    r.wait()
    dt = (time.perf_counter() - t0) / reps
    c1 = r.counters()
    r.close()
    launches = max(c1["launches"] - c0["launches"], 1)
    kernel_ms = (c1["kernelMs"] - c0["kernelMs"]) / launches
    achieved = fpp * (c1["paths"] - c0["paths"]) / launches / (kernel_ms * 1e-3) / 1e12
    return {"numerics": numerics, "value": (c1["paths"] - c0["paths"]) / reps / dt / 1e6, "unit": "Msamples/s",
            "ms_per_step": dt * 1e3, "kernel_ms_per_launch": kernel_ms, "tail_groups_per_launch": int(c1.get("tailGroups", 0)),
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "kernel": "kajo_render_" + numerics, "flops_per_path": fpp}}, got


def scheduler_run_leg(scene, W, H, passes, numerics, frames=8):
    """The path a Kajo user runs: hip::Scheduler::run() behind the reference's plugin interface (renderer/Scheduler.h:12-16),
    driven by the headless kajo_render binary in a CHILD process (its own HIP context; started before this process touches the
    GPU). Every refresh renders `passes` passes, composes, resolves and copies the ARGB8 image into Image::pixels on the host
    (renderer/cpu/Renderer.cpp:73-75 writes Image::pixels in place), synchronously. Reported from the refreshes after the
    first (which loads the kernels and records the launch order)."""
    import tempfile
    exe = os.path.join(ROOT, "kajo_amd", "host", "kajo_render")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        pod = os.path.join(tmp, "scene.pod")
        scene.write_pod(pod)
        cmd = [exe, "-w", str(W), "-h", str(H), "-r", "hip", "--passes", str(passes * (frames + 2)), "--batch", str(passes), "--gpus", "1",
               "-o", "", "--json", "--scene-pod", pod, "--" + numerics]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        return {"error": (p.stderr or p.stdout)[-400:]}
    st = json.loads(p.stdout.strip().splitlines()[-1])
    ms = sorted(st["batch_ms"][2:])
    med = ms[len(ms) // 2]
    n = int(math.sqrt(SPP))
    return {"value": W * H * n * n * passes / (med * 1e-3) / 1e6, "unit": "Msamples/s", "ms_per_frame": med,
            "first_frame_ms": st["batch_ms"][0], "frames_timed": len(ms), "passes_per_frame": passes,
            "includes": "render kernel + compose + resolve + host ARGB8 read-back of %d bytes into Image::pixels, one synchronous refresh per frame" % (W * H * 4),
            "numerics": numerics,
            "command": "kajo_render -w %d -h %d -r hip --passes %d --batch %d --gpus 1 --json --scene-pod <spheres.json fixture> --%s" % (W, H, passes * (frames + 2), passes, numerics)}


STAGE_TAG = "KAJO_BENCH_STAGE"


def note_stage(rank, stage):
    """Every rank says on stderr where it is: what the parent (or a reader of the driver's log) sees last is where a run stopped."""
    STATE["stage"] = stage
    sys.stderr.write("%s rank=%d stage=%s\n" % (STAGE_TAG, rank, stage))
    sys.stderr.flush()


STATE = {"stage": "start", "printed": False}


def stage_from_stderr(lines):
    """-> {rank: last stage it reported}"""
    last = {}
    for l in lines:
        if l.startswith(STAGE_TAG):
            f = dict(kv.split("=", 1) for kv in l.split()[1:] if "=" in kv)
            last[int(f.get("rank", -1))] = f.get("stage", "?")
    return last


def error_line(n_gpus, args, stage, error, stderr_tail=None):
    """The ONE JSON line of a run that failed: the contract's keys with value null, and where / why it stopped."""
    return {"metric": "Msamples/s", "value": None, "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if n_gpus > 1 else None, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "error": error, "stage": stage, "stderr_tail": stderr_tail,
            "config": {"backend": args.backend if n_gpus > 1 else None, "gather_direct": bool(getattr(args, "gather_direct", False))}}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)  # (the first launches of a handle measure and order its blocks: five untimed steps leave them behind)
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c3"],
                    help="auto: c2 (1080p, 16 passes) on one GPU, c3 (4K, 64 passes, strong scaling) on several")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the cpu_baseline / parity / other-mode legs")
    ap.add_argument("--exact", action="store_true", help="time the EXACT kernels (the default: the fastest build that meets BASELINE.json's RMSE < 1e-4)")
    ap.add_argument("--strict", action="store_true", help="time the STRICT kernels (bit-identical to the oracle)")
    ap.add_argument("--fast", action="store_true", help="time the FAST kernels (1.6 x the rate; RMSE ~6e-4 on the timed frame)")
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="N = 1: after the K timed steps, this many seconds of back-to-back "
                    "frames of the same build, reported as `sustained` (0 = skip)")
    ap.add_argument("--passes-per-launch", type=int, default=0)
    ap.add_argument("--no-check", action="store_true", help="N > 1: skip the bit-for-bit check against a one-GPU frame")
    ap.add_argument("--separate-compose", action="store_true",
                    help="N > 1: rank 0 composes the whole float frame and resolves that (two passes over the data, rounds 1-3) instead "
                         "of resolving straight from the gathered tile buffers")
    ap.add_argument("--backend", default="nccl",
                    help="nccl (= RCCL; the graded runs) or gloo: rehearsal of the N > 1 control flow on fewer GPUs than ranks (gather staged "
                         "through host memory, every rank on GPU LOCAL_RANK %% device_count). Any other name fails in init_process_group "
                         "(tests/test_multi_rank_cpu.py uses that to see the error line)")
    ap.add_argument("--gather-direct", action="store_true",
                    help="N > 1, nccl: send straight from the library's own tile buffer instead of a torch-allocated copy of it (the copy is "
                         "there because memory the library allocated is foreign to the process group's allocator: this switch lets a run on "
                         "real hardware tell whether it is needed)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # Start the ranks as a CHILD process; this process has not touched the GPU (importing torch does not).
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # The ranks' stdout is passed on (rank 0 prints the line); their stderr is shown AND kept, so that a failed run -- the first
        # contact with N > 1 hardware must be diagnosable -- still ends in ONE parseable JSON line: which stage, which rank, the
        # tail of what the ranks said.
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        import threading
        err_lines = []

        def pump():
            for line in p.stderr:
                sys.stderr.write(line)
                err_lines.append(line)

        t = threading.Thread(target=pump, daemon=True)
        t.start()
        out_text = p.stdout.read()
        rc = p.wait()
        t.join(timeout=10)
        sys.stdout.write(out_text)
        has_line = any(l.startswith("{") and '"metric"' in l for l in out_text.splitlines())
        if rc != 0 or not has_line:
            if not has_line:  # (a rank-0 failure prints its own error line: not a second one)
                said = [l.strip() for l in err_lines if l.startswith("KAJO_BENCH_ERROR")]
                what = ("the ranks exited with code %d" % rc if rc else "the ranks printed no result line") + ("; " + said[0] if said else "")
                print(json.dumps(error_line(args.gpus, args, stage_from_stderr(err_lines), what, "".join(err_lines[-40:]))), flush=True)
            raise SystemExit(rc or 1)
        raise SystemExit(0)
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if world > 1:
        # A rank that fails takes the others down with it (torch.distributed.run sends them SIGTERM): rank 0 then still prints the line.
        import signal

        def terminated(signum, frame):
            if rank == 0 and not STATE["printed"]:
                STATE["printed"] = True
                print(json.dumps(error_line(world, args, {0: STATE["stage"]}, "rank 0 was terminated (signal %d) at stage '%s': another rank "
                                            "failed or the launcher gave up; the ranks' stderr has their stages" % (signum, STATE["stage"]))), flush=True)
            os._exit(1)

        signal.signal(signal.SIGTERM, terminated)
    try:
        run_ranks(args, world, rank, local_rank)
    except BaseException as e:  # (SystemExit with a message included: "needs a GPU")
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        if world > 1:
            sys.stderr.write("KAJO_BENCH_ERROR rank=%d stage=%s: %s: %s\n" % (rank, STATE["stage"], type(e).__name__, e))
            if rank == 0 and not STATE["printed"]:
                STATE["printed"] = True
                print(json.dumps(error_line(world, args, {0: STATE["stage"]}, "%s: %s" % (type(e).__name__, e))), flush=True)
        raise


def run_ranks(args, world, rank, local_rank):
    numerics = "fast" if args.fast else ("strict" if args.strict else "exact")
    sched_leg = None
    if world == 1 and not args.no_cpu_baseline and args.workload in ("auto", "c2") and torch.cuda.device_count() > 0:
        # the plugin path, in a child process, BEFORE this process initialises the GPU (device_count() does not)
        from kajo_amd.scene import Scene as _Scene
        _z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
        sched_leg = scheduler_run_leg(_Scene.from_npz(_z, "spheres_a169/", "spheres.json 16:9"), *WORKLOADS["c2"][:3], numerics)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend != "nccl":  # (host-side backends need no device: the group comes first, so that its failures are not masked)
            note_stage(rank, "init_process_group(%s)" % args.backend)
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    note_stage(rank, "device")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the backend has no CPU path")
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1 and args.backend == "nccl":
        note_stage(rank, "init_process_group(nccl)")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from kajo_amd import capi
    from kajo_amd.renderer import HipRenderer
    from kajo_amd.scene import Scene
    from kajo_amd.tiles import gather_to_root

    wl = args.workload if args.workload != "auto" else ("c2" if world == 1 else "c3")
    W, H, PASSES, ppl_default, wl_label = WORKLOADS[wl]
    ppl = args.passes_per_launch or ppl_default
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    scene = Scene.from_npz(z, "spheres_a169/", "spheres.json 16:9")

    def factory(w, h, **kw):
        kw.setdefault("strict", numerics == "strict")
        kw.setdefault("exact", numerics == "exact")
        return HipRenderer(scene, w, h, spp=SPP, depth_limit=DEPTH, seed=SEED, device=local_rank, passes_per_launch=ppl, **kw)

    # ONE stream carries the whole step -- render, gather (RCCL orders itself against the current stream on both
    # sides), compose, resolve -- so the next frame's kernel cannot touch a tile buffer that is still being sent.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    r = factory(W, H, tile_index=rank, tile_count=world)
    r.set_stream(stream.cuda_stream)
    ptr, nbytes = r.tile_buffer()
    mine = torch.as_tensor(DevicePtr(ptr, nbytes // 4), device="cuda")
    # the gather sends from a torch-allocated copy of the tile buffer (a device-to-device copy of 16.6 MB at N = 8, ~10 us):
    # memory the library allocated itself is foreign to the process group's allocator, and nothing about RCCL's handling
    # of it could be tried on the one-GPU boxes this was written on
    send = torch.empty_like(mine) if world > 1 else None
    gathered = None
    argb = torch.empty(W * H, dtype=torch.int32, device="cuda") if rank == 0 else None
    if world > 1 and rank == 0:
        gathered = torch.empty(world * (nbytes // 4), dtype=torch.float32, device="cuda")
    L = r._L
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    gather_ms, tail_ms = [], []

    def step(timed=False):
        r.render(PASSES)
        r.wait()
        if world > 1:
            ev[0].record()
            if args.backend == "nccl":
                if args.gather_direct:
                    gather_to_root(dist, mine, gathered, rank, world)
                else:
                    send.copy_(mine)
                    gather_to_root(dist, send, gathered, rank, world)
            else:  # rehearsal: gloo moves host tensors
                host = torch.empty(world * mine.numel()) if rank == 0 else None
                gather_to_root(dist, mine.cpu(), host, rank, world)
                if rank == 0:
                    gathered.copy_(host)
            ev[1].record()
            if rank == 0 and args.separate_compose:
                r.compose(gathered.data_ptr())
        if rank == 0:
            if world > 1 and not args.separate_compose:  # compose + resolve in one pass over the gathered tile buffers
                capi.check(L.kajo_hip_resolve_gathered_argb8_device(r._h, C.c_void_p(gathered.data_ptr()), C.c_void_p(argb.data_ptr())))
            else:
                capi.check(L.kajo_hip_resolve_argb8_device(r._h, C.c_void_p(argb.data_ptr())))
        ev[2].record()
        stream.synchronize()
        if timed and world > 1:
            gather_ms.append(ev[0].elapsed_time(ev[1]))
            tail_ms.append(ev[1].elapsed_time(ev[2]))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note_stage(rank, "first step (render + gather + resolve)" if world > 1 else "warmup")
    for k in range(max(args.warmup, 1 if world > 1 else 0)):  # (N > 1: at least one untimed step, so that the first gather is a stage of its own)
        step()
        if k == 0:
            note_stage(rank, "warmup")
    fence()
    note_stage(rank, "timed steps")
    c0 = r.counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt_local = time.perf_counter() - t0
    c1 = r.counters()
    dt = dt_local
    # The K timed steps are a fraction of a second of GPU time. The same loop for a few seconds more, OUTSIDE the timed region
    # and reported beside it: clocks, power and temperature have settled by then.
    sustained = None
    if world == 1 and args.sustain_seconds > 0 and args.steps > 0:
        per_step = dt_local / args.steps
        k = max(args.steps, int(args.sustain_seconds / max(per_step, 1e-6)))
        ts = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        sdt = time.perf_counter() - ts
        sustained = {"steps": k, "seconds": sdt, "ms_per_step": sdt / k * 1e3}
    per_rank = None
    if world > 1:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine_stats = torch.tensor([(c1["kernelMs"] - c0["kernelMs"]) / max(args.steps, 1), float(np.mean(gather_ms or [0.0])),
                                   (c1["paths"] - c0["paths"]) / max(args.steps, 1)], dtype=torch.float64, device=dev)
        allst = [torch.zeros_like(mine_stats) for _ in range(world)]
        dist.all_gather(allst, mine_stats)
        per_rank = [[float(v) for v in s.tolist()] for s in allst]

    n = int(math.sqrt(SPP))
    paths_per_step = W * H * n * n * PASSES
    value = paths_per_step * args.steps / dt / 1e6

    # ---- N > 1: the composed frame against ONE GPU rendering the same frame alone, bit for bit (2 steps) ---------
    check = None
    if world > 1 and not args.no_check:
        note_stage(rank, "bit-for-bit check against one GPU")
        r.reset()
        for _ in range(2):
            step()
        fence()
        if rank == 0:
            r.compose(gathered.data_ptr())  # (the float frame is only composed when somebody reads it)
            got = r.radiance()
            solo = factory(W, H)
            solo.render(PASSES).wait()  # warm: the first launch also records the launch order
            s0 = solo.counters()
            ts = time.perf_counter()
            solo.render(PASSES).wait()
            solo_dt = time.perf_counter() - ts
            s1 = solo.counters()
            want = solo.radiance()
            want_px = solo.argb8()
            solo.close()
            same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
            same_px = np.array_equal(argb.cpu().numpy().view(np.uint32).reshape(H, W), want_px)
            solo_value = (s1["paths"] - s0["paths"]) / solo_dt / 1e6
            check = {"frame_bit_identical_to_one_gpu": bool(same), "argb8_bit_identical_to_one_gpu": bool(same_px), "steps_compared": 2,
                     "one_gpu_same_frame_value": solo_value, "one_gpu_same_frame_ms": solo_dt * 1e3,
                     "speedup_vs_one_gpu_same_frame": value / solo_value,
                     "efficiency_vs_one_gpu_same_frame": value / solo_value / world}
        fence()

    note_stage(rank, "report")
    out = None
    if rank == 0:
        kernel = "kajo_render_" + numerics
        # work per path from the device counters of a separate, untimed frame (deterministic)
        rc = factory(W, H, tile_index=0, tile_count=world, counters=True)
        cc = rc.render(min(PASSES, 16)).counters()
        rc.close()
        trav = cc["traversals"] / cc["paths"]
        vert = cc["vertices"] / cc["paths"]
        fpp = flops_per_path(scene.n_planes, scene.n_spheres, trav, vert)
        launches = c1["launches"] - c0["launches"]
        kernel_ms = (c1["kernelMs"] - c0["kernelMs"]) / max(launches, 1)
        paths_per_launch = (c1["paths"] - c0["paths"]) / max(launches, 1)
        achieved = fpp * paths_per_launch / (kernel_ms * 1e-3) / 1e12
        owned_px = (c1["paths"] - c0["paths"]) / (n * n * PASSES * args.steps)
        hbm_gbs = 32.0 * owned_px / (kernel_ms * 1e-3) / 1e9  # float4 read + write per pixel per launch
        workload = ("data/spheres.json (parsed-scene fixture), %dx%d, %d spp = %d passes x S=32 (25 camera paths/px/pass), "
                    "depth 8, MIS on; %s" % (W, H, PASSES * SPP, PASSES, wl_label))
        pc = profile_counters(kernel, wl) if world == 1 else None
        roof = {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": pc["hbm_bytes_per_launch"] if pc else None,
                "traffic_source": (pc["source"] + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over this command, on kernels with this source hash)") if pc
                                  else "no counters file for these kernels (tools/profile_round.sh <round> writes profiles/r<NN>_counters.json; the newest one collected on this source hash is used)",
                "kernel": kernel, "kernel_ms_per_launch": kernel_ms, "launches_per_step": launches / args.steps,
                "passes_per_launch": ppl,
                "flops_per_path": fpp, "traversals_per_path": trav, "vertices_per_path": vert,
                "lane_efficiency": cc["traversals"] / max(cc["laneSlots"], 1),
                "hbm": {"achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
                        "algorithmic_bytes_per_launch": 32.0 * owned_px}}
        if pc:
            # FP32 work the launch really executed: the PMC FLOP count x the share of lanes that were active
            roof["executed_flops"] = {"per_launch": pc["executed_fp32_flops_per_launch"],
                                      "tflops": pc["executed_fp32_flops_per_launch"] / (kernel_ms * 1e-3) / 1e12,
                                      "valu_lane_utilisation": pc["valu_lane_utilisation"], "source": pc["source"]}
        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if world > 1 else None,  # one GPU: nothing scales
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "numerics": numerics,
                       "numerics_note": {"exact": "every path takes the CPU oracle's decisions (hits, draws, terminations); only products that scale radiance use the fast forms",
                                         "strict": "bit-identical to the CPU oracle", "fast": "hardware transcendentals and contraction everywhere"}[numerics],
                       "tiles": "64x16 round-robin over ranks", "paths_per_step": paths_per_step,
                       "world_size_seen_by_backend": (dist.get_world_size() if world > 1 else 1),
                       "backend": (args.backend if world > 1 else None),
                       # which library was timed (KAJO_HIP_LIB can point the binding at a diagnostic twin; never a CPU path)
                       "library": os.path.relpath(capi.LIB_PATH, ROOT), "library_version": capi.lib().kajo_hip_version().decode(),
                       "kernel_source_hash": kernel_source_hash(),
                       # launch-shaping knobs are compile-time constants of libkajo_hip.so (no getenv in it); a tools' twin
                       # selected with KAJO_HIP_LIB reads them from the environment, so whatever was set is recorded
                       "tuning_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("KAJO_") and k != "KAJO_HIP_LIB"}},
            "roofline": roof,
            "in_kernel_value": paths_per_launch / (kernel_ms * 1e-3) / 1e6 * world,
            "mtraversals_per_s": value * trav,
            "nominal_rays_x_spp_x_bounces_per_s_M": W * H * n * n * PASSES * DEPTH * args.steps / dt / 1e6,
        }
        if sustained:
            sustained["value"] = paths_per_step * sustained["steps"] / sustained["seconds"] / 1e6
            sustained["unit"] = "Msamples/s"
            out["sustained"] = sustained
        if per_rank:
            km = [p[0] for p in per_rank]
            out["multi_gpu"] = {"kernel_ms_per_step_by_rank": km, "kernel_ms_imbalance": max(km) / max(min(km), 1e-9),
                                "gather_ms_by_rank": [p[1] for p in per_rank],
                                "compose_resolve_ms_rank0": float(np.mean(tail_ms or [0.0])),
                                "compose_resolve": "separate kernels (whole float frame written and read back)" if args.separate_compose
                                                   else "one kernel, straight from the gathered tile buffers",
                                "paths_per_step_by_rank": [p[2] for p in per_rank],
                                "gather_bytes_per_peer": nbytes}
            if check:
                out["multi_gpu"].update(check)
    r.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the parity legs first, the timed CPU baseline last (its libraries are built with the reference's fast-math flags)
        if sched_leg:
            # hip::Scheduler::run() on the same frame with the same numerics build, next to ms_per_step: the difference is the 8 MB
            # host read-back and the synchronous refresh (PCIe-inclusive rate)
            out["scheduler_run"] = sched_leg
        # Every numerics build on the timed frame against the CPU oracle's rendering of THAT frame (all of its pixels).
        want, oracle_s = oracle_frame(scene, W, H, PASSES)
        fpp = out["roofline"]["flops_per_path"]
        modes = {}
        for m in NUMERICS:
            leg, got = mode_leg(scene, W, H, PASSES, ppl, local_rank, m, fpp)
            leg["parity"] = parity_leg(want, oracle_s, got, W, H, PASSES)
            modes[m] = leg
            del got
        del want
        out["parity"] = modes[numerics]["parity"]
        out["cpu_baseline"] = cpu_baseline(scene, W, H)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        for m, leg in modes.items():
            leg["speedup_vs_cpu_baseline"] = leg["value"] / out["cpu_baseline"]["value"]
            if m != numerics:
                out[m + "_mode"] = leg
        meeting = [m for m in NUMERICS if modes[m]["parity"]["meets_north_star_rmse"]]
        out["modes_meeting_north_star_rmse_1e-4"] = meeting
        # BASELINE.json's two targets at once (>= 100 x the CPU backend AND per-pixel RMSE < 1e-4 on the timed frame): the fastest
        # build whose parity leg meets the RMSE figure. Since round 5 that build is the one `value` is measured on (unless --fast /
        # --strict chose another): `headline_is_north_star_mode`.
        if meeting:
            best = max(meeting, key=lambda m: modes[m]["value"])
            b = modes[best]
            out["north_star_mode"] = {"numerics": best, "value": out["value"] if best == numerics else b["value"], "unit": "Msamples/s",
                                      "ms_per_step": out["ms_per_step"] if best == numerics else b["ms_per_step"],
                                      "roofline_frac": out["roofline"]["frac"] if best == numerics else b["roofline"]["frac"],
                                      "rmse": b["parity"]["rmse_clamped01"], "bit_identical_px": b["parity"]["bit_identical_px"],
                                      "px": b["parity"]["px"], "px_off_by_more_than_1e-3": b["parity"]["px_off_by_more_than_1e-3"],
                                      "speedup_vs_cpu_baseline": (out["value"] if best == numerics else b["value"]) / out["cpu_baseline"]["value"]}
            out["headline_is_north_star_mode"] = bool(best == numerics)
        else:
            out["north_star_mode"] = None
            out["headline_is_north_star_mode"] = False
    if rank == 0:
        if world > 1:
            out["config"]["gather_direct"] = bool(args.gather_direct)
        STATE["printed"] = True
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
