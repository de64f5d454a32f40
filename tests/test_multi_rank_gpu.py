"""The RCCL side of the N > 1 path as far as one GPU can host it: a one-rank nccl process group doing exactly the
calls bench.py makes (tests/nccl_single_rank.py), in a child process so that the group does not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_nccl_calls_of_the_bench_with_one_rank():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(here, "nccl_single_rank.py")], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "nccl ok" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])


def test_bench_two_ranks_on_one_gpu_gloo_rehearsal():
    """`python bench.py --gpus 2` starts its own two ranks (child process, torch.distributed.run) and renders BASELINE
    configs[2] -- 3840 x 2160, 64 passes -- strong-scaled; with --backend gloo both ranks share the one GPU of this box
    and the gather is staged through host memory, everything else (tile dealing, one stream per rank carrying render /
    gather / compose / resolve, max-over-ranks timing, per-rank statistics) is the code the RCCL run executes. The
    composed two-rank frame must equal a one-GPU frame bit for bit after two accumulated steps."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and "3840x2160" in out["config"]["workload"]
    assert out["config"]["world_size_seen_by_backend"] == 2
    mg = out["multi_gpu"]
    assert mg["frame_bit_identical_to_one_gpu"] is True and mg["argb8_bit_identical_to_one_gpu"] is True
    assert "one kernel" in mg["compose_resolve"]  # rank 0 resolves straight from the gathered tile buffers
    assert len(mg["kernel_ms_per_step_by_rank"]) == 2 and min(mg["kernel_ms_per_step_by_rank"]) > 0
    assert sum(mg["paths_per_step_by_rank"]) == out["config"]["paths_per_step"]
