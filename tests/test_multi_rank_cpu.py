"""The N > 1 path on the CPU: two gloo ranks deal the tiles, fill their compact buffers, gather to
rank 0 and compose -- every pixel must come back from the rank and slot the device code would use."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, tile, out_path):
    sys.path.insert(0, ROOT)
    from kajo_amd.tiles import TileLayout, gather_to_root
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lay = TileLayout(W, H, world, tile)
    # what a rank's render kernel leaves in its buffer: float4 per slot; here the pixel's own identity
    local = torch.full((lay.slots_per_owner * 4,), -1.0)
    ys, xs = np.mgrid[0:H, 0:W]
    owner, slot = lay.owner_and_slot(xs, ys)
    mine = owner == rank
    buf = local.view(-1, 4).numpy()
    buf[slot[mine], 0] = xs[mine]
    buf[slot[mine], 1] = ys[mine]
    buf[slot[mine], 2] = rank
    buf[slot[mine], 3] = 1.0
    gathered = torch.empty(world * local.numel()) if rank == 0 else None
    gather_to_root(dist, local, gathered, rank, world)
    if rank == 0:
        frame = lay.compose(gathered.view(world, lay.slots_per_owner, 4).numpy())
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H,tile,world", [(200, 70, (32, 8), 2), (1920 // 8, 1080 // 8, (64, 16), 2), (97, 33, (64, 16), 3)])
def test_gloo_gather_and_compose(tmp_path, W, H, tile, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, tile, out), nprocs=world, join=True)
    frame = np.load(out)
    ys, xs = np.mgrid[0:H, 0:W]
    assert np.array_equal(frame[..., 0], xs) and np.array_equal(frame[..., 1], ys)
    assert (frame[..., 3] == 1.0).all()
    # round-robin dealing: tile t belongs to rank t % world
    tiles_x = -(-W // tile[0])
    want_owner = ((ys // tile[1]) * tiles_x + xs // tile[0]) % world
    assert np.array_equal(frame[..., 2], want_owner)


def test_layout_matches_library_sizes():
    """slots_per_owner must equal what kajo_hip_create allocates (capi.cpp) -- checked through the
    C ABI on the GPU in tests/test_hip_parity.py::test_tiling_is_bit_invariant; here the arithmetic."""
    from kajo_amd.tiles import TileLayout
    lay = TileLayout(1920, 1080, 8)
    assert lay.tiles_x == 30 and lay.tiles_y == 68 and lay.n_tiles == 2040
    assert lay.tiles_per_owner == 255 and lay.slots_per_owner == 255 * 1024
    assert sum(lay.owned_pixels(r) for r in range(8)) == 1920 * 1080
    # balance: no rank owns more than 1 % above the mean
    px = [lay.owned_pixels(r) for r in range(8)]
    assert max(px) <= 1.01 * (1920 * 1080 / 8)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start the two ranks itself (a child torch.distributed.run) and
    pass their exit code on. Without a GPU every rank stops at "needs a GPU" (the backend has no CPU path): seeing that
    message from a process that has RANK set proves the launch happened; the non-zero code proves it is propagated."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_multi_rank_gpu.py")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert "bench.py needs a GPU" in p.stderr and "--nproc-per-node" not in p.stdout
    assert p.stderr.count("bench.py needs a GPU") >= 2 or "local_rank: 1" in p.stderr or "rank      : 1" in p.stderr


def _bench(*extra):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", *extra],
                          capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_a_failed_multi_rank_run_still_prints_one_parseable_line():
    """First contact with N > 1 hardware must be diagnosable (the 8-GPU run is the driver's, nobody watches it): when a rank fails --
    here: a backend name torch.distributed does not know -- the run still ends in ONE JSON line with the contract's keys, value null,
    the stage each rank had reached and the error; the exit code is not zero."""
    import json
    p = _bench("--backend", "no_such_backend")
    assert p.returncode != 0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["value"] is None and out["n_gpus"] == 2 and out["metric"] == "Msamples/s" and out["config"]["backend"] == "no_such_backend"
    assert "init_process_group(no_such_backend)" in json.dumps(out["stage"]), out
    assert out["error"] and ("no_such_backend" in out["error"].lower() or "no_such_backend" in (out.get("stderr_tail") or "").lower()), out
    # every rank said where it was
    assert p.stderr.count("KAJO_BENCH_STAGE") >= 2 and "KAJO_BENCH_ERROR" in p.stderr


def test_a_run_without_gpus_names_the_stage():
    """... and the same when the ranks come up but find no device (this container): stage 'device', the message of the product's refusal
    to run on a CPU."""
    import json
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = _bench("--backend", "gloo", "--gather-direct")
    assert p.returncode != 0
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["value"] is None and "device" in json.dumps(out["stage"]) and "needs a GPU" in out["error"] and out["config"]["gather_direct"] is True
