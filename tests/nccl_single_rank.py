"""Helper of test_multi_rank_gpu.py: the RCCL calls bench.py makes for N > 1 (init with device_id, gather of the tile
buffer wrapped through __cuda_array_interface__, MAX all-reduce of the step time, barrier), with a world of ONE rank --
all a 1-GPU box can host. Prints 'nccl ok' on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from kajo_amd.renderer import HipRenderer  # noqa: E402
from kajo_amd.scene import Scene  # noqa: E402
from kajo_amd.tiles import gather_to_root  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
scene = Scene.from_npz(z, "spheres_a169/", "spheres")
r = HipRenderer(scene, 640, 360, tile_index=0, tile_count=1)
ptr, nbytes = r.tile_buffer()
mine = torch.as_tensor(bench.DevicePtr(ptr, nbytes // 4), device="cuda")
gathered = torch.empty(mine.numel(), dtype=torch.float32, device="cuda")
r.render(2).wait()
gather_to_root(dist, mine, gathered, 0, 1)                  # world 1: returns without a collective
send = torch.empty_like(mine)
send.copy_(mine)                                            # bench.py gathers from a torch-allocated copy of the tile buffer
dist.gather(send, list(gathered.chunk(1)), dst=0)          # the collective itself, one-rank world
torch.cuda.current_stream().synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
ok = bool(torch.equal(gathered, mine)) and float(t.item()) == 1.5 and float(mine.abs().sum()) > 0
r.close()
dist.destroy_process_group()
print("nccl ok" if ok else "nccl MISMATCH")
