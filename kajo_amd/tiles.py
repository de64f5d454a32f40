"""Tile dealing for multi-GPU frames: which rank owns which pixel and where it lives in that rank's
compact tile buffer. Host-side mirror of kajoTileSlot (kajo_amd/csrc/render_args.h) and of the
sizes kajo_hip_create derives (kajo_amd/csrc/capi.cpp); used to drive the per-frame gather and to
check it on the CPU (gloo) without a GPU.

The reference splits the image into one contiguous row band per core (renderer/cpu/Scheduler.cpp:32-42,
measured 4.0x on 8 threads: bands are load-imbalanced); here fixed-size tiles are dealt round-robin
so that expensive regions spread evenly over the GPUs.
"""
from __future__ import annotations

import numpy as np


class TileLayout:
    def __init__(self, width: int, height: int, world: int, tile=(64, 16)):
        self.W, self.H, self.world = int(width), int(height), int(world)
        self.tw, self.th = tile
        assert self.tw % 8 == 0 and self.th % 8 == 0 and (self.tw * self.th) % 256 == 0
        self.tiles_x = -(-self.W // self.tw)
        self.tiles_y = -(-self.H // self.th)
        self.n_tiles = self.tiles_x * self.tiles_y
        self.tiles_per_owner = -(-self.n_tiles // self.world)
        self.slots_per_owner = self.tiles_per_owner * self.tw * self.th  # float4 slots, padded: equal on every rank

    def owner_and_slot(self, x, y):
        """Vectorised kajoTileSlot: arrays of pixel coordinates -> (owner rank, slot in its buffer)."""
        x = np.asarray(x)
        y = np.asarray(y)
        tx, ty = x // self.tw, y // self.th
        tile = ty * self.tiles_x + tx
        ix, iy = x - tx * self.tw, y - ty * self.th
        waves_per_tile = (self.tw >> 3) * (self.th >> 3)
        wb = (iy >> 3) * (self.tw >> 3) + (ix >> 3)
        lane = ((iy & 7) << 3) | (ix & 7)
        return tile % self.world, ((tile // self.world) * waves_per_tile + wb) * 64 + lane

    def compose(self, gathered: np.ndarray) -> np.ndarray:
        """gathered: (world, slots_per_owner, C) -> (H, W, C) frame; what kajo_compose does on the device."""
        ys, xs = np.mgrid[0:self.H, 0:self.W]
        owner, slot = self.owner_and_slot(xs, ys)
        return gathered[owner, slot]

    def owned_pixels(self, rank: int) -> int:
        ys, xs = np.mgrid[0:self.H, 0:self.W]
        owner, _ = self.owner_and_slot(xs, ys)
        return int((owner == rank).sum())


def gather_to_root(dist, local, gathered, rank: int, world: int):
    """The one collective on the data path: every rank's tile buffer -> rank 0, in rank order
    (RCCL over xGMI with the nccl backend; gloo on the CPU in tests). `local` is a 1-D tensor of
    the rank's whole (padded) tile buffer; `gathered` a 1-D tensor of world * local.numel() on rank 0."""
    if world == 1:
        return
    dist.gather(local, list(gathered.chunk(world)) if rank == 0 else None, dst=0)
