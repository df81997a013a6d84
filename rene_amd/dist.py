"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The render path shards by construction -- a pixel's result depends only on (x, y, frame seed)
(rene-shader/src/lib.rs:174-176) -- so ranks never exchange data while rendering.  The single
exchange step is the sum of the per-rank framebuffers at the end: every
pixel is owned by exactly one rank under tile sharding, so the reduce adds zeros to the owner's
value and the result is bit-identical to a single-GPU render.
"""
from __future__ import annotations

import os

import numpy as np

from . import abi


def env_rank_world() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def tile_owner_map(xres: int, yres: int, shard_count: int) -> np.ndarray:
    """(yres, xres) array of the owning rank of each *image* pixel (row 0 = top): 32x32 tiles,
    row-major, tile t -> rank t % shard_count.  Same formula as kernels.hip and the oracle."""
    tiles_x = (xres + abi.TILE_SIZE - 1) // abi.TILE_SIZE
    ty = np.arange(yres)[:, None] // abi.TILE_SIZE
    tx = np.arange(xres)[None, :] // abi.TILE_SIZE
    return ((ty * tiles_x + tx) % max(1, shard_count)).astype(np.int32)


def init_process_group(backend: str | None = None):
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world == 1 or dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)


def reduce_framebuffer(fb, dst: int = 0):
    """Sum the per-rank accumulation images onto rank `dst` (ncclReduce over xGMI on GPUs)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb
