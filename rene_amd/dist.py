"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The render path shards by construction -- a sample depends only on (x, y, frame seed)
(rene-shader/src/lib.rs:174-176) -- so ranks never exchange data while rendering.  The single
exchange step is the sum of the per-rank framebuffers at the end.  Two ways
to cut the job: by frames (`frame_block`: every rank renders the whole image for a block of frames;
the exchange is a reduce of the whole image; the sum differs from a single-GPU render only in fp32
summation order) or by tiles (`gather_owned_tiles`: every pixel has exactly one owner, the exchange
moves 1/N of the image per rank and the result is bit-identical to a single-GPU render, but a rank's
launches shrink with N).  bench.py uses frames: at 1024^2 a tile shard of 8 has fewer pixels than the
GPU has lanes.
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


def _tile_view(fb):
    """[3][H][W][4] -> [tiles_y][tiles_x][3][32][32][4] view (H, W multiples of the tile size)."""
    L, H, W, Cc = fb.shape
    t = abi.TILE_SIZE
    return fb.view(L, H // t, t, W // t, t, Cc).permute(1, 3, 0, 2, 4, 5)


def gather_owned_tiles(fb, rank: int, world: int, dst: int = 0):
    """The exchange step, tile-sharded form: every rank sends ONLY the 32x32 tiles it owns to `dst`
    (1/world of the image per rank, all 7 xGMI links of `dst` in parallel) instead of summing whole
    images -- each pixel has exactly one owner, so placing tiles equals the reduce bit for bit while
    moving world x fewer bytes.  Falls back to `reduce_framebuffer` when the image is not a whole
    number of tiles or the tile count does not divide evenly."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world <= 1:
        return fb
    L, H, W, Cc = fb.shape
    t = abi.TILE_SIZE
    n_tiles = (H // t) * (W // t) if H % t == 0 and W % t == 0 else 0
    if n_tiles == 0 or n_tiles % world != 0:
        return reduce_framebuffer(fb, dst)
    tv = _tile_view(fb).reshape(n_tiles, L, t, t, Cc)          # copy (permuted view -> contiguous)
    idx = torch.arange(rank, n_tiles, world, device=fb.device)
    mine = tv.index_select(0, idx).contiguous()                 # [n_tiles/world][3][32][32][4]
    if mine.is_cuda and dist.get_backend() == "gloo":           # rehearsal backend: gloo gathers CPU tensors only
        mine = mine.cpu()
    if rank == dst:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, gather_list=parts, dst=dst)
        full = torch.empty_like(tv)
        for r, part in enumerate(parts):
            full.index_copy_(0, torch.arange(r, n_tiles, world, device=fb.device), part.to(fb.device))
        fb.copy_(full.view(H // t, W // t, L, t, t, Cc).permute(2, 0, 3, 1, 4, 5).reshape(L, H, W, Cc))
    else:
        dist.gather(mine, gather_list=None, dst=dst)
    return fb


def frame_block(rank: int, world: int, total_frames: int) -> tuple[int, int]:
    """Frame sharding in contiguous blocks: rank r renders frames [r * T / world, (r + 1) * T / world)
    of the whole image (sizes differ by at most one frame).  Samples are independent given their frame
    seed (lib.rs:174-176, 512-514), so the ranks never talk while rendering; the exchange step is the
    sum of the partial images (`reduce_framebuffer`).  Blocks, not round robin: a rank then renders
    its share in launches as large as a single GPU's (a launch has a fixed cost of ~0.35 ms: the
    longest single path of its last frames)."""
    world = max(1, world)
    return rank * total_frames // world, (rank + 1) * total_frames // world


def reduce_framebuffer(fb, dst: int = 0):
    """Sum the per-rank accumulation images onto rank `dst` (ncclReduce over xGMI on GPUs)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if fb.is_cuda and dist.get_backend() == "gloo":  # rehearsal backend: reduce through the host
            host = fb.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            fb.copy_(host)
        else:
            dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
    return fb
