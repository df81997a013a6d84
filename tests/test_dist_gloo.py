"""world_size-2 `gloo` rehearsal of the multi-GPU path on CPU: tile sharding + the one framebuffer
reduce.  The per-rank images come from the oracle here (the product path
needs a GPU); what is exercised is rene_amd.dist and the shard definition it shares with the
kernel."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from rene_amd import abi, dist as rdist, scenes
    from oracle import oracle
    rdist.init_process_group("gloo")
    o = oracle.Oracle(scenes.cornell_box(128, 128 if world == 8 else 96))  # 4 x 3 tiles (12 % 2 == 12 % 3 == 0); world 8: 4 x 4
    o.render(0, 2, threads=1, shard_mode=abi.SHARD_TILES, shard_rank=rank, shard_count=world)
    fb = torch.from_numpy(np.stack([o.download(l, 4) for l in range(3)]))
    rdist.gather_owned_tiles(fb, rank, world, dst=0)
    dist.barrier()
    if rank == 0:
        q.put(fb.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_gather_owned_tiles_equals_single(world):
    import torch.multiprocessing as mp
    from rene_amd import scenes
    from oracle import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = oracle.Oracle(scenes.cornell_box(128, 128 if world == 8 else 96))
    o.render(0, 2, threads=1)
    want = np.stack([o.download(l, 4) for l in range(3)])
    assert np.array_equal(got, want)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from rene_amd import abi, dist as rdist, scenes
    from oracle import oracle
    rdist.init_process_group("gloo")
    o = oracle.Oracle(scenes.cornell_box(96, 80))
    o.render(0, 3, threads=1, shard_mode=abi.SHARD_TILES, shard_rank=rank, shard_count=world)
    fb = torch.from_numpy(np.stack([o.download(l, 4) for l in range(3)]))
    rdist.reduce_framebuffer(fb, dst=0)
    dist.barrier()
    if rank == 0:
        q.put(fb.numpy())
    dist.destroy_process_group()


def test_two_rank_tile_shard_reduce_equals_single():
    import torch.multiprocessing as mp
    from rene_amd import scenes
    from oracle import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = oracle.Oracle(scenes.cornell_box(96, 80))
    o.render(0, 3, threads=1)
    want = np.stack([o.download(l, 4) for l in range(3)])
    assert np.array_equal(got, want)  # each pixel has one owner: the reduce adds exact zeros


def test_tile_owner_map_partitions_image():
    from rene_amd.dist import tile_owner_map
    for w, h, n in ((96, 80, 3), (1024, 1024, 8), (33, 31, 2), (1920, 1080, 8)):
        own = tile_owner_map(w, h, n)
        assert own.shape == (h, w) and own.min() == 0 and own.max() == min(n, ((w + 31) // 32) * ((h + 31) // 32)) - 1
        counts = np.bincount(own.ravel(), minlength=n)
        assert counts.sum() == w * h
        if w * h >= 1024 * 1024:
            assert counts.max() / counts.min() < 1.15  # interleaved tiles balance the pixel load


def _frame_worker(rank, world, port, q, total):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from rene_amd import dist as rdist, scenes
    from oracle import oracle
    rdist.init_process_group("gloo")
    o = oracle.Oracle(scenes.cornell_box(64, 48))
    lo, hi = rdist.frame_block(rank, world, total)
    for f0 in range(lo, hi, 2):  # launches of at most 2 frames, like bench.py's frames_per_step
        o.render(f0, min(2, hi - f0), threads=1)
    fb = torch.from_numpy(np.stack([o.download(l, 4) for l in range(3)]))
    rdist.reduce_framebuffer(fb, dst=0)
    dist.barrier()
    if rank == 0:
        q.put(fb.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 7), (3, 8), (8, 16)])
def test_frame_block_reduce_equals_single_up_to_summation_order(world, total):
    """bench.py's N > 1 cut (strong scaling: the job's frames dealt out in contiguous blocks, 8 ranks included) + one reduce of
    the partial images."""
    import torch.multiprocessing as mp
    from rene_amd import scenes
    from rene_amd.dist import frame_block
    from oracle import oracle
    blocks = [frame_block(r, world, total) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == total
    assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
    assert max(b - a for a, b in blocks) - min(b - a for a, b in blocks) <= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_frame_worker, args=(r, world, port, q, total)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o = oracle.Oracle(scenes.cornell_box(64, 48))
    o.render(0, total, threads=1)
    want = np.stack([o.download(l, 4) for l in range(3)])
    # every sample is the same; only the order of the fp32 additions differs (per-rank partial sums)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    assert not np.array_equal(got[0], np.zeros_like(got[0]))
