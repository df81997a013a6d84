"""-m gpu: the N > 1 path on the real HIP kernels -- two fresh child processes share the box's one GPU (RCCL refuses two
ranks on one device, so `gloo` carries the exchange; the exchange inside the library is covered by test_gpu_comm.py),
each renders its shard through the C ABI into a device buffer it owns, and rank 0's image after the exchange equals the
one-rank image: bit for bit for the tile cut, up to fp32 summation order for the frame-block cut."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from test_dist_gloo import _free_port

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, cut):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from rene_amd import abi, api, dist as rdist, scenes
    rdist.init_process_group("gloo")
    s = scenes.cornell_box(160, 96)  # ragged against the 32x32 tiles: 5 x 3 tiles over 2 ranks
    fb = torch.zeros((3, 96, 160, 4), dtype=torch.float32, device="cuda:0")
    frames = 10
    if cut == "tiles":
        r = api.Renderer(s, device=0, shard_mode=abi.SHARD_TILES, shard_rank=rank, shard_count=world, framebuffer_ptr=fb.data_ptr())
        for f0 in range(0, frames, 4):
            r.render(f0, min(4, frames - f0))
    else:
        r = api.Renderer(s, device=0, framebuffer_ptr=fb.data_ptr())
        lo, hi = rdist.frame_block(rank, world, frames)
        for f0 in range(lo, hi, 2):
            r.render(f0, min(2, hi - f0))
    r.sync()
    if cut == "tiles":
        rdist.gather_owned_tiles(fb, rank, world, dst=0)
    else:
        rdist.reduce_framebuffer(fb, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        q.put(fb[..., :3].cpu().numpy())
    r.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("cut,world", [("tiles", 2), ("frames", 2), ("tiles", 4), ("frames", 4)])
def test_ranks_on_the_hip_path_equal_one(cut, world):
    """world 2 and 4 (a GPU box admits six processes on its card; world 8 is rehearsed on the CPU, test_dist_gloo.py): the
    strong-scaling job of bench.py --gpus N -- the SAME image and sample count whatever N -- cut by tiles or by frame blocks."""
    import torch.multiprocessing as mp
    from rene_amd import api, scenes
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, cut)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    with api.Renderer(scenes.cornell_box(160, 96)) as r:
        r.render(0, 10)
        want = np.stack([r.download(l) for l in range(3)])
    if cut == "tiles":
        assert np.array_equal(got, want)  # every pixel has one owner
    else:
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)  # fp32 summation order
