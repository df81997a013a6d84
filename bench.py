#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path (BASELINE.json: Mrays/s and ms/frame at
fixed spp; achieved GB/s vs the HBM roofline).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload: BASELINE.json configs[1], "Cornell-box 1024x1024 @ 1024 spp, 1x MI355X" (synthetic
Cornell-box-shaped scene built in code, rene_amd/scenes.py).  A *step* is one launch of the
persistent render kernel over every pixel this rank owns for `frames_per_step = ceil(1024 / K)`
frames, so the K timed steps always render the full 1024 spp (a little more if K does not divide
1024).  Default K = 4, i.e. 256 frames per launch: a launch ends with its longest paths and a partly idle chip, so
fewer, longer launches are the efficient way to ask for 1024 spp (measured, Grays/s at K = 16 / 8 / 4 / 2:
82.6 / 86.8 / 91.2 / 88.0 -- at K = 2 there is no third launch to hide the second one's tail).  Consecutive
launches overlap on two streams (RENE_FLAG_OVERLAP) and the work-item granularity is picked by rene_tune in the
untimed part; neither changes a bit of the image.  Inputs (scene tables, BVH, frame seeds) are resident in HBM
before the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  Strong scaling: the job is still
K steps = 1024 spp of the whole image; its K * frames_per_step frames are dealt to the ranks in N
contiguous blocks (rene_amd.dist.frame_block), each rank renders its block in launches of at most
frames_per_step frames (two when the block would fit one, so that they overlap), ranks never talk while rendering, and the one exchange step -- an RCCL reduce
(sum) of the [3][H][W][4] f32 partial images onto rank 0 -- is inside the timed region.
RENE_BENCH_SHARD=tiles selects the other cut (32x32 tiles round-robin + a gather of owned tiles: bit-
identical to one GPU, but a rank's launches shrink with N).  value = rays of all ranks / max-over-ranks time.

The JSON line also carries
  roofline     -- dominant kernel (render_kernel): algorithmic bytes per launch (SURVEY.md 8d cache-less
                  model, evaluated from the node/primitive counters of an untimed counting pass over
                  this build's BVH4) / average launch duration from HIP events recorded on the
                  kernel's stream inside librene_hip.so, against the 8 TB/s HBM3E peak; `traffic` =
                  PMC-measured HBM bytes per launch when profiles/ holds them for this config;
  cpu_baseline -- the CPU oracle (a port of rene's integrator; the reference itself has no CPU
                  path and cannot be built here) timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH = HEIGHT = 1024
TARGET_SPP = 1024
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_record(frames_per_step: int, n_gpus: int):
    """The rocprofv3 --pmc summary committed under profiles/ for this launch shape (None if absent or
    collected for a different shape): HBM bytes per render_kernel launch (FETCH_SIZE doubled per the
    gfx950 correction, WRITE_SIZE as is) and the SQ_* instruction counts of the same launch."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return None
    if rec.get("frames_per_step") != frames_per_step or rec.get("n_gpus", 1) != n_gpus:
        return None
    return rec


def pmc_traffic(frames_per_step: int, n_gpus: int):
    rec = pmc_record(frames_per_step, n_gpus)
    return rec.get("hbm_bytes_per_launch") if rec else None


def valu_issue(frames_per_step: int, n_gpus: int, launch_ms: float, cus: int, clock_ghz: float):
    """What actually bounds the kernel: the share of VALU issue cycles in use.  A wave64 VALU
    instruction occupies its SIMD16 for 4 cycles; there are 4 SIMDs per CU.  Instruction count from the
    committed PMC pass (SQ_INSTS_VALU per launch), duration measured live."""
    rec = pmc_record(frames_per_step, n_gpus)
    if not rec or not rec.get("valu_wave_insts_per_launch") or launch_ms <= 0:
        return None
    insts = rec["valu_wave_insts_per_launch"]
    avail = cus * 4 * launch_ms * 1e-3 * clock_ghz * 1e9
    return {"valu_wave_insts_per_launch": insts, "issue_cycles_frac": insts * 4.0 / avail,
            "lane_ops_per_ray": None, "clock_ghz": clock_ghz, "cus": cus,
            "source": f"profiles/{rec.get('tag', '')}_pmc.txt (SQ_INSTS_VALU) / live launch period",
            "model": "4 issue cycles per wave64 VALU instruction at the nominal clock; a fraction above 1 says the hardware "
                     "retires some of them faster (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.03 quad-cycles per instruction "
                     "over a launch that is not busy throughout) -- the issue slots are full either way"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-spp", type=int, default=128, help="frames of the CPU-oracle baseline sample (~10 s on 16 threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from rene_amd import abi, api, dist as rdist, scenes

    rank, world, local = rdist.env_rank_world()
    if world != max(1, args.gpus) and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # Normally one rank per GPU over RCCL.  RENE_DIST_BACKEND=gloo lets several ranks share one GPU
    # (a rehearsal of the N > 1 code path on a 1-GPU box; RCCL refuses two ranks on one device).
    backend = os.environ.get("RENE_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        rdist.init_process_group(backend)
        import torch.distributed as dist

    K, Wm = max(1, args.steps), max(0, args.warmup)
    F = math.ceil(TARGET_SPP / K)
    scene = scenes.cornell_box(WIDTH, HEIGHT)
    packed = scene.to_desc()

    by_tiles = os.environ.get("RENE_BENCH_SHARD", "frames") == "tiles"
    t_rank, t_world = (rank, world) if by_tiles else (0, 1)
    fb = torch.zeros((3, HEIGHT, WIDTH, 4), dtype=torch.float32, device=f"cuda:{local}")
    # consecutive launches overlap on two streams (the next one fills the chip while the last paths of the previous
    # one finish; same image bit for bit) unless RENE_BENCH_OVERLAP=0
    overlap = abi.FLAG_OVERLAP if os.environ.get("RENE_BENCH_OVERLAP", "1") != "0" else 0
    r = api.Renderer(packed, device=local, flags=overlap, shard_mode=abi.SHARD_TILES, shard_rank=t_rank, shard_count=t_world,
                     framebuffer_ptr=fb.data_ptr())
    # this rank's launches: (first_frame, n_frames)
    if by_tiles:
        launches = [(k * F, F) for k in range(K)]
    else:
        lo, hi = rdist.frame_block(rank, world, K * F)
        launches = [(f0, min(F, hi - f0)) for f0 in range(lo, hi, F)]
        if len(launches) == 1 and launches[0][1] >= 2:
            # a block that fits one launch is rendered as two: the second starts while the first finishes its longest
            # paths (measured on one GPU, 128 frames: 8.57 ms as one launch, 8.08 ms as two)
            f0, nf = launches[0]
            launches = [(f0, nf // 2), (f0 + nf // 2, nf - nf // 2)]

    def exchange():
        if world > 1:
            if by_tiles:
                rdist.gather_owned_tiles(fb, rank, world, dst=0)
            else:
                rdist.reduce_framebuffer(fb, dst=0)

    # ---- untimed: algorithmic bytes per ray from the kernel's own counters (same scene, seeds) ----
    # (counted over this build's BVH4 -- RENE_FLAG_FORCE_BVH -- so that the figure does not depend on
    # which intersection back end renders the scene: the small-scene item loop tests every item)
    cf = min(F, 8)
    with api.Renderer(packed, device=local, flags=abi.FLAG_COUNTERS | abi.FLAG_FORCE_BVH, shard_mode=abi.SHARD_TILES,
                      shard_rank=t_rank, shard_count=t_world) as rc:
        rc.render(0, cf)
        cst = rc.stats()
    bytes_per_ray = abi.algorithmic_bytes(cst) / max(1, cst.rays)

    # ---- untimed: work-item granularity for launches of F frames (rene_tune; no bit of the image depends on it) ----
    r.tune(max(nf for _, nf in launches))

    # ---- warmup (kernel + the collective: RCCL sets its rings up lazily), then a clean image ----
    for k in range(Wm):
        r.render(k * F, F)
    r.sync()
    exchange()
    torch.cuda.synchronize()
    r.reset()
    fb.zero_()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for f0, nf in launches:
        r.render(f0, nf)
    r.sync()
    render_s = time.perf_counter() - t0
    exchange()  # the one exchange step (RCCL over xGMI)
    fence()
    elapsed = time.perf_counter() - t0

    st = r.stats()
    rays = torch.tensor([float(st.rays)], dtype=torch.float64, device=f"cuda:{local}")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays, elapsed = float(rays.item()), float(tmax.item())

    if rank == 0:
        img = fb[0, :, :, :3]
        assert bool(torch.isfinite(img).all()) and float(img.mean()) > 0.0, "framebuffer is empty or non-finite"
        spp = K * F
        launch_ms = st.kernel_ms / max(1, st.launches)  # HIP events around each launch, on the launch's stream
        # with RENE_FLAG_OVERLAP two launches are in flight at a time: one launch *lasts* about two launch periods
        # (it waits for the previous launch's waves to retire before its own become resident), so the share of the
        # chip's issue cycles in use is priced on the period, not on the duration
        period_ms = render_s * 1e3 / max(1, st.launches)
        alg_bytes_per_launch = bytes_per_ray * st.rays / max(1, st.launches)
        achieved = alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        prop = torch.cuda.get_device_properties(local)
        valu = valu_issue(F, n_gpus, period_ms, prop.multi_processor_count, 2.4)  # 2.4 GHz: MI355X peak engine clock
        if valu:
            valu["lane_ops_per_ray"] = valu["valu_wave_insts_per_launch"] * 64.0 / max(1.0, st.rays / max(1, st.launches))
        out = {
            "metric": "Mrays/s", "value": total_rays / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3,
            "ms_per_frame": elapsed / spp * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cornell-box {WIDTH}x{HEIGHT} @ {spp} spp", "width": WIDTH, "height": HEIGHT,
                       "spp": spp, "frames_per_step": F, "triangles": scene.n_triangles,
                       "sharding": (f"32x32 tiles round-robin over {n_gpus} GPU(s) + RCCL gather of owned tiles" if by_tiles else
                                    f"{K * F} frames in {n_gpus} contiguous block(s), one per GPU, + RCCL reduce of the partial images"),
                       "seed": abi.DEFAULT_SEED},
            "rays": total_rays, "rays_per_path": total_rays / (WIDTH * HEIGHT * spp),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(F, n_gpus),
                         "kernel": "render_kernel", "launch_ms": launch_ms,
                         "launch_period_ms": period_ms, "launches_in_flight": 2 if overlap else 1,
                         "achieved_per_period": alg_bytes_per_launch / (period_ms * 1e-3) / 1e9 if period_ms > 0 else 0.0,
                         "algorithmic_bytes_per_ray": bytes_per_ray,
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "valu": valu,
                         "note": "cache-less model (SURVEY 8d); the 36-triangle scene is cache resident, "
                                 "so real HBM traffic (`traffic`) is far below it: the kernel is VALU/latency bound. "
                                 "`achieved` = algorithmic bytes per launch / `launch_ms` (event duration of one launch, what "
                                 "rocprofv3 reports per dispatch); consecutive launches overlap on two streams, so a launch "
                                 "completes every `launch_period_ms` and `achieved_per_period` is the rate the chip sustains"},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            from oracle import oracle  # CPU checker used here only as the reported baseline
            o = oracle.Oracle(packed)
            threads = min(len(os.sched_getaffinity(0)), 16)  # the CPU share of a 1-GPU box
            t = time.perf_counter()
            o.render(0, args.cpu_spp, threads=threads)
            dt = time.perf_counter() - t
            so = o.stats()
            out["cpu_baseline"] = {
                "value": so.rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"same scene and seed schedule, {WIDTH}x{HEIGHT} @ {args.cpu_spp} spp "
                          f"(frames 0-{args.cpu_spp - 1}), {so.rays} rays in {dt:.1f} s",
                "algorithmic_bytes_per_ray": abi.algorithmic_bytes(so) / max(1, so.rays),
            }
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
