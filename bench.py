#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render hot path (BASELINE.json: Mrays/s and ms/frame at
fixed spp; achieved rate against the roofline that binds).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--only NAME] [--no-configs] [--no-cpu-baseline]

Headline workload: BASELINE.json configs[1], "Cornell-box 1024x1024 @ 1024 spp, 1x MI355X" (synthetic
Cornell-box-shaped scene built in code, rene_amd/scenes.py).  A *step* is one whole such job: clear the
accumulation image (rene/src/main.rs:1229-1237), render all 1024 frames of every pixel -- ONE launch of the persistent
kernel, every pixel's frames cut into short work items (docs/history.md section 4f) -- and wait for it.  The K timed steps are K
jobs back to back, each timed on its own as well: `step_ms_median` / `step_ms_min`.  Every job renders the same frames,
so its image must be bit-identical to the first one's -- checked after the timed region.  Inputs (scene tables, BVH) are
resident in HBM before the timed region; `value` = rays of all K jobs / elapsed.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  STRONG scaling: the job stays the 1024-spp image of
the N = 1 line.  Default since round 4: north_star's cut -- 32x32 pixel tiles round-robin over the ranks, every rank all 1024
frames of its tiles (with a pixel's frames as eight chains an eighth of the image keeps a million work items in flight:
0.87 - 0.89 render efficiency at N = 8 on one GPU's share), ranks never talk while rendering, and the one
exchange step of a job -- rene_gather_tiles: every rank sends its 1 / N of the image to rank 0 -- is inside the job, hence
inside the timed region; the image is bit-identical to the one-GPU render.  RENE_BENCH_SHARD=frames: contiguous frame blocks +
an RCCL reduce (sum) of the [3][H][W][4] f32 partial images instead; with it RENE_BENCH_SCALING=weak renders an N x 1024 spp
image.  value = rays of all ranks / max-over-ranks time.  A watchdog (class Watchdog) bounds every phase that can wait for
another rank; on overrun rank 0 prints an error record and every rank exits 3.

The JSON line also carries
  roofline     -- for the dominant kernel of the headline job.  The kernel is bound by VALU issue, not by HBM (the
                  36-triangle scene is cache resident): `bound` = "valu", `achieved` = lane-operations per
                  second = rays/s x VALU wave-instructions per ray x 64, transcendentals counted twice (they
                  issue at half rate), `peak` = 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-ops/s (a wave64
                  VALU instruction takes a SIMD two cycles; MI355X_MICROARCH.md "Wave scheduling").  Instruction
                  counts per ray come from the committed rocprofv3 PMC passes (profiles/pmc_per_ray.json:
                  per RAY, so they hold whatever K is; `pmc_stale` says whether the kernel sources have changed since
                  those passes); rates are measured live.  `useful_lane_frac` = frac x lanes active; `sclk_mhz` = the
                  engine clock during the last timed job, measured by the kernel itself (rene_stats.sclk_mhz).  `hbm` inside it = the measured
                  HBM bytes per ray (FETCH_SIZE x 2 + WRITE_SIZE, same passes) x live rays/s against 8 TB/s, and
                  SURVEY 8d's cache-less algorithmic bytes for comparison (not a fraction of anything: the scene
                  lives in the caches).  `traffic` = measured HBM bytes per launch.
  configs      -- N > 1: BASELINE's 8-GPU configurations C4 (dragon-class 1920x1080 @ 1024 spp) and C5 (teapot-class 1920x1080 @ 8192 spp),
                  one job each, sharded over the N ranks like the headline (tiles + one gather), max-over-ranks time.
                  N = 1: the other BASELINE configurations that fit one GPU, one full job each at its own
                  resolution and sample count (C3 veach-mis 1024x1024 @ 4096 spp, C4 dragon-class 1920x1080 @ 1024 spp,
                  C5 teapot-class 1920x1080 @ 8192 spp), each with rays, Mrays/s, ms/frame and the same two fractions.
                  Each runs in a process of its own, before this one touches the GPU, under --config-timeout seconds:
                  one that fails or stalls is reported as an error entry and does not take the headline line with it.
  cpu_baseline -- the CPU oracle (a port of rene's integrator; the reference itself has no CPU
                  path and cannot be built here) timed on ALL of this host's hardware threads on a bounded sample.
"""
from __future__ import annotations

import argparse
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (multi-process GPU work on this pool: the host driver only supports dmabuf IPC)
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
CLOCK_GHZ = 2.4          # peak engine clock, same guide
SIMDS_PER_CU = 4
LANES_PER_SIMD_CYCLE = 32  # SIMD-32: a wave64 VALU instruction occupies its SIMD for 2 cycles


def configurations():
    """name -> (label, scene factory, spp of one job, frames per launch).  The headline is first.  One launch per job
    (tools/job_shapes.py, round 3: Cornell 54.6 ms as one launch of 1024 frames against 55.9 as four serial ones; dragon-class
    716 against 839 ms; the teapot scene 3.82 against 4.31 s -- a launch boundary is a tail)."""
    from rene_amd import scenes
    return {
        "cornell": ("cornell-box 1024x1024 @ 1024 spp", lambda: scenes.cornell_box(1024, 1024), 1024, 1024),
        "veach-mis": ("veach-mis 1024x1024 @ 4096 spp", lambda: scenes.veach_mis(1024, 1024), 4096, 4096),
        "dragon-class": ("dragon-class (870 400 triangles, one distant light) 1920x1080 @ 1024 spp", lambda: scenes.dragon_class(1920, 1080), 1024, 1024),
        "dragon-partial": ("rene's sample_scenes/dragon with the 12 meshes its checkout holds (51 140 triangles, one distant light) through the "
                           "pbrt loader, 1280x720 @ 1024 spp", lambda: scenes.dragon_partial(1280, 720), 1024, 1024),
        "material-zoo": ("material zoo (all seven materials incl. the multi-lobe Uber / Plastic, checkerboard / imagemap / scale textures, spheres, "
                         "an environment light: the five-lobe kernels) 1024x768 @ 256 spp", lambda: scenes.material_zoo(1024, 768), 256, 256),
        "teapot-class": ("teapot-full-class: rene's sample_scenes/teapot (126 050 triangles, Substrate + checkerboard + env map) through the "
                         "pbrt loader, synthetic 1024x512 sky, 1920x1080 @ 8192 spp", lambda: scenes.teapot_full(1920, 1080), 8192, 8192),
    }


def kernel_source_hash() -> str:
    """Hash of everything the device code is built from: what the committed PMC passes were taken on (tools/summarize_profiles.py
    stamps it into profiles/pmc_per_ray.json; a different hash here means the per-ray instruction / byte counts are stale)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rene_amd", "csrc")
    for fn in ("device_code.inc", "render_wf.inc", "device_math.h", "device_scene.h", "kernels.hip", "kernels_bvh.hip", "kernels_vol.hip"):
        h.update(open(os.path.join(d, fn), "rb").read())
    for line in open(os.path.join(d, "Makefile")):
        if line.startswith("HIPFLAGS"):
            h.update(line.encode())
    return h.hexdigest()[:16]


def pmc_per_ray(name: str):
    """The per-ray figures of the committed rocprofv3 PMC passes for this configuration (tools/prof.sh +
    tools/summarize_profiles.py), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_per_ray.json"))).get(name)
    except Exception:
        return None


# What a wave64 VALU instruction occupies a SIMD for, by class, measured on the MI355X (selftest/valu_rate_probe, profiles/r03_valu_rates.txt):
# add / sub / mul / mov / logic 2.4 cycles; VOP3 fma, v_pk_fma, min / max, compares, conversions, VOP3 integer operations 4.2; transcendentals 8.2
ISSUE_CYCLES_FULL_RATE = 2.4
ISSUE_CYCLES_HALF_RATE = 4.2


def rooflines(name: str, rays_per_s: float, cus: int, alg_bytes_per_ray: float | None, sclk_mhz: float | None = None):
    """VALU-issue and HBM fractions of one configuration at the measured ray rate."""
    rec = pmc_per_ray(name)
    peak_lane_ops = cus * SIMDS_PER_CU * LANES_PER_SIMD_CYCLE * CLOCK_GHZ * 1e9
    out = {"valu": None, "hbm": None}
    if rec and rec.get("valu_wave_insts_per_ray"):
        w = rec["valu_wave_insts_per_ray"] + (rec.get("trans_wave_insts_per_ray") or 0.0)  # transcendentals: half rate
        ach = rays_per_s * w * 64.0
        out["valu"] = {"achieved": ach / 1e12, "peak": peak_lane_ops / 1e12, "unit": "Tlane-op/s", "frac": ach / peak_lane_ops,
                       "useful_lane_frac": ach / peak_lane_ops * rec["valu_lanes_active"] if rec.get("valu_lanes_active") else None,
                       "pmc_stale": rec.get("kernel_source_hash") != kernel_source_hash(),
                       "lane_ops_per_ray": rec["valu_wave_insts_per_ray"] * 64.0,
                       "trans_lane_ops_per_ray": (rec.get("trans_wave_insts_per_ray") or 0.0) * 64.0,
                       "lanes_active": rec.get("valu_lanes_active"), "wait_any_frac": rec.get("wait_any_frac"),
                       "source": rec.get("source")}
        # the same rate as cycles of a SIMD per VALU instruction, on the clock the kernels measured (nominal if none): `frac` above prices
        # every instruction at 2 cycles, which no instruction class reaches (2.4 at best, 4.2 for most of what the BVH kernels execute);
        # issue_frac_* = what fraction of a SIMD's cycles the kernel's VALU instructions occupy if all of them were full / half rate --
        # the true occupancy lies between the two (a value above 1 says the mix cannot be all of that class)
        clk = (sclk_mhz or CLOCK_GHZ * 1e3) * 1e6
        cpi = cus * SIMDS_PER_CU * clk / (rays_per_s * rec["valu_wave_insts_per_ray"])
        out["valu"].update({"issue_cycles_per_valu_inst": cpi, "issue_cycles_full_rate": ISSUE_CYCLES_FULL_RATE, "issue_cycles_half_rate": ISSUE_CYCLES_HALF_RATE,
                            "issue_frac_if_all_full_rate": ISSUE_CYCLES_FULL_RATE / cpi, "issue_frac_if_all_half_rate": ISSUE_CYCLES_HALF_RATE / cpi})
    if rec and rec.get("hbm_read_bytes_per_ray") is not None:
        b = rec["hbm_read_bytes_per_ray"] + rec["hbm_write_bytes_per_ray"]
        out["hbm"] = {"achieved": rays_per_s * b / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                      "frac": rays_per_s * b / 1e9 / HBM_PEAK_GBPS, "measured_bytes_per_ray": b,
                      "algorithmic_bytes_per_ray": alg_bytes_per_ray,
                      "algorithmic_GBps_cacheless": rays_per_s * alg_bytes_per_ray / 1e9 if alg_bytes_per_ray else None,
                      "source": rec.get("source")}
    return out


def algorithmic_bytes_per_ray(packed, api, abi, device, frames=4, **shard):
    """SURVEY 8d's cache-less model, evaluated from the kernel's own node / primitive counters over this build's
    BVH4 (RENE_FLAG_FORCE_BVH, so that the figure does not depend on which intersection back end renders the scene)."""
    with api.Renderer(packed, device=device, flags=abi.FLAG_COUNTERS | abi.FLAG_FORCE_BVH, **shard) as rc:
        rc.render(0, frames)
        cst = rc.stats()
    return abi.algorithmic_bytes(cst) / max(1, cst.rays)


def run_config(name: str, device: int = 0):
    """One full job of one of the additional configurations (C3-C5) on `device`: its figures for the `configs` object."""
    import numpy as np
    import torch
    from rene_amd import abi, api
    lab, mk, spp, fpl = configurations()[name]
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    sc = mk()
    pk = sc if hasattr(sc, "byref") else sc.to_desc()
    bpr = algorithmic_bytes_per_ray(pk, api, abi, device, frames=2)
    def job(flags):
        with api.Renderer(pk, device=device, flags=flags) as rr:
            rr.render(0, min(fpl, 64))
            rr.sync()
            rr.reset()
            t1 = time.perf_counter()
            for f0 in range(0, spp, fpl):
                rr.render(f0, min(fpl, spp - f0))
            rr.sync()
            dt = time.perf_counter() - t1
            return dt, rr.stats(), rr.download(0)
    # One job, the library's default path.  (Round 3 timed the BVH configurations twice -- strict frame order and, opt-in, two chains of
    # frames per pixel; since round 4 every context renders a pixel's frames as eight chains, frame f in chain f % 8, and the image is
    # cut-independent: there is one mode.  docs/history.md section 4f.)
    dt, s2, im = job(0)
    assert bool(np.isfinite(im).all()) and float(im.mean()) > 0.0, f"{name}: image empty or non-finite"
    rl2 = rooflines(name, s2.rays / dt, cus, bpr, s2.sclk_mhz or None)
    return {"workload": lab, "width": pk.xres, "height": pk.yres, "spp": spp, "frames_per_launch": fpl,
            "frame_chains": 8,
            "triangles": api.pack_info(pk).n_triangles, "rays": s2.rays, "rays_per_path": s2.rays / max(1, s2.paths),
            "value": s2.rays / dt / 1e6, "unit": "Mrays/s", "seconds": dt, "ms_per_frame": dt / spp * 1e3,
            "launch_ms": s2.kernel_ms / max(1, s2.launches), "launch_period_ms": dt * 1e3 / max(1, s2.launches),
            "kernel": (pmc_per_ray(name) or {}).get("kernel"),
            "valu": rl2["valu"], "hbm": rl2["hbm"]}


def run_config_sharded(name: str, local: int, rank: int, world: int, backend: str, in_library: bool, by_tiles: bool):
    """N > 1: one whole job of one of BASELINE's 8-GPU configurations (C4 dragon-class, C5 teapot-class) sharded over the ranks the
    way the headline job is -- contiguous frame blocks + one reduce onto rank 0 (or tiles + a gather) -- timed between barriers,
    max over ranks.  Every rank must take every step (communicator set-up and the exchange are collectives): a step that can fail
    locally is agreed on first, and a configuration some rank cannot set up is skipped by all."""
    import torch
    import torch.distributed as dist
    from rene_amd import abi, api, dist as rdist
    lab, mk, spp, fpl = configurations()[name]
    dev = f"cuda:{local}"
    ok = torch.ones(1, device=dev)
    r = fb = None
    err = ""
    try:
        sc = mk()
        pk = sc if hasattr(sc, "byref") else sc.to_desc()
        W, H = pk.xres, pk.yres
        t_rank, t_world = (rank, world) if by_tiles else (0, 1)
        fb = torch.zeros((3, H, W, 4), dtype=torch.float32, device=dev)
        r = api.Renderer(pk, device=local, framebuffer_ptr=fb.data_ptr(), shard_mode=abi.SHARD_TILES, shard_rank=t_rank, shard_count=t_world)
        my_uid = api.comm_unique_id() if in_library else None
    except Exception as e:  # noqa: BLE001 -- reported in the line
        err = repr(e)
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if ok.item() <= 0:
        if r is not None:
            r.close()
        return {"workload": lab, "error": err or "another rank could not set this configuration up"}
    if in_library:
        uid = [my_uid if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        r.comm_init(world, rank, uid[0])
    lo, hi = (0, spp) if by_tiles else rdist.frame_block(rank, world, spp)

    def exchange():
        if in_library:
            r.gather_tiles(0) if by_tiles else r.reduce(0)
            r.sync()
        elif by_tiles:
            rdist.gather_owned_tiles(fb, rank, world, dst=0)
        else:
            rdist.reduce_framebuffer(fb, dst=0)
        torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    r.render(lo, min(64, hi - lo))  # warm-up: the kernel's code object, the communicator's rings
    r.sync()
    exchange()
    fence()
    t0 = time.perf_counter()
    r.reset()
    r.render(lo, hi - lo)
    r.sync()
    exchange()
    fence()
    dt = time.perf_counter() - t0
    st = r.stats()
    tot = torch.tensor([float(st.rays), float(st.paths)], dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    finite = bool(torch.isfinite(fb[0, :, :, :3]).all()) and float(fb[0, :, :, :3].mean()) > 0.0 if rank == 0 else True
    r.close()
    dt = float(tmax.item())
    return {"workload": lab, "n_gpus": world, "width": W, "height": H, "spp": spp, "frames_per_gpu": hi - lo,
            "sharding": "tiles + gather" if by_tiles else "frame blocks + reduce",
            "exchange": "RCCL inside librene_hip" if in_library else f"torch.distributed ({backend})",
            "rays": float(tot[0].item()), "rays_per_path": float(tot[0].item() / max(1.0, tot[1].item())),
            "value": float(tot[0].item()) / dt / 1e6, "unit": "Mrays/s", "seconds": dt, "ms_per_frame": dt / spp * 1e3,
            "image_ok": finite}


def configs_in_children(head: str, timeout_s: float):
    """Every additional configuration in a process of its own, one after the other, BEFORE this process touches the GPU
    (a process that has initialised HIP must not be the one that spawns).  A configuration that fails or does not finish in
    time is reported as such instead of taking the headline line down with it."""
    import subprocess
    out = {}
    for name in configurations():
        if name == head:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--config-child", name]
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
            lines = [l for l in p.stdout.splitlines() if l.startswith('{"config"')]
            if p.returncode == 0 and lines:
                out[name] = json.loads(lines[-1])["config"]
            else:
                out[name] = {"error": f"exit code {p.returncode}", "stderr_tail": p.stderr[-400:]}
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"did not finish within {timeout_s:.0f} s (killed)"}
    return out


class Watchdog:
    """N > 1 only: `arm(phase, seconds)` starts a timer; if `disarm()` does not come in time, rank 0 prints ONE JSON error record (same keys as the
    bench line, value null, `error` says which phase hung) and the process leaves with exit code 3 through os._exit -- from the timer's thread, because
    the main thread is then inside a rendezvous or a collective that will not return.  One GPU: does nothing."""

    def __init__(self, rank: int, world: int, n_gpus: int):
        self.rank, self.world, self.n_gpus, self.timer = rank, world, n_gpus, None

    def _fire(self, phase: str, seconds: float):
        msg = f"rank {self.rank} of {self.world}: {phase} did not finish within {seconds:.0f} s"
        print(f"[bench] {msg}; giving up", file=sys.stderr, flush=True)
        if self.rank == 0:
            print(json.dumps({"metric": "Mrays/s", "value": None, "unit": "Mrays/s", "n_gpus": self.n_gpus, "higher_is_better": True,
                              "error": msg, "config": {"workload": "cornell-box 1024x1024 @ 1024 spp"}}), flush=True)
        os._exit(3)

    def arm(self, phase: str, seconds: float):
        self.disarm()
        if self.world > 1 or os.environ.get("RENE_BENCH_TEST_WATCHDOG"):
            import threading
            self.timer = threading.Timer(seconds, self._fire, (phase, seconds))
            self.timer.daemon = True
            self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--only", default=None, help="run this configuration as the timed workload instead of the headline (profiling)")
    ap.add_argument("--no-configs", action="store_true", help="skip the additional configurations (C3-C5)")
    ap.add_argument("--cpu-spp", type=int, default=128, help="frames of the CPU-oracle baseline sample (~10 s on 16 threads; bounded to ~25 s whatever the host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config-child", default=None, help=argparse.SUPPRESS)  # one additional configuration in this process (see configs_in_children)
    ap.add_argument("--config-timeout", type=float, default=300.0, help="seconds an additional configuration may take before it is given up")
    ap.add_argument("--phase-timeout", type=float, default=150.0, help="N > 1: seconds the rendezvous, the communicator set-up and the warm-up may each take (the timed region: six times that) before the run gives up with an error record")
    args = ap.parse_args()
    if args.config_child:
        import torch
        torch.cuda.set_device(0)
        print(json.dumps({"config": run_config(args.config_child, 0)}), flush=True)
        return
    # the additional configurations first, each in its own process (only the one-GPU run reports them)
    child_configs = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and max(1, args.gpus) == 1 and not args.no_configs and not args.only:
        child_configs = configs_in_children("cornell", args.config_timeout)

    import numpy as np
    import torch
    from rene_amd import abi, api, dist as rdist

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
    # N > 1: nothing that can wait for another rank may wait for ever -- the driver has ONE scaling run.  A watchdog thread per rank, armed around
    # the rendezvous, the communicator set-up, the warm-up (first exchange: RCCL builds its rings there) and the timed region: when a phase
    # overruns, rank 0 prints an error record in the bench line's place and every rank leaves with a non-zero exit code (os._exit from the
    # watchdog thread: the main thread may be inside a collective; the launcher then ends the other ranks).  Nothing is re-executed.
    watchdog = Watchdog(rank, world, max(1, args.gpus))
    if world > 1:
        watchdog.arm("rendezvous (torch.distributed.init_process_group)", args.phase_timeout)
        rdist.init_process_group(backend)
        import torch.distributed as dist
        watchdog.disarm()

    cfgs = configurations()
    head = args.only or "cornell"
    if head not in cfgs:
        raise SystemExit(f"--only: unknown configuration {head!r} (have {', '.join(cfgs)})")
    label, make, SPP, F = cfgs[head]
    K, Wm = max(1, args.steps), max(0, args.warmup)
    scene = make()
    packed = scene if hasattr(scene, "byref") else scene.to_desc()  # a loaded pbrt scene is its own table owner
    WIDTH, HEIGHT = packed.xres, packed.yres
    n_triangles = api.pack_info(packed).n_triangles

    # N > 1, default since round 4: north_star's cut -- 32 x 32 pixel TILES round-robin over the GPUs, every rank all 1024 frames of its tiles,
    # one gather of the owned tiles onto rank 0 (1 / N of the image per rank); the image is bit-identical to the one-GPU render.  (With a
    # pixel's frames as eight chains a rank's eighth of the image fills the chip: render efficiency at N = 8 on one GPU's share 0.87, was 0.33;
    # DESIGN.md section 6.)  RENE_BENCH_SHARD=frames: contiguous frame blocks + one reduce of the partial images instead.
    by_tiles = os.environ.get("RENE_BENCH_SHARD", "tiles") == "tiles"
    # STRONG scaling: the job is the N = 1 line's (same image, same 1024 spp).  RENE_BENCH_SCALING=weak (frame blocks only): every GPU renders
    # 1024 frames of an N x 1024 spp image.  The tile cut shrinks a rank's share by construction, so it is always strong.
    weak = world > 1 and not by_tiles and os.environ.get("RENE_BENCH_SCALING", "strong") == "weak"
    JOB_SPP = SPP * world if weak else SPP
    t_rank, t_world = (rank, world) if by_tiles else (0, 1)
    shard = dict(shard_mode=abi.SHARD_TILES, shard_rank=t_rank, shard_count=t_world)
    fb = torch.zeros((3, HEIGHT, WIDTH, 4), dtype=torch.float32, device=f"cuda:{local}")
    r = api.Renderer(packed, device=local, framebuffer_ptr=fb.data_ptr(), **shard)
    # this rank's launches of one job: (first_frame, n_frames) -- one launch for its whole share
    if by_tiles:
        launches = [(f0, min(F, SPP - f0)) for f0 in range(0, SPP, F)]
    else:
        lo, hi = rdist.frame_block(rank, world, JOB_SPP)
        launches = [(f0, min(F, hi - f0)) for f0 in range(lo, hi, F)]

    # The exchange step runs inside the library (rene_reduce / rene_gather_tiles: ncclReduce / ncclSend+Recv on the
    # context's stream); torch.distributed only carries the communicator's 128-byte id and the barriers.  With
    # RENE_DIST_BACKEND=gloo (several ranks on one GPU, which RCCL refuses) the torch-level exchange of rene_amd.dist stands in.
    in_library = world > 1 and backend == "nccl"
    if in_library:
        # every rank must end up on the same path, and ncclCommInitRank blocks until all ranks have called it: first agree
        # that RCCL loads everywhere (rene_comm_unique_id touches nothing but the library), then set the communicator up
        ok = torch.ones(1, device=f"cuda:{local}")
        try:
            my_uid = api.comm_unique_id()
        except Exception as e:  # RCCL missing: the torch-level exchange of rene_amd.dist stands in (and the line says so)
            print(f"[bench] rank {rank}: exchange inside the library unavailable ({e}); using torch.distributed", file=sys.stderr, flush=True)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        in_library = bool(ok.item() > 0)
        if in_library:
            watchdog.arm("communicator set-up (rene_comm_init: ncclCommInitRank on every rank)", args.phase_timeout)
            uid = [my_uid if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            r.comm_init(world, rank, uid[0])
            watchdog.disarm()

    def exchange():
        if world > 1:
            if in_library:
                r.gather_tiles(0) if by_tiles else r.reduce(0)
                r.sync()
            elif by_tiles:
                rdist.gather_owned_tiles(fb, rank, world, dst=0)
            else:
                rdist.reduce_framebuffer(fb, dst=0)

    def job():
        """One whole job: clear, render every frame of this rank's share, wait, exchange."""
        r.reset()
        for f0, nf in launches:
            r.render(f0, nf)
        r.sync()
        exchange()
        if world > 1:
            torch.cuda.synchronize()

    # ---- untimed: algorithmic bytes per ray (a counting pass over the BVH) ----
    bytes_per_ray = algorithmic_bytes_per_ray(packed, api, abi, local, frames=min(F, 8) if head == "cornell" else 2, **shard)

    # ---- warmup (kernel + the collective: RCCL sets its rings up lazily) ----
    if world > 1:
        watchdog.arm("warm-up jobs (the first exchange: RCCL builds its rings)", args.phase_timeout)
    for _ in range(max(Wm, 1 if world > 1 else 0)):  # (N > 1: at least one untimed job, so that the timed region never holds the first exchange)
        job()
    torch.cuda.synchronize()
    watchdog.disarm()
    if world > 1:
        watchdog.arm("timed region + the 8-GPU configurations", 6.0 * args.phase_timeout)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step_ms, first_image, identical = [], None, True
    kernel_ms, n_launches, rays_job = 0.0, 0, 0
    fence()
    sclk = None
    t0 = time.perf_counter()
    for k in range(K):
        t1 = time.perf_counter()
        job()
        step_ms.append((time.perf_counter() - t1) * 1e3)
        if k == 0 or k == K - 1:  # bookkeeping of the first and last job only: a few host-side reads, no device work
            st = r.stats()
            kernel_ms, n_launches, rays_job, sclk = st.kernel_ms, st.launches, st.rays, st.sclk_mhz
    fence()
    elapsed = time.perf_counter() - t0

    # every job renders the same frames: the last image equals a fresh render of the job, bit for bit (one GPU; with
    # N > 1 the exchange changes rank 0's image in place, compared there as well: the reduce is deterministic)
    last = fb.clone()
    torch.cuda.synchronize()  # the copy runs on torch's stream, the job on the context's own
    job()
    torch.cuda.synchronize()
    identical = bool(torch.equal(last[..., :3], fb[..., :3]))

    rays = torch.tensor([float(rays_job) * K], dtype=torch.float64, device=f"cuda:{local}")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}")
    if world > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays, elapsed = float(rays.item()), float(tmax.item())

    # N > 1: BASELINE's two 8-GPU configurations (C4, C5), one sharded job each, after the headline's timed region
    # (RENE_BENCH_MULTI_CONFIGS=0 skips them)
    multi_configs = None
    if world > 1 and not args.only and not args.no_configs and os.environ.get("RENE_BENCH_MULTI_CONFIGS", "1") != "0":
        multi_configs = {}
        for name in ("dragon-class", "teapot-class"):
            multi_configs[name] = run_config_sharded(name, local, rank, world, backend, in_library, by_tiles)

    watchdog.disarm()
    if rank == 0:
        img = fb[0, :, :, :3]
        assert bool(torch.isfinite(img).all()) and float(img.mean()) > 0.0, "framebuffer is empty or non-finite"
        assert identical or world > 1, "two renders of the same job differ"
        prop = torch.cuda.get_device_properties(local)
        cus = prop.multi_processor_count
        rate = total_rays / elapsed
        launch_ms = kernel_ms / max(1, n_launches)  # HIP events around each launch of one job, on the launch's stream
        period_ms = statistics.median(step_ms) / max(1, len(launches))
        rl = rooflines(head, rate / n_gpus, cus, bytes_per_ray, sclk or None)
        rec = pmc_per_ray(head) or {}
        rays_per_launch = rays_job / max(1, n_launches)
        traffic = ((rec["hbm_read_bytes_per_ray"] + rec["hbm_write_bytes_per_ray"]) * rays_per_launch
                   if rec.get("hbm_read_bytes_per_ray") is not None else None)
        valu = rl["valu"] or {"achieved": None, "peak": cus * SIMDS_PER_CU * LANES_PER_SIMD_CYCLE * CLOCK_GHZ / 1e3, "unit": "Tlane-op/s", "frac": None}
        out = {
            "metric": "Mrays/s", "value": rate / 1e6, "unit": "Mrays/s",
            "n_gpus": n_gpus, "steps": K, "warmup": Wm, "ms_per_step": elapsed / K * 1e3,
            "step_ms_median": statistics.median(step_ms), "step_ms_min": min(step_ms),
            "ms_per_frame": elapsed / (K * JOB_SPP) * 1e3,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": WIDTH, "height": HEIGHT,
                       "spp": JOB_SPP, "spp_per_gpu": JOB_SPP // n_gpus if not by_tiles else JOB_SPP, "frames_per_launch": F, "launches_per_step": len(launches), "triangles": n_triangles,
                       "step": "one whole job: clear the image, render every frame, wait for the last launch" + (", exchange" if world > 1 else ""),
                       "sharding": (f"32x32 tiles round-robin over {n_gpus} GPU(s) + RCCL gather of owned tiles" if by_tiles else
                                    f"{JOB_SPP} frames in {n_gpus} contiguous block(s), one per GPU, + RCCL reduce of the partial images"),
                       "exchange": ("none (one GPU)" if world == 1 else "rene_reduce / rene_gather_tiles (RCCL inside librene_hip)" if in_library
                                    else f"torch.distributed ({backend}) on the framebuffer tensor"),
                       "seed": abi.DEFAULT_SEED},
            "rays": total_rays, "rays_per_path": total_rays / (WIDTH * HEIGHT * JOB_SPP * K),
            "jobs_bit_identical": identical,
            "roofline": {"bound": "valu", "achieved": valu["achieved"], "peak": valu["peak"], "unit": valu["unit"],
                         "frac": valu["frac"], "useful_lane_frac": valu.get("useful_lane_frac"), "traffic": traffic,
                         "kernel": rec.get("kernel", "render_kernel"), "launch_ms": launch_ms,
                         "launch_period_ms": period_ms, "launches_in_flight": 1,
                         "sclk_mhz": sclk, "pmc_stale": valu.get("pmc_stale"), "kernel_source_hash": kernel_source_hash(),
                         "valu": rl["valu"], "hbm": rl["hbm"],
                         "issue_cycles_per_valu_inst": valu.get("issue_cycles_per_valu_inst"), "issue_frac_if_all_full_rate": valu.get("issue_frac_if_all_full_rate"),
                         "issue_frac_if_all_half_rate": valu.get("issue_frac_if_all_half_rate"),
                         "note": "the scene is cache resident: VALU issue is the roof that binds, priced at 2 cycles per wave64 "
                                 "VALU instruction on a SIMD-32 (transcendentals 4), nominal 2.4 GHz (`sclk_mhz`: what the clock was "
                                 "during the last timed job, measured by the kernel); rates are the sustained ones (rays of the timed jobs / elapsed); `launch_ms` = "
                                 "HIP events around the launch, on its stream (what rocprofv3 reports per dispatch); `useful_lane_frac` = "
                                 "frac x lanes active; `pmc_stale`: the kernel sources differ from the ones the PMC passes ran on; "
                                 "`hbm.frac` is measured HBM traffic (PMC) against 8 TB/s; `issue_cycles_per_valu_inst` = SIMD cycles per VALU instruction at this rate -- "
                                 "measured instruction costs on this chip are 2.4 (add / mul / mov / logic), 4.2 (VOP3 fma, min / max, compares, conversions) and 8.2 "
                                 "(transcendentals) cycles, profiles/r03_valu_rates.txt: the 2-cycle peak of `frac` is not reachable, `issue_frac_if_all_full_rate` is the "
                                 "fraction of the reachable one if every instruction were of the cheapest class"},
        }
        # ---- the other configurations, one full job each (N = 1 only; measured in child processes before this one started) ----
        if child_configs is not None:
            out["configs"] = child_configs
            # (compact copy inside `config`, which the driver's record of the line keeps: name -> [Mrays/s, VALU issue frac, useful lane frac])
            out["config"]["other_configs"] = {k: ([round(v["value"], 1), (v.get("valu") or {}).get("frac"), (v.get("valu") or {}).get("useful_lane_frac")]
                                                  if isinstance(v, dict) and v.get("value") else [None, None, None]) for k, v in child_configs.items()}
        if multi_configs is not None:
            out["configs"] = multi_configs
        if n_gpus == 1 and not args.no_cpu_baseline:
            from oracle import oracle  # CPU checker used here only as the reported baseline
            o = oracle.Oracle(packed)
            threads = len(os.sched_getaffinity(0))  # all host hardware threads this process may use
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
        # the line's last key, so that a reader who keeps only its tail still has every number that matters: the headline and, compactly, the others
        out["summary"] = {"Mrays/s": round(out["value"], 1), "n_gpus": n_gpus, "ms_per_step": round(out["ms_per_step"], 3),
                          "roofline_frac": out["roofline"]["frac"], "useful_lane_frac": out["roofline"]["useful_lane_frac"], "pmc_stale": out["roofline"]["pmc_stale"],
                          "other_configs [Mrays/s, valu frac, useful lanes]": out["config"].get("other_configs"),
                          "multi_gpu_configs Mrays/s": ({k: (round(v["value"], 1) if isinstance(v, dict) and v.get("value") else None) for k, v in multi_configs.items()}
                                                        if multi_configs is not None else None),
                          "cpu_baseline Mrays/s": (out.get("cpu_baseline") or {}).get("value")}
        print(json.dumps(out), flush=True)
    try:
        r.close()
    except Exception:
        pass
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
