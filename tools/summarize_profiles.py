#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/prof.sh (gpurun_out/prof_<tag>_<name>_<pass>.json + .log) into the
summaries committed under profiles/:

  profiles/<tag>_<name>_kernel_stats.txt   rocprofv3 --kernel-trace --stats: per-kernel calls / total / average
  profiles/<tag>_<name>_pmc.txt            per-kernel PMC sums, one block per pass
  profiles/pmc_per_ray.json[<name>]        per-RAY figures of the dominant kernel, read by bench.py (roofline / configs)

Per ray: the last `launches_per_step` dispatches of the dominant kernel are one whole job (bench.py renders the job
once more after its timed region); their counter sums divided by the job's ray count (from bench.py's JSON line in the
pass's log) hold whatever --steps the bench is run with.

HBM correction (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950
FETCH_SIZE counts 128-B requests at 64 B, so read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is exact.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_THREAD_CYCLES_VALU counts lane-quad-cycles.

usage: tools/summarize_profiles.py TAG NAME
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = ("fetch", "write", "sq1", "sq2", "sq3", "sq4", "sq5", "sq6")


def load_pass(path):
    """tools/extract_pass.py's reduction of one pass: {"top_kernels": [...], "dispatches": [[kernel, dispatch, counter, sum, duration, ...]]}"""
    try:
        return json.load(open(path))
    except Exception:
        return {"top_kernels": [], "dispatches": []}


def bench_line(log):
    try:
        for line in open(log):
            line = line.strip()
            if line.startswith('{"metric"'):
                return json.loads(line)
    except Exception:
        pass
    return None


def main():
    tag, name = sys.argv[1], sys.argv[2]
    g = os.path.join(ROOT, "gpurun_out")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    base = os.path.join(g, f"prof_{tag}_{name}_")

    rows = load_pass(base + "stats.json")["top_kernels"]
    line = bench_line(base + "stats.log")
    cmd = f"python3 bench.py --only {name} --steps {line['steps'] if line else '?'} --warmup 1 --no-cpu-baseline"
    if rows:
        with open(os.path.join(out, f"{tag}_{name}_kernel_stats.txt"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
            f.write("# MI355X (gfx950), ROCm 7.2; durations in microseconds (top_kernels view of the rocpd database)\n")
            if line:
                f.write(f"# bench line of this pass: {line['value']:.0f} Mrays/s, launch_ms (HIP events) {line['roofline']['launch_ms']:.3f}, "
                        f"launch_period_ms {line['roofline']['launch_period_ms']:.3f}\n")
            f.write(f"{'calls':>6} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel\n")
            for nm, calls, total, avg, pct in rows:
                f.write(f"{calls:>6} {total:>14.1f} {avg:>12.1f} {pct:>7.3f}  {nm[:160]}\n")

    per_ray, pmc_lines, kernel = {}, [], None
    for p in PASSES:
        line_p = bench_line(base + p + ".log")
        rs = [r[:5] for r in load_pass(base + p + ".json")["dispatches"]]
        if not rs:
            continue
        # dominant kernel = the one with the largest total duration
        tot = {}
        for kn, did, cn, v, dur in rs:
            tot.setdefault(kn, {})[did] = dur
        kernel_p = max(tot, key=lambda k: sum(tot[k].values()))
        kernel = kernel or kernel_p
        L = line_p["config"]["launches_per_step"] if line_p else 1
        rays_job = line_p["rays"] / line_p["steps"] if line_p else None
        dids = sorted(tot[kernel_p])[-L:]
        pmc_lines.append(f"## pass {p}: {cmd}")
        pmc_lines.append(f"## kernel {kernel_p[:140]}")
        pmc_lines.append(f"## {len(tot[kernel_p])} dispatches; the last {L} are one whole job of {rays_job:.0f} rays" if rays_job else "## (no bench line)")
        sums, alls = {}, {}
        for kn, did, cn, v, dur in rs:
            if kn != kernel_p:
                continue
            alls[cn] = alls.get(cn, 0.0) + v
            if did in dids:
                sums[cn] = sums.get(cn, 0.0) + v
        avg_dur = sum(tot[kernel_p][d] for d in dids) / max(1, len(dids))
        for cn in sorted(sums):
            pmc_lines.append(f"{cn:<26} job_sum={sums[cn]:<20.1f} per_ray={sums[cn] / rays_job if rays_job else float('nan'):<14.6g} all_dispatches_sum={alls[cn]:<20.1f} avg_dur_ns={avg_dur:.0f}")
        if rays_job:
            for cn, v in sums.items():
                per_ray[cn] = v / rays_job
            per_ray["_rate_" + p] = line_p["value"]

    with open(os.path.join(out, f"{tag}_{name}_pmc.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc <counters>, one pass per counter group (tools/prof.sh); never with --stats\n")
        f.write("# FETCH_SIZE / WRITE_SIZE in KiB; SQ_* summed over all shader engines / XCDs; quad-cycle units for *_CYCLES, WAIT_*, ACTIVE_*\n")
        f.write("\n".join(pmc_lines) + "\n")

    if per_ray:
        path = os.path.join(out, "pmc_per_ray.json")
        try:
            allrec = json.load(open(path))
        except Exception:
            allrec = {}
        rec = {"source": f"profiles/{tag}_{name}_pmc.txt", "kernel": (kernel or "").split("(")[0][:120]}
        if "SQ_INSTS_VALU" in per_ray:
            rec["valu_wave_insts_per_ray"] = per_ray["SQ_INSTS_VALU"]
            rec["trans_wave_insts_per_ray"] = per_ray.get("SQ_INSTS_VALU_TRANS_F32")
            rec["salu_wave_insts_per_ray"] = per_ray.get("SQ_INSTS_SALU")
            if per_ray.get("SQ_THREAD_CYCLES_VALU") and per_ray.get("SQ_ACTIVE_INST_VALU"):
                rec["valu_lanes_active"] = per_ray["SQ_THREAD_CYCLES_VALU"] / (64.0 * per_ray["SQ_ACTIVE_INST_VALU"])
        if "SQ_WAIT_ANY" in per_ray and per_ray.get("SQ_WAVE_CYCLES"):
            rec["wait_any_frac"] = per_ray["SQ_WAIT_ANY"] / per_ray["SQ_WAVE_CYCLES"]
            rec["wait_inst_any_frac"] = per_ray.get("SQ_WAIT_INST_ANY", 0.0) / per_ray["SQ_WAVE_CYCLES"]
            rec["active_inst_any_frac"] = per_ray.get("SQ_ACTIVE_INST_ANY", 0.0) / per_ray["SQ_WAVE_CYCLES"]
            rec["smem_insts_per_ray"] = per_ray.get("SQ_INSTS_SMEM")
            rec["vmem_rd_insts_per_ray"] = per_ray.get("SQ_INSTS_VMEM_RD")
            rec["lds_insts_per_ray"] = per_ray.get("SQ_INSTS_LDS")
        if "FETCH_SIZE" in per_ray and "WRITE_SIZE" in per_ray:
            rec["hbm_read_bytes_per_ray"] = per_ray["FETCH_SIZE"] * 1024 * 2
            rec["hbm_write_bytes_per_ray"] = per_ray["WRITE_SIZE"] * 1024
            rec["hbm_correction"] = "read = FETCH_SIZE KiB * 1024 * 2 (gfx950 counts 128-B requests at 64 B); write = WRITE_SIZE KiB * 1024"
        if per_ray.get("TCC_HIT_sum") is not None and per_ray.get("TCC_MISS_sum") is not None:
            rec["l2_hit_rate"] = per_ray["TCC_HIT_sum"] / max(1e-30, per_ray["TCC_HIT_sum"] + per_ray["TCC_MISS_sum"])
        if per_ray.get("SQC_DCACHE_REQ"):
            rec["scalar_cache_req_per_ray"] = per_ray["SQC_DCACHE_REQ"]
            rec["scalar_cache_miss_rate"] = per_ray.get("SQC_DCACHE_MISSES", 0.0) / per_ray["SQC_DCACHE_REQ"]
        rec["mrays_per_s_during_passes"] = {k[6:]: v for k, v in per_ray.items() if k.startswith("_rate_")}
        sys.path.insert(0, ROOT)
        import bench  # the hash of the kernel sources these passes ran on (bench.py: `pmc_stale`)
        rec["kernel_source_hash"] = bench.kernel_source_hash()
        rec["mode"] = "library default (eight frame chains per pixel; the mode every reported rate is measured in)"
        allrec[name] = rec
        json.dump(allrec, open(path, "w"), indent=1, sort_keys=True)
        print(json.dumps(rec, indent=1))
    for fn in (f"{tag}_{name}_kernel_stats.txt", f"{tag}_{name}_pmc.txt"):
        p = os.path.join(out, fn)
        if os.path.exists(p):
            print(open(p).read()[:2500])


if __name__ == "__main__":
    main()
