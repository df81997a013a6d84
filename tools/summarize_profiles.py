#!/usr/bin/env python3
"""Turns the rocprofv3 SQLite outputs under gpurun_out/prof_<tag>_{stats,fetch,write,sq}/ into the
small text/JSON summaries committed under profiles/.

  profiles/<tag>_kernel_stats.txt   rocprofv3 --kernel-trace --stats: per-kernel calls / total / average
  profiles/<tag>_pmc.txt            per-kernel PMC sums (FETCH_SIZE, WRITE_SIZE, SQ_*), separate passes
  profiles/hbm_traffic.json         HBM bytes per render_kernel launch (read by bench.py -> roofline.traffic)

HBM correction (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE counts 128-B requests at 64 B, so read bytes = FETCH_SIZE * 1024 * 2;
WRITE_SIZE is exact.
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def q(path, sql):
    if not os.path.exists(path):
        return []
    db = sqlite3.connect(path)
    try:
        return list(db.execute(sql))
    finally:
        db.close()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    frames_per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cmd = sys.argv[3] if len(sys.argv) > 3 else "python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline"
    g = os.path.join(ROOT, "gpurun_out")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)

    rows = q(os.path.join(g, f"prof_{tag}_stats", "stats_results.db"),
             "select name, total_calls, total_duration, average, percentage from top_kernels") if False else \
        q(os.path.join(g, f"prof_{tag}_stats", "stats_results.db"), "select * from top_kernels")
    with open(os.path.join(out, f"{tag}_kernel_stats.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
        f.write("# MI355X (gfx950), ROCm 7.2; durations in microseconds (top_kernels view of the rocpd database)\n")
        f.write(f"{'calls':>6} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel\n")
        for name, calls, total, avg, pct in rows:
            f.write(f"{calls:>6} {total:>14.1f} {avg:>12.1f} {pct:>7.3f}  {name[:150]}\n")

    pmc_lines = []
    traffic = {}
    for sub, db in (("fetch", "fetch"), ("write", "write"), ("sq", "sq")):
        rs = q(os.path.join(g, f"prof_{tag}_{sub}", f"{db}_results.db"),
               "select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) "
               "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name")
        for name, ctr, n, s, a, dur in rs:
            if "rene::" not in name:
                continue
            pmc_lines.append(f"{ctr:<22} dispatches={n:<4} sum={s:<20.1f} avg_per_dispatch={a:<18.1f} avg_dur_ns={dur:<12.0f} {name[:90]}")
            if "render_kernel" in name and "false" in name.split(",")[2]:
                traffic[ctr] = a
    with open(os.path.join(out, f"{tag}_pmc.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc <counters> -- {cmd}   (one pass per counter group; never with --stats)\n")
        f.write("# FETCH_SIZE / WRITE_SIZE are in KiB; SQ_* are summed over all shader engines\n")
        f.write("\n".join(pmc_lines) + "\n")
    if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
        rd = traffic["FETCH_SIZE"] * 1024 * 2
        wr = traffic["WRITE_SIZE"] * 1024
        rec = {"tag": tag, "frames_per_step": frames_per_step, "n_gpus": 1, "kernel": "render_kernel",
               "fetch_size_kib_avg": traffic["FETCH_SIZE"], "write_size_kib_avg": traffic["WRITE_SIZE"],
               "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
               "hbm_bytes_per_launch": rd + wr,
               "correction": "read = FETCH_SIZE KiB * 1024 * 2 (gfx950 counts 128-B requests at 64 B); write = WRITE_SIZE KiB * 1024",
               "command": cmd}
        if "SQ_INSTS_VALU" in traffic:  # the compute side of the same launch shape, from the SQ_* pass
            rec["valu_wave_insts_per_launch"] = traffic["SQ_INSTS_VALU"]
            rec["salu_wave_insts_per_launch"] = traffic.get("SQ_INSTS_SALU")
            rec["wave_cycles_per_launch"] = traffic.get("SQ_WAVE_CYCLES")
            rec["wait_any_cycles_per_launch"] = traffic.get("SQ_WAIT_INST_ANY")
            rec["waves_per_launch"] = traffic.get("SQ_WAVES")
        json.dump(rec, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
    print(open(os.path.join(out, f"{tag}_kernel_stats.txt")).read()[:1500])
    print(open(os.path.join(out, f"{tag}_pmc.txt")).read()[:3000])


if __name__ == "__main__":
    main()
