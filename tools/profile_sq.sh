#!/bin/bash
# two SQ counter passes over a short bench run; prints per-launch averages for the render kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-sq}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH -d $OUT/prof_${TAG}_a -o a -- python3 $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline > $OUT/prof_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH -d $OUT/prof_${TAG}_b -o b -- python3 $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline > $OUT/prof_${TAG}_b.log 2>&1
python3 - <<PY
import sqlite3
for sub in ("a","b"):
    db=sqlite3.connect("$OUT/prof_${TAG}_%s/%s_results.db"%(sub,sub)); cur=db.cursor()
    for row in cur.execute("select counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%render_kernel<72u, 1, false%' group by counter_name"):
        print(f"{row[0]:<28} n={row[1]:<3} avg={row[2]:.4g}  dur_ms={row[3]/1e6:.2f}")
PY
