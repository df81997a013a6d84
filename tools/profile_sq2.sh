#!/bin/bash
# SQ + TCC counter passes over tools/dev_dragon2.py (BVH kernel on the 870k-triangle scene)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=${1:-dq}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD -d $OUT/prof_${TAG}_a -o a -- python3 $R/tools/dev_dragon2.py > $OUT/prof_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_IFETCH -d $OUT/prof_${TAG}_b -o b -- python3 $R/tools/dev_dragon2.py > $OUT/prof_${TAG}_b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/prof_${TAG}_c -o c -- python3 $R/tools/dev_dragon2.py > $OUT/prof_${TAG}_c.log 2>&1
python3 - <<PY
import sqlite3
for sub in ("a","b","c"):
    try:
        db=sqlite3.connect("$OUT/prof_${TAG}_%s/%s_results.db"%(sub,sub)); cur=db.cursor()
        for row in cur.execute("select counter_name, count(*), max(value), max(duration) from counters_collection where kernel_name like '%render_kernel%' group by counter_name"):
            print(f"{row[0]:<30} n={row[1]:<3} max={row[2]:.4g}  dur_ms={row[3]/1e6:.2f}")
    except Exception as e: print(sub, e)
PY
grep Mrays $OUT/prof_${TAG}_a.log
