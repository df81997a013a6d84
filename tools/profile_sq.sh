#!/bin/bash
# usage: tools/profile_sq.sh TAG script.py [args]  -> SQ counters of the render kernels (one pass, the set profile_bench.sh uses)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/prof_${TAG}_sq -o p -- python3 $R/$1 $2 $3 > $OUT/prof_${TAG}_sq.log 2>&1 || { echo "pass failed"; grep -m2 -i "error\|exceeds" $OUT/prof_${TAG}_sq.log; exit 1; }
grep -h "megakernel" $OUT/prof_${TAG}_sq.log
python3 - <<PY
import sqlite3, os
p = "$OUT/prof_${TAG}_sq/p_results.db"
db = sqlite3.connect(p)
for row in db.execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
    if "rene::" in row[0]: print(row[1], "n=%d sum=%.6g avg_dur_us=%.1f" % (row[2], row[3], row[4]/1e3), row[0][:60])
PY
