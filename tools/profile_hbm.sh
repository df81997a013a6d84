#!/bin/bash
# usage: tools/profile_hbm.sh TAG script.py [args]  -> HBM bytes of the render kernels: FETCH_SIZE and WRITE_SIZE, one pass
# each (KiB; on gfx950 FETCH_SIZE counts 128-byte requests at 64 bytes: read bytes = FETCH_SIZE * 1024 * 2)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/prof_${TAG}_h$i -o p -- python3 $R/$1 $2 $3 > $OUT/prof_${TAG}_h$i.log 2>&1 || { echo "pass $i ($set) failed"; grep -m2 -i "error\|exceeds" $OUT/prof_${TAG}_h$i.log; continue; }
  grep -h "megakernel" $OUT/prof_${TAG}_h$i.log
  python3 - <<PY
import sqlite3, os
p = "$OUT/prof_${TAG}_h$i/p_results.db"
if os.path.exists(p):
    db = sqlite3.connect(p)
    for row in db.execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if "rene::" in row[0]: print(row[1], "n=%d sum_KiB=%.6g avg_dur_us=%.1f" % (row[2], row[3], row[4]/1e3), row[0][:60])
PY
done
