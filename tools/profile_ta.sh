#!/bin/bash
# usage: tools/profile_ta.sh TAG script.py [args]  -> texture-addresser / L1 counters of the render kernels
# (two counters of one hardware block per pass: more than that "exceeds the capabilities of the hardware")
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TAGRAM0_REQ_sum"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/prof_${TAG}_p$i -o p -- python3 $R/$1 $2 $3 > $OUT/prof_${TAG}_p$i.log 2>&1 || { echo "pass $i ($set) failed"; grep -m2 -i "error\|exceeds" $OUT/prof_${TAG}_p$i.log; continue; }
  echo "pass $i done"
  python3 - <<PY
import sqlite3, os
p = "$OUT/prof_${TAG}_p$i/p_results.db"
if os.path.exists(p):
    db = sqlite3.connect(p)
    for row in db.execute("select kernel_name, counter_name, count(*), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if "rene::" in row[0]: print(row[1], "n=%d sum=%.5g avg_dur_us=%.1f" % (row[2], row[3], row[4]/1e3), row[0][:60])
PY
done
