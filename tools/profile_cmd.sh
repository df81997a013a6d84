#!/bin/bash
# usage: tools/profile_cmd.sh TAG script.py  -> rocprofv3 stats + PMC passes of `python3 script.py`
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_stats -o stats -- python3 $R/$1 > $OUT/prof_${TAG}_stats.log 2>&1; echo stats
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_${TAG}_fetch -o fetch -- python3 $R/$1 > $OUT/prof_${TAG}_fetch.log 2>&1; echo fetch
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_${TAG}_write -o write -- python3 $R/$1 > $OUT/prof_${TAG}_write.log 2>&1; echo write
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/prof_${TAG}_sq -o sq -- python3 $R/$1 > $OUT/prof_${TAG}_sq.log 2>&1; echo sq
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY TCC_HIT_sum TCC_MISS_sum -d $OUT/prof_${TAG}_mem -o mem -- python3 $R/$1 > $OUT/prof_${TAG}_mem.log 2>&1; echo mem
