#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on a GPU box (run through gpurun from the repo root).
#   1. --kernel-trace --stats            -> per-kernel time
#   2. --pmc FETCH_SIZE   (own pass)     -> HBM read bytes  (x2: gfx950 correction, MI355X_MICROARCH.md)
#   3. --pmc WRITE_SIZE   (own pass)     -> HBM write bytes
#   4. --pmc SQ_* (own pass)             -> VALU utilisation of the render kernel
# Summaries land in gpurun_out/prof_*; tools/summarize_profiles.py turns them into profiles/*.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-r01}
STEPS=${2:-4}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_stats -o stats -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/prof_${TAG}_stats.log 2>&1
echo "stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_${TAG}_fetch -o fetch -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/prof_${TAG}_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_${TAG}_write -o write -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/prof_${TAG}_write.log 2>&1
echo "write pass done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/prof_${TAG}_sq -o sq -- python3 $R/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/prof_${TAG}_sq.log 2>&1
echo "sq pass done"
ls -R $OUT | head -50
