#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py configuration on a GPU box (run through gpurun from the repo root):
#   tools/prof.sh TAG NAME [STEPS] [PASSES]      e.g.  tools/prof.sh r02a cornell 2 "stats fetch write sq1 sq2 sq3"
# One pass per counter group, --pmc never combined with --stats or any trace domain but --kernel-trace:
#   stats  --kernel-trace --stats                       per-kernel time
#   fetch  FETCH_SIZE  /  write  WRITE_SIZE             HBM bytes (x2 for reads: gfx950 correction, MI355X_MICROARCH.md)
#   sq1    instruction mix + lane utilisation           sq2  where the wave cycles go (wait / issue / active), memory instructions
#   sq3    scalar data cache + L2 hit rate            sq4  average latency of scalar / vector / LDS accesses (LEVEL / INSTS), branches
#   sq5    instruction fetch, per-unit active time    sq6  instruction cache
# Each pass is reduced to gpurun_out/prof_<TAG>_<NAME>_<pass>.json (+ .log) by tools/extract_pass.py;
# tools/summarize_profiles.py turns those into profiles/*.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=$1; NAME=$2; STEPS=${3:-2}; PASSES=${4:-"stats fetch write sq1 sq2 sq3"}
CMD="python3 $R/bench.py --only $NAME --steps $STEPS --warmup 1 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
for p in $PASSES; do
  case $p in
    stats) ARGS="--stats" ;;
    fetch) ARGS="--pmc FETCH_SIZE" ;;
    write) ARGS="--pmc WRITE_SIZE" ;;
    sq1) ARGS="--pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" ;;
    sq2) ARGS="--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS" ;;
    sq4) ARGS="--pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES" ;;
    sq5) ARGS="--pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" ;;
    sq6) ARGS="--pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_REQ SQC_TC_STALL" ;;
    sq3) ARGS="--pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" ;;
    *) echo "unknown pass $p"; exit 2 ;;
  esac
  D=$OUT/prof_${TAG}_${NAME}_$p
  rm -rf $D
  timeout -k 10 420 rocprofv3 --kernel-trace $ARGS -d $D -o p -- $CMD > $D.log 2>&1 || { echo "pass $p failed"; tail -5 $D.log; exit 1; }
  python3 $R/tools/extract_pass.py $D && rm -rf $D
done
