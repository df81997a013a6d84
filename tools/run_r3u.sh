#!/bin/bash
# rocprofv3 passes of the final round-3 kernels (tag r03c): tools/run_r3u.sh NAME...
for n in "$@"; do bash tools/prof.sh r03c $n 2 "stats fetch write sq1 sq2 sq3" || exit 1; echo "done $n"; done
ls gpurun_out | grep prof_r03c | head -40
