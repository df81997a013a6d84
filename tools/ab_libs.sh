#!/bin/bash
# alternating A/B/C... of several builds of the library on a GPU box: tools/ab_libs.sh "lib1.so lib2.so ..." NAME... (whole one-launch jobs through tools/shard_sweep.py, three rounds)
LIBS=$1; shift
for i in 1 2 3; do
  for L in $LIBS; do
    for N in "$@"; do
      RENE_HIP_LIB=$L timeout -k 10 200 python3 tools/shard_sweep.py $N 1 "-,-" 2>&1 | grep -v amdgpu | sed "s/^/$L /"
    done
  done
done
