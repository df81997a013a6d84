#!/bin/bash
# round 3, session 2: a further node step of an iteration only while at least INNER_MIN lanes are at an inner node (variants a8 a16 a24: three steps
# at most; b16 b24: five at most) against the fixed three steps (default build)
O=gpurun_out/r4a; mkdir -p $O
run() {  # run <label> <lib> <config> <shape>
  echo "== $1: $3" >> $O/ab.log
  RENE_HIP_LIB=$2 SHAPES=$4 timeout -k 10 200 python3 tools/job_shapes.py $3 >> $O/ab.log 2>&1
}
for round in 1 2; do
  for v in "" a8 a16 a24 b16 b24; do
    lib=librene_hip${v:+_$v}.so
    run "${v:-new}" $lib dragon-class 1024:i32/32
  done
done
for v in "" a8 a16 a24 b16 b24; do
  lib=librene_hip${v:+_$v}.so
  run "${v:-new}" $lib teapot-class 8192:i256/256
done
grep -v "^\[\|amdgpu.ids" $O/ab.log | paste - - | awk '{print $2, $3, $9, $10, $12, $13}' | sort | tail -60
