#!/bin/bash
# round 3, session 2: kernel arguments fetched together (logic step, item switch) against the committed kernels (variant base), and compiler
# scheduling strategies (variants s1 s2 s3 pm o2)
O=gpurun_out/r3z; mkdir -p $O
run() {  # run <label> <lib> <config> <shape>
  echo "== $1: $3" >> $O/ab.log
  RENE_HIP_LIB=$2 SHAPES=$4 timeout -k 10 200 python3 tools/job_shapes.py $3 >> $O/ab.log 2>&1
}
for round in 1 2; do
  for v in base "" s1 s2 s3 pm o2; do
    lib=librene_hip${v:+_$v}.so
    run "${v:-new}" $lib cornell 1024:i64/64
    run "${v:-new}" $lib dragon-class 1024:i32/32
  done
done
for v in base "" s1 s2 s3 pm o2; do
  lib=librene_hip${v:+_$v}.so
  run "${v:-new}" $lib teapot-class 8192:i256/256
  run "${v:-new}" $lib veach-mis 4096:i256/256
done
grep -v "^\[\|amdgpu.ids" $O/ab.log | paste - - | awk '{print $2, $3, $9, $10, $12, $13}' | sort | tail -60
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_volpath.py tests/test_gpu_edge.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
