#!/bin/bash
O=gpurun_out/r3y; mkdir -p $O
run() {  # run <label> <lib> <config> <shape>
  echo "== $1: $3" >> $O/ab.log
  RENE_HIP_LIB=$2 SHAPES=$4 timeout -k 10 200 python3 tools/job_shapes.py $3 >> $O/ab.log 2>&1
}
for round in 1 2 3; do
  run base librene_hip_base.so cornell 1024:i64/64
  run plain librene_hip_np.so cornell 1024:i64/64
  run pipe librene_hip.so cornell 1024:i64/64
done
run base librene_hip_base.so veach-mis 4096:i256/256
run plain librene_hip_np.so veach-mis 4096:i256/256
grep -v "^\[\|amdgpu.ids" $O/ab.log | tail -40
