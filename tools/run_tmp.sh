mkdir -p gpurun_out/r3p
L=gpurun_out/r3p/thresholds2.log
: > $L
for lm in 2 4 6 8; do
echo "-- LEAF_MIN=$lm (READY_MIN default)" >> $L
RENE_LEAF_MIN=$lm SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
RENE_LEAF_MIN=$lm SHAPES=8192:i256/256 python3 tools/job_shapes.py teapot-class 2>&1 | grep -v amdgpu.ids >> $L
done
for v in ns2 ns4; do
echo "-- node steps per iteration: variant $v, LEAF_MIN=8" >> $L
RENE_HIP_LIB=librene_hip_$v.so RENE_LEAF_MIN=8 SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
RENE_HIP_LIB=librene_hip_$v.so RENE_LEAF_MIN=8 SHAPES=8192:i256/256 python3 tools/job_shapes.py teapot-class 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
