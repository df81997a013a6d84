mkdir -p gpurun_out/r3m
python -m pytest tests -m gpu -x -q > gpurun_out/r3m/pytest.log 2>&1; tail -3 gpurun_out/r3m/pytest.log
L=gpurun_out/r3m/shapes.log
echo "== version register instead of work id: defaults" > $L
for n in cornell veach-mis dragon-class teapot-class; do python3 tools/job_shapes.py $n 2>&1 | grep -v amdgpu.ids | head -0; done
SHAPES=1024:i85/85,1024:i64/64,1024:i128/128 python3 tools/job_shapes.py cornell 2>&1 | grep -v amdgpu.ids >> $L
SHAPES=4096:i341/341 python3 tools/job_shapes.py veach-mis 2>&1 | grep -v amdgpu.ids >> $L
SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
SHAPES=8192:i256/256 python3 tools/job_shapes.py teapot-class 2>&1 | grep -v amdgpu.ids >> $L
echo "== one / two more 16-byte loads of the node's own line per node visit (variants x1, x2)" >> $L
RENE_HIP_LIB=librene_hip_x1.so SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
RENE_HIP_LIB=librene_hip_x2.so SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
RENE_HIP_LIB=librene_hip_x1.so SHAPES=8192:i256/256 python3 tools/job_shapes.py teapot-class 2>&1 | grep -v amdgpu.ids >> $L
cat $L
