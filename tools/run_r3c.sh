mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py -x -q > gpurun_out/r3c/pytest.log 2>&1; tail -3 gpurun_out/r3c/pytest.log
L=gpurun_out/r3c/shapes.log
echo "== poll every 4th, batch 128" > $L
SHAPES=1024:i64/64,1024:i128/128,1024:i256/256,1024:i256/8,1024:i128/8,1024:i64/8 python3 tools/job_shapes.py cornell >> $L 2>&1
echo "== poll every 4th, batch 64" >> $L
RENE_WORK_BATCH=64 SHAPES=1024:i64/64,1024:i128/128,1024:i256/8,1024:i128/8 python3 tools/job_shapes.py cornell >> $L 2>&1
echo "== poll every pass (variant poll1), batch 128" >> $L
RENE_HIP_LIB=librene_hip_poll1.so SHAPES=1024:i64/64,1024:i128/128,1024:i256/8 python3 tools/job_shapes.py cornell >> $L 2>&1
echo "== BVH: poll every 4th" >> $L
SHAPES=1024:i32/32,1024:i32/4,1024:i64/4,1024:i16/4 python3 tools/job_shapes.py dragon-class >> $L 2>&1
SHAPES=8192:i128/8,8192:i256/8,8192:i64/4,8192:i512/8 python3 tools/job_shapes.py teapot-class >> $L 2>&1
echo "== BVH: poll every 4th, batch 64" >> $L
RENE_WORK_BATCH=64 SHAPES=1024:i32/4 python3 tools/job_shapes.py dragon-class >> $L 2>&1
RENE_WORK_BATCH=64 SHAPES=8192:i128/8 python3 tools/job_shapes.py teapot-class >> $L 2>&1
echo "== BVH: poll every pass (variant poll1)" >> $L
RENE_HIP_LIB=librene_hip_poll1.so SHAPES=1024:i32/4 python3 tools/job_shapes.py dragon-class >> $L 2>&1
RENE_HIP_LIB=librene_hip_poll1.so SHAPES=8192:i128/8 python3 tools/job_shapes.py teapot-class >> $L 2>&1
SHAPES=4096:i128/8,4096:i256/8 python3 tools/job_shapes.py veach-mis >> $L 2>&1
grep -v amdgpu.ids $L
