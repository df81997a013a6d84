mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_edge.py -x -q > gpurun_out/r3e/pytest.log 2>&1; tail -5 gpurun_out/r3e/pytest.log
L=gpurun_out/r3e/shapes.log
echo "== shadow fast path: dragon-class, batch 64 / 128" > $L
RENE_WORK_BATCH=64 SHAPES=1024:i32/32,1024:i64/64,1024:i16/16 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
echo "== cornell item sweep, batch 64" >> $L
RENE_WORK_BATCH=64 SHAPES=1024:i32/32,1024:i48/48,1024:i64/64,1024:i96/96,1024:i128/128,1024:i192/192,1024:i256/256,1024:i341/341,1024:i512/512,1024:w,1024:i128/8,1024:i256/8,1024:i341/8,1024:i512/8 python3 tools/job_shapes.py cornell 2>&1 | grep -v amdgpu.ids >> $L
echo "== lane states dragon (fast path)" >> $L
RENE_WORK_BATCH=64 python3 tools/lane_states.py dragon-class 1024:i32/32 2>&1 | grep -v amdgpu.ids | grep "==\|steps\|node visits\|job" >> $L
RENE_WORK_BATCH=64 python3 tools/lane_states.py teapot-class 8192:i256/256 2>&1 | grep -v amdgpu.ids | grep "==\|steps\|node visits\|job" >> $L
cat $L
