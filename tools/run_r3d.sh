mkdir -p gpurun_out/r3d
L=gpurun_out/r3d/shapes.log
echo "== batch sweep, cornell i64/64 and i128/128" > $L
for b in 128 64 32 16; do echo "-- RENE_WORK_BATCH=$b" >> $L; RENE_WORK_BATCH=$b SHAPES=1024:i64/64,1024:i128/128 python3 tools/job_shapes.py cornell 2>&1 | grep -v amdgpu.ids >> $L; done
echo "== batch sweep, veach / dragon / teapot" >> $L
for b in 64 32 16; do echo "-- RENE_WORK_BATCH=$b" >> $L; RENE_WORK_BATCH=$b SHAPES=4096:i128/128 python3 tools/job_shapes.py veach-mis 2>&1 | grep -v amdgpu.ids >> $L; RENE_WORK_BATCH=$b SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L;  RENE_WORK_BATCH=$b SHAPES=8192:i256/256 python3 tools/job_shapes.py teapot-class 2>&1 | grep -v amdgpu.ids >> $L; done
echo "== lane states (counting variants)" >> $L
RENE_WORK_BATCH=128 python3 tools/lane_states.py cornell 1024:i64/64 1024:w 2>&1 | grep -v amdgpu.ids | grep "==\|passes\|job" >> $L
RENE_WORK_BATCH=32 python3 tools/lane_states.py cornell 1024:i64/64 1024:i64/8 1024:i256/256 2>&1 | grep -v amdgpu.ids | grep "==\|passes\|job" >> $L
RENE_WORK_BATCH=32 python3 tools/lane_states.py dragon-class 1024:i32/32 2>&1 | grep -v amdgpu.ids | grep "==\|steps\|node visits\|job" >> $L
cat $L
