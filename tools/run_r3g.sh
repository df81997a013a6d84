mkdir -p gpurun_out/r3g
L=$PWD/gpurun_out/r3g/ab.log
: > $L
for spec in "1024 16" "1024 1"; do
  (cd _r2 && python3 ../tools/ab_cut.py $spec 2>&1 | grep -v amdgpu.ids >> $L)
  (python3 tools/ab_cut.py $spec 2>&1 | grep -v amdgpu.ids >> $L)
  (RENE_WORK_BATCH=64 python3 tools/ab_cut.py $spec 2>&1 | grep -v amdgpu.ids >> $L)
done
RENE_WORK_BATCH=64 SHAPES=1024:i64/64,1024:i96/96,1024:i128/128,1024:i256/256 python3 tools/job_shapes.py cornell 2>&1 | grep -v amdgpu.ids >> $L
RENE_WORK_BATCH=64 SHAPES=4096:i128/128,4096:i256/256 python3 tools/job_shapes.py veach-mis 2>&1 | grep -v amdgpu.ids >> $L
cat $L
python -m pytest tests/test_gpu_t2.py -x -q -s > gpurun_out/r3g/t2.log 2>&1; grep -v "^$" gpurun_out/r3g/t2.log | tail -60
python -m pytest tests -m gpu -x -q > gpurun_out/r3g/pytest.log 2>&1; tail -3 gpurun_out/r3g/pytest.log
