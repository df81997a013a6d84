mkdir -p gpurun_out/r3f
L=$PWD/gpurun_out/r3f/ab.log
: > $L
for spec in "1024 16" "1024 4" "1024 1" "256 4" "256 1"; do
  (cd _r2 && python3 ../tools/ab_cut.py $spec 2>&1 | grep -v amdgpu.ids >> $L)
  (python3 tools/ab_cut.py $spec 2>&1 | grep -v amdgpu.ids >> $L)
  (RENE_WORK_BATCH=64 python3 tools/ab_cut.py $spec 2>&1 | grep -v amdgpu.ids >> $L)
done
cat $L
