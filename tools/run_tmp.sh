mkdir -p gpurun_out/r3q
L=gpurun_out/r3q/short.log
: > $L
SPP=128 SHAPES=128:i8/8,128:i16/16,128:i32/32,128:i64/64,128:w python3 tools/job_shapes.py cornell 2>&1 | grep -v amdgpu.ids >> $L
SPP=256 SHAPES=256:i16/16,256:i32/32,256:i64/64,256:i128/128 python3 tools/job_shapes.py cornell 2>&1 | grep -v amdgpu.ids >> $L
SPP=512 SHAPES=512:i32/32,512:i64/64,512:i128/128 python3 tools/job_shapes.py cornell 2>&1 | grep -v amdgpu.ids >> $L
cat $L
