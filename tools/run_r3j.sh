mkdir -p gpurun_out/r3j
L=gpurun_out/r3j/w5.log
: > $L
echo "== dragon-class, 4 waves (stack 36), default build" >> $L
SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
echo "== dragon-class, 4 waves, stack 26 (LDS only)" >> $L
RENE_STACK_ENTRIES=26 SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
echo "== dragon-class, 5 waves requested (96 VGPRs, 20 B scratch), stack 36: LDS still caps at 4" >> $L
RENE_HIP_LIB=librene_hip_w5.so SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
echo "== dragon-class, 5 waves, stack 26, seeds not in LDS / in LDS" >> $L
RENE_DEBUG=1 RENE_HIP_LIB=librene_hip_w5.so RENE_STACK_ENTRIES=26 SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids | grep "co-resident\|dragon" | sort -u >> $L
RENE_HIP_LIB=librene_hip_w5.so RENE_STACK_ENTRIES=24 SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
RENE_HIP_LIB=librene_hip_w5.so RENE_STACK_ENTRIES=24 RENE_BLOCKS_PER_CU=5 SHAPES=1024:i32/32 python3 tools/job_shapes.py dragon-class 2>&1 | grep -v amdgpu.ids >> $L
cat $L
