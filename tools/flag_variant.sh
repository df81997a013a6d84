#!/bin/bash
# tools/flag_variant.sh NAME "EXTRA HIPCC FLAGS": librene_hip_NAME.so with the kernel units compiled under extra compiler flags (host objects shared
# with the product build) -- A/B runs of compiler options (RENE_HIP_LIB=librene_hip_NAME.so)
cd "$(dirname "$0")/../rene_amd/csrc" || exit 1
N=$1; E=$2
F="-O3 -std=c++17 -fPIC -ffp-contract=on -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 -Wno-everything"
mkdir -p var_$N
for u in kernels kernels_bvh kernels_vol kernels_wave; do /opt/rocm/bin/hipcc $F $E -c -o var_$N/$u.o $u.hip & done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o librene_hip_$N.so var_$N/kernels.o var_$N/kernels_bvh.o var_$N/kernels_vol.o var_$N/kernels_wave.o rene_hip.o scene_pack.o pbrt_loader.o loop_subdiv.o image_io.o -lz
