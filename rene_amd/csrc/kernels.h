// kernels.h -- host-visible launch wrappers of kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rene_hip.h"
#include "device_scene.h"

namespace rene {

struct LaunchConfig {
  uint32_t features = 0;     // FEAT_* of the scene
  uint32_t stack_depth = 16; // LDS traversal stack entries per lane
  uint32_t grid = 0;         // workgroups of the persistent render launch (upper bound)
  uint32_t cus = 0;          // compute units of the device
};

hipError_t launch_render(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st);
hipError_t launch_render_bvh(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st);
hipError_t launch_render_vol(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st);
hipError_t launch_trace(const LaunchConfig& cfg, const SceneView& S, int which, uint32_t n, const float* o,
                        const float* d, float tmin, float tmax, rene_hit* out, hipStream_t st);
hipError_t launch_bsdf_eval(const SceneView& S, uint32_t material, uint32_t n, const float* nrm3, const float* uv,
                            const float* wo3, const float* wi3, const uint32_t* seeds, float* out, hipStream_t st);
hipError_t launch_medium_eval(const SceneView& S, uint32_t medium, uint32_t n, const float* rd3, const float* t_max,
                              const float* wo3, const float* wi3, const uint32_t* seeds, float* out, hipStream_t st);
int render_block_size();

}  // namespace rene
