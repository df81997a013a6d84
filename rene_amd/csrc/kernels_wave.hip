// kernels_wave.hip -- the stage-separated wavefront integrator for BVH scenes (wavefront.inc): launch
// wrappers.  Separate translation unit so that it compiles in parallel with the megakernel families.
#include "device_code.inc"  // opens namespace rene
#include "wavefront.inc"

hipError_t launch_wave_init(const WaveState& Q, hipStream_t st) {
  hipLaunchKernelGGL(wave_init, dim3((Q.n_slots + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, Q);
  return hipGetLastError();
}

hipError_t launch_wave_finish(const RenderParams& P, const WaveState& Q, hipStream_t st) {
  const uint32_t rows = (Q.n_slots + 63u) / 64u;
  hipLaunchKernelGGL(wave_finish, dim3((rows + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, P, Q, rows);
  return hipGetLastError();
}

template <uint32_t FEAT, int MAXL>
static hipError_t wave_rounds(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, const WaveState& Q,
                              uint32_t rounds, hipStream_t st) {
  constexpr bool SPHERES = (FEAT & FEAT_SPHERES) != 0;
  const size_t lds = (size_t)cfg.wave_stack * BLOCK * sizeof(uint32_t);
  const dim3 grid((Q.n_slots + BLOCK - 1) / BLOCK), block(BLOCK);
  const bool count = (P.flags & RENE_FLAG_COUNTERS) != 0, aov = !(P.flags & RENE_FLAG_NO_AOV);
  auto extend = count ? wave_extend<SPHERES, true> : wave_extend<SPHERES, false>;
  auto connect = count ? wave_connect<SPHERES, true> : wave_connect<SPHERES, false>;
  auto shade = count ? wave_shade<FEAT, MAXL, true, true>
                     : (aov ? wave_shade<FEAT, MAXL, false, true> : wave_shade<FEAT, MAXL, false, false>);
  for (uint32_t r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(extend, grid, block, lds, st, S, P, Q);
    hipLaunchKernelGGL(shade, grid, block, 0, st, S, P, Q);
    hipLaunchKernelGGL(connect, grid, block, lds, st, S, P, Q);
  }
  return hipGetLastError();
}

hipError_t launch_wave_rounds(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, const WaveState& Q,
                              uint32_t rounds, hipStream_t st) {
  constexpr uint32_t ALL = FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_LIGHTS | FEAT_BACKGROUND | FEAT_MULTI_LOBE;
  constexpr uint32_t GEN1 = ALL & ~FEAT_MULTI_LOBE;
  const uint32_t f = cfg.features;
  if (!(f & (FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_BACKGROUND | FEAT_MULTI_LOBE)))
    return wave_rounds<FEAT_LIGHTS, 1>(cfg, S, P, Q, rounds, st);
  if (!(f & FEAT_MULTI_LOBE)) return wave_rounds<GEN1, 1>(cfg, S, P, Q, rounds, st);
  return wave_rounds<ALL, 5>(cfg, S, P, Q, rounds, st);
}

}  // namespace rene
