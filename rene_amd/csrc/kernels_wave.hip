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

// co-resident blocks of a persistent traversal pass (cached per kernel and LDS size)
template <typename Kern>
static uint32_t resident_blocks(Kern kernel, size_t lds, uint32_t cus) {
  static std::mutex mu;
  static std::unordered_map<uintptr_t, int> cache;
  const uintptr_t key = reinterpret_cast<uintptr_t>(reinterpret_cast<const void*>(kernel)) ^ (uintptr_t)(lds * 0x9E3779B97F4A7C15ull);
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(key);
  int occ = 0;
  if (it != cache.end()) occ = it->second;
  else {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, BLOCK, lds) != hipSuccess) occ = 0;
    cache[key] = occ;
    if (std::getenv("RENE_DEBUG")) std::fprintf(stderr, "[rene] traversal pass %p lds %zu: %d co-resident blocks per CU\n", reinterpret_cast<const void*>(kernel), lds, occ);
  }
  return std::max(1, occ) * std::max(1u, cus);
}

template <uint32_t FEAT, int MAXL>
static hipError_t wave_rounds(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, const WaveState& Q,
                              uint32_t rounds, hipStream_t st) {
  constexpr bool SPHERES = (FEAT & FEAT_SPHERES) != 0;
  const size_t lds = (size_t)cfg.wave_stack * BLOCK * sizeof(uint32_t);
  const dim3 grid((Q.n_slots + BLOCK - 1) / BLOCK), block(BLOCK);
  const bool count = (P.flags & RENE_FLAG_COUNTERS) != 0, aov = !(P.flags & RENE_FLAG_NO_AOV);
  auto regen = count ? wave_regen<true> : wave_regen<false>;
  auto closest = count ? wave_trace<false, SPHERES, true> : wave_trace<false, SPHERES, false>;
  auto secondary = count ? wave_trace<true, SPHERES, true> : wave_trace<true, SPHERES, false>;
  auto shade = count ? wave_shade<FEAT, MAXL, true, true>
                     : (aov ? wave_shade<FEAT, MAXL, false, true> : wave_shade<FEAT, MAXL, false, false>);
  const uint32_t n1 = Q.n_slots;
  const uint32_t n2 = (Q.max_lights + (S.emit_object_len ? 1u : 0u)) * Q.n_slots;
  auto pass_grid = [&](uint32_t resident, uint32_t ids) {
    uint32_t need = (ids + WAVE_ID_BATCH * (BLOCK / 64) - 1) / (WAVE_ID_BATCH * (BLOCK / 64));
    return std::max(1u, std::min(resident, need));
  };
  const uint32_t g1 = pass_grid(resident_blocks(closest, lds, cfg.cus), n1);
  const uint32_t g2 = n2 ? pass_grid(resident_blocks(secondary, lds, cfg.cus), n2) : 0;
  for (uint32_t r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(regen, grid, block, 0, st, S, P, Q);
    hipLaunchKernelGGL(closest, dim3(g1), block, lds, st, S, P, Q, n1, g1 * (BLOCK / 64));
    hipLaunchKernelGGL(shade, grid, block, 0, st, S, P, Q);
    if (n2) hipLaunchKernelGGL(secondary, dim3(g2), block, lds, st, S, P, Q, n2, g2 * (BLOCK / 64));
    hipLaunchKernelGGL(wave_bounce, grid, block, 0, st, S, P, Q);
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
