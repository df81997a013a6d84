// kernels_vol.hip -- render kernels of the volumetric integrator (Integrator "volpath", lib.rs:477-803):
// render_kernel with FEAT_VOLPATH over the item loop (small scenes) or the while-while BVH traversal (shallow trees), and
// render_kernel_wf with FEAT_VOLPATH (deep trees: traversal restart).
// Separate translation unit so that it compiles in parallel with the path-integrator families.
#include "device_code.inc"  // opens namespace rene

template <uint32_t FEAT, int MAXL>
static hipError_t launch_vol(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P0, hipStream_t st) {
  static_assert(FEAT & FEAT_VOLPATH, "volpath family only");
  size_t lds = (FEAT & FEAT_SMALL) ? 0 : (size_t)cfg.stack_depth * BLOCK * sizeof(uint32_t);
  dim3 grid(cfg.grid), block(BLOCK);
  bool count = (P0.flags & RENE_FLAG_COUNTERS) != 0, aov = !(P0.flags & RENE_FLAG_NO_AOV);
  RenderParams P = P0;
  auto kernel = (count || aov) ? render_kernel<FEAT, MAXL, true, true> : render_kernel<FEAT, MAXL, false, false>;
  // deep trees: the traversal-restart scheduling of render_wf.inc, its walks (tr / tr_emit) as phases of the state machine
  // (shallow ones -- a few hundred nodes -- stay with the while-while loop, as in kernels_bvh.hip); RENE_FLAG_NO_RESTART: A/B tests
  if constexpr (!(FEAT & FEAT_SMALL)) {
    if (!(P.flags & RENE_FLAG_NO_RESTART) && S.main.n_nodes > 512u) {
      kernel = count ? render_kernel_wf<FEAT, MAXL, true, true, false> : (aov ? render_kernel_wf<FEAT, MAXL, false, true, false> : render_kernel_wf<FEAT, MAXL, false, false, false>);
      P.stack_entries = cfg.stack_depth;
    }
  }
  seed_tables_place(P, lds);
  fit_grid(kernel, lds, cfg, P, grid);
  SceneView V = S;
  V.lds_insts = 0;
  hipLaunchKernelGGL(kernel, grid, block, lds, st, V, P);
  return hipGetLastError();
}

hipError_t launch_render_vol(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st) {
  constexpr uint32_t ALL = FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_LIGHTS | FEAT_BACKGROUND | FEAT_MULTI_LOBE | FEAT_VOLPATH;
  constexpr uint32_t GEN1 = ALL & ~FEAT_MULTI_LOBE;
  const uint32_t f = cfg.features;
  const bool small = (f & FEAT_SMALL) != 0;
  if (!(f & (FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_BACKGROUND | FEAT_MULTI_LOBE)))
    return small ? launch_vol<FEAT_LIGHTS | FEAT_VOLPATH | FEAT_SMALL, 1>(cfg, S, P, st)
                 : launch_vol<FEAT_LIGHTS | FEAT_VOLPATH, 1>(cfg, S, P, st);
  if (!(f & FEAT_MULTI_LOBE))
    return small ? launch_vol<GEN1 | FEAT_SMALL, 1>(cfg, S, P, st) : launch_vol<GEN1, 1>(cfg, S, P, st);
  return small ? launch_vol<ALL | FEAT_SMALL, 5>(cfg, S, P, st) : launch_vol<ALL, 5>(cfg, S, P, st);
}

}  // namespace rene
