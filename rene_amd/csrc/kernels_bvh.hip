// kernels_bvh.hip -- the BVH family of render kernels: the traversal-restart state machine
// (render_kernel_wf, render_wf.inc) and, for A/B tests (RENE_FLAG_NO_RESTART), the plain while-while
// kernel.  Separate translation unit so that it compiles in parallel with kernels.hip.
#include "device_code.inc"  // opens namespace rene

template <uint32_t FEAT, int MAXL>
static hipError_t launch_bvh(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P0, hipStream_t st) {
  static_assert(!(FEAT & FEAT_SMALL), "BVH family only");
  size_t lds = (size_t)cfg.stack_depth * BLOCK * sizeof(uint32_t);
  dim3 grid(cfg.grid), block(BLOCK);
  bool count = (P0.flags & RENE_FLAG_COUNTERS) != 0, aov = !(P0.flags & RENE_FLAG_NO_AOV);
  RenderParams P = P0;
  auto kernel = render_kernel_wf<FEAT, MAXL, false, false>;
  // a tree of a few hundred nodes is shallow and its rays stay coherent: the plain while-while loop wins there
  // (forced-BVH Cornell 17.2 vs 10.9, veach-mis 10.8 vs 6.3, zoo 5.1 vs 3.3 Grays/s); deep trees need the restart
  // scheduling (teapot-class 4.4 vs 3.5, dragon-class 4.4 vs 2.0)
  SceneView V = S;
  V.lds_insts = 0;
  if ((P.flags & RENE_FLAG_NO_RESTART) || S.main.n_nodes <= 512u) {
    constexpr uint32_t F0 = FEAT & ~FEAT_NO_EMITTERS;  // (the while-while loop has no instantiation of its own for that bit)
    kernel = (count || aov) ? render_kernel<F0, MAXL, true, true> : render_kernel<F0, MAXL, false, false>;
  } else {
    if (count) kernel = render_kernel_wf<FEAT, MAXL, true, true>;
    else if (aov) kernel = render_kernel_wf<FEAT, MAXL, false, true>;
    // the restart kernels keep the instance records and the distant lights in LDS behind the stack when that still leaves
    // four workgroups per CU (a quarter of 160 KB each): every shaded hit reads its instance, every light loop its light
    P.stack_entries = cfg.stack_depth;
    const size_t tables = (size_t)cfg.n_insts * sizeof(Inst) + (size_t)S.lights_len * sizeof(Light);
    if (!count && aov && cfg.n_insts && lds + tables <= 40u * 1024u && !std::getenv("RENE_NO_LDS_TABLES")) {  // (the knob: A/B tests)
      kernel = render_kernel_wf<FEAT, MAXL, false, true, true>;
      V.lds_insts = cfg.n_insts;
      lds += tables;
    }
  }
  seed_tables_place(P, lds);
  fit_grid(kernel, lds, cfg, P, grid);
  hipLaunchKernelGGL(kernel, grid, block, lds, st, V, P);
  return hipGetLastError();
}

hipError_t launch_render_bvh(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st) {
  constexpr uint32_t ALL = FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_LIGHTS | FEAT_BACKGROUND | FEAT_MULTI_LOBE;
  constexpr uint32_t GEN1 = ALL & ~FEAT_MULTI_LOBE;
  const uint32_t f = cfg.features;
  if (!(f & (FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_BACKGROUND | FEAT_MULTI_LOBE)))
    return (f & FEAT_NO_EMITTERS) ? launch_bvh<FEAT_LIGHTS | FEAT_NO_EMITTERS, 1>(cfg, S, P, st)  // dragon-class: distant lights only
                                  : launch_bvh<FEAT_LIGHTS, 1>(cfg, S, P, st);
  // general single-lobe scenes without spheres and distant lights (teapot-class: Substrate + textures + environment map)
  if (!(f & (FEAT_MULTI_LOBE | FEAT_SPHERES | FEAT_LIGHTS))) {
    if ((f & FEAT_NO_SPECULAR) && (f & FEAT_NO_MICROFACET)) {  // ... and Substrate as the only general material
      constexpr uint32_t SUB = FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_BACKGROUND | FEAT_NO_SPECULAR | FEAT_NO_MICROFACET;
      return (f & FEAT_NO_EMITTERS) ? launch_bvh<SUB | FEAT_NO_EMITTERS, 1>(cfg, S, P, st)  // the teapot scenes: an environment light only
                                    : launch_bvh<SUB, 1>(cfg, S, P, st);
    }
    return launch_bvh<FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_BACKGROUND, 1>(cfg, S, P, st);
  }
  if (!(f & FEAT_MULTI_LOBE)) return launch_bvh<GEN1, 1>(cfg, S, P, st);
  return launch_bvh<ALL, 5>(cfg, S, P, st);
}

}  // namespace rene
