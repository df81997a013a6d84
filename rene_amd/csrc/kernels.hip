// kernels.hip -- launch wrappers + the small kernels (trace, BSDF probe) + the render kernels of the
// small-scene family.  The BVH family (traversal-restart and while-while kernels) is instantiated in
// kernels_bvh.hip.  All device code lives in device_code.inc / render_wf.inc.
#include "device_code.inc"


// -------------------------------------------------------------------------------------------------
// batch closest-hit queries (rene_trace): one lane per ray
// -------------------------------------------------------------------------------------------------
template <bool SMALL>
__global__ void __launch_bounds__(BLOCK) trace_kernel(SceneView S, int which, uint32_t n, const float* o3,
                                                      const float* d3, float tmin, float tmax, rene_hit* out) {
  extern __shared__ uint32_t s_stack[];
  uint32_t* stack = s_stack + threadIdx.x;
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;  // keep the wave converged for the coherent item loop
  LaneCounters lc;
  f3 o = mk3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), d = mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
  const Accel& A = which ? S.emit : S.main;
  HitRec h = trace_accel<SMALL, false, true, false>(A, S.spheres, o, d, tmin, tmax, stack, lc);
  if (!live) return;
  rene_hit r;
  if (h.slot == 0xffffffffu) {
    r.t = -1.0f; r.u = 0.0f; r.v = 0.0f; r.instance = 0; r.primitive = 0;
  } else {
    const float* q = A.isect[h.slot].q;
    r.t = h.t; r.u = h.u; r.v = h.v;
    r.instance = __float_as_uint(q[9]);
    r.primitive = __float_as_uint(q[10]);
  }
  out[i] = r;
}

// -------------------------------------------------------------------------------------------------
// emitter-pdf probe (rene_emitter_pdf): the query of lib.rs:301-318 for a batch of rays, one lane per ray
// -------------------------------------------------------------------------------------------------
template <bool SMALL>
__global__ void __launch_bounds__(BLOCK) emitter_pdf_kernel(SceneView S, uint32_t n, const float* o3, const float* d3, float* out) {
  extern __shared__ uint32_t s_stack[];
  uint32_t* stack = s_stack + threadIdx.x;
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;  // keep the wave converged for the coherent item loop
  LaneCounters lc;
  f3 o = mk3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), d = mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
  HitRec h = trace_accel<SMALL, false, true, false>(S.emit, S.spheres, o, d, 0.001f, 100000.0f, stack, lc);
  const float pdf = emitter_pdf<true>(S, h, o, d);
  if (live) out[i] = pdf;
}
hipError_t launch_emitter_pdf(const LaunchConfig& cfg, const SceneView& S, uint32_t n, const float* o, const float* d, float* out, hipStream_t st) {
  size_t lds = (size_t)cfg.stack_depth * BLOCK * sizeof(uint32_t);
  dim3 grid((n + BLOCK - 1) / BLOCK), block(BLOCK);
  if (cfg.features & FEAT_SMALL) hipLaunchKernelGGL(emitter_pdf_kernel<true>, grid, block, 0, st, S, n, o, d, out);
  else hipLaunchKernelGGL(emitter_pdf_kernel<false>, grid, block, lds, st, S, n, o, d, out);
  return hipGetLastError();
}

// PCG32si on the device (rene_pcg_probe): one lane, n outputs
__global__ void pcg_probe_kernel(uint32_t seed, uint32_t n, uint32_t* out) {
  Pcg r = pcg_new(seed);
  for (uint32_t k = 0; k < n; ++k) out[k] = pcg_u32(r);
}
hipError_t launch_pcg_probe(uint32_t seed, uint32_t n, uint32_t* out, hipStream_t st) {
  hipLaunchKernelGGL(pcg_probe_kernel, dim3(1), dim3(1), 0, st, seed, n, out);
  return hipGetLastError();
}

// frame chains (device_scene.h, CHAINS): the image a call hands out = the chains' images added in chain order, ((c0 + c1) + c2) + ... -- the chains
// themselves are left as they are (they go on accumulating across launches); the alpha channel of the output is 0 (lib.rs:170 never writes it;
// in the chains' images it holds the records' versions); 16 bytes per thread
__global__ void __launch_bounds__(BLOCK) resolve_chains_kernel(const float4* chains, float4* out, size_t n4) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n4) return;
  float4 a = chains[i];
#pragma unroll
  for (uint32_t g = 1; g < CHAINS; ++g) {
    const float4 b = chains[(size_t)g * n4 + i];
    a.x += b.x;
    a.y += b.y;
    a.z += b.z;
  }
  a.w = 0.0f;
  out[i] = a;
}
hipError_t launch_resolve_chains(const float* chains, float* out, size_t image_floats, hipStream_t st) {
  const size_t n4 = image_floats / 4;
  if (n4 == 0) return hipSuccess;
  hipLaunchKernelGGL(resolve_chains_kernel, dim3((unsigned)((n4 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, reinterpret_cast<const float4*>(chains),
                     reinterpret_cast<float4*>(out), n4);
  return hipGetLastError();
}

// The same for a tile shard, over the owned tiles only (a rank of an 8-GPU job owns an eighth of the image: adding -- and, ZERO = true, clearing --
// all eight whole-image chains was a third of a millisecond of a 6 ms share): one thread per (owned tile, layer, texel)
template <bool ZERO>
__global__ void __launch_bounds__(BLOCK) chains_tiles_kernel(float4* chains, float4* out, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_owned,
                                                             uint32_t shard_rank, uint32_t shard_count) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;  // ((k * 3 + layer) * 32 + ty) * 32 + tx
  if (i >= (size_t)n_owned * 3u * RENE_TILE_SIZE * RENE_TILE_SIZE) return;
  const uint32_t tx = (uint32_t)(i & 31u), ty = (uint32_t)((i >> 5) & 31u);
  const uint32_t kl = (uint32_t)(i >> 10), layer = kl % 3u, k = kl / 3u;
  const uint32_t tile = shard_rank + k * shard_count;
  const uint32_t x = (tile % tiles_x) * RENE_TILE_SIZE + tx, y = (tile / tiles_x) * RENE_TILE_SIZE + ty;
  if (x >= W || y >= H) return;
  const size_t at = ((size_t)layer * H + y) * W + x, n4 = (size_t)3 * W * H;
  if (ZERO) {
#pragma unroll
    for (uint32_t g = 0; g < CHAINS; ++g) chains[(size_t)g * n4 + at] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  float4 a = chains[at];
#pragma unroll
  for (uint32_t g = 1; g < CHAINS; ++g) {
    const float4 b = chains[(size_t)g * n4 + at];
    a.x += b.x;
    a.y += b.y;
    a.z += b.z;
  }
  a.w = 0.0f;
  out[at] = a;
}
hipError_t launch_chains_tiles(float* chains, float* out, bool zero, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles,
                               uint32_t shard_rank, uint32_t shard_count, hipStream_t st) {
  const uint32_t n_owned = n_tiles > shard_rank ? (n_tiles - shard_rank + shard_count - 1) / shard_count : 0;
  if (n_owned == 0) return hipSuccess;
  const size_t n = (size_t)n_owned * 3u * RENE_TILE_SIZE * RENE_TILE_SIZE;
  const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK));
  if (zero) hipLaunchKernelGGL(chains_tiles_kernel<true>, grid, dim3(BLOCK), 0, st, reinterpret_cast<float4*>(chains), reinterpret_cast<float4*>(out), width, height, tiles_x, n_owned, shard_rank, shard_count);
  else hipLaunchKernelGGL(chains_tiles_kernel<false>, grid, dim3(BLOCK), 0, st, reinterpret_cast<float4*>(chains), reinterpret_cast<float4*>(out), width, height, tiles_x, n_owned, shard_rank, shard_count);
  return hipGetLastError();
}

// owned tiles <-> packed buffer (rene_gather_tiles): one thread per (owned tile, layer, texel), 16 bytes each
__global__ void __launch_bounds__(BLOCK) pack_tiles_kernel(float* fb, float* packed, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_owned,
                                                            uint32_t shard_rank, uint32_t shard_count, bool unpack) {
  const size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;  // ((k * 3 + layer) * 32 + ty) * 32 + tx
  if (i >= (size_t)n_owned * 3u * RENE_TILE_SIZE * RENE_TILE_SIZE) return;
  const uint32_t tx = (uint32_t)(i & 31u), ty = (uint32_t)((i >> 5) & 31u);
  const uint32_t kl = (uint32_t)(i >> 10), layer = kl % 3u, k = kl / 3u;
  const uint32_t tile = shard_rank + k * shard_count;
  const uint32_t x = (tile % tiles_x) * RENE_TILE_SIZE + tx, y = (tile / tiles_x) * RENE_TILE_SIZE + ty;
  float4* p = reinterpret_cast<float4*>(packed) + i;
  if (x < W && y < H) {
    float4* f = reinterpret_cast<float4*>(fb) + ((size_t)layer * H + y) * W + x;
    if (unpack) *f = *p;
    else *p = *f;
  } else if (!unpack) {
    *p = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
hipError_t launch_pack_tiles(const float* fb, float* packed, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles,
                             uint32_t shard_rank, uint32_t shard_count, bool unpack, hipStream_t st) {
  const uint32_t n_owned = n_tiles > shard_rank ? (n_tiles - shard_rank + shard_count - 1) / shard_count : 0;
  if (n_owned == 0) return hipSuccess;
  const size_t n = (size_t)n_owned * 3u * RENE_TILE_SIZE * RENE_TILE_SIZE;
  hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, const_cast<float*>(fb), packed, width, height, tiles_x,
                     n_owned, shard_rank, shard_count, unpack);
  return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// per-function BSDF probe (rene_bsdf_eval): builds the material's lobes at (normal, uv) exactly like
// the integrator does and returns f(wo,wi), pdf(wo,wi) and one sample_f(wo) drawn from
// PCG32si::new(seed).  out[12] = f.xyz, pdf, s.wi.xyz, s.f.xyz, s.pdf, len
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) bsdf_eval_kernel(SceneView S, uint32_t material, int inst_index, uint32_t n, const float* nrm3,
                                                       const float* uv2_, const float* wo3, const float* wi3,
                                                       const uint32_t* seeds, float* out) {
  constexpr uint32_t ALL = FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_LIGHTS | FEAT_BACKGROUND | FEAT_MULTI_LOBE;
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Inst inst;
  if (inst_index >= 0) {
    inst = S.insts[inst_index];
  } else {
    inst.material = material;
    inst.area_light = 0;
    inst.primitive_count = 1.0f;
    inst.material_type = S.materials[material].type;
    inst.kd[0] = inst.kd[1] = inst.kd[2] = inst.kd[3] = 0.0f;
    inst.emit[0] = inst.emit[1] = inst.emit[2] = inst.emit[3] = 0.0f;
    inst.res_type = 0u;
  }
  f3 normal = normalize(mk3(nrm3[3 * i], nrm3[3 * i + 1], nrm3[3 * i + 2]));
  Bsdf<5> b;
  b.len = 0;
  b.codes = 0xffffffffu;
  b.ng = normal;
  b.onb = onb_from_w(normal);
  compute_bsdf<ALL, 5>(S, inst, uv2{uv2_[2 * i], uv2_[2 * i + 1]}, b);
  f3 wo = mk3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), wi = mk3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]);
  constexpr uint32_t KINDS = lobe_kinds(ALL);  // every lobe kind
  f3 f = bsdf_f<5, KINDS>(b, wo, wi);
  float p = b.len ? bsdf_pdf<5, KINDS>(b, wo, wi) : 0.0f;
  Pcg rng = pcg_new(seeds[i]);
  Sampled sf = bsdf_sample<5, KINDS>(b, wo, rng);
  float* o = out + 12 * (size_t)i;
  o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = p;
  o[4] = sf.wi.x; o[5] = sf.wi.y; o[6] = sf.wi.z;
  o[7] = sf.f.x; o[8] = sf.f.y; o[9] = sf.f.z; o[10] = sf.pdf; o[11] = (float)b.len;
}

hipError_t launch_bsdf_eval(const SceneView& S, uint32_t material, int inst_index, uint32_t n, const float* nrm3, const float* uv,
                            const float* wo3, const float* wi3, const uint32_t* seeds, float* out, hipStream_t st) {
  hipLaunchKernelGGL(bsdf_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, st, S, material, inst_index, n, nrm3, uv, wo3, wi3, seeds, out);
  return hipGetLastError();
}

// per-function medium probe (rene_medium_eval), layout in include/rene_hip.h
__global__ void __launch_bounds__(64) medium_eval_kernel(SceneView S, uint32_t medium, uint32_t n, const float* rd3,
                                                         const float* t_max, const float* wo3, const float* wi3,
                                                         const uint32_t* seeds, float* out) {
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  MediumRec m = load_medium(S, medium);
  f3 rd = mk3(rd3[3 * i], rd3[3 * i + 1], rd3[3 * i + 2]);
  f3 wo = mk3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), wi = mk3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]);
  float* o = out + 16 * (size_t)i;
  const bool vac = m.type == RENE_MEDIUM_VACUUM;  // EnumMedium dispatch, medium.rs:180-208
  f3 tr = vac ? splat(1.0f) : medium_tr(m, rd, t_max[i]);
  Pcg rng = pcg_new(seeds[i]);
  SampledMedium sm{false, splat(0.0f), splat(1.0f)};
  if (!vac) sm = medium_sample(m, splat(0.0f), rd, t_max[i], rng);
  f3 p = vac ? splat(0.0f) : medium_sample_p(m, wo, rng);
  o[0] = tr.x; o[1] = tr.y; o[2] = tr.z; o[3] = vac ? 0.0f : medium_phase(m, wo, wi);
  o[4] = sm.sampled ? 1.0f : 0.0f; o[5] = sm.position.x; o[6] = sm.position.y; o[7] = sm.position.z;
  o[8] = sm.tr.x; o[9] = sm.tr.y; o[10] = sm.tr.z; o[11] = p.x; o[12] = p.y; o[13] = p.z;
  o[14] = __uint_as_float(pcg_u32(rng)); o[15] = 0.0f;
}

hipError_t launch_medium_eval(const SceneView& S, uint32_t medium, uint32_t n, const float* rd3, const float* t_max,
                              const float* wo3, const float* wi3, const uint32_t* seeds, float* out, hipStream_t st) {
  hipLaunchKernelGGL(medium_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, st, S, medium, n, rd3, t_max, wo3, wi3, seeds, out);
  return hipGetLastError();
}

// =================================================================================================
// host-side dispatch
// =================================================================================================
// small-scene family (wave-coherent item loop): instantiated in this translation unit
template <uint32_t FEAT, int MAXL>
static hipError_t launch_small(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P0, hipStream_t st) {
  static_assert(FEAT & FEAT_SMALL, "small family only");
  dim3 grid(cfg.grid), block(BLOCK);
  bool count = (P0.flags & RENE_FLAG_COUNTERS) != 0, aov = !(P0.flags & RENE_FLAG_NO_AOV);
  RenderParams P = P0;
  auto kernel = count ? render_kernel<FEAT, MAXL, true, true> : (aov ? render_kernel<FEAT, MAXL, false, true> : render_kernel<FEAT, MAXL, false, false>);
  static const size_t lds_pad = std::getenv("RENE_LDS_PAD") ? (size_t)std::atoi(std::getenv("RENE_LDS_PAD")) : 0;  // occupancy experiments
  // the scene's LDS image (device_scene.h) + the launch's seed tables
  size_t lds = (size_t)S.small_bytes;
  seed_tables_place(P, lds);
  lds += lds_pad;
  fit_grid(kernel, lds, cfg, P, grid);
  hipLaunchKernelGGL(kernel, grid, block, lds, st, S, P);
  return hipGetLastError();
}

hipError_t launch_render(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st) {
  // Specialisations: Matte-only fast path (Cornell, dragon-class), general single-lobe, general
  // multi-lobe; each with the BVH traversal or, for tiny scenes, the wave-coherent item loop.
  constexpr uint32_t ALL = FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_LIGHTS | FEAT_BACKGROUND | FEAT_MULTI_LOBE;
  constexpr uint32_t GEN1 = ALL & ~FEAT_MULTI_LOBE;
  const uint32_t f = cfg.features;
  if (f & FEAT_VOLPATH) return launch_render_vol(cfg, S, P, st);   // kernels_vol.hip
  if (!(f & FEAT_SMALL)) return launch_render_bvh(cfg, S, P, st);  // kernels_bvh.hip
  if (!(f & (FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_BACKGROUND | FEAT_MULTI_LOBE | FEAT_LIGHTS)))
    return launch_small<FEAT_SMALL, 1>(cfg, S, P, st);  // Matte, triangle emitters only (Cornell)
  if (!(f & (FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_BACKGROUND | FEAT_MULTI_LOBE)))
    return launch_small<FEAT_LIGHTS | FEAT_SMALL, 1>(cfg, S, P, st);
  // general single-lobe scenes without textures, distant lights or a background (veach-mis: Matte + Metal, sphere
  // emitters); with Metal as the only general material the other lobe kinds are compiled out: 118 VGPRs, four waves
  if (!(f & (FEAT_MULTI_LOBE | FEAT_TEXTURES | FEAT_LIGHTS | FEAT_BACKGROUND))) {
    if ((f & FEAT_NO_SPECULAR) && (f & FEAT_NO_BLEND))
      return launch_small<FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_SMALL | FEAT_NO_SPECULAR | FEAT_NO_BLEND, 1>(cfg, S, P, st);
    return launch_small<FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_SMALL, 1>(cfg, S, P, st);
  }
  if (!(f & FEAT_MULTI_LOBE)) return launch_small<GEN1 | FEAT_SMALL, 1>(cfg, S, P, st);
  return launch_small<ALL | FEAT_SMALL, 5>(cfg, S, P, st);
}

hipError_t launch_trace(const LaunchConfig& cfg, const SceneView& S, int which, uint32_t n, const float* o,
                        const float* d, float tmin, float tmax, rene_hit* out, hipStream_t st) {
  size_t lds = (size_t)cfg.stack_depth * BLOCK * sizeof(uint32_t);
  dim3 grid((n + BLOCK - 1) / BLOCK), block(BLOCK);
  if (cfg.features & FEAT_SMALL) hipLaunchKernelGGL(trace_kernel<true>, grid, block, 0, st, S, which, n, o, d, tmin, tmax, out);
  else hipLaunchKernelGGL(trace_kernel<false>, grid, block, lds, st, S, which, n, o, d, tmin, tmax, out);
  return hipGetLastError();
}

int render_block_size() { return BLOCK; }

}  // namespace rene
