// kernels.hip -- the render hot path on gfx950 (CDNA4): ray generation, BVH traversal +
// intersection, hit shading / BSDF sampling, rene's one-sample light/BSDF mixture ("NEE") and
// radiance accumulation.
//
// Reference statement order: rene-shader/src/lib.rs:141-357 (main_ray_generation_path); hit
// shaders lib.rs:805-1066; BxDFs reflection/bxdf.rs; microfacet.rs; fresnel.rs; material.rs;
// texture.rs; light.rs; area_light.rs; surface_sample.rs; camera.rs; rand.rs; math.rs.
//
// Execution model (MI355X-first, not Vulkan's raygen/hit/miss shader split):
//   * one persistent launch renders N frames; a lane owns one pixel at a time and keeps its three
//     accumulation sums in registers for all N frames (one framebuffer read + write per pixel per
//     launch instead of rene's read-modify-write per add_image, lib.rs:165-172);
//   * path regeneration: a lane whose path ended starts the next frame of its pixel (or fetches a
//     new pixel from a wave-aggregated atomic work counter) while its neighbours keep bouncing, so
//     the 64-wide wavefront stays full although path lengths range from 1 to 50 bounces;
//   * sums are added in exactly the reference's order (frame by frame, bounce by bounce), so a
//     pixel's result does not depend on scheduling, on the number of launches or on GPU count.
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_scene.h"
#include "kernels.h"

namespace rene {

constexpr int BLOCK = 256;

// =================================================================================================
// traversal
// =================================================================================================
struct HitRec {
  float t, u, v;
  uint32_t slot;  // 0xffffffff = miss
};

struct LaneCounters {
  uint32_t closest = 0, shadow = 0, emitter = 0, paths = 0, hits = 0, adds = 0, nodes = 0, prims = 0;
};

RENE_DEV float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// slab test of one child box; returns entry distance in tn
RENE_DEV bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz, f3 o, f3 inv,
                   float tmin, float tmax, float& tn) {
  float t0 = (lox - o.x) * inv.x, t1 = (hix - o.x) * inv.x;
  float lo = fminf(t0, t1), hi = fmaxf(t0, t1);
  t0 = (loy - o.y) * inv.y;
  t1 = (hiy - o.y) * inv.y;
  lo = fmaxf(lo, fminf(t0, t1));
  hi = fminf(hi, fmaxf(t0, t1));
  t0 = (loz - o.z) * inv.z;
  t1 = (hiz - o.z) * inv.z;
  lo = fmaxf(lo, fminf(t0, t1));
  hi = fminf(hi, fmaxf(t0, t1));
  lo = fmaxf(lo, tmin);
  hi = fminf(hi, tmax);
  tn = lo;
  // a few ulp of slack: a box must never reject a hit its primitive accepts
  return lo * 0.999998f <= hi * 1.000002f;
}

// Moeller-Trumbore, no culling, tmin <= t, closest wins (first found wins ties)
RENE_DEV void intersect_triangle(const PrimIsect* isect, uint32_t slot, f3 o, f3 d, float tmin,
                                 HitRec& best, float& tmax) {
  const float* q = isect[slot].q;
  float4 a = ldg4(q), b = ldg4(q + 4), c = ldg4(q + 8);
  f3 p0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
  f3 pv = cross(d, e2);
  float det = dot(e1, pv);
  if (det == 0.0f) return;
  float inv_det = 1.0f / det;
  f3 tv = o - p0;
  float u = dot(tv, pv) * inv_det;
  if (u < 0.0f || u > 1.0f) return;
  f3 qv = cross(tv, e1);
  float v = dot(d, qv) * inv_det;
  if (v < 0.0f || u + v > 1.0f) return;
  float t = dot(e2, qv) * inv_det;
  bool accept = t >= tmin && (best.slot == 0xffffffffu ? t <= tmax : t < tmax);
  if (accept) {
    tmax = t;
    best.t = t;
    best.u = u;
    best.v = v;
    best.slot = slot;
  }
}

// sphere_intersection, rene-shader/src/lib.rs:805-839: unit sphere in object space
RENE_DEV void intersect_sphere(const PrimIsect* isect, const Sphere* spheres, uint32_t slot, f3 o, f3 d,
                               float tmin, HitRec& best, float& tmax) {
  uint32_t sidx = __float_as_uint(isect[slot].q[11]);
  const float* w2o = spheres[sidx].w2o;
  f3 oc = aff_point(w2o, o);
  f3 od = aff_vector(w2o, d);
  float a = length_squared(od);
  float half_b = dot(oc, od);
  float c = length_squared(oc) - 1.0f;
  float disc = half_b * half_b - a * c;
  if (disc < 0.0f) return;
  float sq = sqrtf(disc);
  float root0 = (-half_b - sq) / a;
  float root1 = (-half_b + sq) / a;
  float r;
  if (root0 >= tmin && root0 <= tmax) r = root0;
  else if (root1 >= tmin && root1 <= tmax) r = root1;
  else return;
  tmax = r;
  best.t = r;
  best.u = 0.0f;
  best.v = 0.0f;
  best.slot = slot;
}

// While-while BVH2 traversal with a per-lane stack in LDS ([depth][lane] so that a push/pop is
// conflict-free).  ANY: occlusion query, returns at the first accepted hit.
template <bool ANY, bool SPHERES, bool COUNT>
RENE_DEV HitRec traverse(const Accel& A, const Sphere* spheres, f3 o, f3 d, float tmin, float tmax,
                         uint32_t* stack, LaneCounters& lc) {
  HitRec best;
  best.t = 0.0f;
  best.u = 0.0f;
  best.v = 0.0f;
  best.slot = 0xffffffffu;
  const f3 inv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
  int sp = 0;
  uint32_t cur = 0;  // root is always an inner node
  for (;;) {
    // ---- inner nodes ----
    while (!(cur & LEAF_BIT)) {
      const float* q = A.nodes[cur].q;
      float4 q0 = ldg4(q), q1 = ldg4(q + 4), q2 = ldg4(q + 8), q3 = ldg4(q + 12);
      if (COUNT) lc.nodes++;
      float t0, t1;
      bool h0 = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, tmin, tmax, t0);
      bool h1 = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, tmin, tmax, t1);
      uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
      if (h0 && h1) {
        if (t1 < t0) {
          uint32_t tmp = c0;
          c0 = c1;
          c1 = tmp;
        }
        stack[sp * BLOCK] = c1;
        sp++;
        cur = c0;
      } else if (h0) {
        cur = c0;
      } else if (h1) {
        cur = c1;
      } else {
        if (sp == 0) return best;
        sp--;
        cur = stack[sp * BLOCK];
      }
    }
    // ---- leaf ----
    {
      uint32_t first = cur & LEAF_FIRST_MASK;
      uint32_t count = ((cur >> LEAF_COUNT_SHIFT) & 15u) + 1u;
      if (SPHERES && (cur & SPHERE_BIT)) {
        if (COUNT) lc.prims++;
        intersect_sphere(A.isect, spheres, first, o, d, tmin, best, tmax);
      } else {
        for (uint32_t k = 0; k < count; ++k) {
          if (COUNT) lc.prims++;
          intersect_triangle(A.isect, first + k, o, d, tmin, best, tmax);
        }
      }
      if (ANY && best.slot != 0xffffffffu) return best;
      if (sp == 0) return best;
      sp--;
      cur = stack[sp * BLOCK];
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Small scenes: wave-coherent loop over every item (device_scene.h, SmallItem).  The item index is
// wave-uniform, so the 64-byte record is fetched once per wave through the scalar cache (s_load)
// and its fields are SGPR operands of the per-lane arithmetic; there is no stack, no divergent
// control flow and no vector memory traffic.  A parallelogram item covers two triangles.
// -------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) float* cfloat_ptr;  // constant address space -> SMEM loads

template <bool ANY, bool SPHERES, bool COUNT>
RENE_DEV HitRec traverse_small(const Accel& A, const Sphere* spheres, f3 o, f3 d, float tmin, float tmax,
                               LaneCounters& lc) {
  float best_t = tmax, best_s = 0.0f, best_r = 0.0f;
  uint32_t best_item = 0xffffffffu;
  const uint32_t n = A.n_items;
  cfloat_ptr base = (cfloat_ptr)(const void*)A.items;
  for (uint32_t k = 0; k < n; ++k) {
    cfloat_ptr q = base + 16 * k;
    const uint32_t kind = __float_as_uint(q[9]);
    if (COUNT) lc.prims++;
    if (SPHERES && kind == SMALL_SPHERE) {  // sphere_intersection, lib.rs:805-839
      uint32_t slot = __float_as_uint(q[10]);
      uint32_t sidx = __float_as_uint(A.isect[slot].q[11]);
      const float* w2o = spheres[sidx].w2o;
      f3 oc = aff_point(w2o, o);
      f3 od = aff_vector(w2o, d);
      float a = length_squared(od);
      float half_b = dot(oc, od);
      float c = length_squared(oc) - 1.0f;
      float disc = half_b * half_b - a * c;
      if (disc >= 0.0f) {
        float sq = sqrtf(disc);
        float root0 = (-half_b - sq) / a;
        float root1 = (-half_b + sq) / a;
        float r = -1.0f;
        if (root0 >= tmin && root0 <= best_t) r = root0;
        else if (root1 >= tmin && root1 <= best_t) r = root1;
        if (r >= tmin && (best_item == 0xffffffffu ? r <= best_t : r < best_t)) {
          best_t = r;
          best_s = 0.0f;
          best_r = 0.0f;
          best_item = k;
        }
      }
      continue;
    }
    f3 O = mk3(q[0], q[1], q[2]), ea = mk3(q[3], q[4], q[5]), eb = mk3(q[6], q[7], q[8]);
    f3 pv = cross(d, eb);
    float det = dot(ea, pv);
    float inv_det = __builtin_amdgcn_rcpf(det);  // 1 ulp; det == 0 -> inf -> rejected below
    f3 tv = o - O;
    float s = dot(tv, pv) * inv_det;
    f3 qv = cross(tv, ea);
    float r = dot(d, qv) * inv_det;
    float t = dot(eb, qv) * inv_det;
    // triangle: s, r >= 0 and s + r <= 1; parallelogram: s, r in [0, 1].  det == 0 gives inf/nan -> rejected
    bool inside = s >= 0.0f && r >= 0.0f && (kind == SMALL_QUAD ? (s <= 1.0f && r <= 1.0f) : (s + r <= 1.0f));
    bool accept = inside && t >= tmin && (best_item == 0xffffffffu ? t <= best_t : t < best_t);
    if (accept) {
      best_t = t;
      best_s = s;
      best_r = r;
      best_item = k;
    }
  }
  HitRec h;
  h.t = best_t;
  h.u = 0.0f;
  h.v = 0.0f;
  h.slot = 0xffffffffu;
  if (best_item != 0xffffffffu) {
    // map (item, s, r) back to (triangle slot, barycentric u, v) of the BVH path's conventions
    const float* q = A.items[best_item].q;
    float4 m = ldg4(q + 8);   // b.z, kind, slot1, slot2
    float4 pm = ldg4(q + 12);  // perm1, perm2
    bool second = __float_as_uint(m.y) == SMALL_QUAD && best_s + best_r > 1.0f;
    uint32_t perm = __float_as_uint(second ? pm.y : pm.x);
    float w0 = second ? best_s + best_r - 1.0f : 1.0f - best_s - best_r;
    float w1 = second ? 1.0f - best_r : best_s;
    float w2 = second ? 1.0f - best_s : best_r;
    uint32_t iu = perm & 3u, iv = (perm >> 2) & 3u;
    h.u = iu == 0u ? w0 : (iu == 1u ? w1 : w2);
    h.v = iv == 0u ? w0 : (iv == 1u ? w1 : w2);
    h.slot = __float_as_uint(second ? m.w : m.z);
  }
  return h;
}

template <bool SMALL, bool ANY, bool SPHERES, bool COUNT>
RENE_DEV HitRec trace_accel(const Accel& A, const Sphere* spheres, f3 o, f3 d, float tmin, float tmax,
                            uint32_t* stack, LaneCounters& lc) {
  if (SMALL) return traverse_small<ANY, SPHERES, COUNT>(A, spheres, o, d, tmin, tmax, lc);
  return traverse<ANY, SPHERES, COUNT>(A, spheres, o, d, tmin, tmax, stack, lc);
}

// =================================================================================================
// textures / materials / BSDF
// =================================================================================================
struct uv2 {
  float x, y;
};

// asm.rs:26-46 (OpConvertFToU), saturating where SPIR-V is undefined
RENE_DEV uint32_t f32_to_u32(float v) {
  if (!(v > 0.0f)) return 0u;
  if (v >= 4294967296.0f) return 0xffffffffu;
  return (uint32_t)v;
}
RENE_DEV float fract(float v) { return v - floorf(v); }  // GLSL Fract, asm.rs:54-69

// bilinear, REPEAT addressing, texel centres at +0.5 (VK_FILTER_LINEAR, rene/src/main.rs:2390-2397)
RENE_DEV f3 image_sample(const SceneView& S, uint32_t img, float u, float v) {
  ImageRef im = S.images[img];
  const float* base = S.image_pool + im.offset;
  float x = u * (float)im.width - 0.5f, y = v * (float)im.height - 0.5f;
  float fx = floorf(x), fy = floorf(y);
  float ax = x - fx, ay = y - fy;
  int w = (int)im.width, h = (int)im.height;
  int x0 = (int)fx % w, y0 = (int)fy % h;
  if (x0 < 0) x0 += w;
  if (y0 < 0) y0 += h;
  int x1 = x0 + 1 == w ? 0 : x0 + 1, y1 = y0 + 1 == h ? 0 : y0 + 1;
  float4 p00 = ldg4(base + 4 * ((size_t)y0 * w + x0)), p10 = ldg4(base + 4 * ((size_t)y0 * w + x1));
  float4 p01 = ldg4(base + 4 * ((size_t)y1 * w + x0)), p11 = ldg4(base + 4 * ((size_t)y1 * w + x1));
  f3 top = mk3(p00.x, p00.y, p00.z) * (1.0f - ax) + mk3(p10.x, p10.y, p10.z) * ax;
  f3 bot = mk3(p01.x, p01.y, p01.z) * (1.0f - ax) + mk3(p11.x, p11.y, p11.z) * ax;
  return top * (1.0f - ay) + bot * ay;
}

// texture.rs:175-190
template <uint32_t FEAT>
RENE_DEV f3 tex_color_non_recursive(const SceneView& S, uint32_t index, uv2 uv) {
  const Texture& t = S.textures[index];
  if (!(FEAT & FEAT_TEXTURES)) return mk3(t.v0[0], t.v0[1], t.v0[2]);
  uint32_t type = t.type;
  if (type == RENE_TEXTURE_SOLID) return mk3(t.v0[0], t.v0[1], t.v0[2]);
  if (type == RENE_TEXTURE_IMAGEMAP) return image_sample(S, t.u0[0], uv.x, 1.0f - uv.y);  // texture.rs:121-127
  return splat(1.0f);
}
// texture.rs:192-211
template <uint32_t FEAT>
RENE_DEV f3 tex_color(const SceneView& S, uint32_t index, uv2 uv) {
  const Texture& t = S.textures[index];
  if (!(FEAT & FEAT_TEXTURES)) return mk3(t.v0[0], t.v0[1], t.v0[2]);
  uint32_t type = t.type;
  if (type == RENE_TEXTURE_SOLID) return mk3(t.v0[0], t.v0[1], t.v0[2]);
  if (type == RENE_TEXTURE_IMAGEMAP) return image_sample(S, t.u0[0], uv.x, 1.0f - uv.y);
  if (type == RENE_TEXTURE_CHECKERBOARD) {  // texture.rs:97-118
    float x = uv.x * t.v0[0], y = uv.y * t.v0[1];
    uint32_t idx = ((f32_to_u32(x) % 2u == 0u) == (f32_to_u32(y) % 2u == 0u)) ? t.u0[0] : t.u0[1];
    return tex_color_non_recursive<FEAT>(S, idx, uv2{fract(x), fract(y)});
  }
  return tex_color_non_recursive<FEAT>(S, t.u0[0], uv) * tex_color_non_recursive<FEAT>(S, t.u0[1], uv);
}

// ---- local shading frame: onb.rs + math.rs:89-97 ---------------------------------------------------
struct Onb {
  f3 u, v, w;
};
RENE_DEV Onb onb_from_w(f3 w) {
  Onb o;
  o.w = w;
  if (fabsf(w.x) > fabsf(w.y)) o.u = mk3(-w.z, 0.0f, w.x) / sqrtf(w.x * w.x + w.z * w.z);
  else o.u = mk3(0.0f, w.z, -w.y) / sqrtf(w.y * w.y + w.z * w.z);
  o.v = cross(w, o.u);
  return o;
}
RENE_DEV f3 to_world(const Onb& o, f3 a) { return a.x * o.u + a.y * o.v + a.z * o.w; }
RENE_DEV f3 to_local(const Onb& o, f3 a) { return mk3(dot(a, o.u), dot(a, o.v), dot(a, o.w)); }

RENE_DEV float abs_cos_theta(f3 w) { return fabsf(w.z); }
RENE_DEV float cos2_theta(f3 w) { return w.z * w.z; }
RENE_DEV float sin2_theta(f3 w) { return fmaxf(1.0f - w.z * w.z, 0.0f); }
RENE_DEV float sin_theta(f3 w) { return sqrtf(sin2_theta(w)); }
RENE_DEV float tan_theta(f3 w) { return sin_theta(w) / w.z; }
RENE_DEV float tan2_theta(f3 w) { return sin2_theta(w) / cos2_theta(w); }
RENE_DEV float cos_phi(f3 w) {
  float s = sin_theta(w);
  return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f);
}
RENE_DEV float sin_phi(f3 w) {
  float s = sin_theta(w);
  return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f);
}
RENE_DEV bool same_hemisphere(f3 a, f3 b) { return a.z * b.z > 0.0f; }

// math.rs:45-56
RENE_DEV f3 random_cosine_direction(Pcg& rng) {
  float r1 = pcg_f32(rng);
  float r2 = pcg_f32(rng);
  float z = sqrtf(1.0f - r2);
  float r2_sqrt = sqrtf(r2);
  return mk3(cos_2pi(r1) * r2_sqrt, sin_2pi(r1) * r2_sqrt, z);  // phi = 2 pi r1
}
// math.rs:8-20
RENE_DEV f3 random_in_unit_sphere(Pcg& rng) {
  for (;;) {
    float a = pcg_range(rng, -1.0f, 1.0f);
    float b = pcg_range(rng, -1.0f, 1.0f);
    float c = pcg_range(rng, -1.0f, 1.0f);
    f3 v = mk3(a, b, c);
    if (length_squared(v) < 1.0f) return v;
  }
}

// ---- Fresnel: bxdf.rs:138-165, fresnel.rs:78-102 ----------------------------------------------------
RENE_DEV float fr_dielectric(float cos_theta_i, float eta_i, float eta_t) {
  cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
  if (!(cos_theta_i > 0.0f)) {
    float t = eta_i;
    eta_i = eta_t;
    eta_t = t;
  }
  cos_theta_i = fabsf(cos_theta_i);
  float sin_theta_i = sqrtf(1.0f - cos_theta_i * cos_theta_i);
  float sin_theta_t = eta_i / eta_t * sin_theta_i;
  if (sin_theta_t >= 1.0f) return 1.0f;
  float cos_theta_t = sqrtf(1.0f - sin_theta_t * sin_theta_t);
  float r_parl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
  float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
  return 0.5f * (r_parl * r_parl + r_perp * r_perp);
}
RENE_DEV f3 fr_conductor(float cos_theta_i, f3 eta_i, f3 eta_t, f3 k) {
  cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
  f3 eta = eta_t / eta_i;
  f3 eta_k = k / eta_i;
  float c2 = cos_theta_i * cos_theta_i;
  float s2 = 1.0f - c2;
  f3 eta2 = eta * eta;
  f3 eta_k2 = eta_k * eta_k;
  f3 t0 = eta2 - eta_k2 - splat(s2);
  f3 a2plusb2 = sqrt3(t0 * t0 + 4.0f * eta2 * eta_k2);
  f3 t1 = a2plusb2 + splat(c2);
  f3 a = sqrt3(0.5f * (a2plusb2 + t0));
  f3 t2 = 2.0f * cos_theta_i * a;
  f3 rs = (t1 - t2) / (t1 + t2);
  f3 t3 = c2 * a2plusb2 + splat(s2 * s2);
  f3 t4 = t2 * s2;
  f3 rp = rs * (t3 - t4) / (t3 + t4);
  return 0.5f * (rp + rs);
}

enum : uint32_t { FR_CONDUCTOR = 0, FR_NOOP = 1, FR_DIELECTRIC = 2 };
enum : uint32_t { BX_LAMBERT = 0, BX_FRESNEL_SPECULAR, BX_FRESNEL_BLEND, BX_MICROFACET, BX_SPEC_REFL, BX_SPEC_TRANS };
enum : uint32_t { K_REFLECTION = 1, K_TRANSMISSION = 2, K_DIFFUSE = 4 };

// one BxDF lobe (EnumBxdfData, reflection.rs:94-100); only the fields its type reads are set
struct Lobe {
  uint32_t type;
  f3 a;            // albedo / rd / r / t ; FresnelSpecular: ir in a.x
  f3 b;            // rs ; SpecularTransmission: eta_a, eta_b
  float ax, ay;    // Trowbridge-Reitz alpha
  uint32_t fr;     // fresnel type
  f3 eta, k;       // conductor: eta_t, k (eta_i = 1); dielectric: eta.x = eta_i, eta.y = eta_t
};

RENE_DEV f3 fresnel_eval(const Lobe& l, float cos_i) {  // fresnel.rs:160-171
  if (l.fr == FR_NOOP) return splat(1.0f);
  if (l.fr == FR_CONDUCTOR) return fr_conductor(fabsf(cos_i), splat(1.0f), l.eta, l.k);
  return splat(fr_dielectric(cos_i, l.eta.x, l.eta.y));
}

// ---- TrowbridgeReitz, microfacet.rs:46-195 -------------------------------------------------------------
RENE_DEV float roughness_to_alpha(float roughness) {
  roughness = fmaxf(roughness, 1e-3f);
  float x = logf(roughness);
  return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
RENE_DEV float tr_d(float ax, float ay, f3 wh) {
  float t2 = tan2_theta(wh);
  if (isinf(t2)) return 0.0f;
  float c2 = cos2_theta(wh);
  float cp = cos_phi(wh), spv = sin_phi(wh);
  float e = (cp * cp / (ax * ax) + spv * spv / (ay * ay)) * t2;
  return 1.0f / (kPi * ax * ay * (c2 * c2) * (1.0f + e) * (1.0f + e));
}
RENE_DEV float tr_lambda(float ax, float ay, f3 w) {  // Q9: Beckmann's fit
  float abs_tan = fabsf(tan_theta(w));
  if (isinf(abs_tan)) return 0.0f;
  float cp = cos_phi(w), spv = sin_phi(w);
  float alpha = sqrtf(cp * cp * ax * ax + spv * spv * ay * ay);
  float a = 1.0f / (alpha * abs_tan);
  if (a >= 1.6f) return 0.0f;
  return (1.0f - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
}
RENE_DEV float tr_g(float ax, float ay, f3 wo, f3 wi) { return 1.0f / (1.0f + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi)); }
RENE_DEV float tr_pdf(float ax, float ay, f3 wo, f3 wh) {
  return tr_d(ax, ay, wh) * (1.0f / (1.0f + tr_lambda(ax, ay, wo))) * fabsf(dot(wo, wh)) / abs_cos_theta(wo);
}
RENE_DEV void tr_sample11(float cos_theta, Pcg& rng, float& sx, float& sy) {  // microfacet.rs:77-122
  float u1 = pcg_f32(rng);
  float u2 = pcg_f32(rng);
  if (cos_theta > 0.9999f) {
    float r = sqrtf(u1 / (1.0f - u1));
    sx = r * cos_2pi(u2);
    sy = r * sin_2pi(u2);
    return;
  }
  float sin_t = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
  float tan_t = sin_t / cos_theta;
  float a0 = 1.0f / tan_t;
  float g1 = 2.0f / (1.0f + (1.0f + 1.0f / sqrtf(a0 * a0)));  // Q11
  float a = 2.0f * u1 / g1 - 1.0f;
  float tmp = fminf(1.0f / (a * a - 1.0f), 1e10f);
  float b = tan_t;
  float d = sqrtf(fmaxf(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0f));
  float s1 = b * tmp - d, s2 = b * tmp + d;
  sx = (a < 0.0f || s2 > a0) ? s1 : s2;
  float s;
  if (u2 > 0.5f) {
    s = 1.0f;
    u2 = 2.0f * (u2 - 0.5f);
  } else {
    s = -1.0f;
    u2 = 2.0f * (0.5f - u2);
  }
  float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) /
            (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
  sy = s * z * sqrtf(1.0f + sx * sx);
}
RENE_DEV f3 tr_sample_wh(float ax, float ay, f3 wo, Pcg& rng) {  // microfacet.rs:124-138, 176-190
  bool flip = wo.z < 0.0f;
  f3 wi = flip ? -wo : wo;
  f3 ws = normalize(mk3(ax * wi.x, ay * wi.y, wi.z));
  float sx, sy;
  tr_sample11(ws.z, rng, sx, sy);
  float cp = cos_phi(ws), spv = sin_phi(ws);
  float slope_x = cp * sx - spv * sy;
  float slope_y = spv * sx + cp * sy;
  slope_x = ax * slope_x;
  slope_y = ay * slope_y;
  f3 wh = normalize(mk3(-slope_x, -slope_y, 1.0f));
  return flip ? -wh : wh;
}

// ---- BxDFs, bxdf.rs ---------------------------------------------------------------------------------------
RENE_DEV uint32_t lobe_kind(uint32_t type) {
  switch (type) {
    case BX_LAMBERT: case BX_FRESNEL_BLEND: case BX_MICROFACET: return K_REFLECTION | K_DIFFUSE;  // bxdf.rs:83, 262, 357
    case BX_FRESNEL_SPECULAR: return K_REFLECTION | K_TRANSMISSION;                               // bxdf.rs:185
    case BX_SPEC_REFL: return K_REFLECTION;                                                       // bxdf.rs:429
    default: return K_TRANSMISSION;                                                               // bxdf.rs:473
  }
}
RENE_DEV f3 reflect(f3 wo, f3 n) { return -wo + 2.0f * dot(wo, n) * n; }
RENE_DEV bool refract(f3 wi, f3 n, float eta, f3& out) {  // bxdf.rs:121-136
  float cos_i = dot(n, wi);
  float sin2_i = fmaxf(1.0f - cos_i * cos_i, 0.0f);
  float sin2_t = eta * eta * sin2_i;
  if (sin2_t >= 1.0f) {
    out = splat(0.0f);
    return false;
  }
  float cos_t = sqrtf(1.0f - sin2_t);
  out = eta * -wi + (eta * cos_i - cos_t) * n;
  return true;
}
RENE_DEV float pow5(float v) { return (v * v) * (v * v) * v; }

template <bool GENERAL>
RENE_DEV f3 lobe_f(const Lobe& l, f3 wo, f3 wi) {
  if (!GENERAL || l.type == BX_LAMBERT) return l.a * kInvPi;  // bxdf.rs:87-89
  if (l.type == BX_FRESNEL_BLEND) {                           // bxdf.rs:266-290
    f3 diffuse = (28.0f / (23.0f * kPi)) * l.a * (splat(1.0f) - l.b) *
                 (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wi))) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wo)));
    f3 wh = wi + wo;
    if (is_zero(wh)) return splat(0.0f);
    wh = normalize(wh);
    float c = dot(wi, wh);
    f3 schlick = l.b + pow5(1.0f - c) * (splat(1.0f) - l.b);
    f3 specular = tr_d(l.ax, l.ay, wh) / (4.0f * fabsf(c) * fmaxf(abs_cos_theta(wi), abs_cos_theta(wo))) * schlick;
    return diffuse + specular;
  }
  if (l.type == BX_MICROFACET) {  // bxdf.rs:361-381
    float co = abs_cos_theta(wo), ci = abs_cos_theta(wi);
    f3 wh = wi + wo;
    if (ci == 0.0f || co == 0.0f || is_zero(wh)) return splat(0.0f);
    wh = normalize(wh);
    f3 whf = wh.z < 0.0f ? -wh : wh;  // face_forward(wh, +z), bxdf.rs:348-354
    f3 fr = fresnel_eval(l, dot(wi, whf));
    return l.a * tr_d(l.ax, l.ay, wh) * tr_g(l.ax, l.ay, wo, wi) * fr / (4.0f * ci * co);
  }
  return splat(0.0f);
}
template <bool GENERAL>
RENE_DEV float lobe_pdf(const Lobe& l, f3 wo, f3 wi) {
  if (!GENERAL || l.type == BX_LAMBERT) return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0.0f;  // bxdf.rs:107-113
  if (l.type == BX_FRESNEL_BLEND) {  // bxdf.rs:318-328
    if (!same_hemisphere(wo, wi)) return 0.0f;
    f3 wh = normalize(wo + wi);
    return 0.5f * (abs_cos_theta(wi) * kInvPi + tr_pdf(l.ax, l.ay, wo, wh) / (4.0f * dot(wo, wh)));
  }
  if (l.type == BX_MICROFACET) {  // bxdf.rs:407-414
    if (!same_hemisphere(wo, wi)) return 0.0f;
    f3 wh = normalize(wo + wi);
    return tr_pdf(l.ax, l.ay, wo, wh) / (4.0f * dot(wo, wh));
  }
  return 0.0f;
}
struct Sampled {
  f3 wi, f;
  float pdf;
};
RENE_DEV Sampled sampled_default() { return Sampled{splat(0.0f), splat(0.0f), 0.0f}; }

template <bool GENERAL>
RENE_DEV Sampled lobe_sample(const Lobe& l, f3 wo, Pcg& rng) {
  Sampled s;
  if (!GENERAL || l.type == BX_LAMBERT) {  // bxdf.rs:91-105
    f3 wi = random_cosine_direction(rng);
    if (wo.z < 0.0f) wi.z = -wi.z;
    s.wi = wi;
    s.pdf = same_hemisphere(wo, wi) ? abs_cos_theta(wi) * kInvPi : 0.0f;
    s.f = l.a * kInvPi;
    return s;
  }
  switch (l.type) {
    case BX_FRESNEL_SPECULAR: {  // bxdf.rs:193-227
      float ir = l.a.x;
      float fr = fr_dielectric(wo.z, 1.0f, ir);
      if (pcg_f32(rng) < fr) {
        f3 wi = mk3(-wo.x, -wo.y, wo.z);
        s.wi = wi;
        s.f = splat(fr) / abs_cos_theta(wi);
        s.pdf = fr;
        return s;
      }
      float eta_i = wo.z > 0.0f ? 1.0f : ir, eta_t = wo.z > 0.0f ? ir : 1.0f;
      f3 wi;
      bool ok = refract(wo, mk3(0.0f, 0.0f, wo.z > 0.0f ? 1.0f : -1.0f), eta_i / eta_t, wi);
      s.wi = wi;
      s.f = splat(1.0f) * (1.0f - fr) / abs_cos_theta(wi);
      s.pdf = ok ? 1.0f - fr : 0.0f;
      return s;
    }
    case BX_FRESNEL_BLEND: {  // bxdf.rs:292-316
      f3 wi;
      if (pcg_f32(rng) < 0.5f) {
        wi = random_cosine_direction(rng);
        if (wo.z < 0.0f) wi.z = -wi.z;
      } else {
        f3 wh = tr_sample_wh(l.ax, l.ay, wo, rng);
        wi = reflect(wo, wh);
        if (!same_hemisphere(wo, wi)) return sampled_default();
      }
      s.wi = wi;
      s.f = lobe_f<true>(l, wo, wi);
      s.pdf = lobe_pdf<true>(l, wo, wi);
      return s;
    }
    case BX_MICROFACET: {  // bxdf.rs:383-405
      if (wo.z == 0.0f) return sampled_default();
      f3 wh = tr_sample_wh(l.ax, l.ay, wo, rng);
      if (dot(wo, wh) < 0.0f) return sampled_default();
      f3 wi = reflect(wo, wh);
      if (!same_hemisphere(wo, wi)) return sampled_default();
      s.pdf = tr_pdf(l.ax, l.ay, wo, wh) / (4.0f * dot(wo, wh));
      s.wi = wi;
      s.f = lobe_f<true>(l, wo, wi);
      return s;
    }
    case BX_SPEC_REFL: {  // bxdf.rs:437-443
      f3 wi = mk3(-wo.x, -wo.y, wo.z);
      s.wi = wi;
      s.f = fresnel_eval(l, wi.z) * l.a / abs_cos_theta(wi);
      s.pdf = 1.0f;
      return s;
    }
    default: {  // SpecularTransmission, bxdf.rs:481-512
      bool entering = wo.z > 0.0f;
      float eta_a = l.b.x, eta_b = l.b.y;
      float eta_i = entering ? eta_a : eta_b, eta_t = entering ? eta_b : eta_a;
      f3 wi;
      if (!refract(wo, mk3(0.0f, 0.0f, wo.z > 0.0f ? 1.0f : -1.0f), eta_i / eta_t, wi)) return sampled_default();
      float fx = fr_dielectric(wi.z, eta_a, eta_b);
      s.wi = wi;
      s.f = l.a * (splat(1.0f) - splat(fx)) / abs_cos_theta(wi);
      s.pdf = 1.0f;
      return s;
    }
  }
}

// ---- Bsdf, reflection.rs:228-343: up to MAXL lobes held in registers (static indices only) ----------
template <int MAXL>
struct Bsdf {
  f3 ng;
  Onb onb;
  uint32_t len;
  Lobe l[MAXL];
};

template <int MAXL>
RENE_DEV void bsdf_push(Bsdf<MAXL>& b, const Lobe& lobe) {
#pragma unroll
  for (int i = 0; i < MAXL; ++i)
    if ((uint32_t)i == b.len) b.l[i] = lobe;
  b.len++;
}
template <int MAXL, bool GENERAL>
RENE_DEV bool bsdf_contains(const Bsdf<MAXL>& b, uint32_t kind) {  // reflection.rs:267-282
  if (!GENERAL) return b.len > 0 && (kind & (K_REFLECTION | K_DIFFUSE));
  bool r = false;
#pragma unroll
  for (int i = 0; i < MAXL; ++i)
    if ((uint32_t)i < b.len && (lobe_kind(b.l[i].type) & kind)) r = true;
  return r;
}
template <int MAXL, bool GENERAL>
RENE_DEV f3 bsdf_f(const Bsdf<MAXL>& b, f3 wo_world, f3 wi_world) {  // reflection.rs:286-309
  f3 wi = to_local(b.onb, wi_world);
  f3 wo = to_local(b.onb, wo_world);
  if (wo.z == 0.0f) return splat(0.0f);
  bool refl = dot(wi_world, b.ng) * dot(wo_world, b.ng) > 0.0f;
  f3 f = splat(0.0f);
#pragma unroll
  for (int i = 0; i < MAXL; ++i) {
    if ((uint32_t)i < b.len) {
      uint32_t k = GENERAL ? lobe_kind(b.l[i].type) : (K_REFLECTION | K_DIFFUSE);
      if ((refl && (k & K_REFLECTION)) || (!refl && (k & K_TRANSMISSION))) f = f + lobe_f<GENERAL>(b.l[i], wo, wi);
    }
  }
  return f;
}
template <int MAXL, bool GENERAL>
RENE_DEV float bsdf_pdf(const Bsdf<MAXL>& b, f3 wo_world, f3 wi_world) {  // reflection.rs:328-342
  float p = 0.0f;
  f3 wo = to_local(b.onb, wo_world);
  f3 wi = to_local(b.onb, wi_world);
#pragma unroll
  for (int i = 0; i < MAXL; ++i)
    if ((uint32_t)i < b.len) p += lobe_pdf<GENERAL>(b.l[i], wo, wi);
  return p / (float)b.len;
}
template <int MAXL, bool GENERAL>
RENE_DEV Sampled bsdf_sample(const Bsdf<MAXL>& b, f3 wo_world, Pcg& rng) {  // reflection.rs:311-326
  if (b.len == 0) return sampled_default();
  uint32_t index = pcg_u32(rng) % b.len;
  f3 wo = to_local(b.onb, wo_world);
  Sampled s = sampled_default();
  if (MAXL == 1) {
    s = lobe_sample<GENERAL>(b.l[0], wo, rng);
  } else {
    Lobe sel = b.l[0];
#pragma unroll
    for (int i = 1; i < MAXL; ++i)
      if ((uint32_t)i == index) sel = b.l[i];
    s = lobe_sample<GENERAL>(sel, wo, rng);
  }
  s.pdf /= (float)b.len;
  s.wi = to_world(b.onb, s.wi);
  return s;
}

RENE_DEV Lobe lobe_zero() {
  Lobe l;
  l.type = BX_LAMBERT;
  l.a = splat(0.0f);
  l.b = splat(0.0f);
  l.ax = 0.0f;
  l.ay = 0.0f;
  l.fr = FR_CONDUCTOR;
  l.eta = splat(0.0f);
  l.k = splat(0.0f);
  return l;
}

// EnumMaterial::albedo, material.rs:720-737
template <uint32_t FEAT>
RENE_DEV f3 material_albedo(const SceneView& S, const Inst& inst, uv2 uv) {
  if (inst.kd[3] != 0.0f) return mk3(inst.kd[0], inst.kd[1], inst.kd[2]);
  const Material& m = S.materials[inst.material];
  switch (m.type) {
    case RENE_MATERIAL_MATTE: case RENE_MATERIAL_SUBSTRATE: case RENE_MATERIAL_MIRROR:
    case RENE_MATERIAL_UBER: case RENE_MATERIAL_PLASTIC:
      return tex_color<FEAT>(S, m.u0[0], uv);
    case RENE_MATERIAL_METAL: return tex_color<FEAT>(S, m.u0[1], uv);  // k, material.rs:309-316
    default: return splat(0.0f);
  }
}

// EnumMaterial::compute_bsdf, material.rs:739-769
template <uint32_t FEAT, int MAXL>
RENE_DEV void compute_bsdf(const SceneView& S, const Inst& inst, uv2 uv, Bsdf<MAXL>& b) {
  if (inst.kd[3] != 0.0f) {  // Matte over a solid texture (material.rs:127-135), resolved at upload
    Lobe l = lobe_zero();
    l.a = mk3(inst.kd[0], inst.kd[1], inst.kd[2]);
    bsdf_push(b, l);
    return;
  }
  const Material& m = S.materials[inst.material];
  uint32_t type = m.type;
  if (type == RENE_MATERIAL_MATTE) {
    Lobe l = lobe_zero();
    l.a = tex_color<FEAT>(S, m.u0[0], uv);
    bsdf_push(b, l);
    return;
  }
  if (!(FEAT & FEAT_GENERAL_BSDF)) return;
  switch (type) {
    case RENE_MATERIAL_GLASS: {  // material.rs:342-350
      Lobe l = lobe_zero();
      l.type = BX_FRESNEL_SPECULAR;
      l.a.x = m.v0[0];
      bsdf_push(b, l);
      break;
    }
    case RENE_MATERIAL_SUBSTRATE: {  // material.rs:188-216
      Lobe l = lobe_zero();
      l.type = BX_FRESNEL_BLEND;
      l.a = tex_color<FEAT>(S, m.u0[0], uv);
      l.b = tex_color<FEAT>(S, m.u0[1], uv);
      float ru = tex_color<FEAT>(S, m.u0[2], uv).x, rv = tex_color<FEAT>(S, m.u0[3], uv).x;
      if (m.u1[0] != 0) {
        ru = roughness_to_alpha(ru);
        rv = roughness_to_alpha(rv);
      }
      l.ax = ru;
      l.ay = rv;
      bsdf_push(b, l);
      break;
    }
    case RENE_MATERIAL_METAL: {  // material.rs:279-307
      Lobe l = lobe_zero();
      l.type = BX_MICROFACET;
      float ru = tex_color<FEAT>(S, m.u0[2], uv).x, rv = tex_color<FEAT>(S, m.u0[3], uv).x;
      if (m.u1[0] != 0) {
        ru = roughness_to_alpha(ru);
        rv = roughness_to_alpha(rv);
      }
      l.a = splat(1.0f);
      l.ax = ru;
      l.ay = rv;
      l.fr = FR_CONDUCTOR;
      l.eta = tex_color<FEAT>(S, m.u0[0], uv);
      l.k = tex_color<FEAT>(S, m.u0[1], uv);
      bsdf_push(b, l);
      break;
    }
    case RENE_MATERIAL_MIRROR: {  // material.rs:363-373
      Lobe l = lobe_zero();
      l.type = BX_SPEC_REFL;
      l.a = tex_color<FEAT>(S, m.u0[0], uv);
      l.fr = FR_NOOP;
      bsdf_push(b, l);
      break;
    }
    case RENE_MATERIAL_UBER: {  // material.rs:579-630
      if (MAXL < 5) break;
      float e = m.v0[0];
      f3 op = tex_color<FEAT>(S, m.u1[0], uv);
      f3 t = splat(1.0f) - op;
      if (!is_zero(t)) {
        Lobe l = lobe_zero();
        l.type = BX_SPEC_TRANS;
        l.a = t;
        l.b = mk3(1.0f, 1.0f, 0.0f);
        bsdf_push(b, l);
      }
      f3 kd = tex_color<FEAT>(S, m.u0[0], uv);
      if (!is_zero(kd)) {
        Lobe l = lobe_zero();
        l.a = kd;
        bsdf_push(b, l);
      }
      f3 ks = tex_color<FEAT>(S, m.u0[1], uv);
      if (!is_zero(ks)) {
        Lobe l = lobe_zero();
        l.type = BX_MICROFACET;
        float ru = tex_color<FEAT>(S, m.u1[2], uv).x, rv = tex_color<FEAT>(S, m.u1[3], uv).x;
        if (m.u1[1] != 0) {
          ru = roughness_to_alpha(ru);
          rv = roughness_to_alpha(rv);
        }
        l.a = ks;
        l.ax = ru;
        l.ay = rv;
        l.fr = FR_DIELECTRIC;
        l.eta = mk3(1.0f, e, 0.0f);
        bsdf_push(b, l);
      }
      f3 kr = op * tex_color<FEAT>(S, m.u0[2], uv);
      if (!is_zero(kr)) {
        Lobe l = lobe_zero();
        l.type = BX_SPEC_REFL;
        l.a = kr;
        l.fr = FR_DIELECTRIC;
        l.eta = mk3(1.0f, e, 0.0f);
        bsdf_push(b, l);
      }
      f3 kt = op * tex_color<FEAT>(S, m.u0[3], uv);
      if (!is_zero(kt)) {
        Lobe l = lobe_zero();
        l.type = BX_SPEC_TRANS;
        l.a = kt;
        l.b = mk3(1.0f, e, 0.0f);
        bsdf_push(b, l);
      }
      break;
    }
    case RENE_MATERIAL_PLASTIC: {  // material.rs:680-707
      if (MAXL < 2) break;
      f3 kd = tex_color<FEAT>(S, m.u0[0], uv);
      if (!is_zero(kd)) {
        Lobe l = lobe_zero();
        l.a = kd;
        bsdf_push(b, l);
      }
      f3 ks = tex_color<FEAT>(S, m.u0[1], uv);
      if (!is_zero(ks)) {
        Lobe l = lobe_zero();
        l.type = BX_MICROFACET;
        float rough = tex_color<FEAT>(S, m.u0[3], uv).x;
        if (m.u1[2] != 0) rough = roughness_to_alpha(rough);  // Q8: u1.z is never set, material.rs:674-676
        l.a = ks;
        l.ax = rough;
        l.ay = rough;
        l.fr = FR_DIELECTRIC;
        l.eta = mk3(1.5f, 1.0f, 0.0f);
        bsdf_push(b, l);
      }
      break;
    }
    default: break;  // None: no lobes
  }
}

// =================================================================================================
// hit shaders
// =================================================================================================
struct Surface {
  f3 position, normal;  // normal as the closest-hit shader leaves it (lib.rs:944-951 / 874-880)
  uv2 uv;
  uint32_t instance;
};

// math.rs:70-76
RENE_DEV uv2 sphere_uv(f3 p) {
  float theta = acosf(p.z);
  float phi = atan2f(p.y, p.x);
  if (phi < 0.0f) phi = phi + 2.0f * kPi;
  return uv2{phi * 0.5f * kInvPi, (theta - kPi) * -kInvPi};
}

template <bool SPHERES>
RENE_DEV Surface shade_hit(const SceneView& S, const HitRec& h, f3 ro, f3 rd) {
  Surface sf;
  const float* q = S.main.isect[h.slot].q;
  float4 c = ldg4(q + 8);
  sf.instance = __float_as_uint(c.y);
  uint32_t sidx = __float_as_uint(c.w);
  if (SPHERES && sidx != 0xffffffffu) {  // sphere_closest_hit, lib.rs:852-881
    const Sphere& sp = S.spheres[sidx];
    f3 oo = aff_point(sp.w2o, ro), od = aff_vector(sp.w2o, rd);
    f3 ohp = oo + h.t * od;
    sf.position = ro + h.t * rd;
    sf.uv = sphere_uv(ohp);
    // normal = (w2o.x . p, w2o.y . p, w2o.z . p) with w2o.{x,y,z} the COLUMNS of world_to_object
    sf.normal = mk3(sp.w2o[0] * ohp.x + sp.w2o[1] * ohp.y + sp.w2o[2] * ohp.z,
                    sp.w2o[3] * ohp.x + sp.w2o[4] * ohp.y + sp.w2o[5] * ohp.z,
                    sp.w2o[6] * ohp.x + sp.w2o[7] * ohp.y + sp.w2o[8] * ohp.z);
    return sf;
  }
  // triangle_closest_hit, lib.rs:892-952 (vertices are pre-transformed to world space)
  float4 a = ldg4(q), b = ldg4(q + 4);
  f3 p0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
  sf.position = p0 + h.u * e1 + h.v * e2;
  const float* s = S.shade[h.slot].q;
  float4 s0 = ldg4(s), s1 = ldg4(s + 4), s2 = ldg4(s + 8), s3 = ldg4(s + 12);
  float b0 = 1.0f - h.u - h.v;
  sf.normal = normalize(mk3(s0.x, s0.y, s0.z) * b0 + mk3(s1.x, s1.y, s1.z) * h.u + mk3(s2.x, s2.y, s2.z) * h.v);
  sf.uv = uv2{s0.w * b0 + s2.w * h.u + s3.y * h.v, s1.w * b0 + s3.x * h.u + s3.z * h.v};
  return sf;
}

// *_closest_hit_pdf / main_miss_pdf, lib.rs:959-1066
template <bool SPHERES>
RENE_DEV float emitter_pdf(const SceneView& S, const HitRec& h, f3 ro, f3 rd) {
  if (h.slot == 0xffffffffu) return 0.0f;
  const float* q = S.emit.isect[h.slot].q;
  float4 c = ldg4(q + 8);
  uint32_t inst = __float_as_uint(c.y);
  uint32_t sidx = __float_as_uint(c.w);
  if (SPHERES && sidx != 0xffffffffu) {  // Q4, lib.rs:1047-1066
    const float* m = S.spheres[sidx].o2w;
    float radius = (fabsf(m[0]) + fabsf(m[4]) + fabsf(m[8])) / 3.0f;
    f3 center = mk3(m[9], m[10], m[11]);
    float cos_theta_max = sqrtf(fmaxf(1.0f - radius * radius / length_squared(center - ro), 0.0f));
    float solid_angle = 2.0f * kPi * (1.0f - cos_theta_max);
    return 1.0f / solid_angle;
  }
  float4 a = ldg4(q), b = ldg4(q + 4);
  f3 p0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
  f3 hit_pos = p0 + h.u * e1 + h.v * e2;
  float4 pr = ldg4(S.emit_pdf[h.slot].q);
  float distance_squared = length_squared(ro - hit_pos);
  float cosine = fabsf(dot(normalize(rd), mk3(pr.x, pr.y, pr.z)));
  return distance_squared / (cosine * pr.w) / S.insts[inst].primitive_count;
}

// EnumSurfaceSample::sample, surface_sample.rs:69-117
template <bool SPHERES>
RENE_DEV f3 emit_sample(const SceneView& S, uint32_t obj, Pcg& rng) {
  const EmitObject& e = S.emit_objects[obj];
  if (SPHERES && e.type == 1) {
    f3 v = normalize(random_in_unit_sphere(rng));
    return aff_point(e.matrix, v);
  }
  uint32_t p = pcg_u32(rng) % e.prim_count;  // Q6
  const float* q = S.emit_tris[e.first_tri + p].q;
  float4 a = ldg4(q), b = ldg4(q + 4), c = ldg4(q + 8);
  float r = pcg_f32(rng);
  float s = pcg_f32(rng);
  if (r + s > 1.0f) {
    r = 1.0f - r;
    s = 1.0f - s;
  }
  return mk3(a.x, a.y, a.z) * (1.0f - r - s) + mk3(a.w, b.x, b.y) * r + mk3(b.z, b.w, c.x) * s;
}

// =================================================================================================
// the integrator
// =================================================================================================
RENE_DEV uint32_t lane_id() { return __lane_id(); }

// ---- accumulation-image records: 16-byte RGBA, always accessed write-through / L1-bypassing ("sc1") so
// that a pixel's running sums can be handed from one lane to another lane on another CU/XCD inside a
// launch (cdna_hip_programming.md Guideline 16: payload stored sc1 and drained, flag = agent-scope
// atomic, every load of the handed-off bytes an sc1 load; no plain load of the image exists in this kernel)
// (12-byte accesses: the alpha channel is never read or written, like the reference, lib.rs:170)
typedef float v3f __attribute__((ext_vector_type(3)));
RENE_DEV f3 fb_load(const float* p) {
  v3f v;
  asm volatile("global_load_dwordx3 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return mk3(v.x, v.y, v.z);
}
RENE_DEV void fb_store(float* p, f3 a) {
  v3f v = {a.x, a.y, a.z};
  asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
RENE_DEV void fb_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr uint32_t WORK_BATCH = 128;  // work ids a wave takes per global atomic

RENE_DEV unsigned long long wave_sum(uint32_t v) {
  unsigned long long s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  return s;
}

template <uint32_t FEAT, int MAXL, bool COUNT, bool AOV>
__global__ void __launch_bounds__(BLOCK) render_kernel(SceneView S, RenderParams P) {
  constexpr bool SPHERES = (FEAT & FEAT_SPHERES) != 0;
  constexpr bool GENERAL = (FEAT & FEAT_GENERAL_BSDF) != 0;
  constexpr bool SMALL = (FEAT & FEAT_SMALL) != 0;
  extern __shared__ uint32_t s_stack[];  // [stack depth][BLOCK]
  uint32_t* stack = s_stack + threadIdx.x;

  LaneCounters lc;
  const uint32_t W = S.width, H = S.height;
  const float tmin = 0.001f, tmax = 100000.0f;  // lib.rs:182-183

  // work-item state: one item = (pixel, frame range).  A launch of F >= 4 frames cuts every pixel into
  // a long item (frames [0, F - F/4)) and a short one (the rest); all long items are handed out before
  // any short one, so the end of the launch is balanced at a quarter of a pixel's cost.  The short
  // item continues the running sums its long item committed (same summation order as one lane doing
  // all F frames), synchronised through P.item_done[] with the sc1 hand-off described above.
  uint32_t work = 0xffffffffu, px = 0, py = 0, frame = 0;
  bool waiting = false, second = false;
  f3 acc0 = splat(0.0f), acc1 = splat(0.0f), acc2 = splat(0.0f);
  // path state
  bool active = false, done = false;
  f3 ro = splat(0.0f), rd = splat(0.0f), color = splat(0.0f);
  Pcg rng{0}, fw{0};
  int depth = 0;

  const size_t layer_stride = (size_t)W * H * 4;
  const uint32_t F = P.n_frames;
  const bool two_level = P.two_level != 0;
  const uint32_t F0 = two_level ? F - F / 4u : F;
  const uint32_t total_items = two_level ? 2u * P.n_work : P.n_work;
  // wave-uniform batch of work ids
  uint32_t batch_next = 0, batch_end = 0;
  bool exhausted = false;

  uint32_t wait_iters = 0;

  for (;;) {
    // ---- item bookkeeping --------------------------------------------------------------------------
    const uint32_t frame_end = second ? F : F0;
    bool finished = !active && !waiting && work != 0xffffffffu && frame == frame_end;
    if (finished) {
      float* p = P.framebuffer + ((size_t)(H - 1 - py) * W + px) * 4;  // add_image target, lib.rs:166
      if (AOV) {
        fb_store(p + layer_stride, acc1);
        fb_store(p + 2 * layer_stride, acc2);
      }
      fb_store(p, acc0);
      if (two_level && !second) {
        fb_drain();  // the sc1 stores have reached memory before the flag can be seen
        __hip_atomic_store(P.item_done + work, P.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      work = 0xffffffffu;
    }
    bool need = !done && work == 0xffffffffu;
    if (__any(need)) {
      if (batch_next >= batch_end && !exhausted) {  // wave-uniform: refill with one global atomic
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(P.work_counter, WORK_BATCH);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= total_items) {
          exhausted = true;
        } else {
          batch_next = base;
          batch_end = base + WORK_BATCH < total_items ? base + WORK_BATCH : total_items;
        }
      }
      unsigned long long mask = __ballot(need);
      uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane_id()) - 1ull));
      uint32_t id = batch_next + rank;
      bool got = need && !exhausted && id < batch_end;
      uint32_t taken = (uint32_t)__popcll(mask);
      batch_next = batch_next + taken < batch_end ? batch_next + taken : batch_end;
      if (need && exhausted) done = true;
      if (got) {
        bool sec = id >= P.n_work;
        uint32_t w = sec ? id - P.n_work : id;
        // item -> pixel: owned 32x32 tiles, 8x8 micro-tiles inside (a wave starts on one micro-tile)
        uint32_t k = w >> 10, r = w & 1023u;
        uint32_t tile = P.shard_rank + k * P.shard_count;
        uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
        uint32_t sub = r >> 6, l = r & 63u;
        uint32_t x = tx * RENE_TILE_SIZE + (sub & 3u) * 8u + (l & 7u);
        uint32_t yi = ty * RENE_TILE_SIZE + (sub >> 2) * 8u + (l >> 3);  // image row, top first
        if (x < W && yi < H) {
          work = id;
          second = sec;
          px = x;
          py = H - 1 - yi;  // launch_id.y
          frame = sec ? F0 : 0u;
          waiting = true;  // sums are loaded below (a short item first waits for its long item)
          wait_iters = 0;
        }
      }
    }
    if (waiting) {
      bool ready = !second || __hip_atomic_load(P.item_done + (work - P.n_work), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.epoch;
      if (!ready && ++wait_iters > (1u << 22)) {
        // safety net so that every wave can leave: a long item always makes progress, so this bound is
        // never reached; if it were, the host reports RENE_ERR_DEVICE
        atomicAdd(&P.counters[8], 1ull);
        ready = true;
      }
      if (ready) {
        const float* p = P.framebuffer + ((size_t)(H - 1 - py) * W + px) * 4;
        acc0 = fb_load(p);
        if (AOV) {
          acc1 = fb_load(p + layer_stride);
          acc2 = fb_load(p + 2 * layer_stride);
        }
        waiting = false;
      }
    }
    if (__all(done)) break;

    // ---- ray generation, lib.rs:174-189 -----------------------------------------------------------
    if (!active && !waiting && work != 0xffffffffu && frame < (second ? F : F0)) {
      uint32_t seed = P.seeds[frame];
      frame++;
      lc.paths++;
      rng = pcg_new((py * W + px) ^ seed);
      fw = pcg_new(seed);
      float u = ((float)px + pcg_f32(rng)) / (float)(W - 1);  // Q2
      float v = ((float)py + pcg_f32(rng)) / (float)(H - 1);
      f3 origin = m4_point(S.c2w, splat(0.0f));
      f3 target = m4_point(S.proj_inv, mk3(u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f));  // camera.rs:77-90
      target = m4_point(S.c2w, target);
      ro = origin;
      rd = normalize(target - origin);
      color = splat(1.0f);
      depth = 0;
      active = true;
    }

    // ---- one bounce, lib.rs:192-355 ------------------------------------------------------------------
    if (active) {
      lc.closest++;
      HitRec h = trace_accel<SMALL, false, SPHERES, COUNT>(S.main, S.spheres, ro, rd, tmin, tmax, stack, lc);
      if (h.slot == 0xffffffffu) {  // main_miss, lib.rs:120-139, 209-211
        f3 bg = splat(0.0f);
        if (FEAT & FEAT_BACKGROUND) {
          uv2 uv = sphere_uv(normalize(m4_vector(S.bg_matrix, rd)));
          bg = mk3(S.bg_color[0], S.bg_color[1], S.bg_color[2]) * tex_color<FEAT | FEAT_TEXTURES>(S, S.bg_texture, uv);
        }
        acc0 = acc0 + color * bg;
        lc.adds++;
        active = false;
      } else {
        lc.hits++;
        Surface sf = shade_hit<SPHERES>(S, h, ro, rd);
        const Inst& inst = S.insts[sf.instance];
        f3 wo = -normalize(rd);
        f3 normal = normalize(sf.normal);
        f3 position = sf.position;
        Bsdf<MAXL> bsdf;
        bsdf.len = 0;
        bsdf.ng = normal;
        bsdf.onb = onb_from_w(normal);
        compute_bsdf<FEAT, MAXL>(S, inst, sf.uv, bsdf);

        if (inst.emit[3] != 0.0f) {  // lib.rs:225-227, area_light.rs:66-74
          f3 e = dot(wo, normal) > 0.0f ? mk3(inst.emit[0], inst.emit[1], inst.emit[2]) : splat(0.0f);
          acc0 = acc0 + color * e;
          lc.adds++;
        }
        if (depth == 0) {  // lib.rs:229-232
          if (AOV) {
            acc1 = acc1 + normal;
            acc2 = acc2 + material_albedo<FEAT>(S, inst, sf.uv);
          }
          lc.adds += 2;
        }
        if (FEAT & FEAT_LIGHTS) {  // lib.rs:234-272
          for (uint32_t li = 0; li < S.lights_len; ++li) {
            float4 ld = ldg4(S.lights[li].dir), lL = ldg4(S.lights[li].L);
            f3 target = position + mk3(ld.x, ld.y, ld.z);  // light.rs:52-55
            f3 wi = normalize(target - position);
            lc.shadow++;
            HitRec sh = trace_accel<SMALL, true, SPHERES, COUNT>(S.main, S.spheres, position, wi, tmin, 1e5f, stack, lc);
            if (sh.slot == 0xffffffffu) {
              f3 f = bsdf_f<MAXL, GENERAL>(bsdf, wo, wi);
              acc0 = acc0 + color * f * fabsf(dot(wi, normal)) * mk3(lL.x, lL.y, lL.z);
              lc.adds++;
            }
          }
        }
        bool alive = true;
        if (S.emit_object_len > 0 && bsdf_contains<MAXL, GENERAL>(bsdf, K_DIFFUSE)) {  // lib.rs:274-324
          f3 wi, f;
          float pdf;
          if (pcg_f32(fw) > 0.5f) {  // Q3: frame-wide stream
            uint32_t obj = pcg_u32(fw) % S.emit_object_len;
            wi = normalize(emit_sample<SPHERES>(S, obj, fw) - position);
            pdf = bsdf_pdf<MAXL, GENERAL>(bsdf, wi, normal);  // Q1: (wi, normal), lib.rs:287
            f = bsdf_f<MAXL, GENERAL>(bsdf, wo, wi);
          } else {
            Sampled s = bsdf_sample<MAXL, GENERAL>(bsdf, wo, rng);
            wi = s.wi;
            pdf = s.pdf;
            f = s.f;
          }
          ro = position;
          rd = wi;
          lc.emitter++;
          HitRec eh = trace_accel<SMALL, false, SPHERES, COUNT>(S.emit, S.spheres, ro, rd, tmin, tmax, stack, lc);  // Q5
          float pdf_l = emitter_pdf<SPHERES>(S, eh, ro, rd);
          color = color * (f * fabsf(dot(normal, wi)));
          pdf = 0.5f * pdf + 0.5f * pdf_l / (float)S.emit_object_len;
          if (pdf < 1e-5f) alive = false;
          else color = color / pdf;
        } else {  // lib.rs:325-337
          Sampled s = bsdf_sample<MAXL, GENERAL>(bsdf, wo, rng);
          if (s.pdf < 1e-5f) {
            alive = false;
          } else {
            color = color * (s.f * fabsf(dot(normal, s.wi)) / s.pdf);
            ro = position;
            rd = s.wi;
          }
        }
        if (alive && is_zero(color)) alive = false;  // lib.rs:340-342
        if (alive && depth > 12) {                   // lib.rs:345-354
          float rr_coin = pcg_f32(fw);
          float continue_p = max_element(color);
          if (rr_coin > continue_p) alive = false;
          else color = color / continue_p;
        }
        depth++;
        if (depth >= 50) alive = false;  // lib.rs:192 (Q7)
        active = alive;
      }
    }
  }

  // ---- counters: one atomic per wave per counter ---------------------------------------------------
  unsigned long long sums[8] = {wave_sum(lc.closest), wave_sum(lc.shadow), wave_sum(lc.emitter),
                                wave_sum(lc.paths),   wave_sum(lc.hits),   wave_sum(lc.adds),
                                wave_sum(lc.nodes),   wave_sum(lc.prims)};
  if (lane_id() == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (sums[i]) atomicAdd(&P.counters[i], sums[i]);
  }
}

#include "render_wf.inc"

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
// per-function BSDF probe (rene_bsdf_eval): builds the material's lobes at (normal, uv) exactly like
// the integrator does and returns f(wo,wi), pdf(wo,wi) and one sample_f(wo) drawn from
// PCG32si::new(seed).  out[12] = f.xyz, pdf, s.wi.xyz, s.f.xyz, s.pdf, len
// -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) bsdf_eval_kernel(SceneView S, uint32_t material, uint32_t n, const float* nrm3,
                                                       const float* uv2_, const float* wo3, const float* wi3,
                                                       const uint32_t* seeds, float* out) {
  constexpr uint32_t ALL = FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_LIGHTS | FEAT_BACKGROUND | FEAT_MULTI_LOBE;
  uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  Inst inst;
  inst.material = material;
  inst.area_light = 0;
  inst.primitive_count = 1.0f;
  inst.material_type = S.materials[material].type;
  inst.kd[0] = inst.kd[1] = inst.kd[2] = inst.kd[3] = 0.0f;
  inst.emit[0] = inst.emit[1] = inst.emit[2] = inst.emit[3] = 0.0f;
  f3 normal = normalize(mk3(nrm3[3 * i], nrm3[3 * i + 1], nrm3[3 * i + 2]));
  Bsdf<5> b;
  b.len = 0;
  b.ng = normal;
  b.onb = onb_from_w(normal);
  compute_bsdf<ALL, 5>(S, inst, uv2{uv2_[2 * i], uv2_[2 * i + 1]}, b);
  f3 wo = mk3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]), wi = mk3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]);
  f3 f = bsdf_f<5, true>(b, wo, wi);
  float p = b.len ? bsdf_pdf<5, true>(b, wo, wi) : 0.0f;
  Pcg rng = pcg_new(seeds[i]);
  Sampled sf = bsdf_sample<5, true>(b, wo, rng);
  float* o = out + 12 * (size_t)i;
  o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = p;
  o[4] = sf.wi.x; o[5] = sf.wi.y; o[6] = sf.wi.z;
  o[7] = sf.f.x; o[8] = sf.f.y; o[9] = sf.f.z; o[10] = sf.pdf; o[11] = (float)b.len;
}

hipError_t launch_bsdf_eval(const SceneView& S, uint32_t material, uint32_t n, const float* nrm3, const float* uv,
                            const float* wo3, const float* wi3, const uint32_t* seeds, float* out, hipStream_t st) {
  hipLaunchKernelGGL(bsdf_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, st, S, material, n, nrm3, uv, wo3, wi3, seeds, out);
  return hipGetLastError();
}

// =================================================================================================
// host-side dispatch
// =================================================================================================
template <uint32_t FEAT, int MAXL>
static hipError_t launch_feat(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st) {
  size_t lds = (FEAT & FEAT_SMALL) ? 0 : (size_t)cfg.stack_depth * BLOCK * sizeof(uint32_t);
  dim3 grid(cfg.grid), block(BLOCK);
  bool count = (P.flags & RENE_FLAG_COUNTERS) != 0, aov = !(P.flags & RENE_FLAG_NO_AOV);
  if (!(FEAT & FEAT_SMALL) && !(P.flags & RENE_FLAG_NO_RESTART)) {  // BVH scenes: traversal-restart state machine
    if (count) {
      if (aov) hipLaunchKernelGGL((render_kernel_wf<FEAT, MAXL, true, true>), grid, block, lds, st, S, P);
      else hipLaunchKernelGGL((render_kernel_wf<FEAT, MAXL, true, false>), grid, block, lds, st, S, P);
    } else {
      if (aov) hipLaunchKernelGGL((render_kernel_wf<FEAT, MAXL, false, true>), grid, block, lds, st, S, P);
      else hipLaunchKernelGGL((render_kernel_wf<FEAT, MAXL, false, false>), grid, block, lds, st, S, P);
    }
    return hipGetLastError();
  }
  if (count) {
    if (aov) hipLaunchKernelGGL((render_kernel<FEAT, MAXL, true, true>), grid, block, lds, st, S, P);
    else hipLaunchKernelGGL((render_kernel<FEAT, MAXL, true, false>), grid, block, lds, st, S, P);
  } else {
    if (aov) hipLaunchKernelGGL((render_kernel<FEAT, MAXL, false, true>), grid, block, lds, st, S, P);
    else hipLaunchKernelGGL((render_kernel<FEAT, MAXL, false, false>), grid, block, lds, st, S, P);
  }
  return hipGetLastError();
}

hipError_t launch_render(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st) {
  // Specialisations: Matte-only fast path (Cornell, dragon-class), general single-lobe, general
  // multi-lobe; each with the BVH traversal or, for tiny scenes, the wave-coherent item loop.
  constexpr uint32_t ALL = FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_LIGHTS | FEAT_BACKGROUND | FEAT_MULTI_LOBE;
  constexpr uint32_t GEN1 = ALL & ~FEAT_MULTI_LOBE;
  const uint32_t f = cfg.features;
  const bool small = (f & FEAT_SMALL) != 0;
  if (!(f & (FEAT_SPHERES | FEAT_GENERAL_BSDF | FEAT_TEXTURES | FEAT_BACKGROUND | FEAT_MULTI_LOBE)))
    return small ? launch_feat<FEAT_LIGHTS | FEAT_SMALL, 1>(cfg, S, P, st) : launch_feat<FEAT_LIGHTS, 1>(cfg, S, P, st);
  if (!(f & FEAT_MULTI_LOBE))
    return small ? launch_feat<GEN1 | FEAT_SMALL, 1>(cfg, S, P, st) : launch_feat<GEN1, 1>(cfg, S, P, st);
  return small ? launch_feat<ALL | FEAT_SMALL, 5>(cfg, S, P, st) : launch_feat<ALL, 5>(cfg, S, P, st);
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
