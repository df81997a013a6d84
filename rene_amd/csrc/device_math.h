// device_math.h -- small fp32 vector helpers + the PCG32si stream for gfx950 device code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rene {

#define RENE_DEV __device__ __forceinline__

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.318309886183790671538f;
constexpr float kTau = 6.28318530717958647692f;

struct f3 {
  float x, y, z;
};
RENE_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RENE_DEV f3 splat(float s) { return f3{s, s, s}; }
RENE_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RENE_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RENE_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
RENE_DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RENE_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RENE_DEV f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
RENE_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp
RENE_DEV float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32, 1 ulp
RENE_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); } // v_sqrt_f32, 1 ulp
// a / b on the hot path: one v_rcp_f32 and a multiply (<= 2 ulp) instead of the ~8-instruction range-scaled quotient
// the compiler emits for `/` (operands here are pdfs, pixel counts, cosines: far from the denormal / overflow range)
RENE_DEV float qdiv(float a, float b) { return a * fast_rcp(b); }
RENE_DEV f3 operator/(f3 a, float s) {
  float r = fast_rcp(s);
  return {a.x * r, a.y * r, a.z * r};
}
RENE_DEV f3 operator/(f3 a, f3 b) { return {a.x * fast_rcp(b.x), a.y * fast_rcp(b.y), a.z * fast_rcp(b.z)}; }
RENE_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RENE_DEV f3 cross(f3 a, f3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RENE_DEV float length_squared(f3 a) { return dot(a, a); }
RENE_DEV float length(f3 a) { return fast_sqrt(dot(a, a)); }
RENE_DEV f3 normalize(f3 a) {
  float r = fast_rsq(dot(a, a));
  return {a.x * r, a.y * r, a.z * r};
}
RENE_DEV float max_element(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
RENE_DEV bool is_zero(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
RENE_DEV f3 sqrt3(f3 a) { return {fast_sqrt(a.x), fast_sqrt(a.y), fast_sqrt(a.z)}; }
RENE_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }  // GLSL FClamp

// column-major 4x4 (glam Mat4) times point / vector; no perspective divide (camera.rs:79-83)
RENE_DEV f3 m4_point(const float* m, f3 p) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
RENE_DEV f3 m4_vector(const float* m, f3 p) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
          m[2] * p.x + m[6] * p.y + m[10] * p.z};
}
typedef const __attribute__((address_space(4))) float* cfloat_ptr;  // constant address space -> scalar loads
RENE_DEV f3 m4_point(cfloat_ptr m, f3 p) {
  return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
          m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
// c2w . (0, 0, 0), camera.rs:79: read off the matrix.  rene_create refuses a matrix that is not finite, and for a finite one the nine
// products by zero that m4_point(cam, 0) spells out add nothing -- but the compiler must keep them (0 * x is not 0 for every x)
RENE_DEV f3 camera_origin(cfloat_ptr cam) {
#ifdef RENE_CAM_ORIGIN_M4  // A/B switch: the spelled-out form
  return m4_point(cam, f3{0.0f, 0.0f, 0.0f});
#else
  return f3{cam[12], cam[13], cam[14]};
#endif
}
// 3x4 affine stored as x,y,z,w column vectors
RENE_DEV f3 aff_point(const float* m, f3 p) {
  return {p.x * m[0] + p.y * m[3] + p.z * m[6] + m[9], p.x * m[1] + p.y * m[4] + p.z * m[7] + m[10],
          p.x * m[2] + p.y * m[5] + p.z * m[8] + m[11]};
}
RENE_DEV f3 aff_vector(const float* m, f3 p) {
  return {p.x * m[0] + p.y * m[3] + p.z * m[6], p.x * m[1] + p.y * m[4] + p.z * m[7],
          p.x * m[2] + p.y * m[5] + p.z * m[8]};
}

// ---- PCG32si, rene-shader/src/rand.rs:4-52 (integer-exact) -------------------------------------------
struct Pcg {
  uint32_t state;
};
RENE_DEV void pcg_step(Pcg& r) { r.state = r.state * 747796405u + 2891336453u; }
RENE_DEV Pcg pcg_new(uint32_t seed) {  // rand.rs:24-30
  Pcg r{seed};
  pcg_step(r);
  r.state += seed;
  pcg_step(r);
  return r;
}
RENE_DEV uint32_t pcg_u32(Pcg& r) {  // rand.rs:19-22, 32-36
  uint32_t s = r.state;
  pcg_step(r);
  uint32_t word = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (word >> 22) ^ word;
}
// The build-defined seed schedule (SURVEY section 8 d): the seed of global frame g is the g-th next_u32() of
// PCG32si::new(master).  Computed where it is needed by jumping the generator ahead (the LCG's g-fold composition by
// repeated squaring, at most 32 rounds) -- no seed table travels to the device: a launch is a kernel launch and two event
// records, nothing that needs a copy engine or a free CU slot while persistent kernels hold the chip.
// g steps of the generator's LCG as one affine map s -> mul * s + add
RENE_DEV void lcg_pow(uint32_t g, uint32_t& acc_mul, uint32_t& acc_add) {
  uint32_t mul = 747796405u, add = 2891336453u;
  acc_mul = 1u;
  acc_add = 0u;
  for (; g; g >>= 1) {
    if (g & 1u) {
      acc_mul *= mul;
      acc_add = acc_add * mul + add;
    }
    add = (mul + 1u) * add;
    mul *= mul;
  }
}
RENE_DEV uint32_t pcg_output(uint32_t s) {  // rand.rs:19-22: the output permutation of a state
  const uint32_t word = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (word >> 22) ^ word;
}
RENE_DEV uint32_t frame_state(uint32_t state0, uint32_t g) {
  uint32_t m, a;
  lcg_pow(g, m, a);
  return m * state0 + a;
}
RENE_DEV uint32_t frame_seed(uint32_t state0, uint32_t g) { return pcg_output(frame_state(state0, g)); }
RENE_DEV float pcg_f32(Pcg& r) {  // rand.rs:38-47
  return (1.0f / 16777216.0f) * (float)(pcg_u32(r) >> 8);
}
RENE_DEV float pcg_range(Pcg& r, float lo, float hi) { return lo + (hi - lo) * pcg_f32(r); }

// x % n: a 32-bit remainder is ~20 VALU instructions on CDNA; n is 1 or a small power of two in
// most scenes (lobe count, emitter count, triangles of a quad emitter), which is a single AND
// x / n for x < 2^31 and a quotient below 2^22, through the reciprocal the host supplies (inv = 1.0f / n): the float
// estimate is off by at most one, the remainder says which way.  Replaces the ~24-instruction expansion of an
// integer division in the work-item bookkeeping.
RENE_DEV uint32_t udiv_small(uint32_t x, uint32_t n, float inv, uint32_t& rem) {
  uint32_t q = (uint32_t)((float)x * inv);
  int32_t r = (int32_t)(x - q * n);
  if (r < 0) { q--; r += (int32_t)n; }
  if (r >= (int32_t)n) { q++; r -= (int32_t)n; }
  rem = (uint32_t)r;
  return q;
}
RENE_DEV uint32_t umod(uint32_t x, uint32_t n) { return (n & (n - 1u)) == 0u ? (x & (n - 1u)) : (x % n); }

// sin/cos of 2*pi*x for x in [0,1): v_sin_f32 / v_cos_f32 take their argument in revolutions
RENE_DEV float sin_2pi(float x) { return __builtin_amdgcn_sinf(x); }
RENE_DEV float cos_2pi(float x) { return __builtin_amdgcn_cosf(x); }

}  // namespace rene
