#!/usr/bin/env python3
"""Writes rene_amd/csrc/selftest/valu_rate_probe.hip: what a wave64 VALU instruction costs on one SIMD of an MI355X, per mnemonic -- eight
independent chains of ONE instruction, 64 per loop trip, w waves per SIMD, cycles per instruction on the engine clock the kernel measures itself.
The table it prints (profiles/r03_valu_rates.txt) is what tools/valu_price.py prices a kernel's instruction stream with.
    python3 tools/make_valu_probe.py && hipcc --offload-arch=gfx950 -O2 -o rene_amd/csrc/selftest/valu_rate_probe rene_amd/csrc/selftest/valu_rate_probe.hip"""
import os

# (mnemonic as the disassembler prints it, asm template; %A = a 32-bit chain register (in/out), %B %C = 32-bit inputs, %L = a 64-bit chain register,
#  %M = a 64-bit input, %S = an SGPR pair input, %s = an SGPR input, %P = a 2 x f32 chain register, %Q = a 2 x f32 input)
OPS = [
    ("v_mul_f32_e32", "v_mul_f32_e32 %A, %A, %B"), ("v_add_f32_e32", "v_add_f32_e32 %A, %A, %B"), ("v_sub_f32_e32", "v_sub_f32_e32 %A, %A, %B"),
    ("v_fmac_f32_e32", "v_fmac_f32_e32 %A, %B, %C"), ("v_fmac_f32_e64", "v_fmac_f32_e64 %A, %s, %C"), ("v_fma_f32", "v_fma_f32 %A, %A, %B, %C"),
    ("v_fma_f32(sgpr)", "v_fma_f32 %A, %A, %s, 0.5"), ("v_fmaak_f32", "v_fmaak_f32 %A, %A, %B, 0x3f000001"), ("v_fmamk_f32", "v_fmamk_f32 %A, %A, 0x3f800001, %B"),
    ("v_mul_f32_e64", "v_mul_f32_e64 %A, %A, -%B"), ("v_pk_fma_f32", "v_pk_fma_f32 %P, %P, %Q, %Q"),
    ("v_max_f32_e32", "v_max_f32_e32 %A, %A, %B"), ("v_min_f32_e32", "v_min_f32_e32 %A, %A, %B"), ("v_max3_f32", "v_max3_f32 %A, %A, %B, %C"),
    ("v_min3_f32", "v_min3_f32 %A, %A, %B, %C"), ("v_med3_f32", "v_med3_f32 %A, %A, %B, %C"),
    ("v_mov_b32_e32", "v_mov_b32_e32 %A, %B"), ("v_add_u32_e32", "v_add_u32_e32 %A, %A, %B"), ("v_sub_u32_e32", "v_sub_u32_e32 %A, %A, %B"),
    ("v_and_b32_e32", "v_and_b32_e32 %A, %A, %B"), ("v_xor_b32_e32", "v_xor_b32_e32 %A, %A, %B"), ("v_lshlrev_b32_e32", "v_lshlrev_b32_e32 %A, 1, %A"),
    ("v_lshrrev_b32_e32", "v_lshrrev_b32_e32 %A, 1, %A"), ("v_ashrrev_i32_e32", "v_ashrrev_i32_e32 %A, 1, %A"), ("v_max_i32_e32", "v_max_i32_e32 %A, %A, %B"),
    ("v_min_u32_e32", "v_min_u32_e32 %A, %A, %B"), ("v_max_u32_e32", "v_max_u32_e32 %A, %A, %B"), ("v_mul_lo_u32", "v_mul_lo_u32 %A, %A, %B"),
    ("v_mul_hi_u32", "v_mul_hi_u32 %A, %A, %B"), ("v_lshl_add_u32", "v_lshl_add_u32 %A, %A, 1, %B"), ("v_lshl_or_b32", "v_lshl_or_b32 %A, %A, 1, %B"),
    ("v_xad_u32", "v_xad_u32 %A, %A, %B, %C"), ("v_bfi_b32", "v_bfi_b32 %A, %A, %B, %C"), ("v_bfe_u32", "v_bfe_u32 %A, %A, 1, 30"),
    ("v_or3_b32", "v_or3_b32 %A, %A, %B, %C"), ("v_lshl_add_u64", "v_lshl_add_u64 %L, %L, 1, %M"), ("v_lshlrev_b64", "v_lshlrev_b64 %L, 1, %L"),
    ("v_mad_u64_u32", "v_mad_u64_u32 %L, s[100:101], %B, %C, %L"), ("v_mad_i64_i32", "v_mad_i64_i32 %L, s[100:101], %B, %C, %L"),
    ("v_cndmask_b32_e32", "v_cndmask_b32_e32 %A, %A, %B, vcc"), ("v_cndmask_b32_e64", "v_cndmask_b32_e64 %A, %A, %B, %S"),
    ("v_cndmask_b32_e64(vcc)", "v_cndmask_b32_e64 %A, %A, %B, vcc"),
    ("v_cmp_e32+v_cndmask_e32 (a pair)", "v_cmp_lt_f32_e32 vcc, %A, %B\\n\\tv_cndmask_b32_e32 %A, %A, %B, vcc"),
    ("v_cmp_e64+v_cndmask_e64 (a pair)", "v_cmp_lt_f32_e64 s[100:101], %A, %B\\n\\tv_cndmask_b32_e64 %A, %A, %B, s[100:101]"),
    ("cmp_e32 + 3 cndmask_e32 (the four)", "v_cmp_lt_f32_e32 vcc, %A, %B\\n\\tv_cndmask_b32_e32 %A, %A, %B, vcc\\n\\tv_cndmask_b32_e32 %A, %A, %C, vcc\\n\\tv_cndmask_b32_e32 %A, %A, %B, vcc"),
    ("cmp_e64 + 3 cndmask_e64 (the four)", "v_cmp_lt_f32_e64 s[100:101], %A, %B\\n\\tv_cndmask_b32_e64 %A, %A, %B, s[100:101]\\n\\tv_cndmask_b32_e64 %A, %A, %C, s[100:101]\\n\\tv_cndmask_b32_e64 %A, %A, %B, s[100:101]"),
    ("cmp_e32, add, cndmask_e32 (the three)", "v_cmp_lt_f32_e32 vcc, %A, %B\\n\\tv_add_f32_e32 %A, %A, %B\\n\\tv_cndmask_b32_e32 %A, %A, %C, vcc"),
    ("cmp_e32, 3 add, cndmask_e32 (the five)", "v_cmp_lt_f32_e32 vcc, %A, %B\\n\\tv_add_f32_e32 %A, %A, %B\\n\\tv_add_f32_e32 %A, %A, %B\\n\\tv_add_f32_e32 %A, %A, %B\\n\\tv_cndmask_b32_e32 %A, %A, %C, vcc"),
    ("cmp_e64, 3 add, cndmask_e64 (the five)", "v_cmp_lt_f32_e64 s[100:101], %A, %B\\n\\tv_add_f32_e32 %A, %A, %B\\n\\tv_add_f32_e32 %A, %A, %B\\n\\tv_add_f32_e32 %A, %A, %B\\n\\tv_cndmask_b32_e64 %A, %A, %C, s[100:101]"),
    ("cmp_e32, cnd_e32, add, cnd_e32 (the four)", "v_cmp_lt_f32_e32 vcc, %A, %B\\n\\tv_cndmask_b32_e32 %A, %A, %C, vcc\\n\\tv_add_f32_e32 %A, %A, %B\\n\\tv_cndmask_b32_e32 %A, %A, %C, vcc"),
    ("v_add_co_u32_e32", "v_add_co_u32_e32 %A, vcc, %A, %B"),
    ("v_addc_co_u32_e32", "v_addc_co_u32_e32 %A, vcc, %A, %B, vcc"),
    ("v_cmp_lt_f32_e32", "v_cmp_lt_f32_e32 vcc, %A, %B"), ("v_cmp_lt_f32_e64", "v_cmp_lt_f32_e64 s[100:101], %A, %B"),
    ("v_cmp_eq_u32_e32", "v_cmp_eq_u32_e32 vcc, %A, %B"), ("v_cmp_gt_u32_e64", "v_cmp_gt_u32_e64 s[100:101], %A, %B"),
    ("v_cvt_f32_u32_e32", "v_cvt_f32_u32_e32 %A, %A"), ("v_cvt_u32_f32_e32", "v_cvt_u32_f32_e32 %A, %A"), ("v_cvt_i32_f32_e32", "v_cvt_i32_f32_e32 %A, %A"),
    ("v_cvt_f32_ubyte0_e32", "v_cvt_f32_ubyte0_e32 %A, %A"), ("v_cvt_f32_ubyte2_e32", "v_cvt_f32_ubyte2_e32 %A, %A"),
    ("v_floor_f32_e32", "v_floor_f32_e32 %A, %A"), ("v_ldexp_f32", "v_ldexp_f32 %A, %A, %B"), ("v_frexp_mant_f32_e32", "v_frexp_mant_f32_e32 %A, %A"),
    ("v_frexp_exp_i32_f32_e32", "v_frexp_exp_i32_f32_e32 %A, %A"),
    ("v_rcp_f32_e32", "v_rcp_f32_e32 %A, %A"), ("v_rcp_iflag_f32_e32", "v_rcp_iflag_f32_e32 %A, %A"), ("v_rsq_f32_e32", "v_rsq_f32_e32 %A, %A"),
    ("v_sqrt_f32_e32", "v_sqrt_f32_e32 %A, %A"), ("v_sin_f32_e32", "v_sin_f32_e32 %A, %A"), ("v_cos_f32_e32", "v_cos_f32_e32 %A, %A"),
    ("v_exp_f32_e32", "v_exp_f32_e32 %A, %A"), ("v_log_f32_e32", "v_log_f32_e32 %A, %A"),
]


# second set (round 4, `make_valu_probe.py decode`): candidates for a cheaper BVH4 node decode -- what turns a quantised child plane into a float, and
# the packed / half-precision forms of the slab test's arithmetic
OPS_DECODE = [
    ("v_cvt_f32_ubyte1_e32", "v_cvt_f32_ubyte1_e32 %A, %A"), ("v_cvt_f32_ubyte3_e32", "v_cvt_f32_ubyte3_e32 %A, %A"),
    ("v_cvt_f32_f16_e32", "v_cvt_f32_f16_e32 %A, %A"),
    ("v_fma_mix_f32 (f16 lo, f32, f32)", "v_fma_mix_f32 %A, %B, %C, %A op_sel_hi:[1,0,0]"),
    ("v_fma_mix_f32 (f16 hi, f32, f32)", "v_fma_mix_f32 %A, %B, %C, %A op_sel:[1,0,0] op_sel_hi:[1,0,0]"),
    ("v_fma_mix_f32 (all f32)", "v_fma_mix_f32 %A, %B, %C, %A"),
    ("v_pk_fma_f16", "v_pk_fma_f16 %A, %A, %B, %C"), ("v_pk_mul_f32", "v_pk_mul_f32 %P, %P, %Q"), ("v_pk_add_f32", "v_pk_add_f32 %P, %P, %Q"),
    ("v_pk_max_f16", "v_pk_max_f16 %A, %A, %B"), ("v_pk_min_f16", "v_pk_min_f16 %A, %A, %B"),
    ("v_perm_b32", "v_perm_b32 %A, %A, %B, %C"), ("v_dot2_f32_f16", "v_dot2_f32_f16 %A, %B, %C, %A"),
    ("v_or_b32_sdwa (byte1)", "v_or_b32_sdwa %A, %A, %B dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"),
    ("v_and_b32_sdwa (byte2)", "v_and_b32_sdwa %A, %A, %B dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"),
    ("v_add_u32_sdwa (byte3)", "v_add_u32_sdwa %A, %A, %B dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"),
    ("v_cvt_f32_u32_sdwa (byte1)", "v_cvt_f32_u32_sdwa %A, %A dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1"),
    ("v_mul_f32_sdwa (word1)", "v_mul_f32_sdwa %A, %A, %B dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"),
    ("v_cvt_pk_f32_fp8_e32", "v_cvt_pk_f32_fp8_e32 %P, %B"), ("v_cvt_scalef32_pk_f32_fp8", "v_cvt_scalef32_pk_f32_fp8 %P, %B, %C"),
    ("v_max_f16_e32", "v_max_f16_e32 %A, %A, %B"), ("v_alignbit_b32", "v_alignbit_b32 %A, %A, %B, 8"), ("v_mad_u32_u24", "v_mad_u32_u24 %A, %A, %B, %C"),
    ("v_mul_u32_u24_e32", "v_mul_u32_u24_e32 %A, %A, %B"), ("v_or_b32_e32", "v_or_b32_e32 %A, %A, %B"), ("v_and_or_b32", "v_and_or_b32 %A, %A, %B, %C"),
]


def stmt(tpl, k):
    outs, ins = [], []
    t = tpl
    def use(tok, constraint, expr, is_out):
        nonlocal t
        if tok in t:
            idx = len(outs) + len(ins)
    # build operand lists in a fixed order: outputs first
    ops_out, ops_in = [], []
    if "%A" in t: ops_out.append(('"+v"', f"a[{k}]", "%A"))
    if "%L" in t: ops_out.append(('"+v"', f"l[{k}]", "%L"))
    if "%P" in t: ops_out.append(('"+v"', f"p[{k}]", "%P"))
    if "%B" in t: ops_in.append(('"v"', "b", "%B"))
    if "%C" in t: ops_in.append(('"v"', "c", "%C"))
    if "%M" in t: ops_in.append(('"v"', "m", "%M"))
    if "%Q" in t: ops_in.append(('"v"', "q", "%Q"))
    if "%S" in t: ops_in.append(('"s"', "sm", "%S"))
    if "%s" in t: ops_in.append(('"s"', "sc", "%s"))
    for i, (_, _, tok) in enumerate(ops_out + ops_in):
        t = t.replace(tok, f"%{i}")
    clob = ' : "vcc"' if ("vcc," in tpl and (tpl.startswith("v_cmp") or "_co_" in tpl)) else (' : "s100", "s101"' if "s[100:101]" in tpl else "")
    o = ", ".join(f"{c}({e})" for c, e, _ in ops_out)
    i = ", ".join(f"{c}({e})" for c, e, _ in ops_in)
    return f'asm volatile("{t}" : {o} : {i}{clob});'


def main():
    import sys
    global OPS
    which = sys.argv[1] if len(sys.argv) > 1 else "rates"
    if which == "decode":
        OPS = OPS_DECODE
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = []
    for n, (name, tpl) in enumerate(OPS):
        body.append(f"      if constexpr (OP == {n}) {{ " + " ".join(stmt(tpl, k) for k in range(8)) + " }")
    runs = "\n".join(f'  run<{n}>("{name}", d, clk);' for n, (name, _) in enumerate(OPS))
    src = f'''// GENERATED by tools/make_valu_probe.py -- what a wave64 VALU instruction costs on one SIMD of an MI355X, per mnemonic: eight independent chains of ONE
// instruction, 64 per loop trip, w waves per SIMD; cycles per instruction on the engine clock the kernel measures itself (s_memtime against the constant
// 100 MHz s_memrealtime).  hipcc --offload-arch=gfx950 -O2 -o valu_rate_probe valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* clk) {{
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float a[8];
  unsigned long long l[8];
  v2f p[8];
  for (int i = 0; i < 8; ++i) {{ a[i] = threadIdx.x * 1e-3f + i + 1.0f; l[i] = threadIdx.x + i; p[i] = v2f{{a[i], a[i] + 1}}; }}
  float b = 1.0001f, c = 0.9999f;
  unsigned long long m = 3, sm = 0x5555555555555555ull;
  float sc = 1.00001f;
  v2f q = {{b, c}};
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(b) : "vcc");
  for (int i = 0; i < iters; ++i) {{
#pragma unroll
    for (int u = 0; u < 8; ++u) {{
{chr(10).join(body)}
    }}
  }}
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)l[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {{ clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }}
}}
template <int OP>
void run(const char* name, float* d, unsigned long long* clk) {{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  std::printf("%-26s", name);
  for (int waves : {{1, 2, 4, 6}}) {{
    const int blocks = 256 * waves, iters = 8000;
    k<OP><<<blocks, 256>>>(d, 100, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = 100.0 * (double)h[0] / (double)h[1];
    std::printf("  %dw %5.2f", waves, ms * 1e-3 * mhz * 1e6 / ((double)iters * 64 * waves));
  }}
  std::printf("\\n");
}}
int main() {{
  float* d; (void)hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
  unsigned long long* clk; (void)hipMalloc(&clk, 16);
  std::printf("# cycles per wave64 instruction per SIMD, w waves per SIMD all issuing the same instruction (eight independent chains per lane)\\n");
{runs}
  return 0;
}}
'''
    open(os.path.join(root, "rene_amd", "csrc", "selftest", "valu_rate_probe.hip" if which == "rates" else "valu_decode_probe.hip"), "w").write(src)


if __name__ == "__main__":
    main()
