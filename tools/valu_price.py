#!/usr/bin/env python3
"""Prices the VALU instructions of a stretch of gfx950 assembly with the issue costs measured by selftest/valu_rate_probe (profiles/r03_valu_rates.txt):
what the stretch occupies a SIMD for, whatever the waits around it.
    tools/valu_price.py FILE.s [FIRST_LINE LAST_LINE]      (whole file if no range)
Classes (cycles per wave64 instruction per SIMD at two or more waves): FULL 2.4 -- v_add/sub/mul_f32, v_fmaak/fmamk, v_mov, v_add/sub_u32, v_and/or/xor,
v_lshrrev/ashrrev, v_cndmask_b32_e32; TRANS 8.2 -- rcp, rsq, sqrt, sin, cos, exp, log; HALF 4.2 -- everything else the probe measured (v_fma_f32 and
v_fmac_f32 with three register operands, v_pk_fma_f32, min / max / med3, compares, conversions, every VOP3 integer and 64-bit operation, v_cndmask_b32_e64)
and, by default, what it did not."""
import re
import sys

FULL = {"v_mul_f32_e32", "v_add_f32_e32", "v_sub_f32_e32", "v_subrev_f32_e32", "v_fmaak_f32", "v_fmamk_f32", "v_mul_f32_e64", "v_add_f32_e64", "v_sub_f32_e64",
        "v_mov_b32_e32", "v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_not_b32_e32",
        "v_lshrrev_b32_e32", "v_ashrrev_i32_e32", "v_cndmask_b32_e32"}
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_", "v_exp_", "v_log_")
COST = {"full": 2.4, "half": 4.2, "trans": 8.2}


def klass(m):
    if m in FULL:
        return "full"
    if m.startswith(TRANS):
        return "trans"
    return "half"


def price(lines):
    n = {"full": 0, "half": 0, "trans": 0}
    other = 0
    for l in lines:
        l = l.strip()
        m = re.match(r"^(v_[a-z0-9_]+)", l)
        if m and not m.group(1).startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            n[klass(m.group(1))] += 1
        elif re.match(r"^[a-z]", l) and not l.endswith(":"):
            other += 1
    tot = sum(n.values())
    cyc = sum(n[k] * COST[k] for k in n)
    return n, tot, cyc, other


def main():
    lines = open(sys.argv[1]).read().split("\n")
    if len(sys.argv) >= 4:
        lines = lines[int(sys.argv[2]) - 1:int(sys.argv[3])]
    n, tot, cyc, other = price(lines)
    print(f"VALU instructions {tot} (full rate {n['full']}, half rate {n['half']}, transcendental {n['trans']}), other instructions {other}: "
          f"{cyc:.0f} cycles of a SIMD, {cyc / max(1, tot):.2f} per VALU instruction")


if __name__ == "__main__":
    main()
