#!/usr/bin/env python3
"""Registers / spills / occupancy of every render kernel instantiation, from the .res files the build writes
(-Rpass-analysis=kernel-resource-usage).  python3 tools/res_summary.py [dir]"""
import os
import re
import subprocess
import sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rene_amd", "csrc")
for u in ("kernels", "kernels_bvh", "kernels_vol", "kernels_wave"):
    path = os.path.join(d, u + ".res")
    if not os.path.exists(path):
        continue
    t = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
    for b in blocks:
        name = b.split()[0]
        if "render_kernel" not in name and "trace_pass" not in name:
            continue
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"^void rene::", "", dem).split("(")[0]
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{u:12s} {dem:60s} sgpr {g('SGPRs'):>4} vgpr {g('VGPRs'):>4} agpr {g('AGPRs'):>3} scratch {scratch:>4} "
              f"sspill {g('SGPRs Spill'):>3} vspill {g('VGPRs Spill'):>3} occ {occ} lds {lds}")
