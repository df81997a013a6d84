import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rene_amd import scenes, api, abi
from oracle import oracle
for name, s in (("fog", scenes.cornell_fog(64, 64)), ("cornell", scenes.cornell_box(64, 64)), ("veach", scenes.veach_mis(64, 64))):
    o = oracle.Oracle(s)
    rng = np.random.default_rng(5)
    n = 200000
    if name == "veach":
        org = np.stack([rng.uniform(-10, 10, n), rng.uniform(-3, 8, n), rng.uniform(-8, 8, n)], 1).astype(np.float32)
    else:
        org = np.stack([rng.uniform(-.99, .99, n), rng.uniform(0.01, 1.97, n), rng.uniform(-.99, .99, n)], 1).astype(np.float32)
    d = rng.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    with api.Renderer(s) as r:
        hs, ho = r.trace(org, d), o.trace(org, d)
    same = (hs["t"] >= 0) & (ho["t"] >= 0) & (hs["primitive"] == ho["primitive"]) & (hs["instance"] == ho["instance"])
    du, dv = np.abs(hs["u"] - ho["u"])[same], np.abs(hs["v"] - ho["v"])[same]
    bad = (du > 1e-3) | (dv > 1e-3)
    print(name, "same prim:", int(same.sum()), "uv mismatches:", int(bad.sum()))
    idx = np.where(same)[0][bad][:6]
    for i in idx: print("   inst", hs[i]["instance"], "prim", hs[i]["primitive"], "small uv", hs[i]["u"], hs[i]["v"], "oracle uv", ho[i]["u"], ho[i]["v"])
    if bad.sum():
        ii = np.where(same)[0][bad]
        print("   by (inst, prim):", sorted(set(zip(hs[ii]["instance"].tolist(), hs[ii]["primitive"].tolist())))[:40])
