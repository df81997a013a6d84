"""Developer script: GPU-vs-oracle difference statistics per scene (run on a GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rene_amd import scenes, api, abi
from oracle import oracle

def stats(name, s, frames):
    o = oracle.Oracle(s); o.render(0, frames)
    with api.Renderer(s) as r:
        r.render(0, frames); g = r.download(0)
    b = o.download(0)
    d = np.abs(g - b); rel = (d / (1 + np.abs(b))).max(axis=2)
    print(name, "relMSE", float(((g-b)**2).sum()/(b**2).sum()), "mean ratio", float(g.sum()/b.sum()),
          "frac>1e-2", (rel > 1e-2).mean(), ">5e-2", (rel > 5e-2).mean(), ">2e-1", (rel > 0.2).mean(), "exact", (d.max(axis=2) == 0).mean())
    # small relative errors everywhere? distribution of per-pixel relative error
    q = np.quantile(rel, [0.5, 0.9, 0.99, 0.999])
    print("   rel quantiles 50/90/99/99.9:", q)
    H = g.shape[0]
    for k in range(6):
        band = slice(k * H // 6, (k + 1) * H // 6)
        print("   band", k, "frac>1e-2", (rel[band] > 1e-2).mean(), "mean", g[band].mean(), b[band].mean())

stats("veach 160x90x16", scenes.veach_mis(160, 90), 16)
stats("zoo 96x64x32", scenes.material_zoo(96, 64), 32)
