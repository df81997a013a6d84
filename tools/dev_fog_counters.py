import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rene_amd import scenes, api, abi
from oracle import oracle
s = scenes.cornell_fog(96, 96)
o = oracle.Oracle(s); o.render(0, 16); so = o.stats().as_dict(); io = o.download(0)
keys = ("rays_closest", "rays_emitter", "rays_shadow", "hits", "adds")
print("oracle", {k: so[k] for k in keys})
for tag, fl in (("small", 0), ("bvh", abi.FLAG_FORCE_BVH)):
    with api.Renderer(s, flags=abi.FLAG_COUNTERS | fl) as r:
        r.render(0, 16); st = r.stats().as_dict(); im = r.download(0)
    rm = float(((im - io) ** 2).sum() / (io ** 2).sum())
    print(tag, {k: st[k] for k in keys}, "relMSE vs oracle", rm, "mean ratio", float(im.sum() / io.sum()))
