import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rene_amd import scenes, api, abi
s = scenes.material_zoo(64, 48)
force = abi.FLAG_FORCE_BVH
def run(flags, frames, first=0):
    with api.Renderer(s, flags=force | flags) as r:
        r.render(first, frames)
        return [r.download(k) for k in range(3)]
names = {0: "wave", abi.FLAG_WAVEFRONT: "wf", abi.FLAG_NO_RESTART: "ww"}
im = {n: run(f, 7) for f, n in names.items()}
for a, b in (("wave", "wf"), ("wave", "ww"), ("wf", "ww")):
    for k in range(3):
        d = np.argwhere(im[a][k] != im[b][k])
        print(a, b, "layer", k, "mismatches", len(d), d[:4].tolist())
# find the frame
d = np.argwhere(im["wave"][0] != im["wf"][0])
if len(d):
    y, x, c = d[0]
    for f in range(7):
        a = run(0, 1, f)[0][y, x]; b = run(abi.FLAG_WAVEFRONT, 1, f)[0][y, x]; w = run(abi.FLAG_NO_RESTART, 1, f)[0][y, x]
        print("frame", f, a, b, w, "equal" if (a == b).all() and (a == w).all() else "DIFF")
