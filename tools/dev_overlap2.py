"""Developer script: one overlapped zoo run with a configurable size / frames per launch / launch count."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
w, h, F, n = (int(x) for x in sys.argv[1:5])
s = scenes.material_zoo(w, h)
with api.Renderer(s, flags=abi.FLAG_OVERLAP) as r:
    r.render(0, 4); r.sync(); r.reset()
    t0 = time.perf_counter()
    for k in range(n):
        r.render(k * F, F)
    r.sync()
    wall = time.perf_counter() - t0
    try:
        st = r.stats()
        print(sys.argv[1:], f"ok {st.rays / wall / 1e6:.0f} Mrays/s wall {wall*1e3/n:.2f} ms/launch events {st.kernel_ms/n:.2f}", flush=True)
    except Exception as e:
        print(sys.argv[1:], f"FAILED after {wall:.1f} s: {e}", flush=True)
