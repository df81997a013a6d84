"""Developer script: what a rank of an 8-GPU job does -- 128 frames of Cornell 1024^2 -- as 1, 2 or 4 overlapped launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
s = scenes.cornell_box(1024, 1024)
for parts in (1, 2, 4):
    n = 128 // parts
    best = 1e9
    with api.Renderer(s, flags=abi.FLAG_OVERLAP) as r:
        r.tune(n)
        for rep in range(5):
            r.reset()
            t0 = time.perf_counter()
            for k in range(parts):
                r.render(k * n, n)
            r.sync()
            best = min(best, time.perf_counter() - t0)
        st = r.stats()
    print(f"{parts} x {n} frames: {best*1e3:.2f} ms -> {st.rays / best / 1e6:.0f} Mrays/s", flush=True)
