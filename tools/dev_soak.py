"""Developer script: the two 8-GPU north-star scene classes at their full sample counts on one GPU (overlapped, tuned):
no hand-off may time out, the image must be finite."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi
for nm, s, spp, F in (("dragon-class 1920x1080 @ 1024 spp", scenes.dragon_class(1920, 1080), 1024, 64),
                      ("teapot-class 1920x1080 @ 2048 spp", scenes.teapot_class(1920, 1080), 2048, 128)):
    with api.Renderer(s, flags=abi.FLAG_OVERLAP) as r:
        r.tune(F)
        t0 = time.perf_counter()
        for f in range(0, spp, F):
            r.render(f, F)
        r.sync()
        dt = time.perf_counter() - t0
        st = r.stats()  # raises if a hand-off timed out
        img = r.download(0)
        print(f"{nm}: {dt:.2f} s, {st.rays / dt / 1e6:.0f} Mrays/s, {dt * 1e3 / spp:.3f} ms/frame, finite {bool(np.isfinite(img).all())}, mean {img.mean() / spp:.4f}", flush=True)
