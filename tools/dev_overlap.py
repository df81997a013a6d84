"""Developer script: RENE_FLAG_OVERLAP at full size -- wall-clock throughput with and without overlapping
launches, and that the two images are identical bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi

CASES = {"cornell": (lambda: scenes.cornell_box(1024, 1024), 64), "veach": (lambda: scenes.veach_mis(1024, 1024), 64),
         "dragon": (lambda: scenes.dragon_class(1920, 1080), 16), "teapot": (lambda: scenes.teapot_class(1920, 1080), 16),
         "zoo": (lambda: scenes.material_zoo(1024, 768), 32), "fog": (lambda: scenes.cornell_fog(1024, 1024), 16)}
names = sys.argv[1:] or list(CASES)
for nm in names:
    mk, F = CASES[nm]
    s = mk()
    res = []
    for flags in (0, abi.FLAG_OVERLAP):
        with api.Renderer(s, flags=flags) as r:
            r.render(0, 4); r.sync(); r.reset()
            if os.environ.get("RENE_DEV_TUNE") == "1":
                r.tune(F)
            t0 = time.perf_counter()
            for k in range(6):
                r.render(k * F, F)
            r.sync()
            wall = time.perf_counter() - t0
            st = r.stats()
            res.append((st.rays / wall / 1e6, wall * 1e3 / 6, st.kernel_ms / 6, r.download(0)))
    same = np.array_equal(res[0][3], res[1][3])
    print(f"{nm}: serial {res[0][0]:.0f} Mrays/s ({res[0][1]:.2f} ms/launch wall, {res[0][2]:.2f} ms events) | overlap {res[1][0]:.0f} Mrays/s "
          f"({res[1][1]:.2f} ms/launch wall, {res[1][2]:.2f} ms events) | x{res[1][0]/res[0][0]:.3f} | identical: {same}", flush=True)
