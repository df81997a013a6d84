"""Developer script: many short overlapped launches (sizes 1..9 frames, three scene classes) against the same
launches one at a time: images and counters must be identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import scenes, api, abi

rng = np.random.default_rng(1)
for nm, s in (("cornell", scenes.cornell_box(512, 512)), ("dragon", scenes.dragon_class(480, 270, 60, 66)), ("fog", scenes.cornell_fog(256, 256)),
              ("zoo", scenes.material_zoo(256, 192))):
    sizes = rng.integers(1, 10, 300).tolist()
    out = []
    for flags in (0, abi.FLAG_OVERLAP):
        with api.Renderer(s, flags=flags | abi.FLAG_COUNTERS) as r:
            t0 = time.perf_counter()
            f = 0
            for k, n in enumerate(sizes):
                r.render(f, n)
                f += n
                if k % 97 == 96:
                    r.sync()  # joins in the middle
            imgs = [r.download(l) for l in range(3)]
            st = r.stats().as_dict()
            out.append((imgs, st, time.perf_counter() - t0))
    same = all(np.array_equal(a, b) for a, b in zip(out[0][0], out[1][0]))
    keys = ("paths", "rays_closest", "rays_shadow", "rays_emitter", "adds", "hits")
    print(nm, "identical:", same, "counters equal:", all(out[0][1][k] == out[1][1][k] for k in keys),
          f"serial {out[0][2]*1e3:.0f} ms, overlapped {out[1][2]*1e3:.0f} ms", flush=True)
