#!/usr/bin/env python3
"""How a job of N frames should be cut into launches: times whole jobs (reset, launches, sync) of the bench configurations for
several cuts, overlapped or not.  gpurun -- python3 tools/job_shapes.py [NAME...]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rene_amd import abi, api
    import bench
    names = sys.argv[1:] or ["cornell", "dragon-class", "teapot-class"]
    cfgs = bench.configurations()
    for nm in names:
        lab, mk, spp, fpl = cfgs[nm]
        sc = mk()
        pk = sc if hasattr(sc, "byref") else sc.to_desc()
        cuts = [(fpl, abi.FLAG_OVERLAP), (fpl, 0), (spp, 0), (spp // 2, abi.FLAG_OVERLAP), (spp // 2, 0)]
        if os.environ.get("SHAPES"):  # e.g. SHAPES=64,128,256,512: overlapped launches of these sizes only
            cuts = [(int(x), abi.FLAG_OVERLAP) for x in os.environ["SHAPES"].split(",")]
        for cut, flags in cuts:
            cut = max(1, cut)
            with api.Renderer(pk, flags=flags) as r:
                r.tune(cut)
                ts = []
                for k in range(4 if nm != "teapot-class" else 2):
                    r.reset()
                    t0 = time.perf_counter()
                    for f0 in range(0, spp, cut):
                        r.render(f0, min(cut, spp - f0))
                    r.sync()
                    ts.append(time.perf_counter() - t0)
                st = r.stats()
            print(f"{nm}: {spp // cut:3d} launch(es) of {cut:5d} frames, {'overlapped' if flags else 'serial    '}: job {statistics.median(ts) * 1e3:9.2f} ms (min {min(ts) * 1e3:.2f}), "
                  f"{st.rays / statistics.median(ts) / 1e6:9.0f} Mrays/s", flush=True)


if __name__ == "__main__":
    main()
