#!/usr/bin/env python3
"""How a job of N frames should be cut into launches and work items: times whole jobs (reset, launches, sync) of the bench
configurations for several cuts.  gpurun -- python3 tools/job_shapes.py [NAME...]

A cut is  <frames per launch>:<items>  with <items> = i<frames>/<tail> (work items of <frames> frames, the last ones halving
down to <tail> frames: RENE_ITEM_FRAMES / RENE_ITEM_TAIL), u<levels> (uniform items, RENE_LEVELS) or w (one item per pixel
and launch).  SHAPES=256:u4,1024:i64/8,... overrides the built-in list."""
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
        spp = int(os.environ.get("SPP", spp))  # e.g. SPP=128: one rank's share of the 8-GPU strong-scaling job
        sc = mk()
        pk = sc if hasattr(sc, "byref") else sc.to_desc()
        small = nm in ("cornell", "veach-mis")
        cuts = [f"{fpl}:u{4 if small else 8}", f"{spp}:i64/64", f"{spp}:i64/8", f"{spp}:i32/32", f"{spp}:i32/4", f"{spp}:i16/4", f"{spp}:i128/8"]
        if os.environ.get("SHAPES"):
            cuts = os.environ["SHAPES"].split(",")
        for spec in cuts:
            parts = spec.split(":")
            cut, items = max(1, int(parts[0])), parts[1]
            for k in ("RENE_LEVELS", "RENE_ITEM_FRAMES", "RENE_ITEM_TAIL"):
                os.environ.pop(k, None)
            flags = 0
            if items[0] == "i":
                os.environ["RENE_ITEM_FRAMES"], os.environ["RENE_ITEM_TAIL"] = items[1:].split("/")
            elif items[0] == "u":
                os.environ["RENE_LEVELS"] = items[1:]
            else:
                flags |= abi.FLAG_SINGLE_LEVEL
            with api.Renderer(pk, flags=flags) as r:
                r.render(0, 8)
                r.sync()
                ts = []
                for k in range(5 if nm != "teapot-class" else 2):
                    r.reset()
                    t0 = time.perf_counter()
                    for f0 in range(0, spp, cut):
                        r.render(f0, min(cut, spp - f0))
                    r.sync()
                    ts.append(time.perf_counter() - t0)
                st = r.stats()
            print(f"{nm}: {spec:12s} {-(-spp // cut):3d} launch(es): job {statistics.median(ts) * 1e3:9.2f} ms (min {min(ts) * 1e3:.2f}), "
                  f"{st.rays / statistics.median(ts) / 1e6:9.0f} Mrays/s, replays {st.launches - (-(-spp // cut))}", flush=True)


if __name__ == "__main__":
    main()
