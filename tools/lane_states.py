#!/usr/bin/env python3
"""Where the lanes are: one job of a bench configuration on the counting variant of its kernel (RENE_FLAG_COUNTERS), for
several work-item cuts; the library prints the lane states per pass / step under RENE_DEBUG (rene_get_stats).
    gpurun -- python3 tools/lane_states.py NAME [SPEC...]      SPEC as in tools/job_shapes.py, e.g. 1024:i64/64"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RENE_DEBUG"] = "1"


def main():
    from rene_amd import abi, api
    import bench
    nm = sys.argv[1]
    lab, mk, spp, fpl = bench.configurations()[nm]
    sc = mk()
    pk = sc if hasattr(sc, "byref") else sc.to_desc()
    for spec in sys.argv[2:] or [f"{spp}:i64/64"]:
        parts = spec.split(":")
        cut, items = int(parts[0]), parts[1]
        for k in ("RENE_LEVELS", "RENE_ITEM_FRAMES", "RENE_ITEM_TAIL"):
            os.environ.pop(k, None)
        flags = abi.FLAG_COUNTERS
        if items[0] == "i":
            os.environ["RENE_ITEM_FRAMES"], os.environ["RENE_ITEM_TAIL"] = items[1:].split("/")
        elif items[0] == "u":
            os.environ["RENE_LEVELS"] = items[1:]
        else:
            flags |= abi.FLAG_SINGLE_LEVEL
        print(f"== {nm} {spec} (batch {os.environ.get('RENE_WORK_BATCH', 'default')})", file=sys.stderr, flush=True)
        with api.Renderer(pk, flags=flags) as r:
            t0 = time.perf_counter()
            for f0 in range(0, spp, cut):
                r.render(f0, min(cut, spp - f0))
            r.sync()
            dt = time.perf_counter() - t0
            st = r.stats()
        print(f"   job {dt * 1e3:.2f} ms, {st.rays / dt / 1e6:.0f} Mrays/s (counting variant)", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
