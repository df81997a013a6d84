#!/usr/bin/env python3
"""A/B of two checkouts on the same cut: times whole Cornell jobs (1024^2 @ 1024 spp) through whichever tree is the
current directory (both the round-2 and the round-3 Python API understand RENE_LEVELS and RENE_WORK_BATCH... the latter
only round 3).  cd <tree> && python3 <this file> LAUNCH_FRAMES LEVELS [NAME]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.getcwd())


def main():
    fpl, levels = int(sys.argv[1]), sys.argv[2]
    name = sys.argv[3] if len(sys.argv) > 3 else "cornell"
    os.environ["RENE_LEVELS"] = levels
    from rene_amd import api, scenes
    sc = {"cornell": lambda: scenes.cornell_box(1024, 1024), "veach-mis": lambda: scenes.veach_mis(1024, 1024)}[name]()
    spp = 1024
    with api.Renderer(sc) as r:
        r.render(0, 8)
        r.sync()
        ts = []
        for k in range(6):
            r.reset()
            t0 = time.perf_counter()
            for f0 in range(0, spp, fpl):
                r.render(f0, min(fpl, spp - f0))
            r.sync()
            ts.append(time.perf_counter() - t0)
        st = r.stats()
    print(f"{os.path.basename(os.getcwd()) or 'repo'}: {name} {spp // fpl} x {fpl} frames, RENE_LEVELS={levels}, batch {os.environ.get('RENE_WORK_BATCH', '-')}: "
          f"job {statistics.median(ts) * 1e3:.2f} ms (min {min(ts) * 1e3:.2f}), {st.rays / statistics.median(ts) / 1e6:.0f} Mrays/s", flush=True)


if __name__ == "__main__":
    main()
