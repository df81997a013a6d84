#!/usr/bin/env python3
"""GPU box: rank 0's share of a tile-sharded job (1 / N of the 32 x 32 tiles, all frames) as one launch, swept over the work-item knobs
(RENE_ITEM_FRAMES, RENE_ITEM_TAIL): how short a job's items and its halving tail should be when a rank's share is a few milliseconds.
    python3 tools/shard_sweep.py NAME N "item,tail item,tail ..." """
import os
import statistics
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(name, n):
    from rene_amd import abi, api
    import bench
    lab, mk, spp, fpl = bench.configurations()[name]
    sc = mk()
    pk = sc if hasattr(sc, "byref") else sc.to_desc()
    with api.Renderer(pk, shard_mode=abi.SHARD_TILES, shard_rank=0, shard_count=n) as r:
        r.render(0, 16)
        r.sync()
        ts = []
        for k in range(5):
            r.reset()
            t0 = time.perf_counter()
            r.render(0, spp)
            r.sync()
            ts.append(time.perf_counter() - t0)
        st = r.stats()
    print(f"{statistics.median(ts) * 1e3:8.2f} ms (min {min(ts) * 1e3:.2f}), {st.rays / statistics.median(ts) / 1e6:.0f} Mrays/s", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]))
    else:
        name, n, combos = sys.argv[1], sys.argv[2], sys.argv[3].split()
        for c in combos:
            item, tail = c.split(",")
            env = dict(os.environ)
            if item != "-":
                env["RENE_ITEM_FRAMES"] = item
            if tail != "-":
                env["RENE_ITEM_TAIL"] = tail
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, n], env=env, capture_output=True, text=True, timeout=300)
            print(f"{name} tiles 1/{n} item {item} tail {tail}: " + (p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]), flush=True)
