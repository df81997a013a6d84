#!/usr/bin/env python3
"""What one rank of `bench.py --gpus N` renders, timed on ONE GPU: rank 0's share of the strong-scaling Cornell job (1024^2 @ 1024
spp) for N = 1, 2, 4, 8 under both cuts -- contiguous frame blocks (every rank the whole image, 1024 / N frames) and 32x32 tiles
round-robin (every rank 1 / N of the pixels, all 1024 frames) -- as whole jobs (reset, one launch, sync).  The render share of
the scaling efficiency is t_1 / (N t_N); the exchange (one RCCL reduce of 50 MB, or a gather of 50 MB / N per rank) comes on
top and is NOT measured here (it needs N GPUs): DESIGN.md section 6 states what is assumed for it.
    gpurun -- python3 tools/scaling_model.py [NAME]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rene_amd import abi, api, dist as rdist
    import bench
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
    lab, mk, spp, fpl = bench.configurations()[name]
    sc = mk()
    pk = sc if hasattr(sc, "byref") else sc.to_desc()
    base = None
    for cut in ("frames", "tiles"):
        for n in (1, 2, 4, 8):
            shard = dict(shard_mode=abi.SHARD_TILES, shard_rank=0, shard_count=n if cut == "tiles" else 1)
            lo, hi = (0, spp) if cut == "tiles" else rdist.frame_block(0, n, spp)
            with api.Renderer(pk, **shard) as r:
                r.render(0, min(16, hi - lo))
                r.sync()
                ts = []
                for k in range(6):
                    r.reset()
                    t0 = time.perf_counter()
                    r.render(lo, hi - lo)
                    r.sync()
                    ts.append(time.perf_counter() - t0)
                st = r.stats()
            t = statistics.median(ts)
            if base is None:
                base = t
            print(f"{name} {cut:6s} N={n}: rank 0 renders {hi - lo} frames of {'1/%d of the tiles' % n if cut == 'tiles' else 'the whole image'}: "
                  f"{t * 1e3:8.2f} ms (min {min(ts) * 1e3:.2f}), {st.rays / t / 1e6:8.0f} Mrays/s on this rank; render efficiency t1 / (N tN) = {base / (n * t):.3f}", flush=True)


if __name__ == "__main__":
    main()
