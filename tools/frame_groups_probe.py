#!/usr/bin/env python3
"""Is a BVH job bound by its heaviest pixel's chain of frames?  (docs/history.md section 4f: a pixel's frames are rendered one after the other,
so that the sums are added in the reference's order; the job cannot end before its most expensive pixel has gone through all of them.)
Renders one job of a bench configuration (a) as one context, one launch, and (b) as G contexts on the same GPU, context g rendering the
frames f with f % G == g into an image of its own, each launch sized to 1 / G of the chip's workgroup slots (RENE_BLOCKS_PER_CU) so that
the G launches are resident together; the G images are then summed.  The launches of (b) share nothing: none waits for another.
    gpurun -- python3 tools/frame_groups_probe.py NAME [G]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    from rene_amd import abi, api
    import bench
    nm = sys.argv[1]
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    lab, mk, spp, fpl = bench.configurations()[nm]
    sc = mk()
    pk = sc if hasattr(sc, "byref") else sc.to_desc()
    per_cu = 4  # workgroups per CU of the restart kernels (LDS: 40 KB each)

    def job(rs):
        for r in rs:
            r.reset()
        t0 = time.perf_counter()
        for r in rs:
            r.render(0, spp)
        for r in rs:
            r.sync()
        return time.perf_counter() - t0

    os.environ.pop("RENE_BLOCKS_PER_CU", None)
    with api.Renderer(pk) as r:
        job([r])
        ta = min(job([r]) for _ in range(2))
        rays = r.stats().rays
        ref = r.download(0)
    os.environ["RENE_BLOCKS_PER_CU"] = per = str(max(1, per_cu // G))
    rs = [api.Renderer(pk, shard_mode=abi.SHARD_FRAMES, shard_rank=g, shard_count=G) for g in range(G)]
    try:
        job(rs)
        tb = min(job(rs) for _ in range(2))
        rays_b = sum(r.stats().rays for r in rs)
        img = sum(r.download(0).astype(np.float64) for r in rs)
    finally:
        for r in rs:
            r.close()
    rel = float(np.abs(img - ref).max() / max(1e-30, np.abs(ref).max()))
    # (c) RENE_FLAG_FRAME_GROUPS: the two chains inside ONE launch (render_wf.inc), one work counter
    os.environ.pop("RENE_BLOCKS_PER_CU", None)
    with api.Renderer(pk, flags=abi.FLAG_FRAME_GROUPS) as r:
        job([r])
        tc = min(job([r]) for _ in range(2))
        rays_c = r.stats().rays
        img_c = r.download(0)
    rel_c = float(np.abs(img_c - ref).max() / max(1e-30, np.abs(ref).max()))
    print(f"{nm}: RENE_FLAG_FRAME_GROUPS, one launch: {tc * 1e3:.1f} ms ({rays_c / tc / 1e6:.0f} Mrays/s), rays {rays_c}, max |image - default image| / max = {rel_c:.2e}", flush=True)
    print(f"{nm}: one context {ta * 1e3:.1f} ms ({rays / ta / 1e6:.0f} Mrays/s) | {G} contexts, frames f % {G} == g, {per} workgroups per CU each: "
          f"{tb * 1e3:.1f} ms ({rays_b / tb / 1e6:.0f} Mrays/s), rays {rays_b} vs {rays}, max |sum of the partial images - image| / max = {rel:.2e}", flush=True)


if __name__ == "__main__":
    main()
