#!/usr/bin/env python3
"""The J1 gate -- run on a GPU box.

north_star names a wavefront integrator with ray queues compacted in LDS; the megakernels keep path state in registers and their BVH steps run at
0.4 - 0.6 of the lanes.  Before any integrator is rebuilt around queues: what does the TRAVERSAL STAGE of such an integrator run at, on its own, on the
rays of a real frame mix, in the order a wavefront's passes would see them?

  1. rene_ray_dump: every query of F frames of the scene (dragon-class 1920 x 1080 by default), from the traversal-restart kernel itself;
  2. queue order: a pixel slot renders its frames one after the other, a bounce per round, so ray (pixel, frame f, depth d) is traced in round
     start(pixel, f) + d with start = the rounds the pixel's earlier frames took; within a round: all closest-hit queries in slot order (tile-major,
     8 x 8 micro-tiles: what a shading pass over the slots writes), then all shadow queries -- the two traversal passes of a bounce round;
  3. rene_trace_queue over that buffer: persistent pass, dead lanes refilled from the queue (ballot + prefix rank), fp32 origin + fp16 direction (J2's
     payload) or fp32 direction, hits written 16 bytes per ray; swept over the refill threshold, the leaf threshold and the occupancy;
  4. the hits are checked against rene_trace (the while-while probe) on a sample.

GATE: >= 30 Grays/s of traversal on the steady-state rounds (the megakernel's traversal share runs at the equivalent of ~ 19 - 24).
    python3 tools/j1_gate.py [dragon-class|teapot-class] [--frames 3] [--out gpurun_out/j1_gate.txt]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def slot_order(pix, W, H):
    """the kernels' pixel slot of a linear pixel index (image row-major, top row first): 32 x 32 tiles, 8 x 8 micro-tiles inside"""
    x, y = pix % W, pix // W
    tiles_x = (W + 31) // 32
    tile = (y // 32) * tiles_x + x // 32
    sub = ((y % 32) // 8) * 4 + (x % 32) // 8
    return (tile * 1024 + sub * 64 + (y % 8) * 8 + x % 8).astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", nargs="?", default="dragon-class")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    from rene_amd import abi, api
    import bench
    lab, mk, spp, fpl = bench.configurations()[a.name]
    sc = mk()
    pk = sc if hasattr(sc, "byref") else sc.to_desc()
    W, H = pk.xres, pk.yres
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    F = min(a.frames, 8)
    with api.Renderer(pk, flags=abi.FLAG_COUNTERS) as r:
        cap = int(W * H * F * 6.0)
        rays, issued = r.ray_dump(0, F, cap)
        st = r.stats().as_dict()
        assert issued <= cap, (issued, cap)
        assert issued == st["rays"], (issued, st["rays"])
        meta = rays[:, 7].view(np.uint32)
        pix, depth, any_hit, emit, frame = meta & 0x1FFFFF, (meta >> 21) & 63, (meta >> 27) & 1, (meta >> 28) & 1, (meta >> 29) & 7
        say(f"# {lab}: {issued} queries of {F} frames dumped ({issued / (W * H * F):.2f} per path): closest {int((any_hit == 0).sum())}, any-hit {int(any_hit.sum())}, emitter structure {int(emit.sum())}")
        # ---- queue order --------------------------------------------------------------------------------------------------------------------------------
        plen = np.zeros((F, W * H), np.int32)  # rounds each (frame, pixel) path takes = its deepest query + 1
        np.maximum.at(plen, (frame, pix), depth.astype(np.int32) + 1)
        start = np.concatenate([np.zeros((1, W * H), np.int64), np.cumsum(plen, axis=0)[:-1]], axis=0)
        rnd = start[frame, pix] + depth
        slot = slot_order(pix.astype(np.int64), W, H)
        order = np.lexsort((slot, any_hit, rnd))
        rays, rnd, any_hit, emit = rays[order], rnd[order], any_hit[order], emit[order]
        n = rays.shape[0]
        o_tmax = np.ascontiguousarray(rays[:, :4])
        flags = (any_hit | (emit << 1)).astype(np.uint32)
        d32 = np.ascontiguousarray(np.concatenate([rays[:, 4:7], flags.view(np.float32)[:, None]], axis=1))
        h = rays[:, 4:7].astype(np.float16).view(np.uint16).astype(np.uint32)
        d16 = np.ascontiguousarray(np.stack([h[:, 0] | (h[:, 1] << 16), h[:, 2] | (flags << 16)], axis=1).astype(np.uint32))
        counts = np.bincount(rnd.astype(np.int64))
        say("# rays per round: " + " ".join(str(int(c)) for c in counts[:16]) + " ...")
        steady = (rnd >= 3) & (rnd < max(4, min(len(counts) - 1, 3 * F)))  # rounds in which every slot is somewhere inside a path
        lo, hi = int(np.argmax(steady)), int(n - np.argmax(steady[::-1]))
        say(f"# steady-state section: rounds 3 .. {int(rnd[hi - 1])}, {hi - lo} rays (closest and shadow passes alternate, slots at mixed depths)")

        def run(sel, fp16, refill, leaf, bpc=0, hits=False):
            o, d = o_tmax[sel], (d16 if fp16 else d32)[sel]
            # the launch walks the queue several times over (RENE_GATE_PASSES): >= 40 M rays, so that each of the chip's ~ 4 000 waves takes twenty
            # chunks of 512 and the end of the launch is a twentieth of it
            passes = 1 if hits else max(1, int(np.ceil(40e6 / o.shape[0])))
            os.environ["RENE_GATE_PASSES"] = str(passes)
            ms, hh, steps = r.trace_queue(o, d, fp16, refill, leaf, bpc, 3, hits, True)
            nr = o.shape[0] * passes
            dens_n = steps[1] / max(1, 64 * steps[0])
            dens_l = steps[3] / max(1, 64 * steps[2])
            return ms, hh, (nr / ms / 1e6, dens_n, dens_l, steps[1] / nr, steps[3] / nr)

        # ---- correctness first: hits of the queue pass = rene_trace's, on a sample of closest-hit rays of the main structure -----------------------------
        smp = np.nonzero((any_hit == 0) & (emit == 0))[0][:: max(1, n // 200000)][:200000]
        ms, hh, _ = run(smp, False, 16, 6, hits=True)
        ref = r.trace(o_tmax[smp, :3], d32[smp, :3], 0.001, 1e5, 0)  # (the dumped closest queries all carry tmax 1e5)
        miss_q, miss_r = hh[:, 0] < 0, ref["t"] < 0
        same = (miss_q == miss_r) & (miss_q | (np.abs(hh[:, 0] - ref["t"]) <= 1e-6 * np.abs(ref["t"])))
        say(f"# check: {same.mean() * 100:.4f} % of {len(smp)} sampled closest-hit rays have rene_trace's hit (bit-equal t expected: same node test, same leaf test)")
        assert same.mean() > 0.9999
        # ---- the sweep -------------------------------------------------------------------------------------------------------------------------------------------
        sel_all, sel_steady = slice(0, n), slice(lo, hi)
        say("# section | payload | refill_min | leaf_min | blocks/CU | stack | Grays/s | node-step lanes | leaf-step lanes | node visits/ray | leaf visits/ray")
        combos = [(64, 6), (32, 6), (16, 6), (8, 6), (1, 6), (16, 1), (16, 16), (24, 12)]
        if a.quick:
            combos = [(64, 6), (16, 6)]
        for name, sel in (("steady", sel_steady), ("all", sel_all)):
            for fp16 in (True, False):
                for refill, leaf in combos:
                    ms, _, (g, dn, dl, nv, lv) = run(sel, fp16, refill, leaf)
                    say(f"{name:6s} | {'fp16 dir' if fp16 else 'fp32 dir'} | {refill:3d} | {leaf:3d} | auto | scene | {g:7.2f} | {dn:.3f} | {dl:.3f} | {nv:.2f} | {lv:.2f}")
                if a.quick:
                    break
        for bpc in (2, 3):  # fewer workgroups per CU: how much of the rate is occupancy
            ms, hh, (g, dn, dl, nv, lv) = run(sel_steady, True, 16, 6, bpc)
            say(f"steady | fp16 dir |  16 |   6 | {bpc} | scene | {g:7.2f} | {dn:.3f} | {dl:.3f} | {nv:.2f} | {lv:.2f}")
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
