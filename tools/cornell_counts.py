#!/usr/bin/env python3
"""Companion of tools/cornell_offsets.py: is the 2 - 6 % offset between this build's Cornell and rene's published image the
sampling noise of rene's own estimator?  Quirk Q3 (one light / BSDF coin per FRAME, lib.rs:176, 276) makes the energy a 5000-frame image holds at
bounce d proportional to n_d, the number of its frames whose first light-branch coin falls at bounce d - 1 -- a Binomial(5000, 2^-d) number.  This
script (CPU only, oracle = test infrastructure)

  1. computes n_d for this build's seed schedule and checks the model on saved oracle decompositions (tools/cornell_offsets.py --save, several
     master seeds): per-depth components differ by up to 13 % between seeds and agree to 0.3 % once divided by n_d 2^d / N;
  2. regresses rene's image (4 x 4 boxes of its PNG, linear) on the count-normalised components E_d: the fitted nu_d = n_d(rene) 2^d / N against
     their binomial sigmas, and the sum rule sum_d nu_d 2^-d = 1 (every frame has exactly one first light bounce);
  3. compares the pixel noise of rene's PNG with this build's 8-bit image at 5000 spp (tools/t2_full_dump.py) surface by surface.

  python3 tools/cornell_counts.py gpurun_out/r4a/comps_0x52454E45.npz [more npz ...] [--full gpurun_out/r4a/t2full/cornell_full_seed0.npz]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
M32 = np.uint64(0xFFFFFFFF)
A, C = np.uint64(747796405), np.uint64(2891336453)


def step(s):
    return (s * A + C) & M32


def out(s):  # rand.rs:19-22
    w = (((s >> ((s >> np.uint64(28)) + np.uint64(4))) ^ s) * np.uint64(277803737)) & M32
    return ((w >> np.uint64(22)) ^ w) & M32


def new(seed):  # rand.rs:24-30
    seed = seed.astype(np.uint64)
    s = step(seed)
    return step((s + seed) & M32)


def frame_seeds(master, n):  # frame k = k-th output of PCG32si::new(master)
    s, res = new(np.array([master], dtype=np.uint64)), []
    for _ in range(n):
        res.append(out(s)[0])
        s = step(s)
    return np.array(res, dtype=np.uint64)


def first_light(seeds, maxd=40):
    """per frame: the bounce index of its first light-branch coin (coin k of a frame is draw k + 1 of PCG32si::new(seed) while every earlier coin chose BSDF)"""
    s, j = new(seeds), np.full(len(seeds), -1)
    for k in range(maxd):
        coin = (out(s) >> np.uint64(8)).astype(np.float64) * 2.0 ** -24
        s = step(s)
        hit = (coin > 0.5) & (j < 0)
        j[hit] = k
    return j


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("npz", nargs="+")
    ap.add_argument("--full", default="")
    ap.add_argument("--save-components", default="", help="npy: the count-normalised components by add depth, mean over the inputs (tools/cornell_mc.py)")
    a = ap.parse_args()
    import cornell_offsets as co
    from oracle import oracle
    from rene_amd import scenes
    reg, rene = co.regions_and_rene("cornell", scenes.cornell_box(1024, 1024), oracle)
    E, raw = [], []
    print("# 1. frame counts of this build's seed schedule, n_d / (N 2^-d), d = 1..7; per-depth components before / after dividing by them")
    for p in a.npz:
        d = np.load(p)
        N, seed = int(d["frames"]), int(d["seed"])
        Cd = d["comps"].reshape(10, 2, *d["comps"].shape[1:]).sum(axis=1)
        j = first_light(frame_seeds(seed, N))
        n = np.array([(j == k).sum() for k in range(9)])
        print(f"master {seed:#010x}: " + " ".join(f"{x:.3f}" for x in n[:7] / (N / 2.0 ** np.arange(1, 8))))
        En = Cd.copy()
        for dd in range(1, 9):
            En[dd] = Cd[dd] / (n[dd - 1] * 2.0 ** dd / N)
        raw.append(Cd)
        E.append(En)
    ids = {int(r) >> 12: r for r in np.unique(reg) if r >= 0 and (int(r) & 4095) == 0}
    for nm, inst in (("floor", 0), ("ceiling", 1), ("back wall", 2)):
        for dd in (1, 2, 3, 4):
            m = reg == ids[inst]
            print(f"  {nm:9s} add at bounce {dd}: raw R " + " ".join(f"{r[dd][m][:, 0].mean():.5f}" for r in raw) + "   / count: " + " ".join(f"{e[dd][m][:, 0].mean():.5f}" for e in E))
    E = np.mean(E, axis=0)
    if a.save_components:
        np.save(a.save_components, E)
    N = 5000
    m = (reg >= 0)[..., None] & (rene < 0.9) & (rene > 0.01) & ((reg >> 12) != 7)[..., None]
    groups = [[1], [2], [3], [4], [5], [6, 7, 8, 9]]
    G = np.stack([E[g].sum(axis=0)[m] for g in groups], axis=1)
    y = rene[m]
    W = 1 / y
    nu = np.linalg.lstsq(G * W[:, None], y * W, rcond=None)[0]
    sig = [np.sqrt((1 - 2.0 ** -d) / (N * 2.0 ** -d)) for d in range(1, 6)] + [np.sqrt(64.0 / N)]
    print("# 2. rene's image regressed on the count-normalised components (relative least squares over", y.size, "cell-channels)")
    print("nu_d, d = 1..5, 6+   :", " ".join(f"{x:.4f}" for x in nu))
    print("binomial sigma       :", " ".join(f"{x:.4f}" for x in sig))
    print("z = (nu - 1) / sigma :", " ".join(f"{(x - 1) / s:+.2f}" for x, s in zip(nu, sig)))
    print(f"sum rule sum nu_d 2^-d = {sum(nu[k] * 2.0 ** -(k + 1) for k in range(5)) + nu[5] * 2.0 ** -5:.4f} (1 for any 5000 frames of this estimator)")
    full = sum(nu[k] * E[g].sum(axis=0) for k, g in enumerate(groups))
    rows = co.region_table(full, reg, rene)
    keys = co.channels_used(rows)
    print("per region-channel, expectation / rene        :", co.fmt(co.vector(co.region_table(E[1:].sum(axis=0), reg, rene), keys)))
    print("per region-channel, with the fitted counts    :", co.fmt(co.vector(rows, keys)))
    if a.full:
        from PIL import Image
        from scipy import ndimage as ndi
        import t2_regions as T
        ref = os.path.join("/root/reference/images/cornell-box.png")
        if not os.path.exists(ref):
            print("# 3. skipped: rene's PNG is not on this machine")
            return
        print("# 3. pixel noise, std of (pixel - mean of its 8 neighbours) / surface mean, linear light, both images 8-bit after rene's output transform")
        rl = T.to_linear(np.asarray(Image.open(ref)).astype(np.float32) / 255.0)
        ol = T.to_linear(np.load(a.full)["rgb8"].astype(np.float32) / 255.0)
        k = np.ones((3, 3), np.float32) / 8
        k[1, 1] = 0

        def hp(x):
            return (x - np.stack([ndi.convolve(x[..., c], k, mode="nearest") for c in range(3)], axis=2)) / np.sqrt(1 + 1 / 8)
        hr, ho = hp(rl), hp(ol)
        regfull = np.kron(reg, np.ones((4, 4), np.int64))
        for rid in np.unique(reg):
            if rid < 0 or (reg == rid).sum() < 300:
                continue
            mm = ndi.binary_erosion(regfull == rid, iterations=8)
            sr, so = hr[mm].std(axis=0) / rl[mm].mean(axis=0), ho[mm].std(axis=0) / ol[mm].mean(axis=0)
            print(f"  instance {rid >> 12} quad {rid & 4095}: rene {np.round(sr, 4)} this build {np.round(so, 4)} variance ratio {np.round((sr / so) ** 2, 3)} absolute-noise ratio {np.round(hr[mm].std(axis=0) / ho[mm].std(axis=0), 3)}")


if __name__ == "__main__":
    main()
