#!/usr/bin/env python3
"""Third companion of tools/cornell_offsets.py: is the offset against rene's Cornell image tied to WHERE the light branch is taken?
The scene's back / left / right walls carry outward normals (sample_scenes/cornell-box/scene.pbrt), floor, ceiling, blocks and light inward ones, and quirk Q1's
`bsdf.pdf(wi, normal)` is 0 on the former and 1 / pi on the latter -- the one place where the restatement's arithmetic depends on a normal's SIGN.  The oracle
decomposes every add by (bounce, instance at which the light branch that found the light was taken); each component is divided by its frames' count (the
frame-count model of tools/cornell_counts.py) and rene's image is regressed on the components grouped by that instance.  CPU only, ~ 2 min per master seed.
    python3 tools/cornell_vertices.py [--frames 5000] [--seeds 2]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
NAMES = {0: "floor", 1: "ceiling", 2: "back wall", 3: "right wall", 4: "left wall", 5: "short block", 6: "tall block", 7: "light"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=5000)
    ap.add_argument("--seeds", type=int, default=2)
    a = ap.parse_args()
    import cornell_counts as CC
    import cornell_offsets as co
    from oracle import oracle
    from rene_amd import scenes
    reg, rene = co.regions_and_rene("cornell", scenes.cornell_box(1024, 1024), oracle)
    o = oracle.Oracle(scenes.cornell_box(256, 256))
    N, K, D = a.frames, 12, 10
    E = np.zeros((D, K, 256, 256, 3))
    for seed in [0x52454E45, 0x12345678, 0x9E3779B9][:a.seeds]:
        o.reset()
        o.set_experiment(0, 2)
        o.render(0, N, seed=seed)
        j = CC.first_light(CC.frame_seeds(seed, N))
        n = np.array([(j == k).sum() for k in range(D)])
        for d in range(1, D):
            scale = n[d - 1] * 2.0 ** d / N if d < 9 and n[d - 1] else 1.0
            for k in range(K):
                E[d, k] += o.download_decomposition(d, k) / N / scale / a.seeds
    m = (reg >= 0)[..., None] & (rene < 0.9) & (rene > 0.01) & ((reg >> 12) != 7)[..., None]
    y = rene[m]
    W = 1 / y
    print(f"# oracle 256 x 256 @ {N} frames x {a.seeds} master seeds, count-normalised; rene's image regressed (relative least squares over {y.size} cell-channels)")
    share = {k: E[:, k].sum(axis=0)[m].sum() / E.sum(axis=(0, 1))[m].sum() for k in range(K)}
    keys = [k for k in range(K) if share[k] > 1e-3]
    print("# share of the image's energy found through a light branch taken at: " + ", ".join(f"{NAMES.get(k, 'BSDF-branch rays' if k == 11 else k)} {share[k] * 100:.1f} %" for k in keys))

    def fit(groups, label):
        G = np.stack([sum(E[:, k].sum(axis=0) for k in g)[m] for g in groups], axis=1)
        w = np.linalg.lstsq(G * W[:, None], y * W, rcond=None)[0]
        pred = G @ w
        full = sum(wk * sum(E[:, k].sum(axis=0) for k in g) for wk, g in zip(w, groups))
        rows = co.region_table(full, reg, rene)
        v = co.vector(rows, co.channels_used(rows))
        print(f"{label}: weights " + " ".join(f"{x:.3f}" for x in w) + f" | per-surface rms of (fitted / rene - 1) {np.sqrt(((v - 1) ** 2).mean()):.4f}, max {np.abs(v - 1).max():.4f}")
    fit([list(range(K))], "one weight                                                   ")
    fit([[2, 3, 4], [0, 1, 5, 6, 7], [11]], "outward-normal walls | inward-normal surfaces | BSDF rays       ")
    fit([[0], [1], [2, 3, 4], [5, 6], [7, 11]], "floor | ceiling | walls | blocks | rest                        ")
    fit([[0], [2], [3], [4], [5], [6], [1, 7, 11]], "floor | back | right | left | short | tall | rest              ")
    # the same with the bounce as a second factor: direct (the camera's own first hit takes the light branch) against later
    G = []
    for g in ([2, 3, 4], [0, 1, 5, 6, 7, 11]):
        for ds in ([1], [2], list(range(3, D))):
            G.append(sum(E[d, k] for k in g for d in ds)[m])
    G = np.stack(G, axis=1)
    w = np.linalg.lstsq(G * W[:, None], y * W, rcond=None)[0]
    print("(walls | others) x (add at bounce 1 | 2 | 3+): weights " + " ".join(f"{x:.3f}" for x in w))


if __name__ == "__main__":
    main()
