"""Region-wise energy of a render against rene's own published image (tier T2).

An 8 x 8 box-filtered sRGB RMSE lets a one-per-cent energy error in one wall through.  Here the image is cut into the
scene's own surfaces -- every wall, every face of a block, the light, every plate and every light of veach-mis -- by
tracing the camera ray through the centre of every 4 x 4 cell of rene's image with the ORACLE (first hit: instance, quad),
dropping cells whose neighbours see something else, and the mean linear radiance of each region is compared with rene's
(its 8-bit PNG decoded to linear light, tests/golden/rene_<scene>_box4.npy).  Both sides go through the same output
transform (average -> gamma -> round to 8 bits -> clamp, main.rs:1758-1792) and the same decoding, so a saturated light is
saturated on both sides.  Used by test_oracle_render.py (oracle vs rene, CPU) and test_gpu_t2.py (HIP path vs rene).
"""
import os

import numpy as np

from conftest import GOLDEN


def to_linear(s):  # inverse of gamma_correct, main.rs:1768-1774
    return np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)


def rene_box4(name):
    """(sRGB means, linear means), each [H/4][W/4][3] f32, of rene's published render of `name` (cornell / veach_mis)."""
    a = np.load(os.path.join(GOLDEN, f"rene_{name}_box4.npy")).astype(np.float32)
    return a[0], a[1]


def box(a, k):
    h, w, c = a.shape
    return a.reshape(h // k, k, w // k, k, c).mean(axis=(1, 3))


def region_map(oracle_mod, scene_full, k=4):
    """[H/k][W/k] int32: the surface the centre of each k x k cell of the full-size image sees (instance << 12 | quad
    index, quad = two consecutive triangles of a mesh; -1 = nothing, or a cell whose 8 neighbours do not all agree)."""
    o = oracle_mod.Oracle(scene_full)
    W, H = (scene_full.xres, scene_full.yres) if hasattr(scene_full, "xres") else (scene_full.film.xresolution, scene_full.film.yresolution)
    ch, cw = H // k, W // k
    org = np.zeros((ch * cw, 3), np.float32)
    d = np.zeros((ch * cw, 3), np.float32)
    i = 0
    for cy in range(ch):
        yi = k * cy + (k - 1) / 2.0  # image row of the cell centre, top first; launch_id.y = H - 1 - row (lib.rs:178-179)
        for cx in range(cw):
            xi = k * cx + (k - 1) / 2.0
            org[i], d[i] = o.camera_ray((xi + 0.5) / (W - 1), (H - 1 - yi + 0.5) / (H - 1))
            i += 1
    h = o.trace(org, d)
    reg = np.where(h["t"] > 0, (h["instance"].astype(np.int64) << 12) | (h["primitive"].astype(np.int64) >> 1), -1).reshape(ch, cw)
    same = np.ones_like(reg, bool)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            sh = np.roll(np.roll(reg, dy, 0), dx, 1)
            same &= sh == reg
    same[0, :] = same[-1, :] = False
    same[:, 0] = same[:, -1] = False
    return np.where(same, reg, -1).astype(np.int64)


def region_ratios(mine_rgb8, rene_lin4, regions, min_cells=150):
    """{region: (cells, mean linear radiance here / rene's, rene's mean)} over regions of at least min_cells cells;
    mine_rgb8 = this build's image after rene's output transform at the region map's resolution times k (u8, or already box-k)."""
    mine = to_linear(mine_rgb8.astype(np.float32) / 255.0)
    if mine.shape[0] != regions.shape[0]:
        mine = box(mine, mine.shape[0] // regions.shape[0])
    out = {}
    for r in np.unique(regions):
        if r < 0:
            continue
        m = regions == r
        if m.sum() < min_cells:
            continue
        a, b = float(mine[m].mean()), float(rene_lin4[m].mean())
        out[int(r)] = (int(m.sum()), a / b, b)
    return out


def describe(ratios):
    return ", ".join(f"inst {r >> 12} quad {r & 4095}: {c} cells, rene {b:.4f}, ratio {q:.4f}" for r, (c, q, b) in sorted(ratios.items()))


# ---- rene's dragon render against the 12 of its 16 meshes the checkout holds (tests/golden/dragon_partial) -----------------------
def dragon_lit_ratio(mine_rgb8, hit_mask):
    """rene's images/dragon.png is a raw (not denoised) render of sample_scenes/dragon: Matte meshes under ONE distant light on a
    black background -- the only published image that reaches light.rs' distant light (lib.rs:234-272).  Four of the sixteen meshes
    are missing from the checkout (the body and two ground pieces), so shadows and interreflections differ wherever they matter;
    the DIRECTLY LIT surfaces of the meshes that are there do not depend on them to first order.  Over the 4 x 4 cells that this
    build's first hits cover completely (and whose neighbours they cover too) and that are bright but not saturated in rene's
    image (sRGB 0.5 .. 0.98), returns (cells, median of the per-cell ratio of linear radiance mine / rene, lower and upper quartile,
    fraction of the covered cells that lie inside rene's silhouette)."""
    from scipy import ndimage as ndi
    a = np.load(os.path.join(GOLDEN, "rene_dragon_box4.npy")).astype(np.float32)
    rene_srgb, rene_lin = a[0], a[1]
    mine = mine_rgb8.astype(np.float32).mean(axis=2, keepdims=True) / 255.0
    mine_lin = box(to_linear(mine), 4)[..., 0]
    covered = box(hit_mask[..., None].astype(np.float32), 4)[..., 0] == 1.0
    inner = ndi.binary_erosion(covered, iterations=1)
    m = inner & (rene_srgb > 0.5) & (rene_srgb < 0.98)
    r = mine_lin[m] / rene_lin[m]
    q = np.quantile(r, [0.25, 0.5, 0.75])
    return int(m.sum()), float(q[1]), float(q[0]), float(q[2]), float((rene_srgb[covered] > 0.02).mean())
