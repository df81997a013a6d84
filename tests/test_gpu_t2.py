"""-m gpu, tier T2: the HIP path rendered at rene's own resolution and a high sample count, pushed
through rene's output transform (average -> to_rgb8), against rene's OWN published Vulkan renders
(box-filtered 8x8 fixtures under tests/golden/, made by make_rene_image_fixtures.py).

rene's images are not unbiased (its Cornell differs from the Tungsten ground truth by 0.043 sRGB
RMSE, its veach-mis by 0.174 -- SURVEY.md section 6), so a small RMSE here means the GPU kernels
reproduce rene's arithmetic *including its quirks*, on real scenes, end to end."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from rene_amd import api, scenes

pytestmark = pytest.mark.gpu


def _box8(rgb8):
    a = rgb8.astype(np.float32) / 255.0
    h, w, _ = a.shape
    return a.reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3))


def _render(scene, spp, batch=128):
    with api.Renderer(scene) as r:
        for f in range(0, spp, batch):
            r.render(f, min(batch, spp - f))
        return api.to_rgb8(r.download(0), spp)


def test_t2_cornell_vs_renes_render():
    got = _box8(_render(scenes.cornell_box(1024, 1024), 2048))
    want = np.load(os.path.join(GOLDEN, "rene_cornell_box8.npy"))
    rmse = float(np.sqrt(((got - want) ** 2).mean()))
    print("T2 Cornell sRGB RMSE vs rene:", rmse)
    assert rmse < 0.0072  # measured 0.00575 (x 1.25); rene vs Tungsten: 0.043


def test_t2_veach_mis_vs_renes_render():
    got = _box8(_render(scenes.veach_mis(1280, 720), 4096))
    want = np.load(os.path.join(GOLDEN, "rene_veach_mis_box8.npy"))
    rmse = float(np.sqrt(((got - want) ** 2).mean()))
    ratio = float(got.mean() / want.mean())
    print("T2 veach-mis sRGB RMSE vs rene:", rmse, "mean ratio", ratio)
    assert rmse < 0.009 and abs(ratio - 1) < 0.012  # measured 0.00716 (x 1.25) / 0.9919; rene vs Tungsten: 0.174


# ---- rene's own sample count, 4 x 4 boxes, and the energy of every surface -----------------------------
def _t2_regions(name, scene, oracle_mod, spp=5000):
    import t2_regions as T
    srgb4, lin4 = T.rene_box4(name)
    with api.Renderer(scene) as r:
        r.render(0, spp)
        rgb8 = api.to_rgb8(r.download(0), spp)
    mine4 = T.box(rgb8.astype(np.float32) / 255.0, 4)
    rmse = float(np.sqrt(((mine4 - srgb4) ** 2).mean()))
    reg = T.region_map(oracle_mod, scene, 4)
    mine_lin4 = T.box(T.to_linear(rgb8.astype(np.float32) / 255.0), 4)
    rows = []
    for rid in np.unique(reg):
        m = reg == rid
        if rid < 0 or m.sum() < 150:
            continue
        a, b = mine_lin4[m].mean(axis=0), lin4[m].mean(axis=0)
        rows.append((int(rid) >> 12, int(rid) & 4095, int(m.sum()), b, a / np.maximum(b, 1e-9)))
    return rmse, rows


def _show(rows):
    return "\n".join(f"  instance {i} quad {q}: {c} cells, rene linear {np.round(b, 4)}, ratio {np.round(r, 4)}" for i, q, c, b, r in rows)


# Measured on the MI355X in round 3 (5000 frames, fixed seed schedule: reproducible): mean linear radiance of the region here /
# in rene's PNG, per channel, for the channels rene's 8 bits resolve (mean linear in [0.03, 0.9): a quantisation step is then
# below 3 % and the image's own noise dithers it).  (instance, quad) -> ratios.  The build is 1 - 4 % brighter than rene's
# published Cornell on the walls and the floor and 6.4 % on the ceiling (instance 1; there the ratio also varies over the
# surface, 0.95 near the light to 1.06 away from it) -- offsets of the published image, whose code version is unknown (the
# checkout has no history); the ORACLE shows the same offsets (tests/test_oracle_render.py), so they are not the HIP path's.
# The test pins every ratio to +- 1 %: a one-per-cent change of the energy of any one surface fails it.
T2_CORNELL_RATIOS = {
    (0, 0): (1.0205, 1.0169, None), (1, 0): (1.0638, 1.0647, None), (2, 0): (1.0326, 1.0315, 1.0266),
    (3, 0): (1.0284, 1.0276, None), (4, 0): (1.0373, None, None), (5, 1): (None, 1.0135, None),
    (6, 1): (1.0194, 1.0157, None), (6, 4): (0.9888, None, None),
}
T2_VEACH_RATIOS = {
    (0, 1): (0.9978, 0.9979, 0.9979), (0, 5): (1.0279, None, None), (1, 1): (0.9949, 0.9948, 0.9948), (1, 5): (1.0154, None, None),
    (2, 1): (0.9934, 0.9925, 0.9925), (2, 5): (1.0201, None, None), (3, 0): (0.9849, 0.9825, 0.9817), (4, 0): (0.9819, 0.9811, 0.9809),
    (8, 1): (0.9781, 0.9733, 0.9728),
}
T2_CORNELL_BOX4_RMSE = 0.0070  # measured 0.00565 (x 1.24); 8 x 8 boxes at 2048 spp: 0.0057
T2_VEACH_BOX4_RMSE = 0.0114    # measured 0.00919 (x 1.24)


def _check_regions(rows, expected):
    checked = 0
    for i, q, c, b, r in rows:
        for ch in range(3):
            if 0.03 <= b[ch] < 0.9:
                want = expected.get((i, q), (None, None, None))[ch]
                assert want is not None, f"region ({i}, {q}) channel {ch} (rene {b[ch]:.4f}, ratio {r[ch]:.4f}) has no recorded ratio"
                assert abs(r[ch] - want) < 0.01, (i, q, ch, float(r[ch]), want)
                checked += 1
    return checked


def test_t2_cornell_at_renes_5000_spp_box4_and_every_surface(oracle_mod):
    """rene's Cornell at rene's own sample count against rene's PNG: 4 x 4 box sRGB RMSE, and the mean linear radiance of every
    surface the camera sees (walls, floor, ceiling, the visible faces of both blocks; tests/t2_regions.py)."""
    rmse, rows = _t2_regions("cornell", scenes.cornell_box(1024, 1024), oracle_mod)
    print("T2 Cornell 5000 spp, 4 x 4 box sRGB RMSE vs rene:", rmse)
    print(_show(rows))
    assert rmse < T2_CORNELL_BOX4_RMSE
    assert _check_regions(rows, T2_CORNELL_RATIOS) >= 14


def test_t2_veach_mis_at_renes_5000_spp_box4_and_every_surface(oracle_mod):
    """The same for veach-mis (Metal plates, sphere emitters): every plate, the floor, the wall."""
    rmse, rows = _t2_regions("veach_mis", scenes.veach_mis(1280, 720), oracle_mod)
    print("T2 veach-mis 5000 spp, 4 x 4 box sRGB RMSE vs rene:", rmse)
    print(_show(rows))
    assert rmse < T2_VEACH_BOX4_RMSE
    assert _check_regions(rows, T2_VEACH_RATIOS) >= 20


# ---- tier T3: where things are in rene's published teapot render (tests/t3_geometry.py) --------------------------------------
def test_t3_teapot_geometry_vs_renes_render():
    """The HIP path's first-hit albedo of one frame of rene's own teapot scene file at its own 1280 x 720 against rene's published
    render of it, one bit per pixel: checkerboard phase on the floor, the teapot's outline (see test_oracle_render.py's twin)."""
    import t3_geometry as T
    with api.Renderer(scenes.teapot_full(1280, 720)) as r:
        r.render(0, 1)
        g = T.geometry(r.download(2), T.rene_teapot_bright())
    print("T3 teapot geometry, GPU vs rene:", g)
    assert g["floor_pixels"] > 600000 and g["teapot_pixels"] > 200000
    assert 1.0 - g["checker"] < 4.2e-4  # the oracle's figure: 3.3e-4 (same frame seed, same jitter)
    assert g["inside"] > 0.9687 and g["outside"] < 0.0102  # oracle: 0.9750 / 0.0081


def test_t2_dragon_lit_surfaces_vs_renes_render():
    """The HIP path on the 12 dragon meshes the checkout holds, at 1280 x 720 x 1024 spp, against rene's raw render of the whole
    scene: the directly lit surfaces (tests/t2_regions.py, dragon_lit_ratio) -- reference-held radiance for the distant light."""
    import t2_regions as T
    spp = 1024
    with api.Renderer(scenes.dragon_partial(1280, 720)) as r:
        r.render(0, spp)
        rgb8 = api.to_rgb8(r.download(0), spp)
        hit = np.abs(r.download(1)[..., :3]).sum(axis=2) > 0.5 * spp
    n, med, q1, q3, inside = T.dragon_lit_ratio(rgb8, hit)
    print(f"T2 dragon (partial), GPU vs rene: {n} lit cells, linear ratio median {med:.4f} quartiles {q1:.4f} / {q3:.4f}, inside rene's silhouette {inside:.4f}")
    assert n > 3000
    assert abs(med - 0.993) < 0.01 and q1 > 0.94 and q3 < 1.03  # the oracle at 32 spp: 0.9935, 0.9621 / 1.0074
    assert inside > 0.95
