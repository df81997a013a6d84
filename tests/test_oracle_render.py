"""Render-level pins of the oracle: regression fixture, determinism, sharding, multi-launch
additivity, and (when the reference checkout is present) the T2 comparison with rene's own
published Cornell render."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REFERENCE, have_reference
from rene_amd import abi, scenes


@pytest.fixture(scope="module")
def cornell64(oracle_mod):
    return oracle_mod.Oracle(scenes.cornell_box(64, 64))


def test_matches_committed_fixture(cornell64):
    o = cornell64
    o.reset()
    o.render(0, 4)
    got = np.stack([o.download(l) for l in range(3)])
    want = np.load(os.path.join(GOLDEN, "cornell_64x64_4spp_layers.npy"))
    # libm sin/cos may differ by an ulp across glibc builds; path forks are then possible but rare
    bad = np.abs(got - want) > 1e-4 * (1 + np.abs(want))
    assert bad.mean() < 1e-3
    st = o.stats().as_dict()
    ref = json.load(open(os.path.join(GOLDEN, "cornell_64x64_4spp_stats.json")))
    for k, v in ref.items():
        assert abs(st[k] - v) <= max(4, 1e-4 * v), k


def test_threads_and_launch_split_are_bit_identical(cornell64):
    o = cornell64
    o.reset(); o.render(0, 6, threads=1); a = o.download(0)
    o.reset(); o.render(0, 6, threads=4); b = o.download(0)
    o.reset(); o.render(0, 2); o.render(2, 4); c = o.download(0)
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_seed_schedule(cornell64, oracle_mod):
    # frame k uses the k-th next_u32 of PCG32si::new(master): rendering frame 3 alone
    # equals the difference of [0,4) and [0,3) only in exact arithmetic, so test via a 1-frame scene
    o = cornell64
    o.reset(); o.render(3, 1, seed=99); a = o.download(0)
    o.reset(); o.render(3, 1, seed=99); b = o.download(0)
    o.reset(); o.render(3, 1, seed=100); c = o.download(0)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_experiment_switches_are_off_by_default_and_the_decomposition_sums_to_the_image(cornell64):
    """Round 4 (tools/cornell_offsets.py): the oracle-only experiment switches leave the restatement alone -- with none set the image is bit for
    bit the one without the machinery -- and the optional decomposition by (bounce of the add, branch that chose the ray) is a partition of layer 0.
    A switch that IS set changes the image (Q1 undone: the light branch's pdf argument order)."""
    o = cornell64
    o.reset(); o.render(0, 6); plain = o.download(0)
    o.reset(); o.set_experiment(0, True); o.render(0, 6); with_decomp = o.download(0)
    parts = sum(o.download_decomposition(d, br).astype(np.float64) for d in range(10) for br in (0, 1))
    assert np.array_equal(plain, with_decomp)
    assert np.allclose(parts, plain, rtol=1e-5, atol=1e-6)
    assert o.download_decomposition(0, 0).sum() == 0  # nothing is added at bounce 0 through a light-branch ray: the camera ray is not one
    assert o.download_decomposition(1, 0).sum() > 10 * o.download_decomposition(1, 1).sum()  # direct light arrives through the light branch
    o.reset(); o.set_experiment(o.X["pdf_wo_wi"]); o.render(0, 6); q1 = o.download(0)
    o.reset(); o.set_experiment(0); o.render(0, 6); again = o.download(0)
    assert not np.array_equal(q1, plain) and np.array_equal(again, plain)


def test_frame_count_model_of_quirk_q3(oracle_mod):
    """Quirk Q3 (lib.rs:176, 276): one light / BSDF coin per FRAME.  So the energy an image holds at bounce d is proportional to n_d, the number of its
    frames whose first light-branch coin falls at bounce d - 1 -- the reason a 5000-frame image's surfaces scatter by per cents between seeds
    (profiles/r04_cornell_offsets.txt).  Here: the direct light (adds at bounce 1 through the light branch) of two master seeds differs like their n_1,
    and agrees once divided by it."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cornell_counts as CC
    N = 384
    o = oracle_mod.Oracle(scenes.cornell_box(48, 48))
    direct, n1 = [], []
    for seed in (0x52454E45, 0x0BADCAFE):
        o.reset(); o.set_experiment(0, True); o.render(0, N, seed=seed)
        direct.append(float(o.download_decomposition(1, 0).sum()))
        n1.append(int((CC.first_light(CC.frame_seeds(seed, N)) == 0).sum()))
    assert n1[0] != n1[1]
    raw, norm = direct[0] / direct[1], (direct[0] / n1[0]) / (direct[1] / n1[1])
    assert abs(raw - n1[0] / n1[1]) < 0.02 and abs(norm - 1.0) < 0.02, (raw, n1, norm)


@pytest.mark.parametrize("mode", [abi.SHARD_TILES, abi.SHARD_FRAMES])
def test_shards_sum_to_whole(oracle_mod, mode):
    s = scenes.cornell_box(96, 80)  # ragged: 3 x 3 tiles with partial edge tiles
    o = oracle_mod.Oracle(s)
    o.render(0, 4); whole = [o.download(l) for l in range(3)]
    parts = None
    for r in range(3):
        o.reset(); o.render(0, 4, shard_mode=mode, shard_rank=r, shard_count=3)
        cur = [o.download(l) for l in range(3)]
        parts = cur if parts is None else [p + c for p, c in zip(parts, cur)]
    for w, p in zip(whole, parts):
        if mode == abi.SHARD_TILES:
            assert np.array_equal(w, p)  # every pixel owned once: adding zeros is exact
        else:
            np.testing.assert_allclose(w, p, rtol=1e-5, atol=1e-6)  # fp32 summation order differs


def test_tile_owner_map_matches_oracle(oracle_mod):
    from rene_amd.dist import tile_owner_map
    s = scenes.cornell_box(96, 80)
    o = oracle_mod.Oracle(s)
    own = tile_owner_map(96, 80, 3)
    for r in range(3):
        o.reset(); o.render(0, 1, shard_mode=abi.SHARD_TILES, shard_rank=r, shard_count=3)
        nrm = o.download(1)  # first-hit normal layer: non-zero wherever a path hit geometry
        touched = np.abs(nrm).sum(axis=2) > 0
        assert not (touched & (own != r)).any()


def test_image_statistics(cornell64):
    o = cornell64
    o.reset(); o.render(0, 16)
    img = o.download(0) / 16
    assert np.isfinite(img).all() and (img >= 0).all()
    # left wall red, right wall green (scene.pbrt:8-9), emitter row saturates
    left, right = img[24:40, 1:6].mean(axis=(0, 1)), img[24:40, 58:63].mean(axis=(0, 1))
    assert left[0] > 3 * left[1] and right[1] > 1.5 * right[0]
    assert img[:8].max() > 4.0
    nrm = o.download(1) / 16
    assert np.abs(np.linalg.norm(nrm[14, 48]) - 1) < 1e-3  # back wall: constant first-hit normal
    alb = o.download(2) / 16
    np.testing.assert_allclose(alb[14, 48], [0.725, 0.71, 0.68], atol=1e-4)


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference/images/cornell-box.png")
def test_t2_against_renes_published_cornell(oracle_mod):
    """T2: oracle -> average -> to_rgb8 vs images/cornell-box.png (rene's own Vulkan
    render, 1024^2 @ 5000 spp), both box-filtered to suppress Monte-Carlo noise.  rene vs the
    unbiased Tungsten render is 0.043 sRGB RMSE, so 0.02 separates bug-compatible from 'correct'."""
    from PIL import Image
    n = 128
    spp = 192
    o = oracle_mod.Oracle(scenes.cornell_box(n, n))
    o.render(0, spp)
    mine = oracle_mod.to_rgb8(o.download(0), spp).astype(np.float32) / 255
    ref = np.asarray(Image.open(os.path.join(REFERENCE, "images", "cornell-box.png")).convert("RGB"), np.float32) / 255
    k = ref.shape[0] // n
    ref = ref.reshape(n, k, n, k, 3).mean(axis=(1, 3))
    box = lambda x: x.reshape(n // 8, 8, n // 8, 8, 3).mean(axis=(1, 3))
    rmse = float(np.sqrt(((box(mine) - box(ref)) ** 2).mean()))
    tung = np.asarray(Image.open(os.path.join(REFERENCE, "sample_scenes", "cornell-box", "TungstenRender.png")).convert("RGB"), np.float32) / 255
    tung = tung.reshape(n, k, n, k, 3).mean(axis=(1, 3))
    rmse_t = float(np.sqrt(((box(mine) - box(tung)) ** 2).mean()))
    print("T2 sRGB RMSE vs rene:", rmse, " vs Tungsten:", rmse_t)
    assert rmse < 0.0188  # measured 0.0150 (x 1.25)
    assert rmse < rmse_t  # closer to rene than to the unbiased answer


def test_t2_veach_mis_against_renes_published_render(oracle_mod):
    """T2 on the oracle itself for the Metal + sphere-emitter scene (veach-mis vs rene's PNG was checked on
    the GPU side only): oracle -> average -> to_rgb8 at a quarter of rene's resolution, 2x2 box-filtered, against the
    8x8 box-filtered copy of images/veach-mis.png (tests/golden/rene_veach_mis_box8.npy, made by
    make_rene_image_fixtures.py).  rene's image is far from the unbiased one (sRGB RMSE 0.174 vs Tungsten, SURVEY
    section 6), so agreement here pins the restated quirks (Q1, Q3-Q6, Q9, Q11) of the materials Cornell never renders."""
    spp = 1024
    o = oracle_mod.Oracle(scenes.veach_mis(320, 180))
    o.render(0, spp)
    mine = oracle_mod.to_rgb8(o.download(0), spp).astype(np.float32) / 255
    mine = mine.reshape(90, 2, 160, 2, 3).mean(axis=(1, 3))
    want = np.load(os.path.join(GOLDEN, "rene_veach_mis_box8.npy"))
    assert want.shape == mine.shape
    # compared in cells of 40 x 40 of rene's pixels: a quarter-resolution render places its edges (plates, lights) up to
    # two of rene's pixels away, which a finer grid would count as error (0.032 per 8 x 8 cell at any sample count)
    box = lambda x: x.reshape(18, 5, 32, 5, 3).mean(axis=(1, 3))
    rmse = float(np.sqrt(((box(mine) - box(want)) ** 2).mean()))
    ratio = float(mine.mean() / want.mean())
    print("oracle T2 veach-mis sRGB RMSE vs rene:", rmse, "mean ratio", ratio)
    assert rmse < 0.0195 and abs(ratio - 1) < 0.012  # measured 0.0156 (x 1.25) / 0.991; rene vs the unbiased Tungsten image: 0.174


def test_t2_cornell_against_renes_published_render_from_the_fixture(oracle_mod):
    """The Cornell T2 without the reference checkout: the committed 8x8 box-filtered copy of images/cornell-box.png."""
    spp = 192
    o = oracle_mod.Oracle(scenes.cornell_box(128, 128))
    o.render(0, spp)
    mine = oracle_mod.to_rgb8(o.download(0), spp).astype(np.float32) / 255
    want = np.load(os.path.join(GOLDEN, "rene_cornell_box8.npy"))  # 128 x 128 cells of 8 x 8 pixels
    assert want.shape == mine.shape
    box = lambda x: x.reshape(16, 8, 16, 8, 3).mean(axis=(1, 3))
    rmse = float(np.sqrt(((box(mine) - box(want)) ** 2).mean()))
    print("oracle T2 Cornell sRGB RMSE vs rene (fixture):", rmse)
    assert rmse < 0.0188  # measured 0.0150 (x 1.25)


def test_t2_cornell_energy_of_every_surface_against_renes_render(oracle_mod):
    """an 8 x 8 box RMSE lets a one-per-cent energy error in one wall through.  Here every surface the camera
    sees (tests/t2_regions.py: cut by the oracle's own first hits) is compared in mean linear radiance with rene's PNG, in the
    channels its 8 bits resolve.  4096 frames: rene's frame-wide generator (Q3) gives all pixels of a frame the same light /
    BSDF coin and the same point on the light, so a region's mean converges with the number of FRAMES, not of pixels.
    Measured (round 3): the oracle is 1.6 - 2.9 % brighter than rene's image on the walls, the floor, the ceiling and the tall
    block, 0.4 - 0.9 % darker on the right wall -- a systematic offset of the published image (whose code version is not
    known: the checkout has no history) that no depth cap reproduces (cap 8: +0.4 ... +1.7 %; cap 6: -3.5 ... +0.9 %).  The
    bound is 1.25 x the largest measured deviation."""
    import t2_regions as T
    spp = 4096
    reg = T.region_map(oracle_mod, scenes.cornell_box(1024, 1024), 8)
    _, lin4 = T.rene_box4("cornell")
    lin8 = T.box(lin4, 2)
    o = oracle_mod.Oracle(scenes.cornell_box(128, 128))
    o.render(0, spp)
    mine = T.to_linear(oracle_mod.to_rgb8(o.download(0), spp).astype(np.float32) / 255.0)
    checked, worst = 0, 0.0
    for rid in np.unique(reg):
        m = reg == rid
        if rid < 0 or m.sum() < 300:  # (smaller regions are all edge at this resolution; the GPU test has them at 4 x 4)
            continue
        a, b = mine[m].mean(axis=0), lin8[m].mean(axis=0)
        for ch in range(3):
            if 0.03 <= b[ch] < 0.9:
                checked += 1
                worst = max(worst, abs(float(a[ch] / b[ch]) - 1.0))
                assert abs(a[ch] / b[ch] - 1.0) < 0.037, (int(rid) >> 12, int(rid) & 4095, ch, float(a[ch] / b[ch]))
    print("oracle vs rene, region energies: largest deviation", worst, "over", checked, "region-channels")
    assert checked >= 10


# ---- tier T3: where things are in rene's published teapot render (tests/t3_geometry.py) --------------------------------------
def test_t3_teapot_geometry_against_renes_published_render(oracle_mod):
    """The oracle's first-hit albedo of ONE frame of rene's own sample_scenes/teapot/scene.pbrt (through the pbrt loader, at the
    file's 1280 x 720) against images/teapot.png reduced to one bit per pixel (tests/golden/rene_teapot_bright.npy): the
    checkerboard's squares fall where rene's do on 99.97 % of the floor, and the teapot's outline lies within the 2-6-pixel rings
    either side of rene's.  Reference-held data for camera.rs, the triangle hit shader's uv, texture.rs' checkerboard, the PLY
    loader and the traversal on a third scene -- geometry only: the image is denoised and its environment map is not in the checkout."""
    import t3_geometry as T
    o = oracle_mod.Oracle(scenes.teapot_full(1280, 720))
    o.render(0, 1)
    g = T.geometry(o.download(2), T.rene_teapot_bright())
    print("T3 teapot geometry, oracle vs rene:", g)
    assert g["floor_pixels"] > 600000 and g["teapot_pixels"] > 200000
    assert 1.0 - g["checker"] < 4.2e-4  # measured 3.3e-4 (x 1.25): 209 of 635 198 pixels, nearly all in the teapot's contact shadow
    assert g["inside"] > 0.9687 and g["outside"] < 0.0102  # measured 0.9750 / 0.0081 (errors x 1.25)
    # the same render mirrored top to bottom -- what a wrong launch_id.y convention would give -- is at chance
    assert T.geometry(o.download(2)[::-1], T.rene_teapot_bright())["checker"] < 0.6


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference/images/teapot.png")
def test_t3_fixture_is_renes_image():
    """tests/golden/rene_teapot_bright.npy is images/teapot.png thresholded (make_rene_image_fixtures.py), bit for bit."""
    import importlib.util
    import t3_geometry as T
    spec = importlib.util.spec_from_file_location("make_rene_image_fixtures", os.path.join(GOLDEN, "make_rene_image_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    np.testing.assert_array_equal(m.teapot_bright(os.path.join(REFERENCE, "images", "teapot.png")), T.rene_teapot_bright())


def test_t2_dragon_lit_surfaces_against_renes_published_render(oracle_mod):
    """T2 for the distant light (light.rs, lib.rs:234-272), which rene's Cornell and veach-mis images do not reach: rene's raw dragon
    render against the oracle on the 12 meshes of that scene the checkout holds (tests/t2_regions.py, dragon_lit_ratio).  The median
    ratio of linear radiance over 3 300 directly lit 4 x 4 cells is 0.993; pinned to +- 1 % of that (the four missing meshes move the
    quartiles -- their shadows and their bounce light are absent here -- not the median)."""
    import t2_regions as T
    spp = 32
    o = oracle_mod.Oracle(scenes.dragon_partial(1280, 720))
    o.render(0, spp)
    hit = np.abs(o.download(1)[..., :3]).sum(axis=2) > 0.5 * spp  # first-hit normal layer: a hit in every frame
    n, med, q1, q3, inside = T.dragon_lit_ratio(oracle_mod.to_rgb8(o.download(0), spp), hit)
    print(f"T2 dragon (partial), oracle vs rene: {n} lit cells, linear ratio median {med:.4f} quartiles {q1:.4f} / {q3:.4f}, inside rene's silhouette {inside:.4f}")
    assert n > 3000
    assert abs(med - 0.993) < 0.01 and q1 > 0.94 and q3 < 1.03
    assert inside > 0.95
