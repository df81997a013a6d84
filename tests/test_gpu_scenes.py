"""-m gpu: general-BSDF / sphere / texture / light paths of the HIP kernel against the oracle."""
import numpy as np
import pytest

from rene_amd import abi, api, scenes
from test_gpu_parity import aov_check, t1_check

pytestmark = pytest.mark.gpu


def _compare(scene, frames, oracle_mod, frac, relmse, ctol, flags=0):
    o = oracle_mod.Oracle(scene)
    o.render(0, frames)
    with api.Renderer(scene, flags=abi.FLAG_COUNTERS | flags) as r:
        r.render(0, frames)
        so, sg = o.stats().as_dict(), r.stats().as_dict()
        print({k: (sg[k], so[k]) for k in ("rays_closest", "rays_emitter", "rays_shadow", "hits", "adds")})
        assert sg["paths"] == so["paths"]
        for k in ("rays_closest", "rays_emitter", "rays_shadow", "hits", "adds"):
            assert abs(sg[k] - so[k]) <= ctol * so[k] + 4, (k, sg[k], so[k])
        g0, o0 = r.download(0), o.download(0)
        assert np.isfinite(g0).all() == np.isfinite(o0).all()
        fin = np.isfinite(g0).all(axis=2) & np.isfinite(o0).all(axis=2)
        t1_check(np.where(fin[..., None], g0, 0), np.where(fin[..., None], o0, 0), frac=frac, relmse=relmse)
        aov_check(r.download(1), o.download(1), atol=5e-5 * frames, frac=5e-3)
        aov_check(r.download(2), o.download(2), atol=5e-5 * frames, frac=5e-3)
        assert abs(float(g0[fin].sum() / o0[fin].sum()) - 1) < 1e-3
    return sg, so


def test_veach_mis_metal_and_sphere_emitters(oracle_mod):
    # BASELINE config 3 geometry at reduced size: Metal (TrowbridgeReitz + conductor Fresnel),
    # 3 emissive spheres (quirk Q4), the emitter-only structure holding spheres
    s = scenes.veach_mis(160, 90)
    # Tolerance: rene's cone pdf for sphere emitters, 1 / (2 pi (1 - sqrt(1 - r^2/d^2))) (lib.rs:1058-1064),
    # cancels catastrophically in fp32 for the r = 0.05 light (1 - cos ~ 1e-6 against an ulp of 6e-8),
    # so ANY two fp32 evaluations (FMA or not, this GPU or a Vulkan one) differ by 1-30 % on samples that
    # see that light.  tests/test_oracle_fp_sensitivity.py shows the same spread between two CPU builds
    # of the oracle.  Hence: image-level agreement tight (relMSE, mean), per-pixel agreement loose.
    for flags in (0, abi.FLAG_FORCE_BVH):
        sg, so = _compare(s, 16, oracle_mod, frac=0.12, relmse=1e-4, ctol=2e-3, flags=flags)
        assert sg["rays_emitter"] > 0 and sg["rays_shadow"] == 0


def test_material_zoo_every_kind(oracle_mod):
    # glass / mirror / metal / substrate / plastic / uber, checkerboard + imagemap + scale textures,
    # env-map background, a distant light, triangle + sphere emitters, a mirrored instance
    s = scenes.material_zoo(96, 64)
    for flags in (0, abi.FLAG_FORCE_BVH):
        sg, so = _compare(s, 32, oracle_mod, frac=1e-2, relmse=2e-3, ctol=3e-3, flags=flags)
        assert sg["rays_shadow"] > 0


def test_traversal_with_spheres(oracle_mod):
    s = scenes.material_zoo(96, 64)
    o = oracle_mod.Oracle(s)
    rng = np.random.default_rng(2)
    n = 30000
    org = np.stack([rng.uniform(-5, 5, n), rng.uniform(0.1, 3.5, n), rng.uniform(-7, 3, n)], 1).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    for flags in (0, abi.FLAG_FORCE_BVH):
      with api.Renderer(s, flags=flags) as r:
        for which in (0, 1):
            hg, ho = r.trace(org, d, which=which), o.trace(org, d, which=which)
            mg, mo = hg["t"] < 0, ho["t"] < 0
            tie = np.abs(hg["t"] - ho["t"]) <= 1e-5 * (1 + np.abs(ho["t"]))
            bad = (mg != mo) | (~mo & ((hg["instance"] != ho["instance"]) | (hg["primitive"] != ho["primitive"])) & ~tie)
            assert bad.sum() <= 3, bad.sum()
            ok = ~mg & ~mo & (hg["instance"] == ho["instance"])
            np.testing.assert_allclose(hg["t"][ok], ho["t"][ok], rtol=5e-5, atol=2e-6)


def test_cornell_with_distant_light_and_no_emitter(oracle_mod):
    # the dragon-class branch: lights_len = 1, emit_object_len = 0 -> plain BSDF sampling (lib.rs:325-337)
    s = scenes.cornell_box(96, 96)
    s.instances[-1].area_light_index = 0  # switch the quad emitter off
    s.add_light_distant((-0.18862, 0.692312, 0.69651), (0, 0, 0), (8, 8, 8))  # dragon/scene.pbrt:44
    sg, so = _compare(s, 8, oracle_mod, frac=1e-3, relmse=1e-4, ctol=1e-4)
    assert sg["rays_emitter"] == 0 and sg["rays_shadow"] == sg["hits"]


@pytest.mark.parametrize("n_lat,n_lon,res", [(48, 52, (160, 90)), (640, 680, (240, 136))])
def test_dragon_class_bvh_path(oracle_mod, n_lat, n_lon, res):
    """BASELINE config 4 stand-in: a displaced sphere of 2*n_lat*n_lon triangles (870 400 at
    full size) in the Cornell room, Matte, one distant light, no emitter -> deep BVH traversal, shadow
    rays with any-hit early-out, plain BSDF sampling."""
    s = scenes.dragon_class(res[0], res[1], n_lat, n_lon)
    info = api.pack_info(s)
    assert info.n_triangles == 2 * n_lat * n_lon + 20 and not (info.features & 64)  # too big for the item loop
    sg, so = _compare(s, 4, oracle_mod, frac=2e-3, relmse=1e-4, ctol=5e-4)
    assert sg["rays_emitter"] == 0 and sg["rays_shadow"] == sg["hits"]
    # traversal alone, against the oracle's two-level BVH
    o = oracle_mod.Oracle(s)
    rng = np.random.default_rng(4)
    n = 20000
    org = np.tile(np.array([[0, 1, 6.8]], np.float32), (n, 1))
    d = np.stack([rng.uniform(-.25, .25, n), rng.uniform(-.17, .17, n), -np.ones(n)], 1)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    with api.Renderer(s) as r:
        hg, ho = r.trace(org, d), o.trace(org, d)
    tie = np.abs(hg["t"] - ho["t"]) <= 2e-5 * (1 + np.abs(ho["t"]))
    bad = ((hg["t"] < 0) != (ho["t"] < 0)) | ((ho["t"] >= 0) & (hg["primitive"] != ho["primitive"]) & ~tie)
    assert bad.sum() <= 5, bad.sum()


def test_teapot_class_substrate_checkerboard_envmap(oracle_mod):
    """BASELINE config 5 stand-in (rene_amd.scenes.teapot_class): Substrate with alpha = 0.001, a Matte
    floor with a 20 x 20 checkerboard, an environment map sampled by escaping rays only (no emitter, no
    distant light): the general single-lobe traversal-restart kernel with textures and a background."""
    s = scenes.teapot_class(160, 90, n_lat=40, n_lon=42)
    info = api.pack_info(s)
    assert info.n_triangles == 2 * 40 * 42 + 2 and (info.features & 0xff) == (2 | 4 | 16) and (info.features >> 8) == (1 | 4 | 8)  # general, textures, background; no specular, no microfacet lobes, no emit objects
    sg, so = _compare(s, 16, oracle_mod, frac=5e-3, relmse=1e-3, ctol=2e-3)
    assert sg["rays_emitter"] == 0 and sg["rays_shadow"] == 0
    _compare(s, 16, oracle_mod, frac=5e-3, relmse=1e-3, ctol=2e-3, flags=abi.FLAG_NO_RESTART)


@pytest.mark.parametrize("name", ["dragon", "zoo", "teapot", "cornell"])
def test_wavefront_equals_megakernels_bit_for_bit(name):
    """The three schedulings of the BVH integrator -- stage-separated wavefront (RENE_FLAG_WAVEFRONT, wavefront.inc),
    traversal-restart megakernel (default) and while-while megakernel (RENE_FLAG_NO_RESTART) --
    run the same arithmetic in the same order: identical images (all three layers) and identical counters,
    also across launch splits and tile shards."""
    s = {"dragon": lambda: scenes.dragon_class(96, 54, 24, 26), "zoo": lambda: scenes.material_zoo(64, 48),
         "teapot": lambda: scenes.teapot_class(96, 54, 20, 22), "cornell": lambda: scenes.cornell_box(48, 48)}[name]()
    force = abi.FLAG_FORCE_BVH
    imgs, stats = [], []
    for flags in (abi.FLAG_WAVEFRONT, 0, abi.FLAG_NO_RESTART):
        with api.Renderer(s, flags=force | abi.FLAG_COUNTERS | flags) as r:
            r.render(0, 7)
            imgs.append([r.download(k) for k in range(3)])
            st = r.stats().as_dict()
            stats.append({k: st[k] for k in ("rays_closest", "rays_shadow", "rays_emitter", "paths", "hits", "adds", "node_visits", "prim_tests")})
    assert stats[0] == stats[2], stats
    # the traversal-restart kernel traverses speculatively (render_wf.inc, RENE_WF_POSTPONE): a lane puts its leaf aside and goes on
    # with inner nodes until the next leaf step -- same leaves in the same order with the same outcome, a few more node visits
    # and triangle tests (those made before the pending leaf had shortened the ray)
    exact = ("rays_closest", "rays_shadow", "rays_emitter", "paths", "hits", "adds")
    assert {k: stats[1][k] for k in exact} == {k: stats[0][k] for k in exact}, stats
    for k in ("node_visits", "prim_tests"):
        assert stats[0][k] <= stats[1][k] <= 1.2 * stats[0][k], (k, stats)
    for k in range(3):
        np.testing.assert_array_equal(imgs[0][k], imgs[2][k])  # wavefront == while-while megakernel, always
        if name in ("dragon", "cornell"):
            np.testing.assert_array_equal(imgs[0][k], imgs[1][k])
        else:
            # the traversal-restart megakernel re-derives the surface and BSDF after each shadow query; in the
            # general-BSDF instantiations that costs an occasional last-bit difference (seen: 1 value in 9216)
            np.testing.assert_allclose(imgs[0][k], imgs[1][k], rtol=1e-6, atol=1e-7)
            assert (imgs[0][k] != imgs[1][k]).mean() < 1e-3
    with api.Renderer(s, flags=force | abi.FLAG_WAVEFRONT) as r:  # launch split
        r.render(0, 3)
        r.render(3, 4)
        for k in range(3):
            np.testing.assert_array_equal(r.download(k), imgs[0][k])
    acc = np.zeros_like(imgs[0][0])
    for rank in range(3):  # tile shards
        with api.Renderer(s, flags=force | abi.FLAG_NO_AOV | abi.FLAG_WAVEFRONT, shard_mode=abi.SHARD_TILES, shard_rank=rank, shard_count=3) as r:
            r.render(0, 7)
            acc += r.download(0)
    np.testing.assert_array_equal(acc, imgs[0][0])


@pytest.mark.parametrize("name", ["cornell", "cornell-bvh", "veach", "dragon", "teapot", "fog", "zoo", "zoo-bvh", "media-zoo"])
def test_the_overlap_flag_is_accepted_and_changes_nothing(name):
    """RENE_FLAG_OVERLAP (ABI <= 3: consecutive launches on two streams) is accepted and ignored since ABI v4: launches are
    serial, a pixel's running sums pass from launch to launch through the version its records carry, and every bit of the
    three layers and the counters is the same with and without the flag -- on every kernel family, over launches of mixed
    lengths, with a second batch after a sync."""
    s, flags = {"cornell": (lambda: scenes.cornell_box(256, 256), 0),
                "cornell-bvh": (lambda: scenes.cornell_box(128, 128), abi.FLAG_FORCE_BVH | abi.FLAG_NO_RESTART),
                "veach": (lambda: scenes.veach_mis(192, 128), 0),
                "dragon": (lambda: scenes.dragon_class(192, 108, 40, 44), 0),
                "teapot": (lambda: scenes.teapot_class(192, 108, 20, 22), 0),
                "fog": (lambda: scenes.cornell_fog(128, 128), 0),
                "zoo": (lambda: scenes.material_zoo(128, 96), 0),  # multi-lobe kernels: two waves per SIMD, private scratch
                "zoo-bvh": (lambda: scenes.material_zoo(128, 96), abi.FLAG_FORCE_BVH),
                "media-zoo": (lambda: scenes.media_zoo(96, 64), 0)}[name]
    s = s()
    launches = [(0, 16), (16, 16), (32, 3), (35, 1), (36, 8), (44, 20)]  # two-level and single-level launches mixed
    out = []
    for extra in (0, abi.FLAG_OVERLAP):
        with api.Renderer(s, flags=flags | abi.FLAG_COUNTERS | extra) as r:
            for f0, nf in launches:
                r.render(f0, nf)
            imgs = [r.download(k) for k in range(3)]
            out.append((imgs, r.stats().as_dict()))
            # a second batch after the join: the flags of the first batch are what its first items wait for
            r.render(64, 5)
            r.render(69, 5)
            out[-1] = (imgs + [r.download(0)], out[-1][1])
    (a, sa), (b, sb) = out
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    for k in ("rays_closest", "rays_shadow", "rays_emitter", "paths", "hits", "adds", "launches", "frames"):
        assert sa[k] == sb[k], k
    assert np.isfinite(a[0]).all() and a[0].sum() > 0


@pytest.mark.parametrize("name", ["cornell", "dragon", "fog"])
def test_every_work_item_cut_and_rene_tune_give_the_same_bits(name, monkeypatch):
    """A launch cuts each pixel's frames into n work items that hand the running sums on (item_publish /
    item_ready); the order of additions does not depend on n, so neither does any bit of the three layers.
    rene_tune measures which n is fastest, resets the context and leaves nothing else behind."""
    s = {"cornell": lambda: scenes.cornell_box(96, 96), "dragon": lambda: scenes.dragon_class(96, 54, 24, 26),
         "fog": lambda: scenes.cornell_fog(64, 64)}[name]()
    with api.Renderer(s, flags=abi.FLAG_SINGLE_LEVEL | abi.FLAG_COUNTERS) as r:
        r.render(0, 13)
        r.render(13, 7)
        want = [r.download(k) for k in range(3)]
        st = r.stats().as_dict()
    for levels in ("1", "2", "3", "5", "13", "31"):  # 13 frames: uneven cuts, one frame per item, more levels than frames
        monkeypatch.setenv("RENE_LEVELS", levels)
        with api.Renderer(s, flags=abi.FLAG_COUNTERS) as r:
            r.render(0, 13)
            r.render(13, 7)
            for k in range(3):
                np.testing.assert_array_equal(r.download(k), want[k], err_msg=f"levels {levels} layer {k}")
            got = r.stats().as_dict()
            assert {k: got[k] for k in ("paths", "rays_closest", "rays_shadow", "rays_emitter", "adds")} == \
                   {k: st[k] for k in ("paths", "rays_closest", "rays_shadow", "rays_emitter", "adds")}
    monkeypatch.delenv("RENE_LEVELS")
    # the default cut: uniform items whose last ones halve (13 frames, items of 4 down to 1: 4 4 | 3 1 1; items of 13 down to 2: 7 3 3)
    for item, tail in (("4", "1"), ("13", "2"), ("5", "5"), ("3", "2"), ("64", "8")):
        monkeypatch.setenv("RENE_ITEM_FRAMES", item)
        monkeypatch.setenv("RENE_ITEM_TAIL", tail)
        with api.Renderer(s, flags=abi.FLAG_COUNTERS) as r:
            r.render(0, 13)
            r.render(13, 7)
            for k in range(3):
                np.testing.assert_array_equal(r.download(k), want[k], err_msg=f"items of {item} frames halving to {tail}, layer {k}")
            got = r.stats().as_dict()
            assert {k: got[k] for k in ("paths", "rays_closest", "rays_shadow", "rays_emitter", "adds")} == \
                   {k: st[k] for k in ("paths", "rays_closest", "rays_shadow", "rays_emitter", "adds")}
    monkeypatch.delenv("RENE_ITEM_FRAMES")
    monkeypatch.delenv("RENE_ITEM_TAIL")
    for flags in (0, abi.FLAG_OVERLAP):
        with api.Renderer(s, flags=flags) as r:
            r.render(0, 3)  # something to be discarded
            r.tune(13)
            z = r.stats().as_dict()
            assert z["frames"] == 0 and z["launches"] == 0 and not r.download(0).any()
            r.render(0, 13)
            r.render(13, 7)
            for k in range(3):
                np.testing.assert_array_equal(r.download(k), want[k])


# ---- the BASELINE configurations at their own sizes (configs_untested) ---------------------------------
def test_teapot_full_126k_triangles_against_the_oracle(oracle_mod):
    """BASELINE config 5 with the reference's own inputs (tests/golden/teapot: rene's scene.pbrt + its two PLY meshes,
    126 050 triangles, through the pbrt loader) at a size the oracle finishes in seconds."""
    s = scenes.teapot_full(192, 108)
    info = api.pack_info(s)
    assert info.n_triangles == 126050 and not (info.features & 64)
    sg, so = _compare(s, 8, oracle_mod, frac=5e-3, relmse=1e-3, ctol=2e-3)
    assert sg["rays_emitter"] == 0 and sg["rays_shadow"] == 0 and sg["hits"] > 0
    o = oracle_mod.Oracle(s)
    rng = np.random.default_rng(9)
    rays = [o.camera_ray(float(u), float(v)) for u, v in rng.uniform(0.02, 0.98, size=(6000, 2))]
    org, d = np.stack([r[0] for r in rays]).astype(np.float32), np.stack([r[1] for r in rays]).astype(np.float32)
    with api.Renderer(s) as r:
        hg, ho = r.trace(org, d), o.trace(org, d)
    tie = np.abs(hg["t"] - ho["t"]) <= 2e-5 * (1 + np.abs(ho["t"]))
    bad = ((hg["t"] < 0) != (ho["t"] < 0)) | ((ho["t"] >= 0) & ((hg["primitive"] != ho["primitive"]) | (hg["instance"] != ho["instance"])) & ~tie)
    assert (ho["t"] > 0).sum() > 3000 and bad.sum() <= 3, bad.sum()


def test_dragon_partial_real_meshes_against_the_oracle(oracle_mod):
    """rene's own dragon scene with the 12 meshes its checkout holds (tests/golden/dragon_partial, 51 140 triangles of artist
    geometry: long thin triangles, sizes over four orders of magnitude), through the pbrt loader: image, counters and the
    traversal itself (device BVH4 vs the oracle's two-level BVH2, first hits through the film) -- VERDICT r2 item 8."""
    s = scenes.dragon_partial(192, 108)
    info = api.pack_info(s)
    assert info.n_triangles == 51140 and info.lights_len == 1 and not (info.features & 64)
    sg, so = _compare(s, 8, oracle_mod, frac=5e-3, relmse=1e-3, ctol=2e-3)
    assert sg["rays_shadow"] > 0 and sg["rays_emitter"] == 0
    o = oracle_mod.Oracle(s)
    rng = np.random.default_rng(11)  # (a tenth of the film sees geometry: the ground and the dragon's body are among the missing meshes)
    rays = [o.camera_ray(float(u), float(v)) for u, v in rng.uniform(0.02, 0.98, size=(12000, 2))]
    org, d = np.stack([r[0] for r in rays]).astype(np.float32), np.stack([r[1] for r in rays]).astype(np.float32)
    with api.Renderer(s) as r:
        hg, ho = r.trace(org, d), o.trace(org, d)
    tie = np.abs(hg["t"] - ho["t"]) <= 2e-5 * (1 + np.abs(ho["t"]))
    bad = ((hg["t"] < 0) != (ho["t"] < 0)) | ((ho["t"] >= 0) & ((hg["primitive"] != ho["primitive"]) | (hg["instance"] != ho["instance"])) & ~tie)
    assert (ho["t"] > 0).sum() > 800 and bad.sum() <= 3, bad.sum()


@pytest.mark.parametrize("name", ["dragon-class", "teapot-full"])
def test_full_size_configs_hold_their_invariants(name):
    """C4 / C5 at 1920x1080 (the oracle would take minutes here): what does not depend on size -- the image is finite and
    non-empty, a job cut into launches (with and without the ignored overlap flag) and into tile shards is bit-identical, paths = pixels x
    frames, and the per-ray statistics equal those of the size the oracle checks."""
    big = scenes.dragon_class(1920, 1080) if name == "dragon-class" else scenes.teapot_full(1920, 1080)
    small = scenes.dragon_class(240, 136) if name == "dragon-class" else scenes.teapot_full(192, 108)
    frames = 6
    with api.Renderer(big) as r:
        r.render(0, frames)
        whole = [r.download(l) for l in range(3)]
        st = r.stats().as_dict()
    assert st["paths"] == 1920 * 1080 * frames and all(np.isfinite(w).all() for w in whole) and whole[0].mean() > 0
    with api.Renderer(big, flags=abi.FLAG_OVERLAP) as r:
        r.render(0, 2); r.render(2, 3); r.render(5, 1)
        for l in range(3):
            assert np.array_equal(r.download(l), whole[l])
    acc = [np.zeros_like(w) for w in whole]
    for rank in range(2):
        with api.Renderer(big, shard_mode=abi.SHARD_TILES, shard_rank=rank, shard_count=2) as r:
            r.render(0, frames)
            for l in range(3):
                acc[l] += r.download(l)
    for l in range(3):
        assert np.array_equal(acc[l], whole[l])
    with api.Renderer(small) as r:
        r.render(0, frames)
        ss = r.stats().as_dict()
    for k in ("rays_closest", "rays_shadow", "hits", "adds"):  # per path, within a few per cent of the small render
        a, b = st[k] / st["paths"], ss[k] / ss["paths"]
        assert abs(a - b) <= 0.06 * max(b, 1e-9) + 1e-3, (k, a, b)


def test_wavefront_fp16_ray_payload(oracle_mod):
    """BASELINE config 5's "fp16 ray payload" (RENE_FLAG_FP16_PAYLOAD with RENE_FLAG_WAVEFRONT): a slot's ray direction and
    throughput travel as halves.  Same counters up to path forks, and an image within the T1 tolerance of the fp32
    payload's *in the mean* (relMSE <= 1e-3; a direction rounded to 11 bits moves a hit point by up to 1e-3 of its distance, so
    single pixels at edges may differ); the oracle is the third party."""
    s = scenes.teapot_class(160, 90, n_lat=40, n_lon=42)
    frames = 16
    imgs, stats = {}, {}
    for name, flags in (("fp32", abi.FLAG_WAVEFRONT), ("fp16", abi.FLAG_WAVEFRONT | abi.FLAG_FP16_PAYLOAD)):
        with api.Renderer(s, flags=flags | abi.FLAG_COUNTERS) as r:
            r.render(0, frames)
            imgs[name] = r.download(0)
            stats[name] = r.stats().as_dict()
    assert not np.array_equal(imgs["fp16"], imgs["fp32"])  # the flag does something
    for k in ("rays_closest", "hits", "adds"):
        assert abs(stats["fp16"][k] - stats["fp32"][k]) <= 5e-3 * stats["fp32"][k], (k, stats["fp16"][k], stats["fp32"][k])
    a, b = imgs["fp16"], imgs["fp32"]
    assert np.isfinite(a).all()
    assert float(((a - b) ** 2).sum() / (b ** 2).sum()) <= 1e-3 and abs(float(a.sum() / b.sum()) - 1) < 2e-3
    o = oracle_mod.Oracle(s)
    o.render(0, frames)
    ref = o.download(0)
    assert float(((a - ref) ** 2).sum() / (ref ** 2).sum()) <= 2e-3
    # the default integrators keep their payload in registers: the flag changes nothing there
    with api.Renderer(s) as r0, api.Renderer(s, flags=abi.FLAG_FP16_PAYLOAD) as r1:
        r0.render(0, 4); r1.render(0, 4)
        assert np.array_equal(r0.download(0), r1.download(0))


def _zoo_with_a_big_mesh():
    """material_zoo plus a 7 680-triangle displaced sphere of Metal: every material kind, deep enough a tree for the
    traversal-restart kernels (multi-lobe instantiation)."""
    s = scenes.material_zoo(128, 96)
    s.add_triangle_mesh(scenes.displaced_sphere(60, 64), s.add_metal(scenes._VEACH_ETA, scenes._VEACH_K, 0.1, 0.1, remap_roughness=True),
                        ctm=scenes.glam.from_translation((0.0, 1.6, 1.0)))
    return s


def _dragon_with_a_metal_ball():
    """dragon-class (Matte, one distant light) plus a Metal sphere: the general single-lobe restart kernel."""
    s = scenes.dragon_class(160, 90, 40, 44)
    s.add_sphere(0.25, s.add_metal(scenes._VEACH_ETA, scenes._VEACH_K, 0.05, 0.05, remap_roughness=False),
                 ctm=scenes.glam.from_translation((-0.4, 0.25, 0.3)))
    return s


@pytest.mark.parametrize("name", ["dragon", "dragon+metal", "zoo+mesh"])
def test_instance_and_light_tables_in_lds_are_bit_identical(monkeypatch, name):
    """The traversal-restart kernels come in two instantiations: the instance records and the distant lights read from
    global memory, or from a copy in LDS behind the traversal stack (picked when it fits; RENE_NO_LDS_TABLES, read at every
    launch, keeps the first).  Same records, so the same image bit for bit."""
    scene = {"dragon": lambda: scenes.dragon_class(160, 90, 40, 44), "dragon+metal": _dragon_with_a_metal_ball,
             "zoo+mesh": _zoo_with_a_big_mesh}[name]()
    assert not (api.pack_info(scene).features & abi.FEAT_SMALL) and api.pack_info(scene).n_nodes_main > 512
    images = []
    for knob in (None, "1"):
        if knob is None:
            monkeypatch.delenv("RENE_NO_LDS_TABLES", raising=False)
        else:
            monkeypatch.setenv("RENE_NO_LDS_TABLES", knob)
        with api.Renderer(scene) as r:
            r.render(0, 5)
            r.render(5, 3)
            images.append([r.download(l) for l in range(3)])
    for a, b in zip(*images):
        assert np.array_equal(a, b)


def test_soak_many_launches_none_replayed():
    """rounds 1-2 overlapped consecutive launches on two streams and now and then a launch stalled for
    seconds behind the next one's waiters (docs/history.md section 4g); launches are serial now and no launch waits for another.
    Thirty jobs of sixteen launches each at full size on rene's teapot scene: every launch issued is a launch counted --
    none dropped a work item and had to be launched again -- and every job's image equals the first one's, bit for bit."""
    s = scenes.teapot_full(1920, 1080)
    jobs, per_job, frames = 30, 16, 24
    first = None
    with api.Renderer(s) as r:
        for j in range(jobs):
            r.reset()
            for k in range(per_job):
                r.render(k * frames, frames)
            r.sync()
            st = r.stats().as_dict()
            assert st["launches"] == per_job, (j, st["launches"])  # a replayed launch would count again
            if j in (0, jobs // 2, jobs - 1):
                img = r.download(0)
                if first is None:
                    first = img
                else:
                    assert np.array_equal(img, first), j
    assert np.isfinite(first).all() and first.sum() > 0


@pytest.mark.parametrize("name", ["dragon", "teapot", "fog", "cornell", "veach"])
def test_frame_chains_do_not_depend_on_the_cut_and_sum_to_the_frames(name):
    """Frame chains (device_scene.h, CHAINS; include/rene_hip.h, RENE_FLAG_FRAME_GROUPS): a pixel's frames are eight chains -- frame f in chain
    f % 8, each summed in frame order into an image of its own across calls -- and the image handed out is the chains added in chain order.
    The rule is on the frame's NUMBER: every cut of a job into calls gives the same bits (calls of odd lengths, single frames, a sync or a
    download in between), on every kernel family; and the image is the sum of its frames: equal to the float64 sum of the frames rendered
    one at a time up to fp32 rounding."""
    s = {"dragon": lambda: scenes.dragon_class(160, 90, 40, 44), "teapot": lambda: scenes.teapot_class(128, 72, 40, 44),
         "fog": lambda: scenes.dragon_fog(128, 72, 40, 44),  # (fog: Integrator "volpath", the restart kernel of kernels_vol.hip)
         "cornell": lambda: scenes.cornell_box(96, 64), "veach": lambda: scenes.veach_mis(96, 54)}[name]()
    keys = ("rays_closest", "rays_shadow", "rays_emitter", "paths", "hits", "adds")
    F = 35
    with api.Renderer(s) as r:
        r.render(0, F)
        ref = [r.download(k) for k in range(3)]
        sr = r.stats().as_dict()
        per_frame = np.zeros(ref[0].shape, np.float64)
        for f in range(F):
            r.reset()
            r.render(f, 1)
            per_frame += r.download(0)
    np.testing.assert_allclose(ref[0], per_frame, rtol=3e-6, atol=1e-6 * F)
    for cut in ([24, 11], [5, 3, 17, 7, 3], [1] * F, [8, 8, 8, 8, 3], [9, 26]):
        with api.Renderer(s) as g:
            f0 = 0
            for i, n in enumerate(cut):
                g.render(f0, n)
                f0 += n
                if i == 0:
                    g.sync()
                if i == 1:
                    g.download(1)  # (handing the image out adds the chains into the output and leaves them as they are)
            for k in range(3):
                np.testing.assert_array_equal(g.download(k), ref[k], err_msg=f"cut {cut} layer {k}")
            sg = g.stats().as_dict()
        assert {k: sg[k] for k in keys} == {k: sr[k] for k in keys}, cut
    with api.Renderer(s, flags=abi.FLAG_FRAME_GROUPS) as b:  # accepted and ignored
        b.render(0, F)
        np.testing.assert_array_equal(b.download(0), ref[0])
