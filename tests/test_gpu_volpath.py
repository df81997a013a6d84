"""-m gpu: the volumetric integrator (Integrator "volpath", lib.rs:359-803 + medium.rs) on the HIP path,
through the C ABI, against the oracle."""
import numpy as np
import pytest

from rene_amd import abi, api, scenes
from test_gpu_parity import aov_check, t1_check
from test_oracle_volpath import _probe_inputs, _slab_scene

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
        assert np.isfinite(g0).all() and np.isfinite(o0).all()
        t1_check(g0, o0, frac=frac, relmse=relmse)
        aov_check(r.download(1), o.download(1), atol=5e-5 * frames, frac=5e-3)
        aov_check(r.download(2), o.download(2), atol=5e-5 * frames, frac=5e-3)
        assert abs(float(g0.sum() / o0.sum()) - 1) < 2e-3
    return sg, so


def test_medium_functions_match_the_oracle(oracle_mod):
    # per-function parity of tr / phase / sample / sample_p (rene_medium_eval): the RNG consumption is
    # exact (T0: the stream's next u32 after sample + sample_p is identical), the floats agree to a few
    # ulp of v_exp_f32 / v_log_f32 / v_sin_f32
    s = scenes.media_zoo(16, 16)
    o = oracle_mod.Oracle(s)
    n = 50000
    rd, t_max, wo, wi, seeds = _probe_inputs(n, seed=23)
    with api.Renderer(s) as r:
        for idx in (0, 1, 2, 3):
            g, c = r.medium_eval(idx, rd, t_max, wo, wi, seeds), o.medium_eval(idx, rd, t_max, wo, wi, seeds)
            assert (g[:, 14].view(np.uint32) == c[:, 14].view(np.uint32)).all()
            np.testing.assert_allclose(g[:, 0:4], c[:, 0:4], rtol=2e-5, atol=1e-30)
            flip = g[:, 4] != c[:, 4]  # t == t_max to the last bit
            assert flip.sum() <= 2
            np.testing.assert_allclose(g[~flip, 5:8], c[~flip, 5:8], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(g[~flip, 8:11], c[~flip, 8:11], rtol=5e-5, atol=1e-30)
            np.testing.assert_allclose(g[:, 11:14], c[:, 11:14], rtol=0, atol=5e-5)  # sqrt(1 - cos^2) near the poles
        with pytest.raises(api.ReneError) as e:
            r.medium_eval(9, rd[:4], t_max[:4], wo[:4], wi[:4], seeds[:4])
        assert e.value.code == -1
    with api.Renderer(scenes.cornell_box(16, 16)) as r:  # path integrator: no media on the device
        with pytest.raises(api.ReneError):
            r.medium_eval(0, rd[:4], t_max[:4], wo[:4], wi[:4], seeds[:4])


def test_cornell_fog_item_loop_and_bvh(oracle_mod):
    s = scenes.cornell_fog(96, 96)
    assert api.pack_info(s).features == 64 | 128  # Matte-only, FEAT_SMALL | FEAT_VOLPATH
    for flags in (0, abi.FLAG_FORCE_BVH):
        sg, so = _compare(s, 16, oracle_mod, frac=2e-3, relmse=1e-4, ctol=1e-3, flags=flags)
        assert sg["rays_shadow"] > 0 and sg["rays_emitter"] > 0


def test_media_zoo_every_branch(oracle_mod):
    # glass around a medium, a None boundary with nested sphere, anisotropic phase functions, distant light
    # through tr(), sphere + triangle emitters through tr_emit(), textures, the infinite light
    s = scenes.media_zoo(96, 64)
    for flags in (0, abi.FLAG_FORCE_BVH):
        _compare(s, 32, oracle_mod, frac=1e-2, relmse=2e-3, ctol=3e-3, flags=flags)


def test_vacuum_volpath_equals_oracle_and_short_paths_equal_path(oracle_mod):
    s = scenes.cornell_box(64, 64)
    s.integrator = abi.INTEGRATOR_VOLPATH
    _compare(s, 16, oracle_mod, frac=1e-3, relmse=1e-4, ctol=1e-4)


def test_beer_lambert_on_the_device():
    sigma = np.array([0.3, 0.7, 1.3])
    L = np.array([2.0, 3.0, 4.0])
    with api.Renderer(_slab_scene(sigma, (0, 0, 0), depth=1.0, L=L, res=16)) as r:
        r.render(0, 4096)
        img = r.download(0) / 4096
    np.testing.assert_allclose(img[6:10, 6:10].reshape(-1, 3).mean(axis=0), L * np.exp(-sigma), rtol=0.02)


def test_launch_splits_shards_and_aov_flag_are_bit_identical():
    s = scenes.cornell_fog(64, 64)
    with api.Renderer(s) as r:
        r.render(0, 12)
        a = [r.download(k) for k in range(3)]
    with api.Renderer(s) as r:
        r.render(0, 5)
        r.render(5, 7)
        for k in range(3):
            np.testing.assert_array_equal(r.download(k), a[k])
    with api.Renderer(s, flags=abi.FLAG_NO_AOV | abi.FLAG_SINGLE_LEVEL) as r:
        r.render(0, 12)
        np.testing.assert_array_equal(r.download(0), a[0])
    acc = np.zeros_like(a[0])
    for rank in range(2):
        with api.Renderer(s, shard_mode=abi.SHARD_TILES, shard_rank=rank, shard_count=2) as r:
            r.render(0, 12)
            acc += r.download(0)
    np.testing.assert_array_equal(acc, a[0])


def test_full_size_fog_is_consistent_with_the_oracle_checked_size():
    # BASELINE-size volpath frame, checked through a size-independent property.  A pixel's estimator depends
    # on the resolution only through the footprint of its jitter: every 16th pixel of the 1024^2 image and
    # the pixels of a 64^2 render of the same frames (same frame-wide streams, lib.rs:514) are identically
    # distributed up to that footprint.  The estimator is heavy-tailed (1 / pdf at scattering vertices), so
    # compare quantiles of the per-pixel luminance, not means.  64^2 is the size class the oracle
    # comparisons above run at.
    frames = 16
    s = scenes.cornell_fog(1024, 1024)
    with api.Renderer(s, flags=abi.FLAG_COUNTERS) as r:
        r.render(0, frames)
        img = r.download(0) / frames
        st = r.stats().as_dict()
    assert np.isfinite(img).all()
    assert st["paths"] == frames * 1024 * 1024 and st["rays_closest"] > 3 * st["paths"]
    with api.Renderer(scenes.cornell_fog(64, 64)) as r:
        r.render(0, frames)
        ref = r.download(0) / frames
    lum = lambda a: (a @ np.array([0.2126, 0.7152, 0.0722], np.float32)).reshape(-1)
    q = [25, 50, 75, 90]  # 4096 pixels each: the upper quantiles are stable to ~2 %, the lower to ~8 %
    np.testing.assert_allclose(np.percentile(lum(img[8::16, 8::16]), q), np.percentile(lum(ref), q), rtol=0.1)
    np.testing.assert_allclose(np.percentile(lum(img[8::16, 8::16]), q[1:]), np.percentile(lum(ref), q[1:]), rtol=0.05)


# ---- volpath on deep trees: the traversal-restart scheduling (render_wf.inc, FEAT_VOLPATH) ----------------
@pytest.mark.parametrize("emitter", [True, False])
def test_volpath_restart_kernel_equals_the_while_while_kernel_bit_for_bit(emitter):
    """A tree of more than 512 nodes renders through render_kernel_wf with the walks (tr / tr_emit) as phases of its state machine;
    RENE_FLAG_NO_RESTART keeps the while-while loop.  Same arithmetic, same draws, same order of additions: every bit of the
    three layers and every ray / hit / add counter -- with and without an emitter (the emitter sample of a scattering vertex, tr_emit, the
    surface's one-sample mixture), across launch splits."""
    s = scenes.dragon_fog(128, 72, 40, 44, emitter=emitter)
    assert api.pack_info(s).n_nodes_main > 512
    out = []
    for flags in (0, abi.FLAG_NO_RESTART):
        with api.Renderer(s, flags=flags | abi.FLAG_COUNTERS) as r:
            r.render(0, 5)
            r.render(5, 3)
            out.append(([r.download(k) for k in range(3)], r.stats().as_dict()))
    (a, sa), (b, sb) = out
    for k in range(3):
        np.testing.assert_array_equal(a[k], b[k], err_msg=f"layer {k}")
    for k in ("rays_closest", "rays_shadow", "rays_emitter", "paths", "hits", "adds"):
        assert sa[k] == sb[k], k
    for k in ("node_visits", "prim_tests"):  # the restart kernel's speculative traversal (render_wf.inc, RENE_WF_POSTPONE) visits a few more
        assert sb[k] <= sa[k] <= 1.2 * sb[k], k
    assert np.isfinite(a[0]).all() and a[0].sum() > 0 and sa["rays_shadow"] > 0 and (sa["rays_emitter"] > 0) == emitter


def test_volpath_restart_kernel_against_the_oracle(oracle_mod):
    s = scenes.dragon_fog(128, 72, 40, 44)
    sg, so = _compare(s, 8, oracle_mod, frac=5e-3, relmse=1e-3, ctol=2e-3)
    assert sg["rays_emitter"] > 0 and sg["rays_shadow"] > sg["rays_closest"] * 0.5
    # the production instantiation (no counters, first-hit layers on) against the counting one
    with api.Renderer(s) as a, api.Renderer(s, flags=abi.FLAG_COUNTERS) as b:
        a.render(0, 6)
        b.render(0, 6)
        for k in range(3):
            np.testing.assert_array_equal(a.download(k), b.download(k))


def test_volpath_full_size_dragon_fog_holds_its_invariants():
    """1920 x 1080, 870 k triangles, fog: finite and non-empty, a job cut into launches is bit-identical, paths = pixels x frames."""
    s = scenes.dragon_fog(1920, 1080)
    with api.Renderer(s) as r:
        r.render(0, 4)
        whole = r.download(0)
        st = r.stats().as_dict()
        r.reset()
        r.render(0, 1)
        r.render(1, 3)
        assert np.array_equal(r.download(0), whole)
    assert st["paths"] == 1920 * 1080 * 4 and np.isfinite(whole).all() and whole.mean() > 0
