"""-m gpu: edge cases of the render path against the oracle -- empty and tiny inputs, degenerate geometry,
the None material under the path integrator, rays that graze or run parallel to the axes."""
import numpy as np
import pytest

from rene_amd import abi, api, glam, scenes
from rene_amd.scene import Scene, TriangleMesh
from test_gpu_parity import aov_check, t1_check

pytestmark = pytest.mark.gpu


def _both(s, frames, oracle_mod, flags=0):
    o = oracle_mod.Oracle(s)
    o.render(0, frames)
    with api.Renderer(s, flags=abi.FLAG_COUNTERS | flags) as r:
        r.render(0, frames)
        return [r.download(k) for k in range(3)], [o.download(k) for k in range(3)], r.stats().as_dict(), o.stats().as_dict()


def _camera(s, w, h):
    s.set_camera(glam.look_at_lh((0.0, 1.0, -4.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0)), 40.0, w, h)


def test_empty_scene_is_background_only(oracle_mod):
    # no instance at all: every camera ray misses (main_miss, lib.rs:120-139); one ray and one add per path
    for w, h in ((2, 2), (33, 17)):
        s = Scene.new()
        _camera(s, w, h)
        s.set_infinite_light((0.25, 0.5, 0.75))
        for flags in (0, abi.FLAG_WAVEFRONT):
            g, o, sg, so = _both(s, 5, oracle_mod, flags)
            np.testing.assert_array_equal(g[0], np.broadcast_to(np.float32([1.25, 2.5, 3.75]), g[0].shape))
            np.testing.assert_array_equal(g[0], o[0])
            assert (g[1] == 0).all() and (g[2] == 0).all()
            assert sg["rays_closest"] == sg["paths"] == 5 * w * h == so["paths"] and sg["hits"] == 0


def test_degenerate_triangles_are_never_hit(oracle_mod):
    # zero-area triangles (repeated vertex, collinear vertices) next to a real one: no hit, no NaN
    s = Scene.new()
    _camera(s, 48, 32)
    s.set_infinite_light((1.0, 1.0, 1.0))
    m = s.add_matte((0.5, 0.5, 0.5))
    P = [0, 0, 0, 0, 0, 0, 1, 1, 0,   -1, 0, 0, 0, 0, 0, 1, 0, 0,   -1, 0, 1, 1, 0, 1, 0, 1.5, 1]
    s.add_triangle_mesh(TriangleMesh.from_arrays(P, [0, 1, 2, 3, 4, 5, 6, 7, 8]), m)
    for flags in (0, abi.FLAG_FORCE_BVH):
        g, o, sg, so = _both(s, 4, oracle_mod, flags)
        assert np.isfinite(g[0]).all() and sg["paths"] == so["paths"]
        t1_check(g[0], o[0], frac=5e-3, relmse=1e-4)
        assert 0 < sg["hits"] < sg["rays_closest"]  # the real triangle is seen, the degenerate ones never
        aov_check(g[1], o[1], atol=5e-5 * 4, frac=5e-3)


def test_none_material_under_the_path_integrator_ends_the_path(oracle_mod):
    # a None-material surface has no lobes: sample_f returns pdf 0 and the path ends (lib.rs:325-330);
    # its first-hit layers are still written (albedo 0)
    s = Scene.new()
    _camera(s, 40, 30)
    s.set_infinite_light((1.0, 1.0, 1.0))
    s.add_sphere(0.8, 0, ctm=glam.from_translation((0.0, 0.8, 0.0)))
    g, o, sg, so = _both(s, 4, oracle_mod)
    np.testing.assert_allclose(g[0], o[0], rtol=1e-5, atol=1e-6)
    assert sg["rays_closest"] == sg["paths"] == so["rays_closest"]
    always = np.linalg.norm(g[1], axis=2) > 3.9  # all 4 frames of the pixel hit the sphere (unit normals add up)
    assert always.sum() > 50 and (g[0][always] == 0).all()
    assert (g[2] == 0).all()  # albedo of None is 0 (material.rs:727); a miss writes no albedo either


def test_axis_parallel_and_grazing_rays(oracle_mod):
    # rays with exactly-zero direction components (the node tests' safe reciprocal, device_code.inc) and rays
    # lying in the plane of a quad: same hits as the oracle's exact slab tests, and no traversal blow-up
    s = scenes.dragon_class(64, 36, 24, 26)
    o = oracle_mod.Oracle(s)
    rng = np.random.default_rng(9)
    n = 6000
    org = np.stack([rng.uniform(-0.9, 0.9, n), rng.uniform(0.05, 1.9, n), rng.uniform(-0.9, 0.9, n)], 1).astype(np.float32)
    d = np.zeros((n, 3), np.float32)
    axis = rng.integers(0, 3, n)
    d[np.arange(n), axis] = rng.choice([-1.0, 1.0], n)
    two = rng.random(n) < 0.5  # half the rays have one zero component instead of two
    other = (axis + 1) % 3
    d[np.arange(n)[two], other[two]] = rng.uniform(-1, 1, two.sum()).astype(np.float32)
    org[: n // 4, 1] = 0.0  # a quarter start in the floor plane and travel within it
    d[: n // 4, 1] = 0.0
    d[: n // 4, 0] = 1.0
    with api.Renderer(s, flags=abi.FLAG_COUNTERS) as r:
        hg, ho = r.trace(org, d), o.trace(org, d)
    mg, mo = hg["t"] < 0, ho["t"] < 0
    tie = np.abs(hg["t"] - ho["t"]) <= 1e-5 * (1 + np.abs(ho["t"]))
    bad = (mg != mo) | (~mo & (hg["primitive"] != ho["primitive"]) & ~tie)
    assert bad.sum() <= 0.01 * n, bad.sum()  # coplanar starts are ties by construction
    ok = ~mg & ~mo & (hg["primitive"] == ho["primitive"])
    np.testing.assert_allclose(hg["t"][ok], ho["t"][ok], rtol=5e-5, atol=2e-6)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_box_item_merging_on_random_parallelepipeds(oracle_mod, seed):
    """The item loop merges parallelograms into box items (five or six faces of a parallelepiped, any instances).  Random
    scenes of rotated, sheared and mirrored boxes -- closed, with one face missing, sharing a face, built from one mesh or
    from one mesh per face -- plus loose quads and a sphere: every closest hit (instance, primitive, t, u, v) must equal
    the oracle's, which knows nothing about items."""
    rng = np.random.default_rng(seed)
    s = Scene.new()
    _camera(s, 16, 16)
    s.set_infinite_light((1.0, 1.0, 1.0))
    mats = [s.add_matte(tuple(rng.uniform(0.2, 0.9, 3))) for _ in range(3)]

    def rand_affine(scale):
        a = rng.normal(size=(3, 3))
        qm, _ = np.linalg.qr(a)
        shear = np.eye(3) + np.triu(rng.uniform(-0.4, 0.4, (3, 3)), 1)
        m3 = qm @ shear @ np.diag(rng.uniform(0.4, 1.0, 3) * scale * rng.choice([-1.0, 1.0], 3))
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = m3
        m[:3, 3] = rng.uniform(-1.5, 1.5, 3)
        return m.T.copy() if False else m

    def box_faces(lo, hi):
        mesh = scenes._aabb(lo, hi)
        v, idx = mesh.vertices, mesh.indices
        return [TriangleMesh.from_arrays(v[:, 0:3].reshape(-1), idx[6 * f: 6 * f + 6], normals=v[:, 3:6], uvs=v[:, 6:8].reshape(-1))
                for f in range(6)]

    n_boxes = int(rng.integers(2, 5))
    for b in range(n_boxes):
        ctm = glam.from_cols_array(rand_affine(0.5).T.reshape(-1).tolist())
        style = int(rng.integers(0, 4))
        faces = box_faces((-1, -1, -1), (1, 1, 1))
        if style == 0:  # one mesh, six faces
            s.add_triangle_mesh(scenes._aabb((-1, -1, -1), (1, 1, 1)), mats[b % 3], ctm=ctm)
        elif style == 1:  # one instance per face (different materials), all six
            for f in range(6):
                s.add_triangle_mesh(faces[f], mats[f % 3], ctm=ctm)
        elif style == 2:  # five faces: one side open
            skip = int(rng.integers(0, 6))
            for f in range(6):
                if f != skip:
                    s.add_triangle_mesh(faces[f], mats[f % 3], ctm=ctm)
        else:  # two boxes sharing a face (the shared face appears twice)
            s.add_triangle_mesh(scenes._aabb((-1, -1, -1), (0, 1, 1)), mats[0], ctm=ctm)
            s.add_triangle_mesh(scenes._aabb((0, -1, -1), (1, 1, 1)), mats[1], ctm=ctm)
    for _ in range(int(rng.integers(0, 3))):  # loose quads
        ctm = glam.from_cols_array(rand_affine(0.7).T.reshape(-1).tolist())
        s.add_triangle_mesh(TriangleMesh.from_arrays([-1, 0, -1, 1, 0, -1, 1, 0, 1, -1, 0, 1], [0, 1, 2, 0, 2, 3]), mats[0], ctm=ctm)
    s.add_sphere(0.4, mats[1], ctm=glam.from_translation(tuple(rng.uniform(-1, 1, 3))))
    info = api.pack_info(s)
    if not (info.features & 64):
        pytest.skip("scene too large for the item loop")
    n_quads = (info.n_triangles) // 2
    assert info.n_items_main < n_quads + 1  # at least one box was formed (a sphere adds one item)
    o = oracle_mod.Oracle(s)
    n = 60000
    org = rng.uniform(-2.5, 2.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    with api.Renderer(s) as r:
        hg = r.trace(org, d)
    ho = o.trace(org, d)
    mg, mo = hg["t"] < 0, ho["t"] < 0
    tie = np.abs(hg["t"] - ho["t"]) <= 2e-5 * (1 + np.abs(ho["t"]))  # coincident faces of adjacent boxes, edges
    wrong = (mg != mo) | (~mo & ~mg & ~tie)
    assert wrong.sum() <= 3, (int(wrong.sum()), int((mg != mo).sum()))
    same = ~mg & ~mo & (hg["instance"] == ho["instance"]) & (hg["primitive"] == ho["primitive"])
    assert same.sum() > 0.9 * (~mo).sum()  # the rest are ties between coincident / adjacent faces
    np.testing.assert_allclose(hg["t"][same], ho["t"][same], rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(hg["u"][same], ho["u"][same], atol=2e-4)
    np.testing.assert_allclose(hg["v"][same], ho["v"][same], atol=2e-4)


# ---- round 3: long launches, seed tables, the epoch's wrap ---------------------------------------------------------------------
@pytest.mark.parametrize("bvh", [False, True])
def test_long_launches_and_their_seed_tables(oracle_mod, monkeypatch, bvh):
    """One launch renders a whole job.  Up to 1024 frames its seeds sit in LDS one per frame; beyond, as two-level tables
    (the generator's state every 2^s frames + the affine map of j < 2^s frame steps, device_scene.h SEED_TAB_*); with no room in
    LDS every path start composes the jump itself (RENE_NO_LDS_SEEDS): all three must draw the same seeds -- bit-identical images,
    also against launches cut differently, and T1 against the oracle's frames 0 .. 1499."""
    s = scenes.cornell_box(24, 16)
    flags = abi.FLAG_FORCE_BVH if bvh else 0
    with api.Renderer(s, flags=flags) as r:
        r.render(0, 1500)  # two-level tables
        whole = [r.download(l) for l in range(3)]
        r.reset()
        r.render(0, 700)   # direct tables
        r.render(700, 800)
        for l in range(3):
            assert np.array_equal(r.download(l), whole[l])
    monkeypatch.setenv("RENE_NO_LDS_SEEDS", "1")
    with api.Renderer(s, flags=flags) as r:
        r.render(0, 1500)
        for l in range(3):
            assert np.array_equal(r.download(l), whole[l])
    monkeypatch.delenv("RENE_NO_LDS_SEEDS")
    o = oracle_mod.Oracle(s)
    o.render(0, 1500)
    ref = o.download(0)
    assert float(((whole[0] - ref) ** 2).sum() / (ref ** 2).sum()) < 1e-4
    # frame shards deal frames round-robin: the tables then step by `shard_count` frames
    acc = np.zeros_like(whole[0])
    for rank in range(3):
        with api.Renderer(s, flags=flags, shard_mode=abi.SHARD_FRAMES, shard_rank=rank, shard_count=3) as r:
            r.render(0, 1500)
            acc += r.download(0)
    np.testing.assert_allclose(acc, whole[0], rtol=2e-5, atol=1e-4)


def test_requests_longer_than_a_launch_are_cut(oracle_mod):
    """rene_render(first, n) with n beyond 65 536 frames (the most one launch's seed tables cover) becomes several launches."""
    s = scenes.cornell_box(8, 8)
    with api.Renderer(s) as r:
        r.render(0, 70000)
        a, st = r.download(0), r.stats().as_dict()
        r.reset()
        r.render(0, 65536)
        r.render(65536, 70000 - 65536)
        assert np.array_equal(r.download(0), a)
    assert st["launches"] == 2 and st["frames"] == 70000 and st["paths"] == 64 * 70000


@pytest.mark.parametrize("bvh", [False, True])
def test_the_epoch_wraps_without_a_trace(monkeypatch, bvh):
    """A pixel record's version is epoch << 10 | items committed; after 2^22 - 1 launches the epoch starts again and the version
    words are cleared.  RENE_TEST_EPOCH starts a context four launches below the wrap: ten launches across it give the image of
    a fresh context, bit for bit."""
    s = scenes.cornell_box(64, 48)
    flags = abi.FLAG_FORCE_BVH if bvh else 0
    with api.Renderer(s, flags=flags) as r:
        for k in range(10):
            r.render(6 * k, 6)
        want = [r.download(l) for l in range(3)]
    monkeypatch.setenv("RENE_TEST_EPOCH", str((1 << 22) - 1 - 4))
    with api.Renderer(s, flags=flags) as r:
        for k in range(10):
            r.render(6 * k, 6)
        for l in range(3):
            assert np.array_equal(r.download(l), want[l])


def test_a_camera_matrix_that_is_not_finite_is_refused():
    # the kernels read the camera's origin off camera_to_world (c2w . (0, 0, 0), camera.rs:79) instead of multiplying a
    # zero point through it: equal for every finite matrix, so a matrix that is not finite is an argument error
    for field, bad in (("camera_to_world", np.nan), ("projection_inv", np.inf)):
        s = scenes.cornell_box(16, 16)
        m = np.array(getattr(s, field), dtype=np.float64, copy=True)
        m[1, 2] = bad
        setattr(s, field, m)
        with pytest.raises(api.ReneError) as e:
            api.Renderer(s)
        assert e.value.code == -1 and "finite" in str(e.value)


def test_image_map_repeats_over_many_periods(oracle_mod):
    # REPEAT addressing (rene/src/main.rs:2390-2397): a quad whose uv run from -3.6 to 4.4 and from -7.3 to 2.9, so that texel
    # indices lie within one period of the image (wrapped by an add or a subtract), several periods below it and several above
    # (the general remainder); the first-hit albedo layer is the texture's colour at the hit, compared with the oracle's
    rng = np.random.default_rng(5)
    img = rng.random((5, 7, 3), dtype=np.float32)
    for flags in (0, abi.FLAG_FORCE_BVH):
        s = Scene.new()
        _camera(s, 96, 64)
        s.set_infinite_light((1.0, 1.0, 1.0))
        m = s.add_matte(s.add_texture_image_map(img))
        P = [-2.0, -0.5, 0.0,  2.0, -0.5, 0.0,  2.0, 1.5, 0.0,  -2.0, 1.5, 0.0]
        UV = [-3.6, -7.3,  4.4, -7.3,  4.4, 2.9,  -3.6, 2.9]
        s.add_triangle_mesh(TriangleMesh.from_arrays(P, [0, 1, 2, 0, 2, 3], uvs=UV), m)
        g, o, sg, so = _both(s, 4, oracle_mod, flags)
        assert sg["hits"] > 0.3 * sg["paths"]
        # (the quad's uv gradient is 8 periods x 7 texels across it: a hit's barycentrics, equal to the oracle's to ~2e-5, move the look-up by
        # ~1e-3 of a texel, i.e. ~1e-3 of the texture's contrast per frame; a wrong wrap lands on another texel: ~0.3)
        aov_check(g[2], o[2], atol=6e-3 * 4, frac=2e-2)
        assert abs(float(g[2].mean()) - float(o[2].mean())) <= 2e-3 * float(o[2].mean())
        t1_check(g[0], o[0], frac=2e-2, relmse=1e-3)


@pytest.mark.parametrize("name", ["cornell", "dragon"])
def test_tile_shards_of_a_ragged_image_clear_and_add_their_chains_over_the_owned_tiles_only(name):
    """Round 4: a tile shard's eight frame chains hold something in the tiles it owns only, so rene_reset clears and the hand-out adds the
    chains over those tiles alone (kernels.hip, chains_tiles_kernel).  On an image whose size is no multiple of the 32 x 32 tile (partial
    tiles at the right and bottom edges), with three ranks: a job rendered AFTER another job and a reset equals the same job on a fresh
    context bit for bit (nothing of the first job survives in any chain), nothing is written outside the owned tiles, and the ranks'
    images add up to the unsharded one bit for bit."""
    s = scenes.cornell_box(100, 70) if name == "cornell" else scenes.dragon_class(100, 70, 24, 26)
    with api.Renderer(s) as r:
        r.render(0, 11)
        whole = [r.download(l) for l in range(3)]
    acc = [np.zeros_like(w) for w in whole]
    for rank in range(3):
        with api.Renderer(s, shard_mode=abi.SHARD_TILES, shard_rank=rank, shard_count=3) as r:
            r.render(5, 9)          # another job first ...
            r.sync()
            r.reset()               # ... cleared over the owned tiles
            r.render(0, 4)
            r.download(1)           # (a hand-out in the middle of the job)
            r.render(4, 7)
            got = [r.download(l) for l in range(3)]
        with api.Renderer(s, shard_mode=abi.SHARD_TILES, shard_rank=rank, shard_count=3) as f:
            f.render(0, 11)
            for l in range(3):
                assert np.array_equal(got[l], f.download(l)), (rank, l)
        ys, xs = np.nonzero(np.abs(got[1]).sum(axis=2) > 0)  # layer 1: first-hit normals, non-zero wherever the camera ray hit something
        tiles = (ys // 32) * ((100 + 31) // 32) + xs // 32
        assert (tiles % 3 == rank).all(), rank
        for l in range(3):
            acc[l] += got[l]
    for l in range(3):
        assert np.array_equal(acc[l], whole[l]), l
