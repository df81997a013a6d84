"""-m gpu: per-function parity of the device BSDF code (materials -> lobes -> f / pdf / sample_f)
against the oracle, through the rene_bsdf_eval probe.  Tolerance: 2e-4 relative (+ 1e-6 absolute) on
well-conditioned inputs; grazing configurations and the alpha = 0.01 lobe are ill-conditioned in
the reference's own fp32 formulas (1 - cos^2 cancellation in microfacet.rs:141-155) and get 2e-2."""
import numpy as np
import pytest

from rene_amd import abi, api, scenes

pytestmark = pytest.mark.gpu


def _dirs(rng, n, upper=None):
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if upper is not None:
        d[:, 2] = np.abs(d[:, 2]) * (1 if upper else -1)
    return d.astype(np.float32)


def _probe(scene, oracle_mod, material, n, seed, away_from_grazing=0.15):
    rng = np.random.default_rng(seed)
    normals = _dirs(rng, n)
    # wo / wi expressed around the normal so that grazing angles can be excluded
    wo_l, wi_l = _dirs(rng, n, upper=True), _dirs(rng, n)
    keep = (np.abs(wo_l[:, 2]) > away_from_grazing) & (np.abs(wi_l[:, 2]) > away_from_grazing)
    uvs = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    seeds = rng.integers(0, 2 ** 32, n, dtype=np.uint32)
    o = oracle_mod.Oracle(scene)

    def frame(nrm):
        w = nrm / np.linalg.norm(nrm)
        a = np.array([0, w[2], -w[1]]) if abs(w[0]) <= abs(w[1]) else np.array([-w[2], 0, w[0]])
        u = a / np.linalg.norm(a)
        return u, np.cross(w, u), w

    wo = np.zeros_like(wo_l)
    wi = np.zeros_like(wi_l)
    for i in range(n):
        u, v, w = frame(normals[i].astype(np.float64))
        wo[i] = wo_l[i, 0] * u + wo_l[i, 1] * v + wo_l[i, 2] * w
        wi[i] = wi_l[i, 0] * u + wi_l[i, 1] * v + wi_l[i, 2] * w
    with api.Renderer(scene) as r:
        g = r.bsdf_eval(material, normals, uvs, wo, wi, seeds)
    ref = np.zeros_like(g)
    for i in range(n):
        e = o.bsdf_eval(material, normals[i], uvs[i], wo[i], wi[i], int(seeds[i]))
        ref[i] = np.concatenate([e["f"], [e["pdf"]], e["s_wi"], e["s_f"], [e["s_pdf"]], [e["len"]]])
    return g, ref, keep


def _close(g, ref, rtol, atol=1e-6):
    return np.abs(g - ref) <= atol + rtol * np.abs(ref)


ZOO_MATERIALS = {  # index into material_zoo()'s material table -> (name, rtol)
    "matte_checker": 2e-4, "matte_scale": 2e-4, "glass": 2e-4, "mirror": 2e-4, "metal_aniso": 2e-3,
    "substrate": 2e-3, "plastic": 2e-3, "uber": 2e-3,
}


def test_zoo_materials_match_oracle(oracle_mod):
    s = scenes.material_zoo(32, 32)
    # materials[0] is the None sentinel; the zoo adds: matte(checks)=1, matte(scale)=2, glass=3, mirror=4,
    # metal=5, substrate=6, plastic=7, uber=8, matte(light)=9, metal(imap)=10
    names = {1: "matte_checker", 2: "matte_scale", 3: "glass", 4: "mirror", 5: "metal_aniso",
             6: "substrate", 7: "plastic", 8: "uber", 10: "metal_aniso"}
    for mat, name in names.items():
        g, ref, keep = _probe(s, oracle_mod, mat, 600, seed=100 + mat)
        assert np.array_equal(g[:, 11], ref[:, 11]), name  # lobe counts
        rtol = ZOO_MATERIALS[name]
        ok_fp = _close(g[:, :4], ref[:, :4], rtol).all(axis=1)
        assert ok_fp[keep].mean() > 0.995, (name, "f/pdf", ok_fp[keep].mean())
        # samples: identical lobe choice and direction unless a discrete decision flipped
        ok_s = _close(g[:, 4:11], ref[:, 4:11], 10 * rtol, atol=1e-5).all(axis=1)
        assert ok_s[keep].mean() > 0.98, (name, "sample", ok_s[keep].mean())


def test_none_material_has_no_lobes(oracle_mod):
    s = scenes.material_zoo(32, 32)
    g, ref, _ = _probe(s, oracle_mod, 0, 16, seed=3)
    assert (g[:, 11] == 0).all() and (g[:, :11] == 0).all() and np.array_equal(g, ref)


def test_veach_metals(oracle_mod):
    s = scenes.veach_mis(32, 32)
    # materials: 1 diffuse, 2 smooth (alpha .01), 3 glossy (.05), 4 rough (.1), 5 null, 6 super rough (.25)
    for mat, rtol, frac in ((6, 2e-3, 0.99), (4, 5e-3, 0.98), (3, 2e-2, 0.97), (2, 5e-2, 0.9)):
        g, ref, keep = _probe(s, oracle_mod, mat, 800, seed=7 + mat, away_from_grazing=0.25)
        assert np.array_equal(g[:, 11], ref[:, 11])
        ok = _close(g[:, :4], ref[:, :4], rtol, atol=1e-5).all(axis=1)
        assert ok[keep].mean() > frac, (mat, ok[keep].mean())
        s_ok = _close(g[:, 4:7], ref[:, 4:7], 0, atol=5e-3).all(axis=1)  # sampled directions
        assert s_ok[keep].mean() > frac, (mat, "wi", s_ok[keep].mean())


def test_materials_resolved_at_upload_render_bit_identically(monkeypatch):
    """Single-lobe general materials over solid textures are resolved into the instance record when the scene is packed
    (device_scene.h, Inst::res_*); RENE_NO_RESOLVE sends them through the material / texture tables instead.  Same lobes,
    so the same image bit for bit -- on the item-loop kernel (zoo, veach-mis) and on the BVH kernels (the same scenes forced)."""
    for make, flags in ((lambda: scenes.material_zoo(160, 120), 0), (lambda: scenes.material_zoo(160, 120), abi.FLAG_FORCE_BVH),
                        (lambda: scenes.veach_mis(128, 128), 0), (lambda: scenes.veach_mis(128, 128), abi.FLAG_FORCE_BVH)):
        images = []
        for knob in (None, "1"):
            if knob is None:
                monkeypatch.delenv("RENE_NO_RESOLVE", raising=False)
            else:
                monkeypatch.setenv("RENE_NO_RESOLVE", knob)
            with api.Renderer(make(), flags=flags) as r:
                r.render(0, 6)
                images.append([r.download(l) for l in range(3)])
        for a, b in zip(*images):
            assert np.array_equal(a, b)


def test_device_answers_the_committed_per_function_vectors():
    """tests/golden/bxdf_vectors.npz: the device's rene_bsdf_eval against the COMMITTED answers (no oracle at run time): lobe counts
    exact, f / pdf to the material's tolerance, samples unless a discrete decision flipped."""
    import os
    from conftest import GOLDEN
    data = np.load(os.path.join(GOLDEN, "bxdf_vectors.npz"))
    scene = {"zoo": scenes.material_zoo(32, 32), "veach": scenes.veach_mis(32, 32)}
    tol = {"zoo": {1: 2e-4, 2: 2e-4, 3: 2e-4, 4: 2e-4, 5: 2e-3, 6: 2e-3, 7: 2e-3, 8: 2e-3, 10: 2e-3}, "veach": {1: 2e-4, 2: 5e-2, 3: 2e-2, 4: 5e-3, 6: 2e-3}}
    for tag in ("zoo", "veach"):
        with api.Renderer(scene[tag]) as r:
            for key in [k for k in data.files if k.startswith(tag)]:
                m = int(key.split("_")[1])
                v = data[key]
                g = r.bsdf_eval(m, v[:, 0:3].copy(), v[:, 3:5].copy(), v[:, 5:8].copy(), v[:, 8:11].copy(), v[:, 11].copy().view(np.uint32))
                want = v[:, 12:]
                assert np.array_equal(g[:, 11], want[:, 11]), key
                ok = _close(g[:, :4], want[:, :4], tol[tag][m], atol=1e-5).all(axis=1)
                assert ok.mean() > 0.95, (key, "f / pdf", ok.mean())
                ok_s = _close(g[:, 4:11], want[:, 4:11], 10 * tol[tag][m], atol=5e-3).all(axis=1)
                assert ok_s.mean() > 0.9, (key, "sample", ok_s.mean())
