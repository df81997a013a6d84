"""T0 tier: integer-exact pins of the oracle (rene-shader/src/rand.rs, camera.rs)."""
import json
import os

import numpy as np

from conftest import GOLDEN


def test_pcg32si_known_answers(oracle_mod):
    kat = json.load(open(os.path.join(GOLDEN, "pcg32si_kat.json")))
    for seed_s, rec in kat.items():
        seed = int(seed_s)
        assert oracle_mod.pcg_state_after_new(seed) == rec["state_after_new"]
        assert oracle_mod.pcg_u32(seed, 8).tolist() == rec["u32"]
        f = oracle_mod.pcg_f32(seed, 8)
        want = np.array(rec["f32_bits_num"], dtype=np.float64) / 2.0 ** 24
        assert (f.astype(np.float64) == want).all()  # exact: 24-bit integers scale exactly


def test_pcg32si_survey_vectors(oracle_mod):
    # SURVEY.md section 8a, row A2 (restated from rand.rs by the survey, independently of this repo)
    assert oracle_mod.pcg_state_after_new(0) == 0x4712A88E
    assert [hex(v) for v in oracle_mod.pcg_u32(0, 4)] == ["0x22b6b6bc", "0x3bf6e0b1", "0x572f7439", "0x86fc4ddc"]
    assert oracle_mod.pcg_f32(0, 2).tolist() == [0.13560044765472412, 0.2342357635498047]
    assert [hex(v) for v in oracle_mod.pcg_u32(1, 2)] == ["0x1adc3cd6", "0xeb531668"]
    assert [hex(v) for v in oracle_mod.pcg_u32(42, 2)] == ["0x1c271671", "0x2da40d5d"]
    assert [hex(v) for v in oracle_mod.pcg_u32(0xDEADBEEF, 2)] == ["0xa7039ed0", "0x4b895ec7"]


def test_f32_range_and_uniformity(oracle_mod):
    f = oracle_mod.pcg_f32(12345, 200000)
    assert f.min() >= 0.0 and f.max() < 1.0
    assert abs(f.mean() - 0.5) < 5e-3
    hist, _ = np.histogram(f, bins=16, range=(0, 1))
    assert hist.min() > 0.9 * len(f) / 16


def test_camera_rays_cornell(oracle_mod):
    from rene_amd import scenes
    o = oracle_mod.Oracle(scenes.cornell_box(64, 64))
    g = json.load(open(os.path.join(GOLDEN, "cornell_camera_rays.json")))
    for s, t, ro, rd in zip(g["s"], g["t"], g["o"], g["d"]):
        a, b = o.camera_ray(s, t)
        np.testing.assert_allclose(a, ro, rtol=0, atol=1e-6)
        np.testing.assert_allclose(b, rd, rtol=0, atol=1e-6)
    # closed form: dir_cam = (sx * aspect * tan(fov/2), sy * tan(fov/2), 1); the Cornell
    # camera sits at (0, 1, 6.8) looking down -z
    ro, rd = o.camera_ray(0.5, 0.5)
    np.testing.assert_allclose(ro, [0, 1, 6.8], atol=1e-6)
    np.testing.assert_allclose(rd, [0, 0, -1], atol=1e-6)
    ro, rd = o.camera_ray(1.0, 1.0)
    th = np.tan(np.radians(19.5) / 2)
    want = np.array([th, th, -1.0]) / np.sqrt(2 * th * th + 1)
    np.testing.assert_allclose(rd, want, atol=2e-6)
