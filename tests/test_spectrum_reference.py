"""Spectral colours against the reference.

`"spectrum Kd" "file.spd"` goes through from_sampled (rene/src/scene/spectrum.rs:1487-1506) over the CIE 1931 2-degree
tables the reference tabulates (spectrum.rs:5-1467).  The loader carries the same standard table (rene_amd/csrc/cie1931.inc,
written by tools/make_cie_table.py) and restates from_sampled in f32.  Here:
  * without the checkout: the committed table has the CIE's own invariants;
  * with the checkout (`reference` marker): the table literals are parsed out of spectrum.rs AT TEST TIME (nothing of that file
    is kept in the repo), from_sampled / interpolate are restated in numpy f32 over THOSE numbers, and the loader's RGB for
    several .spd files must equal it to 1e-6.
"""
import os
import re

import numpy as np
import pytest

from conftest import REFERENCE, ROOT, have_reference
from rene_amd import loader

SPEC_SCENE = 'Camera "perspective"\nWorldBegin\nLightSource "distant" %s\nShape "sphere"\nWorldEnd\n'


def _committed_table():
    text = open(os.path.join(ROOT, "rene_amd", "csrc", "cie1931.inc")).read()
    out = {}
    for name in ("kCieX", "kCieY", "kCieZ"):
        body = re.search(r"%s\[471\] = \{(.*?)\};" % name, text, re.S).group(1)
        body = re.sub(r"//[^\n]*", "", body)
        out[name] = np.array([float(t.rstrip("f")) for t in re.findall(r"[-+0-9.eE]+f", body)], np.float32)
        assert out[name].shape == (471,)
    out["y_integral"] = np.float32(re.search(r"kCieYIntegral = ([0-9.]+)f", text).group(1))
    return out


def _reference_table():
    text = open(os.path.join(REFERENCE, "rene", "src", "scene", "spectrum.rs")).read()
    out = {}
    for name in ("CIE_X", "CIE_Y", "CIE_Z", "CIE_LAMBDA"):
        body = re.search(r"const %s: \[f32; N_CIE_SAMPLES\] = \[(.*?)\];" % name, text, re.S).group(1)
        out[name] = np.array([float(t) for t in re.findall(r"[-+0-9.eE]+", body)], np.float32)
        assert out[name].shape == (471,)
    out["y_integral"] = np.float32(re.search(r"const CIE_Y_INTEGRAL: f32 = ([0-9.]+);", text).group(1))
    return out


def _light_L(text, base):
    t = loader.parse_pbrt(text, str(base)).tables()
    return np.frombuffer(t["lights"][0].tobytes(), np.float32)[5:8].copy()


def test_committed_cie_table_has_the_standards_invariants():
    t = _committed_table()
    x, y, z = t["kCieX"], t["kCieY"], t["kCieZ"]
    assert y[555 - 360] == np.float32(1.0) and int(np.argmax(y)) == 555 - 360          # V(lambda) peaks at 555 nm with value 1
    sums = [float(np.sum(v.astype(np.float64))) for v in (x, y, z)]
    assert abs(sums[1] - float(t["y_integral"])) < 1e-3                                # CIE_Y_INTEGRAL is the table's own sum
    assert abs(sums[0] / sums[1] - 1) < 5e-4 and abs(sums[2] / sums[1] - 1) < 5e-4     # equal-energy white: x = y = z = 1/3 (table: 1.00008, 1.00033)
    assert (x >= 0).all() and (y >= 0).all() and (z >= 0).all()
    assert int(np.argmax(z)) in range(440 - 360, 452 - 360) and int(np.argmax(x)) in range(595 - 360, 603 - 360)


def _from_sampled(spectrum, T):
    """spectrum.rs:1468-1506 in numpy f32, over the table T parsed from the checkout."""
    f = np.float32
    sp = sorted((f(l), f(v)) for l, v in spectrum)
    ls = [l for l, _ in sp]

    def interpolate(l):  # spectrum.rs:1468-1485 (binary_search: Ok(i) | Err(i) -> i; then segment i .. i + 1)
        if l < sp[0][0]:
            return sp[0][1]
        if l > sp[-1][0]:
            return sp[-1][1]
        i = int(np.searchsorted(np.array(ls, np.float32), l, side="left"))
        t = f(f(l - sp[i][0]) / f(sp[i + 1][0] - sp[i][0]))
        return f(f(f(1.0) - t) * sp[i][1]) + f(t * sp[i + 1][1])

    xyz = [f(0), f(0), f(0)]
    for i in range(471):
        val = f(interpolate(T["CIE_LAMBDA"][i]))
        xyz[0] = f(xyz[0] + f(val * T["CIE_X"][i]))
        xyz[1] = f(xyz[1] + f(val * T["CIE_Y"][i]))
        xyz[2] = f(xyz[2] + f(val * T["CIE_Z"][i]))
    scale = f(f(T["CIE_LAMBDA"][470] - T["CIE_LAMBDA"][0]) / f(T["y_integral"] * f(471)))
    x, y, z = (f(c * scale) for c in xyz)
    return np.array([f(f(f(3.240479) * x) - f(f(1.537150) * y)) - f(f(0.498535) * z),
                     f(f(f(-0.969256) * x) + f(f(1.875991) * y)) + f(f(0.041556) * z),
                     f(f(f(0.055648) * x) - f(f(0.204043) * y)) + f(f(1.057311) * z)], np.float32)


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference/rene/src/scene/spectrum.rs")
def test_committed_table_equals_the_checkouts():
    a, b = _committed_table(), _reference_table()
    for k, r in (("kCieX", "CIE_X"), ("kCieY", "CIE_Y"), ("kCieZ", "CIE_Z")):
        assert np.array_equal(a[k], b[r]), k
    assert a["y_integral"] == b["y_integral"]
    assert np.array_equal(b["CIE_LAMBDA"], np.arange(360, 831, dtype=np.float32))


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference/rene/src/scene/spectrum.rs")
def test_spd_colours_equal_from_sampled_over_the_references_tables(tmp_path, hip_lib):
    T = _reference_table()
    rng = np.random.default_rng(5)
    spds = {
        "flat": [(300.0, 1.0), (500.0, 1.0), (700.0, 1.0), (850.0, 1.0), (900.0, 1.0)],
        "ramp": [(float(l), (l - 300) / 600.0) for l in range(300, 905, 5)],
        "green": [(float(l), float(np.exp(-0.5 * ((l - 535) / 25.0) ** 2))) for l in range(300, 905, 5)],
        "noise": [(float(l), float(v)) for l, v in zip(range(350, 871, 10), rng.uniform(0, 2, 53))],
        "unsorted": [(700.5, 0.2), (340.0, 0.9), (520.25, 1.4), (900.0, 0.1), (610.0, 0.7), (450.75, 0.3), (950.0, 0.05)],
    }
    for name, sp in spds.items():
        (tmp_path / f"{name}.spd").write_text("".join(f"{l:.6f} {v:.6f}\n" for l, v in sp))
        sp_file = [(np.float32(f"{l:.6f}"), np.float32(f"{v:.6f}")) for l, v in sp]  # what both parsers read back
        got = _light_L(SPEC_SCENE % f'"spectrum L" "{name}.spd"', tmp_path)
        want = _from_sampled(sp_file, T)
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6, err_msg=name)
