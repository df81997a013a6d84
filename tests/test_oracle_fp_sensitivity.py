"""How much of the GPU-vs-oracle per-pixel difference is plain fp32 rounding of the reference's own
formulas?  Build the SAME oracle source twice -- without FMA contraction (the parity oracle) and
with it (-ffp-contract=fast -mfma, what any GPU compiler does) -- and compare the two CPU renders.

On Cornell (Matte, well-conditioned) the two agree to ~1e-6.  On veach-mis they do not: the cone pdf
of sphere emitters (rene-shader/src/lib.rs:1058-1064) computes 1 - sqrt(1 - r^2/d^2), which for the
r = 0.05 light is ~1e-6 against an fp32 ulp of 6e-8.  This is why tests/test_gpu_scenes.py uses an
image-level tolerance for that scene."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from rene_amd import abi, scenes


def _has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.fixture(scope="module")
def fma_oracle():
    if not _has_fma():
        pytest.skip("host CPU has no FMA")
    so = os.path.join(ROOT, "oracle", "_variants", "librene_oracle_fma.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    src = os.path.join(ROOT, "oracle", "rene_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-mfma",
                               "-ffp-contract=fast", "-o", so, src])
    L = C.CDLL(so)
    L.oracle_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_void_p)]
    L.oracle_render.argtypes = [C.c_void_p] + [C.c_uint32] * 3 + [C.c_int] + [C.c_uint32] * 3
    L.oracle_download.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.oracle_destroy.argtypes = [C.c_void_p]
    return L


def _render(L, scene, frames):
    p = scene.to_desc()
    h = C.c_void_p()
    assert L.oracle_create(p.byref(), C.byref(h)) == 0
    L.oracle_render(h, abi.DEFAULT_SEED, 0, frames, 0, 0, 0, 1)
    out = np.empty((p.yres, p.xres, 3), np.float32)
    L.oracle_download(h, 0, 3, out.ctypes.data_as(C.c_void_p), out.size)
    L.oracle_destroy(h)
    return out


def _off(a, b, thr=1e-2):
    return float(((np.abs(a - b) / (1 + np.abs(b))).max(axis=2) > thr).mean())


def test_fma_vs_no_fma_on_cpu(fma_oracle, oracle_mod):
    base = oracle_mod.lib()
    c0, c1 = _render(base, scenes.cornell_box(96, 96), 8), _render(fma_oracle, scenes.cornell_box(96, 96), 8)
    v0, v1 = _render(base, scenes.veach_mis(160, 90), 16), _render(fma_oracle, scenes.veach_mis(160, 90), 16)
    print("cornell off:", _off(c0, c1), " veach off:", _off(v0, v1), " veach relMSE:", float(((v0 - v1) ** 2).sum() / (v1 ** 2).sum()))
    assert _off(c0, c1) < 2e-3                     # well-conditioned: rounding stays rounding
    assert 5e-3 < _off(v0, v1) < 0.2               # ill-conditioned cone pdf: percent-level pixels move
    assert abs(float(v0.sum() / v1.sum()) - 1) < 2e-3  # ...but the image as a whole does not
