"""ctypes binding of oracle/librene_oracle.so -- the CPU restatement of rene's integrator.

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rene_amd import abi

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_DIR, "librene_oracle.so")
    src = os.path.join(_DIR, "rene_oracle.cpp")
    hdr = os.path.join(_DIR, "..", "include", "rene_hip.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _DIR, "-B", "librene_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_void_p)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_reset.argtypes = [C.c_void_p]
        L.oracle_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                    C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_download.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.oracle_get_stats.argtypes = [C.c_void_p, C.POINTER(abi.Stats)]
        for name in ("oracle_trace", "oracle_trace_bruteforce"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_float, C.c_float, C.c_void_p]
        L.oracle_emitter_pdf.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_pcg_state_after_new.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
        L.oracle_pcg_u32.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_pcg_f32.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_medium_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_size_t] + [C.c_void_p] * 6
        L.oracle_fr_dielectric.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.oracle_fr_conductor.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_tr_d_lambda.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_roughness_to_alpha.argtypes = [C.c_float, C.c_void_p]
        L.oracle_tex_color.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.oracle_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_set_experiment.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.oracle_set_experiment.restype = None
        L.oracle_download_decomposition.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """CPU reference integrator over one scene; mirrors rene_amd.api.Renderer's surface."""

    def __init__(self, scene):
        self._packed = scene.to_desc() if hasattr(scene, "to_desc") else scene
        self._h = C.c_void_p()
        rc = lib().oracle_create(self._packed.byref(), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"oracle_create failed ({rc}): {lib().oracle_last_error().decode()}")
        self.xres, self.yres = self._packed.xres, self._packed.yres

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def reset(self):
        lib().oracle_reset(self._h)

    def render(self, first_frame: int, n_frames: int, seed: int = abi.DEFAULT_SEED, threads: int = 0,
               shard_mode: int = abi.SHARD_TILES, shard_rank: int = 0, shard_count: int = 1):
        lib().oracle_render(self._h, seed, first_frame, n_frames, threads, shard_mode, shard_rank,
                            shard_count)

    def download(self, layer: int = 0, channels: int = 3) -> np.ndarray:
        out = np.empty((self.yres, self.xres, channels), dtype=np.float32)
        rc = lib().oracle_download(self._h, layer, channels, _p(out), out.size)
        assert rc == 0
        return out

    def stats(self) -> abi.Stats:
        st = abi.Stats()
        lib().oracle_get_stats(self._h, C.byref(st))
        return st

    # one-statement departures from the restatement (Scene::X_* in rene_oracle.cpp; tools/cornell_offsets.py), all off by default
    X = {"pdf_wo_wi": 1 << 0, "pdfl_occluded": 1 << 1, "no_primcount": 1 << 2, "rr_from_3": 1 << 3, "rr_off": 1 << 4, "rr_clamp": 1 << 5,
         "emit_two_sided": 1 << 6, "tie_last": 1 << 7, "pos_from_ray": 1 << 8, "pdfl_front_only": 1 << 9, "pixel_rng": 1 << 10,
         "jitter_w": 1 << 11, "light_pdf_area": 1 << 12, "tmin_1e4": 1 << 13, "cos_from_ng": 1 << 14, "pdf_faceforward": 1 << 15,
         "pdf_zero": 1 << 16}

    def set_experiment(self, bits: int = 0, decomposition=False, depth_cap: int = 0):
        """decomposition: False / True (every layer-0 add kept by bounce and branch) / 2 (by bounce and by the instance the light branch was taken at)"""
        lib().oracle_set_experiment(self._h, (bits | (depth_cap << 24)) & 0xFFFFFFFF, int(decomposition))

    def download_decomposition(self, depth: int, branch: int) -> np.ndarray:
        out = np.empty((self.yres, self.xres, 3), dtype=np.float32)
        assert lib().oracle_download_decomposition(self._h, depth, branch, _p(out)) == 0
        return out

    def trace(self, origins, directions, tmin=0.001, tmax=1e5, which=0, bruteforce=False) -> np.ndarray:
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(o.shape[0], dtype=HIT_DTYPE)
        fn = lib().oracle_trace_bruteforce if bruteforce else lib().oracle_trace
        fn(self._h, which, o.shape[0], _p(o), _p(d), tmin, tmax, _p(out))
        return out

    def emitter_pdf(self, origins, directions):
        """(pdf_l in fp32 as the reference computes it, the sphere formula in fp64) per ray -- lib.rs:301-318, 959-1066."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(o.shape[0], np.float32)
        out64 = np.zeros(o.shape[0], np.float64)
        lib().oracle_emitter_pdf(self._h, o.shape[0], _p(o), _p(d), _p(out), _p(out64))
        return out, out64

    def camera_ray(self, s: float, t: float):
        o = np.zeros(3, np.float32)
        d = np.zeros(3, np.float32)
        lib().oracle_camera_ray(self._h, s, t, _p(o), _p(d))
        return o, d

    def bsdf_eval(self, material_index, n, uv, wo, wi, seed=0) -> dict:
        out = np.zeros(12, np.float32)
        a = [np.ascontiguousarray(v, dtype=np.float32) for v in (n, uv, wo, wi)]
        lib().oracle_bsdf_eval(self._h, material_index, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), seed, _p(out))
        return {"f": out[0:3].copy(), "pdf": float(out[3]), "s_wi": out[4:7].copy(),
                "s_f": out[7:10].copy(), "s_pdf": float(out[10]), "len": int(out[11])}

    def medium_eval(self, medium_index, rd, t_max, wo, wi, seeds) -> np.ndarray:
        """(n, 16) float32, layout of rene_medium_eval (include/rene_hip.h)."""
        a = [np.ascontiguousarray(v, dtype=np.float32) for v in (rd, t_max, wo, wi)]
        sd = np.ascontiguousarray(seeds, dtype=np.uint32)
        out = np.zeros((sd.size, 16), np.float32)
        lib().oracle_medium_eval(self._h, medium_index, sd.size, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(sd), _p(out))
        return out

    def tex_color(self, tex: int, u: float, v: float) -> np.ndarray:
        out = np.zeros(3, np.float32)
        lib().oracle_tex_color(self._h, tex, u, v, _p(out))
        return out

    def info(self) -> dict:
        out = np.zeros(4, np.uint32)
        lib().oracle_scene_info(self._h, _p(out))
        return {"emit_object_len": int(out[0]), "lights_len": int(out[1]),
                "triangles": int(out[2]), "instances": int(out[3])}


HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32),
                      ("instance", np.uint32), ("primitive", np.uint32)])


def pcg_u32(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.uint32)
    lib().oracle_pcg_u32(seed, n, _p(out))
    return out


def pcg_f32(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, np.float32)
    lib().oracle_pcg_f32(seed, n, _p(out))
    return out


def pcg_state_after_new(seed: int) -> int:
    s = C.c_uint32()
    lib().oracle_pcg_state_after_new(seed, C.byref(s))
    return s.value


def to_rgb8(sums: np.ndarray, n_samples: int) -> np.ndarray:
    """average (rene/src/main.rs:1758-1764) + gamma_correct (1768-1774) + to_rgb8 (1785-1792),
    restated in numpy float32."""
    v = (sums.astype(np.float32) / np.float32(n_samples)).astype(np.float32)
    lo = np.float32(12.92) * v
    with np.errstate(invalid="ignore"):
        hi = np.float32(1.055) * np.power(v, np.float32(1.0 / 2.4), dtype=np.float32) - np.float32(0.055)
    g = np.where(v <= np.float32(0.0031308), lo, hi).astype(np.float32)
    r = np.round(np.float32(255.0) * g)  # Rust f32::round: half away from zero
    r = np.where(np.abs(np.float32(255.0) * g - np.trunc(np.float32(255.0) * g)) == 0.5,
                 np.trunc(np.float32(255.0) * g) + np.sign(g), r)
    r = np.nan_to_num(np.clip(r, 0.0, 255.0), nan=0.0)
    return r.astype(np.uint8)
