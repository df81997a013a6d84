"""Python host binding of librene_hip.so (the C ABI of include/rene_hip.h).

The render path has NO CPU fallback: if the HIP library is missing or no GPU is visible, every
entry point raises.  (The CPU restatement under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# RENE_HIP_LIB names another build of the same library in csrc/ (A/B measurements of kernel variants, Makefile `variant`)
LIB_PATH = os.path.join(_CSRC, os.environ.get("RENE_HIP_LIB", "librene_hip.so"))
_LIB = None


class ReneError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{abi.STATUS_NAMES.get(code, code)}: {message}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", _CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ReneError(-3, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback for the render path)")
    # PyTorch bundles its own libamdhip64.so.7; whichever copy is loaded first serves the whole
    # process.  Load torch's first (when torch is installed) so that device pointers and streams can
    # be shared with torch tensors / torch.distributed (bench.py, multi-GPU reduce).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.rene_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.Opts), C.POINTER(vp)]
    L.rene_render.argtypes = [vp, u32, u32]
    L.rene_sync.argtypes = [vp]
    L.rene_download.argtypes = [vp, i32, i32, vp, C.c_size_t]
    L.rene_reset.argtypes = [vp]
    L.rene_tune.argtypes = [vp, C.c_uint32]
    L.rene_framebuffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.rene_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
    L.rene_trace.argtypes = [vp, i32, C.c_size_t, vp, vp, C.c_float, C.c_float, vp]
    L.rene_bsdf_eval.argtypes = [vp, u32, C.c_size_t, vp, vp, vp, vp, vp, vp]
    L.rene_medium_eval.argtypes = [vp, u32, C.c_size_t, vp, vp, vp, vp, vp, vp]
    L.rene_emitter_pdf.argtypes = [vp, C.c_size_t, vp, vp, vp]
    L.rene_pcg_probe.argtypes = [i32, u32, u32, vp]
    L.rene_ray_dump.argtypes = [vp, u32, u32, C.c_size_t, vp, C.POINTER(C.c_uint64)]
    L.rene_trace_queue.argtypes = [vp, C.c_size_t, vp, vp, i32, u32, u32, u32, u32, vp, C.POINTER(C.c_float), vp]
    L.rene_comm_unique_id.argtypes = [vp]
    L.rene_comm_init.argtypes = [vp, i32, i32, vp]
    L.rene_comm_init_all.argtypes = [C.POINTER(vp), i32]
    L.rene_reduce.argtypes = [vp, i32]
    L.rene_gather_tiles.argtypes = [vp, i32]
    L.rene_destroy.argtypes = [vp]
    L.rene_scene_pack_info.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.PackInfo)]
    L.rene_destroy.restype = None
    L.rene_last_error.restype = C.c_char_p
    L.rene_abi_version.restype = u32
    L.rene_to_rgb8.argtypes = [vp, C.c_size_t, u32, vp]
    L.rene_to_rgb8.restype = None
    L.rene_to_aov8.argtypes = [vp, C.c_size_t, u32, i32, vp]
    L.rene_to_aov8.restype = None
    L.rene_frame_seeds.argtypes = [u32, u32, u32, vp]
    L.rene_frame_seeds.restype = None
    L.rene_scene_load_pbrt.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.rene_scene_parse_pbrt.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.rene_scene_get_desc.argtypes = [vp]
    L.rene_scene_get_desc.restype = C.POINTER(abi.SceneDesc)
    L.rene_scene_film_filename.argtypes = [vp]
    L.rene_scene_film_filename.restype = C.c_char_p
    L.rene_scene_free.argtypes = [vp]
    L.rene_scene_free.restype = None
    if L.rene_abi_version() != abi.ABI_VERSION:
        raise ReneError(-1, "librene_hip.so ABI version differs from rene_amd.abi")
    _LIB = L
    return L


def _check(rc: int):
    if rc != 0:
        raise ReneError(rc, lib().rene_last_error().decode(errors="replace"))


HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32),
                      ("instance", np.uint32), ("primitive", np.uint32)])


class Renderer:
    """One render context on one GPU: upload once, render frame ranges, read the 3 layers back
    (rene/src/main.rs:513, 1315-1397, 1453-1623)."""

    def __init__(self, scene, seed: int = abi.DEFAULT_SEED, device: int = 0, flags: int = 0,
                 shard_mode: int = abi.SHARD_TILES, shard_rank: int = 0, shard_count: int = 1,
                 framebuffer_ptr: int | None = None, stream_ptr: int | None = None):
        self._h = C.c_void_p()
        packed = scene if hasattr(scene, "byref") else scene.to_desc()
        self._packed = packed
        o = abi.Opts()
        o.struct_size = C.sizeof(abi.Opts)
        o.seed, o.device, o.flags = seed & 0xFFFFFFFF, device, flags
        self._flags = flags
        o.shard_mode, o.shard_rank, o.shard_count = shard_mode, shard_rank, shard_count
        o.framebuffer = framebuffer_ptr
        o.stream = stream_ptr
        _check(lib().rene_create(packed.byref(), C.byref(o), C.byref(self._h)))
        self.xres, self.yres = packed.xres, packed.yres

    def close(self):
        if getattr(self, "_h", None):
            lib().rene_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, first_frame: int, n_frames: int):
        _check(lib().rene_render(self._h, first_frame, n_frames))

    def sync(self):
        _check(lib().rene_sync(self._h))

    def reset(self):
        _check(lib().rene_reset(self._h))

    def tune(self, n_frames: int):
        """rene_tune: pick the work-item granularity for launches of n_frames frames; resets the image."""
        _check(lib().rene_tune(self._h, n_frames))

    def download(self, layer: int = abi.LAYER_RADIANCE, channels: int = 3) -> np.ndarray:
        out = np.empty((self.yres, self.xres, channels), dtype=np.float32)
        _check(lib().rene_download(self._h, layer, channels, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def stats(self) -> abi.Stats:
        st = abi.Stats()
        _check(lib().rene_get_stats(self._h, C.byref(st)))
        return st

    def framebuffer(self) -> tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t()
        _check(lib().rene_framebuffer(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def trace(self, origins, directions, tmin: float = 0.001, tmax: float = 1e5, which: int = 0) -> np.ndarray:
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        if o.shape != d.shape:
            raise ValueError("origins and directions differ in shape")
        out = np.zeros(o.shape[0], dtype=HIT_DTYPE)
        _check(lib().rene_trace(self._h, which, o.shape[0], o.ctypes.data_as(C.c_void_p),
                                d.ctypes.data_as(C.c_void_p), tmin, tmax, out.ctypes.data_as(C.c_void_p)))
        return out


def _emitter_pdf(self, origins, directions) -> np.ndarray:
    """Device emitter-pdf probe (lib.rs:301-318 + 959-1066): pdf_l of each ray against the emitter-only structure."""
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(o.shape[0], np.float32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    _check(lib().rene_emitter_pdf(self._h, o.shape[0], p(o), p(d), p(out)))
    return out


Renderer.emitter_pdf = _emitter_pdf


def _ray_dump(self, first_frame: int, n_frames: int, capacity: int):
    """(rays [n][8] f32 -- o.xyz, tmax, d.xyz, meta bits --, queries issued) of the frames rendered: rene_ray_dump (the J1 gate's input)."""
    out = np.zeros((capacity, 8), np.float32)
    n = C.c_uint64()
    _check(lib().rene_ray_dump(self._h, first_frame, n_frames, capacity, out.ctypes.data_as(C.c_void_p), C.byref(n)))
    return out[:min(capacity, n.value)], n.value


def _trace_queue(self, o_tmax, d_flags, fp16: bool, refill_min: int = 16, leaf_min: int = 6, blocks_per_cu: int = 0, repeats: int = 3,
                 want_hits: bool = False, want_steps: bool = True):
    """rene_trace_queue: (milliseconds of the fastest launch, hits [n][4] or None, steps5 or None)."""
    o = np.ascontiguousarray(o_tmax, np.float32)
    d = np.ascontiguousarray(d_flags)
    n = o.shape[0]
    hits = np.zeros((n, 4), np.float32) if want_hits else None
    steps = np.zeros(5, np.uint64) if want_steps else None
    ms = C.c_float()
    p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
    _check(lib().rene_trace_queue(self._h, n, p(o), p(d), int(fp16), refill_min, leaf_min, blocks_per_cu, repeats, p(hits), C.byref(ms), p(steps)))
    return ms.value, hits, steps


Renderer.ray_dump = _ray_dump
Renderer.trace_queue = _trace_queue


def _comm_init(self, n_ranks: int, rank: int, unique_id: bytes):
    """Join an RCCL communicator (one context per GPU); `unique_id` = comm_unique_id() of rank 0, handed over by the host."""
    buf = (C.c_uint8 * abi.COMM_ID_BYTES).from_buffer_copy(unique_id)
    _check(lib().rene_comm_init(self._h, n_ranks, rank, buf))


def _reduce(self, root: int = 0):
    _check(lib().rene_reduce(self._h, root))


def _gather_tiles(self, root: int = 0):
    _check(lib().rene_gather_tiles(self._h, root))


Renderer.comm_init = _comm_init
Renderer.reduce = _reduce
Renderer.gather_tiles = _gather_tiles


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * abi.COMM_ID_BYTES)()
    _check(lib().rene_comm_unique_id(buf))
    return bytes(buf)


def pcg_probe(seed: int, n: int, device: int = 0) -> np.ndarray:
    """n outputs of PCG32si::new(seed) computed on the device (rene_pcg_probe)."""
    out = np.zeros(n, np.uint32)
    _check(lib().rene_pcg_probe(device, seed & 0xFFFFFFFF, n, out.ctypes.data_as(C.c_void_p)))
    return out


def _bsdf_eval(self, material_index: int, normals, uvs, wo, wi, seeds) -> np.ndarray:
    """Device BSDF probe: returns (n, 12) = f.rgb, pdf, s_wi.xyz, s_f.rgb, s_pdf, lobe count."""
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (normals, uvs, wo, wi)]
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    n = sd.size
    out = np.zeros((n, 12), np.float32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    _check(lib().rene_bsdf_eval(self._h, material_index, n, p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(sd), p(out)))
    return out


Renderer.bsdf_eval = _bsdf_eval


def _medium_eval(self, medium_index: int, rd, t_max, wo, wi, seeds) -> np.ndarray:
    """Device medium probe (volpath scenes): (n, 16), layout of rene_medium_eval in include/rene_hip.h."""
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (rd, t_max, wo, wi)]
    sd = np.ascontiguousarray(seeds, dtype=np.uint32)
    out = np.zeros((sd.size, 16), np.float32)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    _check(lib().rene_medium_eval(self._h, medium_index, sd.size, p(a[0]), p(a[1]), p(a[2]), p(a[3]), p(sd), p(out)))
    return out


Renderer.medium_eval = _medium_eval


def pack_info(scene) -> abi.PackInfo:
    """Host-only validation + flattening + BVH build (no GPU needed)."""
    packed = scene if hasattr(scene, "byref") else scene.to_desc()
    info = abi.PackInfo()
    _check(lib().rene_scene_pack_info(packed.byref(), C.byref(info)))
    return info


def to_rgb8(sums: np.ndarray, n_samples: int) -> np.ndarray:
    """average + gamma + quantise (rene/src/main.rs:1758-1792)."""
    a = np.ascontiguousarray(sums, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint8)
    lib().rene_to_rgb8(a.ctypes.data_as(C.c_void_p), a.size, n_samples, out.ctypes.data_as(C.c_void_p))
    return out


def to_aov8(sums: np.ndarray, n_samples: int, is_normal: bool) -> np.ndarray:
    a = np.ascontiguousarray(sums, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint8)
    lib().rene_to_aov8(a.ctypes.data_as(C.c_void_p), a.size, n_samples, int(is_normal), out.ctypes.data_as(C.c_void_p))
    return out


def frame_seeds(master_seed: int, first_frame: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    lib().rene_frame_seeds(master_seed & 0xFFFFFFFF, first_frame, n, out.ctypes.data_as(C.c_void_p))
    return out
