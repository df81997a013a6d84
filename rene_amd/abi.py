"""ctypes mirror of include/rene_hip.h (one class per C struct, same field order).

This is the binding a Python host uses; INTEGRATION.md shows the equivalent Rust `#[repr(C)]`
declarations.  tests/test_abi.py checks every struct's size against the compiled header.
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 5
COMM_ID_BYTES = 128  # RENE_COMM_ID_BYTES (an ncclUniqueId)
DEFAULT_SEED = 0x52454E45
TILE_SIZE = 32

# enum mirrors (values = the reference's #[repr(u32)] discriminants, see the header)
SHAPE_TRIANGLE, SHAPE_SPHERE = 0, 1
(MATERIAL_NONE, MATERIAL_MATTE, MATERIAL_GLASS, MATERIAL_SUBSTRATE, MATERIAL_METAL,
 MATERIAL_MIRROR, MATERIAL_UBER, MATERIAL_PLASTIC) = range(8)
TEXTURE_SOLID, TEXTURE_CHECKERBOARD, TEXTURE_IMAGEMAP, TEXTURE_SCALE = range(4)
AREA_LIGHT_NULL, AREA_LIGHT_DIFFUSE = 0, 1
LIGHT_DISTANT = 0
INTEGRATOR_PATH, INTEGRATOR_VOLPATH = 0, 1
LAYER_RADIANCE, LAYER_NORMAL, LAYER_ALBEDO = 0, 1, 2
FLAG_COUNTERS, FLAG_NO_AOV, FLAG_FORCE_BVH, FLAG_SINGLE_LEVEL, FLAG_NO_RESTART, FLAG_DYNAMIC_FIRST, FLAG_WAVEFRONT = 1, 2, 4, 8, 16, 32, 64
FLAG_OVERLAP = 128
FLAG_FP16_PAYLOAD = 256
FLAG_FRAME_GROUPS = 1 << 9
FEAT_SMALL = 64  # rene_pack_info.features: the scene renders through the wave-coherent item loop (include/rene_hip.h)
SHARD_TILES, SHARD_FRAMES = 0, 1

STATUS_NAMES = {
    0: "RENE_OK", -1: "RENE_ERR_INVALID_ARGUMENT", -2: "RENE_ERR_INVALID_SCENE",
    -3: "RENE_ERR_DEVICE", -4: "RENE_ERR_UNSUPPORTED", -5: "RENE_ERR_OUT_OF_MEMORY",
    -6: "RENE_ERR_IO", -7: "RENE_ERR_PARSE",
}

f32, u32, i32, u64 = C.c_float, C.c_uint32, C.c_int32, C.c_uint64


class Vertex(C.Structure):
    _fields_ = [("position", f32 * 3), ("normal", f32 * 3), ("uv", f32 * 2)]


class Mesh(C.Structure):
    _fields_ = [("vertices", C.POINTER(Vertex)), ("indices", C.POINTER(u32)),
                ("n_vertices", u32), ("n_indices", u32)]


class Instance(C.Structure):
    _fields_ = [("shape", u32), ("mesh_index", i32), ("material_index", u32),
                ("area_light_index", u32), ("interior_medium_index", u32),
                ("exterior_medium_index", u32), ("matrix", f32 * 12)]


class Material(C.Structure):
    _fields_ = [("type", u32), ("u0", u32 * 4), ("u1", u32 * 4), ("v0", f32 * 4)]


class Texture(C.Structure):
    _fields_ = [("type", u32), ("u0", u32 * 4), ("v0", f32 * 4)]


class AreaLight(C.Structure):
    _fields_ = [("type", u32), ("v0", f32 * 4)]


class Light(C.Structure):
    _fields_ = [("type", u32), ("v0", f32 * 4), ("v1", f32 * 4)]


MEDIUM_VACUUM, MEDIUM_HOMOGENEOUS = 0, 1


class Medium(C.Structure):
    _fields_ = [("type", u32), ("v0", f32 * 4), ("v1", f32 * 4)]


class Image(C.Structure):
    _fields_ = [("rgba", C.POINTER(f32)), ("width", u32), ("height", u32)]


class Uniform(C.Structure):
    _fields_ = [("camera_to_world", f32 * 16), ("background_matrix", f32 * 16),
                ("background_color", f32 * 4), ("projection_inv", f32 * 16),
                ("background_texture", u32)]


class SceneDesc(C.Structure):
    _fields_ = [("struct_size", u32), ("integrator", u32), ("xresolution", u32),
                ("yresolution", u32), ("uniform", Uniform),
                ("n_instances", u32), ("n_meshes", u32), ("n_materials", u32),
                ("n_textures", u32), ("n_area_lights", u32), ("n_lights", u32),
                ("n_images", u32),
                ("instances", C.POINTER(Instance)), ("meshes", C.POINTER(Mesh)),
                ("materials", C.POINTER(Material)), ("textures", C.POINTER(Texture)),
                ("area_lights", C.POINTER(AreaLight)), ("lights", C.POINTER(Light)),
                ("images", C.POINTER(Image)), ("mediums", C.POINTER(Medium)), ("n_mediums", u32),
                ("reserved", u32)]


class Opts(C.Structure):
    _fields_ = [("struct_size", u32), ("seed", u32), ("device", i32), ("flags", u32),
                ("shard_mode", u32), ("shard_rank", u32), ("shard_count", u32),
                ("reserved", u32), ("framebuffer", C.c_void_p), ("stream", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("rays_closest", u64), ("rays_shadow", u64), ("rays_emitter", u64),
                ("paths", u64), ("bounces", u64), ("hits", u64), ("adds", u64),
                ("node_visits", u64), ("prim_tests", u64), ("frames", u64), ("launches", u64),
                ("kernel_ms", C.c_double), ("last_launch_ms", C.c_double), ("sclk_mhz", C.c_double)]

    @property
    def rays(self) -> int:
        return self.rays_closest + self.rays_shadow + self.rays_emitter

    def as_dict(self) -> dict:
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["rays"] = self.rays
        return d


class PackInfo(C.Structure):
    _fields_ = [("n_instances", u32), ("n_triangles", u32), ("n_spheres", u32),
                ("n_nodes_main", u32), ("n_slots_main", u32), ("depth_main", u32),
                ("n_nodes_emit", u32), ("n_slots_emit", u32), ("depth_emit", u32),
                ("features", u32), ("emit_object_len", u32), ("lights_len", u32),
                ("device_bytes", u64), ("n_items_main", u32), ("n_items_emit", u32)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class Hit(C.Structure):
    _fields_ = [("t", f32), ("u", f32), ("v", f32), ("instance", u32), ("primitive", u32)]


def algorithmic_bytes(stats) -> int:
    """cache-less traffic model,
    B_alg = 64 N_ray + 64 N_node + 48 N_tri + 144 N_hit + 128 N_bounce + 32 N_add."""
    g = (lambda k: stats[k]) if isinstance(stats, dict) else (lambda k: getattr(stats, k))
    n_ray = g("rays_closest") + g("rays_shadow") + g("rays_emitter")
    return (64 * n_ray + 64 * g("node_visits") + 48 * g("prim_tests") + 144 * g("hits")
            + 128 * g("bounces") + 32 * g("adds"))


# every symbol include/rene_hip.h declares (tests check that the shared library exports them all)
EXPORTED_SYMBOLS = [
    "rene_create", "rene_render", "rene_sync", "rene_download", "rene_reset", "rene_tune", "rene_framebuffer",
    "rene_get_stats", "rene_trace", "rene_ray_dump", "rene_trace_queue", "rene_bsdf_eval", "rene_medium_eval", "rene_emitter_pdf", "rene_pcg_probe",
    "rene_comm_unique_id", "rene_comm_init", "rene_comm_init_all", "rene_comm_group_begin", "rene_comm_group_end",
    "rene_reduce", "rene_gather_tiles", "rene_destroy", "rene_scene_pack_info", "rene_last_error", "rene_abi_version",
    "rene_to_rgb8", "rene_to_aov8", "rene_frame_seeds",
    "rene_scene_load_pbrt", "rene_scene_parse_pbrt", "rene_scene_get_desc",
    "rene_scene_film_filename", "rene_scene_free",
]
