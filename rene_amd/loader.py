"""Python face of the C++ pbrt-v3 loader (rene_amd/csrc/pbrt_loader.cpp): `load_pbrt(path)` is
`expand_include + parse_pbrt + Scene::create` of the reference (rene/src/main.rs:107-205)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi, api


class LoadedScene:
    """Owns a `rene_scene*`; usable wherever a packed scene is (Renderer, pack_info, Oracle)."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle
        self._desc_ptr = api.lib().rene_scene_get_desc(handle)
        self.desc = self._desc_ptr.contents
        self.xres, self.yres = self.desc.xresolution, self.desc.yresolution
        self.film_filename = api.lib().rene_scene_film_filename(handle).decode()

    def byref(self):
        return self._desc_ptr

    def close(self):
        if self._h:
            api.lib().rene_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- table views for tests / tools ----
    def tables(self) -> dict:
        return desc_tables(self.desc)


def _arr(ptr, n, ctype):
    if n == 0:
        return np.zeros((0, C.sizeof(ctype)), np.uint8)
    buf = C.string_at(ptr, n * C.sizeof(ctype))
    return np.frombuffer(buf, dtype=np.uint8).reshape(n, C.sizeof(ctype)).copy()


def desc_tables(d: abi.SceneDesc) -> dict:
    """Raw bytes of every table (row per element) + decoded meshes, for equality checks."""
    meshes = []
    for i in range(d.n_meshes):
        m = d.meshes[i]
        v = np.frombuffer(C.string_at(m.vertices, m.n_vertices * 32), dtype=np.float32).reshape(-1, 8).copy()
        ix = np.frombuffer(C.string_at(m.indices, m.n_indices * 4), dtype=np.uint32).copy()
        meshes.append((v, ix))
    uni = np.frombuffer(C.string_at(C.addressof(d.uniform), C.sizeof(abi.Uniform)), dtype=np.float32).copy()
    return {
        "integrator": d.integrator, "res": (d.xresolution, d.yresolution), "uniform": uni,
        "instances": _arr(d.instances, d.n_instances, abi.Instance),
        "materials": _arr(d.materials, d.n_materials, abi.Material),
        "textures": _arr(d.textures, d.n_textures, abi.Texture),
        "area_lights": _arr(d.area_lights, d.n_area_lights, abi.AreaLight),
        "lights": _arr(d.lights, d.n_lights, abi.Light),
        "meshes": meshes, "n_images": d.n_images,
    }


def load_pbrt(path: str) -> LoadedScene:
    h = C.c_void_p()
    api._check(api.lib().rene_scene_load_pbrt(str(path).encode(), C.byref(h)))
    return LoadedScene(h)


def parse_pbrt(text: str, base_dir: str = "") -> LoadedScene:
    h = C.c_void_p()
    api._check(api.lib().rene_scene_parse_pbrt(text.encode(), base_dir.encode(), C.byref(h)))
    return LoadedScene(h)


def scene_to_pbrt(scene) -> str:
    """Serialise a rene_amd.scene.Scene built from named Matte materials + triangle meshes / spheres
    back to pbrt text (used to round-trip the synthetic scenes through the loader in tests)."""
    from . import glam
    w2c = glam.to_cols(scene._world_to_camera)
    out = ['Integrator "volpath"' if scene.integrator == abi.INTEGRATOR_VOLPATH else 'Integrator "path"',
           "Transform [ " + " ".join(repr(float(v)) for v in w2c) + " ]",
           f'Film "image" "integer xresolution" [ {scene.film.xresolution} ] "integer yresolution" '
           f'[ {scene.film.yresolution} ] "string filename" [ "{scene.film.filename}" ]',
           f'Camera "perspective" "float fov" [ {scene._fov_deg!r} ]', "WorldBegin"]
    fl = lambda a: " ".join(repr(float(x)) for x in np.asarray(a).reshape(-1))
    for i, m in enumerate(scene.materials[1:], start=1):
        if m.type != abi.MATERIAL_MATTE:
            raise ValueError("scene_to_pbrt only serialises Matte materials")
        c = scene.textures[m.u0[0]].v0
        out.append(f'\tMakeNamedMaterial "m{i}" "string type" [ "matte" ] "rgb Kd" [ {fl(c[:3])} ]')
    for i, m in enumerate(scene.mediums[1:], start=1):
        out.append(f'\tMakeNamedMedium "med{i}" "string type" [ "homogeneous" ] "rgb sigma_a" [ {fl(m.v0[:3])} ] '
                   f'"rgb sigma_s" [ {fl(m.v1[:3])} ] "float g" [ {float(m.v0[3])!r} ]')
    med = lambda k: f"med{k}" if k else ""
    for inst in scene.instances:
        lines = []
        scoped = bool(inst.area_light_index)
        m = [float(v) for v in inst.matrix]  # Affine3A: x_axis, y_axis, z_axis, translation
        if m != [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]:
            cols = m[0:3] + [0.0] + m[3:6] + [0.0] + m[6:9] + [0.0] + m[9:12] + [1.0]
            lines.append("ConcatTransform [ " + " ".join(repr(v) for v in cols) + " ]")
            scoped = True
        if inst.interior_medium_index or inst.exterior_medium_index:
            lines.append(f'MediumInterface "{med(inst.interior_medium_index)}" "{med(inst.exterior_medium_index)}"')
            scoped = True
        if inst.area_light_index:
            L = scene.area_lights[inst.area_light_index].v0
            lines.append(f'AreaLightSource "diffuse" "rgb L" [ {fl(L[:3])} ]')
        if inst.material_index:
            lines.append(f'NamedMaterial "m{inst.material_index}"')
        else:  # the None material (a pure medium boundary), scene.rs:109
            lines.append('Material "none"')
            scoped = True
        if inst.shape == abi.SHAPE_TRIANGLE:
            mesh = scene.meshes[inst.mesh_index]
            v = mesh.vertices
            lines.append('Shape "trianglemesh" "integer indices" [ ' + " ".join(str(int(k)) for k in mesh.indices) +
                         f' ] "point P" [ {fl(v[:, 0:3])} ] "normal N" [ {fl(v[:, 3:6])} ] "float uv" [ {fl(v[:, 6:8])} ]')
        else:
            raise ValueError("scene_to_pbrt only serialises triangle meshes")
        if scoped:
            out.append("\tAttributeBegin")
            out += ["\t\t" + l for l in lines]
            out.append("\tAttributeEnd")
        else:
            out += ["\t" + l for l in lines]
    out.append("WorldEnd")
    return "\n".join(out) + "\n"
