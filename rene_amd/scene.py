"""Host-side mirror of rene's flat `Scene` (rene/src/scene.rs:36-49) and of the rene-shader
host constructors (`EnumMaterial::new_*`, `EnumTexture::new_*`, ...), producing the
`rene_scene_desc` the C ABI consumes.

The method names, argument order and table layouts follow the reference so that code building a
scene reads like rene/src/scene.rs:170-460.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi, glam

F32 = np.float32


@dataclass
class TriangleMesh:
    """TriangleMesh, rene/src/scene/intermediate_scene.rs:149-153. vertices: (n, 8) f32 rows of
    position(3) normal(3) uv(2); indices: (3m,) u32, mesh-local."""
    vertices: np.ndarray
    indices: np.ndarray

    @staticmethod
    def from_arrays(positions, indices, normals=None, uvs=None) -> "TriangleMesh":
        p = np.asarray(positions, dtype=F32).reshape(-1, 3)
        v = np.zeros((p.shape[0], 8), dtype=F32)
        v[:, 0:3] = p
        if normals is not None:  # absent normals stay exactly zero -> geometric normal, lib.rs:931-935
            v[:, 3:6] = np.asarray(normals, dtype=F32).reshape(-1, 3)
        if uvs is not None:
            v[:, 6:8] = np.asarray(uvs, dtype=F32).reshape(-1, 2)
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.uint32).reshape(-1))
        if idx.size % 3:
            raise ValueError("indices length must be a multiple of 3")  # intermediate_scene.rs:940-944
        if idx.size and idx.max() >= p.shape[0]:
            raise ValueError("index out of range")
        return TriangleMesh(np.ascontiguousarray(v), idx)


@dataclass
class Film:
    """Film defaults, intermediate_scene.rs:162-170."""
    filename: str = "out.png"
    xresolution: int = 640
    yresolution: int = 480


@dataclass
class Scene:
    integrator: int = abi.INTEGRATOR_PATH
    film: Film = field(default_factory=Film)
    camera_to_world: np.ndarray = field(default_factory=glam.identity)
    projection_inv: np.ndarray = field(default_factory=glam.identity)
    background_matrix: np.ndarray = field(default_factory=glam.identity)
    background_color: tuple = (0.0, 0.0, 0.0)
    background_texture: int = 0
    instances: list = field(default_factory=list)   # abi.Instance
    meshes: list = field(default_factory=list)      # TriangleMesh
    materials: list = field(default_factory=list)   # abi.Material
    textures: list = field(default_factory=list)    # abi.Texture
    area_lights: list = field(default_factory=list)  # abi.AreaLight
    lights: list = field(default_factory=list)      # abi.Light
    images: list = field(default_factory=list)      # (h, w, 4) f32 arrays
    mediums: list = field(default_factory=list)     # abi.Medium

    # ---- Scene::create prologue, scene.rs:104-116 ------------------------------------------------
    @staticmethod
    def new() -> "Scene":
        s = Scene()
        s.materials.append(abi.Material(type=abi.MATERIAL_NONE))
        s.area_lights.append(abi.AreaLight(type=abi.AREA_LIGHT_NULL))
        s.mediums.append(abi.Medium(type=abi.MEDIUM_VACUUM))  # scene.rs:111
        s.add_texture_solid((1.0, 1.0, 1.0))  # default infinite-light texture
        return s

    # ---- textures (texture.rs:139-173) -----------------------------------------------------------
    def _push_texture(self, t: abi.Texture) -> int:
        self.textures.append(t)
        return len(self.textures) - 1

    def add_texture_solid(self, color) -> int:
        t = abi.Texture(type=abi.TEXTURE_SOLID)
        t.v0[:] = [color[0], color[1], color[2], 0.0]
        return self._push_texture(t)

    def add_texture_checkerboard(self, tex1: int, tex2: int, uscale: float, vscale: float) -> int:
        t = abi.Texture(type=abi.TEXTURE_CHECKERBOARD)
        t.u0[:] = [tex1, tex2, 0, 0]
        t.v0[:] = [uscale, vscale, 0.0, 0.0]
        return self._push_texture(t)

    def add_texture_image_map(self, image: np.ndarray) -> int:
        image = np.asarray(image, dtype=F32)
        if image.ndim != 3 or image.shape[2] not in (3, 4):
            raise ValueError(f"an image is (h, w, 4) RGBA (or (h, w, 3), padded here): got {image.shape}")
        if image.shape[2] == 3:  # (the C ABI's rene_image is RGBA: four floats per texel)
            image = np.concatenate([image, np.ones(image.shape[:2] + (1,), dtype=F32)], axis=2)
        self.images.append(np.ascontiguousarray(image, dtype=F32))
        t = abi.Texture(type=abi.TEXTURE_IMAGEMAP)
        t.u0[:] = [len(self.images) - 1, 0, 0, 0]
        return self._push_texture(t)

    def add_texture_scale(self, tex1: int, tex2: int) -> int:
        t = abi.Texture(type=abi.TEXTURE_SCALE)
        t.u0[:] = [tex1, tex2, 0, 0]
        return self._push_texture(t)

    def _tex(self, v) -> int:
        """TextureOrColor -> index (Scene::texture, scene.rs:81-98): colours become new solid
        textures, ints are existing texture indices."""
        if isinstance(v, (int, np.integer)):
            return int(v)
        if isinstance(v, (float, np.floating)):
            v = (v, v, v)
        return self.add_texture_solid(v)

    # ---- materials (material.rs:385-493; Scene::material, scene.rs:170-257) ---------------------
    def _push_material(self, m: abi.Material) -> int:
        self.materials.append(m)
        return len(self.materials) - 1

    def add_matte(self, kd=(0.5, 0.5, 0.5)) -> int:
        m = abi.Material(type=abi.MATERIAL_MATTE)
        m.u0[:] = [self._tex(kd), 0, 0, 0]
        return self._push_material(m)

    def add_glass(self, index: float = 1.5) -> int:
        m = abi.Material(type=abi.MATERIAL_GLASS)
        m.v0[:] = [index, 0.0, 0.0, 0.0]
        return self._push_material(m)

    def add_substrate(self, kd=(0.5, 0.5, 0.5), ks=(0.5, 0.5, 0.5), rough_u=0.0, rough_v=0.0,
                      remap_roughness: bool = True) -> int:
        m = abi.Material(type=abi.MATERIAL_SUBSTRATE)
        m.u0[:] = [self._tex(kd), self._tex(ks), self._tex(rough_u), self._tex(rough_v)]
        m.u1[:] = [1 if remap_roughness else 0, 0, 0, 0]
        return self._push_material(m)

    def add_metal(self, eta=(0.19999069, 0.9220846, 1.0998759), k=(3.9046354, 2.4476333, 2.1376526),
                  rough_u=0.01, rough_v=0.01, remap_roughness: bool = True) -> int:
        m = abi.Material(type=abi.MATERIAL_METAL)
        m.u0[:] = [self._tex(eta), self._tex(k), self._tex(rough_u), self._tex(rough_v)]
        m.u1[:] = [1 if remap_roughness else 0, 0, 0, 0]
        return self._push_material(m)

    def add_mirror(self, r=(0.9, 0.9, 0.9)) -> int:
        m = abi.Material(type=abi.MATERIAL_MIRROR)
        m.u0[:] = [self._tex(r), 0, 0, 0]
        return self._push_material(m)

    def add_uber(self, kd=(0.25, 0.25, 0.25), ks=(0.25, 0.25, 0.25), kr=(0.0, 0.0, 0.0),
                 kt=(0.0, 0.0, 0.0), rough_u=0.1, rough_v=0.1, eta: float = 1.5,
                 opacity=(1.0, 1.0, 1.0), remap_roughness: bool = True) -> int:
        # texture creation order follows scene.rs:232-241
        i_kd, i_ks, i_kr, i_kt = self._tex(kd), self._tex(ks), self._tex(kr), self._tex(kt)
        i_ru, i_rv, i_op = self._tex(rough_u), self._tex(rough_v), self._tex(opacity)
        m = abi.Material(type=abi.MATERIAL_UBER)
        m.u0[:] = [i_kd, i_ks, i_kr, i_kt]
        m.u1[:] = [i_op, 1 if remap_roughness else 0, i_ru, i_rv]
        m.v0[:] = [eta, 0.0, 0.0, 0.0]
        return self._push_material(m)

    def add_plastic(self, kd=(0.25, 0.25, 0.25), ks=(0.25, 0.25, 0.25), roughness=0.1,
                    remap_roughness: bool = True) -> int:
        m = abi.Material(type=abi.MATERIAL_PLASTIC)
        m.u0[:] = [self._tex(kd), self._tex(ks), 1 if remap_roughness else 0, self._tex(roughness)]
        return self._push_material(m)

    # ---- lights ------------------------------------------------------------------------------------
    def add_area_light_diffuse(self, L) -> int:
        a = abi.AreaLight(type=abi.AREA_LIGHT_DIFFUSE)
        a.v0[:] = [L[0], L[1], L[2], 0.0]
        self.area_lights.append(a)
        return len(self.area_lights) - 1

    def add_light_distant(self, frm, to, L) -> int:
        """EnumLight::new_distant, light.rs:43-50."""
        d = np.asarray(frm, dtype=F32) - np.asarray(to, dtype=F32)
        d = (d / F32(math.sqrt(float(np.dot(d.astype(np.float64), d.astype(np.float64)))))).astype(F32)
        l = abi.Light(type=abi.LIGHT_DISTANT)
        l.v0[:] = [d[0], d[1], d[2], 0.0]
        l.v1[:] = [L[0], L[1], L[2], 0.0]
        self.lights.append(l)
        return len(self.lights) - 1

    def set_infinite_light(self, color, image: np.ndarray | None = None, ctm=None):
        """LightSource "infinite", scene.rs:367-383."""
        self.background_color = tuple(float(c) for c in color)
        if image is not None:
            self.background_texture = self.add_texture_image_map(image)
            self.background_matrix = glam.inverse(glam.identity() if ctm is None else ctm)

    def add_medium_homogeneous(self, sigma_a, sigma_s, g: float = 0.0) -> int:
        """EnumMedium::new_homogeneous, medium.rs:168-173 (MakeNamedMedium, scene.rs:405-416)."""
        m = abi.Medium(type=abi.MEDIUM_HOMOGENEOUS)
        m.v0[:] = [sigma_a[0], sigma_a[1], sigma_a[2], g]
        m.v1[:] = [sigma_s[0], sigma_s[1], sigma_s[2], 0.0]
        self.mediums.append(m)
        return len(self.mediums) - 1

    # ---- shapes (scene.rs:417-456) ---------------------------------------------------------------
    def add_triangle_mesh(self, mesh: TriangleMesh, material: int, area_light: int = 0,
                          ctm=None, interior: int = 0, exterior: int = 0) -> int:
        self.meshes.append(mesh)
        inst = abi.Instance(shape=abi.SHAPE_TRIANGLE, mesh_index=len(self.meshes) - 1,
                            material_index=material, area_light_index=area_light,
                            interior_medium_index=interior, exterior_medium_index=exterior)
        inst.matrix[:] = glam.affine_from_mat4(glam.identity() if ctm is None else ctm)
        self.instances.append(inst)
        return len(self.instances) - 1

    def add_mesh_instance(self, mesh_index: int, material: int, area_light: int = 0, ctm=None) -> int:
        inst = abi.Instance(shape=abi.SHAPE_TRIANGLE, mesh_index=mesh_index,
                            material_index=material, area_light_index=area_light)
        inst.matrix[:] = glam.affine_from_mat4(glam.identity() if ctm is None else ctm)
        self.instances.append(inst)
        return len(self.instances) - 1

    def add_sphere(self, radius: float, material: int, area_light: int = 0, ctm=None,
                   interior: int = 0, exterior: int = 0) -> int:
        m = glam.mul(glam.identity() if ctm is None else ctm, glam.from_scale((radius,) * 3))
        inst = abi.Instance(shape=abi.SHAPE_SPHERE, mesh_index=-1, material_index=material,
                            area_light_index=area_light, interior_medium_index=interior,
                            exterior_medium_index=exterior)
        inst.matrix[:] = glam.affine_from_mat4(m)
        self.instances.append(inst)
        return len(self.instances) - 1

    # ---- camera / film (scene.rs:155-165) --------------------------------------------------------
    def set_camera(self, world_to_camera: np.ndarray, fov_deg: float, xres: int, yres: int):
        self.film.xresolution, self.film.yresolution = int(xres), int(yres)
        fov = float(F32(fov_deg) * F32(math.pi) / F32(180.0))  # deg_to_radian, intermediate_scene.rs:612-614
        aspect = float(F32(xres) / F32(yres))
        if yres > xres:  # "TODO remove this ad-hoc", scene.rs:156-162
            fov = math.atan(math.tan(fov * 0.5) / xres * yres) * 2.0
        self.projection_inv = glam.inverse(glam.perspective_lh(fov, aspect, 0.01, 1000.0))
        self.camera_to_world = glam.inverse(world_to_camera)
        self._world_to_camera = np.asarray(world_to_camera, dtype=F32)
        self._fov_deg = float(fov_deg)

    def with_resolution(self, xres: int, yres: int) -> "Scene":
        """Re-derive the projection for another Film size (the --width/--height CLI override)."""
        self.set_camera(self._world_to_camera, self._fov_deg, xres, yres)
        return self

    # ---- summary -------------------------------------------------------------------------------------
    @property
    def n_triangles(self) -> int:
        return sum(int(self.meshes[i.mesh_index].indices.size // 3) for i in self.instances
                   if i.shape == abi.SHAPE_TRIANGLE)

    def to_desc(self) -> "PackedScene":
        return PackedScene(self)


class PackedScene:
    """Owns the ctypes arrays behind one `rene_scene_desc` (pointers are borrowed by the C side)."""

    def __init__(self, s: Scene):
        self._keep = []
        d = abi.SceneDesc()
        d.struct_size = C.sizeof(abi.SceneDesc)
        d.integrator = s.integrator
        d.xresolution, d.yresolution = s.film.xresolution, s.film.yresolution
        d.uniform.camera_to_world[:] = glam.to_cols(s.camera_to_world).tolist()
        d.uniform.background_matrix[:] = glam.to_cols(s.background_matrix).tolist()
        d.uniform.projection_inv[:] = glam.to_cols(s.projection_inv).tolist()
        d.uniform.background_color[:] = [*s.background_color, 0.0]
        d.uniform.background_texture = s.background_texture

        def arr(ctype, items):
            a = (ctype * max(1, len(items)))(*items)
            self._keep.append(a)
            return a

        meshes = []
        for m in s.meshes:
            v = np.ascontiguousarray(m.vertices, dtype=F32)
            i = np.ascontiguousarray(m.indices, dtype=np.uint32)
            self._keep += [v, i]
            meshes.append(abi.Mesh(v.ctypes.data_as(C.POINTER(abi.Vertex)),
                                   i.ctypes.data_as(C.POINTER(abi.u32)), v.shape[0], i.size))
        images = []
        for im in s.images:
            a = np.ascontiguousarray(im, dtype=F32)
            self._keep.append(a)
            images.append(abi.Image(a.ctypes.data_as(C.POINTER(abi.f32)), a.shape[1], a.shape[0]))
        d.n_instances, d.instances = len(s.instances), arr(abi.Instance, s.instances)
        d.n_meshes, d.meshes = len(meshes), arr(abi.Mesh, meshes)
        d.n_materials, d.materials = len(s.materials), arr(abi.Material, s.materials)
        d.n_textures, d.textures = len(s.textures), arr(abi.Texture, s.textures)
        d.n_area_lights, d.area_lights = len(s.area_lights), arr(abi.AreaLight, s.area_lights)
        d.n_lights, d.lights = len(s.lights), arr(abi.Light, s.lights)
        d.n_images, d.images = len(images), arr(abi.Image, images)
        d.n_mediums, d.mediums = len(s.mediums), arr(abi.Medium, s.mediums)
        self.desc = d
        self.xres, self.yres = d.xresolution, d.yresolution

    def byref(self):
        return C.byref(self.desc)
