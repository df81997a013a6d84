"""Synthetic benchmark scenes, built in code so that nothing needs
/root/reference at run time.

`cornell_box` is generated from the numbers of Bitterli's CC0 Cornell box as shipped in
sample_scenes/cornell-box/scene.pbrt:2-33 (camera transform, fov, wall quads, the two boxes,
the light quad and its radiance, the Kd values).  The statement order mirrors that file so the
material / texture / instance indices equal what rene's loader produces for it
(tests/test_loader.py checks that when the reference is present).
"""
from __future__ import annotations

import numpy as np

from . import abi, glam
from .scene import Scene, TriangleMesh

F32 = np.float32

_QUAD_IDX = [0, 1, 2, 0, 2, 3]
_QUAD_UV = [0, 0, 1, 0, 1, 1, 0, 1]
_BOX_IDX = [0, 2, 1, 0, 3, 2, 4, 6, 5, 4, 7, 6, 8, 10, 9, 8, 11, 10, 12, 14, 13, 12, 15, 14,
            16, 18, 17, 16, 19, 18, 20, 22, 21, 20, 23, 22]


def _quad(p, n) -> TriangleMesh:
    return TriangleMesh.from_arrays(p, _QUAD_IDX, normals=[n] * 4, uvs=_QUAD_UV)


def _box(p, face_normals) -> TriangleMesh:
    n = [fn for fn in face_normals for _ in range(4)]
    return TriangleMesh.from_arrays(p, _BOX_IDX, normals=n, uvs=_QUAD_UV * 6)


_SHORT_BOX_P = [
    -0.0460751, 0.6, 0.573007, -0.0460751, -2.98023e-08, 0.573007, 0.124253, 0, 0.00310463,
    0.124253, 0.6, 0.00310463, 0.533009, 0, 0.746079, 0.533009, 0.6, 0.746079,
    0.703337, 0.6, 0.176177, 0.703337, 2.98023e-08, 0.176177, 0.533009, 0.6, 0.746079,
    -0.0460751, 0.6, 0.573007, 0.124253, 0.6, 0.00310463, 0.703337, 0.6, 0.176177,
    0.703337, 2.98023e-08, 0.176177, 0.124253, 0, 0.00310463, -0.0460751, -2.98023e-08, 0.573007,
    0.533009, 0, 0.746079, 0.533009, 0, 0.746079, -0.0460751, -2.98023e-08, 0.573007,
    -0.0460751, 0.6, 0.573007, 0.533009, 0.6, 0.746079, 0.703337, 0.6, 0.176177,
    0.124253, 0.6, 0.00310463, 0.124253, 0, 0.00310463, 0.703337, 2.98023e-08, 0.176177]
_SHORT_BOX_N = [
    (-0.958123, -4.18809e-08, -0.286357), (0.958123, 4.18809e-08, 0.286357),
    (-4.37114e-08, 1, -1.91069e-15), (4.37114e-08, -1, 1.91069e-15),
    (-0.286357, -1.25171e-08, 0.958123), (0.286357, 1.25171e-08, -0.958123)]
_TALL_BOX_P = [
    -0.720444, 1.2, -0.473882, -0.720444, 0, -0.473882, -0.146892, 0, -0.673479,
    -0.146892, 1.2, -0.673479, -0.523986, 0, 0.0906493, -0.523986, 1.2, 0.0906492,
    0.0495656, 1.2, -0.108948, 0.0495656, 0, -0.108948, -0.523986, 1.2, 0.0906492,
    -0.720444, 1.2, -0.473882, -0.146892, 1.2, -0.673479, 0.0495656, 1.2, -0.108948,
    0.0495656, 0, -0.108948, -0.146892, 0, -0.673479, -0.720444, 0, -0.473882,
    -0.523986, 0, 0.0906493, -0.523986, 0, 0.0906493, -0.720444, 0, -0.473882,
    -0.720444, 1.2, -0.473882, -0.523986, 1.2, 0.0906492, 0.0495656, 1.2, -0.108948,
    -0.146892, 1.2, -0.673479, -0.146892, 0, -0.673479, 0.0495656, 0, -0.108948]
_TALL_BOX_N = [
    (-0.328669, -4.1283e-08, -0.944445), (0.328669, 4.1283e-08, 0.944445),
    (3.82137e-15, 1, -4.37114e-08), (-3.82137e-15, -1, 4.37114e-08),
    (-0.944445, 1.43666e-08, 0.328669), (0.944445, -1.43666e-08, -0.328669)]

CORNELL_WORLD_TO_CAMERA = [1, -0, -0, -0, -0, 1, -0, -0, -0, -0, -1, -0, -0, -1, 6.8, 1]
CORNELL_FOV_DEG = 19.5


def cornell_box(xres: int = 1024, yres: int = 1024) -> Scene:
    """36 triangles in 8 instances, one quad emitter (L = 17, 12, 4), all Matte."""
    s = Scene.new()
    s.film.filename = "cornell-box.png"
    s.set_camera(glam.from_cols_array(CORNELL_WORLD_TO_CAMERA), CORNELL_FOV_DEG, xres, yres)
    mat = {}
    for name, kd in (("LeftWall", (0.63, 0.065, 0.05)), ("RightWall", (0.14, 0.45, 0.091)),
                     ("Floor", (0.725, 0.71, 0.68)), ("Ceiling", (0.725, 0.71, 0.68)),
                     ("BackWall", (0.725, 0.71, 0.68)), ("ShortBox", (0.725, 0.71, 0.68)),
                     ("TallBox", (0.725, 0.71, 0.68)), ("Light", (0.0, 0.0, 0.0))):
        mat[name] = s.add_matte(kd)  # MakeNamedMaterial ... "matte", scene.pbrt:8-15
    s.add_triangle_mesh(_quad([-1, 1.74846e-07, -1, -1, 1.74846e-07, 1, 1, -1.74846e-07, 1,
                               1, -1.74846e-07, -1], (4.37114e-08, 1, 1.91069e-15)), mat["Floor"])
    s.add_triangle_mesh(_quad([1, 2, 1, -1, 2, 1, -1, 2, -1, 1, 2, -1],
                              (-8.74228e-08, -1, -4.37114e-08)), mat["Ceiling"])
    s.add_triangle_mesh(_quad([-1, 0, -1, -1, 2, -1, 1, 2, -1, 1, 0, -1],
                              (8.74228e-08, -4.37114e-08, -1)), mat["BackWall"])
    s.add_triangle_mesh(_quad([1, 0, -1, 1, 2, -1, 1, 2, 1, 1, 0, 1],
                              (1, -4.37114e-08, 1.31134e-07)), mat["RightWall"])
    s.add_triangle_mesh(_quad([-1, 0, 1, -1, 2, 1, -1, 2, -1, -1, 0, -1],
                              (-1, -4.37114e-08, -4.37114e-08)), mat["LeftWall"])
    s.add_triangle_mesh(_box(_SHORT_BOX_P, _SHORT_BOX_N), mat["ShortBox"])
    s.add_triangle_mesh(_box(_TALL_BOX_P, _TALL_BOX_N), mat["TallBox"])
    light = s.add_area_light_diffuse((17.0, 12.0, 4.0))  # AttributeBegin .. AttributeEnd, scene.pbrt:29-33
    s.add_triangle_mesh(_quad([-0.24, 1.98, -0.22, 0.23, 1.98, -0.22, 0.23, 1.98, 0.16,
                               -0.24, 1.98, 0.16], (-8.74228e-08, -1, 1.86006e-07)),
                        mat["Light"], area_light=light)
    return s


# ---------------------------------------------------------------------------------------------------
# volpath scenes.  The reference ships no scene that uses Integrator
# "volpath" / MakeNamedMedium, so these are parity-test inputs of this repository: the Cornell box with a
# thin fog filling the room and a dense cloud where the short box stood (nested media behind
# None-material boundaries, lib.rs:768-779), and a BVH-sized scene with every volpath branch.
# ---------------------------------------------------------------------------------------------------
def _aabb(lo, hi) -> TriangleMesh:
    """Axis-aligned box, outward vertex normals, 12 triangles (two coplanar per face)."""
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    faces = [((-1, 0, 0), [(x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)]),
             ((1, 0, 0), [(x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)]),
             ((0, -1, 0), [(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)]),
             ((0, 1, 0), [(x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)]),
             ((0, 0, -1), [(x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0)]),
             ((0, 0, 1), [(x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)])]
    P, N, I = [], [], []
    for n, q in faces:
        b = len(P)
        P += q
        N += [n] * 4
        I += [b, b + 1, b + 2, b, b + 2, b + 3]
    return TriangleMesh.from_arrays(np.asarray(P, dtype=F32).reshape(-1), I, normals=N, uvs=_QUAD_UV * 6)


def cornell_fog(xres: int = 1024, yres: int = 1024) -> Scene:
    """Cornell box under Integrator "volpath": 48 triangles; a fog volume (sigma_t = 0.45) fills the
    room below the light, a dense forward-scattering cloud (sigma_t ~ 6, g = 0.4) replaces the short
    box; both are bounded by None-material boxes (MediumInterface inside AttributeBegin/End)."""
    s = cornell_box(xres, yres)
    s.integrator = abi.INTEGRATOR_VOLPATH
    s.film.filename = "cornell-fog.png"
    short = next(i for i, m in enumerate(s.meshes) if len(m.indices) == 36)  # the short box comes first
    inst = next(i for i, it in enumerate(s.instances) if it.mesh_index == short)
    fog = s.add_medium_homogeneous((0.02, 0.02, 0.03), (0.42, 0.42, 0.44), 0.0)
    cloud = s.add_medium_homogeneous((0.3, 0.2, 0.1), (5.5, 5.8, 6.2), 0.4)
    s.instances[inst].material_index = 0  # None: a pure medium boundary
    s.instances[inst].interior_medium_index = cloud
    s.instances[inst].exterior_medium_index = fog
    # lifted 1 cm off the floor: the Cornell box's short box stands ON the floor (bottom face within 3e-8 of it),
    # and with a None material the order in which two coincident surfaces are met decides which medium the floor
    # is in -- a coin flip of rounding that no two intersection formulas resolve alike
    s.instances[inst].matrix[:] = glam.affine_from_mat4(glam.from_translation((0.0, 0.01, 0.0)))
    s.add_triangle_mesh(_aabb((-0.995, 0.005, -0.995), (0.995, 1.9, 0.995)), 0, interior=fog, exterior=0)
    return s


def media_zoo(xres: int = 96, yres: int = 64) -> Scene:
    """Every volpath branch in one BVH-sized scene: distant light through media (tr, lib.rs:359-409),
    area emitters seen from inside media (tr_emit), glass and None boundaries around media, nested
    spheres, anisotropic phase functions, textured surfaces, the infinite light."""
    s = material_zoo(xres, yres)
    s.integrator = abi.INTEGRATOR_VOLPATH
    s.film.filename = "media-zoo.png"
    milk = s.add_medium_homogeneous((0.0011, 0.0024, 0.014), (2.55, 3.21, 3.77), 0.0)  # the loader's defaults
    smoke = s.add_medium_homogeneous((0.8, 0.8, 0.8), (1.2, 1.2, 1.2), -0.3)
    haze = s.add_medium_homogeneous((0.01, 0.01, 0.01), (0.08, 0.09, 0.12), 0.7)
    glass = s.add_glass(1.33)
    s.add_sphere(0.7, glass, ctm=glam.from_translation((-1.6, 0.7, -2.2)), interior=milk)
    s.add_sphere(0.8, 0, ctm=glam.from_translation((1.4, 0.8, -2.4)), interior=smoke, exterior=haze)
    s.add_sphere(0.3, s.add_matte((0.8, 0.3, 0.2)), ctm=glam.from_translation((1.4, 0.8, -2.4)),
                 interior=smoke, exterior=smoke)
    s.add_triangle_mesh(_aabb((-5.5, 0.01, -4.5), (5.5, 3.6, 3.9)), 0, interior=haze, exterior=0)
    return s


# ---------------------------------------------------------------------------------------------------
# veach-mis (BASELINE config 3): numbers from sample_scenes/veach-mis/scene.pbrt (Bitterli, CC0)
# ---------------------------------------------------------------------------------------------------
VEACH_WORLD_TO_CAMERA = [4.37113e-08, -0, -1, -0, -0, 1, -0, -0, -1, -0, -4.37113e-08, -0, -0, -3.5, 28.2792, 1]
VEACH_FOV_DEG = 20.114292
_VEACH_ETA = (0.200438, 0.924033, 1.102212)
_VEACH_K = (3.912949, 2.452848, 2.142188)


def _plate(A, B, Cc, D, n0, n4) -> TriangleMesh:
    """A box extruded along z in [-4, 4] from the cross-section A, B, D, C; vertex order as in the file."""
    def v(p, z):
        return [p[0], p[1], z]
    P = [v(A, -4), v(A, 4), v(B, 4), v(B, -4), v(Cc, 4), v(Cc, -4), v(D, -4), v(D, 4),
         v(Cc, -4), v(A, -4), v(B, -4), v(D, -4), v(D, 4), v(B, 4), v(A, 4), v(Cc, 4),
         v(Cc, 4), v(A, 4), v(A, -4), v(Cc, -4), v(D, -4), v(B, -4), v(B, 4), v(D, 4)]
    neg = lambda n: tuple(-x for x in n)
    return _box(np.asarray(P, dtype=F32).reshape(-1), [n0, neg(n0), (0, 0, -1), (0, 0, 1), n4, neg(n4)])


def veach_mis(xres: int = 1280, yres: int = 720, small_light: bool = True) -> Scene:
    """52 triangles + 3 emissive spheres (r = 1, 0.5, 0.05), four metal plates with
    alpha = 0.01 / 0.05 / 0.1 / 0.25 (`remaproughness false`), two Matte walls.  small_light=False leaves the
    r = 0.05 emitter out (parity tests: rene's fp32 cone pdf cancels catastrophically for it, lib.rs:1058-1064)."""
    s = Scene.new()
    s.film.filename = "veach-mis.png"
    s.set_camera(glam.from_cols_array(VEACH_WORLD_TO_CAMERA), VEACH_FOV_DEG, xres, yres)
    diffuse = s.add_matte((0.5, 0.5, 0.5))
    metal = lambda a: s.add_metal(_VEACH_ETA, _VEACH_K, a, a, remap_roughness=False)
    smooth, glossy, rough = metal(0.01), metal(0.05), metal(0.1)
    null = s.add_matte((0.0, 0.0, 0.0))
    super_rough = metal(0.25)
    s.add_triangle_mesh(_plate((-0.637866, 4.65614), (0.973649, 3.30966), (-0.445511, 4.88636), (1.166, 3.53988),
                               (-0.641183, -0.767388, 0), (-0.767388, 0.641183, 0)), smooth)
    s.add_triangle_mesh(_plate((2.03286, 2.97515), (3.97697, 2.18116), (2.14629, 3.25288), (4.0904, 2.45889),
                               (-0.37809, -0.925769, 0), (-0.925769, 0.37809, 0)), glossy)
    s.add_triangle_mesh(_plate((6.04018, 1.86557), (8.10399, 1.47742), (6.09563, 2.1604), (8.15944, 1.77225),
                               (-0.184835, -0.98277, 0), (-0.98277, 0.184835, 0)), rough)
    s.add_triangle_mesh(_quad([-5, 8.65485e-07, 23.76, 14.8, 8.65485e-07, 23.76, 14.8, -8.65485e-07, -23.76,
                               -5, -8.65485e-07, -23.76], (0, 1, -2.09815e-07)), diffuse)
    s.add_triangle_mesh(_quad([-5, 19.8, 23.76, -5, 0, 23.76, -5, 0, -23.76, -5, 19.8, -23.76],
                              (1, -4.37114e-08, -2.09815e-07)), diffuse)
    for L, z, r in ((7.599088, -2.8, 1.0), (30.396353, 0.0, 0.5), (3039.635254, 2.7, 0.05)):
        if r < 0.1 and not small_light:
            continue
        al = s.add_area_light_diffuse((L, L, L))
        s.add_sphere(r, null, area_light=al, ctm=glam.from_translation((0, 6.5, z)))
    s.add_triangle_mesh(_plate((9.61645, 1.21286), (11.7008, 0.956897), (9.65301, 1.51062), (11.7374, 1.25466),
                               (-0.121887, -0.992544, 0), (-0.992544, 0.121887, 0)), super_rough)
    return s


# ---------------------------------------------------------------------------------------------------
# material zoo: every material / texture / light kind the path integrator supports, in one small
# scene (parity-test input, not a reference scene)
# ---------------------------------------------------------------------------------------------------
def material_zoo(xres: int = 96, yres: int = 64) -> Scene:
    s = Scene.new()
    s.film.filename = "zoo.png"
    s.set_camera(glam.look_at_lh((0.0, 2.2, -7.5), (0.0, 0.8, 0.0), (0.0, 1.0, 0.0)), 38.0, xres, yres)
    rng = np.random.default_rng(5)
    img = np.ones((16, 32, 4), dtype=F32)
    img[..., :3] = rng.uniform(0.1, 1.0, (16, 32, 3))
    sky = np.ones((8, 16, 4), dtype=F32)
    sky[..., 0], sky[..., 1], sky[..., 2] = 0.35, 0.45, np.linspace(0.9, 0.4, 8)[:, None]
    s.set_infinite_light((0.8, 0.8, 0.8), image=sky, ctm=glam.from_axis_angle((0.0, 1.0, 0.0), 0.6))
    s.add_light_distant((-0.3, 0.8, -0.5), (0, 0, 0), (1.5, 1.4, 1.2))
    dark, light = s.add_texture_solid((0.15, 0.15, 0.2)), s.add_texture_solid((0.8, 0.8, 0.7))
    checks = s.add_texture_checkerboard(dark, light, 8.0, 8.0)
    imap = s.add_texture_image_map(img)
    scale = s.add_texture_scale(imap, light)
    floor = TriangleMesh.from_arrays([-6, 0, -6, 6, 0, -6, 6, 0, 6, -6, 0, 6], _QUAD_IDX, uvs=_QUAD_UV)  # no normals
    s.add_triangle_mesh(floor, s.add_matte(checks))
    back = TriangleMesh.from_arrays([-6, 0, 4, 6, 0, 4, 6, 5, 4, -6, 5, 4], _QUAD_IDX, normals=[(0, 0, -1)] * 4, uvs=_QUAD_UV)
    s.add_triangle_mesh(back, s.add_matte(scale))
    mats = [s.add_glass(1.5), s.add_mirror((0.9, 0.85, 0.8)),
            s.add_metal(_VEACH_ETA, _VEACH_K, 0.2, 0.05, remap_roughness=False),
            s.add_substrate((0.6, 0.2, 0.2), (0.04, 0.04, 0.04), 0.05, 0.05, remap_roughness=True),
            s.add_plastic((0.2, 0.5, 0.3), (0.3, 0.3, 0.3), 0.15),
            s.add_uber(kd=(0.3, 0.3, 0.6), ks=(0.2, 0.2, 0.2), kr=(0.1, 0.1, 0.1), kt=(0.3, 0.3, 0.3),
                       opacity=(0.7, 0.7, 0.7), rough_u=0.1, rough_v=0.2, eta=1.4)]
    for i, m in enumerate(mats):
        s.add_sphere(0.6, m, ctm=glam.from_translation((-3.75 + 1.5 * i, 0.6, 0.5 * (i % 2))))
    tri_light = s.add_area_light_diffuse((8.0, 7.0, 6.0))
    lq = TriangleMesh.from_arrays([-1, 4, -1, 1, 4, -1, 1, 4, 1, -1, 4, 1], _QUAD_IDX, normals=[(0, -1, 0)] * 4)
    s.add_triangle_mesh(lq, 0, area_light=tri_light, ctm=glam.mul(glam.from_translation((0.5, 0.0, 0.0)), glam.from_scale((1.5, 1.0, 0.5))))
    sph_light = s.add_area_light_diffuse((20.0, 20.0, 30.0))
    s.add_sphere(0.25, s.add_matte((0, 0, 0)), area_light=sph_light, ctm=glam.from_translation((-3.0, 2.5, -1.0)))
    # an instanced, mirrored (negative determinant) mesh with a textured metal
    cube_p = [-1, 0, -1, 1, 0, -1, 1, 0, 1, -1, 0, 1, -1, 1, -1, 1, 1, -1, 1, 1, 1, -1, 1, 1]
    cube_i = [0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 1, 2, 6, 1, 6, 5, 2, 3, 7, 2, 7, 6, 3, 0, 4, 3, 4, 7]
    cube = TriangleMesh.from_arrays(cube_p, cube_i)
    ci = s.add_triangle_mesh(cube, s.add_metal(imap, _VEACH_K, 0.3, 0.3, remap_roughness=False),
                             ctm=glam.mul(glam.from_translation((3.6, 0.0, 2.2)), glam.from_scale((0.5, 0.8, 0.5))))
    s.add_mesh_instance(s.instances[ci].mesh_index, mats[4],
                        ctm=glam.mul(glam.from_translation((-4.2, 0.0, 2.4)), glam.from_scale((-0.4, 0.6, 0.4))))
    return s


# ---------------------------------------------------------------------------------------------------
# dragon-class (BASELINE config 4): the real dragon cannot be rendered here (4 of its meshes are
# absent from the reference checkout, .MISSING_LARGE_BLOBS), so SURVEY.md section 8d defines a
# stand-in of the same character: ~870 k triangles, all Matte, one distant light, NO area emitter
# (lights_len = 1, emit_object_len = 0 -> the plain-BSDF branch, lib.rs:325-337).
# ---------------------------------------------------------------------------------------------------
def _value_noise(p: np.ndarray, seed: int, octaves: int = 5) -> np.ndarray:
    """Deterministic lattice value noise on points p (n, 3), summed over octaves, in [-1, 1]."""
    def hash01(ix, iy, iz, s):
        h = (ix.astype(np.uint64) * np.uint64(73856093)) ^ (iy.astype(np.uint64) * np.uint64(19349663)) \
            ^ (iz.astype(np.uint64) * np.uint64(83492791)) ^ np.uint64(s * 2654435761 % (1 << 32))
        h = (h ^ (h >> np.uint64(13))) * np.uint64(1274126177) % np.uint64(1 << 32)
        return (h % np.uint64(1 << 24)).astype(np.float64) / float(1 << 24)

    out = np.zeros(p.shape[0], np.float64)
    amp, freq = 1.0, 2.0
    for o in range(octaves):
        q = p.astype(np.float64) * freq + 100.0
        i0 = np.floor(q).astype(np.int64)
        f = q - i0
        f = f * f * (3 - 2 * f)
        acc = np.zeros(p.shape[0], np.float64)
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                    acc += w * hash01(i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz, seed + o)
        out += amp * (2 * acc - 1)
        amp *= 0.5
        freq *= 2.0
    return out / 1.9375


def displaced_sphere(n_lat: int = 640, n_lon: int = 680, radius: float = 0.42, amplitude: float = 0.11,
                     seed: int = 7) -> TriangleMesh:
    """Latitude/longitude sphere (poles trimmed), radially displaced by value noise:
    2 * n_lat * n_lon triangles (870 400 by default), no vertex normals (flat shading)."""
    th = np.linspace(0.02, np.pi - 0.02, n_lat + 1)
    ph = np.linspace(0.0, 2 * np.pi, n_lon + 1)
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], axis=-1).reshape(-1, 3)
    d[np.isclose(P.reshape(-1), 2 * np.pi)] = d[np.isclose(P.reshape(-1), 0.0)]  # close the seam exactly
    r = radius * (1.0 + amplitude * _value_noise(d, seed))
    pos = (d * r[:, None]).astype(F32)
    i = np.arange(n_lat)[:, None] * (n_lon + 1) + np.arange(n_lon)[None, :]
    a, b, c, e = i, i + 1, i + n_lon + 1, i + n_lon + 2
    idx = np.stack([a, c, b, b, c, e], axis=-1).reshape(-1).astype(np.uint32)
    return TriangleMesh.from_arrays(pos, idx)


def dragon_class(xres: int = 1920, yres: int = 1080, n_lat: int = 640, n_lon: int = 680) -> Scene:
    """Cornell room (walls + tall box) whose short box is replaced by the displaced sphere; the quad
    emitter is removed and the dragon scene's distant light added (dragon/scene.pbrt:44)."""
    s = Scene.new()
    s.film.filename = "dragon-class.png"
    s.set_camera(glam.from_cols_array(CORNELL_WORLD_TO_CAMERA), CORNELL_FOV_DEG * 1.0, xres, yres)
    white = s.add_matte((0.725, 0.71, 0.68))
    red, green = s.add_matte((0.63, 0.065, 0.05)), s.add_matte((0.14, 0.45, 0.091))
    clay = s.add_matte((0.79311, 0.79311, 0.79311))  # "Dragon" Kd, dragon/scene.pbrt:11
    s.add_triangle_mesh(_quad([-1, 1.74846e-07, -1, -1, 1.74846e-07, 1, 1, -1.74846e-07, 1,
                               1, -1.74846e-07, -1], (4.37114e-08, 1, 1.91069e-15)), white)
    s.add_triangle_mesh(_quad([-1, 0, -1, -1, 2, -1, 1, 2, -1, 1, 0, -1], (8.74228e-08, -4.37114e-08, -1)), white)
    s.add_triangle_mesh(_quad([1, 0, -1, 1, 2, -1, 1, 2, 1, 1, 0, 1], (1, -4.37114e-08, 1.31134e-07)), green)
    s.add_triangle_mesh(_quad([-1, 0, 1, -1, 2, 1, -1, 2, -1, -1, 0, -1], (-1, -4.37114e-08, -4.37114e-08)), red)
    s.add_triangle_mesh(_box(_TALL_BOX_P, _TALL_BOX_N), white)
    s.add_triangle_mesh(displaced_sphere(n_lat, n_lon), clay, ctm=glam.from_translation((0.33, 0.48, 0.35)))
    s.add_light_distant((-0.18862, 0.692312, 0.69651), (0.0, 0.0, 0.0), (8.0, 8.0, 8.0))
    return s


def dragon_fog(xres: int = 1920, yres: int = 1080, n_lat: int = 640, n_lon: int = 680, emitter: bool = True) -> Scene:
    """The dragon-class room under Integrator "volpath" (SURVEY 8 f1 on a BVH-sized scene): a thin forward-scattering fog fills
    the room, the displaced sphere stands in it, the distant light reaches every vertex through the fog's boundary (tr walks,
    lib.rs:359-409); with `emitter` a dim quad light hangs under where the ceiling would be, so that scattering vertices and
    surfaces also draw their emitter sample (tr_emit, the one-sample mixture)."""
    s = dragon_class(xres, yres, n_lat, n_lon)
    s.integrator = abi.INTEGRATOR_VOLPATH
    s.film.filename = "dragon-fog.png"
    fog = s.add_medium_homogeneous((0.01, 0.01, 0.015), (0.22, 0.22, 0.24), 0.35)
    s.add_triangle_mesh(_aabb((-0.995, 0.005, -0.995), (0.995, 1.95, 0.995)), 0, interior=fog, exterior=0)
    if emitter:
        light = s.add_area_light_diffuse((8.5, 6.0, 2.0))
        s.add_triangle_mesh(_quad([-0.24, 1.9, -0.22, 0.23, 1.9, -0.22, 0.23, 1.9, 0.16, -0.24, 1.9, 0.16], (0, -1, 0)),
                            s.add_matte((0.0, 0.0, 0.0)), area_light=light, interior=fog, exterior=fog)
    return s


# ---------------------------------------------------------------------------------------------------
# teapot-class (BASELINE config 5): sample_scenes/teapot/scene.pbrt is 126 048 triangles of Substrate
# (Ks 0.04, roughness 0.001, remaproughness false) on a Matte floor with a 20 x 20 checkerboard under an
# infinite light with an environment map; its env map is absent from the reference checkout
# (.MISSING_LARGE_BLOBS) and its meshes may not be copied, so this is a stand-in of the same
# character built in code: a displaced sphere of the same triangle count, same materials, same light kind.
# ---------------------------------------------------------------------------------------------------
def teapot_class(xres: int = 1920, yres: int = 1080, n_lat: int = 250, n_lon: int = 252) -> Scene:
    s = Scene.new()
    s.film.filename = "teapot-class.png"
    s.set_camera(glam.look_at_lh((0.0, 1.6, -5.2), (0.0, 0.55, 0.0), (0.0, 1.0, 0.0)), 30.0, xres, yres)
    hh, ww = 64, 128
    v = np.linspace(0.0, 1.0, hh, dtype=F32)[:, None]
    u = np.linspace(0.0, 1.0, ww, dtype=F32)[None, :]
    sky = np.ones((hh, ww, 4), dtype=F32)
    sky[..., 0] = 0.35 + 0.5 * v
    sky[..., 1] = 0.45 + 0.4 * v
    sky[..., 2] = 0.75 + 0.15 * v
    sun = np.exp(-((u - 0.3) ** 2 + (v - 0.75) ** 2) / 0.002).astype(F32)
    sky[..., :3] += 6.0 * sun[..., None]
    s.set_infinite_light((1.0, 1.0, 1.0), image=sky)
    t1, t2 = s.add_texture_solid((0.325, 0.31, 0.25)), s.add_texture_solid((0.725, 0.71, 0.68))
    floor_mat = s.add_matte(s.add_texture_checkerboard(t1, t2, 20.0, 20.0))
    body = s.add_substrate((0.9, 0.9, 0.9), (0.04, 0.04, 0.04), 0.001, 0.001, remap_roughness=False)
    s.add_triangle_mesh(TriangleMesh.from_arrays([-8, 0, -8, 8, 0, -8, 8, 0, 8, -8, 0, 8], _QUAD_IDX,
                                                 normals=[(0, 1, 0)] * 4, uvs=_QUAD_UV), floor_mat)
    s.add_triangle_mesh(displaced_sphere(n_lat, n_lon, radius=0.8, amplitude=0.18, seed=11), body,
                        ctm=glam.from_translation((0.0, 0.95, 0.0)))
    return s


# ---------------------------------------------------------------------------------------------------
# teapot-full-class from the reference's own scene file and meshes (tests/golden/teapot: data fixture)
# ---------------------------------------------------------------------------------------------------
def synthetic_sky(width: int = 1024, height: int = 512) -> np.ndarray:
    """(height, width, 3) float32 environment map: a vertical gradient with a sun -- the stand-in for the scene's
    textures/envmap.pfm, which rene's checkout lacks (.MISSING_LARGE_BLOBS)."""
    v = np.linspace(0.0, 1.0, height, dtype=F32)[:, None]
    u = np.linspace(0.0, 1.0, width, dtype=F32)[None, :]
    sky = np.empty((height, width, 3), dtype=F32)
    sky[..., 0] = 0.35 + 0.5 * v
    sky[..., 1] = 0.45 + 0.4 * v
    sky[..., 2] = 0.75 + 0.15 * v
    sun = np.exp(-((u - 0.3) ** 2 + (v - 0.75) ** 2) / 0.002).astype(F32)
    return (sky + 6.0 * sun[..., None]).astype(F32)


def write_pfm(path: str, rgb: np.ndarray):
    """Colour PFM, little endian, bottom row first (rene/src/scene/pfm_parser.rs:10-61 reads it back)."""
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(f"PF\n{w} {h}\n-1.0\n".encode())
        f.write(np.ascontiguousarray(rgb[::-1], dtype="<f4").tobytes())


def teapot_full(xres: int = 1920, yres: int = 1080, asset_dir: str | None = None):
    """BASELINE config 5 from the reference's own inputs: sample_scenes/teapot/scene.pbrt (Substrate teapot, checkerboard
    Matte floor, infinite light with an environment map; 126 048 triangles) loaded through the pbrt-v3 loader
    (rene_scene_load_pbrt), with the Film size replaced and a synthetic 1024x512 sky as textures/envmap.pfm.
    Returns a rene_amd.loader.LoadedScene."""
    import os
    import re
    import shutil
    import tempfile
    from . import loader
    src = asset_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "teapot")
    tmp = tempfile.mkdtemp(prefix="rene_teapot_")
    try:
        text = open(os.path.join(src, "scene.pbrt")).read()
        text = re.sub(r'"integer xresolution" \[ \d+ \]', f'"integer xresolution" [ {xres} ]', text)
        text = re.sub(r'"integer yresolution" \[ \d+ \]', f'"integer yresolution" [ {yres} ]', text)
        open(os.path.join(tmp, "scene.pbrt"), "w").write(text)
        os.symlink(os.path.join(src, "models"), os.path.join(tmp, "models"))
        os.makedirs(os.path.join(tmp, "textures"))
        write_pfm(os.path.join(tmp, "textures", "envmap.pfm"), synthetic_sky())
        return loader.load_pbrt(os.path.join(tmp, "scene.pbrt"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dragon_partial(xres: int = 1280, yres: int = 720, asset_dir: str | None = None):
    """rene's sample_scenes/dragon with the 12 of its 16 meshes the checkout holds (tests/golden/dragon_partial: 'Dragon' by
    Delatronic, CC-BY 3.0; the body and two ground pieces are missing upstream), through the pbrt-v3 loader, Film size replaced:
    real artist meshes as a BVH datapoint next to the procedural dragon-class stand-in.  Returns a rene_amd.loader.LoadedScene."""
    import os
    import re
    import shutil
    import tempfile
    from . import loader
    src = asset_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dragon_partial")
    tmp = tempfile.mkdtemp(prefix="rene_dragon_")
    try:
        text = open(os.path.join(src, "scene.pbrt")).read()
        text = re.sub(r'"integer xresolution" \[ \d+ \]', f'"integer xresolution" [ {xres} ]', text)
        text = re.sub(r'"integer yresolution" \[ \d+ \]', f'"integer yresolution" [ {yres} ]', text)
        open(os.path.join(tmp, "scene.pbrt"), "w").write(text)
        os.symlink(os.path.join(src, "models"), os.path.join(tmp, "models"))
        return loader.load_pbrt(os.path.join(tmp, "scene.pbrt"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
