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

from . import glam
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
