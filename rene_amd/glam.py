"""Restatement of the few glam 0.20.5 constructors rene's host side uses.

glam is a third-party crate that is not vendored under /root/reference (Cargo.lock pins
`glam 0.20.5`); these follow its documented formulas.  Matrices are numpy float32 arrays of shape
(4, 4) indexed ``m[row, col]``; ``to_cols`` flattens them column-major, which is glam's (and the C
ABI's) storage order.  Call sites in the reference: rene/src/scene.rs:155-165,
rene/src/scene/intermediate_scene.rs:1026-1062.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def identity() -> np.ndarray:
    return np.eye(4, dtype=F32)


def from_cols_array(vals) -> np.ndarray:
    """Mat4::from_cols(x, y, z, w) given 16 numbers in column order (pbrt `Transform [...]`,
    pbrt-parser/src/lib.rs:204-217)."""
    return np.asarray(vals, dtype=F32).reshape(4, 4).T.copy()


def to_cols(m: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(m, dtype=F32).T).reshape(16)


def from_translation(t) -> np.ndarray:
    m = identity()
    m[:3, 3] = np.asarray(t, dtype=F32)
    return m


def from_scale(s) -> np.ndarray:
    m = identity()
    m[0, 0], m[1, 1], m[2, 2] = (F32(v) for v in s)
    return m


def from_axis_angle(axis, angle: float) -> np.ndarray:
    """Mat4::from_axis_angle (axis must be normalised; intermediate_scene.rs:1035-1038)."""
    x, y, z = (float(F32(v)) for v in axis)
    s, c = math.sin(float(F32(angle))), math.cos(float(F32(angle)))
    s, c = float(F32(s)), float(F32(c))
    omc = 1.0 - c
    m = identity()
    m[:3, 0] = [x * x * omc + c, x * y * omc + z * s, x * z * omc - y * s]
    m[:3, 1] = [x * y * omc - z * s, y * y * omc + c, y * z * omc + x * s]
    m[:3, 2] = [x * z * omc + y * s, y * z * omc - x * s, z * z * omc + c]
    return m.astype(F32)


def look_at_lh(eye, center, up) -> np.ndarray:
    """Mat4::look_at_lh = look_to_lh(eye, center - eye, up) (intermediate_scene.rs:1049-1053)."""
    eye = np.asarray(eye, dtype=np.float64)
    d = np.asarray(center, dtype=np.float64) - eye
    f = d / np.linalg.norm(d)
    s = np.cross(np.asarray(up, dtype=np.float64), f)
    s = s / np.linalg.norm(s)
    u = np.cross(f, s)
    m = np.eye(4, dtype=np.float64)
    m[0, :3], m[1, :3], m[2, :3] = s, u, f
    m[0, 3], m[1, 3], m[2, 3] = -np.dot(s, eye), -np.dot(u, eye), -np.dot(f, eye)
    return m.astype(F32)


def perspective_lh(fov_y: float, aspect: float, z_near: float, z_far: float) -> np.ndarray:
    """Mat4::perspective_lh, depth range [0, 1] (rene/src/scene.rs:163-164)."""
    fov_y, aspect, z_near, z_far = (float(F32(v)) for v in (fov_y, aspect, z_near, z_far))
    s, c = math.sin(0.5 * fov_y), math.cos(0.5 * fov_y)
    h = float(F32(c)) / float(F32(s))
    w = h / aspect
    r = z_far / (z_far - z_near)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = w
    m[1, 1] = h
    m[2, 2] = r
    m[3, 2] = 1.0
    m[2, 3] = -r * z_near
    return m.astype(F32)


def inverse(m: np.ndarray) -> np.ndarray:
    """Mat4::inverse.  Evaluated in float64 and rounded once; glam's f32 cofactor expansion
    differs from this by a few ulp, which is below every parity tolerance (the boundary is
    unpinned by any reference test)."""
    return np.linalg.inv(np.asarray(m, dtype=np.float64)).astype(F32)


def mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return (np.asarray(a, dtype=np.float64) @ np.asarray(b, dtype=np.float64)).astype(F32)


def affine_from_mat4(m: np.ndarray) -> np.ndarray:
    """Affine3A::from_mat4 -> 12 floats: x_axis, y_axis, z_axis, translation (scene.rs:296, 419)."""
    m = np.asarray(m, dtype=F32)
    return np.concatenate([m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3]]).astype(F32)
