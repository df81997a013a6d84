"""rene_amd -- MI355X-native render path for hatoo/rene.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of include/rene_hip.h),
the ctypes mirror of that ABI, and the host-side mirror of rene's flat `Scene` tables.
"""
from . import abi, glam  # noqa: F401
from .scene import Film, Scene, TriangleMesh  # noqa: F401
