"""Tier T3: the GEOMETRY of a render against rene's own published image of the same scene file (images/teapot.png =
sample_scenes/teapot/scene.pbrt at 1280 x 720, README.md:45-51 of the reference).

That image is OIDN-denoised and lit by an environment map the checkout does not hold, so its radiance pins nothing.  Where
things ARE in it does: the floor's checkerboard (camera, the floor's transform, the triangle hit shader's uv interpolation,
the checkerboard texture's scale and parity, intermediate_scene.rs / texture.rs) and the outline of the 126 046-triangle
teapot (the PLY loader, the transforms, the traversal) -- reference-held data for a third scene and for the texture path,
which T2's two scenes do not reach.  One bit per pixel of rene's image (tests/golden/rene_teapot_bright.npy: brighter than
the floor's dark squares) is compared with the first-hit albedo layer of ONE frame of this build:

  checker    floor pixels whose 5 x 5 neighbourhood sees one kind of square: light square <=> bright in rene's image
  inside     the ring 2-6 pixels inside the teapot's outline is bright (porcelain against the dark squares)
  outside    the ring 2-6 pixels outside it, where this build sees a dark square, is not

Used by test_oracle_render.py (oracle, CPU) and test_gpu_t2.py (HIP path).
"""
import os

import numpy as np

from conftest import GOLDEN


def rene_teapot_bright():
    """[720][1280] bool from the committed fixture."""
    return np.unpackbits(np.load(os.path.join(GOLDEN, "rene_teapot_bright.npy")), axis=1).astype(bool)


def geometry(albedo, bright):
    """albedo: [720][1280][>= 3] first-hit albedo of one frame (layer 2; top row first); returns the three agreements."""
    from scipy import ndimage as ndi
    r = albedo[..., 0]
    cls = np.where(r > 0.85, 2, np.where(r > 0.5, 1, 0))  # Substrate Kd 0.9 | tex2 0.725 | tex1 0.325 (scene.pbrt)
    uniform = ndi.maximum_filter(cls, 5) == ndi.minimum_filter(cls, 5)
    floor = (cls < 2) & uniform
    teapot = cls == 2
    inner = ndi.binary_erosion(teapot, iterations=2) & ~ndi.binary_erosion(teapot, iterations=6)
    outer = ndi.binary_dilation(teapot, iterations=6) & ~ndi.binary_dilation(teapot, iterations=2) & (cls == 0)
    return {"floor_pixels": int(floor.sum()), "checker": float(((cls == 1) == bright)[floor].mean()),
            "teapot_pixels": int(teapot.sum()), "inside": float(bright[inner].mean()), "outside": float(bright[outer].mean())}
