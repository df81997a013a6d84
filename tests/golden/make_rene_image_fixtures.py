"""Box-filtered copies of rene's OWN published renders (images/cornell-box.png 1024x1024,
images/veach-mis.png 1280x720; README.md:27-59 of the reference, produced by its Vulkan path at
5000 spp) as small golden fixtures for the T2 parity tier.  These are output
data of the reference, not source.  The teapot / teapot-full images are OIDN-denoised, and the teapot scene's environment
map is missing from the checkout: their RADIANCE is not used.  The teapot image's GEOMETRY is (tier T3, tests/t3_geometry.py):
which pixels are brighter than the floor's dark squares -- the checkerboard's phase and the teapot's outline.  The dragon
scene cannot be loaded here (missing meshes).

  rene_<scene>_box8.npy   8 x 8 means of the sRGB values (f32): the round-1 T2 fixtures
  rene_<scene>_box4.npy   [2][H/4][W/4][3] f16: 4 x 4 means of the sRGB values, and 4 x 4 means of the values decoded to
                          linear light (inverse of main.rs:1768-1774) -- the region-wise energy checks of tests/t2_regions.py

  rene_dragon_box4.npy    [2][180][320] f16: images/dragon.png (1280 x 720, NOT denoised: README.md:37-43), channel mean, 4 x 4 means of the sRGB
                          values and of the values decoded to linear light -- the lit surfaces of the 12 meshes the checkout holds
                          (tests/t2_regions.py, dragon_lit_ratio)
  rene_teapot_bright.npy  images/teapot.png (1280 x 720) as ONE bit per pixel, np.packbits of  luminance > 0.51  row by row
                          (Rec. 709 luminance of the sRGB values; the dark squares of the floor end at 0.487, its light squares begin at 0.628)

    python tests/golden/make_rene_image_fixtures.py      # needs /root/reference
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/images"


def load(path):
    return np.asarray(Image.open(path).convert("RGB"), np.float32) / 255.0


def box(a, k):
    h, w, _ = a.shape
    return a[: h // k * k, : w // k * k].reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3)).astype(np.float32)


def box1(a, k):  # one channel
    h, w, _ = a.shape
    return a.reshape(h // k, k, w // k, k).mean(axis=(1, 3)).astype(np.float32)


def to_linear(s):  # inverse of gamma_correct, main.rs:1768-1774
    return np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4).astype(np.float32)


TEAPOT_BRIGHT = 0.51


def teapot_bright(path=os.path.join(REF, "teapot.png")):
    """[720][1280] bool: rene's teapot render brighter than the floor's dark squares."""
    a = load(path)
    return (a @ np.array([0.2126, 0.7152, 0.0722], np.float32)) > TEAPOT_BRIGHT


if __name__ == "__main__":
    d = load(os.path.join(REF, "dragon.png")).mean(axis=2, keepdims=True)
    np.save(os.path.join(HERE, "rene_dragon_box4.npy"), np.stack([box1(d, 4), box1(to_linear(d), 4)]).astype(np.float16))
    np.save(os.path.join(HERE, "rene_teapot_bright.npy"), np.packbits(teapot_bright(), axis=1))  # [720][160] u8
    for name, png in (("cornell", "cornell-box.png"), ("veach_mis", "veach-mis.png")):
        a = load(os.path.join(REF, png))
        np.save(os.path.join(HERE, f"rene_{name}_box8.npy"), box(a, 8))                    # 128x128x3 / 90x160x3
        np.save(os.path.join(HERE, f"rene_{name}_box4.npy"), np.stack([box(a, 4), box(to_linear(a), 4)]).astype(np.float16))
