"""Box-filtered copies of rene's OWN published renders (images/cornell-box.png 1024x1024,
images/veach-mis.png 1280x720; README.md:27-59 of the reference, produced by its Vulkan path at
5000 spp) as small golden fixtures for the T2 parity tier.  These are output
data of the reference, not source.  The teapot / teapot-full images are OIDN-denoised and the
dragon scene cannot be loaded here (missing meshes), so they are not used.

  rene_<scene>_box8.npy   8 x 8 means of the sRGB values (f32): the round-1 T2 fixtures
  rene_<scene>_box4.npy   [2][H/4][W/4][3] f16: 4 x 4 means of the sRGB values, and 4 x 4 means of the values decoded to
                          linear light (inverse of main.rs:1768-1774) -- the region-wise energy checks of tests/t2_regions.py

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


def to_linear(s):  # inverse of gamma_correct, main.rs:1768-1774
    return np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4).astype(np.float32)


if __name__ == "__main__":
    for name, png in (("cornell", "cornell-box.png"), ("veach_mis", "veach-mis.png")):
        a = load(os.path.join(REF, png))
        np.save(os.path.join(HERE, f"rene_{name}_box8.npy"), box(a, 8))                    # 128x128x3 / 90x160x3
        np.save(os.path.join(HERE, f"rene_{name}_box4.npy"), np.stack([box(a, 4), box(to_linear(a), 4)]).astype(np.float16))
