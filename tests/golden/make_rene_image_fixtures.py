"""Box-filtered copies of rene's OWN published renders (images/cornell-box.png 1024x1024,
images/veach-mis.png 1280x720; README.md:27-59 of the reference, produced by its Vulkan path at
5000 spp) as small golden fixtures for the T2 parity tier.  These are output
data of the reference, not source.  The teapot / teapot-full images are OIDN-denoised and the
dragon scene cannot be loaded here (missing meshes), so they are not used.

    python tests/golden/make_rene_image_fixtures.py      # needs /root/reference
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/images"


def box(path, k):
    a = np.asarray(Image.open(path).convert("RGB"), np.float32) / 255.0
    h, w, _ = a.shape
    return a[: h // k * k, : w // k * k].reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3)).astype(np.float32)


if __name__ == "__main__":
    np.save(os.path.join(HERE, "rene_cornell_box8.npy"), box(os.path.join(REF, "cornell-box.png"), 8))   # 128x128x3
    np.save(os.path.join(HERE, "rene_veach_mis_box8.npy"), box(os.path.join(REF, "veach-mis.png"), 8))   # 90x160x3
