#!/usr/bin/env python3
"""GPU box: rene's Cornell at 1024 x 1024 @ 5000 spp, full resolution, for pixel-noise statistics against rene's PNG (tools/cornell_offsets.py's
companion): gpurun_out/<dir>/cornell_full_seed<k>.npz with the 8-bit image (after rene's output transform) and, for seed 0,
the linear sums."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import abi, api, scenes  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/t2full"
os.makedirs(out, exist_ok=True)
s = scenes.cornell_box(1024, 1024)
for k, seed in enumerate((abi.DEFAULT_SEED, 0x9E3779B9)):
    with api.Renderer(s, seed=seed) as r:
        r.render(0, 5000)
        img = r.download(0)
    rgb8 = api.to_rgb8(img, 5000)
    extra = {"lin": (img / 5000).astype(np.float32)} if k == 0 else {}
    np.savez_compressed(os.path.join(out, f"cornell_full_seed{k}.npz"), rgb8=rgb8, **extra)
    print("seed", hex(seed), "finite", bool(np.isfinite(img).all()), flush=True)
