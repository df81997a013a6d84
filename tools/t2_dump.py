#!/usr/bin/env python3
"""GPU box: renders rene's two published raw scenes at rene's own size and 5000 spp under several master seeds and writes the 4 x 4 box means
(linear average before the output transform, and after average -> to_rgb8 -> decode) to gpurun_out/t2dump/<scene>_seed<k>.npz, for the
offline analysis of the per-surface offsets against rene's PNGs (tools/cornell_offsets.py).
  python3 tools/t2_dump.py [cornell|veach_mis ...] [--seeds 4] [--spp 5000]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rene_amd import abi, api, scenes
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["cornell", "veach_mis"])
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--spp", type=int, default=5000)
    ap.add_argument("--out", default="gpurun_out/t2dump")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    mk = {"cornell": lambda: scenes.cornell_box(1024, 1024), "veach_mis": lambda: scenes.veach_mis(1280, 720)}

    def box(x, k=4):
        h, w, c = x.shape
        return x.reshape(h // k, k, w // k, k, c).mean(axis=(1, 3))

    def to_linear(s):
        return np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)

    for nm in a.names:
        s = mk[nm]()
        for k in range(a.seeds):
            seed = abi.DEFAULT_SEED if k == 0 else (0x9E3779B9 * k) & 0xFFFFFFFF
            with api.Renderer(s, seed=seed) as r:
                r.render(0, a.spp)
                img = r.download(0)
                st = r.stats()
            rgb8 = api.to_rgb8(img, a.spp)
            np.savez_compressed(os.path.join(a.out, f"{nm}_seed{k}.npz"), lin=box(img / a.spp).astype(np.float32),
                                dec=box(to_linear(rgb8.astype(np.float32) / 255.0)).astype(np.float32),
                                srgb=box(rgb8.astype(np.float32) / 255.0).astype(np.float32))
            print(f"{nm} seed {seed:#x}: {st.rays / 1e9:.2f} G rays, mean {img.mean() / a.spp:.5f}", flush=True)


if __name__ == "__main__":
    main()
