#!/usr/bin/env python3
"""How much of a BVH step's cost is the memory behind it: the dragon-class room with the displaced sphere at 1/64 ... 1 of its
triangle count (the structure goes from L2-resident to 60 MB), rays/s and node / leaf visits per ray of the counting variant.
    gpurun -- python3 tools/scene_size.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from rene_amd import abi, api, scenes
    for n_lat, n_lon in ((80, 85), (160, 170), (320, 340), (640, 680)):
        s = scenes.dragon_class(1920, 1080, n_lat, n_lon)
        info = api.pack_info(s)
        with api.Renderer(s, flags=abi.FLAG_COUNTERS) as r:
            r.render(0, 4)
            c = r.stats().as_dict()
        with api.Renderer(s) as r:
            r.render(0, 16)
            r.sync()
            r.reset()
            t0 = time.perf_counter()
            r.render(0, 256)
            r.sync()
            dt = time.perf_counter() - t0
            st = r.stats()
        print(f"{info.n_triangles:8d} triangles, {info.n_nodes_main:7d} nodes ({info.device_bytes / 1e6:6.1f} MB on the device), stack bound {info.depth_main}: "
              f"{st.rays / dt / 1e6:7.0f} Mrays/s, {dt * 1e3 / 256:.3f} ms/frame; per ray {c['node_visits'] / c['rays']:.2f} node visits, "
              f"{c['prim_tests'] / c['rays']:.2f} triangle tests, rays/path {st.rays / st.paths:.2f}", flush=True)


if __name__ == "__main__":
    main()
