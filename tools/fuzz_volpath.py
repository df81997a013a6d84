#!/usr/bin/env python3
"""Developer script: random scenes under Integrator "volpath" -- homogeneous media behind None-material boxes and spheres,
a distant light and / or a quad emitter, a displaced sphere that makes the tree deep enough (> 512 nodes) for the
traversal-restart kernel -- rendered by the restart kernel and by the while-while kernel (RENE_FLAG_NO_RESTART), across launch
splits and work-item cuts: every bit of the three layers must agree; against the oracle: T1.
    gpurun -- python3 tools/fuzz_volpath.py [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import abi, api, glam, scenes  # noqa: E402
from rene_amd.scene import Scene, TriangleMesh  # noqa: E402
from oracle import oracle  # noqa: E402


def rand_scene(rng, k):
    s = Scene.new()
    s.integrator = abi.INTEGRATOR_VOLPATH
    w, h = int(rng.integers(48, 96)), int(rng.integers(32, 64))
    s.set_camera(glam.look_at_lh((0.0, 1.1, -5.0), (0.0, 0.7, 0.0), (0.0, 1.0, 0.0)), 38.0, w, h)
    general = k % 2 == 1
    mats = [s.add_matte(tuple(rng.uniform(0.2, 0.9, 3)))]
    if general:
        mats += [s.add_substrate(tuple(rng.uniform(0.2, 0.8, 3)), tuple(rng.uniform(0.05, 0.4, 3)), 0.1, 0.2), s.add_glass(1.4)]
    fl = TriangleMesh.from_arrays(np.float32([[-3, 0, -3], [3, 0, -3], [3, 0, 3], [-3, 0, 3]]), np.uint32([0, 2, 1, 0, 3, 2]))
    s.add_triangle_mesh(fl, mats[0])
    s.add_triangle_mesh(scenes.displaced_sphere(40, 44, radius=0.5, amplitude=0.12, seed=int(rng.integers(1, 99))), int(rng.choice(mats)),
                        ctm=glam.from_translation((float(rng.uniform(-0.8, 0.8)), 0.62, float(rng.uniform(-0.5, 0.8)))))
    fog = s.add_medium_homogeneous(tuple(rng.uniform(0.0, 0.05, 3)), tuple(rng.uniform(0.05, 0.5, 3)), float(rng.uniform(-0.5, 0.7)))
    dense = s.add_medium_homogeneous(tuple(rng.uniform(0.1, 0.6, 3)), tuple(rng.uniform(1.0, 5.0, 3)), float(rng.uniform(-0.3, 0.5)))
    lo, hi = (-2.5, 0.02, -2.5), (2.5, 2.4, 2.5)
    P = np.float32([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                    [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
    I = np.uint32([0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 3, 6, 2, 3, 7, 6, 0, 4, 7, 0, 7, 3, 1, 2, 6, 1, 6, 5])
    s.add_triangle_mesh(TriangleMesh.from_arrays(P, I), 0, interior=fog, exterior=0)  # the fog's boundary: material None
    for _ in range(int(rng.integers(1, 4))):  # dense blobs inside the fog
        c = rng.uniform([-1.4, 0.5, -1.2], [1.4, 1.6, 1.2])
        s.add_sphere(float(rng.uniform(0.15, 0.4)), 0, ctm=glam.from_translation(tuple(c)), interior=dense, exterior=fog)
    for _ in range(int(rng.integers(0, 12))):
        c = rng.uniform([-1.6, 0.1, -1.6], [1.6, 1.8, 1.6])
        v = (c + rng.normal(0, 0.3, (3, 3))).astype(np.float32)
        s.add_triangle_mesh(TriangleMesh.from_arrays(v, np.uint32([0, 1, 2])), int(rng.choice(mats)), interior=fog, exterior=fog)
    kind = k % 3
    if kind != 1:
        al = s.add_area_light_diffuse(tuple(rng.uniform(4, 12, 3)))
        q = np.float32([[-.4, 2.2, -.4], [.4, 2.2, -.4], [.4, 2.2, .4], [-.4, 2.2, .4]])
        s.add_triangle_mesh(TriangleMesh.from_arrays(q, np.uint32([0, 1, 2, 0, 2, 3])), mats[0], area_light=al, interior=fog, exterior=fog)
    if kind != 0:
        s.add_light_distant((1.0, 2.0, -1.5), (0.0, 0.0, 0.0), tuple(rng.uniform(1, 4, 3)))
    return s, general


def main():
    bad = 0
    for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
        rng = np.random.default_rng(7000 + k)
        s, general = rand_scene(rng, k)
        info = api.pack_info(s)
        assert info.n_nodes_main > 512, info.n_nodes_main
        plan = [(0, 4), (4, 3), (7, 5)]
        outs = {}
        for name, flags in (("restart", 0), ("while-while", abi.FLAG_NO_RESTART)):
            for levels in (None, "3"):
                if levels:
                    os.environ["RENE_LEVELS"] = levels
                else:
                    os.environ.pop("RENE_LEVELS", None)
                with api.Renderer(s, flags=flags | abi.FLAG_COUNTERS) as r:
                    for f0, n in plan:
                        r.render(f0, n)
                    outs[(name, levels)] = ([r.download(l) for l in range(3)], r.stats().as_dict())
        os.environ.pop("RENE_LEVELS", None)
        ref, sref = outs[("while-while", None)]
        msgs = []
        for key, (imgs, st) in outs.items():
            for l in range(3):
                if not np.array_equal(imgs[l], ref[l]):
                    msgs.append(f"{key} layer {l}: max |d| {np.abs(imgs[l] - ref[l]).max():.3g}, differing {(imgs[l] != ref[l]).mean():.3g}")
            for c in ("rays_closest", "rays_shadow", "rays_emitter", "hits", "adds"):
                if st[c] != sref[c]:
                    msgs.append(f"{key} counter {c}: {st[c]} vs {sref[c]}")
        o = oracle.Oracle(s)
        o.render(0, 12)
        g, c = outs[("restart", None)][0][0], o.download(0)
        fin = np.isfinite(g).all(axis=-1) & np.isfinite(c).all(axis=-1)
        diff = (np.abs(g - c)[fin] > 1e-2 * (1 + np.abs(c[fin]))).any(axis=-1).mean()
        if diff > (3e-2 if general else 1e-2) or fin.mean() < 0.999:
            msgs.append(f"oracle: {diff:.3g} of the pixels off, finite {fin.mean():.4f}")
        print(f"volpath scene {k}: features {info.features:#x}, {info.n_triangles} triangles, {info.n_nodes_main} nodes, general {general}, "
              f"rays/path {sref['rays'] / sref['paths']:.1f}: {'ok' if not msgs else msgs}", flush=True)
        bad += bool(msgs)
    print("FAILED" if bad else "all scenes agree", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
