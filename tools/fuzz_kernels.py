"""Developer script: random small scenes (triangle soups, quads, spheres, every single-lobe material mix, emitters, distant
lights, backgrounds) rendered by every scheduling that can take them -- item loop, while-while BVH, traversal-restart,
wavefront -- with and without the overlap flag (ignored since ABI v4) and across work-item cuts.  Matte-only scenes must agree bit for bit;
general ones to a last bit (the restart kernel re-derives the surface after a shadow query).  Against the oracle: T1."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rene_amd import abi, api, glam
from rene_amd.scene import Scene, TriangleMesh
from oracle import oracle

def rand_scene(rng, k):
    s = Scene.new()
    w, h = int(rng.integers(40, 90)), int(rng.integers(30, 70))
    s.set_camera(glam.look_at_lh((0.0, 1.0, -5.0), (0.0, 0.6, 0.0), (0.0, 1.0, 0.0)), 38.0, w, h)
    general = k % 3 != 0
    mats = [s.add_matte(tuple(rng.uniform(0.2, 0.9, 3)))]
    if general:
        pick = rng.integers(0, 4)
        if pick in (0, 3): mats.append(s.add_metal(rough_u=float(rng.uniform(0.02, 0.3)), rough_v=float(rng.uniform(0.02, 0.3)), remap_roughness=False))
        if pick in (1, 3): mats.append(s.add_substrate(tuple(rng.uniform(0.2, 0.8, 3)), tuple(rng.uniform(0.05, 0.4, 3)), 0.1, 0.2))
        if pick == 2: mats += [s.add_glass(1.5), s.add_mirror((0.9, 0.9, 0.9))]
    # floor
    fl = TriangleMesh.from_arrays(np.float32([[-3, 0, -3], [3, 0, -3], [3, 0, 3], [-3, 0, 3]]), np.uint32([0, 2, 1, 0, 3, 2]))
    s.add_triangle_mesh(fl, mats[0])
    n_obj = int(rng.integers(1, 70 if k % 2 else 8))  # few: item loop; many: BVH
    for _ in range(n_obj):
        c = rng.uniform([-1.5, 0.1, -1.5], [1.5, 1.6, 1.5])
        if rng.random() < 0.25:
            s.add_sphere(float(rng.uniform(0.08, 0.35)), int(rng.choice(mats)), ctm=glam.from_translation(tuple(c)))
        else:
            v = (c + rng.normal(0, 0.3, (3, 3))).astype(np.float32)
            s.add_triangle_mesh(TriangleMesh.from_arrays(v, np.uint32([0, 1, 2])), int(rng.choice(mats)))
    if rng.random() < 0.6:
        al = s.add_area_light_diffuse(tuple(rng.uniform(4, 12, 3)))
        q = np.float32([[-.4, 1.9, -.4], [.4, 1.9, -.4], [.4, 1.9, .4], [-.4, 1.9, .4]])
        s.add_triangle_mesh(TriangleMesh.from_arrays(q, np.uint32([0, 1, 2, 0, 2, 3])), mats[0], area_light=al)
    if rng.random() < 0.5:
        s.add_light_distant((1.0, 2.0, -1.5), (0.0, 0.0, 0.0), tuple(rng.uniform(1, 4, 3)))
    if rng.random() < 0.5:
        s.set_infinite_light(tuple(rng.uniform(0.1, 0.6, 3)))
    return s, general

bad = 0
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    rng = np.random.default_rng(1000 + k)
    try:
        s, general = rand_scene(rng, k)
    except TypeError as e:
        print("scene builder:", e); raise
    info = api.pack_info(s)
    plan = [(0, 5), (5, 3), (8, 6)]
    outs = {}
    variants = {"default": 0, "bvh-ww": abi.FLAG_FORCE_BVH | abi.FLAG_NO_RESTART, "bvh-restart": abi.FLAG_FORCE_BVH,
                "bvh-wavefront": abi.FLAG_FORCE_BVH | abi.FLAG_WAVEFRONT, "overlap": abi.FLAG_OVERLAP,
                "bvh-restart-overlap": abi.FLAG_FORCE_BVH | abi.FLAG_OVERLAP}
    for name, flags in variants.items():
        for levels in (None, "3"):
            if levels: os.environ["RENE_LEVELS"] = levels
            else: os.environ.pop("RENE_LEVELS", None)
            with api.Renderer(s, flags=flags) as r:
                for f0, n in plan: r.render(f0, n)
                outs[(name, levels)] = [r.download(l) for l in range(3)]
    os.environ.pop("RENE_LEVELS", None)
    ref = outs[("bvh-ww", None)]
    msgs = []
    for key, imgs in outs.items():
        for l in range(3):
            if key[0] in ("default", "overlap") and (info.features & 64):
                ok = np.array_equal(imgs[l], outs[("default", None)][l])  # item loop family among itself
                # vs the BVH family: other intersection arithmetic, so paths fork at silhouettes: T1, not bits
                off = (np.abs(imgs[l] - ref[l]) > 1e-2 * 14 * (1 + np.abs(ref[l]) / 14)).any(axis=-1).mean()
                ok = ok and off < 2e-2
            elif general and "restart" in key[0]:
                ok = np.allclose(imgs[l], ref[l], rtol=1e-5, atol=1e-6) and (imgs[l] != ref[l]).mean() < 5e-3
            else:
                ok = np.array_equal(imgs[l], ref[l])
            if not ok: msgs.append(f"{key} layer {l}: max |d| {np.abs(imgs[l] - ref[l]).max():.3g}, differing {(imgs[l] != ref[l]).mean():.3g}")
    o = oracle.Oracle(s); o.render(0, 14)
    g, c = outs[("default", None)][0], o.download(0)
    diff = (np.abs(g - c) > 1e-2 * (1 + np.abs(c))).any(axis=-1).mean()
    if diff > (2e-2 if general else 5e-3): msgs.append(f"oracle: {diff:.3g} of the pixels off")
    print(f"scene {k}: features {info.features}, {info.n_triangles} triangles, {info.n_spheres} spheres, general {general}: {'ok' if not msgs else msgs}", flush=True)
    bad += bool(msgs)
print("FAILED" if bad else "all scenes agree", bad)
sys.exit(1 if bad else 0)
