"""The C-ABI library loads on a machine without a GPU, exports every symbol the header declares,
its structs have the sizes the ctypes mirror assumes, and its host-only entry points behave."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT
from rene_amd import abi, api, scenes
from rene_amd.scene import Scene, TriangleMesh
from rene_amd import glam

HEADER = os.path.join(ROOT, "include", "rene_hip.h")


def test_header_is_plain_c_and_sizes_match_ctypes():
    names = {"rene_vertex": abi.Vertex, "rene_mesh": abi.Mesh, "rene_instance": abi.Instance,
             "rene_material": abi.Material, "rene_texture": abi.Texture, "rene_area_light": abi.AreaLight,
             "rene_light": abi.Light, "rene_image": abi.Image, "rene_uniform": abi.Uniform,
             "rene_scene_desc": abi.SceneDesc, "rene_opts": abi.Opts, "rene_stats": abi.Stats,
             "rene_hit": abi.Hit, "rene_pack_info": abi.PackInfo, "rene_medium": abi.Medium}
    prog = '#include <stdio.h>\n#include "rene_hip.h"\nint main(void){\n'
    for n in names:
        prog += f'printf("{n} %zu\\n", sizeof({n}));\n'
    prog += 'printf("offset_instances %zu\\n", offsetof(rene_scene_desc, instances));\n'
    prog += 'printf("offset_mediums %zu\\n", offsetof(rene_scene_desc, mediums));\n'
    prog += 'printf("offset_framebuffer %zu\\n", offsetof(rene_opts, framebuffer));\nreturn 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.dirname(HEADER), src, "-o", exe])
        out = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
    for n, cls in names.items():
        assert int(out[n]) == C.sizeof(cls), n
    assert int(out["offset_instances"]) == abi.SceneDesc.instances.offset
    assert int(out["offset_mediums"]) == abi.SceneDesc.mediums.offset
    assert int(out["offset_framebuffer"]) == abi.Opts.framebuffer.offset


def test_library_exports_every_declared_symbol(hip_lib):
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(rene_[a-z0-9_]+)\s*\(", text))
    declared -= {"rene_status"}
    assert declared == set(abi.EXPORTED_SYMBOLS), declared ^ set(abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(hip_lib, name), name
    assert hip_lib.rene_abi_version() == abi.ABI_VERSION


def test_no_oracle_in_product(hip_lib):
    """The product must not link, import or call anything under oracle/."""
    out = subprocess.check_output(["ldd", api.LIB_PATH]).decode()
    assert "oracle" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rene_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                t = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in t and "from oracle" not in t and "oracle/" not in t.replace(
                    "under oracle/", "").replace("oracle/ is", ""), f


def test_frame_seeds_follow_pcg_schedule(hip_lib, oracle_mod):
    got = api.frame_seeds(abi.DEFAULT_SEED, 0, 8)
    assert got.tolist() == oracle_mod.pcg_u32(abi.DEFAULT_SEED, 8).tolist()
    assert api.frame_seeds(7, 5, 3).tolist() == oracle_mod.pcg_u32(7, 8)[5:].tolist()
    # the k-th seed comes from jumping the generator ahead, not from producing its predecessors: far frames cost nothing
    far = api.frame_seeds(7, 4_000_000_000, 6)
    assert far[4:].tolist() == api.frame_seeds(7, 4_000_000_004, 2).tolist()
    assert api.frame_seeds(7, 100_000, 4).tolist() == oracle_mod.pcg_u32(7, 100_004)[100_000:].tolist()


def test_output_transform_matches_restatement(hip_lib, oracle_mod):
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.uniform(0, 2, 4000), rng.uniform(0, 0.004, 500),
                        [0.0, -1.0, 1.0, 0.0031308, np.nan, np.inf, 1e-9, 0.5 / 255 / 12.92 * 16]]).astype(np.float32)
    sums = (v * np.float32(16)).astype(np.float32)
    got = api.to_rgb8(sums, 16)
    want = oracle_mod.to_rgb8(sums, 16)
    # powf may differ by an ulp between libm and numpy: allow off-by-one on a handful of values
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    assert got[-4] == 0 and got[-3] == 255  # NaN -> 0 (Rust `as u8`), inf -> 255
    aov = api.to_aov8(np.array([0.0, 8.0, 16.0, -16.0], np.float32), 16, False)
    assert aov.tolist() == [0, 128, 255, 0]  # (256 * clamp(v, 0, 0.999)) as u8, main.rs:1794-1801
    aovn = api.to_aov8(np.array([0.0, 16.0, -16.0], np.float32), 16, True)
    assert aovn.tolist() == [128, 255, 0]


def test_pack_info_cornell(hip_lib):
    info = api.pack_info(scenes.cornell_box(64, 64)).as_dict()
    assert info["n_triangles"] == 36 and info["n_instances"] == 8 and info["n_spheres"] == 0
    assert info["emit_object_len"] == 1 and info["lights_len"] == 0 and info["n_slots_emit"] == 2
    assert info["features"] == 64  # Matte-only fast path + FEAT_SMALL (wave-coherent item loop)
    # 36 triangles -> 18 parallelograms -> the room (five-sided box) + two blocks + the light; the emitter structure: the light
    assert info["n_items_main"] == 4 and info["n_items_emit"] == 1
    v = api.pack_info(scenes.veach_mis(64, 64)).as_dict()
    assert v["n_items_main"] == 9  # 4 plates (boxes) + 2 walls + 3 spheres
    assert api.pack_info(scenes.dragon_class(64, 36, 24, 26)).n_items_main == 0  # BVH scene
    assert info["n_slots_main"] == 36 and 1 <= info["n_nodes_main"] < 36


def _tiny(**kw):
    s = Scene.new()
    s.set_camera(glam.identity(), 45.0, kw.get("w", 16), kw.get("h", 16))
    return s


def test_validation_errors(hip_lib):
    def code(scene_or_packed):
        with pytest.raises(api.ReneError) as e:
            api.pack_info(scene_or_packed)
        return e.value.code

    s = _tiny()
    s.add_triangle_mesh(TriangleMesh.from_arrays([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 1, 2]), material=99)
    assert code(s) == -2  # material out of range
    s = _tiny()
    s.integrator = 7
    assert code(s) == -2  # only path (0) and volpath (1) exist, main.rs:520-523
    s = _tiny()
    s.integrator = abi.INTEGRATOR_VOLPATH
    m = TriangleMesh.from_arrays([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 1, 2])
    s.add_triangle_mesh(m, 0, interior=3)
    assert code(s) == -2  # medium index out of range
    s = _tiny()
    s.integrator = abi.INTEGRATOR_VOLPATH
    s.add_triangle_mesh(m, 0, interior=s.add_medium_homogeneous((1, 1, 1), (float("nan"), 1, 1)))
    assert code(s) == -2  # non-finite coefficient
    s = _tiny(w=1, h=16)
    assert code(s) == -2  # W-1 division, lib.rs:178
    s = _tiny()
    p = s.to_desc()
    p.desc.struct_size = 12
    assert code(p) == -1  # ABI skew
    s = _tiny()
    m = TriangleMesh.from_arrays([0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 1, 2])
    s.add_triangle_mesh(m, s.add_matte())
    s.instances[0].matrix[:] = [0.0] * 12
    assert code(s) == -2  # singular instance matrix
    s = _tiny()
    s.add_triangle_mesh(m, s.add_matte())
    s.instances[0].mesh_index = 5
    assert code(s) == -2
    # empty scene is valid (every ray misses)
    assert api.pack_info(_tiny()).n_slots_main == 0


def test_pack_info_features(hip_lib):
    s = _tiny()
    s.add_sphere(0.5, s.add_metal(rough_u=0.1, rough_v=0.1, remap_roughness=False),
                 area_light=s.add_area_light_diffuse((1, 1, 1)))
    s.add_light_distant((0, 0, 1), (0, 0, 0), (1, 1, 1))
    info = api.pack_info(s).as_dict()
    assert info["n_spheres"] == 1 and info["emit_object_len"] == 1 and info["lights_len"] == 1
    assert info["features"] & 1 and info["features"] & 2 and info["features"] & 8
    # instancing is flattened: two instances of one mesh -> twice the triangles
    s = _tiny()
    m = TriangleMesh.from_arrays([0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0], [0, 1, 2, 1, 3, 2])
    i0 = s.add_triangle_mesh(m, s.add_matte())
    s.add_mesh_instance(0, 1, ctm=glam.from_translation((3, 0, 0)))
    assert api.pack_info(s).n_triangles == 4


def test_render_path_fails_loudly_without_gpu(hip_lib):
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    with pytest.raises(api.ReneError) as e:
        api.Renderer(scenes.cornell_box(16, 16))
    assert e.value.code == -3  # RENE_ERR_DEVICE: no CPU fallback


def test_item_merging_counts_on_the_host(hip_lib):
    """build_small_items (scene_pack.cpp) without a GPU: triangles -> parallelograms -> box items."""
    def info(build):
        s = _tiny()
        build(s)
        return api.pack_info(s)
    m = lambda s: s.add_matte((0.5, 0.5, 0.5))
    rot = glam.mul(glam.from_translation((0.3, -0.2, 2.0)), glam.mul(glam.from_axis_angle((0.3, 0.8, 0.5), 0.7), glam.from_scale((0.5, 0.9, -0.7))))
    box = scenes._aabb((-1, -1, -1), (1, 1, 1))
    # a closed box of one mesh, rotated, scaled and mirrored: 12 triangles -> 1 item
    assert info(lambda s: s.add_triangle_mesh(box, m(s), ctm=rot)).n_items_main == 1
    # the same box as six instances, one per face
    v, idx = box.vertices, box.indices
    faces = [TriangleMesh.from_arrays(v[:, 0:3].reshape(-1), idx[6 * f: 6 * f + 6], normals=v[:, 3:6], uvs=v[:, 6:8].reshape(-1)) for f in range(6)]
    def six(s, skip=()):
        for f in range(6):
            if f not in skip:
                s.add_triangle_mesh(faces[f], m(s), ctm=rot)
    assert info(six).n_items_main == 1
    assert info(lambda s: six(s, skip=(2,))).n_items_main == 1      # five faces: an open box
    assert info(lambda s: six(s, skip=(2, 3))).n_items_main == 4    # four faces: four parallelograms
    # two triangles that do not form a parallelogram stay two items; a lone triangle is one
    tri2 = TriangleMesh.from_arrays([0, 0, 3, 1, 0, 3, 0, 1, 3, 1.5, 1.2, 3], [0, 1, 2, 1, 3, 2])
    assert info(lambda s: s.add_triangle_mesh(tri2, m(s))).n_items_main == 2
    # more than 64 items: the scene leaves the item loop
    def many(s):
        for k in range(70):
            s.add_triangle_mesh(TriangleMesh.from_arrays([k, 0, 3, k + 0.5, 0, 3, k, 1, 3], [0, 1, 2]), m(s))
    i = info(many)
    assert i.n_items_main == 0 and not (i.features & 64)


def test_missing_rccl_is_an_error_code_not_a_crash():
    """with no RCCL to load, rene_comm_unique_id returns RENE_ERR_UNSUPPORTED with a message (dlerror() is read once;
    reading it twice handed std::string a NULL).  A fresh process, because the library looks for RCCL once."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from rene_amd import api\n"
            "try:\n"
            "    api.comm_unique_id()\n"
            "    print('LOADED')\n"
            "except api.ReneError as e:\n"
            "    print('CODE', e.code, '|', str(e))\n" % ROOT)
    env = dict(os.environ, RENE_RCCL_LIB="/nonexistent/librccl-absent.so")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-500:]
    assert "CODE -4" in p.stdout and "RCCL is not available" in p.stdout and "librccl-absent" in p.stdout, p.stdout
    # several candidates that all fail, as on the default path of a host without RCCL -- the message kept is the FIRST failure's,
    # copied before the next dlopen() rewrites the loader's buffer (it used to be a pointer into that buffer)
    env = dict(os.environ, RENE_RCCL_LIB="/nonexistent/librccl-first.so:/nonexistent/" + "x" * 300 + ".so:/nonexistent/librccl-third.so")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-500:]
    assert "CODE -4" in p.stdout and "librccl-first" in p.stdout and "librccl-third" not in p.stdout, p.stdout


def test_image_maps_are_rgba_and_rgb_is_padded():
    # rene_image is four floats per texel (include/rene_hip.h): an (h, w, 3) array handed to the scene builder is padded,
    # anything else is refused (it used to be passed on as it was, and the packer read past its end)
    import numpy as np
    from rene_amd.scene import Scene
    s = Scene.new()
    t = s.add_texture_image_map(np.full((5, 7, 3), 0.25, dtype=np.float32))
    assert s.images[-1].shape == (5, 7, 4) and (s.images[-1][..., 3] == 1).all() and (s.images[-1][..., :3] == 0.25).all()
    s.add_texture_image_map(np.zeros((2, 2, 4), dtype=np.float32))
    for bad in (np.zeros((4, 4)), np.zeros((4, 4, 2)), np.zeros((4,))):
        with pytest.raises(ValueError):
            s.add_texture_image_map(bad)
    assert t >= 0
