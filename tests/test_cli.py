"""rene's command line over the C ABI (rene/src/main.rs:47-207): option surface on CPU, a full
render on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, have_reference, REFERENCE
from rene_amd import abi, api, loader, scenes

CLI = os.path.join(ROOT, "rene_amd", "csrc", "rene-hip")


@pytest.fixture(scope="module")
def cli(hip_lib):
    if not os.path.exists(CLI):
        api.build()
    return CLI


def test_cli_option_surface(cli, tmp_path):
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([cli, "--denoiser", "bogus", "x.pbrt"], capture_output=True, text=True)
    assert r.returncode == 2
    # scene errors are printed to stdout and the program returns (main.rs:199-205)
    bad = tmp_path / "bad.pbrt"
    bad.write_text('WorldBegin\nShape "cone"\nWorldEnd\n')
    r = subprocess.run([cli, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "Invalid Shape type cone" in r.stdout
    # --frame-groups (round 3's opt-in two chains of frames per pixel) is accepted and means nothing: every job renders eight frame chains
    r = subprocess.run([cli, "--frame-groups", "--spp", "5", "--denoiser", "bogus", "x.pbrt"], capture_output=True, text=True)
    assert r.returncode == 2 and "even --spp" not in r.stderr
    # --dump-module writes the gfx950 code object (the reference dumps its SPIR-V module, main.rs:100-106)
    out = tmp_path / "module.co"
    r = subprocess.run([cli, "--dump-module", str(out)], capture_output=True, text=True)
    data = out.read_bytes()
    assert r.returncode == 0 and data[:24] == b"__CLANG_OFFLOAD_BUNDLE__" and b"gfx950" in data[:4096]


def test_c1_plumbing_loader_to_cpu_integrator(hip_lib, oracle_mod, tmp_path):
    """BASELINE config 1: cornell-box 256x256 @ 16 spp through the loader into the CPU reference
    integrator (plumbing only, no GPU).  Uses the real scene file when the reference checkout exists."""
    if have_reference():
        text = open(os.path.join(REFERENCE, "sample_scenes", "cornell-box", "scene.pbrt")).read()
    else:
        text = loader.scene_to_pbrt(scenes.cornell_box(1024, 1024))
    text = text.replace("[ 1024 ]", "[ 256 ]")
    p = tmp_path / "scene.pbrt"
    p.write_text(text)
    s = loader.load_pbrt(str(p))
    assert (s.xres, s.yres) == (256, 256) and s.film_filename == "cornell-box.png"
    o = oracle_mod.Oracle(s)
    o.render(0, 16)
    img = o.download(0) / 16
    ref = oracle_mod.Oracle(scenes.cornell_box(256, 256))
    ref.render(0, 16)
    np.testing.assert_allclose(img, ref.download(0) / 16, rtol=1e-4, atol=1e-5)
    st = o.stats()
    assert st.paths == 256 * 256 * 16 and st.rays > st.paths


@pytest.mark.gpu
def test_cli_renders_png_identical_to_api(cli, tmp_path):
    from PIL import Image
    syn = scenes.cornell_box(96, 64)
    p = tmp_path / "scene.pbrt"
    p.write_text(loader.scene_to_pbrt(syn))
    out = tmp_path / "o.png"
    r = subprocess.run([cli, str(p), "--spp", "12", "--batch", "5", "--out", str(out),
                        "--aov-normal", str(tmp_path / "n.png"), "--aov-albedo", str(tmp_path / "a.png"),
                        "--denoiser", "oidn"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert "Samples: 12 / 12" in r.stderr and "denoiser" in r.stderr  # progress line + warn-and-ignore
    got = np.asarray(Image.open(out).convert("RGB"))
    with api.Renderer(loader.load_pbrt(str(p))) as rr:
        rr.render(0, 12)
        want = api.to_rgb8(rr.download(0), 12)
        want_n = api.to_aov8(rr.download(1), 12, True)
        want_a = api.to_aov8(rr.download(2), 12, False)
    assert np.array_equal(got, want)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "n.png").convert("RGB")), want_n)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.png").convert("RGB")), want_a)
    # Film filename is the default output; --width/--height override the Film size
    r = subprocess.run([cli, str(p), "--spp", "2", "--width", "48", "--height", "32"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert Image.open(tmp_path / "cornell-box.png").size == (48, 32)


@pytest.mark.gpu
def test_cli_renders_a_volpath_scene_with_media(cli, tmp_path):
    """`Integrator "volpath"` + MakeNamedMedium / MediumInterface from a .pbrt file through rene-hip: same PNG as the
    API renders from the same file, and with `--gpus 2` (tile shards summed on the host) the same image again."""
    from PIL import Image
    p = tmp_path / "fog.pbrt"
    p.write_text(loader.scene_to_pbrt(scenes.cornell_fog(64, 48)))
    out = tmp_path / "fog.png"
    r = subprocess.run([cli, str(p), "--spp", "8", "--out", str(out)], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    got = np.asarray(Image.open(out).convert("RGB"))
    with api.Renderer(loader.load_pbrt(str(p))) as rr:
        rr.render(0, 8)
        want = api.to_rgb8(rr.download(0), 8)
    assert np.array_equal(got, want)
    assert got.mean() > 20  # lit fog, not a black frame
    out2 = tmp_path / "fog2.png"
    r = subprocess.run([cli, str(p), "--spp", "8", "--gpus", "2", "--out", str(out2)], capture_output=True, text=True, cwd=tmp_path)
    if r.returncode == 0:  # one GPU on the test box: the CLI refuses more devices than it sees
        assert np.array_equal(np.asarray(Image.open(out2).convert("RGB")), want)
    else:
        assert "GPU" in r.stderr or "device" in r.stderr
