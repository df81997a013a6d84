"""Image files behind `Texture "imagemap"` / the infinite light (load_image, rene/src/scene/intermediate_scene.rs:631-677
-> rene_amd/csrc/image_io.cpp): TGA and BMP through the LDR branch (RGBA8 -> inverse gamma), OpenEXR through the
HDR branch.  The reference uses the `image` and `exr` crates, which are not in the checkout, and no reference test
reads an image, so parity is unpinned; the decoders are checked against files written here by independent encoders
(struct / numpy / zlib, straight from the format specifications) and -- when /root/reference is present -- against
the reference's own PIZ-compressed EXR renders and their PNG companions."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

from rene_amd import api, loader
from conftest import REFERENCE, have_reference


def load(tmp_path, name):
    text = f'WorldBegin\nTexture "t" "spectrum" "imagemap" "string filename" "{name}"\nMaterial "matte" "texture Kd" "t"\nShape "sphere"\nWorldEnd\n'
    ls = loader.parse_pbrt(text, str(tmp_path))  # owns the tables desc points into
    d = ls.desc
    im = d.images[d.n_images - 1]
    return np.frombuffer(C.string_at(im.rgba, im.height * im.width * 16), np.float32).reshape(im.height, im.width, 4).copy()


def srgb_to_linear(v8):
    v = np.asarray(v8, np.float32) / np.float32(255.0)
    return np.where(v <= 0.04045, v / np.float32(12.92), ((v + np.float32(0.055)) / np.float32(1.055)) ** np.float32(2.4)).astype(np.float32)


def expect_ldr(rgba8):
    out = np.empty(rgba8.shape, np.float32)
    out[..., :3] = srgb_to_linear(rgba8[..., :3])
    out[..., 3] = rgba8[..., 3].astype(np.float32) / np.float32(255.0)
    return out


def _picture(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    img[: h // 2, : w // 2] = (200, 30, 90, 255)  # a flat patch so that run-length packets occur
    return img


# ------------------------------------------------------------------------------------------------ TGA
def tga_bytes(img, kind, rle=False, top_down=False, id_bytes=b""):
    h, w = img.shape[:2]
    rows = img if top_down else img[::-1]
    cmap = b""
    if kind == "bgr":
        px, depth, typ, cm = rows[..., [2, 1, 0]].reshape(-1, 3), 24, 2, (0, 0, 0, 0)
    elif kind == "bgra":
        px, depth, typ, cm = rows[..., [2, 1, 0, 3]].reshape(-1, 4), 32, 2, (0, 0, 0, 0)
    elif kind == "grey":
        px, depth, typ, cm = rows[..., :1].reshape(-1, 1), 8, 3, (0, 0, 0, 0)
    elif kind == "grey-alpha":
        px, depth, typ, cm = rows[..., [0, 3]].reshape(-1, 2), 16, 3, (0, 0, 0, 0)
    else:  # colour map: quantise to 16 colours, first index 3
        pal = np.array([[(37 * i) % 256, (91 * i) % 256, (53 * i) % 256] for i in range(16)], np.uint8)
        idx = (rows[..., 0] % 16).astype(np.uint8)
        px, depth, typ, cm = (idx + 3).reshape(-1, 1), 8, 1, (1, 3, 16, 24)
        cmap = pal[:, ::-1].tobytes()
    if rle:
        typ += 8
        body, i, n = bytearray(), 0, len(px)
        while i < n:
            run = 1
            while i + run < n and run < 128 and (px[i + run] == px[i]).all():
                run += 1
            if run > 1:
                body += bytes([0x80 | (run - 1)]) + px[i].tobytes()
                i += run
            else:
                lit = 1
                while i + lit < n and lit < 128 and not (i + lit + 1 < n and (px[i + lit] == px[i + lit + 1]).all()):
                    lit += 1
                body += bytes([lit - 1]) + px[i:i + lit].tobytes()
                i += lit
        body = bytes(body)
    else:
        body = px.tobytes()
    desc = (0x20 if top_down else 0) | (8 if kind in ("bgra", "grey-alpha") else 0)
    head = struct.pack("<BBBHHBHHHHBB", len(id_bytes), cm[0], typ, cm[1], cm[2], cm[3], 0, 0, w, h, depth, desc)
    return head + id_bytes + cmap + body


@pytest.mark.parametrize("kind", ["bgr", "bgra", "grey", "grey-alpha", "cmap"])
@pytest.mark.parametrize("rle", [False, True])
def test_tga(tmp_path, hip_lib, kind, rle):
    img = _picture(13, 21, 3)
    for top_down in (False, True):
        (tmp_path / "a.tga").write_bytes(tga_bytes(img, kind, rle=rle, top_down=top_down, id_bytes=b"id" if rle else b""))
        got = load(tmp_path, "a.tga")
        want = img.copy()
        if kind == "bgr":
            want[..., 3] = 255
        elif kind == "grey":
            want = np.repeat(img[..., :1], 4, axis=2); want[..., 3] = 255
        elif kind == "grey-alpha":
            want = np.concatenate([np.repeat(img[..., :1], 3, axis=2), img[..., 3:]], axis=2)
        elif kind == "cmap":
            pal = np.array([[(37 * i) % 256, (91 * i) % 256, (53 * i) % 256] for i in range(16)], np.uint8)
            want = np.concatenate([pal[img[..., 0] % 16], np.full(img.shape[:2] + (1,), 255, np.uint8)], axis=2)
        np.testing.assert_allclose(got.reshape(13, 21, 4), expect_ldr(want), rtol=5e-7, atol=0)  # powf, last bit


# ------------------------------------------------------------------------------------------------ BMP
def bmp_bytes(img, bpp, top_down=False, bitfields=False):
    h, w = img.shape[:2]
    rows = img if top_down else img[::-1]
    pal = b""
    if bpp == 8:
        palette = np.array([[(37 * i) % 256, (91 * i) % 256, (53 * i) % 256] for i in range(256)], np.uint8)
        pal = np.concatenate([palette[:, ::-1], np.zeros((256, 1), np.uint8)], axis=1).tobytes()
        px = rows[..., 0]
    elif bpp == 24:
        px = rows[..., [2, 1, 0]].reshape(h, -1)
    else:
        px = (rows[..., [3, 0, 1, 2]] if bitfields else rows[..., [2, 1, 0, 3]]).reshape(h, -1)  # masks below: A, R, G, B bytes
    stride = (px.shape[1] + 3) // 4 * 4
    body = b"".join(r.tobytes() + b"\0" * (stride - px.shape[1]) for r in px)
    masks = struct.pack("<IIII", 0x0000ff00, 0x00ff0000, 0xff000000, 0x000000ff) if bitfields else b""
    dib = struct.pack("<IiiHHIIiiII", 40 + (16 if bitfields else 0), w, -h if top_down else h, 1, bpp, 3 if bitfields else 0, len(body), 2835, 2835, 0, 0) + masks
    off = 14 + len(dib) + len(pal)
    return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + dib + pal + body


@pytest.mark.parametrize("bpp,bitfields", [(8, False), (24, False), (32, False), (32, True)])
def test_bmp(tmp_path, hip_lib, bpp, bitfields):
    img = _picture(9, 14, 4)
    for top_down in (False, True):
        (tmp_path / "a.bmp").write_bytes(bmp_bytes(img, bpp, top_down=top_down, bitfields=bitfields))
        got = load(tmp_path, "a.bmp")
        want = img.copy()
        if bpp == 8:
            palette = np.array([[(37 * i) % 256, (91 * i) % 256, (53 * i) % 256] for i in range(256)], np.uint8)
            want = np.concatenate([palette[img[..., 0]], np.full(img.shape[:2] + (1,), 255, np.uint8)], axis=2)
        elif not bitfields:
            want[..., 3] = 255  # BI_RGB has no alpha
        np.testing.assert_allclose(got.reshape(9, 14, 4), expect_ldr(want), rtol=5e-7, atol=0)


def test_bmp_hostile_masks(tmp_path, hip_lib):
    """a full-width mask used to spin for ever (shift + width = 32), a mask with holes has no meaning, and
    a block offset near 2^64 used to wrap the bounds check of the EXR reader."""
    img = _picture(2, 2, 4)
    good = bmp_bytes(img, 32, bitfields=True)
    at = good.index(struct.pack("<IIII", 0x0000ff00, 0x00ff0000, 0xff000000, 0x000000ff))
    full = good[:at] + struct.pack("<IIII", 0xffffffff, 0x00ff0000, 0xff000000, 0) + good[at + 16:]
    (tmp_path / "full.bmp").write_bytes(full)
    got = load(tmp_path, "full.bmp").reshape(2, 2, 4)  # returns (the top byte of the whole pixel word as red), does not hang
    assert np.isfinite(got).all()
    holes = good[:at] + struct.pack("<IIII", 0x00ff00ff, 0x00ff0000, 0xff000000, 0) + good[at + 16:]
    (tmp_path / "holes.bmp").write_bytes(holes)
    with pytest.raises(api.ReneError) as e:
        load(tmp_path, "holes.bmp")
    assert e.value.code == -6 and "non-contiguous" in str(e.value)
    exr = exr_bytes({"R": (np.zeros((4, 4), np.float16), "half"), "G": (np.zeros((4, 4), np.float16), "half"), "B": (np.zeros((4, 4), np.float16), "half")}, 4, 4, 0)
    # the offset table (4 blocks: one line each without compression) follows the header: its first entry points just past itself
    table = next(p for p in range(len(exr) - 8) if struct.unpack("<Q", exr[p:p + 8])[0] == p + 32)
    bad = exr[:table] + struct.pack("<Q", 2 ** 64 - 4) + exr[table + 8:]
    (tmp_path / "wrap.exr").write_bytes(bad)
    with pytest.raises(api.ReneError) as e:
        load(tmp_path, "wrap.exr")
    assert e.value.code == -6 and "offset" in str(e.value)


# ------------------------------------------------------------------------------------------------ EXR
def _attr(name, typ, value):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value


def exr_bytes(channels, h, w, compression, x0=0, y0=0):
    """channels: {name: (array[h, w], 'half' | 'float' | 'uint')}.  Scan-line file, increasing y."""
    names = sorted(channels)
    tcode = {"uint": 0, "half": 1, "float": 2}
    ndt = {"uint": "<u4", "half": "<f2", "float": "<f4"}
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", tcode[channels[n][1]], 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    head = struct.pack("<II", 20000630, 2)
    head += _attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([compression]))
    head += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box) + _attr("lineOrder", "lineOrder", b"\0")
    head += _attr("pixelAspectRatio", "float", struct.pack("<f", 1)) + _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    head += _attr("screenWindowWidth", "float", struct.pack("<f", 1)) + b"\0"
    lines = {0: 1, 1: 1, 2: 1, 3: 16}[compression]
    blocks = []
    for b0 in range(0, h, lines):
        raw = b"".join(np.asarray(channels[n][0][y], ndt[channels[n][1]]).tobytes() for y in range(b0, min(h, b0 + lines)) for n in names)
        data = raw
        if compression in (1, 2, 3):
            t = np.frombuffer(raw, np.uint8)
            t = np.concatenate([t[0::2], t[1::2]])  # even bytes | odd bytes
            d = t.astype(np.int32)
            d[1:] = (d[1:] - d[:-1] + 128 + 256) % 256  # predictor
            shuffled = d.astype(np.uint8).tobytes()
            if compression == 1:
                out, i = bytearray(), 0
                while i < len(shuffled):
                    run = 1
                    while i + run < len(shuffled) and run < 128 and shuffled[i + run] == shuffled[i]:
                        run += 1
                    if run >= 3:
                        out += struct.pack("b", run - 1) + shuffled[i:i + 1]
                        i += run
                    else:
                        lit = 0
                        while i + lit < len(shuffled) and lit < 127:
                            if i + lit + 2 < len(shuffled) and shuffled[i + lit] == shuffled[i + lit + 1] == shuffled[i + lit + 2]:
                                break
                            lit += 1
                        lit = max(lit, 1)
                        out += struct.pack("b", -lit) + shuffled[i:i + lit]
                        i += lit
                data = bytes(out)
            else:
                data = zlib.compress(shuffled)
            if len(data) >= len(raw):
                data = raw  # the writers fall back to the raw block
        blocks.append(struct.pack("<ii", y0 + b0, len(data)) + data)
    table_at = len(head)
    offs, at = [], table_at + 8 * len(blocks)
    for b in blocks:
        offs.append(at)
        at += len(b)
    return head + b"".join(struct.pack("<Q", o) for o in offs) + b"".join(blocks)


@pytest.mark.parametrize("compression", [0, 1, 2, 3])
def test_exr_scanline_compressions_and_channel_types(tmp_path, hip_lib, compression):
    rng = np.random.default_rng(compression)
    h, w = 37, 23  # not a multiple of the 16-line ZIP block
    r = rng.random((h, w)).astype(np.float16)
    r[:10] = np.float16(0.25)  # compressible rows
    g = (rng.random((h, w)) * 100).astype(np.float32)
    b = rng.integers(0, 70000, (h, w)).astype(np.uint32)
    a = rng.random((h, w)).astype(np.float16)
    r[0, 0], r[0, 1], r[0, 2] = np.float16(6e-8), np.float16(-0.0), np.float16(65504)  # subnormal, signed zero, largest half
    chans = {"R": (r, "half"), "G": (g, "float"), "B": (b, "uint"), "A": (a, "half"), "Z": (g * 2, "float")}  # Z: ignored
    (tmp_path / "a.exr").write_bytes(exr_bytes(chans, h, w, compression, x0=-5, y0=7))
    got = load(tmp_path, "a.exr").reshape(h, w, 4)
    np.testing.assert_array_equal(got[..., 0], r.astype(np.float32))
    np.testing.assert_array_equal(got[..., 1], g)
    np.testing.assert_array_equal(got[..., 2], b.astype(np.float32))
    np.testing.assert_array_equal(got[..., 3], a.astype(np.float32))


def test_exr_without_alpha_and_luminance_only(tmp_path, hip_lib):
    rng = np.random.default_rng(9)
    h, w = 5, 7
    r, g, b = (rng.random((h, w)).astype(np.float16) for _ in range(3))
    (tmp_path / "rgb.exr").write_bytes(exr_bytes({"R": (r, "half"), "G": (g, "half"), "B": (b, "half")}, h, w, 3))
    got = load(tmp_path, "rgb.exr").reshape(h, w, 4)
    np.testing.assert_array_equal(got[..., :3], np.stack([r, g, b], axis=2).astype(np.float32))
    assert (got[..., 3] == 1.0).all()
    (tmp_path / "y.exr").write_bytes(exr_bytes({"Y": (g, "half")}, h, w, 2))
    got = load(tmp_path, "y.exr").reshape(h, w, 4)
    for k in range(3):
        np.testing.assert_array_equal(got[..., k], g.astype(np.float32))


def test_image_errors(tmp_path, hip_lib):
    good = exr_bytes({"R": (np.zeros((4, 4), np.float16), "half"), "G": (np.zeros((4, 4), np.float16), "half"), "B": (np.zeros((4, 4), np.float16), "half")}, 4, 4, 3)
    cases = {"trunc.exr": good[:-7], "magic.exr": b"abcd" + good[4:], "tiled.exr": good[:4] + struct.pack("<I", 2 | 0x200) + good[8:],
             "dwa.exr": good.replace(b"compression\0compression\0\x01\0\0\0\x03", b"compression\0compression\0\x01\0\0\0\x08"),
             "trunc.tga": tga_bytes(_picture(4, 4, 1), "bgr")[:30], "type.tga": bytes([0, 0, 7]) + bytes(15),
             "sig.bmp": b"XX" + bmp_bytes(_picture(4, 4, 1), 24)[2:], "short.bmp": bmp_bytes(_picture(4, 4, 1), 24)[:60]}
    for name, blob in cases.items():
        (tmp_path / name).write_bytes(blob)
        with pytest.raises(api.ReneError) as e:
            load(tmp_path, name)
        assert e.value.code == -6 and "decode error" in str(e.value), (name, str(e.value))
    (tmp_path / "a.gif").write_bytes(b"GIF89a")
    with pytest.raises(api.ReneError) as e:
        load(tmp_path, "a.gif")
    assert e.value.code == -4  # RENE_ERR_UNSUPPORTED, never a guess


@pytest.mark.reference
@pytest.mark.skipif(not have_reference(), reason="needs /root/reference/sample_scenes")
@pytest.mark.parametrize("scene", ["cornell-box", "veach-mis"])
def test_piz_against_the_reference_renders(tmp_path, hip_lib, scene):
    """sample_scenes/*/TungstenRender.exr are PIZ-compressed half RGB files (written by Tungsten, not by rene);
    TungstenRender.png next to each is the same render tone-mapped to 8 bits.  A PIZ decoder that is wrong
    anywhere (Huffman table, run-length symbol, wavelet, value table) gives noise, so: the decoded image must be
    finite and non-negative, and one rising tone curve must relate its values to the PNG's."""
    d = os.path.join(REFERENCE, "sample_scenes", scene)
    os.symlink(os.path.join(d, "TungstenRender.exr"), tmp_path / "t.exr")
    os.symlink(os.path.join(d, "TungstenRender.png"), tmp_path / "t.png")
    hdr = load(tmp_path, "t.exr")
    ldr = load(tmp_path, "t.png")  # linear again: the loader applies inverse gamma
    assert hdr.shape == ldr.shape
    rgb, ref = hdr.reshape(-1, 4)[:, :3], ldr.reshape(-1, 4)[:, :3]
    assert np.isfinite(rgb).all() and (rgb >= 0).all() and (hdr.reshape(-1, 4)[:, 3] == 1).all()
    # the PNG is the EXR through one tone curve applied to every channel value: pooled over pixels and channels,
    # samples of equal HDR value must share their LDR value, and the curve must rise
    x, y = rgb.reshape(-1), ref.reshape(-1)
    ok = (y > 0.02) & (y < 0.8) & (x > 1e-3)
    assert ok.mean() > 0.2
    order = np.argsort(x[ok])
    n = len(order) // 64 * 64
    yb = y[ok][order][:n].reshape(64, -1)
    assert (yb.std(axis=1) / yb.mean(axis=1))[:-4].max() < 0.05  # the last bins span the sparse bright end of the curve
    assert (np.diff(yb.mean(axis=1)) > 0).all()


def test_tga_and_bmp_written_by_pil(tmp_path, hip_lib):
    """A second, independent encoder: PIL's TGA (raw and RLE, RGB / RGBA / L) and BMP (RGB, palette) writers."""
    from PIL import Image
    img = _picture(11, 19, 8)
    cases = {"rgb.tga": (Image.fromarray(img[..., :3], "RGB"), {}), "rgba_rle.tga": (Image.fromarray(img, "RGBA"), {"compression": "tga_rle"}),
             "l_rle.tga": (Image.fromarray(img[..., 0], "L"), {"compression": "tga_rle"}), "rgb.bmp": (Image.fromarray(img[..., :3], "RGB"), {}),
             "pal.bmp": (Image.fromarray(img[..., :3], "RGB").quantize(32), {})}
    for name, (im, kw) in cases.items():
        im.save(tmp_path / name, **kw)
        want = np.asarray(Image.open(tmp_path / name).convert("RGBA"), dtype=np.uint8)
        np.testing.assert_allclose(load(tmp_path, name), expect_ldr(want), rtol=5e-7, atol=0, err_msg=name)


# ------------------------------------------------------------------------------------------------ PNG, the rest of RFC 2083
def png_bytes(smp, depth, ctype, interlace=False, plte=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """smp: [h, w, channels] integer samples at `depth` bits.  Rows are filtered with the types in `filters` in turn."""
    h, w, ch = smp.shape

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xffffffff)

    def scanlines(sub):
        out, prev = b"", None
        sh, sw, _ = sub.shape
        bpp = max(1, ch * depth // 8)
        for y in range(sh):
            row = sub[y].reshape(-1)
            if depth == 16:
                line = np.stack([row >> 8, row & 255], axis=1).reshape(-1).astype(np.uint8)
            elif depth == 8:
                line = row.astype(np.uint8)
            else:
                bits = np.zeros(((len(row) * depth + 7) // 8) * 8, np.uint8)
                for k, v in enumerate(row):
                    for b in range(depth):
                        bits[k * depth + b] = (int(v) >> (depth - 1 - b)) & 1
                line = np.packbits(bits)
            f = filters[y % len(filters)]
            cur = line.astype(np.int32)
            a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            b = prev.astype(np.int32) if prev is not None else np.zeros_like(cur)
            c = np.concatenate([np.zeros(bpp, np.int32), b[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            if f == 0: pred = np.zeros_like(cur)
            elif f == 1: pred = a
            elif f == 2: pred = b
            elif f == 3: pred = (a + b) // 2
            else:
                pa, pb, pc = np.abs(b - c), np.abs(a - c), np.abs(a + b - 2 * c)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
            out += bytes([f]) + ((cur - pred) % 256).astype(np.uint8).tobytes()
            prev = line
        return out

    if interlace:
        passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
        raw = b"".join(scanlines(smp[y0::dy, x0::dx]) for x0, y0, dx, dy in passes if smp[y0::dy, x0::dx].size)
    else:
        raw = scanlines(smp)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        out += chunk(b"PLTE", np.asarray(plte, np.uint8).tobytes())
    if trns is not None:
        out += chunk(b"tRNS", trns)
    return out + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


@pytest.mark.parametrize("interlace", [False, True])
def test_png_every_depth_interlacing_and_colour_keys(tmp_path, hip_lib, interlace):
    rng = np.random.default_rng(11)
    h, w = 11, 13  # smaller than some Adam7 passes' strides: empty and ragged passes occur
    for depth, ctype, ch in ((16, 0, 1), (16, 2, 3), (16, 4, 2), (16, 6, 4), (8, 2, 3), (8, 6, 4), (4, 0, 1), (2, 0, 1), (1, 0, 1), (4, 3, 1), (1, 3, 1)):
        smp = rng.integers(0, 1 << depth, (h, w, ch))
        plte = trns = None
        if ctype == 3:
            plte = rng.integers(0, 256, (1 << depth, 3))
            trns = bytes(rng.integers(0, 256, 1 << (depth - 1)).tolist())  # shorter than the palette: the rest is opaque
        elif ctype == 0 and depth != 2:
            key = int(smp[3, 4, 0])
            trns = struct.pack(">H", key)
        elif ctype == 2:
            key = smp[5, 6]
            trns = struct.pack(">HHH", *[int(v) for v in key])
        (tmp_path / "a.png").write_bytes(png_bytes(smp, depth, ctype, interlace, plte, trns))
        got = load(tmp_path, "a.png")
        to8 = (lambda v: (v + 128) // 257) if depth == 16 else (lambda v: v) if depth == 8 else (lambda v: v * 255 // ((1 << depth) - 1))
        want = np.empty((h, w, 4), np.uint8)
        if ctype == 3:
            want[..., :3] = plte[smp[..., 0]]
            t = np.frombuffer(trns, np.uint8)
            want[..., 3] = np.where(smp[..., 0] < len(t), t[np.minimum(smp[..., 0], len(t) - 1)], 255)
        elif ch <= 2:
            want[..., :3] = to8(smp[..., :1])
            want[..., 3] = to8(smp[..., 1]) if ch == 2 else (np.where(smp[..., 0] == key, 0, 255) if trns else 255)
        else:
            want[..., :3] = to8(smp[..., :3])
            want[..., 3] = to8(smp[..., 3]) if ch == 4 else np.where((smp == key).all(axis=2), 0, 255)
        np.testing.assert_allclose(got, expect_ldr(want), rtol=5e-7, atol=0, err_msg=f"depth {depth} colour type {ctype}")
    # PIL reads what this encoder writes (a check of the encoder, not of the decoder)
    from PIL import Image
    smp = rng.integers(0, 256, (h, w, 3))
    (tmp_path / "p.png").write_bytes(png_bytes(smp, 8, 2, interlace))
    np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / "p.png")), smp.astype(np.uint8))


# ------------------------------------------------------------------------------------------------ JPEG
def _photo(h, w, seed):
    """Smooth content with some structure: what JPEG is for (pure noise would only measure the quantiser)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 9.0) * np.cos(y / 13.0), 128 + 90 * np.cos((x + y) / 17.0), 60 + (x * 180 // w)], axis=2)
    img[h // 3: h // 2, w // 4: w // 2] = (230, 40, 40)  # a hard edge in chroma
    return np.clip(img + rng.normal(0, 4, img.shape), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("progressive", [False, True])
@pytest.mark.parametrize("subsampling", [0, 1, 2])  # 4:4:4, 4:2:2, 4:2:0
def test_jpeg_against_pil(tmp_path, hip_lib, progressive, subsampling):
    """Files written by PIL (libjpeg): baseline and progressive, the three usual chroma samplings, sizes that are
    not multiples of the MCU, grey, restart intervals, optimised Huffman tables.  Compared with PIL's own decode:
    both sides approximate the same inverse DCT and use the same triangle filters for chroma, so they agree to a
    couple of levels (the level-exact result is not defined by the standard)."""
    from PIL import Image
    for (h, w), kw in (((37, 53), {}), ((64, 48), {"optimize": True}), ((17, 9), {"quality": 95}), ((40, 40), {"restart_marker_blocks": 3})):
        img = _photo(h, w, h)
        try:
            Image.fromarray(img, "RGB").save(tmp_path / "a.jpg", quality=kw.pop("quality", 85), progressive=progressive, subsampling=subsampling, **kw)
        except TypeError:
            continue  # a PIL without restart_marker_blocks
        want = np.asarray(Image.open(tmp_path / "a.jpg").convert("RGBA"), dtype=np.int32)
        got8 = np.rint(_linear_to_srgb8(load(tmp_path, "a.jpg"))).astype(np.int32)
        d = np.abs(got8[..., :3] - want[..., :3])
        assert d.max() <= 3 and d.mean() < 0.6, (h, w, kw, d.max(), d.mean())
        assert (got8[..., 3] == 255).all()
    grey = _photo(33, 41, 2)[..., 0]
    Image.fromarray(grey, "L").save(tmp_path / "g.jpg", quality=90, progressive=progressive)
    want = np.asarray(Image.open(tmp_path / "g.jpg").convert("RGBA"), dtype=np.int32)
    got8 = np.rint(_linear_to_srgb8(load(tmp_path, "g.jpg"))).astype(np.int32)
    assert np.abs(got8 - want).max() <= 2


def _linear_to_srgb8(lin):
    """Undo the loader's inverse gamma to get the decoder's 8-bit output back (alpha is linear already)."""
    v = np.asarray(lin, np.float64)
    out = np.where(v <= 0.0031308, v * 12.92, 1.055 * np.power(np.maximum(v, 0), 1 / 2.4) - 0.055) * 255.0
    out[..., 3] = v[..., 3] * 255.0
    return out


def test_jpeg_errors(tmp_path, hip_lib):
    from PIL import Image
    Image.fromarray(_photo(16, 16, 1), "RGB").save(tmp_path / "ok.jpg")
    good = (tmp_path / "ok.jpg").read_bytes()
    sof = good.index(b"\xff\xc0")
    cases = {"sig.jpg": b"XX" + good[2:], "noframe.jpg": good[:sof] + b"\xff\xd9", "twelve.jpg": good[:sof + 4] + b"\x0c" + good[sof + 5:],
             "arith.jpg": good[:sof + 1] + b"\xc9" + good[sof + 2:], "cmyk.jpg": None}
    Image.fromarray(np.zeros((8, 8, 4), np.uint8), "CMYK").save(tmp_path / "cmyk.jpg")
    for name, blob in cases.items():
        if blob is not None:
            (tmp_path / name).write_bytes(blob)
        with pytest.raises(api.ReneError) as e:
            load(tmp_path, name)
        assert e.value.code == -6 and "JPEG decode error" in str(e.value), (name, str(e.value))
