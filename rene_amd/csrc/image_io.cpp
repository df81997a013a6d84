// image_io.cpp -- texture / environment-map file decoders behind the pbrt loader (host only).
//
// Reference: load_image, rene/src/scene/intermediate_scene.rs:631-677 -- ".pfm" goes to rene's own parser
// (pbrt_loader.cpp), ".exr" to the `exr` crate (read_first_rgba_layer_from_file -> [r, g, b, a] f32 per
// pixel, row 0 on top), everything else to the `image` crate (decode -> RGBA8 -> inverse gamma on r, g, b).
// Both crates are registry dependencies absent from the checkout (exr 1.4.1, image 0.24.1), so the formats
// are restated from their published specifications:
//   * TGA (Truevision TGA 2.0): types 1 / 2 / 3 and their RLE forms 9 / 10 / 11; 8-bit grey, 16-bit
//     grey + alpha, 24 / 32-bit BGR(A), 8-bit indices into a 24 / 32-bit colour map; bottom-up unless
//     descriptor bit 5 is set;
//   * BMP (BITMAPINFOHEADER and later): 8-bit palette, 24-bit BGR, 32-bit BGRX (BI_RGB) or masks
//     (BI_BITFIELDS); bottom-up unless the height is negative;
//   * OpenEXR 2 scan-line images ("OpenEXR File Layout", "Technical Introduction to OpenEXR"): HALF / FLOAT /
//     UINT channels R, G, B, A (or Y) without subsampling; compression NONE, RLE, ZIPS, ZIP and PIZ (Huffman
//     + Haar wavelet + value table).  Tiled, deep and multi-part files, PXR24, B44 and DWA are refused.
//   * JPEG: see decode_jpeg at the end of this file.
// PNG lives in pbrt_loader.cpp.  Parity: unpinned (no reference test reads an image file); tests/test_images.py
// decodes files written by independent Python encoders and, where /root/reference is present, the
// reference's own PIZ-compressed EXR renders.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rene {

namespace {

struct Reader {
  const unsigned char* p;
  size_t n, at = 0;
  bool ok = true;
  Reader(const std::string& s) : p(reinterpret_cast<const unsigned char*>(s.data())), n(s.size()) {}
  bool need(size_t k) {
    if (k > n || at > n - k) ok = false;  // no wrap-around for offsets taken from the file
    return ok;
  }
  uint32_t u8() { return need(1) ? p[at++] : 0u; }
  uint32_t u16() {
    if (!need(2)) return 0;
    uint32_t v = p[at] | (p[at + 1] << 8);
    at += 2;
    return v;
  }
  uint32_t u32() {
    if (!need(4)) return 0;
    uint32_t v = p[at] | (p[at + 1] << 8) | (p[at + 2] << 16) | ((uint32_t)p[at + 3] << 24);
    at += 4;
    return v;
  }
  uint64_t u64() {
    uint64_t lo = u32();
    uint64_t hi = u32();
    return lo | (hi << 32);
  }
  std::string cstr() {  // null-terminated
    std::string s;
    while (need(1) && p[at] != 0) s.push_back((char)p[at++]);
    if (ok) at++;
    return s;
  }
};

}  // namespace

// ------------------------------------------------------------------------------------------------- TGA
bool decode_tga(const std::string& data, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgba, std::string& err) {
  Reader r(data);
  const uint32_t id_len = r.u8(), cmap_type = r.u8(), type = r.u8();
  const uint32_t cmap_start = r.u16(), cmap_len = r.u16(), cmap_bits = r.u8();
  r.u16();  // x origin
  r.u16();  // y origin
  w = r.u16();
  h = r.u16();
  const uint32_t depth = r.u8(), desc = r.u8();
  if (!r.ok) { err = "truncated header"; return false; }
  const bool rle = type == 9 || type == 10 || type == 11;
  const uint32_t base = rle ? type - 8 : type;
  if (base < 1 || base > 3) { err = "image type " + std::to_string(type); return false; }
  if (w == 0 || h == 0) { err = "empty image"; return false; }
  r.at += id_len;
  std::vector<unsigned char> cmap;
  uint32_t cmap_bytes = 0;
  if (cmap_type == 1) {
    if (cmap_bits != 24 && cmap_bits != 32) { err = "colour map with " + std::to_string(cmap_bits) + " bits per entry"; return false; }
    cmap_bytes = cmap_bits / 8;
    if (!r.need((size_t)cmap_len * cmap_bytes)) { err = "truncated colour map"; return false; }
    cmap.assign(r.p + r.at, r.p + r.at + (size_t)cmap_len * cmap_bytes);
    r.at += (size_t)cmap_len * cmap_bytes;
  } else if (base == 1) {
    err = "colour-mapped image without a colour map";
    return false;
  }
  uint32_t bpp;
  if (base == 1) {
    if (depth != 8) { err = "colour-map indices with " + std::to_string(depth) + " bits"; return false; }
    bpp = 1;
  } else if (base == 2) {
    if (depth != 24 && depth != 32) { err = "true colour with " + std::to_string(depth) + " bits per pixel"; return false; }
    bpp = depth / 8;
  } else {
    if (depth != 8 && depth != 16) { err = "grey with " + std::to_string(depth) + " bits per pixel"; return false; }
    bpp = depth / 8;
  }
  const size_t n_px = (size_t)w * h;
  std::vector<unsigned char> raw(n_px * bpp);
  if (!rle) {
    if (!r.need(raw.size())) { err = "truncated pixel data"; return false; }
    std::memcpy(raw.data(), r.p + r.at, raw.size());
  } else {
    size_t px = 0;
    while (px < n_px) {  // packets may run across scan lines
      const uint32_t head = r.u8();
      const size_t count = (head & 0x7fu) + 1u;
      if (!r.ok || px + count > n_px) { err = "bad run-length packet"; return false; }
      if (head & 0x80u) {
        if (!r.need(bpp)) { err = "truncated run-length packet"; return false; }
        for (size_t k = 0; k < count; ++k) std::memcpy(&raw[(px + k) * bpp], r.p + r.at, bpp);
        r.at += bpp;
      } else {
        if (!r.need(count * bpp)) { err = "truncated raw packet"; return false; }
        std::memcpy(&raw[px * bpp], r.p + r.at, count * bpp);
        r.at += count * bpp;
      }
      px += count;
    }
  }
  rgba.resize(n_px * 4);
  const bool top_down = (desc & 0x20u) != 0;
  for (uint32_t y = 0; y < h; ++y) {
    const uint32_t sy = top_down ? y : h - 1 - y;
    for (uint32_t x = 0; x < w; ++x) {
      const unsigned char* s = &raw[((size_t)sy * w + x) * bpp];
      unsigned char* d = &rgba[((size_t)y * w + x) * 4];
      if (base == 1) {
        const uint32_t idx = s[0];
        if (idx < cmap_start || idx - cmap_start >= cmap_len) { err = "colour-map index out of range"; return false; }
        const unsigned char* c = &cmap[(size_t)(idx - cmap_start) * cmap_bytes];
        d[0] = c[2]; d[1] = c[1]; d[2] = c[0]; d[3] = cmap_bytes == 4 ? c[3] : 255;
      } else if (base == 2) {
        d[0] = s[2]; d[1] = s[1]; d[2] = s[0]; d[3] = bpp == 4 ? s[3] : 255;
      } else {
        d[0] = d[1] = d[2] = s[0];
        d[3] = bpp == 2 ? s[1] : 255;
      }
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------- BMP
bool decode_bmp(const std::string& data, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgba, std::string& err) {
  Reader r(data);
  if (r.u8() != 'B' || r.u8() != 'M') { err = "signature"; return false; }
  r.u32();
  r.u32();
  const uint32_t off_bits = r.u32();
  const size_t dib_at = r.at;
  const uint32_t dib = r.u32();
  if (!r.ok || dib < 40) { err = "unsupported header"; return false; }
  const int32_t iw = (int32_t)r.u32(), ih = (int32_t)r.u32();
  r.u16();
  const uint32_t bpp = r.u16(), comp = r.u32();
  r.u32();
  r.u32();
  r.u32();
  uint32_t clr_used = r.u32();
  r.u32();
  if (!r.ok || iw <= 0 || ih == 0) { err = "bad dimensions"; return false; }
  w = (uint32_t)iw;
  h = (uint32_t)(ih < 0 ? -(int64_t)ih : ih);
  const bool top_down = ih < 0;
  uint32_t mask[4] = {0x00ff0000u, 0x0000ff00u, 0x000000ffu, 0u};
  if (comp == 3) {
    if (bpp != 32) { err = "bit fields with " + std::to_string(bpp) + " bits per pixel"; return false; }
    for (int k = 0; k < 3; ++k) mask[k] = r.u32();  // after a 40-byte header, or inside a V4 / V5 header
    if (dib >= 56) mask[3] = r.u32();
  } else if (comp != 0) {
    err = "compression " + std::to_string(comp);
    return false;
  }
  if (bpp != 8 && bpp != 24 && bpp != 32) { err = std::to_string(bpp) + " bits per pixel"; return false; }
  std::vector<unsigned char> pal;
  if (bpp == 8) {
    if (clr_used == 0) clr_used = 256;
    r.at = dib_at + dib;
    if (clr_used > 256 || !r.need((size_t)clr_used * 4)) { err = "palette"; return false; }
    pal.assign(r.p + r.at, r.p + r.at + (size_t)clr_used * 4);
  }
  const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
  if ((size_t)off_bits + stride * h > data.size()) { err = "truncated pixel data"; return false; }
  // channel masks must be contiguous runs of bits (0 = channel absent); shift + width never exceeds 32
  for (int k = 0; k < 4; ++k) {
    const uint32_t m = mask[k];
    if (m != 0) {
      const uint32_t low = m & (~m + 1u), run = m + low;  // adding the lowest set bit clears a contiguous run
      if ((run & m) != 0) { err = "non-contiguous channel mask"; return false; }
    }
  }
  auto field = [](uint32_t v, uint32_t m) -> unsigned char {
    if (m == 0) return 255;
    const int shift = __builtin_ctz(m), width = __builtin_popcount(m);
    const uint32_t x = (v & m) >> shift;
    return (unsigned char)(width >= 8 ? x >> (width - 8) : (x * 255u) / ((1u << width) - 1u));
  };
  rgba.resize((size_t)w * h * 4);
  for (uint32_t y = 0; y < h; ++y) {
    const unsigned char* row = r.p + off_bits + stride * (top_down ? y : h - 1 - y);
    for (uint32_t x = 0; x < w; ++x) {
      unsigned char* d = &rgba[((size_t)y * w + x) * 4];
      if (bpp == 8) {
        const uint32_t i = row[x];
        if ((size_t)i * 4 + 3 >= pal.size()) { err = "palette index out of range"; return false; }
        d[0] = pal[i * 4 + 2]; d[1] = pal[i * 4 + 1]; d[2] = pal[i * 4]; d[3] = 255;
      } else if (bpp == 24) {
        d[0] = row[3 * x + 2]; d[1] = row[3 * x + 1]; d[2] = row[3 * x]; d[3] = 255;
      } else {
        const uint32_t v = row[4 * x] | (row[4 * x + 1] << 8) | (row[4 * x + 2] << 16) | ((uint32_t)row[4 * x + 3] << 24);
        d[0] = field(v, mask[0]); d[1] = field(v, mask[1]); d[2] = field(v, mask[2]);
        d[3] = comp == 3 ? field(v, mask[3]) : 255;
      }
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------------------- EXR
namespace {

float half_to_float(uint16_t hbits) {
  const uint32_t s = (hbits >> 15) & 1u, e = (hbits >> 10) & 31u, m = hbits & 1023u;
  uint32_t out;
  if (e == 0) {
    if (m == 0) {
      out = s << 31;
    } else {  // subnormal: normalise
      int ee = -1;
      uint32_t mm = m;
      do {
        ee++;
        mm <<= 1;
      } while (!(mm & 1024u));
      out = (s << 31) | ((uint32_t)(127 - 15 - ee) << 23) | ((mm & 1023u) << 13);
    }
  } else if (e == 31) {
    out = (s << 31) | 0x7f800000u | (m << 13);
  } else {
    out = (s << 31) | ((e + 127 - 15) << 23) | (m << 13);
  }
  float f;
  std::memcpy(&f, &out, 4);
  return f;
}

// the byte shuffle ZIP and RLE blocks are stored in: differences, then even bytes | odd bytes
void unpredict_and_interleave(std::vector<unsigned char>& t, unsigned char* out) {
  const size_t n = t.size();
  for (size_t i = 1; i < n; ++i) t[i] = (unsigned char)(t[i - 1] + t[i] - 128);
  const size_t half = (n + 1) / 2;
  for (size_t i = 0; i < n; ++i) out[i] = (i & 1) ? t[half + i / 2] : t[i / 2];
}

bool inflate_block(const unsigned char* src, size_t n_src, std::vector<unsigned char>& dst) {
  uLongf len = (uLongf)dst.size();
  return uncompress(dst.data(), &len, src, (uLong)n_src) == Z_OK && len == dst.size();
}

bool unrle_block(const unsigned char* src, size_t n_src, std::vector<unsigned char>& dst) {
  size_t o = 0, i = 0;
  while (i < n_src) {
    const int c = (signed char)src[i++];
    if (c < 0) {
      const size_t k = (size_t)(-c);
      if (i + k > n_src || o + k > dst.size()) return false;
      std::memcpy(&dst[o], src + i, k);
      i += k;
      o += k;
    } else {
      const size_t k = (size_t)c + 1;
      if (i >= n_src || o + k > dst.size()) return false;
      std::memset(&dst[o], src[i++], k);
      o += k;
    }
  }
  return o == dst.size();
}

// ---- PIZ: canonical Huffman (ImfHuf), 2-D Haar wavelet with 14- or 16-bit arithmetic (ImfWav), value table ----
struct BitReader {
  const unsigned char* p;
  size_t n, at = 0;
  uint64_t acc = 0;
  int bits = 0;
  uint32_t get(int k) {  // MSB first; reads zeros past the end
    while (bits < k) {
      acc = (acc << 8) | (at < n ? p[at] : 0u);
      at++;
      bits += 8;
    }
    bits -= k;
    return (uint32_t)((acc >> bits) & ((1ull << k) - 1ull));
  }
};

bool huf_uncompress(const unsigned char* src, size_t n_src, std::vector<uint16_t>& out) {
  if (n_src == 0) return out.empty();
  if (n_src < 20) return false;
  auto u32 = [&](size_t a) { return (uint32_t)src[a] | (src[a + 1] << 8) | (src[a + 2] << 16) | ((uint32_t)src[a + 3] << 24); };
  const uint32_t im = u32(0), iM = u32(4), n_bits = u32(12);
  constexpr uint32_t ENC_SIZE = (1u << 16) + 1u;
  if (im >= ENC_SIZE || iM >= ENC_SIZE || im > iM) return false;
  // code lengths, six bits each; 59..62 = short runs of zeros (2..5), 63 = long run (6 + next eight bits)
  std::vector<unsigned char> len(ENC_SIZE, 0);
  BitReader br{src + 20, n_src - 20};
  for (uint32_t s = im; s <= iM;) {
    const uint32_t l = br.get(6);
    if (l == 63) {
      uint32_t run = br.get(8) + 6;
      if (s + run > iM + 1) return false;
      s += run;
    } else if (l >= 59) {
      uint32_t run = l - 59 + 2;
      if (s + run > iM + 1) return false;
      s += run;
    } else {
      len[s++] = (unsigned char)l;
    }
  }
  // canonical codes: shorter codes have the numerically larger prefixes (built from length 58 downwards)
  uint64_t count[59] = {0}, first[59] = {0};
  for (uint32_t s = im; s <= iM; ++s) count[len[s]]++;
  {
    uint64_t c = 0;
    for (int l = 58; l > 0; --l) {
      const uint64_t nc = (c + count[l]) >> 1;
      first[l] = c;
      c = nc;
    }
  }
  std::vector<uint32_t> offset(60, 0), symbols;
  for (int l = 1; l <= 58; ++l) offset[l + 1] = offset[l] + (uint32_t)count[l];
  symbols.resize(offset[59]);
  {
    std::vector<uint32_t> fill(offset.begin(), offset.end());
    for (uint32_t s = im; s <= iM; ++s)
      if (len[s]) symbols[fill[len[s]]++] = s;
  }
  // header: im, iM, table length in bytes, number of data bits, reserved; the data bits follow the table
  const size_t data_at = 20 + (size_t)u32(8);
  if (data_at > n_src || ((uint64_t)n_bits + 7) / 8 > n_src - data_at) return false;
  BitReader dr{src + data_at, n_src - data_at};
  uint64_t left = n_bits;
  size_t o = 0;
  const uint32_t rlc = iM;
  while (o < out.size()) {
    uint64_t code = 0;
    int l = 0;
    uint32_t sym = 0xffffffffu;
    while (l < 58) {
      if (left == 0) return false;
      code = (code << 1) | dr.get(1);
      left--;
      l++;
      if (count[l] && code >= first[l] && code - first[l] < count[l]) {
        sym = symbols[offset[l] + (uint32_t)(code - first[l])];
        break;
      }
    }
    if (sym == 0xffffffffu) return false;
    if (sym == rlc) {
      if (left < 8 || o == 0) return false;
      uint32_t run = dr.get(8);
      left -= 8;
      if (o + run > out.size()) return false;
      const uint16_t v = out[o - 1];
      while (run--) out[o++] = v;
    } else {
      out[o++] = (uint16_t)sym;
    }
  }
  return true;
}

inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int ls = (int16_t)l, hs = (int16_t)h;
  const int ai = ls + (hs & 1) + (hs >> 1);
  a = (uint16_t)(int16_t)ai;
  b = (uint16_t)(int16_t)(ai - hs);
}
inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
  const int m = l, d = h;
  const int bb = (m - (d >> 1)) & 0xffff;
  const int aa = (d + bb - (1 << 15)) & 0xffff;
  b = (uint16_t)bb;
  a = (uint16_t)aa;
}

void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
  const bool w14 = mx < (1 << 14);
  const int n = nx > ny ? ny : nx;
  int p = 1, p2;
  while (p <= n) p <<= 1;
  p >>= 1;
  p2 = p;
  p >>= 1;
  while (p >= 1) {
    uint16_t* py = in;
    uint16_t* ey = in + (ptrdiff_t)oy * (ny - p2);
    const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
    uint16_t i00, i01, i10, i11;
    for (; py <= ey; py += oy2) {
      uint16_t* px = py;
      uint16_t* ex = py + (ptrdiff_t)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t *p01 = px + ox1, *p10 = px + oy1, *p11 = p10 + ox1;
        if (w14) {
          wdec14(*px, *p10, i00, i10);
          wdec14(*p01, *p11, i01, i11);
          wdec14(i00, i01, *px, *p01);
          wdec14(i10, i11, *p10, *p11);
        } else {
          wdec16(*px, *p10, i00, i10);
          wdec16(*p01, *p11, i01, i11);
          wdec16(i00, i01, *px, *p01);
          wdec16(i10, i11, *p10, *p11);
        }
      }
      if (nx & p) {
        uint16_t* p10 = px + oy1;
        if (w14) wdec14(*px, *p10, i00, *p10);
        else wdec16(*px, *p10, i00, *p10);
        *px = i00;
      }
    }
    if (ny & p) {
      uint16_t* px = py;
      uint16_t* ex = py + (ptrdiff_t)ox * (nx - p2);
      for (; px <= ex; px += ox2) {
        uint16_t* p01 = px + ox1;
        if (w14) wdec14(*px, *p01, i00, *p01);
        else wdec16(*px, *p01, i00, *p01);
        *px = i00;
      }
    }
    p2 = p;
    p >>= 1;
  }
}

struct Channel {
  std::string name;
  uint32_t type;  // 0 uint, 1 half, 2 float
  uint32_t size;  // bytes per sample
};

bool unpiz_block(const unsigned char* src, size_t n_src, const std::vector<Channel>& ch, uint32_t nx, uint32_t ny, std::vector<unsigned char>& dst) {
  if (n_src < 4) return false;
  std::vector<unsigned char> bitmap(8192, 0);
  const uint32_t min_nz = src[0] | (src[1] << 8), max_nz = src[2] | (src[3] << 8);
  size_t at = 4;
  if (min_nz <= max_nz) {
    if (max_nz >= 8192 || at + (max_nz - min_nz + 1) > n_src) return false;
    std::memcpy(&bitmap[min_nz], src + at, max_nz - min_nz + 1);
    at += max_nz - min_nz + 1;
  }
  std::vector<uint16_t> lut(65536, 0);
  uint32_t k = 0;
  for (uint32_t i = 0; i < 65536; ++i)
    if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
  const uint16_t max_value = (uint16_t)(k - 1);
  if (at + 4 > n_src) return false;
  const uint32_t huf_len = src[at] | (src[at + 1] << 8) | (src[at + 2] << 16) | ((uint32_t)src[at + 3] << 24);
  at += 4;
  if (at + huf_len > n_src) return false;
  size_t n_words = 0;
  for (const Channel& c : ch) n_words += (size_t)nx * ny * (c.size / 2);
  if (n_words * 2 != dst.size()) return false;
  std::vector<uint16_t> tmp(n_words);
  if (!huf_uncompress(src + at, huf_len, tmp)) return false;
  std::vector<size_t> start(ch.size());
  size_t pos = 0;
  for (size_t c = 0; c < ch.size(); ++c) {
    start[c] = pos;
    const int words = (int)(ch[c].size / 2);
    for (int j = 0; j < words; ++j) wav2_decode(&tmp[pos + j], (int)nx, words, (int)ny, (int)nx * words, max_value);
    pos += (size_t)nx * ny * words;
  }
  for (uint16_t& v : tmp) v = lut[v];
  // back to the scan-line layout: per line, the channels one after the other
  unsigned char* o = dst.data();
  std::vector<size_t> run(start);
  for (uint32_t y = 0; y < ny; ++y)
    for (size_t c = 0; c < ch.size(); ++c) {
      const size_t words = (size_t)nx * (ch[c].size / 2);
      std::memcpy(o, &tmp[run[c]], words * 2);  // little-endian host
      o += words * 2;
      run[c] += words;
    }
  return true;
}

}  // namespace

bool decode_exr(const std::string& data, uint32_t& w, uint32_t& h, std::vector<float>& rgba, std::string& err) {
  Reader r(data);
  if (r.u32() != 20000630u) { err = "magic number"; return false; }
  const uint32_t version = r.u32();
  if ((version & 0xffu) != 2) { err = "file format version " + std::to_string(version & 0xffu); return false; }
  if (version & 0x200u) { err = "tiled images are not supported"; return false; }
  if (version & 0x1800u) { err = "deep / multi-part files are not supported"; return false; }
  std::vector<Channel> ch;
  int32_t dw[4] = {0, 0, -1, -1};
  uint32_t compression = 0, line_order = 0;
  bool have_channels = false, have_window = false;
  for (;;) {
    std::string name = r.cstr();
    if (!r.ok) { err = "truncated header"; return false; }
    if (name.empty()) break;
    std::string type = r.cstr();
    const uint32_t size = r.u32();
    if (!r.need(size)) { err = "truncated attribute " + name; return false; }
    const size_t end = r.at + size;
    if (name == "channels" && type == "chlist") {
      for (;;) {
        std::string cn = r.cstr();
        if (!r.ok || r.at > end) { err = "bad channel list"; return false; }
        if (cn.empty()) break;
        Channel c;
        c.name = cn;
        c.type = r.u32();
        r.u32();  // pLinear + reserved
        const uint32_t xs = r.u32(), ys = r.u32();
        if (c.type > 2) { err = "channel type " + std::to_string(c.type); return false; }
        if (xs != 1 || ys != 1) { err = "subsampled channel " + cn; return false; }
        c.size = c.type == 1 ? 2u : 4u;
        ch.push_back(c);
      }
      have_channels = true;
    } else if (name == "compression") {
      compression = r.u8();
    } else if (name == "dataWindow" && size == 16) {
      for (int k = 0; k < 4; ++k) dw[k] = (int32_t)r.u32();
      have_window = true;
    } else if (name == "lineOrder") {
      line_order = r.u8();
    }
    r.at = end;
  }
  if (!have_channels || !have_window || ch.empty()) { err = "header lacks channels / dataWindow"; return false; }
  if (dw[2] < dw[0] || dw[3] < dw[1]) { err = "empty data window"; return false; }
  const uint64_t W = (uint64_t)((int64_t)dw[2] - dw[0] + 1), H = (uint64_t)((int64_t)dw[3] - dw[1] + 1);
  if (W > 65536 || H > 65536) { err = "image too large"; return false; }
  w = (uint32_t)W;
  h = (uint32_t)H;
  uint32_t lines_per_block;
  switch (compression) {
    case 0: case 1: case 2: lines_per_block = 1; break;   // NONE, RLE, ZIPS
    case 3: lines_per_block = 16; break;                  // ZIP
    case 4: lines_per_block = 32; break;                  // PIZ
    default: err = "compression method " + std::to_string(compression) + " (PXR24 / B44 / DWA) is not supported"; return false;
  }
  (void)line_order;  // blocks carry their y coordinate; any order of blocks decodes the same
  int idx[4] = {-1, -1, -1, -1};  // R G B A, or Y for grey
  int y_idx = -1;
  size_t line_bytes = 0;
  std::vector<size_t> ch_off(ch.size());
  for (size_t c = 0; c < ch.size(); ++c) {
    ch_off[c] = line_bytes;
    line_bytes += (size_t)w * ch[c].size;
    if (ch[c].name == "R") idx[0] = (int)c;
    else if (ch[c].name == "G") idx[1] = (int)c;
    else if (ch[c].name == "B") idx[2] = (int)c;
    else if (ch[c].name == "A") idx[3] = (int)c;
    else if (ch[c].name == "Y") y_idx = (int)c;
  }
  if (idx[0] < 0 || idx[1] < 0 || idx[2] < 0) {
    if (y_idx < 0) { err = "no R, G, B (or Y) channels in the first layer"; return false; }
    idx[0] = idx[1] = idx[2] = y_idx;
  }
  const uint32_t n_blocks = (h + lines_per_block - 1) / lines_per_block;
  if (!r.need((size_t)n_blocks * 8)) { err = "truncated offset table"; return false; }
  std::vector<uint64_t> offsets(n_blocks);
  for (uint32_t b = 0; b < n_blocks; ++b) offsets[b] = r.u64();
  rgba.assign((size_t)w * h * 4, 0.0f);
  std::vector<unsigned char> raw, tmp;
  for (uint32_t b = 0; b < n_blocks; ++b) {
    if (offsets[b] >= data.size()) { err = "block offset outside the file"; return false; }
    r.at = (size_t)offsets[b];
    const int32_t y0 = (int32_t)r.u32();
    const uint32_t n_src = r.u32();
    if (!r.ok || !r.need(n_src)) { err = "truncated block"; return false; }
    if (y0 < dw[1] || y0 > dw[3]) { err = "block outside the data window"; return false; }
    const uint32_t first = (uint32_t)(y0 - dw[1]);
    const uint32_t ny = std::min(lines_per_block, h - first);
    raw.assign(line_bytes * ny, 0);
    const unsigned char* src = r.p + r.at;
    bool ok = true;
    if (n_src == raw.size() && compression != 4) {  // stored uncompressed (also what compressors fall back to)
      std::memcpy(raw.data(), src, raw.size());
    } else if (compression == 0) {
      ok = false;
    } else if (compression == 1) {
      tmp.assign(raw.size(), 0);
      ok = unrle_block(src, n_src, tmp);
      if (ok) unpredict_and_interleave(tmp, raw.data());
    } else if (compression == 2 || compression == 3) {
      tmp.assign(raw.size(), 0);
      ok = inflate_block(src, n_src, tmp);
      if (ok) unpredict_and_interleave(tmp, raw.data());
    } else {
      if (n_src == raw.size()) std::memcpy(raw.data(), src, raw.size());
      else ok = unpiz_block(src, n_src, ch, w, ny, raw);
    }
    if (!ok) { err = "corrupt block at y = " + std::to_string(y0); return false; }
    for (uint32_t ly = 0; ly < ny; ++ly) {
      const unsigned char* line = raw.data() + line_bytes * ly;
      float* out = &rgba[(size_t)(first + ly) * w * 4];
      for (int k = 0; k < 4; ++k) {
        if (idx[k] < 0) {
          for (uint32_t x = 0; x < w; ++x) out[4 * x + k] = 1.0f;  // no alpha channel: opaque
          continue;
        }
        const Channel& c = ch[idx[k]];
        const unsigned char* s = line + ch_off[idx[k]];
        for (uint32_t x = 0; x < w; ++x) {
          float v;
          if (c.type == 1) {
            uint16_t hb;
            std::memcpy(&hb, s + 2 * x, 2);
            v = half_to_float(hb);
          } else if (c.type == 2) {
            std::memcpy(&v, s + 4 * x, 4);
          } else {
            uint32_t u;
            std::memcpy(&u, s + 4 * x, 4);
            v = (float)u;
          }
          out[4 * x + k] = v;
        }
      }
    }
  }
  return true;
}

}  // namespace rene

// ------------------------------------------------------------------------------------------------- JPEG
// ITU-T T.81 (JPEG), 8-bit Huffman-coded frames: baseline / extended sequential (SOF0, SOF1) and progressive
// (SOF2: spectral selection + successive approximation), one (grey) or three (YCbCr, or RGB when an Adobe APP14
// marker says so) components, restart intervals, any sampling factors up to 2 x 2.  The inverse DCT is evaluated in
// floating point (the transform libjpeg's and the jpeg-decoder crate's fixed-point kernels approximate to within one
// level); chroma is brought to full resolution with the triangle filters both use for 2:1 horizontal and 2 x 2
// sampling ("fancy upsampling"), by replication otherwise.  Arithmetic-coded, lossless, 12-bit and four-component
// files are refused.
namespace rene {

namespace {

const unsigned char kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool present = false;
  int mincode[17], maxcode[18], valptr[17];
  unsigned char vals[256];
};

struct JComp {
  int id = 0, h = 1, v = 1, tq = 0;
  int bw = 0, bh = 0;  // blocks per line / column (padded to the MCU grid)
  int pw = 0, ph = 0;  // sample dimensions of the component (ceil(W * h / hmax))
  std::vector<int16_t> coef;
  int pred = 0, td = 0, ta = 0;
};

struct JBits {
  const unsigned char* p;
  size_t n, at;
  uint32_t acc = 0;
  int bits = 0;
  bool hit_marker = false;
  void fill() {
    while (bits <= 24) {
      unsigned c = 0;
      if (!hit_marker && at < n) {
        c = p[at];
        if (c == 0xff) {
          unsigned d = at + 1 < n ? p[at + 1] : 0xd9u;
          if (d == 0) at += 2;               // stuffed zero
          else { hit_marker = true; c = 0; }  // a marker ends the entropy-coded segment: feed zeros
        } else {
          at++;
        }
      }
      acc |= c << (24 - bits);
      bits += 8;
    }
  }
  int get(int k) {
    if (k == 0) return 0;
    fill();
    int v = (int)(acc >> (32 - k));
    acc <<= k;
    bits -= k;
    return v;
  }
  int bit() { return get(1); }
  void reset() { acc = 0; bits = 0; hit_marker = false; }
};

int jdecode(JBits& b, const Huff& h) {
  int code = 0;
  for (int l = 1; l <= 16; ++l) {
    code = (code << 1) | b.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}
inline int jextend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

void idct8x8(const float* in, unsigned char* out, int stride) {
  static float c[8][8];
  static bool init = false;
  if (!init) {
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u) c[x][u] = (float)((u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
    init = true;
  }
  float tmp[64];
  for (int y = 0; y < 8; ++y)      // rows: over u
    for (int x = 0; x < 8; ++x) {
      float s = 0.0f;
      for (int u = 0; u < 8; ++u) s += c[x][u] * in[y * 8 + u];
      tmp[y * 8 + x] = s;
    }
  for (int x = 0; x < 8; ++x)
    for (int y = 0; y < 8; ++y) {
      float s = 0.0f;
      for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
      int q = (int)std::floor(s + 128.5f);
      out[y * stride + x] = (unsigned char)(q < 0 ? 0 : q > 255 ? 255 : q);
    }
}

}  // namespace

bool decode_jpeg(const std::string& data, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgba, std::string& err) {
  const unsigned char* p = reinterpret_cast<const unsigned char*>(data.data());
  const size_t n = data.size();
  if (n < 4 || p[0] != 0xff || p[1] != 0xd8) { err = "signature"; return false; }
  uint16_t qt[4][64] = {};
  bool have_qt[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  std::vector<JComp> comp;
  bool progressive = false, have_frame = false, done = false;
  int hmax = 1, vmax = 1, restart = 0, adobe_transform = -1;
  int mcux = 0, mcuy = 0;
  size_t at = 2;
  auto be16 = [&](size_t a) { return (unsigned)((p[a] << 8) | p[a + 1]); };
  while (!done) {
    while (at < n && p[at] != 0xff) at++;  // garbage between segments is tolerated like the decoders do
    while (at < n && p[at] == 0xff) at++;
    if (at >= n) break;
    const unsigned m = p[at++];
    if (m == 0xd9) break;
    if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
    if (at + 2 > n) { err = "truncated segment"; return false; }
    const size_t len = be16(at);
    if (len < 2 || at + len > n) { err = "truncated segment"; return false; }
    const size_t seg = at + 2, seg_end = at + len;
    if (m == 0xdb) {  // DQT
      for (size_t a = seg; a < seg_end;) {
        const int pq = p[a] >> 4, tq = p[a] & 15;
        a++;
        if (tq > 3 || a + (pq ? 128 : 64) > seg_end) { err = "quantisation table"; return false; }
        for (int k = 0; k < 64; ++k) {
          qt[tq][k] = pq ? (uint16_t)be16(a) : p[a];
          a += pq ? 2 : 1;
        }
        have_qt[tq] = true;
      }
    } else if (m == 0xc4) {  // DHT
      for (size_t a = seg; a < seg_end;) {
        if (a + 17 > seg_end) { err = "Huffman table"; return false; }
        const int tc = p[a] >> 4, th = p[a] & 15;
        if (tc > 1 || th > 3) { err = "Huffman table"; return false; }
        Huff& t = tc ? ac[th] : dc[th];
        int counts[17] = {0}, total = 0;
        for (int l = 1; l <= 16; ++l) total += counts[l] = p[a + l];
        a += 17;
        if (total > 256 || a + total > seg_end) { err = "Huffman table"; return false; }
        std::memcpy(t.vals, p + a, total);
        a += total;
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
          t.valptr[l] = k;
          t.mincode[l] = code;
          code += counts[l];
          k += counts[l];
          t.maxcode[l] = counts[l] ? code - 1 : -1;
          code <<= 1;
        }
        t.present = true;
      }
    } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // SOF
      if (have_frame) { err = "second frame"; return false; }
      if (len < 8) { err = "frame header"; return false; }
      if (p[seg] != 8) { err = std::to_string(p[seg]) + "-bit samples"; return false; }
      h = be16(seg + 1);
      w = be16(seg + 3);
      const int nc = p[seg + 5];
      if (!w || !h) { err = "empty image"; return false; }
      if (nc != 1 && nc != 3) { err = std::to_string(nc) + " components"; return false; }
      if (seg + 6 + 3 * (size_t)nc > seg_end) { err = "frame header"; return false; }
      comp.resize(nc);
      for (int c = 0; c < nc; ++c) {
        comp[c].id = p[seg + 6 + 3 * c];
        comp[c].h = p[seg + 7 + 3 * c] >> 4;
        comp[c].v = p[seg + 7 + 3 * c] & 15;
        comp[c].tq = p[seg + 8 + 3 * c];
        if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) { err = "sampling factors"; return false; }
        hmax = std::max(hmax, comp[c].h);
        vmax = std::max(vmax, comp[c].v);
      }
      mcux = ((int)w + 8 * hmax - 1) / (8 * hmax);
      mcuy = ((int)h + 8 * vmax - 1) / (8 * vmax);
      for (JComp& c : comp) {
        c.bw = mcux * c.h;
        c.bh = mcuy * c.v;
        c.pw = ((int)w * c.h + hmax - 1) / hmax;
        c.ph = ((int)h * c.v + vmax - 1) / vmax;
        c.coef.assign((size_t)c.bw * c.bh * 64, 0);
      }
      progressive = m == 0xc2;
      have_frame = true;
    } else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) {
      err = "lossless / hierarchical / arithmetic-coded JPEG";
      return false;
    } else if (m == 0xdd) {  // DRI
      restart = (int)be16(seg);
    } else if (m == 0xee) {  // Adobe
      if (len >= 14 && std::memcmp(p + seg, "Adobe", 5) == 0) adobe_transform = p[seg + 11];
    } else if (m == 0xda) {  // SOS + entropy-coded data
      if (!have_frame) { err = "scan before frame"; return false; }
      const int ns = p[seg];
      if (ns < 1 || ns > (int)comp.size() || seg + 1 + 2 * (size_t)ns + 3 > seg_end) { err = "scan header"; return false; }
      int order[3];
      for (int k = 0; k < ns; ++k) {
        int id = p[seg + 1 + 2 * k], ci = -1;
        for (size_t c = 0; c < comp.size(); ++c)
          if (comp[c].id == id) ci = (int)c;
        if (ci < 0) { err = "scan component"; return false; }
        order[k] = ci;
        comp[ci].td = p[seg + 2 + 2 * k] >> 4;
        comp[ci].ta = p[seg + 2 + 2 * k] & 15;
        if (comp[ci].td > 3 || comp[ci].ta > 3) { err = "scan tables"; return false; }
      }
      int Ss = p[seg + 1 + 2 * ns], Se = p[seg + 2 + 2 * ns], Ah = p[seg + 3 + 2 * ns] >> 4, Al = p[seg + 3 + 2 * ns] & 15;
      if (!progressive) { Ss = 0; Se = 63; Ah = Al = 0; }
      if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0 && progressive) || (Ss > 0 && ns != 1) || Al > 13) { err = "spectral selection"; return false; }
      for (int k = 0; k < ns; ++k) {
        const JComp& c = comp[order[k]];
        const bool needs_dc = Ss == 0 && Ah == 0;                  // baseline blocks, first DC scans
        const bool needs_ac = progressive ? Ss > 0 : true;         // baseline blocks, every AC scan
        if ((needs_dc && !dc[c.td].present) || (needs_ac && !ac[c.ta].present)) {
          err = "missing Huffman table";
          return false;
        }
      }
      JBits b{p, n, seg_end};
      int eobrun = 0, to_restart = restart;
      for (JComp& c : comp) c.pred = 0;
      bool bad = false;
      auto block = [&](JComp& c, int bx, int by) {
        int16_t* q = &c.coef[((size_t)by * c.bw + bx) * 64];
        if (!progressive) {
          int t = jdecode(b, dc[c.td]);
          if (t < 0 || t > 11) { bad = true; return; }
          c.pred += jextend(b.get(t), t);
          q[0] = (int16_t)c.pred;
          for (int k = 1; k < 64;) {
            int rs = jdecode(b, ac[c.ta]);
            if (rs < 0) { bad = true; return; }
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
              if (r == 15) { k += 16; continue; }
              break;
            }
            k += r;
            if (k > 63) { bad = true; return; }
            q[kZigZag[k]] = (int16_t)jextend(b.get(s), s);
            k++;
          }
        } else if (Ss == 0) {
          if (Ah == 0) {
            int t = jdecode(b, dc[c.td]);
            if (t < 0 || t > 11) { bad = true; return; }
            c.pred += jextend(b.get(t), t);
            q[0] = (int16_t)(c.pred * (1 << Al));
          } else if (b.bit()) {
            q[0] |= (int16_t)(1 << Al);
          }
        } else if (Ah == 0) {
          if (eobrun > 0) { eobrun--; return; }
          for (int k = Ss; k <= Se;) {
            int rs = jdecode(b, ac[c.ta]);
            if (rs < 0) { bad = true; return; }
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
              if (r < 15) {
                eobrun = (1 << r) - 1 + (r ? b.get(r) : 0);
                break;
              }
              k += 16;
              continue;
            }
            k += r;
            if (k > Se) { bad = true; return; }
            q[kZigZag[k]] = (int16_t)(jextend(b.get(s), s) * (1 << Al));
            k++;
          }
        } else {  // AC refinement, T.81 G.1.2.3
          const int p1 = 1 << Al, m1 = -(1 << Al);
          int k = Ss;
          auto refine = [&](int16_t& v) {
            if (b.bit() && (v & p1) == 0) v = (int16_t)(v >= 0 ? v + p1 : v + m1);
          };
          if (eobrun == 0) {
            for (; k <= Se; ++k) {
              int rs = jdecode(b, ac[c.ta]);
              if (rs < 0) { bad = true; return; }
              int r = rs >> 4, s = rs & 15, value = 0;
              if (s) {
                value = b.bit() ? p1 : m1;
              } else if (r != 15) {
                eobrun = (1 << r) + (r ? b.get(r) : 0);
                break;
              }
              while (k <= Se) {
                int16_t& v = q[kZigZag[k]];
                if (v != 0) refine(v);
                else if (--r < 0) break;
                k++;
              }
              if (s && k <= Se) q[kZigZag[k]] = (int16_t)value;
            }
          }
          if (eobrun > 0) {
            for (; k <= Se; ++k) {
              int16_t& v = q[kZigZag[k]];
              if (v != 0) refine(v);
            }
            eobrun--;
          }
        }
      };
      auto at_restart = [&]() -> bool {  // true when the data ended
        if (!restart || --to_restart > 0) return false;
        // byte-align, swallow the RSTn marker, reset the predictors
        b.reset();
        size_t a = b.at;
        while (a + 1 < n && !(p[a] == 0xff && p[a + 1] >= 0xd0 && p[a + 1] <= 0xd7)) {
          if (p[a] == 0xff && p[a + 1] != 0 && p[a + 1] != 0xff) return true;  // some other marker: the scan is over
          a++;
        }
        if (a + 1 >= n) return true;
        b.at = a + 2;
        for (JComp& c : comp) c.pred = 0;
        eobrun = 0;
        to_restart = restart;
        return false;
      };
      if (ns == 1) {
        JComp& c = comp[order[0]];
        const int nbx = (c.pw + 7) / 8, nby = (c.ph + 7) / 8;  // a non-interleaved scan covers the component's own blocks
        bool over = false;
        for (int by = 0; by < nby && !bad && !over; ++by)
          for (int bx = 0; bx < nbx && !bad && !over; ++bx) {
            block(c, bx, by);
            over = at_restart() && !(by == nby - 1 && bx == nbx - 1);
          }
      } else {
        bool over = false;
        for (int my = 0; my < mcuy && !bad && !over; ++my)
          for (int mx = 0; mx < mcux && !bad && !over; ++mx) {
            for (int k = 0; k < ns && !bad; ++k) {
              JComp& c = comp[order[k]];
              for (int v = 0; v < c.v && !bad; ++v)
                for (int hh = 0; hh < c.h && !bad; ++hh) block(c, mx * c.h + hh, my * c.v + v);
            }
            over = at_restart() && !(my == mcuy - 1 && mx == mcux - 1);
          }
      }
      if (bad) { err = "corrupt entropy-coded data"; return false; }
      // continue after the entropy-coded segment: at the marker that ended it
      b.reset();
      at = b.at;
      while (at + 1 < n && !(p[at] == 0xff && p[at + 1] != 0 && p[at + 1] != 0xff && !(p[at + 1] >= 0xd0 && p[at + 1] <= 0xd7))) at++;
      continue;
    }
    at = seg_end;
  }
  if (!have_frame) { err = "no frame"; return false; }
  // dequantise + inverse DCT into component planes (padded to whole blocks)
  std::vector<std::vector<unsigned char>> plane(comp.size());
  for (size_t ci = 0; ci < comp.size(); ++ci) {
    JComp& c = comp[ci];
    if (!have_qt[c.tq]) { err = "missing quantisation table"; return false; }
    const int stride = c.bw * 8;
    plane[ci].assign((size_t)stride * c.bh * 8, 0);
    float blk[64];
    for (int by = 0; by < c.bh; ++by)
      for (int bx = 0; bx < c.bw; ++bx) {
        const int16_t* q = &c.coef[((size_t)by * c.bw + bx) * 64];
        for (int k = 0; k < 64; ++k) blk[kZigZag[k]] = (float)q[kZigZag[k]] * (float)qt[c.tq][k];
        idct8x8(blk, &plane[ci][(size_t)by * 8 * stride + bx * 8], stride);
      }
  }
  // chroma to full resolution
  auto sample = [&](size_t ci, int x, int y) -> int {  // clamped fetch inside the component's true extent
    const JComp& c = comp[ci];
    x = x < 0 ? 0 : x >= c.pw ? c.pw - 1 : x;
    y = y < 0 ? 0 : y >= c.ph ? c.ph - 1 : y;
    return plane[ci][(size_t)y * c.bw * 8 + x];
  };
  std::vector<std::vector<unsigned char>> full(comp.size());
  for (size_t ci = 0; ci < comp.size(); ++ci) {
    const JComp& c = comp[ci];
    full[ci].resize((size_t)w * h);
    const int fx = hmax / c.h, fy = vmax / c.v;
    for (uint32_t y = 0; y < h; ++y)
      for (uint32_t x = 0; x < w; ++x) {
        int v;
        if (fx == 1 && fy == 1) {
          v = sample(ci, (int)x, (int)y);
        } else if (fx == 2 && fy == 1) {  // h2v1 triangle filter: 3/4 nearer, 1/4 farther, alternating rounding
          const int i = (int)x >> 1;
          if ((int)x == 0 || (int)x == 2 * c.pw - 1) v = sample(ci, i, (int)y);
          else v = (x & 1) ? (3 * sample(ci, i, (int)y) + sample(ci, i + 1, (int)y) + 2) >> 2 : (3 * sample(ci, i, (int)y) + sample(ci, i - 1, (int)y) + 1) >> 2;
        } else if (fx == 2 && fy == 2) {  // h2v2: the same filter in both directions, sixteenths
          const int i = (int)x >> 1, j = (int)y >> 1, jn = (y & 1) ? j + 1 : j - 1;
          auto col = [&](int ii) { return 3 * sample(ci, ii, j) + sample(ci, ii, jn); };
          const int cur = col(i);
          if ((int)x == 0 || (int)x == 2 * c.pw - 1) v = (4 * cur + 8) >> 4;
          else v = (x & 1) ? (3 * cur + col(i + 1) + 7) >> 4 : (3 * cur + col(i - 1) + 8) >> 4;
        } else {
          v = sample(ci, (int)x / fx, (int)y / fy);
        }
        full[ci][(size_t)y * w + x] = (unsigned char)v;
      }
  }
  rgba.resize((size_t)w * h * 4);
  const bool ycc = comp.size() == 3 && adobe_transform != 0;
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    unsigned char* d = &rgba[i * 4];
    if (comp.size() == 1) {
      d[0] = d[1] = d[2] = full[0][i];
    } else if (!ycc) {
      d[0] = full[0][i]; d[1] = full[1][i]; d[2] = full[2][i];
    } else {
      const float Y = full[0][i], cb = (float)full[1][i] - 128.0f, cr = (float)full[2][i] - 128.0f;
      auto clamp8 = [](float f) { int q = (int)std::floor(f + 0.5f); return (unsigned char)(q < 0 ? 0 : q > 255 ? 255 : q); };
      d[0] = clamp8(Y + 1.402f * cr);
      d[1] = clamp8(Y - 0.344136f * cb - 0.714136f * cr);
      d[2] = clamp8(Y + 1.772f * cb);
    }
    d[3] = 255;
  }
  return true;
}

}  // namespace rene
