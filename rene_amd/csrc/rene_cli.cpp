// rene_cli.cpp -- `rene-hip`: rene's command line (rene/src/main.rs:47-207, 1613-1687) over the C ABI.
//
//   rene-hip <scene.pbrt> [--aov-normal PATH] [--aov-albedo PATH] [--denoiser none|optix|oidn]
//            [--dump-module PATH]                       <- the reference's five options (main.rs:54-71)
//            [--spp N] [--seed S] [--width W] [--height H] [--gpus G] [--batch B] [--out PATH] [--frame-groups]
//
// The reference hard-codes 5000 samples in batches of 100 (main.rs:80-81); --spp / --batch default
// to those.  Output name = Film "filename" (+ ".png" when it ends in ".exr", main.rs:1651-1656).
// --gpus G renders on G devices from this one process: one context per device, 32x32 tiles dealt
// round-robin, the per-device images summed on the host (each pixel has exactly one owner).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rene_hip.h"

namespace {

// ---- minimal PNG writer: 8-bit RGB, zlib "stored" blocks (no compression library needed) -----------
uint32_t crc_table[256];
void crc_init() {
  for (uint32_t n = 0; n < 256; ++n) {
    uint32_t c = n;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
    crc_table[n] = c;
  }
}
uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0xffffffffu) {
  for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c;
}
void be32(std::vector<uint8_t>& v, uint32_t x) {
  for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}
void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
  be32(out, (uint32_t)data.size());
  std::vector<uint8_t> td(type, type + 4);
  td.insert(td.end(), data.begin(), data.end());
  out.insert(out.end(), td.begin(), td.end());
  be32(out, crc32(td.data(), td.size()) ^ 0xffffffffu);
}
bool write_png(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h) {
  crc_init();
  std::vector<uint8_t> raw;
  raw.reserve((size_t)h * (3 * w + 1));
  for (uint32_t y = 0; y < h; ++y) {
    raw.push_back(0);  // filter: none
    raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  size_t pos = 0;
  uint32_t a = 1, b = 0;
  for (uint8_t c : raw) {
    a = (a + c) % 65521u;
    b = (b + a) % 65521u;
  }
  while (pos < raw.size() || raw.empty()) {
    size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back((uint8_t)(n & 0xff));
    z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xff));
    z.push_back((uint8_t)((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
    if (raw.empty()) break;
  }
  be32(z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  be32(ihdr, w);
  be32(ihdr, h);
  const uint8_t tail[5] = {8, 2, 0, 0, 0};
  ihdr.insert(ihdr.end(), tail, tail + 5);
  chunk(out, "IHDR", ihdr);
  chunk(out, "IDAT", z);
  chunk(out, "IEND", {});
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  std::fclose(f);
  return ok;
}

int die(const char* what) {
  std::fprintf(stderr, "\nrene-hip: %s: %s\n", what, rene_last_error());
  return 1;
}

void usage() {
  std::fprintf(stderr,
               "usage: rene-hip <pbrt file> [--aov-normal PATH] [--aov-albedo PATH] [--denoiser none|optix|oidn]\n"
               "                [--dump-module PATH] [--spp N] [--seed S] [--width W] [--height H] [--gpus G]\n"
               "                [--batch B] [--out PATH] [--frame-groups]\n");
}

}  // namespace

int main(int argc, char** argv) {
  auto t_start = std::chrono::steady_clock::now();
  std::string pbrt_path, aov_normal, aov_albedo, denoiser = "none", dump_module, out_override;
  uint32_t spp = 5000, batch = 100, seed = RENE_DEFAULT_SEED, width = 0, height = 0, gpus = 1;
  bool frame_groups = false;  // --frame-groups (round 3's opt-in): accepted and ignored, every context renders eight frame chains per pixel (ABI v5)
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto val = [&](const char* name) -> const char* {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "rene-hip: %s needs a value\n", name);
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "--aov-normal") aov_normal = val("--aov-normal");
    else if (a == "--aov-albedo") aov_albedo = val("--aov-albedo");
    else if (a == "--denoiser") denoiser = val("--denoiser");
    else if (a == "--dump-module") dump_module = val("--dump-module");
    else if (a == "--spp") spp = (uint32_t)std::strtoul(val("--spp"), nullptr, 0);
    else if (a == "--batch") batch = (uint32_t)std::strtoul(val("--batch"), nullptr, 0);
    else if (a == "--seed") seed = (uint32_t)std::strtoul(val("--seed"), nullptr, 0);
    else if (a == "--width") width = (uint32_t)std::strtoul(val("--width"), nullptr, 0);
    else if (a == "--height") height = (uint32_t)std::strtoul(val("--height"), nullptr, 0);
    else if (a == "--gpus") gpus = (uint32_t)std::strtoul(val("--gpus"), nullptr, 0);
    else if (a == "--out") out_override = val("--out");
    else if (a == "--frame-groups") frame_groups = true;
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "rene-hip: unknown option %s\n", a.c_str()); usage(); return 2; }
    else pbrt_path = a;
  }
  if (denoiser != "none" && denoiser != "optix" && denoiser != "oidn") {
    std::fprintf(stderr, "rene-hip: invalid --denoiser %s\n", denoiser.c_str());
    return 2;
  }
  if (denoiser != "none")  // main.rs:86-98: warn and ignore when not built in
    std::fprintf(stderr, "WARN %s denoiser was enabled but this build has no denoiser. Ignore.\n", denoiser.c_str());
  if (!dump_module.empty()) {  // main.rs:100-106 dumps the SPIR-V module; here: the gfx950 code object
    std::string self = argv[0];
    size_t slash = self.rfind('/');
    std::string co = (slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/rene_kernels.co";
    FILE* in = std::fopen(co.c_str(), "rb");
    if (!in) { std::fprintf(stderr, "rene-hip: cannot open %s\n", co.c_str()); return 1; }
    FILE* o = std::fopen(dump_module.c_str(), "wb");
    if (!o) { std::fclose(in); std::fprintf(stderr, "rene-hip: cannot write %s\n", dump_module.c_str()); return 1; }
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, in)) > 0) std::fwrite(buf, 1, n, o);
    std::fclose(in);
    std::fclose(o);
    return 0;
  }
  if (pbrt_path.empty() || spp == 0 || batch == 0 || gpus == 0) { usage(); return 2; }

  rene_scene* scene = nullptr;
  if (rene_scene_load_pbrt(pbrt_path.c_str(), &scene) != RENE_OK) {
    std::printf("%s\n", rene_last_error());  // main.rs:199-205 prints the error and returns
    return 1;
  }
  rene_scene_desc desc = *rene_scene_get_desc(scene);
  if ((width && width != desc.xresolution) || (height && height != desc.yresolution)) {
    // additive override of the Film size: rescale the projection's x axis to the new aspect
    // (projection_inv = diag(aspect*tan, tan, ..), scene.rs:163-164)
    uint32_t nw = width ? width : desc.xresolution, nh = height ? height : desc.yresolution;
    float old_aspect = (float)desc.xresolution / (float)desc.yresolution, new_aspect = (float)nw / (float)nh;
    desc.uniform.projection_inv[0] *= new_aspect / old_aspect;
    desc.xresolution = nw;
    desc.yresolution = nh;
  }
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t).count();
  };
  std::fprintf(stderr, "INFO Scene parsed (%lld ms)\n", ms_since(t_start));

  auto t_load = std::chrono::steady_clock::now();
  std::vector<rene_ctx*> ctx(gpus, nullptr);
  for (uint32_t g = 0; g < gpus; ++g) {
    rene_opts o{};
    o.struct_size = sizeof(o);
    o.seed = seed;
    o.device = (int32_t)g;
    o.shard_mode = RENE_SHARD_TILES;
    o.shard_rank = g;
    o.shard_count = gpus;
    if (frame_groups) o.flags |= RENE_FLAG_FRAME_GROUPS;
    // all three layers are accumulated whether or not --aov-* asks for the files, like the reference's raygen
    // (lib.rs:229-232); RENE_FLAG_NO_AOV would save little and its Matte item-loop kernel happens to be the slower one
    if (rene_create(&desc, &o, &ctx[g]) != RENE_OK) return die("rene_create");
  }
  std::fprintf(stderr, "INFO Scene loaded (%lld ms)\n", ms_since(t_load));

  uint32_t sampled = 0;
  const auto t_render = std::chrono::steady_clock::now();
  while (sampled < spp) {  // main.rs:1315-1397
    uint32_t n = std::min(spp - sampled, batch);
    auto now = std::chrono::steady_clock::now();
    for (uint32_t g = 0; g < gpus; ++g)
      if (rene_render(ctx[g], sampled, n) != RENE_OK) return die("rene_render");
    sampled += n;
    // queue_wait_idle after every batch, main.rs:1389: the progress line then says what the batch took.  (A batch is one launch,
    // and a launch ends on the longest paths of its last work items: --batch N with N = --spp renders the job as ONE launch,
    // a few per cent faster than fifty batches of 100.)
    for (uint32_t g = 0; g < gpus; ++g)
      if (rene_sync(ctx[g]) != RENE_OK) return die("rene_sync");
    std::fprintf(stderr, "\rSamples: %u / %u (%lld ms)", sampled, spp, ms_since(now));
  }
  std::fprintf(stderr, "\n");
  const double render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_render).count();

  // The exchange step of a multi-GPU render: every GPU sends the 32x32 tiles it owns to GPU 0 over xGMI (RCCL inside
  // the library, rene_gather_tiles; one process, one communicator over the `gpus` contexts).  Only where RCCL is
  // missing do the per-GPU images meet on the host instead.
  bool gathered = false;
  if (gpus > 1) {
    if (rene_comm_init_all(ctx.data(), (int)gpus) == RENE_OK) {
      bool ok = rene_comm_group_begin() == RENE_OK;
      for (uint32_t g = 0; g < gpus && ok; ++g) ok = rene_gather_tiles(ctx[g], 0) == RENE_OK;
      ok = rene_comm_group_end() == RENE_OK && ok;
      if (!ok) return die("rene_gather_tiles");
      for (uint32_t g = 0; g < gpus; ++g)
        if (rene_sync(ctx[g]) != RENE_OK) return die("rene_sync");
      gathered = true;
    } else {
      std::fprintf(stderr, "WARN %s -- summing the per-GPU images on the host\n", rene_last_error());
    }
  }
  const size_t n_px = (size_t)desc.xresolution * desc.yresolution;
  auto layer = [&](int l, std::vector<float>& sum) -> bool {
    sum.assign(n_px * 3, 0.0f);
    if (gathered || gpus == 1) return rene_download(ctx[0], l, 3, sum.data(), sum.size()) == RENE_OK;
    std::vector<float> part(n_px * 3);
    for (uint32_t g = 0; g < gpus; ++g) {
      if (rene_download(ctx[g], l, 3, part.data(), part.size()) != RENE_OK) return false;
      for (size_t i = 0; i < sum.size(); ++i) sum[i] += part[i];
    }
    return true;
  };
  uint64_t rays = 0;
  double kernel_ms = 0.0;
  for (uint32_t g = 0; g < gpus; ++g) {
    rene_stats st;
    if (rene_get_stats(ctx[g], &st) != RENE_OK) return die("rene_get_stats");  // e.g. a hand-off timed out: no image is written
    rays += st.rays_closest + st.rays_shadow + st.rays_emitter;
    kernel_ms = std::max(kernel_ms, st.kernel_ms);
  }
  std::vector<float> img;
  std::vector<uint8_t> rgb(n_px * 3);
  if (!layer(RENE_LAYER_RADIANCE, img)) return die("rene_download");
  rene_to_rgb8(img.data(), img.size(), spp, rgb.data());  // average + to_rgb8, main.rs:1621, 1649
  std::string filename = out_override.empty() ? rene_scene_film_filename(scene) : out_override;
  if (filename.size() >= 4 && filename.compare(filename.size() - 4, 4, ".exr") == 0) {
    std::fprintf(stderr, "INFO .exr output is not yet supported. Save as .png\n");  // main.rs:1651-1656
    filename += ".png";
  }
  if (!write_png(filename, rgb.data(), desc.xresolution, desc.yresolution)) {
    std::fprintf(stderr, "rene-hip: cannot write %s\n", filename.c_str());
    return 1;
  }
  if (!aov_normal.empty()) {  // main.rs:1667-1676
    if (!layer(RENE_LAYER_NORMAL, img)) return die("rene_download");
    rene_to_aov8(img.data(), img.size(), spp, 1, rgb.data());
    if (!write_png(aov_normal, rgb.data(), desc.xresolution, desc.yresolution)) return 1;
  }
  if (!aov_albedo.empty()) {  // main.rs:1678-1687
    if (!layer(RENE_LAYER_ALBEDO, img)) return die("rene_download");
    rene_to_aov8(img.data(), img.size(), spp, 0, rgb.data());
    if (!write_png(aov_albedo, rgb.data(), desc.xresolution, desc.yresolution)) return 1;
  }
  for (rene_ctx* c : ctx) rene_destroy(c);
  rene_scene_free(scene);
  std::fprintf(stderr, "INFO %llu rays, %.1f Mrays/s (%.1f ms of rendering on %u GPU(s); launch durations add up to %.1f ms)\n",
               (unsigned long long)rays, render_ms > 0 ? rays / render_ms / 1e3 : 0.0, render_ms, gpus, kernel_ms);
  std::fprintf(stderr, "INFO End (%lld ms)\n", ms_since(t_start));
  return 0;
}
