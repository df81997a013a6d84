// rene_hip.cpp -- implementation of the C ABI declared in include/rene_hip.h (render path part).
// Compiled with hipcc; owns all device memory; never throws or aborts across the boundary.
#include "../../include/rene_hip.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is loaded on demand (rccl() below)

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "device_scene.h"
#include "kernels.h"
#include "scene_pack.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
  g_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(e_ == hipErrorOutOfMemory ? RENE_ERR_OUT_OF_MEMORY : RENE_ERR_DEVICE,            \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                              \
  } while (0)

// PCG32si restated for the host-side seed schedule (rene-shader/src/rand.rs:4-52)
struct HostPcg {
  uint32_t s;
  explicit HostPcg(uint32_t seed) : s(seed) {
    step();
    s += seed;
    step();
  }
  void step() { s = s * 747796405u + 2891336453u; }
  uint32_t next() {
    uint32_t o = s;
    step();
    uint32_t w = ((o >> ((o >> 28) + 4u)) ^ o) * 277803737u;
    return (w >> 22) ^ w;
  }
  // n steps at once (the LCG's n-fold composition by repeated squaring): frame k's seed is the k-th output, and a
  // caller may ask for frame four billion without anybody having produced the ones before it
  void skip(uint64_t n) {
    uint32_t mul = 747796405u, add = 2891336453u, acc_mul = 1u, acc_add = 0u;
    for (; n; n >>= 1) {
      if (n & 1u) {
        acc_mul *= mul;
        acc_add = acc_add * mul + add;
      }
      add = (mul + 1u) * add;
      mul *= mul;
    }
    s = acc_mul * s + acc_add;
  }
};

}  // namespace

namespace rene {
void set_last_error(const std::string& msg) { g_error = msg; }
thread_local uint32_t g_launched_blocks = 0;
}

// ---- RCCL, loaded on demand: a host that never shards keeps running where librccl.so is absent -----------------
namespace {
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};
Rccl* rccl() {
  static Rccl* r = [] {
    Rccl* x = new Rccl();
    // a process that already holds an RCCL (PyTorch ships its own copy) uses that one; else the ROCm installation's
    // (RENE_RCCL_LIB=<path> names the library instead: hosts that keep it elsewhere, and the test of the missing-library path)
    const char* over = std::getenv("RENE_RCCL_LIB");
    const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::vector<const char*> names;
    static std::vector<std::string> over_names;  // RENE_RCCL_LIB may hold several candidates, ':'-separated, tried in order like the defaults
    if (over && *over) {
      std::string all(over);
      for (size_t b = 0; b <= all.size();) {
        size_t e = all.find(':', b);
        if (e == std::string::npos) e = all.size();
        if (e > b) over_names.push_back(all.substr(b, e - b));
        b = e + 1;
      }
      for (const std::string& n : over_names) names.push_back(n.c_str());
    } else names.assign(defaults, defaults + 3);
    for (const char* n : names) {
      x->handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (x->handle) break;
    }
    std::string why;  // a COPY: dlerror()'s buffer belongs to the loader and the next failing dlopen() rewrites it
    for (const char* n : names) {
      if (x->handle) break;
      x->handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (!x->handle && why.empty()) {
        if (const char* m = dlerror()) why = m;
      }
    }
    if (!x->handle) {
      x->error = std::string("RCCL is not available: ") + (why.empty() ? "librccl.so not found" : why);
      return x;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
      void* p = dlsym(x->handle, name);
      if (!p) { ok = false; x->error = std::string("RCCL lacks ") + name; }
      return p;
    };
    x->GetUniqueId = reinterpret_cast<decltype(x->GetUniqueId)>(sym("ncclGetUniqueId"));
    x->CommInitRank = reinterpret_cast<decltype(x->CommInitRank)>(sym("ncclCommInitRank"));
    x->CommInitAll = reinterpret_cast<decltype(x->CommInitAll)>(sym("ncclCommInitAll"));
    x->CommDestroy = reinterpret_cast<decltype(x->CommDestroy)>(sym("ncclCommDestroy"));
    x->Reduce = reinterpret_cast<decltype(x->Reduce)>(sym("ncclReduce"));
    x->Send = reinterpret_cast<decltype(x->Send)>(sym("ncclSend"));
    x->Recv = reinterpret_cast<decltype(x->Recv)>(sym("ncclRecv"));
    x->GroupStart = reinterpret_cast<decltype(x->GroupStart)>(sym("ncclGroupStart"));
    x->GroupEnd = reinterpret_cast<decltype(x->GroupEnd)>(sym("ncclGroupEnd"));
    x->GetErrorString = reinterpret_cast<decltype(x->GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(x->handle); x->handle = nullptr; }
    return x;
  }();
  return r;
}
#define RCCL_TRY(expr)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return fail(RENE_ERR_DEVICE, std::string(#expr) + ": " + R->GetErrorString(r_)); \
  } while (0)
}  // namespace


// Host-side waits poll (hipEventQuery / hipStreamQuery read the completion signal) instead of blocking in the runtime: a
// blocked wait depends on a wake-up from the driver, and on this pool a job of many launches was seen to sit in
// one for minutes now and then (docs/history.md section 4g); polling costs one host thread a few microseconds of latency.
static hipError_t wait_event(hipEvent_t ev) {
  const auto t0 = std::chrono::steady_clock::now();
  int told = 0;
  for (uint64_t spins = 0;; ++spins) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if ((spins & 0xfffu) == 0xfffu && std::getenv("RENE_DEBUG")) {
      const double s_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (s_ > 10.0 * (told + 1)) {
        ++told;
        std::fprintf(stderr, "[rene] still waiting for a launch to complete after %.0f s (its completion signal has not fired)\n", s_);
      }
    }
    if (spins < 2000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}
static hipError_t wait_stream(hipStream_t st) {
  for (uint64_t spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    if (spins < 2000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

struct rene_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  rene_opts opts{};
  std::vector<void*> allocations;
  rene::SceneView view{};
  rene::LaunchConfig cfg{};
  uint32_t width = 0, height = 0, tiles_x = 0, n_tiles = 0, n_work = 0, n_materials = 0, n_mediums = 0;
  float* fb = nullptr;
  bool own_fb = false;
  size_t fb_floats = 0;
  // frame chains (device_scene.h, CHAINS): the kernels accumulate into `chains`, [CHAINS][3][H][W][4], the library's own; `fb` -- the context's or
  // the caller's -- is the image handed out: the chains added in chain order whenever a drain finds launches since the last one (`fb_stale`)
  float* chains = nullptr;
  bool fb_stale = false;
  float* ray_dump = nullptr;     // rene_ray_dump in progress: where the counting restart kernel records its queries
  uint32_t ray_dump_cap = 0;
  static constexpr uint32_t kCounters = 60;  // launches between two drains: each takes its own zeroed work counter
  uint32_t* d_work_counters = nullptr;        // [kCounters]
  unsigned long long* d_wave_times = nullptr; // RENE_DEBUG: [kCounters][8192][2]
  uint32_t counters_used = 0;
  uint32_t epoch = 0, prev_final = 0;
  // frames per work item: kWholeLaunch = one item per pixel and launch; 0 = not tuned (rene_tune picks): 64 for the
  // item-loop kernels, whose item switches cost a memory round trip of the whole wave, 32 for the BVH kernels, where
  // pixels differ more in cost and a lane that waits is a lane the ballots miss
  static constexpr uint32_t kWholeLaunch = 0xffffffffu;
  uint32_t item_frames = 0;
  std::vector<uint32_t> inst_material;  // material index of every instance (rene_bsdf_eval looks an instance of its material up)
  uint32_t* d_item_done = nullptr;  // [CHAINS][H][W] versions, traversal-restart kernels only (device_code.inc, item_flag_publish)
  unsigned long long* d_counters = nullptr;
  // stage-separated wavefront integrator (BVH scenes): path state in HBM + a pinned word for the host loop
  bool wavefront = false;
  rene::WaveState wave{};
  uint32_t* h_done = nullptr;
  // per-launch resources that must outlive the asynchronous launch
  struct Pending {
    hipEvent_t start, stop;
    uint32_t epoch;
    bool replayable = false;    // a persistent render launch: what it was launched with, should it have to be launched again
    rene::RenderParams P{};
    rene::LaunchConfig cfg{};
  };
  uint64_t replays = 0;         // launches launched again by drain() (RENE_DEBUG prints them)
  std::deque<Pending> pending;
  uint64_t frames = 0, launches = 0, owned_pixels = 0, paths = 0;
  double kernel_ms = 0.0, last_ms = 0.0;
  bool handoff_failed = false;
  std::string handoff_detail;
  // multi-GPU exchange (rene_comm_*): the RCCL communicator this context belongs to
  ncclComm_t comm = nullptr;
  int comm_ranks = 0, comm_rank = -1;
  bool exchanged = false;  // an exchange has put other ranks' sums into `fb`, which the next drain would overwrite with this context's chains: rene_reset before rendering again
  float* tile_buf = nullptr;  // rene_gather_tiles: packed owned tiles (root: of every rank)
  float* h_stage = nullptr;   // pinned host staging of one layer (rene_download)
  void* h_upload = nullptr;   // pinned host staging of the scene upload (rene_create), released when it is done
  int unpack_root = -1;       // >= 0: tiles received by rene_gather_tiles wait in tile_buf to be placed (flush_exchange)
  size_t tile_buf_floats = 0;
  // seed schedule cache: seeds[k] = k-th next_u32 of PCG32si::new(master)

  template <class T>
  int upload(const std::vector<T>& v, const T** out) {
    size_t bytes = std::max<size_t>(1, v.size()) * sizeof(T);
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    allocations.push_back(p);
    if (!v.empty()) {
      // through pinned staging, in pieces: a copy straight from pageable memory makes the runtime register the caller's
      // pages with the driver for the duration of the copy (see rene_download)
      constexpr size_t kPiece = 8u << 20;
      if (!h_upload) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_upload), kPiece, hipHostMallocDefault));
      const char* src = reinterpret_cast<const char*>(v.data());
      const size_t total = v.size() * sizeof(T);
      for (size_t off = 0; off < total; off += kPiece) {
        const size_t n = std::min(kPiece, total - off);
        std::memcpy(h_upload, src + off, n);
        HIP_TRY(hipMemcpy(static_cast<char*>(p) + off, h_upload, n, hipMemcpyHostToDevice));
      }
    } else {
      HIP_TRY(hipMemset(p, 0, bytes));
    }
    *out = static_cast<const T*>(p);
    return RENE_OK;
  }

  // root of a rene_gather_tiles: place the other ranks' tiles (enqueued behind the receives on `stream`)
  int flush_exchange() {
    if (unpack_root < 0) return RENE_OK;
    const uint32_t n = (uint32_t)comm_ranks;
    const size_t tile_floats = (size_t)3 * RENE_TILE_SIZE * RENE_TILE_SIZE * 4;
    size_t off = 0;
    for (uint32_t r = 0; r < n; ++r) {
      const uint32_t owned_r = n_tiles > r ? (n_tiles - r + n - 1) / n : 0u;
      if ((int)r == unpack_root || owned_r == 0) continue;
      hipError_t e = rene::launch_pack_tiles(fb, tile_buf + off, width, height, tiles_x, n_tiles, r, n, true, stream);
      if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_gather_tiles unpack: ") + hipGetErrorString(e));
      off += owned_r * tile_floats;
    }
    unpack_root = -1;
    return RENE_OK;
  }

  // Zeroes device memory and WAITS for it.  (hipMemset on the null stream returns before the fill has run -- it is a kernel, and
  // behind a persistent launch on another queue it gets a slot only when that launch's first waves leave: the work counters
  // of launches already running were then zeroed under them, their ids handed out a second time, and the duplicates waited
  // for versions that had passed.  docs/history.md section 4g.)
  hipError_t zero_now(void* p, size_t bytes) {
    hipError_t e = hipMemsetAsync(p, 0, bytes, stream);
    return e == hipSuccess ? wait_stream(stream) : e;
  }

  int drain() {  // wait for the stream(s) and fold finished launches into the timing totals
    {
      int rc_ = flush_exchange();
      if (rc_ != RENE_OK) return rc_;
    }
    HIP_TRY(wait_stream(stream));
    const bool had_launches = !pending.empty();
    if (counters_used && d_wave_times) {  // RENE_DEBUG: per launch, when its waves started and ended (ms since the first start)
      std::vector<unsigned long long> t((size_t)counters_used * 8192 * 2);
      if (hipMemcpy(t.data(), d_wave_times, t.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
        unsigned long long t0 = ~0ull;
        for (auto v : t) if (v && v < t0) t0 = v;
        for (uint32_t k = 0; k < counters_used; ++k) {
          std::vector<double> st, en;
          for (uint32_t w = 0; w < 8192; ++w) {
            const unsigned long long a = t[((size_t)k * 8192 + w) * 2], b = t[((size_t)k * 8192 + w) * 2 + 1];
            if (a) st.push_back((a - t0) * 1e-5);
            if (b) en.push_back((b - t0) * 1e-5);
          }
          std::sort(st.begin(), st.end());
          std::sort(en.begin(), en.end());
          auto q = [](const std::vector<double>& v, double f) { return v.empty() ? -1.0 : v[(size_t)(f * (v.size() - 1))]; };
          std::fprintf(stderr, "[rene] launch slot %u: %zu waves; starts min/50%%/90%%/max %.3f %.3f %.3f %.3f ms; ends min/10%%/50%%/90%%/max %.3f %.3f %.3f %.3f %.3f ms\n", k, st.size(),
                       q(st, 0), q(st, .5), q(st, .9), q(st, 1), q(en, 0), q(en, .1), q(en, .5), q(en, .9), q(en, 1));
        }
      }
      zero_now(d_wave_times, (size_t)counters_used * 8192 * 2 * 8);
    }
    if (counters_used) {  // the stream is idle: the work counters can be handed out again
      HIP_TRY(zero_now(d_work_counters, kCounters * sizeof(uint32_t)));
      counters_used = 0;
    }
    // Items whose hand-off did not come were DROPPED by their lanes (device_code.inc: the waves that render the awaited
    // item can be parked by the driver behind this launch's own -- a queue eviction restores the queues in its own order --
    // and from then on every waiter of this and of the following launches drops too).  Nothing wrong has been added to the
    // image: the launches since the last sync are launched again, one at a time on an idle device, in their order and with
    // their own parameters; an item that was committed the first time finds its pixel's version ahead of it and is skipped.
    if (had_launches && d_counters && !handoff_failed) {
      for (int attempt = 0; attempt < 3; ++attempt) {
        unsigned long long dropped = 0;
        HIP_TRY(hipMemcpy(&dropped, d_counters + 8, sizeof(dropped), hipMemcpyDeviceToHost));
        if (!dropped) break;
        bool all = true;
        for (const Pending& p : pending) all = all && p.replayable;
        if (!all) break;  // (the wavefront integrator's launches are not of this kind)
        if (std::getenv("RENE_DEBUG")) {
          unsigned long long t[4] = {0, 0, 0, 0};
          hipMemcpy(t, d_counters + 8, sizeof(t), hipMemcpyDeviceToHost);
          std::fprintf(stderr, "[rene] %llu work items were dropped (the first: pixel (%llu, %llu), in launch %llu, wanted version %llu, saw %llu): launching the last %zu launch(es), %u..%u, again, serially (attempt %d)\n",
                       dropped, t[1] & 0x3fffull, (t[1] >> 14) & 0x3fffull, t[2] >> 32, t[3], t[2] & 0xffffffffull, pending.size(), pending.front().epoch, pending.back().epoch, attempt + 1);
        }
        HIP_TRY(zero_now(d_counters + 8, 4 * sizeof(unsigned long long)));
        for (Pending& p : pending) {
          HIP_TRY(zero_now(d_work_counters, kCounters * sizeof(uint32_t)));
          rene::RenderParams P = p.P;
          P.flags &= ~rene::RENE_FLAG_INTERNAL_TEST_DROP;
          rene::g_launched_blocks = p.cfg.grid;
          const auto tr = std::chrono::steady_clock::now();
          hipError_t e = rene::launch_render(p.cfg, view, P, stream);
          if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("render launch (replay): ") + hipGetErrorString(e));
          HIP_TRY(wait_stream(stream));
          kernel_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr).count();  // rene_stats.launches counts it too
          ++replays;
        }
        HIP_TRY(zero_now(d_work_counters, kCounters * sizeof(uint32_t)));
      }
    }
    while (!pending.empty()) {
      Pending& p = pending.front();
      float ms = 0.0f;
      HIP_TRY(hipEventElapsedTime(&ms, p.start, p.stop));
      if (std::getenv("RENE_DEBUG")) std::fprintf(stderr, "[rene] launch %u: %.3f ms\n", p.epoch, ms);
      kernel_ms += ms;
      last_ms = ms;
      hipEventDestroy(p.start);
      hipEventDestroy(p.stop);
      pending.pop_front();
    }
    // items still dropped after three replays on an idle device: something else is wrong, and every call that hands
    // results to the caller (rene_sync, rene_download, rene_get_stats, rene_reduce) must say so
    if (had_launches && d_counters && !handoff_failed) {
      unsigned long long t[4] = {0, 0, 0, 0};
      HIP_TRY(hipMemcpy(t, d_counters + 8, sizeof(t), hipMemcpyDeviceToHost));
      handoff_failed = t[0] != 0;
      if (handoff_failed)
        handoff_detail = " [" + std::to_string(t[0]) + " lanes gave up; the first: pixel (" + std::to_string(t[1] & 0x3fffull) + ", " + std::to_string((t[1] >> 14) & 0x3fffull) + "), in launch " +
                         std::to_string(t[2] >> 32) + ", wanted version " + std::to_string(t[3]) + ", saw " + std::to_string(t[2] & 0xffffffffu) + "]";
    }
    if (handoff_failed) return fail(RENE_ERR_DEVICE, "work items were dropped inside the render kernel and replaying their launches did not complete them (results invalid; rene_reset clears the condition)" + handoff_detail);
    if (fb_stale) {  // frame chains: the image handed out = the chains added in chain order (the chains go on accumulating)
      const bool tiles = opts.shard_mode == RENE_SHARD_TILES && opts.shard_count > 1;  // (a tile shard: only the tiles it owns hold anything)
      hipError_t e = tiles ? rene::launch_chains_tiles(chains, fb, false, width, height, tiles_x, n_tiles, opts.shard_rank, opts.shard_count, stream)
                           : rene::launch_resolve_chains(chains, fb, fb_floats, stream);
      if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("resolve_chains: ") + hipGetErrorString(e));
      HIP_TRY(wait_stream(stream));
      fb_stale = false;
    }
    return RENE_OK;
  }
};

// No exception crosses the C boundary: the entry points that allocate host memory run under this guard.
template <class F>
static int guarded(F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(RENE_ERR_OUT_OF_MEMORY, "host allocation failed");
  } catch (const std::exception& e) {
    return fail(RENE_ERR_DEVICE, std::string("unexpected exception: ") + e.what());
  }
}

extern "C" {

const char* rene_last_error(void) { return g_error.c_str(); }
uint32_t rene_abi_version(void) { return RENE_ABI_VERSION; }

void rene_frame_seeds(uint32_t master_seed, uint32_t first_frame, uint32_t n, uint32_t* out) {
  HostPcg g(master_seed);
  g.skip(first_frame);
  for (uint32_t k = 0; k < n; ++k) out[k] = g.next();
}

static int rene_scene_pack_info_impl(const rene_scene_desc* scene, rene_pack_info* out) {
  if (!scene || !out) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_scene_pack_info: NULL argument");
  rene::PackedScene ps;
  std::string err;
  int rc = rene::pack_scene(scene, ps, err);
  if (rc != RENE_OK) return fail(rc, err);
  std::memset(out, 0, sizeof(*out));
  out->n_instances = (uint32_t)ps.insts.size();
  out->n_triangles = ps.n_triangles;
  out->n_spheres = (uint32_t)ps.spheres.size();
  out->n_nodes_main = (uint32_t)ps.main.nodes.size();
  out->n_slots_main = (uint32_t)ps.main.isect.size();
  out->depth_main = ps.main.depth;
  out->n_nodes_emit = (uint32_t)ps.emit.nodes.size();
  out->n_slots_emit = (uint32_t)ps.emit.isect.size();
  out->depth_emit = ps.emit.depth;
  out->features = ps.features;
  out->emit_object_len = (uint32_t)ps.emit_objects.size();
  out->lights_len = (uint32_t)ps.lights.size();
  out->n_items_main = (ps.features & rene::FEAT_SMALL) ? ps.main.n_loop : 0u;
  out->n_items_emit = (ps.features & rene::FEAT_SMALL) ? ps.emit.n_loop : 0u;
  out->device_bytes = ps.main.nodes.size() * sizeof(rene::Node) + ps.main.isect.size() * sizeof(rene::PrimIsect) +
                      ps.emit.nodes.size() * sizeof(rene::Node) + ps.emit.isect.size() * sizeof(rene::PrimIsect) +
                      ps.shade.size() * sizeof(rene::PrimShade) + ps.emit_pdf.size() * sizeof(rene::EmitPdf) +
                      ps.spheres.size() * sizeof(rene::Sphere) + ps.insts.size() * sizeof(rene::Inst) +
                      ps.emit_objects.size() * sizeof(rene::EmitObject) + ps.emit_tris.size() * sizeof(rene::EmitTri) +
                      ps.materials.size() * sizeof(rene::Material) + ps.textures.size() * sizeof(rene::Texture) +
                      ps.lights.size() * sizeof(rene::Light) + ps.image_pool.size() * sizeof(float) +
                      ps.mediums.size() * sizeof(rene::Medium) + ps.inst_medium.size() * sizeof(rene::InstMedium);
  return RENE_OK;
}

static int rene_create_impl(const rene_scene_desc* scene, const rene_opts* opts, rene_ctx** out) {
  if (!scene || !out) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_create: NULL argument");
  *out = nullptr;
  rene_opts o{};
  o.struct_size = sizeof(rene_opts);
  o.seed = RENE_DEFAULT_SEED;
  if (opts) {
    if (opts->struct_size != sizeof(rene_opts))
      return fail(RENE_ERR_INVALID_ARGUMENT, "rene_opts.struct_size mismatch (ABI skew)");
    o = *opts;
  }
  if (o.shard_count == 0) o.shard_count = 1;
  if (o.shard_rank >= o.shard_count) return fail(RENE_ERR_INVALID_ARGUMENT, "shard_rank >= shard_count");
  if (o.shard_mode > RENE_SHARD_FRAMES) return fail(RENE_ERR_INVALID_ARGUMENT, "unknown shard_mode");

  rene::PackedScene ps;
  std::string err;
  int rc = rene::pack_scene(scene, ps, err);
  if (rc != RENE_OK) return fail(rc, err);

  int n_dev = 0;
  HIP_TRY(hipGetDeviceCount(&n_dev));
  if (n_dev <= 0) return fail(RENE_ERR_DEVICE, "no HIP device visible (the render path has no CPU fallback)");
  if (o.device < 0 || o.device >= n_dev) return fail(RENE_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  HIP_TRY(hipSetDevice(o.device));
  if (ps.width > rene::MAX_RESOLUTION || ps.height > rene::MAX_RESOLUTION)  // a lane keeps its pixel and chain as x | y << 14 | chain << 28
    return fail(RENE_ERR_INVALID_ARGUMENT, "resolutions above 16384 are not supported");
  for (int i = 0; i < 16; ++i)  // the kernels read the camera's origin off the matrix instead of multiplying a zero point through it
    if (!std::isfinite(ps.uniform.camera_to_world[i]) || !std::isfinite(ps.uniform.projection_inv[i]))
      return fail(RENE_ERR_INVALID_ARGUMENT, "rene_uniform: camera_to_world / projection_inv must be finite");

  std::unique_ptr<rene_ctx> c(new rene_ctx());
  c->device = o.device;
  c->opts = o;
  c->width = ps.width;
  c->height = ps.height;
  c->n_materials = (uint32_t)ps.materials.size();
  c->n_mediums = (uint32_t)ps.mediums.size();  // 0 unless the integrator is volpath
  struct Cleanup {
    std::unique_ptr<rene_ctx>& c;
    bool armed = true;
    ~Cleanup() {
      if (armed && c) {
        rene_ctx* p = c.release();
        rene_destroy(p);
      }
    }
  } cleanup{c};

  if (o.stream) {
    c->stream = static_cast<hipStream_t>(o.stream);
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }

  rene::SceneView& v = c->view;
#define UP(vec, field)                              \
  do {                                              \
    int rc_ = c->upload(vec, &field);               \
    if (rc_ != RENE_OK) return rc_;                 \
  } while (0)
  UP(ps.main.nodes, v.main.nodes);
  UP(ps.main.isect, v.main.isect);
  UP(ps.emit.nodes, v.emit.nodes);
  UP(ps.emit.isect, v.emit.isect);
  UP(ps.main.items, v.main.items);
  UP(ps.emit.items, v.emit.items);
  v.main.n_top = ps.main.n_top;
  v.emit.n_top = ps.emit.n_top;
  v.main.n_items = ps.main.n_loop;  // the loop's items; the auxiliary records of box items follow them
  v.emit.n_items = ps.emit.n_loop;
  v.main.n_nodes = (uint32_t)ps.main.nodes.size();
  v.main.n_slots = (uint32_t)ps.main.isect.size();
  v.emit.n_nodes = (uint32_t)ps.emit.nodes.size();
  v.emit.n_slots = (uint32_t)ps.emit.isect.size();
  UP(ps.shade, v.shade);
  UP(ps.emit_pdf, v.emit_pdf);
  UP(ps.spheres, v.spheres);
  UP(ps.insts, v.insts);
  c->inst_material.clear();
  for (const rene::Inst& in : ps.insts) c->inst_material.push_back(in.material);
  UP(ps.emit_objects, v.emit_objects);
  UP(ps.emit_tris, v.emit_tris);
  UP(ps.materials, v.materials);
  UP(ps.textures, v.textures);
  UP(ps.lights, v.lights);
  UP(ps.mediums, v.mediums);
  UP(ps.inst_medium, v.inst_medium);
  UP(ps.images, v.images);
  UP(ps.image_pool, v.image_pool);
  UP(ps.small_image, v.small_image);
#undef UP
  v.small_bytes = (uint32_t)(ps.small_image.size() * sizeof(float));
  for (uint32_t k = 0; k < rene::SMALL_OFF_COUNT; ++k) v.small_off[k] = ps.small_off[k];
  {
    std::vector<rene::Uniforms> u(1);
    std::memset(&u[0], 0, sizeof(u[0]));
    std::memcpy(u[0].c2w, ps.uniform.camera_to_world, 64);
    std::memcpy(u[0].proj_inv, ps.uniform.projection_inv, 64);
    std::memcpy(u[0].bg_matrix, ps.uniform.background_matrix, 64);
    std::memcpy(u[0].bg_color, ps.uniform.background_color, 16);
    u[0].bg_kind = 0;
    if (ps.uniform.background_texture < ps.textures.size() && !std::getenv("RENE_NO_RESOLVE")) {
      const rene::Texture& t = ps.textures[ps.uniform.background_texture];
      if (t.type == RENE_TEXTURE_SOLID) {
        u[0].bg_kind = 1;
        std::memcpy(u[0].bg_solid, t.v0, 12);
      } else if (t.type == RENE_TEXTURE_IMAGEMAP && t.u0[0] < ps.images.size()) {
        u[0].bg_kind = 2;
        u[0].bg_image_offset = ps.images[t.u0[0]].offset;
        u[0].bg_image_width = ps.images[t.u0[0]].width;
        u[0].bg_image_height = ps.images[t.u0[0]].height;
      }
    }
    int rc_ = c->upload(u, &v.uni);
    if (rc_ != RENE_OK) return rc_;
  }
  v.bg_texture = ps.uniform.background_texture;
  v.lights_len = (uint32_t)ps.lights.size();                // rene/src/scene.rs:166
  v.emit_object_len = (uint32_t)ps.emit_objects.size();     // rene/src/main.rs:3279
  v.width = ps.width;
  v.height = ps.height;

  // traversal stack: enough for the deeper of the two trees, in LDS, [depth][256 lanes]
  uint32_t depth = std::max(ps.main.depth, ps.emit.depth);
  uint32_t stack = 16;
  while (stack < depth) stack += 4;  // 40 entries x 1024 lanes x 4 B = the whole 160 KB of a CU at 4 waves per SIMD
  if (stack > 96) return fail(RENE_ERR_UNSUPPORTED, "BVH deeper than the 96-entry traversal stack");
  // (occupancy experiments: RENE_STACK_ENTRIES may only GROW the stack -- a stack shallower than the tree's worst case would let a deep query write
  // past its column into other lanes' entries and the LDS tables: refused)
  if (const char* e = std::getenv("RENE_STACK_ENTRIES")) {
    const int want = std::atoi(e);
    if (want < (int)depth) return fail(RENE_ERR_INVALID_ARGUMENT, "RENE_STACK_ENTRIES is below the traversal stack this scene's BVH can need (" + std::to_string(depth) + " entries)");
    stack = (uint32_t)std::min(96, want);
  }
  c->cfg.features = ps.features;
  if (o.flags & RENE_FLAG_FORCE_BVH) c->cfg.features &= ~rene::FEAT_SMALL;
  c->cfg.stack_depth = stack;
  c->cfg.n_insts = (uint32_t)ps.insts.size();
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, o.device));
  c->cfg.grid = (uint32_t)prop.multiProcessorCount * 8u;  // persistent launch: upper bound, clamped to the co-resident blocks at launch
  c->cfg.cus = (uint32_t)prop.multiProcessorCount;
  if (const char* e = std::getenv("RENE_BLOCKS_PER_CU")) {  // tuning knob (experiments only)
    int b = std::atoi(e);
    if (b > 0 && b <= 64) c->cfg.grid = (uint32_t)prop.multiProcessorCount * (uint32_t)b;
  }

  c->tiles_x = (ps.width + RENE_TILE_SIZE - 1) / RENE_TILE_SIZE;
  uint32_t tiles_y = (ps.height + RENE_TILE_SIZE - 1) / RENE_TILE_SIZE;
  c->n_tiles = c->tiles_x * tiles_y;
  uint32_t tile_rank = 0, tile_count = 1;
  if (o.shard_mode == RENE_SHARD_TILES) {
    tile_rank = o.shard_rank;
    tile_count = o.shard_count;
  }
  uint32_t owned = c->n_tiles > tile_rank ? (c->n_tiles - tile_rank + tile_count - 1) / tile_count : 0;
  // work ids are 32-bit: id = level * n_work + (pixel slot * CHAINS + chain) with room for 32 levels, and udiv_small needs ids below 2^31
  if ((uint64_t)owned * RENE_TILE_SIZE * RENE_TILE_SIZE * rene::CHAINS * 32ull >= (1ull << 31))
    return fail(RENE_ERR_UNSUPPORTED, "image too large: more than 2^23 pixels per GPU (shard it by tiles)");
  c->n_work = owned * RENE_TILE_SIZE * RENE_TILE_SIZE;
  // pixels of the image inside the owned tiles: paths per rendered frame (the kernels do not count what the host knows)
  c->owned_pixels = 0;
  for (uint32_t t = tile_rank; t < c->n_tiles; t += tile_count) {
    const uint32_t x0 = (t % c->tiles_x) * RENE_TILE_SIZE, y0 = (t / c->tiles_x) * RENE_TILE_SIZE;
    c->owned_pixels += (uint64_t)std::min<uint32_t>(RENE_TILE_SIZE, ps.width - x0) * std::min<uint32_t>(RENE_TILE_SIZE, ps.height - y0);
  }

  c->fb_floats = (size_t)3 * ps.width * ps.height * 4;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->chains), (size_t)rene::CHAINS * c->fb_floats * sizeof(float)));  // frame chains (device_scene.h)
  if (o.framebuffer) {
    c->fb = static_cast<float*>(o.framebuffer);
  } else {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->fb), c->fb_floats * sizeof(float)));
    c->own_fb = true;
  }
  // Which integrator renders this scene: the item-loop megakernel (small scenes), the volpath megakernel,
  // or -- for everything that needs the BVH -- the traversal-restart megakernel; RENE_FLAG_WAVEFRONT selects
  // the stage-separated wavefront (wavefront.inc) instead, unless the scene has more distant lights than
  // it keeps shadow-ray slots for.
  constexpr uint32_t kWaveMaxLights = 4;
  c->cfg.wave_stack = std::max(1u, depth);
  c->wavefront = !(c->cfg.features & (rene::FEAT_SMALL | rene::FEAT_VOLPATH)) &&
                 (o.flags & RENE_FLAG_WAVEFRONT) && !(o.flags & RENE_FLAG_NO_RESTART) && ps.lights.size() <= kWaveMaxLights &&
                 c->n_work > 0;
  if (c->wavefront) {
    rene::WaveState& q = c->wave;
    q.n_slots = c->n_work;
    q.max_lights = (uint32_t)ps.lights.size();
    q.fp16_payload = (o.flags & RENE_FLAG_FP16_PAYLOAD) ? 1u : 0u;
    const size_t n = q.n_slots;
    auto dev = [&](void** p, size_t bytes) {
      hipError_t e = hipMalloc(p, std::max<size_t>(16, bytes));
      if (e == hipSuccess) c->allocations.push_back(*p);
      return e;
    };
    HIP_TRY(dev(reinterpret_cast<void**>(&q.ro), n * 16));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.rd), n * 16));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.color), n * 16));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.ctl), n * 16));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.hit), n * 16));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.sh_wi), n * 16 * q.max_lights));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.sh_c), n * 16 * q.max_lights));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.status), n * 4));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.n_done), 4));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.trace_counter), 8));
    HIP_TRY(dev(reinterpret_cast<void**>(&q.wave_sums), ((n + 255) / 256) * 4 * 8 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_done), sizeof(uint32_t)));
  }
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_work_counters), rene_ctx::kCounters * sizeof(uint32_t)));
  HIP_TRY(hipMemsetAsync(c->d_work_counters, 0, rene_ctx::kCounters * sizeof(uint32_t), c->stream));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_counters), 32 * sizeof(unsigned long long)));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_item_done), (size_t)rene::CHAINS * ps.width * ps.height * sizeof(uint32_t)));  // one version word per pixel and chain
  HIP_TRY(hipMemsetAsync(c->d_item_done, 0, (size_t)rene::CHAINS * ps.width * ps.height * sizeof(uint32_t), c->stream));
  HIP_TRY(hipMemsetAsync(c->chains, 0, (size_t)rene::CHAINS * c->fb_floats * sizeof(float), c->stream));  // main.rs:1229-1237
  HIP_TRY(hipMemsetAsync(c->fb, 0, c->fb_floats * sizeof(float), c->stream));
  HIP_TRY(hipMemsetAsync(c->d_counters, 0, 32 * sizeof(unsigned long long), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipDeviceSynchronize());  // everything the uploads left on the null stream (the fills of empty tables) has run: see zero_now

  if (c->h_upload) {
    hipHostFree(c->h_upload);
    c->h_upload = nullptr;
  }
  // tests: start the launch numbering just below the wrap of the 22-bit epoch (MAX_EPOCH), so that a handful of launches cross it
  if (const char* e = std::getenv("RENE_TEST_EPOCH")) c->epoch = (uint32_t)std::min<unsigned long>(rene::MAX_EPOCH, std::strtoul(e, nullptr, 0));
  cleanup.armed = false;
  *out = c.release();
  return RENE_OK;
}

void rene_destroy(rene_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  for (auto& p : c->pending) {
    hipEventDestroy(p.start);
    hipEventDestroy(p.stop);
  }
  for (void* p : c->allocations) hipFree(p);
  if (c->own_fb && c->fb) hipFree(c->fb);
  if (c->chains) hipFree(c->chains);
  if (c->d_work_counters) hipFree(c->d_work_counters);
  if (c->d_wave_times) hipFree(c->d_wave_times);
  if (c->d_counters) hipFree(c->d_counters);
  if (c->d_item_done) hipFree(c->d_item_done);
  if (c->h_done) hipHostFree(c->h_done);
  if (c->h_stage) hipHostFree(c->h_stage);
  if (c->h_upload) hipHostFree(c->h_upload);
  if (c->tile_buf) hipFree(c->tile_buf);
  if (c->comm && rccl()->handle) rccl()->CommDestroy(c->comm);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete c;
}

static int rene_render_impl(rene_ctx* c, uint32_t first_frame, uint32_t n_frames) {
  if (!c) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_render: NULL context");
  if (n_frames == 0) return RENE_OK;
  if (c->exchanged) return fail(RENE_ERR_INVALID_ARGUMENT, "the image has been through rene_reduce / rene_gather_tiles: rene_reset before rendering again");
  if ((uint64_t)first_frame + n_frames > 0xffffffffull) return fail(RENE_ERR_INVALID_ARGUMENT, "frame range overflows u32");
  if (n_frames > rene::MAX_LAUNCH_FRAMES) {  // one launch renders at most this many frames (its seed tables)
    for (uint32_t done = 0; done < n_frames;) {
      const uint32_t n = std::min(rene::MAX_LAUNCH_FRAMES, n_frames - done);
      int rc = rene_render_impl(c, first_frame + done, n);
      if (rc != RENE_OK) return rc;
      done += n;
    }
    return RENE_OK;
  }
  HIP_TRY(hipSetDevice(c->device));
  // which frames of [first_frame, first_frame + n_frames) are this context's: all of them, or under RENE_SHARD_FRAMES those
  // with f % shard_count == shard_rank.  The kernels compute the frames' seeds themselves (device_math.h, frame_seed).
  uint32_t my_first = first_frame, my_stride = 1, my_count = n_frames;
  if (c->opts.shard_mode == RENE_SHARD_FRAMES && c->opts.shard_count > 1) {
    const uint32_t n = c->opts.shard_count, r = c->opts.shard_rank;
    const uint32_t skip = (r + n - first_frame % n) % n;  // frames before the first one with f % n == r
    my_first = first_frame + skip;
    my_stride = n;
    my_count = skip < n_frames ? (n_frames - skip + n - 1) / n : 0;
  }
  c->frames += n_frames;
  if (my_count == 0 || c->n_work == 0) return RENE_OK;
  c->paths += (uint64_t)my_count * c->owned_pixels;

  if (c->epoch >= rene::MAX_EPOCH) {  // the hand-off flags are cleared when the epoch wraps: nothing may be in flight then
    int rc = c->drain();
    if (rc != RENE_OK) return rc;
  }
  // A launch is a kernel launch between two event records -- no allocation, no copy, no memset: those need a copy engine
  // or a free CU slot, which a persistent launch that holds the chip gives up only when it ends.  Hence the pool of work
  // counters (zeroed again in drain()) instead of one that is reset per launch.
  if (c->counters_used >= rene_ctx::kCounters) {
    int rc = c->drain();
    if (rc != RENE_OK) return rc;
  }
  // Launches are serial on the context's one stream.  (Rounds 1-2 alternated consecutive launches between two streams so that
  // the next launch filled the slots the previous one's tail vacated; the waves of the later launch then waited, holding
  // their slots, for pixels of the earlier one -- and when the driver evicted and restored the process's queues, the earlier
  // launch's waves could find their slots taken: a stall of seconds, docs/history.md section 4g.  One launch per job in short work
  // items has the same tail to hide -- none between launches -- and no launch ever waits for another.)
  hipStream_t stream = c->stream;
  uint32_t* work_counter = c->d_work_counters + c->counters_used;
  if (std::getenv("RENE_DEBUG") && !c->d_wave_times) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_wave_times), (size_t)rene_ctx::kCounters * 8192 * 2 * 8));
    HIP_TRY(c->zero_now(c->d_wave_times, (size_t)rene_ctx::kCounters * 8192 * 2 * 8));
  }
  rene_ctx::Pending pend{};
  hipError_t e = hipEventCreate(&pend.start);
  if (e == hipSuccess) e = hipEventCreate(&pend.stop);
  if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_render setup: ") + hipGetErrorString(e));
  rene::RenderParams P{};
  P.framebuffer = c->chains;
  P.seed_state0 = HostPcg(c->opts.seed).s;
  P.first_frame = my_first;
  P.frame_stride = my_stride;
  P.work_counter = work_counter;
  P.wave_times = c->d_wave_times ? c->d_wave_times + (size_t)c->counters_used * 8192 * 2 : nullptr;
  P.item_done = c->d_item_done;
  P.ray_dump = c->ray_dump;
  P.ray_dump_cap = c->ray_dump_cap;
  P.counters = c->d_counters;
  P.n_frames = my_count;
  // frame chains (device_scene.h): global frame f belongs to chain (f / frame_stride) % CHAINS -- a rule on the frame's number, so that a pixel's
  // chains hold the same sums however a job is cut into calls; a level's work ids: pixel slot * CHAINS + chain
  P.n_work = c->n_work * rene::CHAINS;
  P.chain_phase = (my_first / my_stride) & (rene::CHAINS - 1u);
  P.group_frames = (my_count + rene::CHAINS - 1u) / rene::CHAINS;
  P.shard_rank = c->opts.shard_mode == RENE_SHARD_TILES ? c->opts.shard_rank : 0;
  P.shard_count = c->opts.shard_mode == RENE_SHARD_TILES ? c->opts.shard_count : 1;
  P.tiles_x = c->tiles_x;
  P.inv_n_work = 1.0f / (float)std::max(1u, P.n_work);
  P.inv_tiles_x = 1.0f / (float)std::max(1u, c->tiles_x);
  P.n_tiles = c->n_tiles;
  P.flags = c->opts.flags;
  // No work item may belong to a wave that is not resident: items wait for their predecessors, and a statically owned
  // first batch of a wave that has not been given a slot -- the device shared with another process, or the previous
  // launch still draining -- can be waited for by every wave that has one (seen: two processes' counting launches on
  // one GPU, each holding the slots the other's missing waves needed).  With every batch taken from the atomic
  // counter the item a lane waits for is always in the hands of a resident lane.  (The static first batch saved
  // 0.05 ms per launch.)
  P.flags |= RENE_FLAG_DYNAMIC_FIRST;
  if (const char* e = std::getenv("RENE_TEST_DROP"))  // fault injection (tests): the context's launch number e drops some of its items
    if ((uint32_t)std::atoi(e) == c->epoch + 1u) P.flags |= rene::RENE_FLAG_INTERNAL_TEST_DROP;
  // every pixel's frames in work items (device_code.inc, item_frames): uniform items of `item` frames, the last one or two of
  // them cut into halving items down to `tail` frames; RENE_LEVELS=<n> (tests, A/B measurements) cuts into n uniform items
  {
    const uint32_t F = P.group_frames;  // (what the items cut: the frames of one chain)
    // untuned: sixteen items per pixel and launch for the item-loop kernels, 32 for the BVH kernels, at least 16 frames each
    // (measured, one launch per job, MI355X: Cornell 1024 frames flat from 64 to 96 frames per item, veach-mis 4096 frames best at
    // 256 - 341, dragon-class 1024 at 32, the teapot scene 8192 at 256: it is the number of item switches per pixel that a launch
    // pays for, and the length of its last item -- and a BVH scene's pixels differ more in cost);
    // no halving tail by default (tail = item): it buys nothing once the hand-off waits are rare (docs/history.md section 4f)
    // (short launches -- one rank's share of a multi-GPU job -- want few, long items: Cornell 128 frames, 8 / 16 / 32 / 64 frames per
    // item: 6.89 / 6.59 / 6.59 / 6.41 ms; 256 frames, 16 / 32 / 64 / 128: 13.30 / 13.13 / 13.21 / 12.84; 512 frames, 32 / 64 / 128: 24.93 / 24.74 / 25.23)
    // (frame groups: a chain has half the frames and wants items as long as the undivided job's, or longer -- dragon-class, two chains of 512
    // frames: items of 16 / 32 / 64 frames 671 / 653 / 642 ms; the teapot scene, two chains of 4096: 128 / 256 / 512 / 1024 frames 3139 / 3106 / 3179 / 3157 ms)
    // (frame chains, round 4: F is what ONE of a pixel's CHAINS chains renders in this launch; the same item LENGTHS as before -- sixteen / 32 items
    // per pixel and launch over all its chains)
    // (BVH kernels, re-swept with chains on dragon-class, a chain's share F = 128 frames: the whole 2 M-pixel image wants items of 64 frames -- 648 ms
    // against 654 at 32 and 667 at 16 -- and an eighth of its tiles items of 16 -- 93.3 ms against 98.8 at 32 and 108 at 64: what matters is how
    // many items the context's lanes share, so the item shrinks with the pixels the context owns, F / 2 at 2 M pixels down to F / 8)
    const uint32_t bvh_div = c->owned_pixels >= (3u << 19) ? 2u : c->owned_pixels >= (3u << 18) ? 4u : 8u;
    uint32_t item = c->item_frames ? c->item_frames : ((c->cfg.features & rene::FEAT_SMALL) ? std::max(64u, F / (16u / rene::CHAINS)) : std::max(16u, F / bvh_div));
    uint32_t tail = item;
    // (... and for the BVH kernels a halving tail: with chains the end of the job is the end of its last items, not the heaviest pixel's chain --
    // dragon-class, two chains: items of 64 frames 651 ms, halving down to 8 frames 641; the teapot scene 256 -> 16 frames: 3140 -> 3092 ms)
    if (!(c->cfg.features & rene::FEAT_SMALL) && !c->item_frames) tail = std::max(4u, item / 8u);
    if (const char* e = std::getenv("RENE_ITEM_FRAMES")) item = (uint32_t)std::max(1, std::atoi(e));  // tuning knobs
    if (const char* e = std::getenv("RENE_ITEM_TAIL")) tail = (uint32_t)std::max(1, std::atoi(e));
    if (const char* e = std::getenv("RENE_LEVELS")) {
      const uint32_t levels = std::min((uint32_t)std::max(1, std::min((int)rene::MAX_LEVELS, std::atoi(e))), F);
      item = (F + levels - 1) / levels;
      tail = item;
    }
    if (c->item_frames == rene_ctx::kWholeLaunch || (c->opts.flags & RENE_FLAG_SINGLE_LEVEL) || F < 4) item = tail = F;
    // work ids are 32-bit (level * n_work + slot < 2^31) and a version counts at most MAX_LEVELS items
    const uint32_t max_levels = std::max(1u, std::min(rene::MAX_LEVELS, (uint32_t)(0x7fffffffu / std::max(1u, P.n_work))));
    item = std::min(std::max(item, 1u), F);
    for (;;) {
      uint32_t K = F / item, R = F - K * item, H = R ? 1u : 0u;  // K uniform items, then H halving items over the rest R
      if (tail < item && F >= 2 * item) {  // the last uniform item joins the rest: R in [item, 2 item)
        K -= 1;
        R += item;
        H = 1;
        while (H < 16u && (R >> H) >= tail) ++H;  // the last one has ceil(R / 2^(H-1)) >= tail frames
      } else if (tail < item && K == 1 && R == 0) {  // a launch of one item's length: halve that
        K = 0;
        R = F;
        H = 1;
        while (H < 16u && (R >> H) >= tail) ++H;
      }
      if (K + H <= max_levels) {
        P.level_step = item;
        P.n_uniform = K;
        P.n_levels = K + H;
        break;
      }
      item += (item + 7) / 8;  // too many levels: longer items
    }
  }
  P.prev_final = c->prev_final;
  if (c->epoch >= rene::MAX_EPOCH) {  // (drained above) the epoch wraps: every pixel record back to version 0
    hipMemset2DAsync(c->chains + 3, 4 * sizeof(float), 0, sizeof(float), (size_t)rene::CHAINS * c->fb_floats / 4, stream);
    hipMemsetAsync(c->d_item_done, 0, (size_t)rene::CHAINS * c->width * c->height * sizeof(uint32_t), stream);
    c->epoch = 0;
    P.prev_final = 0;
  }
  const uint32_t saved_epoch = c->epoch, saved_prev_final = c->prev_final;
  P.epoch = ++c->epoch;
  c->prev_final = (P.epoch << rene::VERSION_LEVEL_BITS) | P.n_levels;
  // swept with the BVH4 (tools/dev_sweep4.py): dragon-class (Matte) peaks at 24 / 12 (4.96 Grays/s; 20 / 16 gave 4.6);
  // teapot-class, whose logic step is the general-BSDF one, keeps gaining up to ~44 waiting lanes (5.3 vs 4.7)
  P.ready_min = (c->cfg.features & rene::FEAT_GENERAL_BSDF) ? 40 : 24;
  P.leaf_min = 6;  // re-swept in round 3 on whole one-launch jobs: flat from 2 to 12 (dragon-class 686 - 689 ms, the teapot scene 3748 - 3789)
  if (const char* e = std::getenv("RENE_READY_MIN")) P.ready_min = (uint32_t)std::max(1, std::atoi(e));  // tuning knobs
  if (const char* e = std::getenv("RENE_LEAF_MIN")) P.leaf_min = (uint32_t)std::max(1, std::atoi(e));
  if (c->wavefront) {
    // Host-driven rounds of three kernels (wavefront.inc) until every slot has rendered its frames.  The
    // number of rounds is the largest number of bounces any pixel needs over the launch's frames, known
    // only to the device: run a batch sized from the frame count, then poll the done counter (a 4-byte
    // copy + one stream sync per batch).  rene_render is therefore synchronous for these scenes.
    rene::LaunchConfig cfg = c->cfg;
    hipEventRecord(pend.start, c->stream);
    e = rene::launch_wave_init(c->wave, c->stream);
    uint32_t batch = std::max(8u, 2u * P.n_frames);
    uint64_t rounds = 0;
    const uint64_t max_rounds = 64ull + 51ull * P.n_frames;  // depth cap 50 (lib.rs:192) + regeneration rounds
    while (e == hipSuccess) {
      e = rene::launch_wave_rounds(cfg, c->view, P, c->wave, batch, c->stream);
      rounds += batch;
      if (e == hipSuccess) e = hipMemcpyAsync(c->h_done, c->wave.n_done, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess || *c->h_done >= c->wave.n_slots) break;
      if (rounds > max_rounds) {
        hipEventRecord(pend.stop, c->stream);
        c->pending.push_back(pend);
        return fail(RENE_ERR_DEVICE, "wavefront integrator: pixels left unfinished after the maximum number of rounds");
      }
      batch = std::max(8u, P.n_frames / 4u);
    }
    if (e == hipSuccess) e = rene::launch_wave_finish(P, c->wave, c->stream);
    hipEventRecord(pend.stop, c->stream);
    c->pending.push_back(pend);
    c->launches++;
    c->fb_stale = true;
    if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("wavefront launch: ") + hipGetErrorString(e));
    return c->drain();
  }
  rene::LaunchConfig cfg = c->cfg;
  // launch no more lanes than there are work items; hand items out in batches small enough that every
  // launched wave gets some (a tile shard of a small image has fewer items than the chip has lanes)
  const uint32_t total_items = P.n_levels * P.n_work;
  uint32_t blocks_needed = (total_items + rene::render_block_size() - 1) / rene::render_block_size();
  cfg.grid = std::max(1u, std::min(cfg.grid, blocks_needed));
  const uint32_t waves = cfg.grid * (uint32_t)(rene::render_block_size() / 64);
  // 64 ids = one wave's worth: every id a wave takes is rendered at once.  (With 128 the second half of a batch sat reserved
  // until lanes of that wave came free, its pixels started late, and the items that continue from them -- handed out one sweep
  // of the image later -- found them unfinished: Cornell 52.1 -> 48.5 ms per job, docs/history.md section 4f.)
  P.work_batch = 64;
  if (const char* e = std::getenv("RENE_WORK_BATCH")) P.work_batch = (uint32_t)std::max(16, std::min(1024, std::atoi(e)));  // tuning knob
  while (P.work_batch > 16 && (uint64_t)P.work_batch * waves * 2u > total_items) P.work_batch >>= 1;
  hipEventRecord(pend.start, stream);
  rene::g_launched_blocks = cfg.grid;
  e = rene::launch_render(cfg, c->view, P, stream);
  if (e != hipSuccess) {
    // nothing was launched: no pending entry (a replay must not launch what the caller was told failed), no counter slot,
    // and the next launch must not wait for versions this one would have written
    hipEventDestroy(pend.start);
    hipEventDestroy(pend.stop);
    c->epoch = saved_epoch;
    c->prev_final = saved_prev_final;
    c->frames -= n_frames;
    c->paths -= (uint64_t)my_count * c->owned_pixels;
    return fail(RENE_ERR_DEVICE, std::string("render launch: ") + hipGetErrorString(e));
  }
  hipEventRecord(pend.stop, stream);
  pend.epoch = P.epoch;
  pend.replayable = true;
  pend.P = P;
  pend.cfg = cfg;
  c->counters_used++;
  c->pending.push_back(pend);
  c->launches++;
  c->fb_stale = true;
  if (c->pending.size() >= rene_ctx::kCounters) return c->drain();
  return RENE_OK;
}

int rene_sync(rene_ctx* c) {
  if (!c) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_sync: NULL context");
  HIP_TRY(hipSetDevice(c->device));
  return c->drain();
}

int rene_reset(rene_ctx* c) {
  if (!c) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_reset: NULL context");
  HIP_TRY(hipSetDevice(c->device));
  int rc = c->drain();
  if (rc != RENE_OK && !c->handoff_failed) return rc;
  c->handoff_failed = false;  // the counters are cleared below and the image starts again from zero
  c->exchanged = false;
  if (c->opts.shard_mode == RENE_SHARD_TILES && c->opts.shard_count > 1) {  // a tile shard's chains hold something in the tiles it owns only
    hipError_t e = rene::launch_chains_tiles(c->chains, c->fb, true, c->width, c->height, c->tiles_x, c->n_tiles, c->opts.shard_rank, c->opts.shard_count, c->stream);
    if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_reset: ") + hipGetErrorString(e));
  } else {
    HIP_TRY(hipMemsetAsync(c->chains, 0, (size_t)rene::CHAINS * c->fb_floats * sizeof(float), c->stream));
  }
  HIP_TRY(hipMemsetAsync(c->fb, 0, c->fb_floats * sizeof(float), c->stream));
  c->fb_stale = false;
  HIP_TRY(hipMemsetAsync(c->d_counters, 0, 32 * sizeof(unsigned long long), c->stream));
  HIP_TRY(hipMemsetAsync(c->d_item_done, 0, (size_t)rene::CHAINS * c->width * c->height * sizeof(uint32_t), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->prev_final = 0;  // the pixel records carry version 0 again
  c->frames = 0;
  c->paths = 0;
  c->launches = 0;
  c->replays = 0;
  c->kernel_ms = 0.0;
  c->last_ms = 0.0;
  return RENE_OK;
}

int rene_tune(rene_ctx* c, uint32_t n_frames) {
  if (!c) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_tune: NULL context");
  if (c->wavefront || n_frames < 4 || c->n_work == 0 || (c->opts.flags & RENE_FLAG_SINGLE_LEVEL)) return RENE_OK;
  HIP_TRY(hipSetDevice(c->device));
  int rc = c->drain();
  if (rc != RENE_OK) return rc;
  const uint32_t saved = c->item_frames;
  uint32_t best = saved;
  double best_ms = 0.0;
  // candidates: one item per pixel, chain and launch, then items of 256 / 128 / 64 / 32 / 16 frames (of the chain's share of the launch)
  const uint32_t items[6] = {rene_ctx::kWholeLaunch, 256u, 128u, 64u, 32u, 16u};
  const uint32_t chain_frames = (n_frames + rene::CHAINS - 1u) / rene::CHAINS;
  for (int i = 0; i < 6; ++i) {
    if (i > 0 && items[i] * 2u > chain_frames) continue;
    c->item_frames = items[i];
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 3 && rc == RENE_OK; ++k) rc = rene_render(c, 0, n_frames);  // what is rendered does not matter
    if (rc == RENE_OK) rc = c->drain();
    if (rc != RENE_OK) {
      c->item_frames = saved;
      return rc;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (i == 0 || ms < 0.985 * best_ms) {  // finer items must pay for their bookkeeping
      best_ms = ms;
      best = c->item_frames;
    }
  }
  c->item_frames = best;
  if (std::getenv("RENE_DEBUG"))
    std::fprintf(stderr, "[rene] tuned: work items of %u frames for launches of %u frames\n", best == rene_ctx::kWholeLaunch ? n_frames : best, n_frames);
  return rene_reset(c);
}

int rene_framebuffer(rene_ctx* c, void** device_ptr, size_t* n_floats) {
  if (!c || !device_ptr) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_framebuffer: NULL argument");
  *device_ptr = c->fb;
  if (n_floats) *n_floats = c->fb_floats;
  HIP_TRY(hipSetDevice(c->device));
  {  // frame chains: the image is the chains added together, which a drain does (not in stream order)
    int rc = c->drain();
    if (rc != RENE_OK) return rc;
  }
  return c->flush_exchange();
}

static int rene_download_impl(rene_ctx* c, int layer, int channels, float* dst, size_t dst_floats) {
  if (!c || !dst) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_download: NULL argument");
  if (layer < 0 || layer >= RENE_LAYER_COUNT) return fail(RENE_ERR_INVALID_ARGUMENT, "layer out of range");
  if (channels != 3 && channels != 4) return fail(RENE_ERR_INVALID_ARGUMENT, "channels must be 3 or 4");
  size_t n = (size_t)c->width * c->height;
  if (dst_floats < n * (size_t)channels) return fail(RENE_ERR_INVALID_ARGUMENT, "destination too small");
  HIP_TRY(hipSetDevice(c->device));
  int rc = c->drain();
  if (rc != RENE_OK) return rc;
  const float* src = c->fb + (size_t)layer * n * 4;
  // through a pinned staging buffer the context keeps: a copy into pageable memory makes the runtime pin and unpin the
  // destination's pages on the fly (a registration of user memory with the driver per call)
  if (!c->h_stage) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), n * 4 * sizeof(float), hipHostMallocDefault));
  HIP_TRY(hipMemcpy(c->h_stage, src, n * 4 * sizeof(float), hipMemcpyDeviceToHost));
  const float* tmp = c->h_stage;
  if (channels == 4) {
    for (size_t i = 0; i < n; ++i) {
      dst[4 * i] = tmp[4 * i];
      dst[4 * i + 1] = tmp[4 * i + 1];
      dst[4 * i + 2] = tmp[4 * i + 2];
      dst[4 * i + 3] = 0.0f;  // the device keeps a record's version there; rene's alpha stays 0 (lib.rs:170)
    }
    return RENE_OK;
  }
  for (size_t i = 0; i < n; ++i) {  // f32_4_to_3, rene/src/main.rs:1749-1756
    dst[3 * i] = tmp[4 * i];
    dst[3 * i + 1] = tmp[4 * i + 1];
    dst[3 * i + 2] = tmp[4 * i + 2];
  }
  return RENE_OK;
}

int rene_get_stats(rene_ctx* c, rene_stats* out) {
  if (!c || !out) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_get_stats: NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  int rc = c->drain();
  if (rc != RENE_OK) return rc;
  unsigned long long h[32];
  HIP_TRY(hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
  if (std::getenv("RENE_DEBUG") && h[23])  // RENE_FLAG_COUNTERS on the megakernels of device_code.inc: where the lanes of a pass were
    std::fprintf(stderr, "[rene] passes %llu (wave executions of the loop); lanes per pass: on a path %.3f, waiting for a hand-off %.3f, out of work %.3f, starting a path %.3f; "
                         "passes with an item switch %.4f, with a poll %.4f\n", h[12], (double)h[13] / (64.0 * (double)h[12]), (double)h[14] / (64.0 * (double)h[12]),
                 (double)h[15] / (64.0 * (double)h[12]), (double)h[16] / (64.0 * (double)h[12]), (double)h[17] / (double)h[12], (double)h[18] / (double)h[12]);
  else if (std::getenv("RENE_DEBUG") && h[12])  // RENE_FLAG_COUNTERS on the traversal-restart kernel: lanes active per step kind
    std::fprintf(stderr, "[rene] steps (wave executions, lanes, lanes / 64 per execution): node %llu %llu %.3f | leaf %llu %llu %.3f | logic %llu %llu %.3f | iterations %llu\n",
                 h[12], h[6], (double)h[6] / (64.0 * (double)h[12]), h[13], h[16], (double)h[16] / (64.0 * (double)std::max(1ull, h[13])), h[14], h[15],
                 (double)h[15] / (64.0 * (double)std::max(1ull, h[14])), h[17]);
  if (std::getenv("RENE_DEBUG") && h[12] && !h[23] && h[17])  // ... and by the node step's position in its iteration: a wave's density decays from step to step
    std::fprintf(stderr, "[rene] lanes at an inner node per iteration, at its first / second / third node step: %.3f %.3f %.3f of 64\n",
                 (double)h[28] / (64.0 * (double)h[17]), (double)h[29] / (64.0 * (double)h[17]), (double)h[30] / (64.0 * (double)h[17]));
  if (std::getenv("RENE_DEBUG") && h[12] && !h[23])
    std::fprintf(stderr, "[rene] node visits %llu: nothing hit %llu (%.3f), reached by a pop %llu (%.3f), both %llu (%.3f); in the top levels %llu (%.3f); deepest stack %llu entries\n", h[6], h[18],
                 (double)h[18] / (double)h[6], h[19], (double)h[19] / (double)h[6], h[20], (double)h[20] / (double)h[6], h[22], (double)h[22] / (double)h[6], h[21]);
  std::memset(out, 0, sizeof(*out));
  out->rays_closest = h[0];
  out->rays_shadow = h[1];
  out->rays_emitter = h[2];
  out->paths = c->paths;  // frames rendered x pixels owned
  out->hits = h[4];
  out->bounces = h[4];
  out->adds = h[5];
  out->node_visits = h[6];
  out->prim_tests = h[7];
  out->frames = c->frames;
  out->launches = c->launches + c->replays;  // launches that had to be launched again (drain) count twice
  out->kernel_ms = c->kernel_ms;
  out->last_launch_ms = c->last_ms;
  // the engine clock while the launches ran: shader-clock ticks (s_memtime) per tick of the constant 100 MHz clock
  // (s_memrealtime) over the lifetime of one wave of each launch (device_code.inc, clock_probe)
  out->sclk_mhz = h[25] ? 100.0 * (double)h[24] / (double)h[25] : 0.0;
  return RENE_OK;
}

int rene_trace(rene_ctx* c, int which, size_t n, const float* origins, const float* directions, float tmin,
               float tmax, rene_hit* out) {
  if (!c || (n && (!origins || !directions || !out))) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_trace: NULL argument");
  if (which != 0 && which != 1) return fail(RENE_ERR_INVALID_ARGUMENT, "which must be 0 (main) or 1 (emitters)");
  // the traversal orders children by the bit pattern of their (non-negative) entry distance
  if (!(tmin >= 0.0f) || !(tmax >= tmin)) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_trace: need 0 <= tmin <= tmax");
  if (n == 0) return RENE_OK;
  if (n > 0x7fffffffull) return fail(RENE_ERR_INVALID_ARGUMENT, "too many rays in one batch");
  HIP_TRY(hipSetDevice(c->device));
  float *d_o = nullptr, *d_d = nullptr;
  rene_hit* d_h = nullptr;
  auto cleanup = [&]() {
    hipFree(d_o);
    hipFree(d_d);
    hipFree(d_h);
  };
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_o), n * 12);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_d), n * 12);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_h), n * sizeof(rene_hit));
  if (e == hipSuccess) e = hipMemcpyAsync(d_o, origins, n * 12, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_d, directions, n * 12, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = rene::launch_trace(c->cfg, c->view, which, (uint32_t)n, d_o, d_d, tmin, tmax, d_h, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_h, n * sizeof(rene_hit), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  cleanup();
  if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_trace: ") + hipGetErrorString(e));
  return RENE_OK;
}

// ---- the J1 gate (probes; DESIGN.md section 9, tools/j1_gate.py) ---------------------------------------------------------------------------------
int rene_ray_dump(rene_ctx* c, uint32_t first_frame, uint32_t n_frames, size_t capacity, float* rays8, uint64_t* n_issued) {
  if (!c || !rays8 || !n_issued || capacity == 0 || capacity > 0x7fffffffull / 8) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_ray_dump: bad argument");
  if (!(c->opts.flags & RENE_FLAG_COUNTERS) || c->wavefront || (c->cfg.features & (rene::FEAT_SMALL | rene::FEAT_VOLPATH)) || (c->opts.flags & RENE_FLAG_NO_RESTART) ||
      c->view.main.n_nodes <= 512u)
    return fail(RENE_ERR_UNSUPPORTED, "rene_ray_dump: a context with RENE_FLAG_COUNTERS whose scene the traversal-restart kernel renders (path integrator, more than 512 BVH nodes)");
  HIP_TRY(hipSetDevice(c->device));
  int rc = c->drain();
  if (rc != RENE_OK) return rc;
  float* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), (capacity + 1) * 8 * sizeof(float)));
  hipError_t e = c->zero_now(d, 8 * sizeof(float));
  if (e != hipSuccess) { hipFree(d); return fail(RENE_ERR_DEVICE, std::string("rene_ray_dump: ") + hipGetErrorString(e)); }
  c->ray_dump = d;
  c->ray_dump_cap = (uint32_t)capacity;
  rc = rene_render(c, first_frame, n_frames);
  if (rc == RENE_OK) rc = c->drain();
  c->ray_dump = nullptr;
  c->ray_dump_cap = 0;
  uint32_t issued = 0;
  if (rc == RENE_OK) {
    e = hipMemcpy(&issued, d, sizeof(issued), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(rays8, d + 8, std::min<size_t>(issued, capacity) * 8 * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(RENE_ERR_DEVICE, std::string("rene_ray_dump: ") + hipGetErrorString(e));
  }
  hipFree(d);
  *n_issued = issued;
  return rc;
}

int rene_trace_queue(rene_ctx* c, size_t n, const float* o_tmax4, const void* d_flags, int fp16, uint32_t refill_min, uint32_t leaf_min, uint32_t blocks_per_cu,
                     uint32_t repeats, float* hits4, float* ms_out, uint64_t* steps5) {
  if (!c || !n || !o_tmax4 || !d_flags || n > 0x7fffffffull) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_trace_queue: bad argument");
  if (c->cfg.features & rene::FEAT_SMALL) return fail(RENE_ERR_UNSUPPORTED, "rene_trace_queue: BVH scenes only");
  HIP_TRY(hipSetDevice(c->device));
  int rc = c->drain();
  if (rc != RENE_OK) return rc;
  float *d_o = nullptr, *d_h = nullptr;
  uint32_t *d_d = nullptr, *d_cnt = nullptr;
  unsigned long long* d_steps = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  auto cleanup = [&]() {
    hipFree(d_o); hipFree(d_h); hipFree(d_d); hipFree(d_cnt); hipFree(d_steps);
    if (ev0) hipEventDestroy(ev0);
    if (ev1) hipEventDestroy(ev1);
  };
  const size_t dbytes = n * (fp16 ? 8 : 16);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_o), n * 16);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_d), dbytes);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_h), n * 16);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_cnt), 64 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_steps), 8 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemcpy(d_o, o_tmax4, n * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_d, d_flags, dbytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipEventCreate(&ev0);
  if (e == hipSuccess) e = hipEventCreate(&ev1);
  float best_ms = 0.0f;
  for (uint32_t k = 0; e == hipSuccess && k < std::max(1u, repeats); ++k) {
    e = c->zero_now(d_cnt, 64 * sizeof(uint32_t));
    if (e == hipSuccess) e = c->zero_now(d_steps, 8 * sizeof(unsigned long long));
    rene::TraceQueue Q{};
    Q.o_tmax = d_o; Q.d_flags = d_d; Q.hits = d_h; Q.counter = d_cnt; Q.n = (uint32_t)n;
    Q.refill_min = std::max(1u, std::min(64u, refill_min)); Q.leaf_min = std::max(1u, leaf_min); Q.fp16 = fp16 ? 1u : 0u;
    Q.stack_entries = c->cfg.stack_depth;
    Q.passes = 1;
    if (const char* ev = std::getenv("RENE_GATE_PASSES")) Q.passes = (uint32_t)std::max(1, std::min((int)(0x7fffffffull / n), std::atoi(ev)));  // (probe knob: a longer launch)
    if (e == hipSuccess) e = hipEventRecord(ev0, c->stream);
    if (e == hipSuccess) e = rene::launch_trace_queue(c->cfg, c->view, Q, blocks_per_cu, steps5 ? d_steps : nullptr, c->stream);
    if (e == hipSuccess) e = hipEventRecord(ev1, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev0, ev1);
    if (k == 0 || ms < best_ms) best_ms = ms;
  }
  if (e == hipSuccess && hits4) e = hipMemcpy(hits4, d_h, n * 16, hipMemcpyDeviceToHost);
  if (e == hipSuccess && steps5) e = hipMemcpy(steps5, d_steps, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_trace_queue: ") + hipGetErrorString(e));
  if (ms_out) *ms_out = best_ms;
  return RENE_OK;
}

int rene_bsdf_eval(rene_ctx* c, uint32_t material_index, size_t n, const float* normals3, const float* uvs2,
                   const float* wo3, const float* wi3, const uint32_t* seeds, float* out12) {
  if (!c || (n && (!normals3 || !uvs2 || !wo3 || !wi3 || !seeds || !out12)))
    return fail(RENE_ERR_INVALID_ARGUMENT, "rene_bsdf_eval: NULL argument");
  if (material_index >= c->n_materials) return fail(RENE_ERR_INVALID_ARGUMENT, "material_index out of range");
  if (n == 0) return RENE_OK;
  if (n > (1u << 24)) return fail(RENE_ERR_INVALID_ARGUMENT, "too many items in one batch");
  HIP_TRY(hipSetDevice(c->device));
  const size_t sizes[6] = {n * 12, n * 8, n * 12, n * 12, n * 4, n * 48};
  const void* src[5] = {normals3, uvs2, wo3, wi3, seeds};
  void* d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipMalloc(&d[i], sizes[i]);
  for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipMemcpyAsync(d[i], src[i], sizes[i], hipMemcpyHostToDevice, c->stream);
  // the first instance that uses the material lends its record, so that the probe runs what a render runs (the material
  // resolved at upload, Inst::res_*); a material no instance uses goes through the material / texture tables
  int inst_index = -1;
  for (size_t k = 0; k < c->inst_material.size() && inst_index < 0; ++k)
    if (c->inst_material[k] == material_index) inst_index = (int)k;
  if (e == hipSuccess)
    e = rene::launch_bsdf_eval(c->view, material_index, inst_index, (uint32_t)n, (const float*)d[0], (const float*)d[1],
                               (const float*)d[2], (const float*)d[3], (const uint32_t*)d[4], (float*)d[5], c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out12, d[5], sizes[5], hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  for (int i = 0; i < 6; ++i) hipFree(d[i]);
  if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_bsdf_eval: ") + hipGetErrorString(e));
  return RENE_OK;
}

int rene_medium_eval(rene_ctx* c, uint32_t medium_index, size_t n, const float* rd3, const float* t_max,
                     const float* wo3, const float* wi3, const uint32_t* seeds, float* out16) {
  if (!c || (n && (!rd3 || !t_max || !wo3 || !wi3 || !seeds || !out16)))
    return fail(RENE_ERR_INVALID_ARGUMENT, "rene_medium_eval: NULL argument");
  if (medium_index >= c->n_mediums) return fail(RENE_ERR_INVALID_ARGUMENT, "medium_index out of range (media exist only under the volpath integrator)");
  if (n == 0) return RENE_OK;
  if (n > (1u << 24)) return fail(RENE_ERR_INVALID_ARGUMENT, "too many items in one batch");
  HIP_TRY(hipSetDevice(c->device));
  const size_t sizes[6] = {n * 12, n * 4, n * 12, n * 12, n * 4, n * 64};
  const void* src[5] = {rd3, t_max, wo3, wi3, seeds};
  void* d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipMalloc(&d[i], sizes[i]);
  for (int i = 0; i < 5 && e == hipSuccess; ++i) e = hipMemcpyAsync(d[i], src[i], sizes[i], hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess)
    e = rene::launch_medium_eval(c->view, medium_index, (uint32_t)n, (const float*)d[0], (const float*)d[1],
                                 (const float*)d[2], (const float*)d[3], (const uint32_t*)d[4], (float*)d[5], c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out16, d[5], sizes[5], hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  for (int i = 0; i < 6; ++i) hipFree(d[i]);
  if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_medium_eval: ") + hipGetErrorString(e));
  return RENE_OK;
}

int rene_emitter_pdf(rene_ctx* c, size_t n, const float* origins, const float* directions, float* out) {
  if (!c || (n && (!origins || !directions || !out))) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_emitter_pdf: NULL argument");
  if (n == 0) return RENE_OK;
  if (n > 0x7fffffffull) return fail(RENE_ERR_INVALID_ARGUMENT, "too many rays in one batch");
  HIP_TRY(hipSetDevice(c->device));
  float *d_o = nullptr, *d_d = nullptr, *d_p = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_o), n * 12);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_d), n * 12);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_p), n * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(d_o, origins, n * 12, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_d, directions, n * 12, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = rene::launch_emitter_pdf(c->cfg, c->view, (uint32_t)n, d_o, d_d, d_p, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_p, n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_o);
  hipFree(d_d);
  hipFree(d_p);
  if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_emitter_pdf: ") + hipGetErrorString(e));
  return RENE_OK;
}

int rene_pcg_probe(int device, uint32_t seed, uint32_t n, uint32_t* out) {
  if (n && !out) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_pcg_probe: NULL argument");
  if (n == 0) return RENE_OK;
  int n_dev = 0;
  HIP_TRY(hipGetDeviceCount(&n_dev));
  if (device < 0 || device >= n_dev) return fail(n_dev <= 0 ? RENE_ERR_DEVICE : RENE_ERR_INVALID_ARGUMENT, "rene_pcg_probe: no such HIP device");
  HIP_TRY(hipSetDevice(device));
  uint32_t* d = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), (size_t)n * 4));
  hipError_t e = rene::launch_pcg_probe(seed, n, d, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, d, (size_t)n * 4, hipMemcpyDeviceToHost);
  hipFree(d);
  if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_pcg_probe: ") + hipGetErrorString(e));
  return RENE_OK;
}

// ---- multi-GPU exchange step: RCCL over xGMI (include/rene_hip.h) ------------------------------------------------
static_assert(RENE_COMM_ID_BYTES == sizeof(ncclUniqueId), "rene_comm_unique_id hands out an ncclUniqueId");

int rene_comm_unique_id(uint8_t id[RENE_COMM_ID_BYTES]) {
  if (!id) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_comm_unique_id: NULL argument");
  Rccl* R = rccl();
  if (!R->handle) return fail(RENE_ERR_UNSUPPORTED, R->error);
  ncclUniqueId u;
  RCCL_TRY(R->GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return RENE_OK;
}

int rene_comm_init(rene_ctx* c, int n_ranks, int rank, const uint8_t id[RENE_COMM_ID_BYTES]) {
  if (!c || !id) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_comm_init: NULL argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_comm_init: rank out of range");
  if (c->comm) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_comm_init: the context already belongs to a communicator");
  Rccl* R = rccl();
  if (!R->handle) return fail(RENE_ERR_UNSUPPORTED, R->error);
  HIP_TRY(hipSetDevice(c->device));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  RCCL_TRY(R->CommInitRank(&c->comm, n_ranks, u, rank));
  c->comm_ranks = n_ranks;
  c->comm_rank = rank;
  return RENE_OK;
}

int rene_comm_init_all(rene_ctx** ctxs, int n) {
  if (!ctxs || n < 1) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_comm_init_all: NULL argument");
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i] || ctxs[i]->comm) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_comm_init_all: NULL context or context already in a communicator");
    devs[i] = ctxs[i]->device;
  }
  Rccl* R = rccl();
  if (!R->handle) return fail(RENE_ERR_UNSUPPORTED, R->error);
  std::vector<ncclComm_t> comms(n);
  RCCL_TRY(R->CommInitAll(comms.data(), n, devs.data()));
  for (int i = 0; i < n; ++i) {
    ctxs[i]->comm = comms[i];
    ctxs[i]->comm_ranks = n;
    ctxs[i]->comm_rank = i;
  }
  return RENE_OK;
}

int rene_comm_group_begin(void) {
  Rccl* R = rccl();
  if (!R->handle) return fail(RENE_ERR_UNSUPPORTED, R->error);
  RCCL_TRY(R->GroupStart());
  return RENE_OK;
}
int rene_comm_group_end(void) {
  Rccl* R = rccl();
  if (!R->handle) return fail(RENE_ERR_UNSUPPORTED, R->error);
  RCCL_TRY(R->GroupEnd());
  return RENE_OK;
}

int rene_reduce(rene_ctx* c, int root) {
  if (!c) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_reduce: NULL context");
  if (!c->comm) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_reduce: rene_comm_init first");
  if (root < 0 || root >= c->comm_ranks) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_reduce: root out of range");
  Rccl* R = rccl();
  HIP_TRY(hipSetDevice(c->device));
  {  // every launch has completed (and any dropped work item has been rendered by a replay) before the image is summed
    int rc = c->drain();
    if (rc != RENE_OK) return rc;
  }
  c->exchanged = true;
  RCCL_TRY(R->Reduce(c->fb, c->fb, c->fb_floats, ncclFloat, ncclSum, root, c->comm, c->stream));
  return RENE_OK;
}

int rene_gather_tiles(rene_ctx* c, int root) {
  if (!c) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_gather_tiles: NULL context");
  if (!c->comm) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_gather_tiles: rene_comm_init first");
  if (root < 0 || root >= c->comm_ranks) return fail(RENE_ERR_INVALID_ARGUMENT, "rene_gather_tiles: root out of range");
  const uint32_t n = (uint32_t)c->comm_ranks;
  if (c->opts.shard_mode != RENE_SHARD_TILES || c->opts.shard_count != n || c->opts.shard_rank != (uint32_t)c->comm_rank)
    return fail(RENE_ERR_INVALID_ARGUMENT, "rene_gather_tiles: the context must be tile-sharded with shard_count == n_ranks and shard_rank == rank");
  Rccl* R = rccl();
  HIP_TRY(hipSetDevice(c->device));
  {  // as rene_reduce: what is packed and sent is a complete image
    int rc = c->drain();
    if (rc != RENE_OK) return rc;
  }
  c->exchanged = true;
  const size_t tile_floats = (size_t)3 * RENE_TILE_SIZE * RENE_TILE_SIZE * 4;
  auto owned = [&](uint32_t r) { return c->n_tiles > r ? (c->n_tiles - r + n - 1) / n : 0u; };
  const bool is_root = c->comm_rank == root;
  // packed staging: a sender its own tiles; the root those of every other rank, one after the other
  size_t need = 0;
  if (n == 1) need = owned(0) * tile_floats;
  else if (is_root) { for (uint32_t r = 0; r < n; ++r) if ((int)r != root) need += owned(r) * tile_floats; }
  else need = owned((uint32_t)c->comm_rank) * tile_floats;
  if (need > c->tile_buf_floats) {
    if (c->tile_buf) hipFree(c->tile_buf);
    c->tile_buf = nullptr;
    c->tile_buf_floats = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->tile_buf), std::max<size_t>(16, need * sizeof(float))));
    c->tile_buf_floats = need;
  }
  if (n == 1) {
    // a communicator of one has nothing to send; it still packs its tiles, clears the image and places them again, so
    // that the two kernels of the exchange run (and are tested) wherever the library does
    hipError_t e = rene::launch_pack_tiles(c->fb, c->tile_buf, c->width, c->height, c->tiles_x, c->n_tiles, 0u, 1u, false, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->fb, 0, c->fb_floats * sizeof(float), c->stream);
    if (e == hipSuccess) e = rene::launch_pack_tiles(c->fb, c->tile_buf, c->width, c->height, c->tiles_x, c->n_tiles, 0u, 1u, true, c->stream);
    if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_gather_tiles: ") + hipGetErrorString(e));
    return RENE_OK;
  }
  if (!is_root) {
    hipError_t e = rene::launch_pack_tiles(c->fb, c->tile_buf, c->width, c->height, c->tiles_x, c->n_tiles, (uint32_t)c->comm_rank, n, false, c->stream);
    if (e != hipSuccess) return fail(RENE_ERR_DEVICE, std::string("rene_gather_tiles pack: ") + hipGetErrorString(e));
    if (need) RCCL_TRY(R->Send(c->tile_buf, need, ncclFloat, root, c->comm, c->stream));
    return RENE_OK;
  }
  RCCL_TRY(R->GroupStart());
  size_t off = 0;
  for (uint32_t r = 0; r < n; ++r) {
    if ((int)r == root || owned(r) == 0) continue;
    ncclResult_t rr = R->Recv(c->tile_buf + off, owned(r) * tile_floats, ncclFloat, (int)r, c->comm, c->stream);
    if (rr != ncclSuccess) { R->GroupEnd(); return fail(RENE_ERR_DEVICE, std::string("ncclRecv: ") + R->GetErrorString(rr)); }
    off += owned(r) * tile_floats;
  }
  RCCL_TRY(R->GroupEnd());
  // the received tiles are placed by the next call that waits for or hands out the image (rene_sync, rene_download,
  // rene_get_stats, rene_framebuffer): inside a caller's rene_comm_group_begin / _end the receives above are only
  // enqueued when the outermost group ends, and the placing kernels must come behind them on the stream
  c->unpack_root = root;
  return RENE_OK;
}

// ---- output transform, rene/src/main.rs:1758-1810 ---------------------------------------------------------
static float gamma_correct(float v) {  // main.rs:1768-1774
  if (v <= 0.0031308f) return 12.92f * v;
  return 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
}
static uint8_t sat_u8(float v) {  // Rust `as u8`: saturating, NaN -> 0
  if (!(v > 0.0f)) return 0;
  if (v >= 255.0f) return 255;
  return (uint8_t)v;
}
void rene_to_rgb8(const float* sums, size_t n_floats, uint32_t n_samples, uint8_t* out) {
  const float denom = (float)n_samples;
  for (size_t i = 0; i < n_floats; ++i) {
    float v = sums[i] / denom;                              // average, main.rs:1758-1764
    float r = std::round(255.0f * gamma_correct(v));        // to_rgb8, main.rs:1785-1792
    out[i] = sat_u8(std::fmin(std::fmax(r, 0.0f), 255.0f));
  }
}
void rene_to_aov8(const float* sums, size_t n_floats, uint32_t n_samples, int is_normal, uint8_t* out) {
  const float denom = (float)n_samples;
  for (size_t i = 0; i < n_floats; ++i) {
    float v = sums[i] / denom;
    if (is_normal) v = v * 0.5f + 0.5f;                                 // to_aov_normal, main.rs:1803-1810
    out[i] = sat_u8(256.0f * std::fmin(std::fmax(v, 0.0f), 0.999f));  // to_aov, main.rs:1794-1801
  }
}

int rene_scene_pack_info(const rene_scene_desc* scene, rene_pack_info* out) { return guarded([&] { return rene_scene_pack_info_impl(scene, out); }); }
int rene_create(const rene_scene_desc* scene, const rene_opts* opts, rene_ctx** out) { return guarded([&] { return rene_create_impl(scene, opts, out); }); }
int rene_render(rene_ctx* c, uint32_t first_frame, uint32_t n_frames) { return guarded([&] { return rene_render_impl(c, first_frame, n_frames); }); }
int rene_download(rene_ctx* c, int layer, int channels, float* dst, size_t dst_floats) { return guarded([&] { return rene_download_impl(c, layer, channels, dst, dst_floats); }); }
}  // extern "C"
