// kernels.h -- host-visible launch wrappers of kernels.hip
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rene_hip.h"
#include "device_scene.h"

namespace rene {

constexpr uint32_t RENE_FLAG_INTERNAL_TEST_DROP = 1u << 29;  // RenderParams.flags, set by rene_render under RENE_TEST_DROP=<launch>: some items of that
                                                             // launch are dropped as if their hand-off had timed out (tests of the replay)

struct LaunchConfig {
  uint32_t features = 0;     // FEAT_* of the scene
  uint32_t stack_depth = 16; // LDS traversal stack entries per lane
  uint32_t grid = 0;         // workgroups of the persistent render launch (upper bound)
  uint32_t cus = 0;          // compute units of the device
  uint32_t wave_stack = 0;   // wavefront integrator: LDS traversal stack entries per lane (exact tree depth)
  uint32_t n_insts = 0;      // instances of the scene (the restart kernels keep a small table of them in LDS)
};

// workgroups of the calling thread's most recent persistent render launch after fit_grid's clamp (rene_hip.cpp)
extern thread_local uint32_t g_launched_blocks;
hipError_t launch_render(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st);
hipError_t launch_render_bvh(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st);
hipError_t launch_render_vol(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, hipStream_t st);
// stage-separated wavefront integrator (wavefront.inc, kernels_wave.hip): path state in HBM, SoA
struct WaveState {
  float4* ro;        // xyz ray origin
  float4* rd;        // xyz ray direction
  float4* color;     // rgb throughput; w = BSDF pdf awaiting the emitter-pdf ray
  uint4* ctl;        // x rng state, y frame-wide rng state, z depth, w next frame of this slot
  float4* hit;       // t, u, v, bits(slot)
  float4* sh_wi;     // [max_lights][n_slots] shadow ray direction (xyz)
  float4* sh_c;      // [max_lights][n_slots] contribution if unoccluded (rgb)
  uint32_t* status;  // WS_* bits
  uint32_t* n_done;  // slots that have finished all their frames
  uint32_t* trace_counter;  // [2] next ray id of the closest-hit / secondary traversal pass
  unsigned long long* wave_sums;  // [waves][8] per-wave counter rows (folded into the context's counters at the end)
  uint32_t n_slots;
  uint32_t max_lights;
  uint32_t fp16_payload;  // RENE_FLAG_FP16_PAYLOAD: direction and throughput of a slot as six halves in ONE float4 (Q.rd), Q.color unused
  uint32_t pad_;
};
hipError_t launch_wave_init(const WaveState& Q, hipStream_t st);
hipError_t launch_wave_finish(const RenderParams& P, const WaveState& Q, hipStream_t st);
hipError_t launch_wave_rounds(const LaunchConfig& cfg, const SceneView& S, const RenderParams& P, const WaveState& Q,
                              uint32_t rounds, hipStream_t st);
hipError_t launch_trace(const LaunchConfig& cfg, const SceneView& S, int which, uint32_t n, const float* o,
                        const float* d, float tmin, float tmax, rene_hit* out, hipStream_t st);
hipError_t launch_bsdf_eval(const SceneView& S, uint32_t material, int inst_index, uint32_t n, const float* nrm3, const float* uv,
                            const float* wo3, const float* wi3, const uint32_t* seeds, float* out, hipStream_t st);
hipError_t launch_medium_eval(const SceneView& S, uint32_t medium, uint32_t n, const float* rd3, const float* t_max,
                              const float* wo3, const float* wi3, const uint32_t* seeds, float* out, hipStream_t st);
hipError_t launch_emitter_pdf(const LaunchConfig& cfg, const SceneView& S, uint32_t n, const float* o, const float* d, float* out, hipStream_t st);
hipError_t launch_pcg_probe(uint32_t seed, uint32_t n, uint32_t* out, hipStream_t st);
// tile-sharded exchange: the 32x32 tiles with index % shard_count == shard_rank of a [3][H][W][4] image <-> a packed
// buffer [owned tile][layer][32][32][4] (out-of-image texels of edge tiles are zero / skipped)
hipError_t launch_pack_tiles(const float* fb, float* packed, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles,
                             uint32_t shard_rank, uint32_t shard_count, bool unpack, hipStream_t st);
// rene_trace_queue (probe: the J1 gate): a traversal-only persistent pass over a queue-ordered SoA ray buffer
struct TraceQueue {
  const float* o_tmax;      // [n][4] origin, tmax
  const uint32_t* d_flags;  // [n][2] direction as three halves + flags in the fourth (bit 0: any-hit, bit 1: emitter-only structure) -- or,
                            // fp32 payload, [n][4] direction as three floats + flags
  float* hits;              // [n][4] t (-1: miss), u, v, bits(slot)
  uint32_t* counter;        // next ray of the queue (zeroed by the host)
  uint32_t n;
  uint32_t refill_min;      // dead lanes a wave gathers before it fetches rays for them (64: only when the whole wave is done)
  uint32_t leaf_min;        // lanes at a leaf before the leaf step runs
  uint32_t fp16;            // direction payload: 1 = three halves (8 bytes), 0 = three floats (16 bytes)
  uint32_t stack_entries;
  uint32_t passes;          // the queue is traversed this many times over in one launch (>= 1)
};
// rene_trace_queue (probe, kernels_gate.hip): a traversal-only persistent pass over a queue-ordered ray buffer; step_counters (optional): 5 x u64
hipError_t launch_trace_queue(const LaunchConfig& cfg, const SceneView& S, const TraceQueue& Q, uint32_t blocks_per_cu, unsigned long long* step_counters, hipStream_t st);
// frame chains: out[3][H][W][4] = the CHAINS images of `chains` added in chain order (alpha 0); the chains are left as they are
hipError_t launch_resolve_chains(const float* chains, float* out, size_t image_floats, hipStream_t st);
// the same over the tiles a tile shard owns (tile t with t % shard_count == shard_rank); zero = true: clears the chains there instead
hipError_t launch_chains_tiles(float* chains, float* out, bool zero, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles,
                               uint32_t shard_rank, uint32_t shard_count, hipStream_t st);
int render_block_size();

}  // namespace rene
