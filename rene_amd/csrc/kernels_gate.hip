// kernels_gate.hip -- the J1 gate: what does TRAVERSAL ALONE cost when its rays come from a queue?
// north_star names "a persistent-threads wavefront integrator with ray queues sorted / compacted in LDS"; the megakernels keep path state in
// registers and fill lanes by regeneration instead, and their BVH steps run at 0.4 - 0.6 of the lanes.  This unit measures the other design's
// traversal stage in isolation, on rays dumped from a real frame mix (rene_ray_dump) and laid out the way a wavefront's shading pass would
// have queued them: a persistent pass over a queue-ordered SoA buffer -- 16-byte origin + tmax, 8-byte direction as three halves + flags (J2's
// payload) or 16-byte fp32 direction -- whose lanes take the next rays of the queue as soon as `refill_min` of a wave's lanes are free (one atomic
// per refill, ballot + prefix rank: the dead lanes are re-packed with live rays, which is what compacting the queue in LDS buys), the same node /
// leaf steps as the traversal-restart kernel (node4_test, intersect_leaf; leaves postponed until leaf_min lanes have one), hits written 16 bytes
// per ray.  No shading, no path state: the number is an upper bound on what a wavefront integrator's two traversal passes can run at.
#include "device_code.inc"  // opens namespace rene

typedef __fp16 gate_half2 __attribute__((ext_vector_type(2)));
RENE_DEV void unpack_h2(float w, float& a, float& b) {
  const gate_half2 h = __builtin_bit_cast(gate_half2, w);
  a = (float)h.x;
  b = (float)h.y;
}
#ifndef RENE_GATE_CHUNK
#define RENE_GATE_CHUNK 512
#endif
#ifndef RENE_GATE_NODE_STEPS
#define RENE_GATE_NODE_STEPS 3
#endif
template <bool FP16>
__global__ void __launch_bounds__(BLOCK) trace_queue_kernel(SceneView S, TraceQueue Q, unsigned long long* step_counters) {
  extern __shared__ uint32_t s_stack[];  // [depth][BLOCK]
  uint32_t* stack = s_stack + threadIdx.x;
  const float tmin = 0.001f;
  bool q_active = false, q_any = false, q_emit = false, exhausted = false;
  f3 qo = splat(0.0f), qd = splat(0.0f), qinv = splat(0.0f);
  float q_tmax = 0.0f;
  uint32_t cur = 0, mine = 0xffffffffu;
  int sp = 0;
  HitRec best;
  best.t = best.u = best.v = 0.0f;
  best.slot = 0xffffffffu;
  uint32_t st_node = 0, st_leaf = 0, st_iter = 0, ln_node = 0, ln_leaf = 0;
  // A wave owns a CHUNK of the queue at a time and walks it with a cursor of its own: ONE global atomic per RENE_GATE_CHUNK rays.  (The first version
  // took every refill's rays from the global counter -- an atomic per refill per wave, a million of them on one address for ten million rays -- and ran at
  // 1.6 - 4.6 Grays/s, FASTER the less it refilled: atomics on one address retire at ~ 80 M/s on this chip, 12 ns each, whoever issues them.  The
  // stage-separated wavefront of round 2 -- wavefront.inc, 1.2 Grays/s, "an empty pass 97 us" -- handed out its ray ids the same way.)
  uint32_t chunk_next = 0, chunk_end = 0;  // wave-uniform
  for (;;) {
    // ---- refill: the free lanes take the next rays of the wave's chunk, in lane order (a wave's rays stay neighbours in the queue) ----
    const unsigned long long free_mask = __ballot(!q_active);
    const uint32_t n_free = (uint32_t)__popcll(free_mask);
    if (!exhausted && (n_free >= Q.refill_min || n_free == 64u)) {
      if (chunk_next >= chunk_end) {
        uint32_t base = 0;
        if (lane_id() == 0) base = atomicAdd(Q.counter, (uint32_t)RENE_GATE_CHUNK);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t total = Q.n * Q.passes;  // (the queue is walked `passes` times over: a launch long enough for every wave to take many chunks)
        chunk_next = base < total ? base : total;
        chunk_end = base + RENE_GATE_CHUNK < total ? base + RENE_GATE_CHUNK : total;
        if (chunk_next >= chunk_end) exhausted = true;
      }
      const uint32_t vidx = chunk_next + (uint32_t)__popcll(free_mask & ((1ull << lane_id()) - 1ull));
      const uint32_t idx = vidx % Q.n;
      const uint32_t chunk_limit = chunk_end;
      chunk_next = chunk_next + n_free < chunk_end ? chunk_next + n_free : chunk_end;
      if (!q_active && vidx < chunk_limit) {
        const float4 a = ldg4(Q.o_tmax + 4 * (size_t)idx);
        uint32_t flags;
        if (FP16) {
          const uint32_t w0 = Q.d_flags[2 * (size_t)idx], w1 = Q.d_flags[2 * (size_t)idx + 1];
          float dx, dy, dz, unused;
          unpack_h2(__uint_as_float(w0), dx, dy);
          unpack_h2(__uint_as_float(w1), dz, unused);
          qd = mk3(dx, dy, dz);
          flags = w1 >> 16;
        } else {
          const float4 b = ldg4(reinterpret_cast<const float*>(Q.d_flags) + 4 * (size_t)idx);
          qd = mk3(b.x, b.y, b.z);
          flags = __float_as_uint(b.w);
        }
        qo = mk3(a.x, a.y, a.z);
        q_tmax = a.w;
        qinv = safe_inv(qd);
        q_any = (flags & 1u) != 0u;
        q_emit = (flags & 2u) != 0u;
        cur = 0;
        sp = 0;
        best.slot = 0xffffffffu;
        best.t = -1.0f;
        mine = idx;
        q_active = true;
      }
    }
    if (!__any(q_active)) {
      if (exhausted) break;
      continue;
    }
    st_iter++;
    // ---- inner-node steps ----
#pragma unroll
    for (int rep = 0; rep < RENE_GATE_NODE_STEPS; ++rep) {
      const bool at_inner = q_active && !(cur & LEAF_BIT);
      if (__any(at_inner)) st_node++;
      if (at_inner) {
        ln_node++;
        float t4[4];
        uint32_t w4[4];
        node4_test((q_emit ? S.emit.nodes : S.main.nodes)[cur].q, qo, qinv, tmin, q_tmax, t4, w4);
        if (!node4_descend(t4, w4, stack, sp, cur)) {
          if (sp == 0) {
            q_active = false;
          } else {
            sp--;
            cur = stack[sp * BLOCK];
          }
        }
      }
    }
    // ---- leaf step, postponed until enough lanes have one ----
    {
      const bool at_leaf = q_active && (cur & LEAF_BIT) != 0;
      const uint32_t n_leaf = (uint32_t)__popcll(__ballot(at_leaf));
      const uint32_t n_inner = (uint32_t)__popcll(__ballot(q_active && !(cur & LEAF_BIT)));
      if (n_leaf != 0 && (n_leaf >= Q.leaf_min || n_inner == 0)) {
        st_leaf++;
        if (at_leaf) {
          ln_leaf++;
          const PrimIsect* isect = q_emit ? S.emit.isect : S.main.isect;
          const uint32_t first = cur & LEAF_FIRST_MASK, count = ((cur >> LEAF_COUNT_SHIFT) & 15u) + 1u;
          if (cur & SPHERE_BIT) {
            intersect_sphere(isect, S.spheres, first, qo, qd, tmin, best, q_tmax);
          } else {
            uint32_t n_tests = 0;
            intersect_leaf(isect, first, count, qo, qd, tmin, best, q_tmax, n_tests);
          }
          if ((q_any && best.slot != 0xffffffffu) || sp == 0) {
            q_active = false;
          } else {
            sp--;
            cur = stack[sp * BLOCK];
          }
        }
      }
    }
    // ---- a completed query: its hit, 16 bytes ----
    if (!q_active && mine != 0xffffffffu) {
      float4* out = reinterpret_cast<float4*>(Q.hits) + mine;
      *out = make_float4(best.slot == 0xffffffffu ? -1.0f : best.t, best.u, best.v, __uint_as_float(best.slot));
      mine = 0xffffffffu;
    }
  }
  if (step_counters) {  // how full the steps were: wave-steps and lane-steps per kind
    unsigned long long a = wave_sum(ln_node), b = wave_sum(ln_leaf);
    if (lane_id() == 0) {
      atomicAdd(&step_counters[0], (unsigned long long)st_node);
      atomicAdd(&step_counters[1], a);
      atomicAdd(&step_counters[2], (unsigned long long)st_leaf);
      atomicAdd(&step_counters[3], b);
      atomicAdd(&step_counters[4], (unsigned long long)st_iter);
    }
  }
}

hipError_t launch_trace_queue(const LaunchConfig& cfg, const SceneView& S, const TraceQueue& Q, uint32_t blocks_per_cu, unsigned long long* step_counters, hipStream_t st) {
  const size_t lds = (size_t)Q.stack_entries * BLOCK * sizeof(uint32_t);
  auto kernel = Q.fp16 ? trace_queue_kernel<true> : trace_queue_kernel<false>;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, BLOCK, lds) != hipSuccess || occ < 1) occ = 1;
  if (blocks_per_cu && (int)blocks_per_cu < occ) occ = (int)blocks_per_cu;
  const uint32_t need = (Q.n + BLOCK - 1) / BLOCK;
  dim3 grid(std::max(1u, std::min(need, std::max(1u, cfg.cus) * (uint32_t)occ))), block(BLOCK);
  hipLaunchKernelGGL(kernel, grid, block, lds, st, S, Q, step_counters);
  return hipGetLastError();
}

}  // namespace rene
