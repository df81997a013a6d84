// gather_bench.hip -- what does a divergent 64-byte record fetch cost on gfx950?  (developer microbenchmark behind the
// node-fetch design of the BVH kernels; tools/, not part of the library)
//   mode 0: every lane reads its own 64-byte record with four global_load_dwordx4 (64 different lines per instruction)
//   mode 1: the four lanes of a quad read the four quarters of ONE record per instruction (16 lines per instruction,
//           each line read whole by adjacent lanes); four instructions cover the quad's four records
//   mode 2: like 0, but only `active` of 64 lanes take part (traversal kernels run at ~40 % lane utilisation)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) gather(const v4f* table, uint32_t n_rec, int iters, int active, float* out) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t idx = hash(blockIdx.x * 256u + threadIdx.x) % n_rec;
  float acc = 0.0f;
  if (MODE == 2 && (int)lane >= active) { out[blockIdx.x * 256 + threadIdx.x] = 0.0f; return; }
  for (int it = 0; it < iters; ++it) {
    v4f a, b, c, d;
    if (MODE == 1) {
      const uint32_t q = lane & 3u, base = lane & ~3u;
      const uint32_t n0 = __shfl(idx, base + 0), n1 = __shfl(idx, base + 1), n2 = __shfl(idx, base + 2), n3 = __shfl(idx, base + 3);
      a = table[(size_t)n0 * 4 + q];
      b = table[(size_t)n1 * 4 + q];
      c = table[(size_t)n2 * 4 + q];
      d = table[(size_t)n3 * 4 + q];
    } else {
      const v4f* p = table + (size_t)idx * 4;
      a = p[0]; b = p[1]; c = p[2]; d = p[3];
    }
    const float s = a.x + b.y + c.z + d.w;
    acc += s;
    idx = hash(idx + __float_as_uint(s)) % n_rec;  // the next record depends on the data, like a traversal
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int iters = 256;
  float* out;
  hipMalloc(&out, 256 * 2048 * 4 * sizeof(float));
  for (size_t mb : {2ul, 16ul, 200ul}) {
    const uint32_t n_rec = (uint32_t)(mb * 1024 * 1024 / 64);
    std::vector<float> h((size_t)n_rec * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 1e-3f;
    v4f* table;
    hipMalloc(&table, h.size() * 4);
    hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int waves = 4; waves <= 8; waves += 4) {
      const int blocks = 256 * waves;  // `waves` per SIMD
      auto run = [&](int mode, int active) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, active, out);
          if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, active, out);
          if (mode == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, active, out);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double recs = (double)blocks * 256 * (mode == 2 ? active / 64.0 : 1.0) * iters;
        std::printf("table %4zu MB, %d waves/SIMD, mode %d active %2d: %7.3f ms, %6.1f G records/s, %5.2f TB/s, %.1f cycles per wave-step at 2.4 GHz\n", mb, waves, mode, active, ms,
                    recs / ms / 1e6, recs * 64 / ms / 1e9, ms * 1e-3 * 2.4e9 / iters);
      };
      run(0, 64);
      run(1, 64);
      run(2, 24);
    }
    hipFree(table);
  }
  return 0;
}
