// gather_bench2.hip -- second microbenchmark of the CU's vector-memory path on gfx950: what one divergent load instruction
// costs as a function of (a) its width, (b) how many lanes take part and (c) WHICH lanes (a contiguous run, or spread over
// the wave).  Dependent chains as in gather_bench.hip; 16 waves per CU; cost = CU cycles per load instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// W dwords per instruction, L instructions per step (consecutive pieces of one 64-byte record; L * W <= 16)
template <int W, int L, bool DEP, bool QUAD>
__global__ void __launch_bounds__(256) gather(const float* table, uint32_t n_rec, int iters, unsigned long long mask, float* out) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t idx = hash(blockIdx.x * 256u + threadIdx.x) % n_rec;
  float acc = 0.0f;
  if (!((mask >> lane) & 1ull)) { out[blockIdx.x * 256 + threadIdx.x] = 0.0f; return; }
  for (int it = 0; it < iters; ++it) {
    // QUAD: the four lanes of a quad read the four quarters of the record of the quad's first lane (one 64-byte segment per quad)
    const uint32_t rec = QUAD ? (uint32_t)__shfl((int)idx, (int)(lane & ~3u)) : idx;
    const float* p = table + (size_t)rec * 16 + (QUAD ? 4 * (lane & 3u) : 0);
    float s = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      if (W == 4) { v4f a = *reinterpret_cast<const v4f*>(p + 4 * l); s += a.x + a.w; }
      if (W == 2) { v2f a = *reinterpret_cast<const v2f*>(p + 2 * l); s += a.x + a.y; }
      if (W == 1) { s += p[l]; }
    }
    acc += s;
    idx = hash(idx + (DEP ? __float_as_uint(s) : (uint32_t)it)) % n_rec;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static unsigned long long contiguous(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }
static unsigned long long spread(int n) {  // n lanes, evenly spaced
  unsigned long long m = 0;
  for (int i = 0; i < n; ++i) m |= 1ull << ((i * 64) / n);
  return m;
}
static unsigned long long quads(int n) {  // n lanes as n/4 whole quads, evenly spaced
  unsigned long long m = 0;
  for (int i = 0; i < n / 4; ++i) m |= 0xfull << (4 * ((i * 16) / (n / 4)));
  return m;
}

int main() {
  const int iters = 256, blocks = 256 * 4;
  float* out;
  hipMalloc(&out, 256 * 2048 * 4 * sizeof(float));
  for (size_t kb : {8ul, 2048ul, 65536ul}) {
    const uint32_t n_rec = (uint32_t)(kb * 1024 / 64);
    std::vector<float> h((size_t)n_rec * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 1e-3f;
    float* table;
    hipMalloc(&table, h.size() * 4);
    hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    auto run = [&](int W, int L, const char* pat, int n, unsigned long long mask, bool dep = true, bool quad = false) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
#define GO(w, l, d, q) if (W == w && L == l && dep == d && quad == q) hipLaunchKernelGGL((gather<w, l, d, q>), dim3(blocks), dim3(256), 0, 0, table, n_rec, iters, mask, out)
        GO(4, 4, true, false); GO(4, 1, true, false); GO(2, 1, true, false); GO(1, 1, true, false); GO(4, 2, true, false); GO(1, 4, true, false);
        GO(4, 4, false, false); GO(4, 1, false, false); GO(1, 1, false, false); GO(4, 1, false, true); GO(4, 3, false, false); GO(4, 2, false, false);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double cyc = ms * 1e-3 * 2.4e9 / iters / 16.0;  // CU cycles per step (16 waves share the CU)
      std::printf("table %5zu KB  %s%s x%d * %d  %-10s %2d lanes: %7.3f ms  %6.1f cycles per step per CU, %5.1f per instruction\n", kb, dep ? "chain" : "indep", quad ? " quad" : "     ", W, L, pat, n, ms, cyc, cyc / L);
    };
    for (int n : {4, 8, 16, 32, 64}) {
      run(4, 4, "contiguous", n, contiguous(n));
      if (n < 64) run(4, 4, "spread", n, spread(n));
      if (n < 64) run(4, 4, "quads", n, quads(n));
    }
    for (int n : {16, 64}) {
      run(4, 1, "contiguous", n, contiguous(n));
      run(4, 2, "contiguous", n, contiguous(n));
      run(2, 1, "contiguous", n, contiguous(n));
      run(1, 1, "contiguous", n, contiguous(n));
      run(1, 4, "contiguous", n, contiguous(n));
      if (n < 64) run(4, 1, "spread", n, spread(n));
      if (n < 64) run(1, 1, "spread", n, spread(n));
    }
    for (int n : {8, 16, 24, 32, 64}) {
      run(4, 4, "contiguous", n, contiguous(n), false);
      if (n < 64) run(4, 4, "spread", n, spread(n), false);
      run(4, 1, "contiguous", n, contiguous(n), false);
      run(1, 1, "contiguous", n, contiguous(n), false);
      run(4, 1, "contiguous", n, contiguous(n), false, true);
    }
    run(4, 3, "contiguous", 64, contiguous(64), false);
    run(4, 2, "contiguous", 64, contiguous(64), false);
    run(4, 3, "contiguous", 24, contiguous(24), false);
    run(4, 2, "contiguous", 24, contiguous(24), false);
    hipFree(table);
  }
  return 0;
}
