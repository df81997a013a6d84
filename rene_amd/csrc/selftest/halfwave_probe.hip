// Does a wave64 VALU instruction whose EXEC mask covers only one half of the wave (lanes 0-31) issue in half the time on a SIMD-32?
// Decides whether packing a step's active lanes into one half of a wave could pay.  hipcc --offload-arch=gfx950 -O2 -o halfwave_probe halfwave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(float* out, int mode, int iters, unsigned long long* clk) {
  // the engine clock during the run: shader-clock ticks (s_memtime) per tick of the constant 100 MHz clock (s_memrealtime), wave 0 of workgroup 0
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const bool on = mode == 0 ? true : mode == 1 ? lane < 32 : mode == 2 ? (lane & 1) == 0 : lane < 16;
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  if (on) {
    for (int i = 0; i < iters; i += 8) {  // 64 FMAs per trip: the loop's three scalar instructions are 5 % of the stream
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f); a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
        a4 = fmaf(a4, 1.0001f, 0.5f); a5 = fmaf(a5, 1.0001f, 0.5f); a6 = fmaf(a6, 1.0001f, 0.5f); a7 = fmaf(a7, 1.0001f, 0.5f);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 256 * 8 * 4 * sizeof(float));
  unsigned long long* clk; (void)hipMalloc(&clk, 16); unsigned long long h[2];
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[] = {"all 64 lanes", "lanes 0-31", "even lanes", "lanes 0-15"};
  for (int waves : {1, 2, 3, 4, 6, 8})
    for (int mode = 0; mode < 4; ++mode) {
      const int blocks = 256 * waves;  // `waves` workgroups of four waves per CU = `waves` waves per SIMD
      k<<<blocks, 256>>>(d, mode, 1000, clk);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      k<<<blocks, 256>>>(d, mode, 200000, clk);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double mhz = 100.0 * (double)h[0] / (double)h[1];
      std::printf("%d wave(s) per SIMD, %-13s: %8.3f ms, engine clock %4.0f MHz: %.2f cycles per wave-instruction\n", waves, names[mode], ms, mhz, ms * 1e-3 * mhz * 1e6 / (200000.0 * 8 * waves));
    }
  return 0;
}
