// memset_probe.hip -- is hipMemset (null stream, device memory) complete when it returns, while a persistent kernel on a
// non-blocking stream holds every CU slot?  (the question behind rene_ctx::zero_now, docs/history.md section 4g)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>

__global__ void __launch_bounds__(256) spin(unsigned long long ticks, const unsigned* watched, unsigned* seen_zero_at) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t;
  bool noted = false;
  while ((t = __builtin_amdgcn_s_memrealtime()) - t0 < ticks) {
    if (!noted && blockIdx.x == 0 && threadIdx.x == 0 && __atomic_load_n(watched, __ATOMIC_RELAXED) == 0u) {
      *seen_zero_at = (unsigned)((t - t0) / 100000ull);  // ms on the 100 MHz clock
      noted = true;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned *d, *seen;
  hipMalloc(&d, 4096);
  hipMalloc(&seen, 4);
  hipMemsetAsync(d, 0xff, 4096, s);
  hipMemsetAsync(seen, 0xff, 4, s);
  hipStreamSynchronize(s);
  for (int blocks_per_cu : {8, 2}) {  // 8 x 256 threads: every wave slot of a CU; 2: a quarter of them
    hipMemsetAsync(d, 0xff, 4096, s);
    hipMemsetAsync(seen, 0xff, 4, s);
    hipStreamSynchronize(s);
    hipLaunchKernelGGL(spin, dim3(prop.multiProcessorCount * blocks_per_cu), dim3(256), 0, s, 200000000ull /* 2 s */, d, seen);
    std::this_thread::sleep_for(std::chrono::milliseconds(50));  // the kernel is resident
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMemset(d, 0, 4096);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    hipStreamSynchronize(s);
    unsigned at = 0;
    hipMemcpy(&at, seen, 4, hipMemcpyDeviceToHost);
    std::printf("%d blocks of 256 per CU spinning for 2 s: hipMemset returned %s after %.3f ms; the kernel saw the zero %s\n", blocks_per_cu,
                hipGetErrorString(e), ms, at == 0xffffffffu ? "never (the fill ran after the kernel had ended)" : (std::to_string(at) + " ms after its start").c_str());
  }
  return 0;
}
