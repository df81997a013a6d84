// record_tear.hip -- checks on the device it runs on the one hardware property the work-item hand-off of the
// render_kernel family relies on (device_code.inc, fb_store / fb_load1 / fb_load3): an aligned 16-byte
// global_store_dwordx4 ... sc0 sc1 of one lane is seen by an aligned 16-byte global_load_dwordx4 ... sc0 sc1 of
// another lane -- on another CU, on another XCD -- either entirely or not at all.
//
// Writers (half of the workgroups) rewrite a set of 16-byte records over and over with {v, v ^ A, v ^ B, v} for a
// running v (the fourth dword is the "version", as in the renderer); readers (the other half, different workgroups,
// so different CUs and -- blocks being dealt round-robin over the XCDs -- different XCDs) load the same records and
// count every one whose four dwords do not belong together.  Prints "records_read torn" and exits 0 iff torn == 0.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) tear_kernel(uint32_t* records, uint32_t n_records, uint32_t rounds, unsigned long long* out) {
  const uint32_t lane = threadIdx.x, pair = blockIdx.x >> 1;
  const bool writer = (blockIdx.x & 1u) == 0u;
  unsigned long long reads = 0, torn = 0;
  for (uint32_t r = 0; r < rounds; ++r) {
    // both blocks of a pair work on the same 256 records, each lane on its own; the writer's v changes every round
    uint32_t* p = records + (size_t)((pair * 256u + lane) % n_records) * 4u;
    if (writer) {
      const uint32_t v = r * 2654435761u + lane * 40503u + 1u;
      u4 rec = {v, v ^ 0x5a5a5a5au, v ^ 0x0f0f0f0fu, v};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(rec) : "memory");
    } else {
      u4 rec;
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(rec) : "v"(p) : "memory");
      reads++;
      const uint32_t v = rec.w;
      if (v != 0u && (rec.x != v || rec.y != (v ^ 0x5a5a5a5au) || rec.z != (v ^ 0x0f0f0f0fu))) torn++;
    }
  }
  if (!writer) {
    atomicAdd(&out[0], reads);
    atomicAdd(&out[1], torn);
  }
}

int main(int argc, char** argv) {
  const uint32_t rounds = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 20000u;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { std::fprintf(stderr, "no device\n"); return 2; }
  const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 2u;  // one writer and one reader block per pair, all co-resident
  const uint32_t n_records = (blocks / 2u) * 256u;
  uint32_t* records = nullptr;
  unsigned long long* out = nullptr;
  if (hipMalloc(&records, (size_t)n_records * 16) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) return 2;
  (void)hipMemset(records, 0, (size_t)n_records * 16);
  (void)hipMemset(out, 0, 16);
  hipLaunchKernelGGL(tear_kernel, dim3(blocks), dim3(256), 0, 0, records, n_records, rounds, out);
  if (hipDeviceSynchronize() != hipSuccess) { std::fprintf(stderr, "kernel failed\n"); return 2; }
  unsigned long long h[2] = {0, 0};
  (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
  std::printf("%llu %llu\n", h[0], h[1]);
  return h[1] == 0 ? 0 : 1;
}
