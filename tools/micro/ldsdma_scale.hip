// Micro-benchmark: is the ~73 GB/s per CU of the L2 -> LDS path (global_load_lds_dwordx4, 8 rows x 128 B per wave-instruction) a
// per-CU limit or the CU's share of a chip-wide one?  One 512-thread workgroup per CU (LDS-limited), G workgroups, every workgroup
// of an XCD streams the same 1 MiB window (L2 hits, the way 32 panel workgroups stream one weight matrix).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/ldsdma_scale.hip -o tools/micro/ldsdma_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(512) void k(const char* __restrict__ buf, int iters, int same) {
  __shared__ __attribute__((aligned(16))) char smem[150 * 1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t window = 1u << 20;
  const char* base = buf + (same ? (size_t)(blockIdx.x & 7) : (size_t)blockIdx.x) * window;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {      // 40 instructions = 40 KiB per "stage" and workgroup
      const size_t piece = (size_t)((it * 5 + j) * 8 + w);
      const size_t off = ((piece % 18) * 128 + (piece / 18) * 8 * 2304 + (size_t)(lane >> 3) * 2304 + (lane & 7) * 16) % (window - 4096);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (off & ~(size_t)15)),
                                       (__attribute__((address_space(3))) void*)(smem + ((w * 5 + j) * 1024 + (it % 3) * 40960)), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
  char* buf;
  CK(hipMalloc(&buf, 260u << 20)); CK(hipMemset(buf, 1, 260u << 20));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int same = 1; same >= 0; --same)
    for (int g : {8, 32, 64, 128, 192, 256}) {
      const int iters = 4000;
      k<<<g, 512>>>(buf, 50, same);
      CK(hipEventRecord(a));
      k<<<g, 512>>>(buf, iters, same);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double total = (double)g * iters * 40 * 1024;
      printf("%s window, %3d workgroups: %.3f ms  %.2f TB/s  %.1f GB/s per CU\n", same ? "one window per XCD (L2 hits)" : "own 1 MiB window per workgroup", g, ms,
             total / ms / 1e9, total / g / ms / 1e6);
    }
  return 0;
}
