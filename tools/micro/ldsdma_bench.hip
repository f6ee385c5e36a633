// Micro-benchmark: what rate does the L2 -> LDS path (global_load_lds_dwordx4) sustain per CU, by
// access shape?  One 256-thread..768-thread workgroup per CU streams a small (L2-resident) buffer.
//   shape 0: 16 rows x 64 B per wave-instruction, rows 1152 B apart (the GEMM's current chunk)
//   shape 1:  8 rows x 128 B, rows 1152 B apart (full cache lines)
//   shape 2:  1 KiB contiguous per wave-instruction (blocked operand layout)
//   shape 3:  global_load_dwordx4 to VGPRs, 1 KiB contiguous (no LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int SHAPE>
__global__ __launch_bounds__(768) void k(const char* __restrict__ buf, size_t bytes, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[96 * 1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // each workgroup walks its own window so that 32 CUs of an XCD share lines like GEMM tiles do
  size_t base = ((size_t)(blockIdx.x >> 3) * 131072) % (bytes / 2);
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      size_t off;
      const size_t chunk = (size_t)((it * 8 + j) * nw + w);
      if (SHAPE == 0) off = (chunk % 18) * 64 + (chunk / 18) * 16 * 1152 + (size_t)(lane >> 2) * 1152 + (lane & 3) * 16;
      else if (SHAPE == 1) off = (chunk % 9) * 128 + (chunk / 9) * 8 * 1152 + (size_t)(lane >> 3) * 1152 + (lane & 7) * 16;
      else off = chunk * 1024 + lane * 16;
      off = (base + off) % (bytes - 4096);
      off &= ~(size_t)15;
      if (SHAPE == 3) {
        const float4 v = *(const float4*)(buf + off);
        acc += v.x + v.y + v.z + v.w;
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + off),
                                         (__attribute__((address_space(3))) void*)(smem + ((w * 8 + j) % 96) * 1024), 16, 0, 0);
      }
    }
    if (SHAPE != 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 12345.f) sink[0] = acc + smem[threadIdx.x];
}

template <int SHAPE>
void run(const char* buf, size_t bytes, int waves, float* sink) {
  const int iters = 400;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<SHAPE><<<256, waves * 64>>>(buf, bytes, 10, sink);
  CK(hipEventRecord(a));
  k<SHAPE><<<256, waves * 64>>>(buf, bytes, iters, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double total = 256.0 * waves * iters * 8 * 1024;
  printf("shape %d waves %2d: %.3f ms  %.2f TB/s  %.1f GB/s per CU\n", SHAPE, waves, ms, total / ms / 1e9, total / 256 / ms / 1e6);
}

int main() {
  const size_t bytes = 24u << 20;  // 24 MiB: L2 (8 x 4 MiB) + MALL resident
  char* buf; float* sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes)); CK(hipMalloc(&sink, 4));
  for (int waves : {4, 8, 12}) {
    run<0>(buf, bytes, waves, sink); run<1>(buf, bytes, waves, sink); run<2>(buf, bytes, waves, sink); run<3>(buf, bytes, waves, sink);
  }
  return 0;
}
