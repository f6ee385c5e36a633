// Micro-benchmark: what one `global_load_lds_dwordx4` (64 lanes x 16 B = 1 KiB into LDS) costs the wave that issues it, and the rate a
// CU reaches, for the access pattern of the GEMM kernels here: lines STREAMED (touched once per sharer, never again) while S workgroups
// of one XCD ask for the same lines at about the same time (qkv_attn_fused.hip: 8 workgroups share a pair group's activation rows, 4 a
// head's weight rows; ffn_fused.hip: 32 share the weight rows).  One workgroup per CU, 256 workgroups; workgroup b lives on XCD b % 8
// (round-robin dispatch), so the S sharers of a region are blocks of equal b % 8 and consecutive b / 8.
//   kind 0: LDS-DMA            kind 1: global_load_dwordx4 into registers (no LDS write)     kind 2: LDS-DMA with 40 MFMAs per wave and round
//   `issue` = cycles between the first and after the last load instruction of a round, per instruction: the time the wave could not
//   have issued anything else.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/ldsdma_share.hip -o tools/micro/ldsdma_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PER, int KIND>
__global__ __launch_bounds__(1024) void k(const char* __restrict__ buf, size_t region_bytes, int iters, int sharers, int dma_waves, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[144 * 1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w >= dma_waves) return;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const char* base = buf + ((size_t)xcd + 8 * (size_t)(slot / sharers)) * region_bytes;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(lane * 0.01f + i); fb[i] = (_Float16)(1.f - i * 0.1f); }
  u32x4 x = {0, 0, 0, 0};
  long long issue_cycles = 0;
  const int stage = dma_waves * PER;              // 1 KiB pieces per round of the workgroup
  int ring = 0;
  for (int it = 0; it < iters; ++it) {
    ring = ring == 2 ? 0 : ring + 1;
    u32x4 v[PER];
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const size_t piece = (size_t)it * stage + (size_t)w * PER + j;
      const size_t off = ((piece * 1024) & (region_bytes - 1)) + (size_t)lane * 16;       // region_bytes is a power of two
      if (KIND == 1) {
        v[j] = __builtin_nontemporal_load((const u32x4*)(base + off));
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                         (__attribute__((address_space(3))) void*)(smem + ((w * PER + j) & 47) * 1024 + ring * 48 * 1024), 16, 0, 0);
      }
    }
    issue_cycles += __builtin_readcyclecounter() - t0;
    if (KIND == 2) {
#pragma unroll
      for (int q = 0; q < 40; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc, 0, 0, 0);
    }
    if (KIND == 1) {
#pragma unroll
      for (int j = 0; j < PER; ++j) x ^= v[j];                           // the compiler waits here
    } else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 123.456f || x[0] == 0x12345u) sink[threadIdx.x] = acc[1] + x[1];
  if (lane == 0 && blockIdx.x == 17 && w == 0) sink[1024] = (float)issue_cycles / (float)(iters * PER);
}

template <int PER, int KIND>
void run(const char* buf, float* sink, size_t total, int sharers, int dma_waves, bool resident, const char* what) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int g = 256;
  const int iters = 16 * 1024 / (dma_waves * PER);                      // about 16 MiB per workgroup
  const size_t regions = (size_t)8 * (32 / sharers);
  const size_t region_bytes = resident ? ((size_t)256 << 10) : total / regions;   // resident: 256 KiB per region, 8 regions: all hits after the first pass
  k<PER, KIND><<<g, 64 * dma_waves>>>(buf, region_bytes, 20, sharers, dma_waves, sink);
  CK(hipEventRecord(a));
  k<PER, KIND><<<g, 64 * dma_waves>>>(buf, region_bytes, iters, sharers, dma_waves, sink);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms, ic; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipMemcpy(&ic, sink + 1024, 4, hipMemcpyDeviceToHost));
  const double bytes = (double)g * iters * dma_waves * PER * 1024;
  printf("%-34s %2d waves x %2d per round, %2d sharers%s: issue %4.0f cycles/instr  %6.2f TB/s  %5.1f GB/s per CU\n", what, dma_waves, PER, sharers,
         resident ? " (L2-resident)" : "", ic, bytes / ms / 1e9, bytes / g / ms / 1e6);
}

int main() {
  char* buf;
  float* sink;
  const size_t total = (size_t)4 << 30;        // 4 GiB: far beyond the 256 MiB Infinity Cache
  CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total));
  CK(hipMalloc(&sink, 8192));
  for (int sharers : {1, 2, 8, 32}) run<5, 0>(buf, sink, total, sharers, 8, false, "LDS-DMA");
  for (int waves : {1, 2, 4, 8, 16}) run<5, 0>(buf, sink, total, 8, waves, false, "LDS-DMA");
  for (int waves : {1, 4, 8, 16}) run<5, 0>(buf, sink, total, 32, waves, true, "LDS-DMA");
  run<2, 0>(buf, sink, total, 8, 8, false, "LDS-DMA");
  run<10, 0>(buf, sink, total, 8, 8, false, "LDS-DMA");
  run<16, 0>(buf, sink, total, 8, 8, false, "LDS-DMA");
  for (int waves : {1, 4, 8, 16}) run<5, 1>(buf, sink, total, 8, waves, false, "global_load_dwordx4 to registers");
  run<10, 1>(buf, sink, total, 8, 8, false, "global_load_dwordx4 to registers");
  run<5, 1>(buf, sink, total, 32, 8, true, "global_load_dwordx4 to registers");
  for (int waves : {4, 8, 16}) run<5, 2>(buf, sink, total, 8, waves, false, "LDS-DMA + 40 MFMA per round");
  return 0;
}
