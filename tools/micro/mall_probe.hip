// Probe (gfx950): is a buffer that one kernel has just WRITTEN served from the Infinity Cache when the next kernel reads it,
// and do repeated overwrites of a small buffer stay out of HBM?  For footprints around the 256 MB cache: time of
// (a) a store-only kernel over the buffer, (b) a load-only kernel right behind it, (c) a load-only kernel after a 2 GB
// flush of another buffer.  Build: hipcc -O3 --offload-arch=gfx950 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wr(f4* dst, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = f4{v, v, v, v};
}
__global__ __launch_bounds__(256) void rd(const f4* src, size_t n, float* sink) {
  f4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) sink[0] = acc[0];
}
int main() {
  const size_t big = (size_t)2048 << 20;
  f4 *buf, *flush; float* sink;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&flush, big)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, big)); CK(hipMemset(flush, 0, big));
  hipEvent_t ev[4];
  for (auto& x : ev) CK(hipEventCreate(&x));
  for (size_t mb : {32, 64, 100, 150, 200, 300, 600, 2048}) {
    const size_t n = (mb << 20) / 16;
    float tw = 0, tr = 0, trc = 0;
    const int reps = 8;
    for (int r = 0; r < reps + 1; ++r) {
      CK(hipEventRecord(ev[0]));
      wr<<<8192, 256>>>(buf, n, (float)r);
      CK(hipEventRecord(ev[1]));
      rd<<<8192, 256>>>(buf, n, sink);
      CK(hipEventRecord(ev[2]));
      CK(hipEventSynchronize(ev[2]));
      float a, b; CK(hipEventElapsedTime(&a, ev[0], ev[1])); CK(hipEventElapsedTime(&b, ev[1], ev[2]));
      if (r) { tw += a; tr += b; }
      wr<<<8192, 256>>>(flush, big / 16, 1.f);         // push everything out of the cache
      CK(hipEventRecord(ev[0]));
      rd<<<8192, 256>>>(buf, n, sink);
      CK(hipEventRecord(ev[1])); CK(hipEventSynchronize(ev[1]));
      CK(hipEventElapsedTime(&a, ev[0], ev[1]));
      if (r) trc += a;
    }
    const double gb = (double)(mb << 20) / 1e9;
    printf("%5zu MB: write %.2f TB/s | read right behind the write %.2f TB/s | read after a 2 GB flush %.2f TB/s\n", mb,
           gb * reps / tw, gb * reps / tr, gb * reps / trc);
  }
  return 0;
}
