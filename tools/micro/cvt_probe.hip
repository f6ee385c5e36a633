// Probe (gfx950): v_cvt_scalef32_pk_fp8_f32 / _f16 -- direction of the scale, saturation -- and packed f16 conversion.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
typedef short v2s __attribute__((ext_vector_type(2)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
static float e4m3(uint8_t v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; float x = (e == 15 && m == 7) ? NAN : e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7); return s ? -x : x; }
__global__ void k(const float* x, const float* sc, uint32_t* out, int n) {
  int i = threadIdx.x;
  if (i >= n) return;
  v2s o = {0, 0};
  o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, x[i], 0.f, sc[i], false);
  out[i] = (uint16_t)o[0];
  v2h h = {(_Float16)x[i], (_Float16)0};
  v2s o2 = {0, 0};
  o2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o2, h, sc[i], false);
  out[32 + i] = (uint16_t)o2[0];
}
int main() {
  const float xs[10] = {1.f, 3.f, 100.f, 480.f, 1000.f, 1e6f, -500.f, 0.01f, 7.3f, 28.1f};
  const float ss[10] = {1.f, 0.5f, 0.0625f, 1.f, 1.f, 1.f, 2.f, 0.0625f, 0.0625f, 0.0625f};
  float *dx, *ds; uint32_t* dout;
  hipMalloc(&dx, 40); hipMalloc(&ds, 40); hipMalloc(&dout, 256);
  hipMemcpy(dx, xs, 40, hipMemcpyHostToDevice); hipMemcpy(ds, ss, 40, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, ds, dout, 10);
  uint32_t out[64]; hipMemcpy(out, dout, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 10; ++i) printf("x %g scale %g: from f32 0x%02x = %g | from f16 0x%02x = %g\n", xs[i], ss[i], out[i] & 0xff, e4m3(out[i] & 0xff), out[32 + i] & 0xff, e4m3(out[32 + i] & 0xff));
  return 0;
}
