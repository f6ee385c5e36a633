// Probe (gfx950): does MODE.FP16_OVFL (hardware register MODE, bit 23) make the f32 -> e4m3 conversions (v_cvt_scalef32_pk_fp8_f32,
// v_cvt_pk_fp8_f32) and the f32 -> f16 conversion saturate instead of producing NaN / Inf?  (the mixed-row producers clamp
// every value to +-448 with v_med3_f32 today: 8 of ~22 conversion instructions per 4 values)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
typedef short v2s __attribute__((ext_vector_type(2)));
static float e4m3(uint8_t v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; float x = (e == 15 && m == 7) ? NAN : e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7); return s ? -x : x; }
__global__ void k(const float* x, uint32_t* out, int n, int ovfl) {
  int i = threadIdx.x;
  if (ovfl) __builtin_amdgcn_s_setreg((1 /*HW_REG_MODE*/) | (23 << 6) | ((1 - 1) << 11), 1);   // MODE[23] = FP16_OVFL
  if (i >= n) return;
  v2s o = {0, 0};
  o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, x[i], 0.f, 1.f, false);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false);
  _Float16 h = (_Float16)x[i];
  out[i] = (uint16_t)o[0] & 0xff;
  out[32 + i] = r & 0xff;
  out[64 + i] = __builtin_bit_cast(uint16_t, h);
}
int main() {
  const float xs[12] = {1.f, 447.f, 448.f, 449.f, 480.f, 1000.f, 1e6f, -500.f, INFINITY, NAN, 70000.f, -1e9f};
  float* dx; uint32_t* dout;
  hipMalloc(&dx, 48); hipMalloc(&dout, 96 * 4);
  hipMemcpy(dx, xs, 48, hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    k<<<1, 64>>>(dx, dout, 12, ovfl);
    uint32_t out[96]; hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    printf("FP16_OVFL = %d\n", ovfl);
    for (int i = 0; i < 12; ++i) {
      _Float16 h; uint16_t hb = (uint16_t)out[64 + i]; __builtin_memcpy(&h, &hb, 2);
      printf("  x %-10g cvt_scalef32_pk_fp8 0x%02x = %-6g  cvt_pk_fp8 0x%02x = %-6g  f16 0x%04x = %g\n", xs[i], out[i], e4m3(out[i]), out[32 + i], e4m3(out[32 + i]), hb, (double)(float)h);
    }
  }
  return 0;
}
