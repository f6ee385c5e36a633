// Probe (gfx950): semantics of the block-scaled K = 128 MFMA with fp6 (e2m3) operands, of the fp6 conversions and of the 12-byte LDS-DMA.
//   part 1: v_mfma_scale_f32_16x16x128_f8f6f4, cbsz = blgp = 2: where value p (0..31) of a lane sits in its 6 registers, the e2m3 code,
//           and WHICH lane's scale byte multiplies it (one-hot operand against all ones, one lane's scale doubled); the same map for the
//           e4m3 form (cbsz = blgp = 0: profiles/r02_mfma_mix_probe.txt found per-k-block scales not where the data map suggests)
//   part 2: v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_fp6_f16: order of the 32 results, rounding, saturation, scale operand;
//           e2m3 code from the e4m3 conversion of x / 64 (the squeeze the mixed-row producers would use), compared with the native conversion
//   part 3: global_load_lds_dwordx3: LDS destination of lane l
// Build: hipcc -O3 --offload-arch=gfx950 -o fp6_probe fp6_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u6 __attribute__((ext_vector_type(6)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h32 __attribute__((ext_vector_type(32)));

static float e2m3_to_float(unsigned c) {
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float x = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -x : x;
}

// case = (fq, p, ls): operand A = zero but for value p of lane (row 0, fq) = 1.0; B = all ones; the scale of lane ls is 2, all others 1.
// FMT 2: fp6, 6-bit fields at bit 6 p of the lane's 192 bits; FMT 0: e4m3, byte p of the lane's 32.  SIDE 0: the one-hot operand is the
// FIRST matrix operand (its row = the result's row), 1: the second (its row = the result's column).
template <int FMT, int SIDE>
__global__ void map_kernel(float* out) {
  const int l = threadIdx.x, c = blockIdx.x;
  const int fq = c >> 11, p = (c >> 6) & 31, ls = c & 63;
  v8i hot, ones;
  for (int j = 0; j < 8; ++j) hot[j] = 0;
  if (FMT == 2) {
    // 32 x code 0x08 (1.0): bit pattern of 001000 repeated
    unsigned bits[6] = {0, 0, 0, 0, 0, 0};
    for (int q = 0; q < 32; ++q) { const int b = 6 * q + 3; bits[b >> 5] |= 1u << (b & 31); }
    for (int j = 0; j < 6; ++j) ones[j] = (int)bits[j];
    ones[6] = ones[7] = 0;
    if (l == 16 * fq) { const int b = 6 * p + 3; hot[b >> 5] = (int)(1u << (b & 31)); }
  } else {
    for (int j = 0; j < 8; ++j) ones[j] = 0x38383838;
    if (l == 16 * fq) hot[p >> 2] = 0x38 << (8 * (p & 3));
  }
  const int s_hot = l == ls ? 128 : 127, s_one = 127;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  if (SIDE == 0) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(hot, ones, acc, FMT, FMT, 0, s_hot, 0, s_one);
  else acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, hot, acc, FMT, FMT, 0, s_one, 0, s_hot);
  // SIDE 0: row 0 of the result = registers 0 of lanes 0..15 (row = 4 (l >> 4) + reg, column = l & 15); SIDE 1: column 0 = lanes 0, 16, 32, 48
  out[c * 64 + l] = acc[0];      // (every lane stores: an MFMA under a divergent branch runs with the other lanes' operands unset)
}

// one code per launch in value 0 of lane 0 against ones: the decoded value of every 6-bit code
__global__ void code_kernel(float* out) {
  const int l = threadIdx.x, c = blockIdx.x;
  v8i hot, ones;
  unsigned bits[6] = {0, 0, 0, 0, 0, 0};
  for (int q = 0; q < 32; ++q) { const int b = 6 * q + 3; bits[b >> 5] |= 1u << (b & 31); }
  for (int j = 0; j < 8; ++j) { hot[j] = 0; ones[j] = j < 6 ? (int)bits[j] : 0; }
  if (l == 0) hot[0] = c;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(hot, ones, acc, 2, 2, 0, 127, 0, 127);
  out[c * 64 + l] = acc[0];
}

__global__ void cvt_kernel(const float* x, float scale, unsigned* out32, unsigned* out16, unsigned* out8) {
  const int l = threadIdx.x;
  f16v a, b;
  h32 h;
  for (int i = 0; i < 16; ++i) { a[i] = x[l * 32 + i]; b[i] = x[l * 32 + 16 + i]; }
  for (int i = 0; i < 32; ++i) h[i] = (_Float16)x[l * 32 + i];
  const u6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
  const u6 rh = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, scale);
  for (int i = 0; i < 6; ++i) { out32[l * 6 + i] = r[i]; out16[l * 6 + i] = rh[i]; }
  // the e4m3 route: e4m3(x / (64 scale)) has exponent field <= 3 for |x / scale| <= 7.5: code = sign << 5 | low five bits
  typedef short v2s __attribute__((ext_vector_type(2)));
  for (int i = 0; i < 32; i += 2) {
    v2s t = {0, 0};
    t = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(t, x[l * 32 + i], x[l * 32 + i + 1], 64.f * scale, false);
    out8[l * 16 + i / 2] = (unsigned)(unsigned short)t[0];
  }
}

__global__ void dma_kernel(const unsigned* src, unsigned* out) {
  __shared__ unsigned sm[512];      // (64 lanes x 16 bytes = 256 dwords are enough for either hypothesis)
  const int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) sm[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)sm;
  // lane l asks for the 12 bytes at src + 16 l (so that source and destination strides differ)
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx3 %1, %2\n\ts_waitcnt vmcnt(0)" ::"s"(lds0), "v"((unsigned)(l * 16)), "s"(src) : "memory");
  __syncthreads();
  for (int i = l; i < 512; i += 64) out[i] = sm[i];
}

int main() {
  float* dout;
  CK(hipMalloc(&dout, 4096 * 4));
  std::vector<float> h(4096);
  // ---- part 1 ----
  CK(hipFree(dout));
  CK(hipMalloc(&dout, 8192 * 64 * 4));
  h.resize(8192 * 64);
  for (int fmt : {2, 0})
    for (int side : {0, 1}) {
      if (fmt == 2 && side == 0) map_kernel<2, 0><<<8192, 64>>>(dout);
      if (fmt == 2 && side == 1) map_kernel<2, 1><<<8192, 64>>>(dout);
      if (fmt == 0 && side == 0) map_kernel<0, 0><<<8192, 64>>>(dout);
      if (fmt == 0 && side == 1) map_kernel<0, 1><<<8192, 64>>>(dout);
      CK(hipMemcpy(h.data(), dout, 8192 * 64 * 4, hipMemcpyDeviceToHost));
      printf("part1 %s, one-hot %s operand (row 0): the lane whose scale multiplies value p of lane group fq (-1: none, -2: several, -3: a product that is neither 1 nor 2)\n",
             fmt == 2 ? "fp6 e2m3" : "fp8 e4m3", side == 0 ? "first" : "second");
      for (int fq = 0; fq < 4; ++fq) {
        printf("  fq %d:", fq);
        for (int p = 0; p < 32; ++p) {
          int who = -1, n = 0;
          bool bad = false;
          for (int ls = 0; ls < 64; ++ls) {
            const float v = h[(size_t)((fq << 11) | (p << 6) | ls) * 64];
            if (v == 2.f) { who = ls; ++n; } else if (v != 1.f) bad = true;
          }
          printf(" %d", bad ? -3 : n == 1 ? who : n == 0 ? -1 : -2);
        }
        printf("\n");
      }
    }
  code_kernel<<<64, 64>>>(dout);
  CK(hipMemcpy(h.data(), dout, 64 * 64 * 4, hipMemcpyDeviceToHost));
  {
    int bad = 0;
    for (int c = 0; c < 64; ++c) if (h[c * 64] != e2m3_to_float(c)) { if (bad < 8) printf("  code 0x%02x decodes to %g, e2m3 says %g\n", c, h[c * 64], e2m3_to_float(c)); ++bad; }
    printf("part1 e2m3 codes (sign << 5 | exponent << 3 | mantissa, bias 1): %d / 64 differ\n", bad);
  }
  // ---- part 2 ----
  {
    std::vector<float> x(64 * 32);
    // lane 0: 0.125 * i (exact codes, i = 0..31 -> 0 .. 3.875); lane 1: negative and large; lane 2: rounding cases; lane 3: a ramp for the order
    const float l2[32] = {0.0624f, 0.0625f, 0.0626f, 0.1874f, 0.1875f, 0.1876f, 0.9374f, 0.9375f, 0.9376f, 1.0624f, 1.0625f, 1.0626f, 1.1875f, 3.875f, 3.876f, 4.25f,
                          6.75f, 7.25f, 7.5f, 7.74f, 7.75f, 7.76f, 8.f, 9.f, 100.f, 1e6f, INFINITY, NAN, -7.75f, -8.f, -100.f, -0.0625f};
    for (int i = 0; i < 32; ++i) {
      x[i] = 0.125f * i;
      x[32 + i] = -0.25f * i;
      x[64 + i] = l2[i];
      x[96 + i] = (i & 1 ? -1.f : 1.f) * (0.125f + 0.125f * (i % 8)) * (float)(1 << (i / 8 % 3));
    }
    for (int i = 128; i < 64 * 32; ++i) x[i] = (float)((rand() % 2001) - 1000) / 128.0f;      // multiples of 1/128 in [-7.8, 7.8]
    float* dx; unsigned *d32, *d16, *d8;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&d32, 64 * 6 * 4)); CK(hipMalloc(&d16, 64 * 6 * 4)); CK(hipMalloc(&d8, 64 * 16 * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    for (float scale : {1.0f, 0.25f}) {
      cvt_kernel<<<1, 64>>>(dx, scale, d32, d16, d8);
      std::vector<unsigned> r32(64 * 6), r16(64 * 6), r8(64 * 16);
      CK(hipMemcpy(r32.data(), d32, r32.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r16.data(), d16, r16.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r8.data(), d8, r8.size() * 4, hipMemcpyDeviceToHost));
      auto field = [](const unsigned* w, int p) { const int b = 6 * p; unsigned long long v = w[b >> 5] | ((unsigned long long)((b >> 5) < 5 ? w[(b >> 5) + 1] : 0u) << 32); return (unsigned)(v >> (b & 31)) & 63u; };
      printf("part2 scale %g: lane 0 (input 0.125 i): 2xpk16_fp6_f32 fields ->", scale);
      for (int p = 0; p < 32; ++p) printf(" %g", e2m3_to_float(field(&r32[0], p)) * scale);
      printf("\n                        pk32_fp6_f16 fields ->");
      for (int p = 0; p < 32; ++p) printf(" %g", e2m3_to_float(field(&r16[0], p)) * scale);
      printf("\n");
      if (scale == 1.0f) {
        printf("part2 lane 3 (ramp), raw dwords f32 form: %08x %08x %08x %08x %08x %08x\n", r32[18], r32[19], r32[20], r32[21], r32[22], r32[23]);
        printf("part2 rounding / saturation (f32 form | f16 form | e4m3 squeeze):\n");
        for (int i = 0; i < 32; ++i) {
          const unsigned c8 = (r8[2 * 16 + i / 2] >> (8 * (i & 1))) & 0xff;
          printf("   %12g -> %7g | %7g | e4m3 0x%02x -> %7g\n", l2[i], e2m3_to_float(field(&r32[12], i < 16 ? 2 * i : 2 * (i - 16) + 1)), e2m3_to_float(field(&r16[12], i)), c8,
                 e2m3_to_float(((c8 >> 2) & 0x20) | (c8 & 0x1f)));
        }
      }
      // agreement of the three routes on the random lanes, assuming field p = input p (checked above by eye)
      int d_16 = 0, d_8 = 0, n = 0, big8 = 0;
      for (int l = 4; l < 64; ++l)
        for (int i = 0; i < 32; ++i) {
          const unsigned a = field(&r32[l * 6], i < 16 ? 2 * i : 2 * (i - 16) + 1), b = field(&r16[l * 6], i);      // (the f32 form interleaves its two operands)
          const unsigned c8 = (r8[l * 16 + i / 2] >> (8 * (i & 1))) & 0xff;
          const unsigned c = ((c8 >> 2) & 0x20) | (c8 & 0x1f);
          ++n;
          if (a != b) ++d_16;
          if (fabsf(x[l * 32 + i] / scale) <= 7.5f) { if (a != c) { if (d_8 < 5) printf("   x %g: native 0x%02x squeeze 0x%02x (e4m3 0x%02x)\n", x[l * 32 + i], a, c, c8); ++d_8; } } else ++big8;
        }
      printf("part2 scale %g: %d values: f16 form differs from f32 form on %d; e4m3 squeeze differs on %d of the %d with |x / scale| <= 7.5\n", scale, n, d_16, d_8, n - big8);
    }
  }
  // ---- part 3 ----
  {
    std::vector<unsigned> src(1024);
    for (int i = 0; i < 1024; ++i) src[i] = i;
    unsigned *ds, *dd;
    CK(hipMalloc(&ds, 4096)); CK(hipMalloc(&dd, 2048));
    CK(hipMemcpy(ds, src.data(), 4096, hipMemcpyHostToDevice));
    dma_kernel<<<1, 64>>>(ds, dd);
    std::vector<unsigned> o(512);
    CK(hipMemcpy(o.data(), dd, 2048, hipMemcpyDeviceToHost));
    int bad12 = 0, bad16 = 0, holes = 0;
    for (int l = 0; l < 64; ++l) {
      for (int j = 0; j < 3; ++j) {
        if (o[3 * l + j] != (unsigned)(4 * l + j)) ++bad12;      // hypothesis A: lane l's 12 bytes at m0 + 12 l
        if (o[4 * l + j] != (unsigned)(4 * l + j)) ++bad16;      // hypothesis B: at m0 + 16 l (a 4-byte hole behind every lane's piece)
      }
      holes += o[4 * l + 3] == 0xdeadbeefu;
    }
    printf("part3 global_load_lds_dwordx3, lane l's 12 bytes: at LDS m0 + 12 l: %d / 192 dwords differ; at m0 + 16 l: %d / 192 differ, %d / 64 of the dwords behind the pieces untouched\n",
           bad12, bad16, holes);
  }
  return 0;
}
