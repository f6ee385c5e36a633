// Shared device helpers for the VETO relation-head kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// hipGetLastError() is per host thread and shared with every other user of the HIP runtime in the
// process (PyTorch): clear a stale value before the launch so the check after it is about THIS launch.
#define VETO_LAUNCH(...)            \
  do {                              \
    (void)hipGetLastError();        \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

namespace veto {

constexpr int kDim = 576;        // T_INPUT_DIM; forced by proj_d(512)+proj_v(64), model_veto.py:105-113
constexpr int kTokens = 19;      // cls + 16 patches + location + class, model_veto.py:56-63
constexpr int kPatchTokens = 16;
constexpr int kPosDim = 128;     // pos_embed Linear(4,128), roi_relation_predictors.py:4042-4047
constexpr int kWave = 64;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// "Split row" operand format of every GEMM operand (activations and weights): a logical row of K
// fp32 values is stored as 2K bf16, in blocks of 32 k's: [hi(k0..k0+31) | lo(k0..k0+31)], i.e. 64 B of
// hi followed by 64 B of lo.  One k-step (BK = 32) of one row is therefore ONE full 128-byte line,
// and a row is contiguous (2304 B for K = 576).  Measured on MI355X (tools/micro/ldsdma_bench.hip):
// 8 waves/CU pulling L2-resident data through global_load_lds sustain 64 GB/s per CU with 128-byte
// row pieces vs 49 GB/s with the 64-byte pieces of separate hi/lo planes.
__host__ __device__ __forceinline__ int split_index(int k) { return ((k >> 5) << 6) + (k & 31); }  // hi; lo = +32

// x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significand bits in two bf16.
__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}

// ------------------------------------------------------------------------------------------------
// "Mixed row" operand format (precision mode VETO_MIXED): fp16 main product + e4m3 correction terms.
//
// x = h + l with h = fp16(x): 11 significand bits in h, the residual l is 2^-11 |x| or smaller.  For a product
// sum_k a_k w_k the main term sum ah wh runs on v_mfma_f32_16x16x32_f16, and the two first-order corrections
// sum (al wh + ah wl) are 2^-11 of it, so 4 significand bits of each factor are enough for a 2^-16 result: they run as
// ONE v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (128 = 64 k's x the two corrections), which issues at twice
// the bf16 rate.  Matrix-pipe time per 64 k's and output block: 64 cycles against 96 for the 3-term split-bf16 product
// (profiles/r02_mfma_mix_probe.txt); measured logit error 5-7e-5 against 2-3e-5 (tools/precision_study.py).
//
// A logical row of K fp32 values (K % 64 == 0) is 4K bytes, in blocks of 64 k's = 256 B:
//   [ h: 64 x fp16 (128 B) | 16 groups of 8 B, group j = { X(4 j .. 4 j + 3), Y(4 j .. 4 j + 3) }: 64 + 64 x e4m3 (128 B) ]
//   activation rows: X = e4m3(2^11 (a - h)),  Y = e4m3(a)
//   weight rows:     X = e4m3(2^e h),         Y = e4m3(2^(e+11) (w - h)),   e per tensor (kept on the device)
// so that sum X_a X_w + Y_a Y_w = 2^(11+e) (al wh + ah wl); the MFMA's E8M0 scale operand undoes the 2^(11+e).
// The activation scales are fixed powers of two chosen for RANGE: e4m3 keeps its four significant bits down to 2^-6 and decays
// gracefully below (subnormals: an absolute error of 2^-10, i.e. 2^-21 |w| in a correction term), so nothing is gained by
// scaling activations up, while the clamp at 448 sits at |a| = 448 for both planes (round 2 used 2^15 / 2^4, i.e. |a| <= 28:
// trained LayerNorm gains and FeedForward outliers exceed that; emulated GEMM error on N(0,1) rows with 2 % of the entries
// x 30: 1.8e-4 relative to sum |a||w| then, 1.4e-5 now, 3.0e-6 on plain N(0,1) rows with either choice).  Beyond |a| = 448 an
// element degrades to the fp16 class (2^-11); fp16 itself overflows at 65504.
// (The K = 128 MFMA sums the products of corresponding bytes of the two operands' 128-byte stage rows, so ANY byte order that
// activation and weight rows share is right; this one lets a producer that owns 4 consecutive columns write its X and Y
// bytes as ONE 8-byte store and one that owns 8 columns as one 16-byte store.)
// e4m3 conversions do NOT saturate on gfx950 by default (480 -> NaN, profiles/r02_mfma_mix_probe.txt); they do with
// MODE.FP16_OVFL = 1 (saturating_conversions_on() below), which every producer of mixed rows sets.
// One GEMM stage (128 B of a row) is either the h part or the X|Y part of a block: same addressing as the split rows.
enum OperandFmt { FMT_SPLIT = 0, FMT_MIXED = 1 };
#ifndef VETO_MIX_ACT_HI_EXP
#define VETO_MIX_ACT_HI_EXP 0         // (-DVETO_MIX_ACT_HI_EXP=4 rebuilds round 2's scales for the A/B in tests/test_gpu_parity.py's docstring)
#endif
constexpr int kMixActHiExp = VETO_MIX_ACT_HI_EXP;   // activation value scale 2^0, residual scale 2^11: both planes clamp at |a| = 448
constexpr int kMixActExp = kMixActHiExp + 11;
constexpr int kMixWLoShift = kMixActExp - kMixActHiExp;   // weight residual scale = weight value scale x 2^11
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// Saturating conversions (round 3, tools/micro/ovfl_probe.hip): with MODE.FP16_OVFL = 1 the f32 -> e4m3 conversions saturate at
// +-448 (0x7e) instead of producing NaN, and f32 -> f16 saturates at +-65504 instead of Inf (measured on gfx950 for
// v_cvt_scalef32_pk_fp8_f32, v_cvt_pk_fp8_f32 and v_cvt_f16_f32; a NaN stays a NaN).  Every kernel that writes mixed rows sets the
// bit on entry (the mode is per wave), and the v_med3_f32 clamps in front of every conversion -- 8 of ~22 instructions per 4
// values in the VALU-bound producers -- are gone.  -DVETO_FP16_OVFL=0 rebuilds the clamped form.
#ifndef VETO_FP16_OVFL
#define VETO_FP16_OVFL 1
#endif
__device__ __forceinline__ void saturating_conversions_on() {
#if VETO_FP16_OVFL
  __builtin_amdgcn_s_setreg(1 /* HW_REG_MODE */ | (23 << 6) /* bit 23: FP16_OVFL */ | ((1 - 1) << 11), 1);
#endif
}
__device__ __forceinline__ float clamp448(float x) {
#if VETO_FP16_OVFL
  return x;
#else
  return __builtin_amdgcn_fmed3f(x, -448.f, 448.f);
#endif
}
__device__ __forceinline__ uint32_t pack_e4m3x4(float a, float b, float c, float d) {
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(a), clamp448(b), 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(c), clamp448(d), r, true);
  return (uint32_t)r;
}
// e4m3(x * 2^EXP) for four values with the power-of-two scale folded into the conversion (v_cvt_scalef32_pk_fp8_f32 converts
// x / scale; tools/micro/cvt_probe.hip); the clamp moves in front of the scale: |x| <= 448 * 2^-EXP
template <int EXP>
__device__ __forceinline__ uint32_t pack_e4m3x4_scaled(float a, float b, float c, float d) {
  typedef short v2s __attribute__((ext_vector_type(2)));
  constexpr float lim = 448.f / (float)(1 << EXP), inv = 1.f / (float)(1 << EXP);
  int any;                                  // (both halves are written below: no instruction to initialise the register)
  asm("" : "=v"(any));
  v2s r = __builtin_bit_cast(v2s, any);
#if VETO_FP16_OVFL
  (void)lim;
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, a, b, inv, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, c, d, inv, true);
#else
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, __builtin_amdgcn_fmed3f(a, -lim, lim), __builtin_amdgcn_fmed3f(b, -lim, lim), inv, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, __builtin_amdgcn_fmed3f(c, -lim, lim), __builtin_amdgcn_fmed3f(d, -lim, lim), inv, true);
#endif
  return __builtin_bit_cast(uint32_t, r);
}
// byte offset of column k's fp16 inside a mixed row; its X byte is at mixed_x_offset(k), its Y byte 4 further
__host__ __device__ __forceinline__ int mixed_h_offset(int k) { return ((k >> 6) << 8) + ((k & 63) << 1); }
__host__ __device__ __forceinline__ int mixed_x_offset(int k) { return ((k >> 6) << 8) + 128 + (((k & 63) >> 2) << 3) + (k & 3); }
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// The mixed-row image of 4 consecutive activation values: h = their fp16 (two dwords), xy = {e4m3 X of the fp16 residual, e4m3 Y
// of the value}.  The residual v - float(h) comes from v_fma_mix_f32 (fp16 operand read in place, exact result) instead of a
// conversion back and a subtraction: 12 instructions per 4 values instead of 16 -- the producers of mixed rows are VALU-bound.
__device__ __forceinline__ void mixed_pack4(f32x4 v, u32x2& h, u32x2& xy) {
  float l[4];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    uint32_t hp;
    // (volatile: the conversion reads MODE -- rounding, FP16_OVFL -- which an asm statement cannot name as an input; this keeps it
    // behind saturating_conversions_on())
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(v[2 * p]), "v"(v[2 * p + 1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l[2 * p]) : "v"(hp), "v"(v[2 * p]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l[2 * p + 1]) : "v"(hp), "v"(v[2 * p + 1]));
    h[p] = hp;
  }
  xy = u32x2{pack_e4m3x4_scaled<kMixActExp>(l[0], l[1], l[2], l[3]), pack_e4m3x4_scaled<kMixActHiExp>(v[0], v[1], v[2], v[3])};
}

// 4 consecutive columns (col % 4 == 0) of an ACTIVATION row in either operand format; `row` = first byte of the row
template <int FMT>
__device__ __forceinline__ void store_act4(__bf16* row, int col, f32x4 v) {
  if constexpr (FMT == FMT_SPLIT) {
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16 h, l;
      split_bf16(v[e], h, l);
      hi[e] = h;
      lo[e] = l;
    }
    __bf16* d = row + split_index(col);
    *(bf16x4*)d = hi;
    *(bf16x4*)(d + 32) = lo;
  } else {
    char* base = (char*)row;
    u32x2 h, xy;
    mixed_pack4(v, h, xy);
    *(u32x2*)(base + mixed_h_offset(col)) = h;
    *(u32x2*)(base + mixed_x_offset(col)) = xy;
  }
}

// ---- 3-byte floats ("f24": sign, 8 exponent bits, 15 mantissa bits = the top three bytes of the fp32 rounded to nearest even).
// q / k / v between the QKV projection and the attention kernel: the attention products split every operand into bf16 hi + bf16 lo
// (a 16-bit significand), so a 16-bit significand in memory loses nothing the kernel keeps -- hi + lo of an f24 value is exact --
// and the 2 GB matrix a layer writes and reads back becomes 1.5 GB.  Four values = three dwords.
typedef __attribute__((ext_vector_type(3))) uint32_t u32x3;
__device__ __forceinline__ u32x3 pack_f24x4(f32x4 v) {
  uint32_t u[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t b = __float_as_uint(v[e]);
    const uint32_t r = b + 0x7fu + ((b >> 8) & 1u);   // round to nearest even at bit 8 (the low byte is dropped below)
    // a NaN keeps its bits (v_cmp_class + v_cndmask): the increment would carry a payload of all ones out of the mantissa and
    // store +-0; Inf (mantissa 0) is unchanged by the rounding either way
    u[e] = __builtin_isnan(v[e]) ? (b | 0x00400000u) : r;
  }
  // bytes, low address first: u0.b1 u0.b2 u0.b3 | u1.b1 u1.b2 u1.b3 | u2.b1 .. | u3.b1 u3.b2 u3.b3   (v_perm_b32: selector byte
  // values 0..3 pick bytes of the SECOND operand, 4..7 of the first)
  return u32x3{__builtin_amdgcn_perm(u[1], u[0], 0x05030201u), __builtin_amdgcn_perm(u[2], u[1], 0x06050302u),
               __builtin_amdgcn_perm(u[3], u[2], 0x07060503u)};
}
__device__ __forceinline__ f32x4 unpack_f24x4(uint32_t d0, uint32_t d1, uint32_t d2) {
  // value e occupies bytes 3e .. 3e+2 of the 12; 0x0c selects a zero byte
  return f32x4{__uint_as_float(d0 << 8), __uint_as_float(__builtin_amdgcn_perm(d1, d0, 0x0504030cu)),
               __uint_as_float(__builtin_amdgcn_perm(d2, d1, 0x0403020cu)), __uint_as_float(d2 & 0xffffff00u)};
}

// 8 consecutive columns (col % 8 == 0) of a mixed activation row: two 16-byte stores
// NT: non-temporal stores (rows that are next read a launch or more later: DESIGN.md section 7.4, cache policy)
template <bool NT = false>
__device__ __forceinline__ void store_act8_mixed(__bf16* row, int col, f32x4 v0, f32x4 v1) {
  char* base = (char*)row;
  u32x2 h0, h1, xy0, xy1;
  mixed_pack4(v0, h0, xy0);
  mixed_pack4(v1, h1, xy1);
  if constexpr (NT) {
    __builtin_nontemporal_store(u32x4{h0[0], h0[1], h1[0], h1[1]}, (u32x4*)(base + mixed_h_offset(col)));
    __builtin_nontemporal_store(u32x4{xy0[0], xy0[1], xy1[0], xy1[1]}, (u32x4*)(base + mixed_x_offset(col)));
  } else {
    *(u32x4*)(base + mixed_h_offset(col)) = u32x4{h0[0], h0[1], h1[0], h1[1]};
    *(u32x4*)(base + mixed_x_offset(col)) = u32x4{xy0[0], xy0[1], xy1[0], xy1[1]};
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Exact-erf GELU (nn.GELU default, model_veto.py:140).  erf through Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, i.e. below the 2^-17 noise of the split-bf16 products around it) instead of
// the ~30-instruction libm erff: the FeedForward epilogue applies it to 96 values per lane.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);  // raw v_rcp_f32 (1 ulp); __frcp_rn expands to a 10-instruction IEEE division
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// d/dx of the exact-erf GELU, Phi(x) + x phi(x), with the same erf approximation: exp(-z^2) of z = |x| / sqrt(2) IS exp(-x^2 / 2), so the
// density shares the one exponential (|error| <= 1e-7 absolute; the training path's fc2 input-gradient epilogue)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __expf(-z * z);
  const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * e, x));
  return cdf + x * (0.39894228040143267794f * e);
}

// The same function through ONE exponential: x Phi(x) = x sigmoid(g(x)) with g = logit(Phi) fitted by an odd polynomial of
// degree 11 on [-6, 6] (minimax-reweighted least squares, tools/precision_study.py --gelu; |error| <= 1e-6 absolute in fp32,
// beyond +-6 Phi is 0 / 1 to 1e-9).  10 VALU + 2 transcendental instructions against 17 + 2: the FeedForward epilogue of the
// inference path is VALU-bound (96 values per lane).  The coefficients carry the factor -log2(e) of v_exp_f32.
__device__ __forceinline__ float gelu_sigmoid(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -6.f, 6.f);
  const float x2 = xc * xc;
  float p = 1.3906235096783348e-07f;
  p = fmaf(p, x2, -7.393736268568318e-06f);
  p = fmaf(p, x2, 0.000129951280541718f);
  p = fmaf(p, x2, 0.00019161769887432456f);
  p = fmaf(p, x2, -0.10496557503938675f);
  p = fmaf(p, x2, -2.3021585941314697f);
  const float e = __builtin_amdgcn_exp2f(p * xc);             // exp(-g(x)), <= 2^23
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Two values at a time on the packed fp32 instructions (v_pk_mul / v_pk_fma / v_pk_add: two lanes' worth of work per issue slot);
// same operations in the same order as gelu_sigmoid, so the results are bit-identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_sigmoid2(f32x2 x) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -6.f, 6.f), __builtin_amdgcn_fmed3f(x[1], -6.f, 6.f)};
  const f32x2 x2 = xc * xc;
  auto both = [](float c) { return f32x2{c, c}; };
  f32x2 p = both(1.3906235096783348e-07f);
  p = __builtin_elementwise_fma(p, x2, both(-7.393736268568318e-06f));
  p = __builtin_elementwise_fma(p, x2, both(0.000129951280541718f));
  p = __builtin_elementwise_fma(p, x2, both(0.00019161769887432456f));
  p = __builtin_elementwise_fma(p, x2, both(-0.10496557503938675f));
  p = __builtin_elementwise_fma(p, x2, both(-2.3021585941314697f));
  const f32x2 t = p * xc;
  const f32x2 d = both(1.0f) + f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  return x * f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

// Counter-based dropout mask (training path): element `idx` of dropout site `seed` is kept iff the top 24 bits of a
// splitmix64 hash reach the threshold p * 2^24.  Forward and backward recompute the same mask; nothing is stored.
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx, unsigned thresh) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 40) >= thresh;
}

}  // namespace veto
