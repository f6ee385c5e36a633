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
