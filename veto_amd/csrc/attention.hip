// Per-pair multi-head self-attention over the 19 relation tokens (model_veto.py:85-96):
//   dots = q k^T * dh^-0.5 ; attn = softmax(dots, -1) ; out = attn v ; heads merged 'b h n d -> b n (h d)'.
// One wave per (pair, head).  q/k/v of the head are staged in LDS in fp32; the 19x19 scores, the
// softmax and the PV product are computed in fp32 on the vector ALU (QK^T + AV are 0.8 % of the
// path's FLOPs).  The result is written as the hi/lo bf16 planes the
// out-projection GEMM consumes.  cls_only = last layer: only token 0's query is needed
// (model_veto.py:23 consumes x[:, 0] only), k/v still cover all 19 tokens.
#include "common.h"
#include "kernels.h"

namespace veto {

namespace {

constexpr int kWavesPerBlock = 4;

__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a, int dh, int ldh /* dh + 4 */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * kWavesPerBlock + w;
  const int per_wave = 3 * kTokens * ldh + kTokens * 20;
  float* sq = (float*)smem_raw + (size_t)w * per_wave;
  float* sk = sq + kTokens * ldh;
  float* sv = sk + kTokens * ldh;
  float* sp = sv + kTokens * ldh;  // [19][20]
  const bool active = gw < (long)a.n_pair * a.heads;
  const int pair = active ? (int)(gw / a.heads) : 0;
  const int head = active ? (int)(gw % a.heads) : 0;
  const int d4n = dh >> 2;
  const float* base = a.qkv + (size_t)pair * kTokens * (3 * kDim) + head * dh;

  // stage q, k, v [19][dh] -> LDS
  const int per_mat = kTokens * d4n;
  for (int e = lane; e < 3 * per_mat; e += 64) {
    const int mat = e / per_mat, rem = e % per_mat, i = rem / d4n, d4 = rem % d4n;
    const f32x4 v = *(const f32x4*)(base + (size_t)i * (3 * kDim) + mat * kDim + d4 * 4);
    *(f32x4*)(sq + mat * kTokens * ldh + i * ldh + d4 * 4) = v;
  }
  __syncthreads();

  const float scale = 1.0f / sqrtf((float)dh);
  const int nq = a.cls_only ? 1 : kTokens;
  for (int e = lane; e < nq * kTokens; e += 64) {
    const int i = e / kTokens, j = e % kTokens;
    float acc = 0.f;
    for (int d4 = 0; d4 < d4n; ++d4) {
      const f32x4 qv = *(const f32x4*)(sq + i * ldh + d4 * 4);
      const f32x4 kv = *(const f32x4*)(sk + j * ldh + d4 * 4);
      acc += qv[0] * kv[0];
      acc += qv[1] * kv[1];
      acc += qv[2] * kv[2];
      acc += qv[3] * kv[3];
    }
    sp[i * 20 + j] = acc * scale;
  }
  __syncthreads();
  if (lane < nq) {
    float* row = sp + lane * 20;
    float mx = row[0];
    for (int j = 1; j < kTokens; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.f;
    for (int j = 0; j < kTokens; ++j) { const float ev = expf(row[j] - mx); row[j] = ev; sum += ev; }
    const float inv = 1.f / sum;
    for (int j = 0; j < kTokens; ++j) row[j] *= inv;
  }
  __syncthreads();
  if (!active) return;
  for (int e = lane; e < nq * d4n; e += 64) {
    const int i = e / d4n, d4 = e % d4n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < kTokens; ++j) {
      const float pj = sp[i * 20 + j];
      const f32x4 vv = *(const f32x4*)(sv + j * ldh + d4 * 4);
      acc += pj * vv;
    }
    const size_t row = a.cls_only ? (size_t)pair : (size_t)pair * kTokens + i;
    const size_t off = row * kDim + head * dh + d4 * 4;
    bf16x4 hi, lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      __bf16 h, l;
      split_bf16(acc[c], h, l);
      hi[c] = h;
      lo[c] = l;
    }
    *(bf16x4*)(a.o_hi + off) = hi;
    *(bf16x4*)(a.o_lo + off) = lo;
  }
}

}  // namespace

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  if (kDim % a.heads != 0) return hipErrorInvalidValue;
  const int dh = kDim / a.heads;
  if (dh % 4 != 0) return hipErrorInvalidValue;
  const int ldh = dh + 4;
  const size_t lds = (size_t)kWavesPerBlock * (3 * kTokens * ldh + kTokens * 20) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attention_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const long waves = (long)a.n_pair * a.heads;
  const unsigned blocks = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
  VETO_LAUNCH(attention_kernel, dim3(blocks), dim3(256), lds, s, a, dh, ldh);
  return hipGetLastError();
}

}  // namespace veto
