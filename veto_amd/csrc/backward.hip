// Backward building blocks of the relation transformer; chained by veto_backward
// (veto_abi.hip) and exposed one by one through test hooks that are checked against autograd.
//   attention_backward   model_veto.py:85-96   (dQ, dK, dV) from dOut and the saved q, k, v of one (pair, head)
//   layernorm_backward   model_veto.py:125-132 dx, and per-block partial sums of dgamma / dbeta
//   gelu_backward        model_veto.py:140     dpre = dh * gelu'(pre), exact-erf GELU
//   column_sums          bias gradients: db[n] = sum over the token rows of dy[:, n] (two-stage, fixed order)
#include "common.h"
#include "kernels.h"

#include <cstdlib>

namespace veto {

namespace {

constexpr int kColChunks = 256;    // row chunks of the two-stage column sums (fixed order -> deterministic)

constexpr int kAttnBwdThreads = 320;

// One workgroup of five waves per (pair, head), everything in fp32.  With S = q k^T * scale, P = softmax(S), O = P v:
//   dV = P^T dO,  dP = dO v^T,  dS = P * (dP - rowsum(dP * P)),  dQ = dS k * scale,  dK = dS^T q * scale.
// All five products are register-blocked 4 x 4 with 16-byte LDS reads (2 bytes of LDS traffic per FMA; the first
// version read two scalars per FMA and was bound by the LDS array at 2.2 ms per launch):
//   A  S, dP   [20 x 20] outputs in 25 tiles of 4 x 4, the head dimension cut into NS slices -> 2 * NS * 25 work items,
//              partial sums in LDS, folded in slice order (deterministic)
//   B  softmax / dS: 16 lanes per row, reductions by DPP row rotations
//   C  dQ, dK, dV  [20 x DH] outputs: work item = (product, 4 tokens, 4 head columns), contraction over the 19 tokens
// Token row 19 of every image is zero padding.
// all-reduce (sum or max) over the 16 lanes of a DPP row by rotations row_ror:8/4/2/1; every lane of the row must be active
template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <bool MAX>
__device__ __forceinline__ float row_all(float v) {
  float o;
  o = dpp_rot<0x128>(v); v = MAX ? fmaxf(v, o) : v + o;
  o = dpp_rot<0x124>(v); v = MAX ? fmaxf(v, o) : v + o;
  o = dpp_rot<0x122>(v); v = MAX ? fmaxf(v, o) : v + o;
  o = dpp_rot<0x121>(v); v = MAX ? fmaxf(v, o) : v + o;
  return v;
}

template <int DH>
__global__ __launch_bounds__(kAttnBwdThreads) void attention_backward_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                             float* __restrict__ dqkv, __bf16* __restrict__ dqkv_split,
                                                                             int n_pair, int heads, int cls_only) {
  constexpr int TP = 20, LD = DH + 4, D4 = DH / 4, MAT = TP * LD;
  constexpr int NS = DH == 144 ? 3 : 6, SL = DH / NS;       // slices of the head dimension in phase A
  constexpr int CHUNKS = 4 * kTokens * D4, ROUNDS = (CHUNKS + kAttnBwdThreads - 1) / kAttnBwdThreads;
  static_assert(DH % 12 == 0 && SL % 4 == 0, "head dimension");
  __shared__ __attribute__((aligned(16))) float img[4 * MAT];          // q, k, v, dO as [20][LD]
  __shared__ __attribute__((aligned(16))) float part[2 * NS * TP * TP];
  __shared__ __attribute__((aligned(16))) float p[TP * TP], ds[TP * TP], dst_t[TP * TP];
  const int tid = threadIdx.x;
  const long total = (long)n_pair * heads;
  const float* q = img;
  const float* k = img + MAT;
  const float* v = img + 2 * MAT;
  const float* go = img + 3 * MAT;
  for (int e = tid; e < 4 * LD; e += kAttnBwdThreads) img[(e / LD) * MAT + kTokens * LD + e % LD] = 0.f;   // padding row 19
  // Persistent workgroups: the operands of the NEXT (pair, head) are fetched into registers while this one is computed
  // (one workgroup per item spent 0.8 of its 2.3 ms waiting for these loads and 0.7 ms on workgroup turnover).
  f32x4 pre[ROUNDS];
  auto fetch = [&](long item) {
    const int pair = (int)(item / heads), head = (int)(item % heads);
    const float* src = qkv + (size_t)pair * kTokens * (3 * kDim) + head * DH;
    // cls_only (last layer, model_veto.py:23 consumes x[:, 0] only): dout is compact [n_pair, 576] (the CLS query's row), the
    // queries and output gradients of tokens 1..18 do not exist and enter as zeros -> dQ rows 1..18 = 0, dK / dV from row 0
    const float* gsrc = dout + (cls_only ? (size_t)pair * kDim : (size_t)pair * kTokens * kDim) + head * DH;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int e = tid + r * kAttnBwdThreads;
      if (e < CHUNKS) {
        const int mat = e / (kTokens * D4), rem = e % (kTokens * D4), t = rem / D4, c = rem % D4;
        if (cls_only && t > 0 && (mat == 0 || mat == 3)) pre[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        else pre[r] = mat < 3 ? *(const f32x4*)(src + (size_t)t * 3 * kDim + mat * kDim + 4 * c) : *(const f32x4*)(gsrc + (size_t)t * kDim + 4 * c);
      }
    }
  };
  // contiguous item ranges: the heads of one pair (adjacent 288-byte pieces of the same rows) follow each other on one CU
  const long per_wg = (total + gridDim.x - 1) / gridDim.x;
  long item = blockIdx.x * per_wg;
  const long last = item + per_wg < total ? item + per_wg : total;
  if (item < last) fetch(item);
  for (; item < last; ++item) {
  const int pair = (int)(item / heads), head = (int)(item % heads);
  __syncthreads();                 // the previous item's phase C has finished reading the images
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = tid + r * kAttnBwdThreads;
    if (e < CHUNKS) {
      const int mat = e / (kTokens * D4), rem = e % (kTokens * D4), t = rem / D4, c = rem % D4;
      *(f32x4*)(img + mat * MAT + t * LD + 4 * c) = pre[r];
    }
  }
  if (item + 1 < last) fetch(item + 1);
  __syncthreads();
  // ---- A: partial S = q k^T and dP = dO v^T ---------------------------------------------------------------------------
  const int nt = cls_only ? 5 : 25;        // cls_only: only query row 0 exists -> the five tiles of the first row group
  for (int w = tid; w < 2 * NS * nt; w += kAttnBwdThreads) {
    const int prod = w / (NS * nt), rem = w % (NS * nt), sl = rem / nt, tile = rem % nt;
    const int i0 = (tile / 5) * 4, j0 = (tile % 5) * 4;
    const float* A = (prod ? go : q) + i0 * LD + sl * SL;
    const float* B = (prod ? v : k) + j0 * LD + sl * SL;
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int d = 0; d < SL; d += 4) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[r] = *(const f32x4*)(A + r * LD + d); b[r] = *(const f32x4*)(B + r * LD + d); }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] += (a[r][0] * b[c][0] + a[r][1] * b[c][1]) + (a[r][2] * b[c][2] + a[r][3] * b[c][3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) *(f32x4*)(part + (prod * NS + sl) * (TP * TP) + (i0 + r) * TP + j0) = acc[r];
  }
  __syncthreads();
  // ---- B: P = softmax(S * scale) per row, dS = P * (dP - sum_j dP P); 16 lanes per row (columns j and j + 16), all 20
  // rows at once; the row reductions are DPP rotations inside the 16-lane row (no LDS crossbar, a few cycles each)
  const float scale = 1.0f / sqrtf((float)DH);
  {
    const int i = tid >> 4, j = tid & 15;
    const int nq = cls_only ? 1 : kTokens;   // query rows that exist; the others get P = dS = 0
    const bool v1 = i < nq, v2 = i < nq && j < 3;
    float s1 = 0.f, d1 = 0.f, s2 = 0.f, d2 = 0.f;
    if (v1) {
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) { s1 += part[sl * (TP * TP) + i * TP + j]; d1 += part[(NS + sl) * (TP * TP) + i * TP + j]; }
    }
    if (v2) {
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) { s2 += part[sl * (TP * TP) + i * TP + 16 + j]; d2 += part[(NS + sl) * (TP * TP) + i * TP + 16 + j]; }
    }
    s1 = v1 ? s1 * scale : -INFINITY;
    s2 = v2 ? s2 * scale : -INFINITY;
    const float mx = row_all<true>(fmaxf(s1, s2));
    const float e1 = v1 ? expf(s1 - mx) : 0.f, e2 = v2 ? expf(s2 - mx) : 0.f;
    const float sum = row_all<false>(e1 + e2), dotr = row_all<false>(e1 * d1 + e2 * d2);
    const float inv = v1 ? 1.f / sum : 0.f, dot = dotr * inv;
    const float p1 = e1 * inv, p2 = e2 * inv;
    const float g1 = p1 * (d1 - dot), g2 = p2 * (d2 - dot);
    p[i * TP + j] = p1;
    ds[i * TP + j] = g1;
    dst_t[j * TP + i] = g1;
    if (j < 4) {          // columns 16 .. 19 (19 = padding, zero)
      p[i * TP + 16 + j] = p2;
      ds[i * TP + 16 + j] = g2;
      dst_t[(16 + j) * TP + i] = g2;
    }
  }
  __syncthreads();
  // ---- C: dQ[t] = scale sum_j dS[t][j] k[j], dK[t] = scale sum_i dS[i][t] q[i], dV[t] = sum_i P[i][t] dO[i] ------------
  float* dst = dqkv + (size_t)pair * kTokens * (3 * kDim) + head * DH;
  for (int w = tid; w < 3 * 5 * D4; w += kAttnBwdThreads) {
    const int prod = w / (5 * D4), rem = w % (5 * D4), t0 = (rem / D4) * 4, c = rem % D4;
    const float* coef = (prod == 0 ? dst_t : prod == 1 ? ds : p) + t0;      // [contraction index][token t0 .. t0 + 3]
    const float* mat = (prod == 0 ? k : prod == 1 ? q : go) + 4 * c;
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    // cls_only: dQ rows 1..18 are zero (written as such), dK / dV reduce over the one query row
    const int jn = !cls_only ? kTokens : prod > 0 ? 1 : t0 > 0 ? 0 : kTokens;
#pragma unroll 4
    for (int j = 0; j < jn; ++j) {
      const f32x4 cf = *(const f32x4*)(coef + j * TP);
      const f32x4 m = *(const f32x4*)(mat + j * LD);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += cf[r] * m;
    }
    const float f = prod < 2 ? scale : 1.f;
    if (dqkv_split) {   // training path: straight to the split rows [M, 2 * 1728] the QKV weight / input gradient GEMMs read
      const int col = prod * kDim + head * DH + 4 * c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (t0 + r >= kTokens) continue;
        const f32x4 o = acc[r] * f;
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 hh, ll;
          split_bf16(o[e], hh, ll);
          hi[e] = hh;
          lo[e] = ll;
        }
        __bf16* d = dqkv_split + ((size_t)pair * kTokens + t0 + r) * (2 * 3 * kDim) + split_index(col);
        *(bf16x4*)d = hi;
        *(bf16x4*)(d + 32) = lo;
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (t0 + r < kTokens) *(f32x4*)(dst + (size_t)(t0 + r) * 3 * kDim + prod * kDim + 4 * c) = acc[r] * f;
  }
  }  // items
}

// ---- MFMA form (head widths 72 and 96, all 19 queries) ----------------------------------------------------------------
// One wave per (pair, head), all five products on v_mfma_f32_32x32x16_bf16 in the 3-term split-bf16 scheme of the forward
// (attention.hip).  Both orientations of the scores are computed, because an accumulator serves as the A operand of the next
// product only along its ROW index:
//   lane = query:  S^T = K Q^T, dP^T = V dO^T  -> softmax statistics (max, 1/sum, D = sum_j P dP) in-lane, dS [query x key]
//                  = the A operand of dQ = dS K
//   lane = key:    S = Q K^T, dP = dO V^T      -> P, dS rebuilt with the statistics (through LDS): the A operands of
//                  dV = P^T dO and dK = dS^T Q
// The B operands (K, dO, Q with the TOKEN as contraction index) are read from the same row-major hi / lo images the score
// products use, through ds_read_b64_tr_b16 (4 tokens x 16 columns per 16-lane group, column-major): no transposed copies.
// Element e of k-step s of an accumulator-operand is token 16 s + 8 (e >> 2) + 4 h + (e & 3) (h = lane >> 5), which is the
// order the transposed reads are issued in.  Image row 19 is zero and stands in for tokens 19..31.
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef short bw_s16x4 __attribute__((ext_vector_type(4)));
typedef short bw_s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 lds_tr_pair(const char* p0, const char* p1) {
  const bw_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bw_s16x4*)p0);
  const bw_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bw_s16x4*)p1);
  const bw_s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// QF24: qkv holds 3-byte floats (common.h: rows of 1728 x 3 bytes) -- lossless here: every operand is split into bf16 hi + lo anyway
template <int DH, bool QF24 = false>
__global__ __launch_bounds__(128) void attention_backward_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                      float* __restrict__ dqkv, __bf16* __restrict__ dqkv_split,
                                                                      int n_pair, int heads, int cls_only) {
  constexpr int DHP = (DH + 15) / 16 * 16;   // contraction extent of the score products (zero padded)
  constexpr int RB = DHP * 2;                // bytes per image row (bf16)
  constexpr int NT = (DH + 31) / 32;         // 32-wide output tiles over the head dimension
  constexpr int PLANE = 20 * RB;             // 19 token rows + the zero row
  constexpr int WAVE_LDS = 8 * PLANE + 3 * 32 * 4 + 64;   // q, k, v, dO (hi, lo), the statistics, slack for the padded tr reads
  constexpr int CH = DH / 8, PER_MAT = kTokens * CH, ROUNDS = (4 * PER_MAT + 63) / 64;
  static_assert(DH % 8 == 0 && kTokens * DH * 4 <= 2 * PLANE, "layout");
  __shared__ __attribute__((aligned(16))) char smem[2 * WAVE_LDS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * 2 + w, total = (long)n_pair * heads;
  const bool active = gw < total;
  const long item = active ? gw : total - 1;
  const int pair = (int)(item / heads), head = (int)(item % heads);
  char* base = smem + w * WAVE_LDS;
  auto plane = [&](int mat, int lo) { return base + (2 * mat + lo) * PLANE; };   // mat: 0 q, 1 k, 2 v, 3 dO
  float* stat = (float*)(base + 8 * PLANE);                                       // [3][32]: max, 1 / sum, D
  const float* src0 = qkv + (size_t)pair * kTokens * (3 * kDim) + head * DH;
  const char* src24 = (const char*)qkv + ((size_t)pair * kTokens * (3 * kDim) + head * DH) * 3;      // (QF24)
  // cls_only (last layer): dout is compact [n_pair, 576]; the queries / output gradients of tokens 1..18 enter as zeros
  const float* gsrc = dout + (cls_only ? (size_t)pair * kDim : (size_t)pair * kTokens * kDim) + head * DH;

  // ---- global -> registers (all loads in flight) -> bf16 hi / lo images -------------------------------------------------
  f32x4 ld[ROUNDS][2];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = lane + 64 * r;
    const int mat = e / PER_MAT, rem = e % PER_MAT, i = rem / CH, c = rem % CH;
    if (cls_only && i > 0 && (mat == 0 || mat == 3)) {
      ld[r][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      ld[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (e < 4 * PER_MAT) {
      if (QF24 && mat < 3) {      // 8 values = 24 bytes (8-byte aligned: 16 + 8)
        const char* sp = src24 + ((size_t)i * (3 * kDim) + mat * kDim + c * 8) * 3;
        const u32x2 d01 = *(const u32x2*)sp, d23 = *(const u32x2*)(sp + 8), d45 = *(const u32x2*)(sp + 16);
        ld[r][0] = unpack_f24x4(d01[0], d01[1], d23[0]);
        ld[r][1] = unpack_f24x4(d23[1], d45[0], d45[1]);
      } else {
      const float* src = mat < 3 ? src0 + (size_t)i * (3 * kDim) + mat * kDim + c * 8 : gsrc + (size_t)i * kDim + c * 8;
      ld[r][0] = *(const f32x4*)src;
      ld[r][1] = *(const f32x4*)(src + 4);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int e = lane + 64 * r;
    if (e >= 4 * PER_MAT) continue;
    const int mat = e / PER_MAT, rem = e % PER_MAT, i = rem / CH, c = rem % CH;
    bf16x8 hi, lo;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      __bf16 hh, ll;
      split_bf16(ld[r][t >> 2][t & 3], hh, ll);
      hi[t] = hh;
      lo[t] = ll;
    }
    *(bf16x8*)(plane(mat, 0) + i * RB + c * 16) = hi;
    *(bf16x8*)(plane(mat, 1) + i * RB + c * 16) = lo;
  }
  for (int idx = lane; idx < 8 * (RB / 4); idx += 64)      // the zero row of every plane
    *(uint32_t*)(base + (idx / (RB / 4)) * PLANE + kTokens * RB + 4 * (idx % (RB / 4))) = 0u;
  if (DHP > DH) {                                            // the contraction padding of the token rows
    for (int idx = lane; idx < 8 * kTokens; idx += 64) {
      char* dst = base + (idx / kTokens) * PLANE + (idx % kTokens) * RB + DH * 2;
#pragma unroll
      for (int t = 0; t < (DHP - DH) / 2; ++t) *(uint32_t*)(dst + 4 * t) = 0u;
    }
  }
  for (int idx = lane; idx < 16; idx += 64) *(uint32_t*)(base + 8 * PLANE + 3 * 32 * 4 + 4 * idx) = 0u;   // slack behind the last plane
  __syncthreads();

  // ---- the four score products ---------------------------------------------------------------------------------------------
  const int r = lane & 31, h = lane >> 5;
  const int rr = r < kTokens ? r : kTokens;        // lanes beyond the 19 tokens take the zero row
  f32x16 st, dpt, sk, dpk;                         // S^T, dP^T (lane = query) ; S, dP (lane = key)
#pragma unroll
  for (int t = 0; t < 16; ++t) { st[t] = 0.f; dpt[t] = 0.f; sk[t] = 0.f; dpk[t] = 0.f; }
#pragma unroll
  for (int s = 0; s < DHP / 16; ++s) {
    const int off = rr * RB + (16 * s + 8 * h) * 2;
    const bf16x8 qh = *(const bf16x8*)(plane(0, 0) + off), ql = *(const bf16x8*)(plane(0, 1) + off);
    const bf16x8 kh = *(const bf16x8*)(plane(1, 0) + off), kl = *(const bf16x8*)(plane(1, 1) + off);
    const bf16x8 vh = *(const bf16x8*)(plane(2, 0) + off), vl = *(const bf16x8*)(plane(2, 1) + off);
    const bf16x8 gh = *(const bf16x8*)(plane(3, 0) + off), gl = *(const bf16x8*)(plane(3, 1) + off);
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, st, 0, 0, 0);      // rows = keys, columns = queries
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, st, 0, 0, 0);
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, st, 0, 0, 0);
    dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, gh, dpt, 0, 0, 0);
    dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, gl, dpt, 0, 0, 0);
    dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, gh, dpt, 0, 0, 0);
    sk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ql, kh, sk, 0, 0, 0);      // rows = queries, columns = keys
    sk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh, kl, sk, 0, 0, 0);
    sk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qh, kh, sk, 0, 0, 0);
    dpk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, vh, dpk, 0, 0, 0);
    dpk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, vl, dpk, 0, 0, 0);
    dpk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, vh, dpk, 0, 0, 0);
  }

  // ---- lane = query: softmax over the keys (16 registers here + 16 in lane ^ 32), D, dS ------------------------------------
  const float scale = 1.0f / sqrtf((float)DH);
  auto tok = [&](int t) { return (t & 3) + 8 * (t >> 2) + 4 * h; };      // token of accumulator register t
  bf16x8 dsh[2], dsl[2];                                                // dS [query x key] as the A operand of dQ
  {
    float p[16];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      p[t] = tok(t) < kTokens ? st[t] * scale : -INFINITY;
      mx = fmaxf(mx, p[t]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      p[t] = expf(p[t] - mx);
      sum += p[t];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      p[t] *= inv;
      dsum += p[t] * dpt[t];
    }
    dsum += __shfl_xor(dsum, 32, 64);
    if (h == 0) { stat[r] = mx; stat[32 + r] = inv; stat[64 + r] = dsum; }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      __bf16 hh, ll;
      split_bf16(p[t] * (dpt[t] - dsum), hh, ll);
      dsh[t >> 3][t & 7] = hh;
      dsl[t >> 3][t & 7] = ll;
    }
  }
  __syncthreads();
  // ---- lane = key: P and dS from the statistics of each register's query --------------------------------------------------
  bf16x8 pkh[2], pkl[2], dkh[2], dkl[2];                                // P^T, dS^T [key x query] as A operands of dV, dK
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = tok(t);
    const float pv = (r < kTokens && i < kTokens) ? expf(sk[t] * scale - stat[i]) * stat[32 + i] : 0.f;
    const float dv = pv * (dpk[t] - stat[64 + i]);
    __bf16 hh, ll;
    split_bf16(pv, hh, ll);
    pkh[t >> 3][t & 7] = hh;
    pkl[t >> 3][t & 7] = ll;
    split_bf16(dv, hh, ll);
    dkh[t >> 3][t & 7] = hh;
    dkl[t >> 3][t & 7] = ll;
  }

  // ---- dQ = dS K, dK = dS^T Q, dV = P^T dO: B operands by transposed reads of the row-major images --------------------------
  const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  auto tr_b = [&](int mat, int lo, int n, int s) {          // B fragment: columns 32 n + (lane & 31), tokens of k-step s
    const char* pl = plane(mat, lo) + (32 * n + 16 * (g4 & 1) + 4 * tp) * 2;
    int t0 = 16 * s + 4 * h + tq, t1 = t0 + 8;
    t0 = t0 < kTokens ? t0 : kTokens;
    t1 = t1 < kTokens ? t1 : kTokens;
    return lds_tr_pair(pl + t0 * RB, pl + t1 * RB);
  };
  float* o_lds = (float*)plane(2, 0);                        // [19][DH] fp32 staging in the (dead) V planes
  float* dst = dqkv ? dqkv + (size_t)pair * kTokens * (3 * kDim) + head * DH : nullptr;
#pragma unroll
  for (int prod = 0; prod < 3; ++prod) {
    const int bmat = prod == 0 ? 1 : prod == 1 ? 0 : 3;      // B = K, Q, dO
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      f32x16 o;
#pragma unroll
      for (int t = 0; t < 16; ++t) o[t] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 bh = tr_b(bmat, 0, n, s), bl = tr_b(bmat, 1, n, s);
        const bf16x8 ah = prod == 0 ? dsh[s] : prod == 1 ? dkh[s] : pkh[s];
        const bf16x8 al = prod == 0 ? dsl[s] : prod == 1 ? dkl[s] : pkl[s];
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, o, 0, 0, 0);
      }
      const int d = 32 * n + r;
      if (d < DH) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int i = tok(t);
          if (i < kTokens) o_lds[i * DH + d] = o[t] * (prod < 2 ? scale : 1.f);
        }
      }
    }
    __syncthreads();
    if (active) {
      for (int e = lane; e < kTokens * CH; e += 64) {
        const int i = e / CH, c = e % CH;
        const f32x4 v0 = *(const f32x4*)(o_lds + i * DH + c * 8), v1 = *(const f32x4*)(o_lds + i * DH + c * 8 + 4);
        if (dqkv_split) {
          bf16x8 hi, lo;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            __bf16 hh, ll;
            split_bf16(v0[t], hh, ll);
            hi[t] = hh;
            lo[t] = ll;
            split_bf16(v1[t], hh, ll);
            hi[4 + t] = hh;
            lo[4 + t] = ll;
          }
          __bf16* d2 = dqkv_split + ((size_t)pair * kTokens + i) * (2 * 3 * kDim) + split_index(prod * kDim + head * DH + c * 8);
          *(bf16x8*)d2 = hi;
          *(bf16x8*)(d2 + 32) = lo;
        } else {
          float* d2 = dst + (size_t)i * (3 * kDim) + prod * kDim + c * 8;
          *(f32x4*)d2 = v0;
          *(f32x4*)(d2 + 4) = v1;
        }
      }
    }
    __syncthreads();
  }
}


// LayerNorm backward over 576-wide rows, half a wave per row (lane q holds the 8-byte chunks q + 32 j; 16 lanes per row
// with 16-byte chunks needed 244 VGPRs = 2 waves / SIMD and ran at 2.9 TB/s):
//   xhat = (x - mean) * rstd, g = dy * gamma, dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dres)
// dgamma / dbeta: a thread always owns the same 18 columns, so it adds its kLnRowsPerBlock / 8 rows in registers;
// the 8 row slots of the block then meet in LDS once per block, and the block writes one partial row [2, 576]
// that a two-stage column sum folds in a fixed order.
constexpr int kLnRowsPerBlock = 64;
constexpr int kLnLanes = 32;                       // lanes per row: 18 values per lane and tensor (9 x 8-byte accesses)
constexpr int kLnSlots = 256 / kLnLanes;           // rows in flight per block
typedef float f32x2 __attribute__((ext_vector_type(2)));

// SPLIT (round 6): the result is the gradient matrix of the Linear behind this LayerNorm in the backward chain, so its split rows (the
// operand of that Linear's two gradient GEMMs) and the column sums of every 32 rows (its bias gradient's partials, [rows / 32, 576]) leave
// from here, with the forward's dropout mask applied where the Linear's output was dropped (drop_thresh != 0: the attention out
// projection) -- the preparation pass that read the matrix back (train.hip, prep_grad_kernel) is gone for these Linears.
template <bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void layernorm_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ gamma, const float* dres,
                                                                 float* dx, float* __restrict__ partial, int rows,      // dres may be dx
                                                                 __bf16* __restrict__ split_out, float* __restrict__ colp,
                                                                 unsigned long long drop_seed, unsigned drop_thresh, float drop_scale) {
  __shared__ float s_dg[kDim], s_db[kDim], s_w[kDim];
  __shared__ float s_cs[SPLIT ? 2 : 1][SPLIT ? kDim : 1];
  for (int c = threadIdx.x; c < kDim; c += 256) {
    s_dg[c] = 0.f; s_db[c] = 0.f; s_w[c] = gamma[c];
    if (SPLIT) { s_cs[0][c] = 0.f; s_cs[SPLIT ? 1 : 0][c] = 0.f; }
  }
  __syncthreads();
  f32x2 acc_c[SPLIT ? 9 : 1];
  if (SPLIT) {
#pragma unroll
    for (int j = 0; j < 9; ++j) acc_c[j] = f32x2{0.f, 0.f};
  }
  const int q = threadIdx.x & (kLnLanes - 1);
  auto gsum = [](float t) {
#pragma unroll
    for (int o = kLnLanes / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  };
  f32x2 acc_g[9], acc_b[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    acc_g[j] = f32x2{0.f, 0.f};
    acc_b[j] = f32x2{0.f, 0.f};
  }
  // (SPLIT: the column sums of rows 0..31 and 32..63 of the block are flushed behind iterations 3 and 7)
  auto flush_cs = [&](int grp) {
    if constexpr (SPLIT) {
#pragma unroll
      for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          atomicAdd(&s_cs[grp][2 * (q + kLnLanes * j) + e], acc_c[j][e]);
          acc_c[j][e] = 0.f;
        }
    }
  };
  for (int it = 0; it < kLnRowsPerBlock / kLnSlots; ++it) {
    const int row = blockIdx.x * kLnRowsPerBlock + it * kLnSlots + threadIdx.x / kLnLanes;
    if (row >= rows) {      // uniform per half wave
      if (SPLIT && (it & 3) == 3) flush_cs(it >> 2);
      continue;
    }
    const float* xr = x + (size_t)row * kDim;
    const float* gr = dy + (size_t)row * kDim;
    f32x2 xv[9], gv[9];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      xv[j] = *(const f32x2*)(xr + 2 * (q + kLnLanes * j));
      gv[j] = *(const f32x2*)(gr + 2 * (q + kLnLanes * j));
      s += xv[j][0] + xv[j][1];
    }
    const float mean = gsum(s) * (1.f / kDim);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e) { const float d = xv[j][e] - mean; sq += d * d; }
    const float rstd = 1.f / sqrtf(gsum(sq) * (1.f / kDim) + 1e-5f);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const f32x2 w = *(const f32x2*)(s_w + 2 * (q + kLnLanes * j));
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float xh = (xv[j][e] - mean) * rstd;
        acc_g[j][e] += gv[j][e] * xh;
        acc_b[j][e] += gv[j][e];
        const float g = gv[j][e] * w[e];
        xv[j][e] = xh;      // keep xhat
        gv[j][e] = g;       // keep g
        sg += g;
        sgx += g * xh;
      }
    }
    const float mg = gsum(sg) * (1.f / kDim), mgx = gsum(sgx) * (1.f / kDim);
    float* dr = dx + (size_t)row * kDim;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      f32x2 o;
#pragma unroll
      for (int e = 0; e < 2; ++e) o[e] = rstd * (gv[j][e] - mg - xv[j][e] * mgx);
      if (dres) o += *(const f32x2*)(dres + (size_t)row * kDim + 2 * (q + kLnLanes * j));
      *(f32x2*)(dr + 2 * (q + kLnLanes * j)) = o;
      if constexpr (SPLIT) {
        const int c = 2 * (q + kLnLanes * j);
        if (drop_thresh) {
#pragma unroll
          for (int e = 0; e < 2; ++e) o[e] = dropout_keep(drop_seed, (unsigned long long)row * kDim + c + e, drop_thresh) ? o[e] * drop_scale : 0.f;
        }
        acc_c[j] += o;
        __bf16 h0, l0, h1, l1;
        split_bf16(o[0], h0, l0);
        split_bf16(o[1], h1, l1);
        __bf16* d = split_out + (size_t)row * (2 * kDim) + split_index(c);
        *(bf16x2*)d = bf16x2{h0, h1};
        *(bf16x2*)(d + 32) = bf16x2{l0, l1};
      }
    }
    if (SPLIT && (it & 3) == 3) flush_cs(it >> 2);
  }
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 2 * (q + kLnLanes * j) + e;
      atomicAdd(&s_dg[c], acc_g[j][e]);    // kLnSlots row slots per column, once per block
      atomicAdd(&s_db[c], acc_b[j][e]);
    }
  __syncthreads();
  float* pr = partial + (size_t)blockIdx.x * 2 * kDim;
  for (int c = threadIdx.x; c < kDim; c += 256) { pr[c] = s_dg[c]; pr[kDim + c] = s_db[c]; }
  if constexpr (SPLIT) {
    float* cp = colp + (size_t)blockIdx.x * 2 * kDim;      // two rows of partials per block: rows 0..31, rows 32..63
    for (int c = threadIdx.x; c < kDim; c += 256) { cp[c] = s_cs[0][c]; cp[kDim + c] = s_cs[1][c]; }
  }
}

// out[c] = sum over `n_rows` rows of src[r][c], rows folded in order, in double: used for the LayerNorm partials and
// (through column_partial_kernel) for bias gradients
// 32 columns x 8 row groups per block: a thread folds its contiguous eighth of the rows in order, the eight partial sums
// meet in LDS in group order (fixed order -> deterministic; one thread per column over all rows was a 256-long chain of
// dependent loads, 62 us per call and 28 calls per training step)
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ src, long ld, int n_rows, int n_cols,
                                                        float* __restrict__ out) {
  __shared__ double s_part[8][32];
  const int cl = threadIdx.x & 31, g = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const int per = (n_rows + 7) / 8, r0 = g * per, r1 = r0 + per < n_rows ? r0 + per : n_rows;
  double acc = 0.0;
  if (c < n_cols)
    for (int r = r0; r < r1; ++r) acc += (double)src[(size_t)r * ld + c];
  s_part[g][cl] = acc;
  __syncthreads();
  if (g == 0 && c < n_cols) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += s_part[k][cl];
    out[c] = (float)t;
  }
}

// stage 1 of a column sum over many rows: block b adds rows [b*chunk, (b+1)*chunk) of dy[:, n_cols]
__global__ __launch_bounds__(256) void column_partial_kernel(const float* __restrict__ dy, long ld, int rows, int n_cols, int chunk,
                                                             float* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n_cols) return;
  const int r0 = blockIdx.y * chunk;
  const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc += dy[(size_t)r * ld + c];
  partial[(size_t)blockIdx.y * n_cols + c] = acc;
}

// dpre = dh * gelu'(pre), gelu'(x) = Phi(x) + x phi(x) with the exact erf (libm erff: this kernel is HBM-bound)
__global__ __launch_bounds__(256) void gelu_backward_kernel(const float* __restrict__ pre, const float* __restrict__ dh,
                                                            float* __restrict__ dpre, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n4; i += stride) {
    const f32x4 x = ((const f32x4*)pre)[i], g = ((const f32x4*)dh)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float cdf = 0.5f * (1.f + erff(x[e] * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x[e] * x[e]);
      o[e] = g[e] * (cdf + x[e] * pdf);
    }
    ((f32x4*)dpre)[i] = o;
  }
}

}  // namespace

// persistent grid = what the device holds at once (CUs x resident workgroups of this instantiation), found once
template <int DH>
static hipError_t launch_attention_backward_dh(const float* qkv, const float* dout, float* dqkv, __bf16* dqkv_split, int n_pair, int heads,
                                               int cls_only, hipStream_t s) {
  static int resident = 0;
  if (!resident) {
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, attention_backward_kernel<DH>, kAttnBwdThreads, 0);
    if (e != hipSuccess) return e;
    resident = cus * (per_cu > 0 ? per_cu : 1);
  }
  const long total = (long)n_pair * heads;
  const unsigned blocks = (unsigned)(total < resident ? total : resident);
  VETO_LAUNCH(attention_backward_kernel<DH>, dim3(blocks), dim3(kAttnBwdThreads), 0, s, qkv, dout, dqkv, dqkv_split, n_pair, heads, cls_only);
  return hipGetLastError();
}

hipError_t launch_attention_backward(const float* qkv, const float* dout, float* dqkv, __bf16* dqkv_split, int n_pair, int heads, int cls_only,
                                     hipStream_t s, bool qkv_f24) {
  if (heads <= 0 || kDim % heads != 0 || n_pair <= 0 || (!dqkv == !dqkv_split)) return hipErrorInvalidValue;
  const int dh = kDim / heads;
  if (qkv_f24 && dh != 72 && dh != 96) return hipErrorInvalidValue;      // (the MFMA form only)
  if (dh == 72 || dh == 96) {
    const unsigned blocks = (unsigned)(((long)n_pair * heads + 1) / 2);
    if (qkv_f24) {
      if (dh == 72) VETO_LAUNCH((attention_backward_mfma_kernel<72, true>), dim3(blocks), dim3(128), 0, s, qkv, dout, dqkv, dqkv_split, n_pair, heads, cls_only);
      else VETO_LAUNCH((attention_backward_mfma_kernel<96, true>), dim3(blocks), dim3(128), 0, s, qkv, dout, dqkv, dqkv_split, n_pair, heads, cls_only);
    } else if (dh == 72) VETO_LAUNCH(attention_backward_mfma_kernel<72>, dim3(blocks), dim3(128), 0, s, qkv, dout, dqkv, dqkv_split, n_pair, heads, cls_only);
    else VETO_LAUNCH(attention_backward_mfma_kernel<96>, dim3(blocks), dim3(128), 0, s, qkv, dout, dqkv, dqkv_split, n_pair, heads, cls_only);
    return hipGetLastError();
  }
  if (dh == 72) return launch_attention_backward_dh<72>(qkv, dout, dqkv, dqkv_split, n_pair, heads, cls_only, s);
  if (dh == 96) return launch_attention_backward_dh<96>(qkv, dout, dqkv, dqkv_split, n_pair, heads, cls_only, s);
  if (dh == 144) return launch_attention_backward_dh<144>(qkv, dout, dqkv, dqkv_split, n_pair, heads, cls_only, s);
  return hipErrorInvalidValue;
}

// partial rows of the LayerNorm parameter gradients + the scratch of the two-stage column sum over them
size_t layernorm_backward_partial_floats(int rows) {
  return (size_t)((rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock) * 2 * kDim + (size_t)kColChunks * 2 * kDim;
}

int layernorm_backward_col_partials(int rows) { return (rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock * 2; }

hipError_t launch_layernorm_backward(const float* x, const float* dy, const float* gamma, const float* dres, float* dx,
                                     float* dgamma_dbeta, float* partial, int rows, hipStream_t s, __bf16* split_out, float* colp,
                                     unsigned long long drop_seed, unsigned drop_thresh, float drop_scale) {
  const int blocks = (rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock;
  if (!split_out != !colp) return hipErrorInvalidValue;
  if (split_out) VETO_LAUNCH(layernorm_backward_kernel<true>, dim3(blocks), dim3(256), 0, s, x, dy, gamma, dres, dx, partial, rows, split_out, colp,
                             drop_seed, drop_thresh, drop_scale);
  else VETO_LAUNCH(layernorm_backward_kernel<false>, dim3(blocks), dim3(256), 0, s, x, dy, gamma, dres, dx, partial, rows, nullptr, nullptr, 0ull, 0u, 1.f);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_column_sums(partial, 2 * kDim, blocks, 2 * kDim, dgamma_dbeta, partial + (size_t)blocks * 2 * kDim, kColChunks, s);
}

int column_sums_chunks() { return kColChunks; }

hipError_t launch_column_sums(const float* dy, long ld, int rows, int n_cols, float* out, float* partial, int n_chunks, hipStream_t s) {
  const int chunk = (rows + n_chunks - 1) / n_chunks;
  VETO_LAUNCH(column_partial_kernel, dim3((n_cols + 255) / 256, n_chunks), dim3(256), 0, s, dy, ld, rows, n_cols, chunk, partial);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  VETO_LAUNCH(fold_rows_kernel, dim3((n_cols + 31) / 32), dim3(256), 0, s, partial, (long)n_cols, n_chunks, n_cols, out);
  return hipGetLastError();
}

hipError_t launch_gelu_backward(const float* pre, const float* dh, float* dpre, size_t n, hipStream_t s) {
  if (n % 4 != 0) return hipErrorInvalidValue;
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  VETO_LAUNCH(gelu_backward_kernel, dim3(blocks), dim3(256), 0, s, pre, dh, dpre, n4);
  return hipGetLastError();
}

}  // namespace veto
