// Memory-bound kernels of the VETO relation head: weight preparation, the per-object stage
// (box/class embeddings, patchify), pair-index build, pair gather + token assembly (+ fused
// LayerNorm of layer 0), stand-alone LayerNorm, and the classifier head.
//
// Reference semantics (all paths relative to pysgg/modeling/roi_heads/relation_head/):
//   pair indices      roi_relation_predictors.py:4104-4115, sampling.py:31-52
//   box embedding     roi_relation_predictors.py:4042-4047,4097-4102; model_mpv2.py:341-345;
//                     structures/bounding_box.py:60-78
//   class embedding   roi_relation_predictors.py:4086-4095
//   loc/class proj    roi_relation_predictors.py:4118-4121
//   patchify          model_veto.py:109-110
//   token assembly    model_veto.py:56-63
//   LayerNorm         model_veto.py:125-132 (nn.LayerNorm, eps 1e-5)
//   rel_out           roi_relation_predictors.py:4125
#include "common.h"
#include "kernels.h"

// token assembly: the fp32 token rows leave through non-temporal stores (they are next read by the layer tail, a whole attention launch later:
// 0.239 -> 0.215 ms, and the table attention behind it 0.60 -> 0.57 ms -- its per-object tables stay cached)
#ifndef ASM_NT
#define ASM_NT 1
#endif
namespace veto {

namespace {

// ------------------------------------------------------------------------------------------------
// weight preparation
// ------------------------------------------------------------------------------------------------
// fp32 [rows, K] -> split rows [rows, 2K] (common.h)
__global__ void split_rows_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n, int K) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const size_t row = i / K;
    const int k = (int)(i % K);
    __bf16 h, l;
    split_bf16(src[i], h, l);
    __bf16* d = dst + row * (2 * (size_t)K) + split_index(k);
    d[0] = h;
    d[32] = l;
  }
}

// fp32 [rows, K] -> mixed ACTIVATION rows (common.h); test hook of the mixed GEMM
__global__ void mixed_act_rows_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n, int K) {
  saturating_conversions_on();
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
    const size_t row = i / K;
    store_act4<FMT_MIXED>(dst + row * (2 * (size_t)K), (int)(i % K), *(const f32x4*)(src + i));
  }
}

// Mixed weight rows (common.h).  Pass 1: max |w| as float bits (non-negative floats order like unsigned integers).
__global__ void absmax_bits_kernel(const float* __restrict__ src, size_t n, unsigned* __restrict__ out) {
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned b = __float_as_uint(src[i]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
// e = the largest exponent in [0, 24] with 2^e * max|w| <= 448 (an all-zero tensor gets 24)
__device__ __forceinline__ int weight_exp_of(unsigned maxbits) {
  const float mx = __uint_as_float(maxbits);
  int e = 24;
  while (e > 0 && !(ldexpf(mx, e) <= 448.f)) --e;
  return e;
}
// Pass 2: 4 consecutive k's per thread: h = fp16(w) (8 B), { X = e4m3(2^e h), Y = e4m3(2^(e+11) (w - h)) } (8 B)
__global__ void mixed_weight_rows_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n, int K,
                                         const int* __restrict__ maxbits) {
  saturating_conversions_on();
  const int e = weight_exp_of((unsigned)*maxbits);
  const float sx = ldexpf(1.f, e), sy = ldexpf(1.f, e + kMixWLoShift);
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
    const size_t row = i / K;
    const int k = (int)(i % K);
    const f32x4 w = *(const f32x4*)(src + i);
    char* base = (char*)dst + row * (4 * (size_t)K);
    f16x4 h;
    float x[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = (_Float16)w[j];
      x[j] = (float)h[j] * sx;
      y[j] = (w[j] - (float)h[j]) * sy;
    }
    *(f16x4*)(base + mixed_h_offset(k)) = h;
    *(u32x2*)(base + mixed_x_offset(k)) = u32x2{pack_e4m3x4(x[0], x[1], x[2], x[3]), pack_e4m3x4(y[0], y[1], y[2], y[3])};
  }
}
// Pass 3: replace the float bits by the exponent itself (what the GEMM reads)
__global__ void finish_weight_exp_kernel(int* e) { *e = weight_exp_of((unsigned)*e); }

// fp32 [M, ld] -> split rows of the transpose [N, 2*Mp]: a 32 (m) x 64 (n) tile goes through LDS; every output
// row n receives one whole 32-k block, [8 x 4 hi | 8 x 4 lo] bf16 = 128 contiguous bytes, as 16-byte pieces.
__global__ __launch_bounds__(256) void transpose_split_kernel(const float* __restrict__ src, long ld, int M, int N,
                                                              __bf16* __restrict__ dst, int Mp) {
  __shared__ float t[32][65];
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 64, tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + ty * 8 + i, n = n0 + tx;
    t[ty * 8 + i][tx] = (m < M && n < N) ? src[(size_t)m * ld + n] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int id = tid + 256 * r, nl = id >> 3, piece = id & 7;
    if (n0 + nl >= N) continue;
    bf16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 h, l;
      split_bf16(t[8 * (piece & 3) + e][nl], h, l);
      out[e] = piece < 4 ? h : l;
    }
    __bf16* d = dst + (size_t)(n0 + nl) * (2 * (size_t)Mp) + (size_t)(m0 >> 5) * 64 + (piece < 4 ? 8 * piece : 32 + 8 * (piece - 4));
    *(bf16x8*)d = out;
  }
}

// proj(cat(subj, obj)) is linear, so it splits into a subject and an object partial product per
// OBJECT (restructuring (a)).  Row j of the combined weight is one column of
// the per-object table: [subj: depth 0..511 | rgb 512..575 | obj: depth 576..1087 | rgb 1088..1151];
// K = [depth patch features 0..1023 | rgb patch features 1024..2047], feature = (p1*2+p2)*256 + c.
// NB the crossed naming of the reference: proj_d (512 wide) acts on DEPTH, proj_v (64 wide) on RGB.
__global__ void build_patch_weight_kernel(const float* __restrict__ wd, const float* __restrict__ bd,
                                          const float* __restrict__ wv, const float* __restrict__ bv,
                                          __bf16* __restrict__ dst, float* __restrict__ bias_cat) {
  const int j = blockIdx.x;  // 0..1151
  const int half = j / kDim, jj = j % kDim;
  for (int kk = threadIdx.x; kk < 2048; kk += blockDim.x) {
    const int mod = kk >> 10, f = kk & 1023, pp = f >> 8, c = f & 255;
    float v = 0.f;
    if (jj < 512 && mod == 0) v = wd[(size_t)jj * 2048 + pp * 512 + half * 256 + c];
    if (jj >= 512 && mod == 1) v = wv[(size_t)(jj - 512) * 2048 + pp * 512 + half * 256 + c];
    __bf16 h, l;
    split_bf16(v, h, l);
    __bf16* d = dst + (size_t)j * 4096 + split_index(kk);
    d[0] = h;
    d[32] = l;
  }
  if (threadIdx.x == 0) bias_cat[j] = half == 0 ? (jj < 512 ? bd[jj] : bv[jj - 512]) : 0.f;
}

// Wcat^T as a GEMM weight operand: row kk (patch feature), column j (table column) = Wcat[j][kk] of the kernel above
__global__ void build_patch_weight_t_kernel(const float* __restrict__ wd, const float* __restrict__ wv, __bf16* __restrict__ dst) {
  const int kk = blockIdx.x;  // 0..2111
  const int mod = kk >> 10, f = kk & 1023, pp = f >> 8, c = f & 255;
  for (int j = threadIdx.x; j < 2 * kDim; j += blockDim.x) {
    const int half = j / kDim, jj = j % kDim;
    float v = 0.f;
    if (kk < 2048) {
      if (jj < 512 && mod == 0) v = wd[(size_t)jj * 2048 + pp * 512 + half * 256 + c];
      if (jj >= 512 && mod == 1) v = wv[(size_t)(jj - 512) * 2048 + pp * 512 + half * 256 + c];
    }
    __bf16 h, l;
    split_bf16(v, h, l);
    __bf16* d = dst + (size_t)kk * (2 * 2 * kDim) + split_index(j);
    d[0] = h;
    d[32] = l;
  }
}

__global__ void transpose_pair_proj_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                           int kin) {
  const int k = blockIdx.x;  // 0..kin-1
  for (int jj = threadIdx.x; jj < 2 * kDim; jj += blockDim.x) {
    const int half = jj / kDim, j = jj % kDim;
    dst[(size_t)k * (2 * kDim) + jj] = src[(size_t)j * (2 * kin) + half * kin + k];
  }
}

__global__ void transpose_head_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_out) {
  const int k = blockIdx.x;  // 0..575
  for (int c = threadIdx.x; c < n_out; c += blockDim.x) dst[(size_t)k * n_out + c] = src[(size_t)c * kDim + k];
}

// ------------------------------------------------------------------------------------------------
// per-object stage
// ------------------------------------------------------------------------------------------------
// Batch statistics of the four box features (centre x, centre y, w, h; same formula as obj_prep_kernel) for
// BatchNorm1d(4) in training mode: one workgroup, two passes (mean, then centred squares) in double precision.
__global__ __launch_bounds__(256) void bn_batch_stats_kernel(const float* __restrict__ boxes, int box_mode, int n_obj,
                                                             float* __restrict__ out) {
  __shared__ double s_acc[256][4];
  __shared__ double s_mean[4];
  const int tid = threadIdx.x;
  auto feat = [&](int n, double (&v)[4]) {
    const float* b = boxes + (size_t)n * 4;
    float w, h;
    if (box_mode == 0) { w = b[2] - b[0] + 1.f; h = b[3] - b[1] + 1.f; } else { w = b[2]; h = b[3]; }
    v[0] = b[0] + 0.5f * w; v[1] = b[1] + 0.5f * h; v[2] = w; v[3] = h;
  };
  for (int pass = 0; pass < 2; ++pass) {
    double acc[4] = {0, 0, 0, 0};
    for (int n = tid; n < n_obj; n += 256) {
      double v[4];
      feat(n, v);
      for (int k = 0; k < 4; ++k) { const double d = pass == 0 ? v[k] : v[k] - s_mean[k]; acc[k] += pass == 0 ? d : d * d; }
    }
    for (int k = 0; k < 4; ++k) s_acc[tid][k] = acc[k];
    __syncthreads();
    if (tid < 4) {
      double t = 0;
      for (int i = 0; i < 256; ++i) t += s_acc[i][tid];
      if (pass == 0) {
        s_mean[tid] = t / n_obj;
        out[tid] = (float)s_mean[tid];
      } else {
        out[4 + tid] = (float)(t / n_obj);
        out[8 + tid] = (float)(n_obj > 1 ? t / (n_obj - 1) : t);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void obj_prep_kernel(ObjPrepArgs a) {
  __shared__ float s_box[4];
  __shared__ float s_pos[kPosDim];
  __shared__ float s_emb[256];
  __shared__ float s_prob[256];
  __shared__ float s_red[2];
  const int n = blockIdx.x, tid = threadIdx.x;

  if (tid < 4) {
    const float* b = a.boxes + (size_t)n * 4;
    float w, h;
    if (a.box_mode == 0) { w = b[2] - b[0] + 1.f; h = b[3] - b[1] + 1.f; }  // xyxy -> xywh, +1 convention
    else { w = b[2]; h = b[3]; }
    float v = tid == 0 ? b[0] + 0.5f * w : tid == 1 ? b[1] + 0.5f * h : tid == 2 ? w : h;
    // BatchNorm1d(4) in eval mode: running statistics, eps 1e-5
    s_box[tid] = (v - a.bn_mean[tid]) / sqrtf(a.bn_var[tid] + 1e-5f) * a.bn_w[tid] + a.bn_b[tid];
  }
  if (a.obj_logits) {
    for (int c = tid; c < a.num_obj_cls; c += 256) s_prob[c] = a.obj_logits[(size_t)n * a.num_obj_cls + c];
  }
  __syncthreads();
  if (tid < kPosDim) {
    float acc = a.pos_b[tid];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += a.pos_w[tid * 4 + k] * s_box[k];
    s_pos[tid] = fmaxf(acc, 0.f);
    if (a.drop_thresh)   // training: Dropout(0.1) of pos_embed (roi_relation_predictors.py:4042-4047)
      s_pos[tid] = dropout_keep(a.drop_seed, (unsigned long long)n * kPosDim + tid, a.drop_thresh) ? s_pos[tid] * a.drop_scale : 0.f;
    if (a.pos_out && blockIdx.y == 0) a.pos_out[(size_t)n * kPosDim + tid] = s_pos[tid];
  }
  if (a.obj_logits) {
    if (tid == 0) {
      float mx = -INFINITY;
      for (int c = 0; c < a.num_obj_cls; ++c) mx = fmaxf(mx, s_prob[c]);
      float sum = 0.f;
      for (int c = 0; c < a.num_obj_cls; ++c) sum += expf(s_prob[c] - mx);
      s_red[0] = mx;
      s_red[1] = sum;
    }
    __syncthreads();
    const float mx = s_red[0], inv = 1.f / s_red[1];
    __syncthreads();
    for (int c = tid; c < a.num_obj_cls; c += 256) s_prob[c] = expf(s_prob[c] - mx) * inv;
    __syncthreads();
    for (int e = tid; e < a.embed_dim; e += 256) {
      float acc = 0.f;
      for (int c = 0; c < a.num_obj_cls; ++c) acc += s_prob[c] * a.embed[(size_t)c * a.embed_dim + e];
      s_emb[e] = acc;
    }
  } else {
    const int64_t lab = a.labels[n];
    for (int e = tid; e < a.embed_dim; e += 256) s_emb[e] = a.embed[(size_t)lab * a.embed_dim + e];
  }
  __syncthreads();
  // blockIdx.y selects which 256 of the 1152 output columns this workgroup produces (the cheap box /
  // class embedding above is recomputed per workgroup): 5x more workgroups hide the k-loop latency.
  float* out = a.lc + (size_t)n * 2 * (2 * kDim);
  const int jj = blockIdx.y * 256 + tid;
  if (jj < 2 * kDim) {
    // (both loops are chains of one L2 load + one FMA per k: unrolled, so that 16 weight loads are in flight per thread -- the
    // kernel is pure latency: 53 -> 2x us for 432 objects.  Same order of additions.)
    float acc = jj < kDim ? a.loc_b[jj] : 0.f;
#pragma unroll 16
    for (int k = 0; k < kPosDim; ++k) acc += a.loc_wt[(size_t)k * (2 * kDim) + jj] * s_pos[k];
    out[jj] = acc;
    float acc2 = jj < kDim ? a.cls_b[jj] : 0.f;
#pragma unroll 8
    for (int k = 0; k < a.embed_dim; ++k) acc2 += a.cls_wt[(size_t)k * (2 * kDim) + jj] * s_emb[k];
    out[2 * kDim + jj] = acc2;
  }
}

// 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)', p = 2, 8x8 maps, one thread per channel.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ depth,
                                                       const float* __restrict__ rgb,
                                                       __bf16* __restrict__ dst) {
  const int n = blockIdx.x, mod = blockIdx.y, c = threadIdx.x;
  const float* in = (mod == 0 ? depth : rgb) + ((size_t)n * 256 + c) * 64;
  float v[64];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const f32x4 t = *(const f32x4*)(in + 4 * i);
    v[4 * i] = t[0]; v[4 * i + 1] = t[1]; v[4 * i + 2] = t[2]; v[4 * i + 3] = t[3];
  }
#pragma unroll
  for (int y = 0; y < 8; ++y)
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int row = n * 16 + (y >> 1) * 4 + (x >> 1);
      const int col = mod * 1024 + ((y & 1) * 2 + (x & 1)) * 256 + c;
      __bf16 h, l;
      split_bf16(v[y * 8 + x], h, l);
      __bf16* d = dst + (size_t)row * 4096 + split_index(col);
      d[0] = h;
      d[32] = l;
    }
}

// the inverse index map of patchify_kernel, on fp32 gradients: thread c of block (n, modality) gathers the 64 pixels of
// channel c from the 16 patch rows of object n (every pixel receives exactly one contribution: patches do not overlap)
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ dpa, long ld, float* __restrict__ d_depth,
                                                         float* __restrict__ d_rgb) {
  const int n = blockIdx.x, mod = blockIdx.y, c = threadIdx.x;
  float* out = (mod == 0 ? d_depth : d_rgb);
  if (!out) return;
  out += ((size_t)n * 256 + c) * 64;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    f32x4 lo, hi;
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int row = n * 16 + (y >> 1) * 4 + (x >> 1);
      const int col = mod * 1024 + ((y & 1) * 2 + (x & 1)) * 256 + c;
      const float v = dpa[(size_t)row * ld + col];
      if (x < 4) lo[x] = v; else hi[x - 4] = v;
    }
    *(f32x4*)(out + y * 8) = lo;
    *(f32x4*)(out + y * 8 + 4) = hi;
  }
}

// ------------------------------------------------------------------------------------------------
// pair stage
// ------------------------------------------------------------------------------------------------
__global__ void pair_indices_kernel(const int64_t* __restrict__ rel_pairs,
                                    const int32_t* __restrict__ img_obj_off,
                                    const int32_t* __restrict__ img_pair_off, int n_img, int n_pair,
                                    int32_t* __restrict__ subj, int32_t* __restrict__ obj,
                                    int64_t* __restrict__ subj64, int64_t* __restrict__ obj64) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pair) return;
  int lo = 0, hi = n_img - 1;  // largest i with img_pair_off[i] <= p
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (img_pair_off[mid] <= p) lo = mid; else hi = mid - 1;
  }
  const int64_t off = img_obj_off[lo];
  const int64_t s = rel_pairs[2 * (size_t)p] + off, o = rel_pairs[2 * (size_t)p + 1] + off;
  subj[p] = (int32_t)s;
  obj[p] = (int32_t)o;
  if (subj64) subj64[p] = s;
  if (obj64) obj64[p] = o;
}

// nonzero(ones(n,n) - eye(n)) in row-major order; [[0,0]] placeholder when empty.
__global__ void enumerate_pairs_kernel(int n, int64_t* __restrict__ out) {
  const long total = n > 1 ? (long)n * (n - 1) : 1;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  if (n <= 1) { out[0] = 0; out[1] = 0; return; }
  const long i = p / (n - 1), r = p % (n - 1);
  out[2 * p] = i;
  out[2 * p + 1] = r + (r >= i ? 1 : 0);
}

// A 576-wide row is handled by a QUARTER wave: lane q (0..15) of the group holds the float4 chunks
// q + 16 j, j = 0..8 (16-byte global accesses, 256 contiguous bytes per group and j).  A wave works
// on 4 rows, a 256-thread block on 16.
struct RowQ { f32x4 v[9]; };

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LayerNorm (eps 1e-5, model_veto.py:125-132) of the row in registers -> split-row operand
__device__ __forceinline__ void rowq_stats(const RowQ& r, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 9; ++j) s += (r.v[j][0] + r.v[j][1]) + (r.v[j][2] + r.v[j][3]);
  mean = group16_sum(s) * (1.f / kDim);
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = r.v[j][e] - mean; sq += d * d; }
  rstd = 1.f / sqrtf(group16_sum(sq) * (1.f / kDim) + 1e-5f);
}

template <int FMT = FMT_SPLIT>
__device__ __forceinline__ void rowq_normalized_store(const RowQ& r, int q, float mean, float rstd, const float* __restrict__ w,
                                                      const float* __restrict__ b, __bf16* __restrict__ dst) {
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int c = 4 * (q + 16 * j);
    const f32x4 wv = *(const f32x4*)(w + c), bv = *(const f32x4*)(b + c);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (r.v[j][e] - mean) * rstd * wv[e] + bv[e];
    store_act4<FMT>(dst, c, y);  // 4 consecutive columns stay inside one 32-k block
  }
}

template <int FMT = FMT_SPLIT>
__device__ __forceinline__ void rowq_layernorm_store(const RowQ& r, int q, const float* __restrict__ w,
                                                     const float* __restrict__ b, __bf16* __restrict__ dst) {
  float mean, rstd;
  rowq_stats(r, mean, rstd);
  rowq_normalized_store<FMT>(r, q, mean, rstd, w, b, dst);
}

// Token assembly (model_veto.py:56-63 with the per-object partial products of section 4 of DESIGN.md):
// one quarter wave per token row.  Reads of the per-object tables hit L2 / Infinity Cache (each object
// row is re-used by 2(N-1) pairs); the writes (x fp32 + LN(x) split) are the HBM stream.
// STATS_ONLY (layer 0 in the per-object form): the kernel writes the token rows and their LayerNorm statistics only; the
// LayerNorm'ed split rows of tokens 17 / 18 come from a strided layernorm_kernel launch behind it.  Keeping that store path
// (gamma / beta of nine chunks, the split conversion) out of this instantiation takes it from 138 to ~70 registers, i.e.
// from 3 to 7 waves per SIMD: the kernel is a latency-bound gather (measured 2.3 TB/s of writes at 3 waves per SIMD).
template <bool STATS_ONLY>
__global__ __launch_bounds__(256) void assemble_kernel(AssembleArgs a) {
  // Workgroups are dealt round-robin to the 8 XCDs; XCD x takes the x-th contiguous eighth of the token rows, i.e. the pairs of
  // one or two images, whose per-object rows (2.65 MB per 36-object image) then stay in THAT XCD's 4 MB L2.
  const long per_xcd = gridDim.x >> 3;
  const long row = ((long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * 16 + (threadIdx.x >> 4);
  if (row >= (long)a.n_pair * kTokens) return;
  const int q = threadIdx.x & 15;
  const int p = (int)(row / kTokens), t = (int)(row % kTokens);
  const int s = a.subj[p], o = a.obj[p];
  const float *ps = nullptr, *po = nullptr;
  if (t >= 1 && t <= kPatchTokens) {
    ps = a.patch_tab + ((size_t)s * 16 + (t - 1)) * (2 * kDim);
    po = a.patch_tab + ((size_t)o * 16 + (t - 1)) * (2 * kDim) + kDim;
  } else if (t > kPatchTokens) {
    const int which = t - kPatchTokens - 1;  // 0 location, 1 class
    ps = a.lc + ((size_t)s * 2 + which) * (2 * kDim);
    po = a.lc + ((size_t)o * 2 + which) * (2 * kDim) + kDim;
  }
  RowQ r;
  float* xr = a.x + (size_t)row * kDim;
  char* xr24 = (char*)a.x + (size_t)row * (kDim * 3);      // x_f24: rows of 3-byte floats
  // Three groups of three 16-byte chunks: with all nine chunks of the three sources in flight at once the kernel needed 138
  // registers (3 waves per SIMD) and was bound by the latency of its gathers (0.32 ms for 0.73 GB written = 2.3 TB/s, against
  // 5.5 TB/s for plain store streams, profiles/r02_store_bw.txt); a group keeps 9 loads in flight per lane and leaves the
  // overlap to the other waves of the SIMD.
#pragma unroll
  for (int g3 = 0; g3 < 3; ++g3) {
    f32x4 vs[3], vo[3], vp[3];
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int c = 4 * (q + 16 * (g3 * 3 + jj));
      vp[jj] = *(const f32x4*)(a.pos_embedding + c);
      if (t == 0) {
        vs[jj] = *(const f32x4*)(a.cls_token + c);
        vo[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        vs[jj] = *(const f32x4*)(ps + c);
        vo[jj] = *(const f32x4*)(po + c);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
      const int j = g3 * 3 + jj, c = 4 * (q + 16 * j);
      f32x4 v = t == 0 ? vs[jj] : vs[jj] + vo[jj];
      if (t > kPatchTokens) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);  // ReLU of location / class projection
      }
      v += vp[jj];
      if (a.drop_thresh) {   // training: pos_drop (model_veto.py:63)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = dropout_keep(a.drop_seed, (unsigned long long)row * kDim + c + e, a.drop_thresh) ? v[e] * a.drop_scale : 0.f;
      }
      r.v[j] = v;
      if (a.x_f24) {      // (the statistics below are those of the fp32 row: the tables of layer 0 are normalised with them)
#if ASM_NT
        __builtin_nontemporal_store(pack_f24x4(v), (u32x3*)(xr24 + 3 * c));
#else
        *(u32x3*)(xr24 + 3 * c) = pack_f24x4(v);
#endif
        continue;
      }
#if ASM_NT
      __builtin_nontemporal_store(v, (f32x4*)(xr + c));
#else
      *(f32x4*)(xr + c) = v;
#endif
    }
    asm volatile("" ::: "memory");   // keeps the next group's loads behind this group's stores (bounds the registers)
  }
  if constexpr (STATS_ONLY) {
    // layer 0 in the per-object form (qkv0_combine_kernel): the patch-token rows and the CLS row only need their LayerNorm
    // statistics; the two ReLU'd rows (location, class) are not linear in the per-object tables and keep the split-row path
    // (written by the layernorm launch behind this kernel)
    float mean, rstd;
    rowq_stats(r, mean, rstd);
    if (q == 0) *(float2*)(a.stats + (size_t)row * 2) = float2{mean, rstd};
  } else {
    rowq_layernorm_store(r, q, a.ln_w, a.ln_b, a.a + (size_t)row * (2 * kDim));
  }
}

// ---- layer 0, per-object form of LayerNorm + QKV ---------------------------------------------
// x = S[s, t] + O[o, t] + pos for the 16 patch tokens, and the row mean of a sum is the sum of the row means, so with
// W' = Wqkv diag(gamma) and every term centred on its own row mean (Sc = S - mean(S) etc.):
//   LN(x) Wqkv^T = rstd (x - mean(x)) W'^T + c2 = rstd (SW[s, t] + OW[o, t]) + c2,
//   SW = Sc W'^T + b0 (b0 = W' (pos - mean(pos))),  OW = Oc W'^T,  c2 = Wqkv beta
// SW / OW are per-OBJECT tables ([n_obj*16, 1728], two small GEMMs over centred split rows, centre_split_kernel), c2, b0 and
// the constant CLS row are weight-only vectors (vec = [c2 | b0 | qkv_cls]); only rstd depends on the pair.  The consumer is
// the layer-0 attention itself (attention.hip, TAB) or, for head widths without an MFMA attention, qkv0_combine_kernel, which
// materialises the qkv rows of tokens 0..16: one wave per row.  The rows of tokens 17 / 18 (ReLU'd, not linear in the tables)
// are written by two GEMMs over their LayerNorm'ed split rows.
__global__ __launch_bounds__(256) void qkv0_combine_kernel(const float* __restrict__ sw, const float* __restrict__ ow,
                                                           const float* __restrict__ stats, const float* __restrict__ vec,
                                                           const int32_t* __restrict__ subj, const int32_t* __restrict__ obj,
                                                           float* __restrict__ qkv, int n_pair) {
  constexpr int N = 3 * kDim, N4 = N / 4;
  __shared__ __attribute__((aligned(16))) float s_vec[N];
  for (int i = threadIdx.x; i < N4; i += 256) ((f32x4*)s_vec)[i] = ((const f32x4*)vec)[i];
  __syncthreads();
  const f32x4* c2 = (const f32x4*)s_vec;
  const int lane = threadIdx.x & 63;
  // Workgroups are dealt round-robin to the 8 XCDs: XCD x walks the x-th contiguous eighth of the rows (one or two images), its
  // workgroups side by side, so that the per-object rows of those images (8 MB per 36-object image) are fetched into ONE L2
  const long rows = (long)n_pair * (kPatchTokens + 1), per_xcd = (rows + 7) / 8;
  const long r_begin = (blockIdx.x & 7) * per_xcd, r_end = r_begin + per_xcd < rows ? r_begin + per_xcd : rows;
  const long stride = (long)(gridDim.x >> 3) * 4;
  for (long r = r_begin + (long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6); r < r_end; r += stride) {
    const int p = (int)(r / (kPatchTokens + 1)), t = (int)(r % (kPatchTokens + 1));
    f32x4* dst = (f32x4*)(qkv + ((size_t)p * kTokens + t) * N);
    if (t == 0) {
      const f32x4* src = (const f32x4*)(vec + 2 * N);
      for (int i = lane; i < N4; i += 64) dst[i] = src[i];
      continue;
    }
    const float rstd = stats[((size_t)p * kTokens + t) * 2 + 1];
    const f32x4* a = (const f32x4*)(sw + ((size_t)subj[p] * kPatchTokens + t - 1) * N);
    const f32x4* b = (const f32x4*)(ow + ((size_t)obj[p] * kPatchTokens + t - 1) * N);
#pragma unroll
    for (int k = 0; k < (N4 + 63) / 64; ++k) {
      const int i = lane + 64 * k;
      if (i < N4) dst[i] = rstd * (a[i] + b[i]) + c2[i];
    }
  }
}

// patch_tab fp32 [rows, 1152] (subject half | object half) -> split rows [rows, 2*1152] of the halves, each centred on its own
// row mean: the A operands of the two table GEMMs.  A quarter wave per half row.
__global__ __launch_bounds__(256) void centre_split_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int rows) {
  const long hr = (long)blockIdx.x * 16 + (threadIdx.x >> 4);     // half-row index
  if (hr >= 2L * rows) return;
  const int q = threadIdx.x & 15;
  const float* xr = src + (size_t)hr * kDim;                       // row hr / 2, half hr & 1: contiguous in [rows, 1152]
  RowQ r;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    r.v[j] = *(const f32x4*)(xr + 4 * (q + 16 * j));
    s += (r.v[j][0] + r.v[j][1]) + (r.v[j][2] + r.v[j][3]);
  }
  const float mean = group16_sum(s) * (1.f / kDim);
  __bf16* d0 = dst + (size_t)hr * (2 * kDim);                      // split row of [rows, 2*1152]: half h starts at element h * 1152
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int c = 4 * (q + 16 * j);
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16 h, l;
      split_bf16(r.v[j][e] - mean, h, l);
      hi[e] = h;
      lo[e] = l;
    }
    __bf16* d = d0 + split_index(c);
    *(bf16x4*)d = hi;
    *(bf16x4*)(d + 32) = lo;
  }
}

// weight-only vectors of the per-object layer-0 form: vec = [c2 | b0 | qkv_cls], each [1728]; wq = Wqkv [1728, 576], wp = Wqkv diag(gamma)
__global__ __launch_bounds__(64) void qkv0_consts_kernel(const float* __restrict__ wq, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ pos,
                                                         const float* __restrict__ cls, float* __restrict__ wp, float* __restrict__ vec) {
  const int n = blockIdx.x, lane = threadIdx.x, N = 3 * kDim;
  // row mean of pos_embedding, and the LayerNorm statistics of the constant CLS row x = cls_token + pos_embedding (every block
  // recomputes them: 576 values)
  double s = 0.0, sp = 0.0, sq = 0.0;
  for (int k = lane; k < kDim; k += 64) { s += (double)(cls[k] + pos[k]); sp += (double)pos[k]; }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); sp += __shfl_xor(sp, o, 64); }
  const double mean = s / kDim, mean_pos = sp / kDim;
  for (int k = lane; k < kDim; k += 64) { const double d = (double)(cls[k] + pos[k]) - mean; sq += d * d; }
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const double rstd = 1.0 / sqrt(sq / kDim + 1e-5);
  double c2 = 0.0, b0 = 0.0, qc = 0.0;
  for (int k = lane; k < kDim; k += 64) {
    const float w = wq[(size_t)n * kDim + k], g = w * gamma[k];
    wp[(size_t)n * kDim + k] = g;
    c2 += (double)w * (double)beta[k];
    b0 += (double)g * ((double)pos[k] - mean_pos);
    qc += (double)g * ((double)(cls[k] + pos[k]) - mean);
  }
  for (int o = 32; o > 0; o >>= 1) {
    c2 += __shfl_xor(c2, o, 64);
    b0 += __shfl_xor(b0, o, 64);
    qc += __shfl_xor(qc, o, 64);
  }
  if (lane == 0) {
    vec[n] = (float)c2;
    vec[N + n] = (float)b0;
    vec[2 * N + n] = (float)(rstd * qc + c2);
  }
}

// XF24: the rows are 3-byte floats (common.h), ldx counts ELEMENTS all the same (a row starts at byte 3 * row * ldx)
template <int FMT, bool XF24 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        __bf16* __restrict__ dst, int rows, long ldd) {
  saturating_conversions_on();
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= rows) return;
  const int q = threadIdx.x & 15;
  RowQ r;
  if constexpr (XF24) {
    const char* xr = (const char*)x + (size_t)row * ldx * 3;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const u32x3 d = *(const u32x3*)(xr + 12 * (q + 16 * j));
      r.v[j] = unpack_f24x4(d[0], d[1], d[2]);
    }
  } else {
    const float* xr = x + (size_t)row * ldx;
#pragma unroll
    for (int j = 0; j < 9; ++j) r.v[j] = *(const f32x4*)(xr + 4 * (q + 16 * j));
  }
  rowq_layernorm_store<FMT>(r, q, w, b, dst + (size_t)row * ldd);
}

// 3-byte floats -> fp32 (debug outputs of a path that keeps its residual stream in 3-byte floats); n = number of values, a multiple of 4
__global__ __launch_bounds__(256) void unpack_f24_kernel(const char* __restrict__ src, float* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const u32x3 d = *(const u32x3*)(src + i * 12);
    *(f32x4*)(dst + i * 4) = unpack_f24x4(d[0], d[1], d[2]);
  }
}

__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                            unsigned long long seed, unsigned thresh, float scale) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) y[i] = dropout_keep(seed, i, thresh) ? x[i] * scale : 0.f;
}

// logits[p][c] = cls[p] . W[c] + b[c]; weights pre-transposed to [576][n_out]; 4 pairs per block.
__global__ __launch_bounds__(128) void head_kernel(const float* __restrict__ cls,
                                                   const float* __restrict__ wt,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   int n_pair, int n_out, long ld) {
  __shared__ float s_cls[4][kDim];
  const int p0 = blockIdx.x * 4;
  for (int i = threadIdx.x; i < 4 * kDim; i += 128) {
    const int pp = i / kDim, k = i % kDim;
    s_cls[pp][k] = p0 + pp < n_pair ? cls[(size_t)(p0 + pp) * ld + k] : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n_out; c += 128) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < kDim; ++k) {
      const float wv = wt[(size_t)k * n_out + c];
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) acc[pp] += wv * s_cls[pp][k];
    }
    const float bv = bias[c];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp)
      if (p0 + pp < n_pair) out[(size_t)(p0 + pp) * n_out + c] = acc[pp] + bv;
  }
}

}  // namespace

hipError_t launch_split_rows(const float* src, __bf16* dst, size_t rows, int K, hipStream_t s) {
  if (K % 32 != 0) return hipErrorInvalidValue;
  const size_t n = rows * (size_t)K;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  VETO_LAUNCH(split_rows_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n, K);
  return hipGetLastError();
}

// Weights of the folded last layer in their block form (veto_abi.hip): `which` = 0 Wq with every head's dh rows padded to dhp
// [H dhp, 576]; 1 the per-head transposes of Wk, block-diagonal [H 576, H dhp] (row (h, c), column (h, d) = Wk[h dh + d][c]);
// 2 Wv, block-diagonal [H dhp, H 576] (row (h, d), column (h, c) = Wv[h dh + d][c]); 3 Wo with every head's dh columns padded
// [576, H dhp].  qkv = the layer's to_qkv weight [1728, 576] (q rows, k rows, v rows), wo = to_out weight [576, 576].  fp32 out.
namespace {
__global__ __launch_bounds__(256) void fold_blocks_kernel(const float* __restrict__ qkv, const float* __restrict__ wo, float* __restrict__ out,
                                                          int which, int H, int dh, int dhp, size_t n) {
  const int np = H * dhp, hk = H * kDim;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    float v = 0.f;
    if (which == 0) {
      const int r = (int)(e / kDim), c = (int)(e % kDim), h = r / dhp, d = r % dhp;
      if (d < dh) v = qkv[(size_t)(h * dh + d) * kDim + c];
    } else if (which == 1) {
      const int r = (int)(e / np), k = (int)(e % np), h = r / kDim, c = r % kDim, hk2 = k / dhp, d = k % dhp;
      if (hk2 == h && d < dh) v = qkv[(size_t)(kDim + h * dh + d) * kDim + c];
    } else if (which == 2) {
      const int r = (int)(e / hk), k = (int)(e % hk), h = r / dhp, d = r % dhp, hk2 = k / kDim, c = k % kDim;
      if (hk2 == h && d < dh) v = qkv[(size_t)(2 * kDim + h * dh + d) * kDim + c];
    } else {
      const int r = (int)(e / np), k = (int)(e % np), h = k / dhp, d = k % dhp;
      if (d < dh) v = wo[(size_t)r * kDim + h * dh + d];
    }
    out[e] = v;
  }
}
}  // namespace

hipError_t launch_fold_blocks(const float* qkv, const float* wo, float* out, int which, int heads, int dhp, hipStream_t s) {
  if (heads <= 0 || kDim % heads != 0 || dhp < kDim / heads || which < 0 || which > 3) return hipErrorInvalidValue;
  const int dh = kDim / heads, np = heads * dhp;
  const size_t n = which == 0 ? (size_t)np * kDim : which == 1 ? (size_t)heads * kDim * np : which == 2 ? (size_t)np * heads * kDim : (size_t)kDim * np;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  VETO_LAUNCH(fold_blocks_kernel, dim3(blocks), dim3(256), 0, s, qkv, wo, out, which, heads, dh, dhp, n);
  return hipGetLastError();
}

hipError_t launch_build_patch_weight_t(const float* wd, const float* wv, __bf16* dst, hipStream_t s) {
  VETO_LAUNCH(build_patch_weight_t_kernel, dim3(kPatchTRows), dim3(256), 0, s, wd, wv, dst);
  return hipGetLastError();
}

hipError_t launch_unpatchify(const float* dpa, long ld, float* d_depth, float* d_rgb, int n_obj, hipStream_t s) {
  VETO_LAUNCH(unpatchify_kernel, dim3(n_obj, 2), dim3(256), 0, s, dpa, ld, d_depth, d_rgb);
  return hipGetLastError();
}

hipError_t launch_transpose_split(const float* src, long ld, int M, int N, __bf16* dst, int Mp, hipStream_t s) {
  if (Mp % 32 != 0 || Mp < M) return hipErrorInvalidValue;
  VETO_LAUNCH(transpose_split_kernel, dim3(Mp / 32, (N + 63) / 64), dim3(256), 0, s, src, ld, M, N, dst, Mp);
  return hipGetLastError();
}

hipError_t launch_build_patch_weight(const float* wd, const float* bd, const float* wv, const float* bv,
                                     __bf16* dst, float* bias_cat, hipStream_t s) {
  VETO_LAUNCH(build_patch_weight_kernel, dim3(2 * kDim), dim3(256), 0, s, wd, bd, wv, bv, dst, bias_cat);
  return hipGetLastError();
}

hipError_t launch_transpose_pair_proj(const float* src, float* dst, int kin, hipStream_t s) {
  VETO_LAUNCH(transpose_pair_proj_kernel, dim3(kin), dim3(256), 0, s, src, dst, kin);
  return hipGetLastError();
}

hipError_t launch_transpose_head(const float* src, float* dst, int n_out, hipStream_t s) {
  VETO_LAUNCH(transpose_head_kernel, dim3(kDim), dim3(128), 0, s, src, dst, n_out);
  return hipGetLastError();
}

hipError_t launch_dropout_apply(const float* x, float* y, size_t rows, int n_cols, unsigned long long seed, unsigned thresh,
                                float scale, hipStream_t s) {
  const size_t n = rows * (size_t)n_cols;
  const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  VETO_LAUNCH(dropout_apply_kernel, dim3(blocks), dim3(256), 0, s, x, y, n, seed, thresh, scale);
  return hipGetLastError();
}

hipError_t launch_bn_batch_stats(const float* boxes, int box_mode, int n_obj, float* out, hipStream_t s) {
  VETO_LAUNCH(bn_batch_stats_kernel, dim3(1), dim3(256), 0, s, boxes, box_mode, n_obj, out);
  return hipGetLastError();
}

hipError_t launch_obj_prep(const ObjPrepArgs& a, hipStream_t s) {
  if (a.num_obj_cls > 256 || a.embed_dim > 256) return hipErrorInvalidValue;
  VETO_LAUNCH(obj_prep_kernel, dim3(a.n_obj, (2 * kDim + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_patchify(const float* depth, const float* rgb, __bf16* dst, int n_obj, hipStream_t s) {
  VETO_LAUNCH(patchify_kernel, dim3(n_obj, 2), dim3(256), 0, s, depth, rgb, dst);
  return hipGetLastError();
}

hipError_t launch_pair_indices(const int64_t* rel_pairs, const int32_t* img_obj_off,
                               const int32_t* img_pair_off, int n_img, int n_pair, int32_t* subj,
                               int32_t* obj, int64_t* subj64, int64_t* obj64, hipStream_t s) {
  VETO_LAUNCH(pair_indices_kernel, dim3((n_pair + 255) / 256), dim3(256), 0, s, rel_pairs,
                     img_obj_off, img_pair_off, n_img, n_pair, subj, obj, subj64, obj64);
  return hipGetLastError();
}

hipError_t launch_enumerate_pairs(int n, int64_t* out, hipStream_t s) {
  const long total = n > 1 ? (long)n * (n - 1) : 1;
  VETO_LAUNCH(enumerate_pairs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n, out);
  return hipGetLastError();
}

hipError_t launch_assemble(const AssembleArgs& a, hipStream_t s) {
  const long blocks = ((long)a.n_pair * kTokens + 15) / 16;
  if (a.stats) {
    VETO_LAUNCH(assemble_kernel<true>, dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(256), 0, s, a);
    // LayerNorm'ed split rows of the location / class tokens (rows 19 p + 17, 19 p + 18): 2 of 19 rows, read back from x
    for (int t = kPatchTokens + 1; t < kTokens; ++t) {
      const float* xt = a.x_f24 ? (const float*)((const char*)a.x + (size_t)t * kDim * 3) : a.x + (size_t)t * kDim;
      hipError_t e = launch_layernorm(xt, (long)kTokens * kDim, a.ln_w, a.ln_b, a.a + (size_t)t * 2 * kDim, a.n_pair, s,
                                      a.a_fmt == FMT_MIXED ? FMT_MIXED : FMT_SPLIT, (long)kTokens * 2 * kDim, a.x_f24 != 0);
      if (e != hipSuccess) return e;
    }
  } else {
    if (a.x_f24) return hipErrorInvalidValue;      // (3-byte token rows: the per-object form of layer 0 only)
    VETO_LAUNCH(assemble_kernel<false>, dim3((unsigned)((blocks + 7) / 8 * 8)), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

hipError_t launch_unpack_f24(const void* src, float* dst, size_t n, hipStream_t s) {
  if (n % 4 != 0) return hipErrorInvalidValue;
  const size_t n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  VETO_LAUNCH(unpack_f24_kernel, dim3(blocks), dim3(256), 0, s, (const char*)src, dst, n4);
  return hipGetLastError();
}

hipError_t launch_qkv0_combine(const float* sw, const float* ow, const float* stats, const float* vec, const int32_t* subj,
                               const int32_t* obj, float* qkv, int n_pair, hipStream_t s) {
  const long rows = (long)n_pair * (kPatchTokens + 1);
  const unsigned blocks = (unsigned)(rows / 4 + 8 < 4096 ? (rows / 4 + 8) / 8 * 8 : 4096);   // a multiple of 8 (XCDs)
  VETO_LAUNCH(qkv0_combine_kernel, dim3(blocks), dim3(256), 0, s, sw, ow, stats, vec, subj, obj, qkv, n_pair);
  return hipGetLastError();
}

// Saturation audit of mixed rows (VETO_MIXED activations, common.h; veto_forward_saturation): counts, over `rows` rows of K elements
// (row r at base + r * stride bytes), the fp16 values at +-65504 and the e4m3 bytes at +-448 of the value plane (Y) and of the
// residual plane (X).  A thread takes 16 bytes at a time; counters = {elements, fp16, value plane, residual plane}.
// The counts are UPPER BOUNDS of the clamped elements: an encoding at the clamp also holds the values that merely round to it
// (|a| in [432, 464) for e4m3, the last fp16 binade's top), and the NaN / Inf encodings (fp16 exponent all ones, e4m3 0x7f) are
// counted with them -- a NaN operand is the case that actually breaks a result and must not read as "no saturation".
__global__ __launch_bounds__(256) void count_saturation_kernel(const char* base, long stride, int rows, int K, unsigned long long* counters) {
  const int per_row = K / 4;                      // 16-byte pieces of a row (4 K bytes)
  const long total = (long)rows * per_row;
  unsigned nf = 0, nv = 0, nr = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / per_row), c = (int)(i % per_row);
    const u32x4 v = *(const u32x4*)(base + (size_t)r * stride + (size_t)c * 16);
    if ((c & 15) < 8) {                           // the fp16 half of a 256-byte block
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        nf += (v[e] & 0x7fffu) >= 0x7bffu;
        nf += ((v[e] >> 16) & 0x7fffu) >= 0x7bffu;
      }
    } else {                                      // groups of 8 bytes: 4 x X (residual), 4 x Y (value)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const unsigned hit = ((v[e] >> (8 * b)) & 0x7fu) >= 0x7eu;
          if (e & 1) nv += hit; else nr += hit;
        }
    }
  }
  unsigned long long f = nf, vv = nv, rr = nr;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    f += __shfl_xor(f, o, 64);
    vv += __shfl_xor(vv, o, 64);
    rr += __shfl_xor(rr, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (f) atomicAdd(counters + 1, f);
    if (vv) atomicAdd(counters + 2, vv);
    if (rr) atomicAdd(counters + 3, rr);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(counters, (unsigned long long)rows * K);
  }
}

hipError_t launch_centre_split(const float* patch_tab, __bf16* dst, int rows, hipStream_t s) {
  VETO_LAUNCH(centre_split_kernel, dim3((unsigned)((2L * rows + 15) / 16)), dim3(256), 0, s, patch_tab, dst, rows);
  return hipGetLastError();
}

hipError_t launch_qkv0_consts(const float* wq, const float* gamma, const float* beta, const float* pos, const float* cls, float* wp,
                              float* vec, hipStream_t s) {
  VETO_LAUNCH(qkv0_consts_kernel, dim3(3 * kDim), dim3(64), 0, s, wq, gamma, beta, pos, cls, wp, vec);
  return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, long ldx, const float* w, const float* b, __bf16* dst, int rows,
                            hipStream_t s, int fmt, long ldd, bool x_f24) {
  if (ldd == 0) ldd = 2 * kDim;
  if (x_f24) {
    if (fmt == FMT_MIXED) VETO_LAUNCH((layernorm_kernel<FMT_MIXED, true>), dim3((rows + 15) / 16), dim3(256), 0, s, x, ldx, w, b, dst, rows, ldd);
    else VETO_LAUNCH((layernorm_kernel<FMT_SPLIT, true>), dim3((rows + 15) / 16), dim3(256), 0, s, x, ldx, w, b, dst, rows, ldd);
    return hipGetLastError();
  }
  if (fmt == FMT_MIXED) VETO_LAUNCH(layernorm_kernel<FMT_MIXED>, dim3((rows + 15) / 16), dim3(256), 0, s, x, ldx, w, b, dst, rows, ldd);
  else VETO_LAUNCH(layernorm_kernel<FMT_SPLIT>, dim3((rows + 15) / 16), dim3(256), 0, s, x, ldx, w, b, dst, rows, ldd);
  return hipGetLastError();
}

hipError_t launch_mixed_act_rows(const float* src, __bf16* dst, size_t rows, int K, hipStream_t s) {
  if (K % 64 != 0) return hipErrorInvalidValue;
  const size_t n = rows * (size_t)K;
  const int blocks = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 : 2048);
  VETO_LAUNCH(mixed_act_rows_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n, K);
  return hipGetLastError();
}

hipError_t launch_mixed_weight_rows(const float* src, __bf16* dst, size_t rows, int K, int* exp_out, hipStream_t s) {
  if (K % 64 != 0) return hipErrorInvalidValue;
  const size_t n = rows * (size_t)K;
  const int blocks = (int)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 : 2048);
  hipError_t e = hipMemsetAsync(exp_out, 0, sizeof(int), s);
  if (e != hipSuccess) return e;
  VETO_LAUNCH(absmax_bits_kernel, dim3(blocks), dim3(256), 0, s, src, n, (unsigned*)exp_out);
  VETO_LAUNCH(mixed_weight_rows_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n, K, exp_out);
  VETO_LAUNCH(finish_weight_exp_kernel, dim3(1), dim3(1), 0, s, exp_out);
  return hipGetLastError();
}

hipError_t launch_count_saturation(const void* base, long stride_bytes, int rows, int K, unsigned long long* counters, hipStream_t s) {
  if (rows <= 0 || K % 64 != 0 || !counters) return hipErrorInvalidValue;
  const long pieces = (long)rows * (K / 4);
  const int blocks = (int)((pieces + 255) / 256 < 4096 ? (pieces + 255) / 256 : 4096);
  VETO_LAUNCH(count_saturation_kernel, dim3(blocks), dim3(256), 0, s, (const char*)base, stride_bytes, rows, K, counters);
  return hipGetLastError();
}

hipError_t launch_head(const float* cls, const float* wt, const float* bias, float* out, int n_pair,
                       int n_out, hipStream_t s, long ld) {
  VETO_LAUNCH(head_kernel, dim3((n_pair + 3) / 4), dim3(128), 0, s, cls, wt, bias, out, n_pair, n_out, ld);
  return hipGetLastError();
}

}  // namespace veto
