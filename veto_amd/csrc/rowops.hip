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
    if (a.pos_out) a.pos_out[(size_t)n * kPosDim + tid] = s_pos[tid];
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
  float* out = a.lc + (size_t)n * 2 * (2 * kDim);
  for (int jj = tid; jj < 2 * kDim; jj += 256) {
    float acc = jj < kDim ? a.loc_b[jj] : 0.f;
    for (int k = 0; k < kPosDim; ++k) acc += a.loc_wt[(size_t)k * (2 * kDim) + jj] * s_pos[k];
    out[jj] = acc;
    float acc2 = jj < kDim ? a.cls_b[jj] : 0.f;
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

// A 576-wide row lives in one wave as float2 pairs: pair index lane + 64*i (i < 4) covers columns
// 0..511, lanes 0..31 additionally hold pair 256 + lane (columns 512..575).
struct RowRegs { float v[10]; };

__device__ __forceinline__ int row_col(int lane, int i) { return i < 4 ? 2 * (lane + 64 * i) : 512 + 2 * lane; }

__device__ __forceinline__ void row_layernorm_store(const RowRegs& r, int lane, const float* __restrict__ w,
                                                    const float* __restrict__ b, __bf16* __restrict__ dst) {
  const bool tail = lane < 32;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += r.v[i];
  if (tail) s += r.v[8] + r.v[9];
  const float mean = wave_sum(s) * (1.f / kDim);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float d = r.v[i] - mean; q += d * d; }
  if (tail) { const float d0 = r.v[8] - mean, d1 = r.v[9] - mean; q += d0 * d0 + d1 * d1; }
  const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / kDim) + 1e-5f);
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    if (i == 4 && !tail) break;
    const int c = row_col(lane, i);
    const float y0 = (r.v[2 * i] - mean) * rstd * w[c] + b[c];
    const float y1 = (r.v[2 * i + 1] - mean) * rstd * w[c + 1] + b[c + 1];
    __bf16 h0, l0, h1, l1;
    split_bf16(y0, h0, l0);
    split_bf16(y1, h1, l1);
    __bf16* d = dst + split_index(c);  // c is even: the pair stays inside one 32-k block
    *(bf16x2*)d = bf16x2{h0, h1};
    *(bf16x2*)(d + 32) = bf16x2{l0, l1};
  }
}

// One workgroup per pair; wave w builds token rows w, w+4, ...  Reads of the per-object tables hit
// L2 / Infinity Cache (each object row is re-used by 2(N-1) pairs); writes are the HBM stream.
__global__ __launch_bounds__(256) void assemble_kernel(AssembleArgs a) {
  const int p = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int s = a.subj[p], o = a.obj[p];
  const bool tail = lane < 32;
  for (int t = w; t < kTokens; t += 4) {
    RowRegs r;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i == 4 && !tail) { r.v[8] = 0.f; r.v[9] = 0.f; break; }
      const int c = row_col(lane, i);
      float2 v;
      const float2 pe = *(const float2*)(a.pos_embedding + c);
      if (t == 0) {
        v = *(const float2*)(a.cls_token + c);
      } else if (t <= kPatchTokens) {
        const float2 vs = *(const float2*)(a.patch_tab + ((size_t)s * 16 + (t - 1)) * (2 * kDim) + c);
        const float2 vo = *(const float2*)(a.patch_tab + ((size_t)o * 16 + (t - 1)) * (2 * kDim) + kDim + c);
        v.x = vs.x + vo.x;
        v.y = vs.y + vo.y;
      } else {
        const int which = t - kPatchTokens - 1;  // 0 location, 1 class
        const float2 vs = *(const float2*)(a.lc + ((size_t)s * 2 + which) * (2 * kDim) + c);
        const float2 vo = *(const float2*)(a.lc + ((size_t)o * 2 + which) * (2 * kDim) + kDim + c);
        v.x = fmaxf(vs.x + vo.x, 0.f);
        v.y = fmaxf(vs.y + vo.y, 0.f);
      }
      v.x += pe.x;
      v.y += pe.y;
      r.v[2 * i] = v.x;
      r.v[2 * i + 1] = v.y;
      *(float2*)(a.x + ((size_t)p * kTokens + t) * kDim + c) = v;
    }
    const size_t row = (size_t)p * kTokens + t;
    row_layernorm_store(r, lane, a.ln_w, a.ln_b, a.a + row * (2 * kDim));
  }
}

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b,
                                                        __bf16* __restrict__ dst, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + (size_t)row * ldx;
  RowRegs r;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    if (i == 4 && lane >= 32) { r.v[8] = 0.f; r.v[9] = 0.f; break; }
    const float2 v = *(const float2*)(xr + row_col(lane, i));
    r.v[2 * i] = v.x;
    r.v[2 * i + 1] = v.y;
  }
  row_layernorm_store(r, lane, w, b, dst + (size_t)row * (2 * kDim));
}

// logits[p][c] = cls[p] . W[c] + b[c]; weights pre-transposed to [576][n_out]; 4 pairs per block.
__global__ __launch_bounds__(128) void head_kernel(const float* __restrict__ cls,
                                                   const float* __restrict__ wt,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   int n_pair, int n_out) {
  __shared__ float s_cls[4][kDim];
  const int p0 = blockIdx.x * 4;
  for (int i = threadIdx.x; i < 4 * kDim; i += 128) {
    const int pp = i / kDim, k = i % kDim;
    s_cls[pp][k] = p0 + pp < n_pair ? cls[(size_t)(p0 + pp) * kDim + k] : 0.f;
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

hipError_t launch_obj_prep(const ObjPrepArgs& a, hipStream_t s) {
  if (a.num_obj_cls > 256 || a.embed_dim > 256) return hipErrorInvalidValue;
  VETO_LAUNCH(obj_prep_kernel, dim3(a.n_obj), dim3(256), 0, s, a);
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
  VETO_LAUNCH(assemble_kernel, dim3(a.n_pair), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, long ldx, const float* w, const float* b, __bf16* dst, int rows,
                            hipStream_t s) {
  VETO_LAUNCH(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, w, b, dst, rows);
  return hipGetLastError();
}

hipError_t launch_head(const float* cls, const float* wt, const float* bias, float* out, int n_pair,
                       int n_out, hipStream_t s) {
  VETO_LAUNCH(head_kernel, dim3((n_pair + 3) / 4), dim3(128), 0, s, cls, wt, bias, out, n_pair, n_out);
  return hipGetLastError();
}

}  // namespace veto
