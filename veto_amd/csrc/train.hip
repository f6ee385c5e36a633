// Kernels of the training path around the GEMMs: what the forward needs to keep
// activations (un-fused GELU + split, so the pre-activation survives), the one-pass preparation of a gradient matrix for
// the two GEMMs of a Linear's backward, the backward of the classifier head, of the token
// assembly (scatter of the token gradients into the per-object tables) and of the per-object stage (the small
// projections, BatchNorm, embeddings), and the inverse of the weight re-layouts of rowops.hip.
// The per-object matrices are tiny (a few hundred objects x <= 1152 columns): plain fp32 kernels.
#include "common.h"
#include "kernels.h"

namespace veto {

namespace {

// ---- one pass over a gradient matrix dY fp32 [M, N]: split rows (the operand of the input-gradient GEMM and, read through
// transposing LDS loads, of the weight-gradient GEMM) and, optionally, per-block column sums (bias gradient, folded later).
// 32 (m) x 64 (n) tile through LDS.
// XF folds the elementwise op that precedes the Linear's output in the backward chain into the load: XF_GELU multiplies by
// gelu'(pre) (model_veto.py:140, exact erf), XF_DROP applies the counter-based dropout mask of the forward (element m * N + n).
template <int XF>
__global__ __launch_bounds__(256) void prep_grad_kernel(const float* __restrict__ src, long ld, int M, int N, __bf16* __restrict__ rows_out,
                                                        float* __restrict__ col_partial, GradXform xf) {
  __shared__ float t[32][65];
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 64, tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + ty * 8 + i, n = n0 + tx;
    float v = 0.f;
    if (m < M && n < N) {
      v = src[(size_t)m * ld + n];
      if constexpr (XF == XF_GELU) {
        const float x = xf.pre[(size_t)m * ld + n];
        const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
        v = v * (cdf + x * pdf);
      }
      if constexpr (XF == XF_DROP) v = dropout_keep(xf.seed, (size_t)m * N + n, xf.thresh) ? v * xf.scale : 0.f;
    }
    t[ty * 8 + i][tx] = v;
  }
  __syncthreads();
  // split rows: row m, the two 32-column blocks of this tile; one 8-column piece per thread
  {
    const int ml = tid >> 3, piece = tid & 7;       // 32 rows x 8 pieces of 8 columns
    const int m = m0 + ml, c = n0 + piece * 8;
    if (m < M && c < N) {
      bf16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 h, l;
        split_bf16(t[ml][piece * 8 + e], h, l);
        hi[e] = h;
        lo[e] = l;
      }
      __bf16* d = rows_out + (size_t)m * (2 * (size_t)N) + split_index(c);
      *(bf16x8*)d = hi;
      *(bf16x8*)(d + 32) = lo;
    }
  }
  if (col_partial && tid < 64 && n0 + tid < N) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += t[i][tid];
    col_partial[(size_t)blockIdx.x * N + n0 + tid] = acc;
  }
}

// ---- hid = split(gelu(pre)), pre fp32 [rows, n_cols] (n_cols % 4 == 0) --------------------------------------------
__global__ __launch_bounds__(256) void gelu_split_kernel(const float* __restrict__ pre, __bf16* __restrict__ dst, size_t rows,
                                                         int n_cols) {
  const int per_row = n_cols / 4;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = rows * per_row, stride = (size_t)gridDim.x * 256;
  for (; i < total; i += stride) {
    const size_t row = i / per_row;
    const int c = (int)(i % per_row) * 4;
    const f32x4 v = *(const f32x4*)(pre + row * n_cols + c);
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16 hh, ll;
      split_bf16(gelu_erf(v[e]), hh, ll);
      h[e] = hh;
      l[e] = ll;
    }
    __bf16* d = dst + row * (2 * (size_t)n_cols) + split_index(c);
    *(bf16x4*)d = h;
    *(bf16x4*)(d + 32) = l;
  }
}

// ---- classifier head: logits = cls . W^T + b, cls = row p*19 of x ------------------------------------------------
// dx[p, k] = sum_c dlogits[p, c] W[c, k]  (row p at dx + p * ld)
__global__ __launch_bounds__(576) void head_dcls_kernel(const float* __restrict__ dlogits, const float* __restrict__ w,
                                                        float* __restrict__ dx, int n_out, long ld) {
  extern __shared__ float s_g[];
  const int p = blockIdx.x, k = threadIdx.x;
  for (int c = k; c < n_out; c += 576) s_g[c] = dlogits[(size_t)p * n_out + c];
  __syncthreads();
  float acc = 0.f;
  for (int c = 0; c < n_out; ++c) acc += s_g[c] * w[(size_t)c * kDim + k];
  dx[(size_t)p * ld + k] = acc;
}

// dW[c, k] = sum_p dlogits[p, c] x[p, k] (x row p at x + p * ld): pair chunk blockIdx.y writes partial[chunk, c, k]; a fixed-order fold
// over the chunks follows (launch_column_sums on the [chunks, n_out*576] matrix)
__global__ __launch_bounds__(576) void head_dw_kernel(const float* __restrict__ dlogits, const float* __restrict__ x,
                                                      float* __restrict__ partial, int n_pair, int n_out, int chunk, long ld) {
  const int c = blockIdx.x, k = threadIdx.x;
  const int p0 = blockIdx.y * chunk, p1 = p0 + chunk < n_pair ? p0 + chunk : n_pair;
  float acc = 0.f;
  for (int p = p0; p < p1; ++p) acc += dlogits[(size_t)p * n_out + c] * x[(size_t)p * ld + k];
  partial[((size_t)blockIdx.y * n_out + c) * kDim + k] = acc;
}

// ---- token assembly backward: token gradients -> per-object tables -----------------------------------------------
// token t of pair (s, o): 1..16 = patch_tab[s, t-1, :576] + patch_tab[o, t-1, 576:]; 17 / 18 = relu(lc[s, w, :576] +
// lc[o, w, 576:]) (+ pos_embedding[t] everywhere; token 0 = cls_token: both handled by column sums)
__global__ __launch_bounds__(256) void assemble_backward_kernel(const float* __restrict__ dx, const int32_t* __restrict__ subj,
                                                                const int32_t* __restrict__ obj, const float* __restrict__ lc,
                                                                float* __restrict__ dpatch, float* __restrict__ dlc, int n_pair) {
  const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= (long)n_pair * kTokens) return;
  const int q = threadIdx.x & 15;
  const int p = (int)(row / kTokens), t = (int)(row % kTokens);
  if (t == 0) return;
  const int s = subj[p], o = obj[p];
  const float* g = dx + (size_t)row * kDim;
  if (t <= kPatchTokens) {
    float* ds = dpatch + ((size_t)s * 16 + (t - 1)) * (2 * kDim);
    float* dob = dpatch + ((size_t)o * 16 + (t - 1)) * (2 * kDim) + kDim;
    for (int c = q; c < kDim; c += 16) {
      const float v = g[c];
      unsafeAtomicAdd(ds + c, v);
      unsafeAtomicAdd(dob + c, v);
    }
  } else {
    const int which = t - kPatchTokens - 1;
    const size_t is = ((size_t)s * 2 + which) * (2 * kDim), io = ((size_t)o * 2 + which) * (2 * kDim) + kDim;
    for (int c = q; c < kDim; c += 16) {
      if (lc[is + c] + lc[io + c] > 0.f) {       // ReLU of the location / class projection
        const float v = g[c];
        unsafeAtomicAdd(dlc + is + c, v);
        unsafeAtomicAdd(dlc + io + c, v);
      }
    }
  }
}

// ---- small fp32 matrix products over the objects ------------------------------------------------------------------
// C[i, j] = sum_r A[r, i] B[r, j]   (A: [n, lda], B: [n, ldb], C: [ka, kb] with ldc)
__global__ __launch_bounds__(256) void sgemm_tn_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                       float* __restrict__ c, long ldc, int n, int ka, int kb) {
  const int i = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= kb) return;
  float acc = 0.f;
  for (int r = 0; r < n; ++r) acc += a[(size_t)r * lda + i] * b[(size_t)r * ldb + j];
  c[(size_t)i * ldc + j] = acc;
}

// C[r, i] = sum_j A[r, j] B[i, j]   (A: [n, kj] lda, B: [ki, kj] ldb, C: [n, ki] ldc)
__global__ __launch_bounds__(256) void sgemm_nt_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                       float* __restrict__ c, long ldc, int n, int ki, int kj) {
  const int r = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ki) return;
  float acc = 0.f;
  for (int j = 0; j < kj; ++j) acc += a[(size_t)r * lda + j] * b[(size_t)i * ldb + j];
  c[(size_t)r * ldc + i] = acc;
}

// C[r, i] = sum_j A[r, j] B[j, i]   (A: [n, kj] lda, B: [kj, ki] ldb, C: [n, ki] ldc)
__global__ __launch_bounds__(256) void sgemm_nn_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                       float* __restrict__ c, long ldc, int n, int ki, int kj) {
  const int r = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ki) return;
  float acc = 0.f;
  for (int j = 0; j < kj; ++j) acc += a[(size_t)r * lda + j] * b[(size_t)j * ldb + i];
  c[(size_t)r * ldc + i] = acc;
}

// row softmax of [n, c] (c <= 256), one workgroup per row: the soft object embedding of sgcls (:4092-4095)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int c) {
  __shared__ float s_v[256], s_red[2];
  const int n = blockIdx.x, tid = threadIdx.x;
  s_v[tid] = tid < c ? x[(size_t)n * c + tid] : -INFINITY;
  __syncthreads();
  if (tid == 0) {
    float mx = -INFINITY, sum = 0.f;
    for (int i = 0; i < c; ++i) mx = fmaxf(mx, s_v[i]);
    for (int i = 0; i < c; ++i) sum += expf(s_v[i] - mx);
    s_red[0] = mx;
    s_red[1] = sum;
  }
  __syncthreads();
  if (tid < c) y[(size_t)n * c + tid] = expf(s_v[tid] - s_red[0]) * (1.f / s_red[1]);
}

// x[n, k] = relu(x[n, k] + b[k])
__global__ __launch_bounds__(256) void bias_relu_kernel(float* __restrict__ x, const float* __restrict__ b, int n, int k) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * k) return;
  x[i] = fmaxf(x[i] + b[i % k], 0.f);
}

// ---- per-object stage backward: recompute the normalised box, the position pre-activation and apply ReLU' ------
// in:  dpos [n, 128] (gradient of the ReLU'd position embedding);  out: dpre [n, 128] (masked), xhat [n, 4],
//      bn_out [n, 4], dbn_out [n, 4] = dpre . pos_w
__global__ __launch_bounds__(128) void obj_pos_backward_kernel(const float* __restrict__ boxes, int box_mode, const float* __restrict__ stats,
                                                               const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                               const float* __restrict__ pos_w, const float* __restrict__ pos_b,
                                                               const float* __restrict__ dpos, float* __restrict__ dpre,
                                                               float* __restrict__ xhat, float* __restrict__ bn_out,
                                                               float* __restrict__ dbn_out) {
  __shared__ float s_x[4], s_y[4], s_d[kPosDim];
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid < 4) {
    const float* b = boxes + (size_t)n * 4;
    float w, h;
    if (box_mode == 0) { w = b[2] - b[0] + 1.f; h = b[3] - b[1] + 1.f; } else { w = b[2]; h = b[3]; }
    const float v = tid == 0 ? b[0] + 0.5f * w : tid == 1 ? b[1] + 0.5f * h : tid == 2 ? w : h;
    s_x[tid] = (v - stats[tid]) / sqrtf(stats[4 + tid] + 1e-5f);
    s_y[tid] = s_x[tid] * bn_w[tid] + bn_b[tid];
    xhat[(size_t)n * 4 + tid] = s_x[tid];
    bn_out[(size_t)n * 4 + tid] = s_y[tid];
  }
  __syncthreads();
  float pre = pos_b[tid];
#pragma unroll
  for (int k = 0; k < 4; ++k) pre += pos_w[tid * 4 + k] * s_y[k];
  const float d = pre > 0.f ? dpos[(size_t)n * kPosDim + tid] : 0.f;
  s_d[tid] = d;
  dpre[(size_t)n * kPosDim + tid] = d;
  __syncthreads();
  if (tid < 4) {
    float acc = 0.f;
    for (int k = 0; k < kPosDim; ++k) acc += s_d[k] * pos_w[k * 4 + tid];
    dbn_out[(size_t)n * 4 + tid] = acc;
  }
}

// BatchNorm1d(4) affine gradients: dgamma[c] = sum_n dy[n, c] xhat[n, c], dbeta[c] = sum_n dy[n, c]
__global__ void bn_affine_backward_kernel(const float* __restrict__ dy, const float* __restrict__ xhat, int n, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta) {
  const int c = threadIdx.x;
  if (c >= 4) return;
  double g = 0.0, b = 0.0;
  for (int r = 0; r < n; ++r) { g += (double)dy[(size_t)r * 4 + c] * xhat[(size_t)r * 4 + c]; b += dy[(size_t)r * 4 + c]; }
  dgamma[c] = (float)g;
  dbeta[c] = (float)b;
}

// emb[n, :] = E[label[n], :]
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ labels, int dim,
                                                          float* __restrict__ out) {
  const int n = blockIdx.x;
  for (int e = threadIdx.x; e < dim; e += 256) out[(size_t)n * dim + e] = table[(size_t)labels[n] * dim + e];
}

// dE[label[n], :] += demb[n, :]
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ demb, const int64_t* __restrict__ labels, int dim,
                                                           float* __restrict__ dtable) {
  const int n = blockIdx.x;
  for (int e = threadIdx.x; e < dim; e += 256) unsafeAtomicAdd(dtable + (size_t)labels[n] * dim + e, demb[(size_t)n * dim + e]);
}

// inverse of transpose_pair_proj_kernel: dW[j, half*kin + k] = dWt[k, half*576 + j]
__global__ __launch_bounds__(256) void untranspose_pair_proj_kernel(const float* __restrict__ dwt, float* __restrict__ dw, int kin) {
  const int k = blockIdx.x;
  for (int jj = threadIdx.x; jj < 2 * kDim; jj += 256) {
    const int half = jj / kDim, j = jj % kDim;
    dw[(size_t)j * (2 * kin) + half * kin + k] = dwt[(size_t)k * (2 * kDim) + jj];
  }
}

// inverse of build_patch_weight_kernel: dWcatT [2048, 1152] (row = patch feature kk, column = table column j) ->
// proj_d.weight.grad [512, 2048], proj_v.weight.grad [64, 2048]
__global__ __launch_bounds__(256) void patch_weight_grad_kernel(const float* __restrict__ dwcat_t, float* __restrict__ dwd, float* __restrict__ dwv) {
  const int j = blockIdx.x;  // 0..1151
  const int half = j / kDim, jj = j % kDim;
  for (int f = threadIdx.x; f < 1024; f += 256) {
    const int pp = f >> 8, c = f & 255;
    if (jj < 512) dwd[(size_t)jj * 2048 + pp * 512 + half * 256 + c] = dwcat_t[(size_t)f * (2 * kDim) + j];
    else dwv[(size_t)(jj - 512) * 2048 + pp * 512 + half * 256 + c] = dwcat_t[(size_t)(1024 + f) * (2 * kDim) + j];
  }
}

}  // namespace

hipError_t launch_prep_grad(const float* src, long ld, int M, int N, __bf16* rows_out, int Mp, float* col_partial, const GradXform& xf,
                            hipStream_t s) {
  if (Mp % 32 != 0 || Mp < M || N % 32 != 0) return hipErrorInvalidValue;
  const dim3 grid(Mp / 32, (N + 63) / 64);
  if (xf.mode == XF_GELU) {
    if (!xf.pre) return hipErrorInvalidValue;
    VETO_LAUNCH(prep_grad_kernel<XF_GELU>, grid, dim3(256), 0, s, src, ld, M, N, rows_out, col_partial, xf);
  } else if (xf.mode == XF_DROP) {
    if (ld != N) return hipErrorInvalidValue;
    VETO_LAUNCH(prep_grad_kernel<XF_DROP>, grid, dim3(256), 0, s, src, ld, M, N, rows_out, col_partial, xf);
  } else {
    VETO_LAUNCH(prep_grad_kernel<XF_NONE>, grid, dim3(256), 0, s, src, ld, M, N, rows_out, col_partial, xf);
  }
  return hipGetLastError();
}

hipError_t launch_gelu_split(const float* pre, __bf16* dst, size_t rows, int n_cols, hipStream_t s) {
  if (n_cols % 32 != 0) return hipErrorInvalidValue;
  const size_t total = rows * (n_cols / 4);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  VETO_LAUNCH(gelu_split_kernel, dim3(blocks), dim3(256), 0, s, pre, dst, rows, n_cols);
  return hipGetLastError();
}

size_t head_backward_partial_floats(int n_out) {
  const size_t a = (size_t)(64 + 8) * n_out * kDim, b = (size_t)column_sums_chunks() * n_out;
  return a + b;
}

hipError_t launch_head_backward(const float* dlogits, const float* w, const float* x, float* dx, float* dw, float* db, float* partial,
                                int n_pair, int n_out, long ld, hipStream_t s) {
  VETO_LAUNCH(head_dcls_kernel, dim3(n_pair), dim3(576), (size_t)n_out * 4, s, dlogits, w, dx, n_out, ld);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int n_chunks = 64, chunk = (n_pair + n_chunks - 1) / n_chunks;
  VETO_LAUNCH(head_dw_kernel, dim3(n_out, n_chunks), dim3(576), 0, s, dlogits, x, partial, n_pair, n_out, chunk, ld);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  float* scratch = partial + (size_t)n_chunks * n_out * kDim;
  if ((e = launch_column_sums(partial, (long)n_out * kDim, n_chunks, n_out * kDim, dw, scratch, 8, s)) != hipSuccess) return e;
  return launch_column_sums(dlogits, n_out, n_pair, n_out, db, scratch, column_sums_chunks(), s);
}

hipError_t launch_assemble_backward(const float* dx, const int32_t* subj, const int32_t* obj, const float* lc, float* dpatch, float* dlc,
                                    int n_pair, hipStream_t s) {
  VETO_LAUNCH(assemble_backward_kernel, dim3((unsigned)(((long)n_pair * kTokens + 15) / 16)), dim3(256), 0, s, dx, subj, obj, lc, dpatch,
              dlc, n_pair);
  return hipGetLastError();
}

hipError_t launch_sgemm_tn(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int n, int ka, int kb, hipStream_t s) {
  VETO_LAUNCH(sgemm_tn_kernel, dim3((kb + 255) / 256, ka), dim3(256), 0, s, a, lda, b, ldb, c, ldc, n, ka, kb);
  return hipGetLastError();
}

hipError_t launch_sgemm_nt(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int n, int ki, int kj, hipStream_t s) {
  VETO_LAUNCH(sgemm_nt_kernel, dim3((ki + 255) / 256, n), dim3(256), 0, s, a, lda, b, ldb, c, ldc, n, ki, kj);
  return hipGetLastError();
}

hipError_t launch_obj_pos_backward(const float* boxes, int box_mode, const float* stats, const float* bn_w, const float* bn_b,
                                   const float* pos_w, const float* pos_b, const float* dpos, float* dpre, float* xhat, float* bn_out,
                                   float* dbn_out, float* dgamma, float* dbeta, int n_obj, hipStream_t s) {
  VETO_LAUNCH(obj_pos_backward_kernel, dim3(n_obj), dim3(kPosDim), 0, s, boxes, box_mode, stats, bn_w, bn_b, pos_w, pos_b, dpos, dpre, xhat,
              bn_out, dbn_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  VETO_LAUNCH(bn_affine_backward_kernel, dim3(1), dim3(64), 0, s, dbn_out, xhat, n_obj, dgamma, dbeta);
  return hipGetLastError();
}

hipError_t launch_sgemm_nn(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int n, int ki, int kj, hipStream_t s) {
  VETO_LAUNCH(sgemm_nn_kernel, dim3((ki + 255) / 256, n), dim3(256), 0, s, a, lda, b, ldb, c, ldc, n, ki, kj);
  return hipGetLastError();
}

hipError_t launch_softmax_rows(const float* x, float* y, int n, int c, hipStream_t s) {
  if (c > 256) return hipErrorInvalidValue;
  VETO_LAUNCH(softmax_rows_kernel, dim3(n), dim3(256), 0, s, x, y, c);
  return hipGetLastError();
}

hipError_t launch_bias_relu(float* x, const float* b, int n, int k, hipStream_t s) {
  VETO_LAUNCH(bias_relu_kernel, dim3((n * k + 255) / 256), dim3(256), 0, s, x, b, n, k);
  return hipGetLastError();
}

hipError_t launch_gather_rows(const float* table, const int64_t* labels, int dim, float* out, int n, hipStream_t s) {
  VETO_LAUNCH(gather_rows_kernel, dim3(n), dim3(256), 0, s, table, labels, dim, out);
  return hipGetLastError();
}

hipError_t launch_scatter_rows(const float* demb, const int64_t* labels, int dim, float* dtable, int n, hipStream_t s) {
  VETO_LAUNCH(scatter_rows_kernel, dim3(n), dim3(256), 0, s, demb, labels, dim, dtable);
  return hipGetLastError();
}

hipError_t launch_untranspose_pair_proj(const float* dwt, float* dw, int kin, hipStream_t s) {
  VETO_LAUNCH(untranspose_pair_proj_kernel, dim3(kin), dim3(256), 0, s, dwt, dw, kin);
  return hipGetLastError();
}

hipError_t launch_patch_weight_grad(const float* dwcat_t, float* dwd, float* dwv, hipStream_t s) {
  VETO_LAUNCH(patch_weight_grad_kernel, dim3(2 * kDim), dim3(256), 0, s, dwcat_t, dwd, dwv);
  return hipGetLastError();
}

}  // namespace veto
