// Backward building blocks of the relation transformer (groundwork: these kernels are
// exposed through test hooks and checked against autograd; the training path that chains them is not built yet).
//   attention_backward   model_veto.py:85-96   (dQ, dK, dV) from dOut and the saved q, k, v of one (pair, head)
//   layernorm_backward   model_veto.py:125-132 dx, and per-block partial sums of dgamma / dbeta
//   gelu_backward        model_veto.py:140     dpre = dh * gelu'(pre), exact-erf GELU
//   column_sums          bias gradients: db[n] = sum over the token rows of dy[:, n] (two-stage, fixed order)
#include "common.h"
#include "kernels.h"

namespace veto {

namespace {

constexpr int kColChunks = 256;    // row chunks of the two-stage column sums (fixed order -> deterministic)

constexpr int kAttnBwdThreads = 256;

// One workgroup of four waves per (pair, head), everything in fp32.  With S = q k^T * scale, P = softmax(S), O = P v:
//   dV = P^T dO,  dP = dO v^T,  dS = P * (dP - rowsum(dP * P)),  dQ = dS k * scale,  dK = dS^T q * scale.
template <int DH>
__global__ __launch_bounds__(kAttnBwdThreads) void attention_backward_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                float* __restrict__ dqkv, int n_pair, int heads) {
  constexpr int LD = DH + 1;                  // padded row stride: column walks hit distinct banks
  __shared__ float q[kTokens * LD], k[kTokens * LD], v[kTokens * LD], go[kTokens * LD];
  __shared__ float p[kTokens * 20], ds[kTokens * 20];
  const int item = blockIdx.x, pair = item / heads, head = item % heads, lane = threadIdx.x;   // index inside the workgroup that owns this (pair, head)
  if (pair >= n_pair) return;
  const float* src = qkv + (size_t)pair * kTokens * (3 * kDim) + head * DH;
  const float* gsrc = dout + (size_t)pair * kTokens * kDim + head * DH;
  for (int e = lane; e < kTokens * DH; e += kAttnBwdThreads) {
    const int t = e / DH, d = e % DH;
    q[t * LD + d] = src[(size_t)t * 3 * kDim + d];
    k[t * LD + d] = src[(size_t)t * 3 * kDim + kDim + d];
    v[t * LD + d] = src[(size_t)t * 3 * kDim + 2 * kDim + d];
    go[t * LD + d] = gsrc[(size_t)t * kDim + d];
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)DH);
  for (int e = lane; e < kTokens * kTokens; e += kAttnBwdThreads) {   // S and dP
    const int i = e / kTokens, j = e % kTokens;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < DH; ++d) { s += q[i * LD + d] * k[j * LD + d]; dp += go[i * LD + d] * v[j * LD + d]; }
    p[i * 20 + j] = s * scale;
    ds[i * 20 + j] = dp;
  }
  __syncthreads();
  if (lane < kTokens) {                                   // row softmax, then dS = P * (dP - sum_j dP P)
    float* pr = p + lane * 20;
    float* dr = ds + lane * 20;
    float mx = -INFINITY;
    for (int j = 0; j < kTokens; ++j) mx = fmaxf(mx, pr[j]);
    float sum = 0.f;
    for (int j = 0; j < kTokens; ++j) { pr[j] = expf(pr[j] - mx); sum += pr[j]; }
    const float inv = 1.f / sum;
    float dot = 0.f;
    for (int j = 0; j < kTokens; ++j) { pr[j] *= inv; dot += dr[j] * pr[j]; }
    for (int j = 0; j < kTokens; ++j) dr[j] = pr[j] * (dr[j] - dot);
  }
  __syncthreads();
  float* dst = dqkv + (size_t)pair * kTokens * (3 * kDim) + head * DH;
  for (int e = lane; e < kTokens * DH; e += kAttnBwdThreads) {
    const int t = e / DH, d = e % DH;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    for (int j = 0; j < kTokens; ++j) {
      dq += ds[t * 20 + j] * k[j * LD + d];     // dQ[t] = sum_j dS[t][j] k[j]
      dk += ds[j * 20 + t] * q[j * LD + d];     // dK[t] = sum_i dS[i][t] q[i]
      dv += p[j * 20 + t] * go[j * LD + d];     // dV[t] = sum_i P[i][t] dO[i]
    }
    dst[(size_t)t * 3 * kDim + d] = dq * scale;
    dst[(size_t)t * 3 * kDim + kDim + d] = dk * scale;
    dst[(size_t)t * 3 * kDim + 2 * kDim + d] = dv;
  }
}

// LayerNorm backward over 576-wide rows, one quarter wave per row (lane q holds chunks q + 16 j as the forward):
//   xhat = (x - mean) * rstd, g = dy * gamma, dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dres)
// dgamma / dbeta: a thread always owns the same 36 columns, so it adds its kLnRowsPerBlock / 16 rows in registers;
// the 16 row slots of the block then meet in LDS once per block, and the block writes one partial row [2, 576]
// that a two-stage column sum folds in a fixed order.
constexpr int kLnRowsPerBlock = 64;

__global__ __launch_bounds__(256) void layernorm_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const float* __restrict__ gamma, const float* __restrict__ dres,
                                                                 float* __restrict__ dx, float* __restrict__ partial, int rows) {
  __shared__ float s_dg[kDim], s_db[kDim];
  for (int c = threadIdx.x; c < kDim; c += 256) { s_dg[c] = 0.f; s_db[c] = 0.f; }
  __syncthreads();
  const int q = threadIdx.x & 15;
  auto gsum = [](float t) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
  };
  f32x4 acc_g[9], acc_b[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { acc_g[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_b[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int it = 0; it < kLnRowsPerBlock / 16; ++it) {
    const int row = blockIdx.x * kLnRowsPerBlock + it * 16 + (threadIdx.x >> 4);
    if (row >= rows) continue;      // uniform per quarter wave
    const float* xr = x + (size_t)row * kDim;
    const float* gr = dy + (size_t)row * kDim;
    f32x4 xv[9], gv[9];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      xv[j] = *(const f32x4*)(xr + 4 * (q + 16 * j));
      gv[j] = *(const f32x4*)(gr + 4 * (q + 16 * j));
      s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]);
    }
    const float mean = gsum(s) * (1.f / kDim);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = xv[j][e] - mean; sq += d * d; }
    const float rstd = 1.f / sqrtf(gsum(sq) * (1.f / kDim) + 1e-5f);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const f32x4 w = *(const f32x4*)(gamma + 4 * (q + 16 * j));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
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
      const int c = 4 * (q + 16 * j);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (gv[j][e] - mg - xv[j][e] * mgx);
      if (dres) o += *(const f32x4*)(dres + (size_t)row * kDim + c);
      *(f32x4*)(dr + c) = o;
    }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * (q + 16 * j) + e;
      atomicAdd(&s_dg[c], acc_g[j][e]);    // 16 row slots per column, once per block
      atomicAdd(&s_db[c], acc_b[j][e]);
    }
  __syncthreads();
  float* pr = partial + (size_t)blockIdx.x * 2 * kDim;
  for (int c = threadIdx.x; c < kDim; c += 256) { pr[c] = s_dg[c]; pr[kDim + c] = s_db[c]; }
}

// out[c] = sum over `n_rows` rows of src[r][c], rows folded in order, in double: used for the LayerNorm partials and
// (through column_partial_kernel) for bias gradients
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ src, long ld, int n_rows, int n_cols,
                                                        float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n_cols) return;
  double acc = 0.0;
  for (int r = 0; r < n_rows; ++r) acc += (double)src[(size_t)r * ld + c];
  out[c] = (float)acc;
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

hipError_t launch_attention_backward(const float* qkv, const float* dout, float* dqkv, int n_pair, int heads, hipStream_t s) {
  if (heads <= 0 || kDim % heads != 0) return hipErrorInvalidValue;
  const int dh = kDim / heads;
  const unsigned blocks = (unsigned)((long)n_pair * heads);
  if (dh == 72) VETO_LAUNCH(attention_backward_kernel<72>, dim3(blocks), dim3(kAttnBwdThreads), 0, s, qkv, dout, dqkv, n_pair, heads);
  else if (dh == 96) VETO_LAUNCH(attention_backward_kernel<96>, dim3(blocks), dim3(kAttnBwdThreads), 0, s, qkv, dout, dqkv, n_pair, heads);
  else if (dh == 144) VETO_LAUNCH(attention_backward_kernel<144>, dim3(blocks), dim3(kAttnBwdThreads), 0, s, qkv, dout, dqkv, n_pair, heads);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// partial rows of the LayerNorm parameter gradients + the scratch of the two-stage column sum over them
size_t layernorm_backward_partial_floats(int rows) {
  return (size_t)((rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock) * 2 * kDim + (size_t)kColChunks * 2 * kDim;
}

hipError_t launch_layernorm_backward(const float* x, const float* dy, const float* gamma, const float* dres, float* dx,
                                     float* dgamma_dbeta, float* partial, int rows, hipStream_t s) {
  const int blocks = (rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock;
  VETO_LAUNCH(layernorm_backward_kernel, dim3(blocks), dim3(256), 0, s, x, dy, gamma, dres, dx, partial, rows);
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
  VETO_LAUNCH(fold_rows_kernel, dim3((n_cols + 255) / 256), dim3(256), 0, s, partial, (long)n_cols, n_chunks, n_cols, out);
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
