// Split-bf16 MFMA GEMM for the token-wise Linear layers of the VETO relation transformer
// (to_qkv / to_out / FeedForward of model_veto.py:78-96,137-143 and the per-object patch
// projection derived from model_veto.py:103-113).
//
//   C[M,N] = A[M,K] . W[N,K]^T        (nn.Linear layout: both operands K-contiguous)
//
// A and W are stored as split rows (common.h): per row and per 32 k's, 64 B of bf16 hi then 64 B of
// bf16 lo, x ~= hi + lo (16 significand bits).
// PRECISE (NTERMS=3): A_hi.W_hi + A_lo.W_hi + A_hi.W_lo, fp32 accumulate  -> ~2^-16 relative,
// which is what the 1e-3 logit bar needs (plain bf16 misses it by 14x).
// FAST (NTERMS=1): A_hi.W_hi only.
//
// This file: the homogeneous kernel (every wave loads and computes) and the dispatcher.
// gemm_split_ps.hip: persistent + loader-wave variant of the same tile.  Both give bit-identical C.
//
// Tile 256(M) x 192(N) x 32(K), 8 waves (4 along M x 2 along N), v_mfma_f32_16x16x32_bf16.
// The MFMA is issued "swapped" (weights as the A operand, activations as the B operand) so that
// each lane ends up with 4 CONSECUTIVE output columns of one row -> 16-byte epilogue accesses.
// LDS: two stages of {A rows, W rows} x 128 B = 2 x 56 KiB, filled by global_load_lds_dwordx4 (no
// VGPR staging): one wave-instruction moves 8 rows x 128 B (8 full cache lines).  Inside a row the
// eight 16-byte slots (4 hi k-chunks, 4 lo k-chunks) are XOR-swizzled with (row>>1)&7 -- on the
// per-lane SOURCE address for the DMA (its LDS destination is lane-linear) and again on the
// ds_read_b128 address -- which makes every fragment read bank-conflict free.
#include "common.h"
#include "kernels.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace veto {

namespace {

constexpr int BM = 256, BN = 192, BK = 32;
constexpr int NWAVES = 8;
constexpr int kStageBytes = (BM + BN) * 128;     // 57344
constexpr int kWOff = BM * 128;                  // W rows follow the A rows inside a stage
constexpr int NCHUNK = (BM + BN) / 8;            // 56 chunks of 8 rows x 128 B
constexpr int CA = BM / 8;
constexpr int CPW = NCHUNK / NWAVES;             // 7 per wave and stage

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int NTERMS, int EPI>
__global__ __launch_bounds__(64 * NWAVES, 2) void gemm_split_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6) & (NWAVES - 1);
  const int wm = w >> 1, wn = w & 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of logical tiles; consecutive logical tiles walk N fastest and therefore re-use
  // the same 256-row activation panel out of that XCD's L2.  Speed only, never correctness.
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int tile_n = logical % g.tiles_n;
  const int tile_m = logical / g.tiles_n;

  const int K = g.K;
  // ---- per-lane source pointers of this wave's 8-row chunks (7 per stage) ----------------------
  const int rr = lane >> 3;                       // row inside the chunk
  const char* src[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = w + NWAVES * i;                 // chunk id, wave-uniform; CA is a multiple of NWAVES
    const int r16 = ((c & 1) << 3) + rr;          // row inside its 16-row MFMA tile
    const int slot = (lane & 7) ^ ((r16 >> 1) & 7);  // source 16-byte slot that lands in LDS slot lane&7
    const __bf16* base;
    long ld;
    int row;
    if (c < CA) {
      base = g.a; ld = g.lda; row = tile_m * BM + c * 8 + rr;
      if (g.lda != 2 * K && row >= g.M) row = g.M - 1;  // strided (unpadded) A rows: clamp
    } else {
      base = g.w; ld = 2 * K; row = tile_n * BN + (c - CA) * 8 + rr;
    }
    src[i] = (const char*)(base + (size_t)row * ld) + slot * 16;
  }
  auto load_stage = [&](int stage, int kt) {
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      glds16(src[i] + (size_t)kt * 128, smem + stage * kStageBytes + (w + NWAVES * i) * 1024);
  };

  // ---- fragment read offsets: row fr of a 16-row tile, hi k-chunk fq; lo = same ^ 64 -------------
  const int fr = lane & 15, fq = lane >> 4;
  const int frag_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
  const int a_off = (wm * 64) * 128 + frag_off;           // activation rows of this wave (4 tiles of 16)
  const int w_off = kWOff + (wn * 96) * 128 + frag_off;   // weight rows of this wave (6 tiles of 16)

  f32x4 acc[6][4];
#pragma unroll
  for (int n = 0; n < 6; ++n)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  bf16x8 ah[4], al[4], wh[6], wl[6];

  // One k-step.  Software pipeline inside the step: while the 12 MFMAs of weight tile n run, the
  // fragments of tile n+1 are read from LDS and (in the first 4 groups) two LDS-DMA chunks of the
  // NEXT stage are issued, so the vector-memory issue cost hides under the matrix pipe instead of
  // preceding it.  sched_group_barrier pins that interleave in the emitted code.
  auto k_step = [&](int kt, auto load_next) {
    constexpr bool LOAD = decltype(load_next)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* st = smem + (kt & 1) * kStageBytes;
    char* nst = smem + ((kt + 1) & 1) * kStageBytes;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      ah[m] = *(const bf16x8*)(st + a_off + m * 2048);
      if (NTERMS == 3) al[m] = *(const bf16x8*)(st + ((a_off + m * 2048) ^ 64));
    }
    wh[0] = *(const bf16x8*)(st + w_off);
    if (NTERMS == 3) wl[0] = *(const bf16x8*)(st + (w_off ^ 64));
    constexpr int GL = (CPW + 1) / 2;  // groups that carry global loads (2 chunks each)
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      if (n < 5) {
        wh[n + 1] = *(const bf16x8*)(st + w_off + (n + 1) * 2048);
        if (NTERMS == 3) wl[n + 1] = *(const bf16x8*)(st + ((w_off + (n + 1) * 2048) ^ 64));
      }
      if (LOAD && n < GL) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = 2 * n + j;
          if (i < CPW) glds16(src[i] + (size_t)(kt + 1) * 128, nst + (w + NWAVES * i) * 1024);
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (NTERMS == 3) {
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[n], ah[m], acc[n][m], 0, 0, 0);
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], al[m], acc[n][m], 0, 0, 0);
        }
        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], ah[m], acc[n][m], 0, 0, 0);
      }
    }
    if (NTERMS == 3) {  // DS_READ 0x100, VMEM 0x10, MFMA 0x8
      __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        if (n < 5) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        if (LOAD && n < GL - 1) __builtin_amdgcn_sched_group_barrier(0x10, 2, 0);
        if (LOAD && n == GL - 1) __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);  // CPW = 7: last group has one
        __builtin_amdgcn_sched_group_barrier(0x8, 12, 0);
      }
    }
  };

  load_stage(0, 0);
  for (int kt = 0; kt < nk - 1; ++kt) k_step(kt, std::true_type{});
  k_step(nk - 1, std::false_type{});

  // ---- epilogue: lane holds C[row = m-tile row lane&15][4 consecutive cols (lane>>4)*4..+3] ----
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = tile_m * BM + wm * 64 + m * 16 + (lane & 15);
    if (row >= g.M) continue;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const int col = tile_n * BN + wn * 96 + n * 16 + (lane >> 4) * 4;
      f32x4 v = acc[n][m];
      if (g.bias) v += *(const f32x4*)(g.bias + col);
      if (EPI == EPI_RESID) v += *(const f32x4*)(g.resid + (size_t)row * g.ldr + col);
      if (EPI == EPI_GELU_SPLIT) {
        bf16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 h, l;
          split_bf16(gelu_sigmoid(v[e]), h, l);
          hi[e] = h;
          lo[e] = l;
        }
        __bf16* dst = g.c_split + (size_t)row * g.ldc + split_index(col);
        *(bf16x4*)dst = hi;
        *(bf16x4*)(dst + 32) = lo;
      } else {
        *(f32x4*)(g.c + (size_t)row * g.ldc + col) = v;
      }
    }
  }
}

template <int NTERMS>
hipError_t launch_terms(GemmArgs g, int epi, hipStream_t s) {
  dim3 grid(g.tiles_m * g.tiles_n), block(64 * NWAVES);
  switch (epi) {
    case EPI_F32: VETO_LAUNCH((gemm_split_kernel<NTERMS, EPI_F32>), grid, block, 0, s, g); break;
    case EPI_RESID: VETO_LAUNCH((gemm_split_kernel<NTERMS, EPI_RESID>), grid, block, 0, s, g); break;
    case EPI_GELU_SPLIT: VETO_LAUNCH((gemm_split_kernel<NTERMS, EPI_GELU_SPLIT>), grid, block, 0, s, g); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

// Row padding every contiguous A-operand buffer must have.
int gemm_rows_padded(int m) { return (m + BM - 1) / BM * BM; }

hipError_t launch_gemm_split(GemmArgs g, int epi, int precision, hipStream_t s) {
  if (g.N % BN != 0 || g.K % BK != 0 || g.M <= 0) return hipErrorInvalidValue;
  if (g.lda == 0) g.lda = 2 * (long)g.K;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = g.N / BN;
  // the training-only forms (split-K / atomic, row-major weight gradients, dropout before the residual) exist in the
  // persistent kernel only
  if (g.tn || epi == EPI_ATOMIC || epi == EPI_RESID_DROP) return launch_gemm_split_ps(g, epi, precision, s);
  // VETO_GEMM_VARIANT (A/B knob): "ps" persistent + loader waves (default), "plain" homogeneous waves.
  static const bool plain = [] {
    const char* e = getenv("VETO_GEMM_VARIANT");
    return e && !strcmp(e, "plain");
  }();
  if (!plain) return launch_gemm_split_ps(g, epi, precision, s);
  return precision == 0 ? launch_terms<3>(g, epi, s) : launch_terms<1>(g, epi, s);
}

}  // namespace veto
