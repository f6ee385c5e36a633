// MFMA GEMM for the token-wise Linear layers of the VETO relation transformer (to_qkv / to_out / FeedForward of
// model_veto.py:78-96,137-143 and the per-object patch projection derived from model_veto.py:103-113):
//
//   C[M,N] = A[M,K] . W[N,K]^T        (nn.Linear layout: both operands K-contiguous)
//
// Operands are split rows or mixed rows (common.h).
//   NTERMS == 3 (VETO_PRECISE): split rows, A_hi.W_hi + A_lo.W_hi + A_hi.W_lo on v_mfma_f32_16x16x32_bf16, fp32 accumulate
//                -> ~2^-16 relative (plain bf16 misses the 1e-3 logit bar by 14x).
//   NTERMS == 1 (VETO_FAST): A_hi.W_hi only (reported, never trusted).
//   NTERMS == 2 (VETO_MIXED): mixed rows, fp16 main product + e4m3 correction terms (below).
//
// Tile 256(M) x 192(N) x 32(K), 8 consumer waves (4 along M x 2 along N) + 4 loader waves.  The MFMA is issued "swapped"
// (weights as the A operand, activations as the B operand) so that each lane ends up with 4 CONSECUTIVE output columns of one
// row -> 16-byte epilogue accesses.  LDS stages are filled by global_load_lds_dwordx4 (no VGPR staging): one wave-instruction
// moves 8 rows x 128 B (8 full cache lines).  Inside a row the eight 16-byte slots are XOR-swizzled with (row>>1)&7 -- on the
// per-lane SOURCE address of the DMA (its LDS destination is lane-linear) and again on the ds_read_b128 address -- which makes
// every fragment read bank-conflict free.
//
//  * Loader waves.  4 of the 12 waves of a workgroup only issue the LDS-DMA (global_load_lds) of
//    the next stage; the 8 consumer waves run nothing but ds_read_b128 + MFMA between barriers.
//  * Persistent workgroups.  One workgroup per CU walks a strided list of tiles and treats
//    (tile, k-step) as ONE stream of stages: the loaders run into the next tile while the consumers
//    are still in the epilogue, and the epilogue stores are fire-and-forget.
//
// Protocol (one s_barrier per stage s, all 12 waves take part).  LDS holds THREE activation stages
// and TWO weight stages (3 x 32 KiB + 2 x 24 KiB = 144 KiB): the activation panel streams from
// HBM / Infinity Cache and gets two k-steps of flight time, the weights come from L2 and get one.
//   loader:   A(0) W(0) A(1); for s: { vmcnt(8): all but A(s+1) landed; barrier B_s; W(s+1); A(s+2) }
//   consumer:                 for s: {                                   barrier B_s; compute(s) }
// (vmcnt counts in issue order: A(s) and W(s) are older than A(s+1), whose 8 chunks stay in flight
// ACROSS the barrier -- the loader's issue time overlaps the flight of the previous stage instead of
// adding to it.)  B_s = "A(s), W(s) have landed" + "the buffers stage s-1 used are free" (every
// consumer finished stage s-1; its fragment reads were consumed by its own MFMAs before it arrived).
//
// Tile order: workgroup b lives on XCD b%8 (round-robin dispatch; speed only) and in round i takes
// tile ((i*8 + b%8)*32 + b/8): the 32 workgroups of an XCD work on 32 consecutive tiles (N fastest),
// i.e. on ~3.5 activation panels x all weight panels, which is what their shared L2 then holds.
//
// NTERMS == 2 is the VETO_MIXED instantiation (operands in the mixed-row format of common.h): the stages of a tile alternate
// between the fp16 part of a 64-k block (two v_mfma_f32_16x16x32_f16 per output block) and its e4m3 part (one
// v_mfma_scale_f32_16x16x128_f8f6f4 whose E8M0 scale undoes the operands' power-of-two scaling).  Loaders, LDS image,
// fragment reads and barriers are the same: a stage is 128 bytes of every row in either format.
//
// -DVETO_GEMM_STAMPS builds a diagnostic copy that accumulates s_memtime deltas per phase (barrier
// wait / MFMA phase / epilogue; loader: vmcnt wait / barrier / issue) and prints their means.
#include "common.h"
#include "kernels.h"

#include <cstdio>
#include <cstdlib>

namespace veto {

namespace {

constexpr int BM = 256, BN = 192, BK = 32;
constexpr int NCONS = 8, NLOAD = 4;
constexpr int kABytes = BM * 128;             // one A stage: 256 rows x 128 B = 32 KiB (3 of them)
constexpr int kWBytes = BN * 128;             // one W stage: 192 rows x 128 B = 24 KiB (2 of them)
constexpr int kDumpOff = 3 * kABytes + 2 * kWBytes;
constexpr int kBiasOff = kDumpOff + 1024;     // three 768-byte bias slices (tile index mod 3), filled by loader wave 0
constexpr int kBiasSlices = 3;                // the loaders run at most two STAGES ahead: two tiles when a tile is one k-step
constexpr int kLdsBytes = kBiasOff + kBiasSlices * BN * 4;  // 144 KiB + a dump area for prefetches + the bias slices
constexpr int kPrefSteps = BN * 4 / 128;      // 6: 128-byte lines per residual row of a tile
constexpr int CPA = BM / 8 / NLOAD;           // 8 A chunks (8 rows x 128 B) per loader wave and stage
constexpr int CPWL = BN / 8 / NLOAD;          // 6 W chunks
#ifndef VETO_GEMM_EPI_T
#define VETO_GEMM_EPI_T 1
#endif
constexpr bool kEpiT = VETO_GEMM_EPI_T;       // epilogue lane transposition (A/B knob; results identical)

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

#ifdef VETO_GEMM_STAMPS
__device__ unsigned long long g_stamps[256 * 8];
__device__ __forceinline__ unsigned long long stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(x) x = stamp()
#define ACC(a, t1, t0) a += (t1) - (t0)
#else
#define STAMP(x)
#define ACC(a, t1, t0)
#endif

__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int V> struct IntTag { static constexpr int value = V; };

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// 4 rows x 16 columns of 16-bit elements, delivered column-major: lane i of a 16-lane group gets column i of the 4 rows
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
// one MFMA operand fragment (8 reduction indices of one row / column) from an image whose ROWS are reduction indices:
// two transposed reads, `pitch4` = 4 image rows apart
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* p, int pitch4) {
  const s16x4 a = lds_tr16(p), b = lds_tr16(p + pitch4);
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int NTERMS, int EPI_T, bool TN = false>
__global__ __launch_bounds__(64 * (NCONS + NLOAD), 3) void gemm_split_ps_kernel(GemmArgs g) {
  constexpr bool kDrop = EPI_T == EPI_RESID_DROP;          // training-only instantiation: dropout before the residual add
  constexpr bool kMixed = NTERMS == 2;                     // fp16 + e4m3 operands (mixed rows)
  if constexpr (kMixed) saturating_conversions_on();       // (only these instantiations can write mixed rows: the fc1 epilogue)
  static_assert(!(kMixed && TN), "the mixed-row format has no transposed (weight-gradient) form");
  constexpr bool kGeluBwd = EPI_T == EPI_GELU_BWD;         // training-only: the residual machinery reads the pre-activation, the result leaves as split rows
  constexpr int EPI = kDrop || kGeluBwd ? (int)EPI_RESID : EPI_T;
  __shared__ __attribute__((aligned(16))) char smem[kLdsBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int ksp = EPI == EPI_ATOMIC && g.k_splits > 1 ? g.k_splits : 1;
  const int out_tiles = g.tiles_m * g.tiles_n;
  const int ntiles = out_tiles * ksp;        // split-K: (k range, output tile), output tile fastest
  const int K = g.K;
  const int nk = g.kb_tiles > 0 ? g.kb_steps : K / BK / ksp;   // k-steps per tile (block-diagonal weights: the steps of the tile's block)
  // tile of round `it` for this workgroup; gridDim.x is a multiple of 8 (launcher)
  const int per_xcd = gridDim.x >> 3;
  auto tile_of = [&](int it) { return (it * 8 + (b & 7)) * per_xcd + (b >> 3); };
  int my_tiles = 0;
  while (tile_of(my_tiles) < ntiles) ++my_tiles;  // tile_of is increasing in `it`
  if (my_tiles == 0) return;
  const int nstages = my_tiles * nk;

  if (w >= NCONS) {
    // ------------------------------- loader wave ------------------------------------------------
    const int lw = (w - NCONS) & (NLOAD - 1);
    const int rr = lane >> 3;
    const char* srcA[CPA];
    const char* srcW[CPWL];
    auto src_of = [&](const __bf16* base, long ld, int row, int c) {
      const int r16 = ((c & 1) << 3) + rr;
      const int slot = (lane & 7) ^ ((r16 >> 1) & 7);
      return (const char*)(base + (size_t)row * ld) + slot * 16;
    };
    // TN: a stage is 32 reduction rows; an A row is the 1 KiB of one split row that covers the tile's 256 columns (one
    // instruction, lane = LDS slot), a W row the 768 bytes of its 192 columns (an instruction covers 64 consecutive slots of
    // the stage).  The 16-byte chunk c of image row r sits in slot c ^ tn_swz(r): the transposed reads of a 32-lane half
    // touch 8 rows x 2 chunks and must land on 16 different bank groups.
    auto tn_swz = [](int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); };
    const char* tnA = nullptr;   // tile's column offset applied
    const char* tnW = nullptr;
    int tn_rowA = 0, tn_rowW = 0;
    int tn_xa[2] = {0, 0}, tn_wr[CPWL], tn_wx[CPWL];
    if constexpr (TN) {
      tn_xa[0] = (lane ^ tn_swz(lw)) << 4;          // stage row lw + 4 i: (row & 3) = lw, (row >> 3) & 1 = (i >> 1) & 1
      tn_xa[1] = (lane ^ tn_swz(lw + 8)) << 4;
#pragma unroll
      for (int i = 0; i < CPWL; ++i) {
        const int slot = (lw + NLOAD * i) * 64 + lane, r = slot / 48, cc = slot % 48;
        tn_wr[i] = r;
        tn_wx[i] = (cc ^ tn_swz(r)) << 4;
      }
    }
    auto setupA = [&](int tile) {
      int k0 = (tile / out_tiles) * nk;   // first k-step of this tile's range
      tile %= out_tiles;
      const int tile_m = tile / g.tiles_n;
      if (g.kb_tiles > 0) k0 = ((tile % g.tiles_n) / g.kb_tiles) * nk;
      if constexpr (TN) {
        tnA = (const char*)g.a + (size_t)tile_m * (BM * 4);
        tn_rowA = k0 * BK;
        return;
      }
#pragma unroll
      for (int i = 0; i < CPA; ++i) {
        const int c = lw + NLOAD * i;
        int row = tile_m * BM + c * 8 + rr;
        if (g.lda != 2 * K && row >= g.M) row = g.M - 1;  // strided (unpadded) A rows
        srcA[i] = src_of(g.a, g.lda, row, c) + (size_t)k0 * 128;
      }
    };
    auto setupW = [&](int tile) {
      int k0 = (tile / out_tiles) * nk;
      tile %= out_tiles;
      const int tile_n = tile % g.tiles_n;
      if (g.kb_tiles > 0) k0 = (tile_n / g.kb_tiles) * nk;
      if constexpr (TN) {
        tnW = (const char*)g.w + (size_t)tile_n * (BN * 4);
        tn_rowW = k0 * BK;
        return;
      }
#pragma unroll
      for (int i = 0; i < CPWL; ++i) {
        const int c = lw + NLOAD * i;
        srcW[i] = src_of(g.w, 2 * K, tile_n * BN + c * 8 + rr, c) + (size_t)k0 * 128;
      }
    };
    // cursors of the next A stage / W stage to issue (the stream of stages crosses tile boundaries)
    int itA = 0, ktA = 0, bufA = 0, itW = 0, ktW = 0, bufW = 0;
    auto issueA = [&]() {
      // first stage of a tile: loader wave 0 also fetches the tile's 192 bias values into slice (tile index mod 3) (the
      // epilogue reads them from LDS: no bias registers held across it, no vector-memory wait behind its own stores).  Issued
      // BEFORE the A chunks: older than them in vmcnt order, so the stage's counted wait covers it.
      if (ktA == 0 && lw == 0 && g.bias && lane < BN / 4) {
        const int tn = (tile_of(itA) % out_tiles) % g.tiles_n;
        glds16((const char*)(g.bias + tn * BN) + lane * 16, smem + kBiasOff + (itA % kBiasSlices) * (BN * 4));
      }
      char* dst = smem + bufA * kABytes + lw * 1024;
      if constexpr (TN) {
#pragma unroll
        for (int i = 0; i < CPA; ++i) {
          const int row = tn_rowA + ktA * BK + lw + NLOAD * i;        // wave-uniform
          const char* p = row < g.k_valid ? tnA + (size_t)row * (size_t)(g.lda * 2) : (const char*)g.zero;
          glds16(p + tn_xa[(i >> 1) & 1], dst + NLOAD * i * 1024);
        }
      } else {
#pragma unroll
      for (int i = 0; i < CPA; ++i) glds16(srcA[i] + (size_t)ktA * 128, dst + NLOAD * i * 1024);
      }
      bufA = bufA == 2 ? 0 : bufA + 1;
      if (++ktA == nk) {
        ktA = 0;
        if (++itA < my_tiles) setupA(tile_of(itA));
      }
    };
    auto issueW = [&]() {
      char* dst = smem + 3 * kABytes + bufW * kWBytes + lw * 1024;
      if constexpr (TN) {
#pragma unroll
        for (int i = 0; i < CPWL; ++i) {
          const int row = tn_rowW + ktW * BK + tn_wr[i];               // per lane: an instruction spans two image rows
          const char* p = row < g.k_valid ? tnW + (size_t)row * (size_t)(g.ldw * 2) : (const char*)g.zero;
          glds16(p + tn_wx[i], dst + NLOAD * i * 1024);
        }
      } else {
#pragma unroll
      for (int i = 0; i < CPWL; ++i) glds16(srcW[i] + (size_t)ktW * 128, dst + NLOAD * i * 1024);
      }
      bufW ^= 1;
      if (++ktW == nk) {
        ktW = 0;
        if (++itW < my_tiles) setupW(tile_of(itW));
      }
    };
    // EPI_RESID: the epilogue reads a 256 x 192 fp32 residual tile that nobody has touched since the
    // previous kernel, i.e. from HBM, synchronously, on all CUs at once (46k cycles per tile against 7k
    // for a store-only epilogue).  During the last 6 k-steps of a tile the loader waves touch one
    // 128-byte line per lane of that tile (4-byte LDS-DMA into a dump area) so that it is L2 / MALL
    // resident when the consumers ask for it.  Pure prefetch: results do not depend on it.
    auto prefetch_resid = [&](int tile, int j) {   // EPI_RESID only: never split-K
      const int tile_n = tile % g.tiles_n, tile_m = tile / g.tiles_n;
      int row = tile_m * BM + lw * 64 + lane;
      if (row >= g.M) row = g.M - 1;
      const float* p = g.resid + (size_t)row * g.ldr + tile_n * BN + j * 32;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + kDumpOff + lw * 256), 4, 0, 0);
    };
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, t_begin = 0, a_vm = 0, a_bar = 0, a_iss = 0;
    (void)tA; (void)tB; (void)tC; (void)tD; (void)t_begin; (void)a_vm; (void)a_bar; (void)a_iss;
    STAMP(t_begin);
    setupA(tile_of(0));
    setupW(tile_of(0));
    issueA();                   // A(0)
    issueW();                   // W(0)
    if (nstages > 1) issueA();  // A(1)
    bool pref_in_flight = false;  // a residual prefetch was the youngest request of the last iteration
    int itC = 0, ktC = 0;         // (tile, k-step) the consumers work on after barrier B_s
    for (int s = 0; s < nstages; ++s) {
      STAMP(tA);
      // outstanding, oldest first: ..., A(s), W(s), A(s+1) [, prefetch]: everything but A(s+1)'s CPA
      // chunks (and the prefetch, which gets another k-step) must be done
      if (s + 1 >= nstages) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (pref_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPA + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPA) : "memory");
      STAMP(tB);
      wg_barrier();
      STAMP(tC);
      if (s + 1 < nstages) issueW();  // W(s+1) -> the W buffer stage s-1 used
      if (s + 2 < nstages) issueA();  // A(s+2) -> the A buffer stage s-1 used
      pref_in_flight = false;
      if (EPI == EPI_RESID && s + 2 < nstages && ktC >= nk - kPrefSteps) {
        prefetch_resid(tile_of(itC), ktC - (nk - kPrefSteps));
        pref_in_flight = true;
      }
      if (++ktC == nk) { ktC = 0; ++itC; }
      STAMP(tD);
      ACC(a_vm, tB, tA); ACC(a_bar, tC, tB); ACC(a_iss, tD, tC);
    }
#ifdef VETO_GEMM_STAMPS
    if (lw == 0 && lane == 0) {
      g_stamps[b * 8 + 4] = a_vm; g_stamps[b * 8 + 5] = a_bar; g_stamps[b * 8 + 6] = a_iss; g_stamps[b * 8 + 7] = tD - t_begin;
    }
#endif
    return;
  }

  // --------------------------------- consumer wave ------------------------------------------------
  const int wm = w >> 1, wn = w & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int frag_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
  const int a_off = (wm * 64) * 128 + frag_off;
  const int w_off = 3 * kABytes + (wn * 96) * 128 + frag_off;
  // TN images (rows = reduction index): lane 16 g + 4 q + p of a transposed read supplies row 8 g + q (+ 4 for the second
  // read), 8-byte half p & 1 of chunk (c + (p >> 1)) ^ tn_swz(row) -- the loader's slot rule
  const int tn_g = lane >> 4, tn_q = (lane >> 2) & 3, tn_p = lane & 3;
  const int tn_y = ((2 * (tn_q | ((tn_g & 1) << 2))) ^ (tn_p >> 1)) << 4;
  const int tn_a = (8 * tn_g + tn_q) * (BM * 4) + tn_y + (tn_p & 1) * 8 + wm * 256;   // chunk bits join by XOR below
  const int tn_w = 3 * kABytes + (8 * tn_g + tn_q) * (BN * 4) + (tn_p & 1) * 8;

  // E8M0 scale of the e4m3 MFMA: the operands carry 2^(15 + e), e = the weight tensor's exponent (device memory)
  const int mix_scale = kMixed ? (127 - kMixActExp - __builtin_amdgcn_readfirstlane(*g.w_exp)) * 0x01010101 : 0;
  (void)mix_scale;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t_begin = 0, a_bar = 0, a_cmp = 0, a_epi = 0;
  (void)t0; (void)t1; (void)t2; (void)t3; (void)t_begin; (void)a_bar; (void)a_cmp; (void)a_epi;
  STAMP(t_begin);
  int s = 0, s3 = 0;  // stage counter and s % 3
  for (int it = 0; it < my_tiles; ++it) {
    const int tile = tile_of(it) % out_tiles;
    const int tile_n = tile % g.tiles_n, tile_m = tile / g.tiles_n;
    f32x4 acc[6][4];
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 ah[4], al[4], wh[6], wl[6];
    i32x4 fa0[4], fa1[4], fw0[6], fw1[6];   // kMixed: slot g | slot 4+g of the stage row = the 32-byte e4m3 operand, or two fp16 fragments
    // one stage; KIND (kMixed only): 0 = the fp16 part of a 64-k block, 1 = its e4m3 part
    auto stage = [&](auto kind_tag) {
      constexpr int KIND = decltype(kind_tag)::value;
      STAMP(t0);
      wg_barrier();
      STAMP(t1);
      const char* sa = smem + s3 * kABytes;
      const char* sw = smem + (s & 1) * kWBytes;
      // Fragment reads are issued in the order the MFMAs need them, two at a time, one group ahead of
      // their use: the first MFMA waits for two LDS reads (not twelve), and all 12 waves of the
      // workgroup hitting the LDS right after the barrier overlap with matrix work instead of
      // preceding it.  NTERMS == 1 skips the lo fragments.
      auto rdA = [&](int m) {
        if constexpr (kMixed) {
          fa0[m] = *(const i32x4*)(sa + a_off + m * 2048);
          fa1[m] = *(const i32x4*)(sa + ((a_off + m * 2048) ^ 64));
        } else if constexpr (TN) {
          const int o = tn_a ^ (((m >> 1) * 8 + (m & 1) * 2) << 4);
          ah[m] = lds_tr_frag(sa + o, 4 * BM * 4);
          if (NTERMS == 3) al[m] = lds_tr_frag(sa + (o ^ 64), 4 * BM * 4);
        } else {
          ah[m] = *(const bf16x8*)(sa + a_off + m * 2048);
          if (NTERMS == 3) al[m] = *(const bf16x8*)(sa + ((a_off + m * 2048) ^ 64));
        }
      };
      auto rdW = [&](int n) {
        if constexpr (kMixed) {
          fw0[n] = *(const i32x4*)(sw + w_off + n * 2048);
          fw1[n] = *(const i32x4*)(sw + ((w_off + n * 2048) ^ 64));
        } else if constexpr (TN) {
          const int o = tn_w + ((((wn * 3 + (n >> 1)) * 8 + (n & 1) * 2) << 4) ^ tn_y);
          wh[n] = lds_tr_frag(sw + o, 4 * BN * 4);
          if (NTERMS == 3) wl[n] = lds_tr_frag(sw + (o ^ 64), 4 * BN * 4);
        } else {
          wh[n] = *(const bf16x8*)(sw + w_off + n * 2048);
          if (NTERMS == 3) wl[n] = *(const bf16x8*)(sw + ((w_off + n * 2048) ^ 64));
        }
      };
      auto mma = [&](int n, int m) {
        if constexpr (kMixed) {
          if constexpr (KIND == 0) {
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fw0[n]), __builtin_bit_cast(f16x8, fa0[m]), acc[n][m], 0, 0, 0);
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fw1[n]), __builtin_bit_cast(f16x8, fa1[m]), acc[n][m], 0, 0, 0);
          } else {
            const i32x8 w8 = __builtin_shufflevector(fw0[n], fw1[n], 0, 1, 2, 3, 4, 5, 6, 7), a8 = __builtin_shufflevector(fa0[m], fa1[m], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[n][m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w8, a8, acc[n][m], 0, 0, 0, mix_scale, 0, 0x7f7f7f7f);
          }
        } else {
          if (NTERMS == 3) {
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[n], ah[m], acc[n][m], 0, 0, 0);
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], al[m], acc[n][m], 0, 0, 0);
          }
          acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], ah[m], acc[n][m], 0, 0, 0);
        }
      };
      rdW(0);
      rdA(0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m < 3) rdA(m + 1); else rdW(1);
        mma(0, m);
      }
#pragma unroll
      for (int n = 1; n < 6; ++n) {
        if (n < 5) rdW(n + 1);
#pragma unroll
        for (int m = 0; m < 4; ++m) mma(n, m);
      }
      if (NTERMS >= 2) {  // pin the interleave: DS_READ 0x100, MFMA 0x8
        constexpr int R = TN ? 2 : 1;   // a transposed fragment is two reads
        constexpr int MPB = kMixed ? (KIND == 0 ? 2 : 1) : 3;   // MFMAs per output block and stage
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * R, 0);   // W0 (hi, lo), A0 (hi, lo)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2 * R, 0); // A(m+1) or W1
          __builtin_amdgcn_sched_group_barrier(0x8, MPB, 0);   // tile (0, m)
        }
#pragma unroll
        for (int n = 1; n < 6; ++n) {
          if (n < 5) __builtin_amdgcn_sched_group_barrier(0x100, 2 * R, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 4 * MPB, 0);
        }
      }
      STAMP(t2);
      ACC(a_bar, t1, t0); ACC(a_cmp, t2, t1);
      ++s;
      s3 = s3 == 2 ? 0 : s3 + 1;
    };
    if constexpr (kMixed) {
      for (int kt = 0; kt < nk; kt += 2) {   // nk is even: a 64-k block is two stages
        stage(IntTag<0>());
        stage(IntTag<1>());
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) stage(IntTag<0>());
    }

    // epilogue.  The MFMA leaves lane l with C[row l&15 of the 16-row group][4 consecutive columns of chunk
    // l>>4]: four neighbouring lanes hold four DIFFERENT rows, so a 16-byte store per lane reaches memory as
    // 64 separate 16-byte pieces.  kEpiT: a ds_bpermute per value (lane 4r+q takes the value of lane 16q+r)
    // turns that into lane = (row r = l>>2, chunk q = l&3): every quad of lanes writes 64 contiguous bytes of
    // one row, and the residual is read the same way.  Values and the order of the additions are unchanged.
    // Stores are not waited for.  EPI_RESID: every residual read of the tile is issued before its first store (two
    // phases, below; c may alias resid, which stops the compiler from moving loads across stores by itself).
    // (the lane id is laundered through an empty asm so that the epilogue's lane-dependent offsets are recomputed per tile
    // instead of being hoisted out of the persistent loop, where they would stay live -- and spill -- across the k-loop)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int er = kEpiT ? lane_e >> 2 : lane_e & 15;   // row inside a 16-row group
    const int eq = kEpiT ? lane_e & 3 : lane_e >> 4;    // 16-byte chunk inside a 16-column group
    const int perm_addr = ((eq << 4) + er) << 2;    // byte address of the source lane for ds_bpermute
    const int row0 = tile_m * BM + wm * 64 + er;
    const int col0 = tile_n * BN + wn * 96 + eq * 4;
    const char* bias_lds = smem + kBiasOff + (it % kBiasSlices) * (BN * 4) + (wn * 96 + eq * 4) * 4;
    // the epilogue walks 8 units of (16-row group m, half h of the 6 column groups); EPI_RESID: the residual of unit
    // u+1 is in flight while unit u is combined (two buffers of 3 x 4 registers next to the 96 accumulators)
    f32x4 res[2][3];
    auto load_res = [&](int u, f32x4 (&r)[3]) {
      int row = row0 + (u >> 1) * 16;
      if (row >= g.M) row = g.M - 1;  // clamp: the value is never stored
#pragma unroll
      for (int j = 0; j < 3; ++j) r[j] = *(const f32x4*)(g.resid + (size_t)row * g.ldr + col0 + ((u & 1) * 3 + j) * 16);
    };
    if (EPI == EPI_RESID) load_res(0, res[0]);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = u >> 1, h = u & 1;
      if (EPI == EPI_RESID && u < 7) load_res(u + 1, res[(u + 1) & 1]);
      const int row = row0 + m * 16;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int n = h * 3 + j;
        f32x4 t = acc[n][m];
        if (kEpiT) {   // executed by every lane: rows past M hold values other lanes need
#pragma unroll
          for (int e = 0; e < 4; ++e)
            t[e] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_addr, __float_as_int(acc[n][m][e])));
        }
        if (kGeluBwd) {
          // dH * gelu'(pre); rows past M contribute zeros to the column sums below (they are never stored)
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = row < g.M ? t[e] * gelu_erf_grad(res[u & 1][j][e]) : 0.f;
          acc[n][m] = v;
        } else if (EPI == EPI_RESID && !kDrop) {
          // every lane consumes its residual load (rows past M read a clamped row): a load left pending on a skipped
          // path would have to be waited for -- behind this tile's stores -- when the k-loop reuses its register
          f32x4 v = t + res[u & 1][j];
          if (g.bias) v += *(const f32x4*)(bias_lds + n * 64);
          acc[n][m] = v;    // stored below, after the last residual load
        } else if (row < g.M) {
          const int col = col0 + n * 16;
          f32x4 v = t;
          if (g.bias) v += *(const f32x4*)(bias_lds + n * 64);
          if (kDrop) {   // training: Dropout behind the attention out projection
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = dropout_keep(g.drop_seed, (unsigned long long)row * g.N + col + e, g.drop_thresh) ? v[e] * g.drop_scale : 0.f;
          }
          if (EPI == EPI_RESID) v += res[u & 1][j];
          if (EPI == EPI_GELU_SPLIT || EPI == EPI_SPLIT) {   // the operand format of the next GEMM = this one's
            if (EPI == EPI_GELU_SPLIT) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_sigmoid(v[e]);
            }
            store_act4<kMixed ? FMT_MIXED : FMT_SPLIT>(g.c_split + (size_t)row * g.ldc, col, v);
          } else if (EPI == EPI_PRE_GELU) {   // training: the pre-activation survives for gelu', its GELU is fc2's operand (was a pass of its own)
            *(f32x4*)(g.c + (size_t)row * g.ldc_f32 + col) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            store_act4<FMT_SPLIT>(g.c_split + (size_t)row * g.ldc, col, v);
          } else if (EPI == EPI_ATOMIC) {
            float* dstc = g.c + (size_t)row * g.ldc + col;
#pragma unroll
            for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dstc + e, v[e]);
          } else if (EPI == EPI_RESID) {
            acc[n][m] = v;    // (training, dropout form) stored below, after the last residual load
          } else if (EPI == EPI_F24) {
            *(u32x3*)((char*)g.c + ((size_t)row * g.ldc + col) * 3) = pack_f24x4(v);
          } else {
            *(f32x4*)(g.c + (size_t)row * g.ldc + col) = v;
          }
        }
      }
    }
    if (kGeluBwd) {
      // column sums of this wave's 64 rows (bias gradient partials): per column block the four row groups, then the 16 lanes that hold the
      // block's rows (lane = 4 row + chunk with the transposition, 16 chunk + row without); a fixed order: deterministic
      float* cp = g.col_partial + (size_t)(tile_m * 4 + wm) * g.N + col0;
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        f32x4 sum = (acc[n][0] + acc[n][1]) + (acc[n][2] + acc[n][3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = sum[e];
          if (kEpiT) { x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64); }
          else { x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64); }
          sum[e] = x;
        }
        if (er == 0) *(f32x4*)(cp + n * 16) = sum;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = u >> 1, h = u & 1;
        const int row = row0 + m * 16;
        if (row < g.M) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int n = h * 3 + j;
            store_act4<FMT_SPLIT>(g.c_split + (size_t)row * g.ldc, col0 + n * 16, acc[n][m]);
          }
        }
      }
    } else if (EPI == EPI_RESID) {
      // Two phases: loads, stores and LDS-DMA share ONE in-order counter per wave, so a residual load issued behind a store
      // cannot return before that store has drained -- and with every CU in its epilogue at once the stores drain slowly (the
      // store-only epilogue of a 256 x 192 tile takes ~10 k cycles).  Interleaved (load group m+1, store group m) the eight units
      // of a tile paid that round trip one after the other (~24 k cycles per tile); with every store behind the last load the
      // loads see an empty queue and the stores of this tile drain under the next tile's k-loop.
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = u >> 1, h = u & 1;
        const int row = row0 + m * 16;
        if (row < g.M) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const int n = h * 3 + j;
            *(f32x4*)(g.c + (size_t)row * g.ldc + col0 + n * 16) = acc[n][m];
          }
        }
      }
    }
    STAMP(t3);
    ACC(a_epi, t3, t2);
  }
#ifdef VETO_GEMM_STAMPS
  if (w == 0 && lane == 0) {
    g_stamps[b * 8 + 0] = a_bar; g_stamps[b * 8 + 1] = a_cmp; g_stamps[b * 8 + 2] = a_epi; g_stamps[b * 8 + 3] = t3 - t_begin;
  }
#endif
}

template <int NTERMS>
hipError_t launch_ps_terms(GemmArgs g, int epi, int nblocks, hipStream_t s) {
  dim3 grid(nblocks), block(64 * (NCONS + NLOAD));
  if constexpr (NTERMS == 2) {   // inference forms only
    switch (epi) {
      case EPI_F32: VETO_LAUNCH((gemm_split_ps_kernel<2, EPI_F32>), grid, block, 0, s, g); break;
      case EPI_F24: VETO_LAUNCH((gemm_split_ps_kernel<2, EPI_F24>), grid, block, 0, s, g); break;
      case EPI_RESID: VETO_LAUNCH((gemm_split_ps_kernel<2, EPI_RESID>), grid, block, 0, s, g); break;
      case EPI_GELU_SPLIT: VETO_LAUNCH((gemm_split_ps_kernel<2, EPI_GELU_SPLIT>), grid, block, 0, s, g); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  } else
  switch (epi) {
    case EPI_F32: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_F32>), grid, block, 0, s, g); break;
    case EPI_F24: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_F24>), grid, block, 0, s, g); break;
    case EPI_RESID: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_RESID>), grid, block, 0, s, g); break;
    case EPI_GELU_SPLIT: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_GELU_SPLIT>), grid, block, 0, s, g); break;
    case EPI_SPLIT: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_SPLIT>), grid, block, 0, s, g); break;
    case EPI_PRE_GELU:
      if constexpr (NTERMS == 3) { VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_PRE_GELU>), grid, block, 0, s, g); break; }
      else return hipErrorInvalidValue;
    case EPI_GELU_BWD:
      if constexpr (NTERMS == 3) {
        if (!g.resid || !g.c_split || !g.col_partial || g.bias) return hipErrorInvalidValue;
        VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_GELU_BWD>), grid, block, 0, s, g);
        break;
      } else return hipErrorInvalidValue;
    case EPI_ATOMIC:
      if (g.tn) VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_ATOMIC, true>), grid, block, 0, s, g);
      else VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_ATOMIC>), grid, block, 0, s, g);
      break;
    case EPI_RESID_DROP: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_RESID_DROP>), grid, block, 0, s, g); break;
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
  return launch_gemm_split_ps(g, epi, precision, s);
}

// g.tiles_m / g.tiles_n / g.lda are already filled by launch_gemm_split.
hipError_t launch_gemm_split_ps(GemmArgs g, int epi, int precision, hipStream_t s) {
  int num_cu = device_cu_count();
  if (num_cu < 1) return hipErrorInvalidDevice;
  num_cu = num_cu / 8 * 8;
  if (num_cu < 8) num_cu = 8;
  const int ksp = epi == EPI_ATOMIC && g.k_splits > 1 ? g.k_splits : 1;
  if ((g.K / BK) % ksp != 0) return hipErrorInvalidValue;
  if (g.kb_tiles > 0 && (ksp != 1 || g.tn || g.fmt == FMT_MIXED || g.kb_steps <= 0 || g.tiles_n % g.kb_tiles != 0 ||
                         (g.tiles_n / g.kb_tiles) * g.kb_steps != g.K / BK))
    return hipErrorInvalidValue;
  if (g.tn && (epi != EPI_ATOMIC || !g.zero || g.lda <= 0 || g.ldw <= 0 || g.k_valid <= 0 || g.k_valid > g.K)) return hipErrorInvalidValue;
  const int ntiles = g.tiles_m * g.tiles_n * ksp;
  int nblocks = num_cu;  // one persistent workgroup per CU (LDS: 112 KiB each)
  if (ntiles < nblocks) nblocks = (ntiles + 7) / 8 * 8;
  if (g.fmt == FMT_MIXED && (g.tn || !g.w_exp || (g.K / BK) % 2 != 0 || ksp != 1)) return hipErrorInvalidValue;
  hipError_t rc = g.fmt == FMT_MIXED ? launch_ps_terms<2>(g, epi, nblocks, s)
                  : precision == 0 ? launch_ps_terms<3>(g, epi, nblocks, s) : launch_ps_terms<1>(g, epi, nblocks, s);
#ifdef VETO_GEMM_STAMPS
  {
    static unsigned long long host[256 * 8];
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(host));
    double sum[8] = {0};
    for (int bb = 0; bb < nblocks && bb < 256; ++bb)
      for (int k = 0; k < 8; ++k) sum[k] += (double)host[bb * 8 + k];
    fprintf(stderr, "[stamps M%d N%d K%d epi%d] consumer: barrier %.0f compute %.0f epilogue %.0f total %.0f | loader: vmwait %.0f "
            "barrier %.0f issue %.0f total %.0f (mean cycles per workgroup; %d tiles x %d k-steps each)\n", g.M, g.N, g.K, epi,
            sum[0] / nblocks, sum[1] / nblocks, sum[2] / nblocks, sum[3] / nblocks, sum[4] / nblocks, sum[5] / nblocks,
            sum[6] / nblocks, sum[7] / nblocks, (ntiles + nblocks - 1) / nblocks, g.K / BK);
  }
#endif
  return rc;
}

}  // namespace veto
