// Split-bf16 MFMA GEMM, persistent + wave-specialised variant.
//
// Same tile (256x192x32), split-row operands, LDS image, swizzle, MFMA order and epilogue as
// gemm_split.hip (results are bit-identical); two structural differences:
//
//  * Loader waves.  4 of the 12 waves of a workgroup only issue the LDS-DMA (global_load_lds) of
//    the next stage; the 8 consumer waves run nothing but ds_read_b128 + MFMA between barriers.
//  * Persistent workgroups.  One workgroup per CU walks a strided list of tiles and treats
//    (tile, k-step) as ONE stream of stages: the loaders run into the next tile while the consumers
//    are still in the epilogue, and the epilogue stores are fire-and-forget.
//
// Protocol (2 LDS stages, one s_barrier per stage s, all 12 waves take part):
//   loader:   issue(0); for s: { vmcnt(0); barrier B_s; issue(s+1) }
//   consumer:           for s: {           barrier B_s; compute(s); [epilogue at the end of a tile] }
// B_s = "stage s has landed" (every loader waited for its own DMAs) + "buffer (s+1)&1 is free" (every
// consumer finished stage s-1; its fragment reads were consumed by its own MFMAs before it arrived).
//
// Tile order: workgroup b lives on XCD b%8 (round-robin dispatch; speed only) and in round i takes
// tile ((i*8 + b%8)*32 + b/8): the 32 workgroups of an XCD work on 32 consecutive tiles (N fastest),
// i.e. on ~3.5 activation panels x all weight panels, which is what their shared L2 then holds.
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
constexpr int kStageBytes = (BM + BN) * 128;  // 57344
constexpr int kWOff = BM * 128;
constexpr int NCHUNK = (BM + BN) / 8;         // 56 chunks of 8 rows x 128 B per stage
constexpr int CA = BM / 8;
constexpr int CPL = NCHUNK / NLOAD;           // 14 per loader wave

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

template <int NTERMS, int EPI>
__global__ __launch_bounds__(64 * (NCONS + NLOAD), 3) void gemm_split_ps_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int K = g.K;
  const int nk = K / BK;
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
    const char* src[CPL];
    auto setup = [&](int tile) {
      const int tile_n = tile % g.tiles_n, tile_m = tile / g.tiles_n;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = lw + NLOAD * i;  // CA is a multiple of NLOAD
        const int r16 = ((c & 1) << 3) + rr;
        const int slot = (lane & 7) ^ ((r16 >> 1) & 7);
        const __bf16* base;
        long ld;
        int row;
        if (c < CA) {
          base = g.a; ld = g.lda; row = tile_m * BM + c * 8 + rr;
          if (g.lda != 2 * K && row >= g.M) row = g.M - 1;  // strided (unpadded) A rows
        } else {
          base = g.w; ld = 2 * K; row = tile_n * BN + (c - CA) * 8 + rr;
        }
        src[i] = (const char*)(base + (size_t)row * ld) + slot * 16;
      }
    };
    auto issue = [&](int s, int kt) {
      char* dst = smem + (s & 1) * kStageBytes + lw * 1024;
#pragma unroll
      for (int i = 0; i < CPL; ++i) glds16(src[i] + (size_t)kt * 128, dst + NLOAD * i * 1024);
    };
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, t_begin = 0, a_vm = 0, a_bar = 0, a_iss = 0;
    (void)tA; (void)tB; (void)tC; (void)tD; (void)t_begin; (void)a_vm; (void)a_bar; (void)a_iss;
    STAMP(t_begin);
    setup(tile_of(0));
    issue(0, 0);
    int it = 0, kt = 0;
    for (int s = 0; s < nstages; ++s) {
      STAMP(tA);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(tB);
      wg_barrier();
      STAMP(tC);
      if (++kt == nk) {
        kt = 0;
        ++it;
        if (it < my_tiles) setup(tile_of(it));
      }
      if (s + 1 < nstages) issue(s + 1, kt);
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
  const int w_off = kWOff + (wn * 96) * 128 + frag_off;

  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t_begin = 0, a_bar = 0, a_cmp = 0, a_epi = 0;
  (void)t0; (void)t1; (void)t2; (void)t3; (void)t_begin; (void)a_bar; (void)a_cmp; (void)a_epi;
  STAMP(t_begin);
  int s = 0;
  for (int it = 0; it < my_tiles; ++it) {
    const int tile = tile_of(it);
    const int tile_n = tile % g.tiles_n, tile_m = tile / g.tiles_n;
    f32x4 acc[6][4];
#pragma unroll
    for (int n = 0; n < 6; ++n)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 ah[4], al[4], wh[6], wl[6];
    for (int kt = 0; kt < nk; ++kt, ++s) {
      STAMP(t0);
      wg_barrier();
      STAMP(t1);
      const char* st = smem + (s & 1) * kStageBytes;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        ah[m] = *(const bf16x8*)(st + a_off + m * 2048);
        if (NTERMS == 3) al[m] = *(const bf16x8*)(st + ((a_off + m * 2048) ^ 64));
      }
      wh[0] = *(const bf16x8*)(st + w_off);
      if (NTERMS == 3) wl[0] = *(const bf16x8*)(st + (w_off ^ 64));
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        if (n < 5) {
          wh[n + 1] = *(const bf16x8*)(st + w_off + (n + 1) * 2048);
          if (NTERMS == 3) wl[n + 1] = *(const bf16x8*)(st + ((w_off + (n + 1) * 2048) ^ 64));
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
      if (NTERMS == 3) {  // DS_READ 0x100, MFMA 0x8: fragments of tile n+1 under the MFMAs of tile n
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
        for (int n = 0; n < 6; ++n) {
          if (n < 5) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x8, 12, 0);
        }
      }
      STAMP(t2);
      ACC(a_bar, t1, t0); ACC(a_cmp, t2, t1);
    }

    // epilogue: lane holds C[row lane&15 of m-tile][4 consecutive columns]; stores are not waited for
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
            split_bf16(gelu_erf(v[e]), h, l);
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
  switch (epi) {
    case EPI_F32: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_F32>), grid, block, 0, s, g); break;
    case EPI_RESID: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_RESID>), grid, block, 0, s, g); break;
    case EPI_GELU_SPLIT: VETO_LAUNCH((gemm_split_ps_kernel<NTERMS, EPI_GELU_SPLIT>), grid, block, 0, s, g); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

// g.tiles_m / g.tiles_n / g.lda are already filled by launch_gemm_split.
hipError_t launch_gemm_split_ps(GemmArgs g, int epi, int precision, hipStream_t s) {
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
    num_cu = prop.multiProcessorCount / 8 * 8;
    if (num_cu < 8) num_cu = 8;
  }
  const int ntiles = g.tiles_m * g.tiles_n;
  int nblocks = num_cu;  // one persistent workgroup per CU (LDS: 112 KiB each)
  if (ntiles < nblocks) nblocks = (ntiles + 7) / 8 * 8;
  hipError_t rc = precision == 0 ? launch_ps_terms<3>(g, epi, nblocks, s) : launch_ps_terms<1>(g, epi, nblocks, s);
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
