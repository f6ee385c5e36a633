// Per-pair multi-head self-attention over the 19 relation tokens (model_veto.py:85-96):
//   dots = q k^T * dh^-0.5 ; attn = softmax(dots, -1) ; out = attn v ; heads merged 'b h n d -> b n (h d)'.
// One wave per (pair, head).  The result is written as the hi/lo bf16 planes the out-projection
// GEMM consumes.  cls_only = last layer: only token 0's query is needed (model_veto.py:23 consumes
// x[:, 0] only), k/v still cover all 19 tokens.
//
// attention_mfma_kernel<DH> (head dims 72 and 96 = the 8-head and 6-head configurations): QK^T and
// PV on v_mfma_f32_32x32x16_bf16 in the same 3-term split-bf16 scheme as the GEMMs.  The scores are
// computed TRANSPOSED (S^T = K Q^T) so that one query's 19 scores sit in the 16 accumulator
// registers of lanes l and l+32: the softmax is in-register plus one cross-half shuffle, and the
// probabilities are already the A operand of the PV product (accumulator-as-operand, no LDS trip).
// V is transposed on its way into LDS so the PV B operand is two 8-byte reads.
// attention_kernel (generic head dim): fp32 on the vector ALU, LDS staged.
// the attention output rows (mixed rows, read next by the layer tail) leave through non-temporal stores: table attention 0.60 -> 0.57 ms
#ifndef ACT8_NT
#define ACT8_NT 1
#endif
// folded CLS attention: bit 1 = the residual rows (read once) through non-temporal loads (0.254 -> 0.242 ms, adopted), bit 2 = the abar rows through
// non-temporal stores (the GEMM behind reads them: 0.268 ms, not adopted)
#ifndef CLS_NT
#define CLS_NT 1
#endif
#include "common.h"
#include "kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

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
    bf16x4 hi, lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      __bf16 h, l;
      split_bf16(acc[c], h, l);
      hi[c] = h;
      lo[c] = l;
    }
    __bf16* dst = a.o + row * (2 * kDim) + split_index(head * dh + d4 * 4);
    *(bf16x4*)dst = hi;
    *(bf16x4*)(dst + 32) = lo;
  }
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
// Two fp32 values -> packed 16-bit hi and lo parts, x ~= hi + lo: the operand planes of the attention products (three MFMA terms hi hi + hi lo + lo hi).
// ATT_F16_SPLIT (round 5, as qkv_attn_fused.hip): fp16 hi (11 significant bits) + fp16 lo, 22 bits in all, FOUR vector instructions per value pair
// (v_cvt_pk_f16_f32, two v_fma_mix_f32 that read the packed halves in place, v_cvt_pk_f16_f32); q / k / v are O(10) (LayerNorm'ed rows times weights), far
// inside the fp16 range (the conversions saturate: MODE.FP16_OVFL is set), a lo part below 2^-24 is lost -- less than a bf16 lo keeps of such a value.
// 0: bf16 hi + bf16 lo (16 bits, six instructions; rounds 1-4).
#ifndef ATT_F16_SPLIT
#define ATT_F16_SPLIT 1
#endif
__device__ __forceinline__ void att_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
#if ATT_F16_SPLIT
  uint32_t h, l;
  float l0, l1;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));      // (volatile: reads MODE, see mixed_pack4 in common.h)
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(b));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(l0), "v"(l1));
  hi = h;
  lo = l;
#else
  __bf16 h0, l0, h1, l1;
  split_bf16(a, h0, l0);
  split_bf16(b, h1, l1);
  hi = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
  lo = (uint32_t)__builtin_bit_cast(unsigned short, l0) | ((uint32_t)__builtin_bit_cast(unsigned short, l1) << 16);
#endif
}
typedef _Float16 att_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 att_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
#if ATT_F16_SPLIT
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(att_f16x8, a), __builtin_bit_cast(att_f16x8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
typedef short att_s16x4 __attribute__((ext_vector_type(4)));
typedef short att_s16x8 __attribute__((ext_vector_type(8)));

// TAB: layer 0 in the per-object form -- q / k / v of the 16 patch tokens never exist in memory: their chunks are formed on load
// from the two per-object table rows (L2 / Infinity Cache), the row's rstd and c2 (AttnArgs::sw ...).  (Giving each XCD a
// contiguous eighth of the items, so that an image's table rows meet in one L2, measured neutral.)
// F24: q / k / v arrive as 3-byte floats (common.h): a chunk of 8 values is 24 bytes, and hi + lo of such a value is exact.
template <int V> struct IntTag { static constexpr int value = V; };
template <int I, int N, class F>
__device__ __forceinline__ void att_static_for(F&& f) {
  if constexpr (I < N) {
    f(IntTag<I>());
    att_static_for<I + 1, N>(f);
  }
}
// waves (= items) per workgroup of attention_mfma_kernel
#ifndef ATT_WPB
#define ATT_WPB 2
#endif
#ifndef TAB_SPLIT_LOADS
#define TAB_SPLIT_LOADS 1      // 0: every chunk of the table form requested up front (round 4; two waves per SIMD)
#endif
// -DVETO_ATT_STAMPS: a diagnostic build that sums s_memtime deltas per phase over all waves (tools/att_stamps.py prints them): where an
// item's time goes -- issue of the gathers, the wait for their first use, conversion, S^T, softmax, V conversion, P V, the output stores
#ifdef VETO_ATT_STAMPS
__device__ unsigned long long g_att_stamps[8192 * 10];      // one slot per item (mod 8192): plain stores, no atomics in the timed kernel
#define ATT_T(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
                      __builtin_amdgcn_sched_barrier(0); att_dt[k] = t_ - att_t; att_t = t_; } while (0)
#else
#define ATT_T(k)
#endif
template <int DH, bool TAB = false, bool F24 = false>
__global__ __launch_bounds__(64 * ATT_WPB, TAB && (!TAB_SPLIT_LOADS || DH > 72) ? 2 : 3) void attention_mfma_kernel(AttnArgs a) {
  static_assert(!(TAB && F24), "the per-object form reads fp32 tables");
  saturating_conversions_on();   // (the mixed-row output path converts without clamps, common.h)
#ifdef VETO_ATT_STAMPS
  unsigned long long att_t, att_dt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(att_t)::"memory");
#endif
  constexpr int KS = (DH + 15) / 16;        // k-steps of QK^T (the upper half of the last one may lie behind the row: masked in registers)
  // bytes per row of the Q / K / V images (bf16; round 5: no contraction padding).  The pitch decides the bank conflicts of the fragment reads
  // (ds_read_b128: 16 rows per lane group, bank = (a / 4) mod 64): 144 B (DH = 72) puts the 16 rows of a group on 16 different bank quadruples;
  // round 4's 160 B (rows padded to 80 values) put them on 8, and 192 B (DH = 96) on 4: 16 bytes of padding there.
  constexpr int RB = DH * 2 + (DH % 32 == 0 ? 16 : 0);
  constexpr int NT = (DH + 31) / 32;        // 32-wide output tiles of PV
  constexpr int QK_PLANE = kTokens * RB;
  // LDS of a wave, in two phases (round 4): the four Q / K images; then, once S^T = K Q^T has been issued, the two V images and the fp32
  // output rows OVER them (V waits in registers meanwhile): the kernel is bound by the bytes it has in flight (measured at DH = 72 by padding
  // the LDS: 4 / 6 / 8 waves per CU = 2.55 / 1.97 / 1.72 ms for the step's three launches; nothing beyond 8).  Round 5: V is stored ROW-MAJOR
  // like Q and K (20 rows: row 19 is zero and stands in for keys 19..31) and read as the B operand of P V through ds_read_b64_tr_b16 in the
  // key order an accumulator-operand has (the scheme of qkv_attn_fused.hip).  Rounds 1-4 scattered a TRANSPOSED image with 16 two-byte
  // stores per chunk whose stride (320 B) put the lanes of a store on two banks: 54 % of the kernel's LDS cycles were bank conflicts
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r05_d_pmc_write.txt).
  constexpr int V_PLANE = (kTokens + 1) * RB;
  constexpr int O_BYTES = kTokens * DH * 4;
  constexpr int PHASE2 = 2 * V_PLANE + O_BYTES;
  // TAB (round 6): behind the images, the head's c2 (3 DH floats) and the pair's 19 rstd values, staged ONCE per item -- they were a global
  // load per chunk and per token of every round (27 vector-memory instructions per item on a kernel that is bound by exactly those)
  constexpr int kConstBytes = TAB ? (3 * DH * 4 + 32 * 4) : 0;
  constexpr int kImgBytes = ((4 * QK_PLANE > PHASE2 ? 4 * QK_PLANE : PHASE2) + 64 + 15) & ~15;      // (+64: over-reads behind the last row stay inside)
  constexpr int WAVE_LDS = kImgBytes + kConstBytes;
  constexpr int CH = DH / 8;                // 8-element chunks per row
  constexpr int PER_MAT = kTokens * CH;
  constexpr int ROUNDS = (3 * PER_MAT + 63) / 64;
  static_assert(DH % 8 == 0 && QK_PLANE % 16 == 0 && V_PLANE % 16 == 0, "layout");
  __shared__ __attribute__((aligned(16))) char smem[ATT_WPB * WAVE_LDS];

  // A wave works on its own LDS region: no workgroup barrier anywhere (LDS operations of one wave complete in order; the two
  // barriers of the first version cost 3 % of the launch).  Every wave has ONE item (persistent waves that prefetch their next
  // item's operands were measured 13 % slower: the dispatcher's refill of finished workgroups spreads the memory phases better
  // than waves that march in step).
  // (the wave index as a SCALAR: item, pair, head and the table rows of the pair's objects are then wave-uniform to the compiler -- scalar loads of the
  // indices, SGPR base + 32-bit lane offset for every gather instead of 64-bit vector address arithmetic per load)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long item = (long)blockIdx.x * ATT_WPB + w;
  if (item >= (long)a.n_pair * a.heads) return;
  char* base = smem + w * WAVE_LDS;
  char* q_hi = base;
  char* q_lo = base + QK_PLANE;
  char* k_hi = base + 2 * QK_PLANE;
  char* k_lo = base + 3 * QK_PLANE;
  char* v_hi = base;                   // (second phase)
  char* v_lo = v_hi + V_PLANE;

  // ---- global -> registers (all loads in flight), then -> bf16 hi/lo LDS images ----------------
  // TAB (round 6): the gathers go out in 16-byte PIECES, consecutive lanes on consecutive pieces of a row (a row of the head's slice is
  // DH / 4 pieces = 288 contiguous bytes): one wave instruction then touches ~11 cache lines.  Rounds 3-5 gave a lane a 32-byte chunk as
  // two loads (+0, +16): each of the two instructions touched the ~23 lines of all 64 chunks, and the in-kernel timeline
  // (profiles/r06_attention_stamps.txt) showed the kernel bound by the issue of its vector-memory instructions -- 8.3 k of an item's
  // 29.8 k cycles issuing 36 gathers, 10 k more in the conversion phase waiting for the per-chunk loads of c2 and rstd.
  constexpr int PC = DH / 4;                        // pieces per row
  constexpr int PER_MAT4 = kTokens * PC;
  constexpr int ROUNDS4 = (3 * PER_MAT4 + 63) / 64;
  f32x4 ld[TAB ? 1 : ROUNDS][2];      // (not TAB) a lane's 32-byte chunk of round r
  f32x4 lt[TAB ? ROUNDS4 : 1], lo_[TAB ? ROUNDS4 : 1];      // TAB: subject-side piece (or the plain row's), object-side piece
  // item -> (pair, head), head fastest: neighbouring waves read adjacent 288-byte slices of the SAME table rows.  ATT_PAIR_FASTEST=1 (TAB only;
  // measured SLOWER in round 6: 0.50 against 0.44 ms per launch): consecutive items = consecutive pairs of one head, i.e. the same subject-side
  // rows for a run of pairs but a head's slice alone of every row
#ifndef ATT_PAIR_FASTEST
#define ATT_PAIR_FASTEST 0
#endif
  const int pair = TAB && ATT_PAIR_FASTEST ? (int)(item % a.n_pair) : (int)(item / a.heads);
  const int head = TAB && ATT_PAIR_FASTEST ? (int)(item / a.n_pair) : (int)(item % a.heads);
  const float* src0 = a.qkv + (size_t)pair * kTokens * (3 * kDim) + head * DH;
  const float* tab_s = nullptr;
  const float* tab_o = nullptr;
  if constexpr (TAB) {
    tab_s = a.sw + (size_t)a.subj[pair] * kPatchTokens * (3 * kDim) + head * DH;
    tab_o = a.ow + (size_t)a.obj[pair] * kPatchTokens * (3 * kDim) + head * DH;
  }
  // TAB: two table rows per piece are twice the registers in flight, so the pieces go out in two groups: the rounds that hold a Q or K
  // piece now, the V-only rounds behind the Q / K conversion (their latency then lies under S^T and the softmax) -- the kernel runs three
  // waves per SIMD without spilling (round 4 had two).
  constexpr int kLateFrom = TAB ? (TAB_SPLIT_LOADS ? (2 * PER_MAT4 + 63) / 64 : ROUNDS4) : ROUNDS;     // first round without a Q / K piece
  constexpr int kRounds = TAB ? ROUNDS4 : ROUNDS;
  float* const c2_lds = (float*)(base + kImgBytes);                 // TAB: [3 DH] c2 of this head | [32] rstd of the pair's token rows
  float* const rs_lds = c2_lds + 3 * DH;
  if constexpr (TAB) {
    // (one or two vector-memory instructions for c2, 16 bytes per lane, and one for the 19 rstd values)
    constexpr int NC2 = 3 * DH / 4;      // 54 pieces (eight heads), 72 (six: a second, partial instruction)
#pragma unroll
    for (int idx0 = 0; idx0 < NC2; idx0 += 64) {
      const int idx = idx0 + lane;
      if (idx < NC2) {
        const int mat = idx / PC, c4 = idx % PC;
        *(f32x4*)(c2_lds + idx * 4) = *(const f32x4*)(a.vec + head * DH + mat * kDim + c4 * 4);
      }
    }
    if (lane < kTokens) rs_lds[lane] = a.stats[((size_t)pair * kTokens + lane) * 2 + 1];
  }
  auto load_round = [&](auto r_tag) {
    constexpr int r = decltype(r_tag)::value;
    const int e = lane + 64 * r;
    if constexpr (TAB) {
      const int mat = e / PER_MAT4, rem = e % PER_MAT4, i = rem / PC, c4 = rem % PC;
      const bool need = e < 3 * PER_MAT4;
      lo_[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (need && i >= 1 && i <= kPatchTokens) {
        // (32-bit byte offsets from wave-uniform bases: the saddr form of the loads, no 64-bit vector arithmetic)
        const unsigned off = (unsigned)(((i - 1) * (3 * kDim) + mat * kDim + c4 * 4) * 4);
        // (the store to lt[r] last, as in the other branches: the compiler merges the branches' trailing stores, and a merged store to
        // "lt[r] or lo_[r]" keeps both arrays in scratch memory)
        lo_[r] = *(const f32x4*)((const char*)tab_o + off);
        lt[r] = *(const f32x4*)((const char*)tab_s + off);
      } else if (need) {
        const unsigned off = (unsigned)((mat * kDim + c4 * 4) * 4) + (i == 0 ? 0u : (unsigned)(i * (3 * kDim) * 4));
        const char* src = i == 0 ? (const char*)(a.vec + 2 * (3 * kDim) + head * DH) : (const char*)src0;
        lt[r] = *(const f32x4*)(src + off);
      } else {
        lt[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
    const int mat = e / PER_MAT, rem = e % PER_MAT, i = rem / CH, c = rem % CH;
    const bool need = e < 3 * PER_MAT && !(a.cls_only && mat == 0 && i > 0);
    if (F24 && need) {
      // (the six dwords are kept packed in ld[r][0] / the first half of ld[r][1] until the conversion below)
      const char* src = (const char*)a.qkv + (((size_t)pair * kTokens + i) * (3 * kDim) + mat * kDim + head * DH + c * 8) * 3;
      // 24 bytes at an 8-byte-aligned address as 16 + 8 (two vector-memory instructions per chunk instead of three: 27 -> 18 load
      // instructions per lane; unaligned dwordx4 loads are legal on gfx950).  Measured neutral to slightly positive.
      typedef u32x4 u32x4_a8 __attribute__((aligned(8)));
      const u32x4 d01 = *(const u32x4_a8*)src;
      const u32x2 d2 = *(const u32x2*)(src + 16);
      ld[r][0] = f32x4{__uint_as_float(d01[0]), __uint_as_float(d01[1]), __uint_as_float(d01[2]), __uint_as_float(d01[3])};
      ld[r][1] = f32x4{__uint_as_float(d2[0]), __uint_as_float(d2[1]), 0.f, 0.f};
    } else if (need) {
      const unsigned off = (unsigned)((mat * kDim + c * 8) * 4) + (unsigned)(i * (3 * kDim) * 4);
      ld[r][0] = *(const f32x4*)((const char*)src0 + off);
      ld[r][1] = *(const f32x4*)((const char*)src0 + off + 16u);
    } else {
      ld[r][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      ld[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    }
  };
  att_static_for<0, kLateFrom>(load_round);
  ATT_T(0);      // index arithmetic + issue of the early gathers
  // Conversion of the loaded chunks / pieces into the 16-bit hi / lo images.  PHASE 0: the Q and K ones (and, TAB, the table arithmetic of
  // every early round's pieces, so that only its result stays in registers); PHASE 1, behind S^T: the V ones into the V images.
  auto convert = [&](auto phase_tag) {
    constexpr int PHASE = decltype(phase_tag)::value;
    if constexpr (TAB) {
      att_static_for<0, ROUNDS4>([&](auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        // (rounds that hold no piece of this phase; phase 0 visits the early rounds' V pieces too, for the table arithmetic)
        if constexpr (!(PHASE == 0 ? r >= kLateFrom : 64 * r + 63 < 2 * PER_MAT4)) {
          const int e = lane + 64 * r;
          const int mat = e / PER_MAT4, rem = e % PER_MAT4, i = rem / PC, c4 = rem % PC;
          if (e < 3 * PER_MAT4) {
            // (the table arithmetic of a round runs in the phase that first sees the round's data: phase 0 for the early rounds, phase 1 for the late)
            if ((PHASE == 0 ? r < kLateFrom : r >= kLateFrom) && i >= 1 && i <= kPatchTokens)      // rstd (SW + OW) + c2
              lt[r] = rs_lds[i] * (lt[r] + lo_[r]) + *(const f32x4*)(c2_lds + mat * DH + c4 * 4);
            if (PHASE == 0 ? mat < 2 : mat == 2) {
              uint32_t h0, l0, h1, l1;
              att_split2(lt[r][0], lt[r][1], h0, l0);
              att_split2(lt[r][2], lt[r][3], h1, l1);
              char* dst = (PHASE == 0 ? (mat == 0 ? q_hi : k_hi) : v_hi) + i * RB + c4 * 8;
              *(u32x2*)dst = u32x2{h0, h1};
              *(u32x2*)(dst + (PHASE == 0 ? QK_PLANE : V_PLANE)) = u32x2{l0, l1};
            }
          }
        }
      });
    } else {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (PHASE == 0 ? 64 * r >= 2 * PER_MAT : 64 * r + 63 < 2 * PER_MAT) continue;      // (rounds that hold no chunk of this phase)
      const int e = lane + 64 * r;
      if (e >= 3 * PER_MAT) continue;
      const int mat = e / PER_MAT, rem = e % PER_MAT, i = rem / CH, c = rem % CH;
      if (PHASE == 0 ? mat == 2 : mat < 2) continue;
      f32x4 v0 = ld[r][0], v1 = ld[r][1];
      if constexpr (F24) {      // (rows that were not read hold zeros, which unpack to zeros)
        v0 = unpack_f24x4(__float_as_uint(ld[r][0][0]), __float_as_uint(ld[r][0][1]), __float_as_uint(ld[r][0][2]));
        v1 = unpack_f24x4(__float_as_uint(ld[r][0][3]), __float_as_uint(ld[r][1][0]), __float_as_uint(ld[r][1][1]));
      }
      uint32_t h4[4], l4[4];
      att_split2(v0[0], v0[1], h4[0], l4[0]);
      att_split2(v0[2], v0[3], h4[1], l4[1]);
      att_split2(v1[0], v1[1], h4[2], l4[2]);
      att_split2(v1[2], v1[3], h4[3], l4[3]);
      const u32x4 hi = {h4[0], h4[1], h4[2], h4[3]}, lo = {l4[0], l4[1], l4[2], l4[3]};
      if (PHASE == 0) {
        char* dst = (mat == 0 ? q_hi : k_hi) + i * RB + c * 16;
        *(u32x4*)dst = hi;
        *(u32x4*)(dst + QK_PLANE) = lo;
      } else {
        char* dst = v_hi + i * RB + c * 16;
        *(u32x4*)dst = hi;
        *(u32x4*)(dst + V_PLANE) = lo;
      }
    }
    }
  };
#ifdef VETO_ATT_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_T(1);      // wait for the early gathers (all of them: the stamped build drains here)
#endif
  convert(IntTag<0>());
  ATT_T(2);      // table arithmetic + Q / K conversion + image stores
  att_static_for<kLateFrom, kRounds>(load_round);      // (TAB: the V-only rounds; in flight under S^T)
  ATT_T(3);      // issue of the late gathers

  // ---- S^T = K Q^T: row = key j, column = query i ----------------------------------------------
  const int r = lane & 31, h = lane >> 5;
  const int rr = r < kTokens ? r : 0;  // lanes beyond the 19 tokens re-read row 0; their outputs are masked
  f32x16 st;
#pragma unroll
  for (int t = 0; t < 16; ++t) st[t] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int off = rr * RB + (16 * s + 8 * h) * 2;
    u32x4 kh = *(const u32x4*)(k_hi + off), kl = *(const u32x4*)(k_lo + off);
    u32x4 qh = *(const u32x4*)(q_hi + off), ql = *(const u32x4*)(q_lo + off);
    if (16 * s + 8 >= DH) {      // the upper half of the last k-step lies behind the row: zeros (both operands: 0 x NaN is NaN)
      const u32x4 z = {0u, 0u, 0u, 0u};
      if (h) { kh = z; kl = z; qh = z; ql = z; }
    }
    st = att_mfma(kl, qh, st);
    st = att_mfma(kh, ql, st);
    st = att_mfma(kh, qh, st);
  }

  ATT_T(4);      // S^T (fragment reads + 15 / 18 MFMAs)
  // ---- softmax over the keys of query (lane & 31): 16 registers here + 16 in lane ^ 32 ---------
  const float scale = 1.0f / sqrtf((float)DH);
  float p[16];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int j = (t & 3) + 8 * (t >> 2) + 4 * h;
    p[t] = j < kTokens ? st[t] * scale : -INFINITY;
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
  u32x4 ph[2], pl[2];
#pragma unroll
  for (int t = 0; t < 16; t += 2) {
    uint32_t h2, l2;
    att_split2(p[t] * inv, p[t + 1] * inv, h2, l2);
    ph[t >> 3][(t & 7) >> 1] = h2;      // (constant indices: the loop is unrolled)
    pl[t >> 3][(t & 7) >> 1] = l2;
  }

  ATT_T(5);      // softmax + probability split
  // ---- the V^T images over the Q / K images: every ds_read of those has returned (each was waited for in front of its MFMA, and a
  // wave's LDS operations complete in order)
#ifdef VETO_ATT_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_T(6);      // wait for the late gathers
#endif
  convert(IntTag<1>());
  ATT_T(7);      // V table arithmetic + conversion + image stores
  if (lane < 2 * (RB / 16)) {       // image row 19: the zero row (keys 19..31)
    const int pln = lane >= RB / 16;
    *(u32x4*)(base + pln * V_PLANE + kTokens * RB + (lane - pln * (RB / 16)) * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- O = P V: A operand = P^T accumulators, whose element e of k-step s is key 16 s + 8 (e >> 2) + 4 h + (e & 3); the B operand is read
  // in that order from the row-major V images (transposed reads: lane 16 g + 4 q + p supplies row q of the 4, 8-byte piece p)
  float* o_lds = (float*)(base + 2 * V_PLANE);  // [19][DH] fp32, behind the V images
  {
    const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    int trow[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ta0 = 16 * s + 4 * h + tq, ta1 = ta0 + 8;
      trow[s][0] = (ta0 < kTokens ? ta0 : kTokens) * RB + (16 * (g4 & 1) + 4 * tp) * 2;
      trow[s][1] = (ta1 < kTokens ? ta1 : kTokens) * RB + (16 * (g4 & 1) + 4 * tp) * 2;
    }
    auto tr_pair = [](const char* p0, const char* p1) {
      const att_s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)p0);
      const att_s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)p1);
      return __builtin_bit_cast(u32x4, (att_s16x8)__builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int d = 32 * n + r;
      f32x16 o;
#pragma unroll
      for (int t = 0; t < 16; ++t) o[t] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4 vh = tr_pair(v_hi + trow[s][0] + 64 * n, v_hi + trow[s][1] + 64 * n);
        const u32x4 vl = tr_pair(v_lo + trow[s][0] + 64 * n, v_lo + trow[s][1] + 64 * n);
        o = att_mfma(pl[s], vh, o);
        o = att_mfma(ph[s], vl, o);
        o = att_mfma(ph[s], vh, o);
      }
      // register t holds query (t & 3) + 8 (t >> 2) + 4 h: t < 8 always a token, t = 8..10 only in the lower half wave (16..18), t >= 11 never
      if (32 * n + 32 <= DH || d < DH) {
        float* op = o_lds + 4 * h * DH + d;
#pragma unroll
        for (int t = 0; t < 8; ++t) op[((t & 3) + 8 * (t >> 2)) * DH] = o[t];
        if (!h) {
#pragma unroll
          for (int t = 8; t < 11; ++t) op[((t & 3) + 8 * (t >> 2)) * DH] = o[t];
        }
      }
    }
  }
  ATT_T(8);      // P V (transposed fragment reads + 18 MFMAs) + staging of the output rows
  const int nq = a.cls_only ? 1 : kTokens;
#pragma unroll
  for (int e0 = 0; e0 < (kTokens * CH + 63) / 64 * 64; e0 += 64) {     // (a fixed trip count: the compiler can then count the stores
    const int e = e0 + lane;                                            // that are younger than the prefetched loads)
    if (e >= nq * CH) continue;
    const int i = e / CH, c = e % CH;
    const f32x4 v0 = *(const f32x4*)(o_lds + i * DH + c * 8);
    const f32x4 v1 = *(const f32x4*)(o_lds + i * DH + c * 8 + 4);
    const size_t row = a.cls_only ? (size_t)pair : (size_t)pair * kTokens + i;
    if (a.o_fmt == FMT_MIXED) {   // wave-uniform
      store_act8_mixed<ACT8_NT != 0>(a.o + row * (2 * kDim), head * DH + c * 8, v0, v1);
      continue;
    }
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
    __bf16* dst = a.o + row * (2 * kDim) + split_index(head * DH + c * 8);
    *(bf16x8*)dst = hi;
    *(bf16x8*)(dst + 32) = lo;
  }
#ifdef VETO_ATT_STAMPS
  ATT_T(9);      // output conversion + stores (issue; the stores drain behind the wave's end)
  if (lane == 0) {
    unsigned long long* slot = g_att_stamps + (size_t)(item & 8191) * 10;
    for (int k = 0; k < 10; ++k) slot[k] = att_dt[k];
  }
#endif
}


// sum over the 16 lanes of a DPP row (row_ror 8 / 4 / 2 / 1): VALU-speed, no LDS crossbar; all lanes must be active
template <int CTRL>
__device__ __forceinline__ float dpp_rot16(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_rot16<0x128>(v);
  v += dpp_rot16<0x124>(v);
  v += dpp_rot16<0x122>(v);
  v += dpp_rot16<0x121>(v);
  return v;
}

// ---- last layer, folded form -----------------------------------------------------------------
// Only the CLS query of the last layer is consumed, and for ONE query the key / value projections can be moved off the 19
// tokens: with a_j = LN1(x_j) (the 576-vector of token j),
//   score[h][j] = q0_h . k_{j,h} = (W_k,h^T q0_h) . a_j = u_h . a_j,      u = a_0 . Mcat,  M_h = W_q,h^T W_k,h   (576 x 576)
//   out         = sum_h W_o,h (sum_j p[h][j] W_v,h a_j) = sum_h N_h abar_h,  abar_h = sum_j p[h][j] a_j,  N_h = W_o,h W_v,h
// so the [19 n_pair, 1152] key / value GEMM becomes two GEMMs over the n_pair CLS rows (u: K = 576, N = 576 H; out: K = 576 H,
// N = 576; M_h, N_h are products of weights, built once) around this kernel: per pair, scores of the H heads against the 19
// tokens, softmax, and the H probability-weighted token means.  One workgroup per pair; wave w owns heads w, w + 4, ...;
// lane l owns columns l + 64 i; the token vectors a_j = LN1(x_j) are computed here from the residual stream and staged in LDS.
constexpr int kFoldMaxHeads = 12;     // heads * 19 <= 256: one softmax element per thread

__global__ __launch_bounds__(256) void cls_fold_attention_kernel(const float* __restrict__ x, const float* __restrict__ ln_w,
                                                                 const float* __restrict__ ln_b, const float* __restrict__ u,
                                                                 __bf16* __restrict__ abar, int n_pair, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) float a_s[kTokens * kDim];
  __shared__ float s_row[kFoldMaxHeads][4][kTokens];
  __shared__ float p_s[kFoldMaxHeads][kTokens + 1];
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (pair >= n_pair) return;
  // a_j = LayerNorm1(x_j) of this pair's 19 token rows, computed here (the last layer has no LayerNorm launch of its own for
  // them): a quarter wave per row, 9 16-byte chunks per lane, two-pass statistics as in layernorm_kernel
  {
    const int q = tid & 15;
    const float* xp = x + (size_t)pair * kTokens * kDim;
    for (int j = tid >> 4; j < kTokens; j += 16) {
      f32x4 v[9];
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        v[i] = *(const f32x4*)(xp + (size_t)j * kDim + 4 * (q + 16 * i));
        sm += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      }
      const float mean = row16_sum(sm) * (1.f / kDim);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
      const float rstd = 1.f / sqrtf(row16_sum(sq) * (1.f / kDim) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int c = 4 * (q + 16 * i);
        const f32x4 g = *(const f32x4*)(ln_w + c), bb = *(const f32x4*)(ln_b + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
        *(f32x4*)(a_s + j * kDim + c) = o;
      }
    }
  }
  __syncthreads();
  const float* up = u + (size_t)pair * heads * kDim;
  // scores: a wave takes its heads two at a time (h, h + 4) so that a token value read from LDS serves both; partial dot
  // products over this lane's 9 columns, then all-reduced inside the 16-lane rows by DPP rotations
  for (int h = w; h < heads; h += 8) {
    const int h2 = h + 4 < heads ? h + 4 : h;      // no partner: head h twice, second copy not stored
    float acc[kTokens], acc2[kTokens];
#pragma unroll
    for (int j = 0; j < kTokens; ++j) { acc[j] = 0.f; acc2[j] = 0.f; }
#pragma unroll 1
    for (int i = 0; i < kDim / 64; ++i) {
      const int c = lane + 64 * i;
      const float uv = up[h * kDim + c], uv2 = up[h2 * kDim + c];
#pragma unroll
      for (int j = 0; j < kTokens; ++j) {
        const float av = a_s[j * kDim + c];
        acc[j] += uv * av;
        acc2[j] += uv2 * av;
      }
    }
#pragma unroll
    for (int j = 0; j < kTokens; ++j) {
      const float v = row16_sum(acc[j]), v2 = row16_sum(acc2[j]);
      if ((lane & 15) == 0) {
        s_row[h][lane >> 4][j] = v;
        if (h2 != h) s_row[h2][lane >> 4][j] = v2;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < heads * kTokens; e += 256) {     // fold the four row partials (fixed order)
    const int h = e / kTokens, j = e % kTokens;
    p_s[h][j] = ((s_row[h][0][j] + s_row[h][1][j]) + (s_row[h][2][j] + s_row[h][3][j])) * scale;
  }
  __syncthreads();
  float pmine = 0.f;
  const int sh = tid / kTokens, sj = tid % kTokens;        // softmax over the 19 tokens of a head (model_veto.py:91)
  if (tid < heads * kTokens) {                              // heads <= kFoldMaxHeads: one element per thread
    float mx = -INFINITY;
#pragma unroll 1
    for (int t = 0; t < kTokens; ++t) mx = fmaxf(mx, p_s[sh][t]);
    float sum = 0.f;
#pragma unroll 1
    for (int t = 0; t < kTokens; ++t) sum += expf(p_s[sh][t] - mx);
    pmine = expf(p_s[sh][sj] - mx) / sum;
  }
  __syncthreads();
  if (tid < heads * kTokens) p_s[sh][sj] = pmine;
  __syncthreads();
  __bf16* dst = abar + (size_t)pair * (2 * (size_t)heads * kDim);
  for (int h = w; h < heads; h += 8) {      // abar_h = sum_j p[h][j] a_j, written as split rows (the A operand of the out GEMM)
    const int h2 = h + 4 < heads ? h + 4 : h;
    float pj[kTokens], pj2[kTokens];
#pragma unroll
    for (int j = 0; j < kTokens; ++j) { pj[j] = p_s[h][j]; pj2[j] = p_s[h2][j]; }
#pragma unroll 1
    for (int i = 0; i < kDim / 64; ++i) {
      const int c = lane + 64 * i;
      float o = 0.f, o2 = 0.f;
#pragma unroll
      for (int j = 0; j < kTokens; ++j) {
        const float av = a_s[j * kDim + c];
        o += pj[j] * av;
        o2 += pj2[j] * av;
      }
      __bf16 hh, ll;
      split_bf16(o, hh, ll);
      __bf16* d = dst + split_index(h * kDim + c);
      d[0] = hh;
      d[32] = ll;
      if (h2 != h) {
        split_bf16(o2, hh, ll);
        d = dst + split_index(h2 * kDim + c);
        d[0] = hh;
        d[32] = ll;
      }
    }
  }
}

// The same kernel with both contractions on the matrix cores (round 3; the VALU form above took 0.17 ms for the scores -- a DPP
// reduction tree per (head, token) behind the FMAs -- and 0.09 ms for the weighted means of a 0.39 ms launch).  The token vectors
// a_j go to LDS as bf16 hi / lo images (row = token, 20 rows: row 19 is zero and stands in for tokens 19..31; the row pitch of
// 1168 bytes puts the 16 rows of a fragment read on 16 different bank quadruples), both products are 16x16x32 MFMAs in the 3-term
// split-bf16 scheme of the GEMMs:
//   scores[h][j] = sum_c u[h][c] a[j][c]   A operand = u (lane (m, g): head m, 8 columns at 32 s + 8 g, read from memory as it is,
//                                          heads >= H are zero rows), B operand = the image rows (ds_read_b128), k-steps s dealt
//                                          over the 4 waves, partial tiles summed through LDS in a fixed order
//   abar[h][c]   = sum_j p[h][j] a[j][c]   A operand = p (one k-step: 32 token slots), B operand = the SAME images read through
//                                          ds_read_b64_tr_b16 (token = contraction index: 4 rows x 16 columns per 16-lane group,
//                                          delivered column-major), 36 column tiles dealt over the waves
// The result tiles are staged through LDS (over the images, which are dead by then) so that abar leaves as 128-byte runs.
constexpr int kFoldPitch = kDim * 2 + 16;            // bytes per image row
constexpr int kFoldImg = 20 * kFoldPitch;            // one plane (hi or lo)
typedef short cf_s16x4 __attribute__((ext_vector_type(4)));
typedef short cf_s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    __bf16 hh, ll;
    split_bf16(t < 4 ? v0[t] : v1[t - 4], hh, ll);
    hi[t] = hh;
    lo[t] = ll;
  }
}

#ifdef VETO_CLS_STAMPS
__device__ unsigned long long g_cls_stamps[16];
#define CST(k) do { if (blockIdx.x == 5000 && tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_cls_stamps[k] = t_; } } while (0)
#else
#define CST(k)
#endif
template <bool U24>     // U24: u arrives as 3-byte floats (common.h), rows of 3 * heads * 576 bytes
__global__ __launch_bounds__(256, 3) void cls_fold_attention_mfma_kernel(const float* __restrict__ x, const float* __restrict__ ln_w,
                                                                      const float* __restrict__ ln_b, const float* __restrict__ u,
                                                                      __bf16* __restrict__ abar, int n_pair, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) char img[2 * kFoldImg];      // a_j as bf16 hi | lo; later the fp32 staging of abar
  __shared__ float s_part[4][kFoldMaxHeads][20];                        // per-wave partial score tiles [head][token] (the valid part)
  __shared__ float p_s[16][32];                                         // probabilities, zero outside [heads][19]
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (pair >= n_pair) return;
  CST(0);
  char* a_hi = img;
  char* a_lo = img + kFoldImg;
  const int fm = lane & 15, fg = lane >> 4;           // MFMA operand lane: row / column fm, k group fg
  // ---- every global read of the workgroup up front: the token rows 0..15 (a quarter wave per row, 9 16-byte chunks per lane), the
  // rows 16..18 (one per wave 0..2, lane l takes the columns l + 64 i), and this wave's k-steps of u in operand layout
  const int q = tid & 15, j0 = tid >> 4;
  const float* xp = x + (size_t)pair * kTokens * kDim;
  f32x4 vrow[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
#if CLS_NT & 1
    vrow[i] = __builtin_nontemporal_load((const f32x4*)(xp + (size_t)j0 * kDim + 4 * (q + 16 * i)));
#else
    vrow[i] = *(const f32x4*)(xp + (size_t)j0 * kDim + 4 * (q + 16 * i));
#endif
  }
  float vx[9];
  if (16 + w < kTokens) {
#pragma unroll
    for (int i = 0; i < 9; ++i) vx[i] = xp[(size_t)(16 + w) * kDim + lane + 64 * i];
  }
  constexpr int KS = kDim / 32;                       // 18 k-steps of the score product
  f32x4 uf[(KS + 3) / 4][2];
  const size_t uoff = (size_t)pair * heads * kDim + (size_t)(fm < heads ? fm : 0) * kDim + 8 * fg;     // in elements
  const float* up = u + uoff;
#pragma unroll
  for (int i = 0; i < (KS + 3) / 4; ++i) {
    const int sidx = w + 4 * i;
    if (U24 && sidx < KS && fm < heads) {       // 8 values = 24 bytes, kept packed until the operand is formed
      const char* src = (const char*)u + (uoff + 32 * sidx) * 3;
      const u32x2 d0 = *(const u32x2*)src, d1 = *(const u32x2*)(src + 8), d2 = *(const u32x2*)(src + 16);
      uf[i][0] = f32x4{__uint_as_float(d0[0]), __uint_as_float(d0[1]), __uint_as_float(d1[0]), __uint_as_float(d1[1])};
      uf[i][1] = f32x4{__uint_as_float(d2[0]), __uint_as_float(d2[1]), 0.f, 0.f};
    } else if (sidx < KS && fm < heads) {
      uf[i][0] = *(const f32x4*)(up + 32 * sidx);
      uf[i][1] = *(const f32x4*)(up + 32 * sidx + 4);
    } else {
      uf[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      uf[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  CST(1);
  for (int i = tid; i < 16 * 32; i += 256) (&p_s[0][0])[i] = 0.f;
  for (int i = tid; i < kFoldPitch / 4; i += 256) {            // the zero row of both planes
    *(uint32_t*)(a_hi + 19 * kFoldPitch + 4 * i) = 0u;
    *(uint32_t*)(a_lo + 19 * kFoldPitch + 4 * i) = 0u;
  }
  CST(2);
  // ---- a_j = LayerNorm1(x_j) (two-pass statistics as in layernorm_kernel) -> the images
  {
    const int j = j0;
    f32x4 (&v)[9] = vrow;
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) sm += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = row16_sum(sm) * (1.f / kDim);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    const float rstd = 1.f / sqrtf(row16_sum(sq) * (1.f / kDim) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int c = 4 * (q + 16 * i);
      const f32x4 g = *(const f32x4*)(ln_w + c), bb = *(const f32x4*)(ln_b + c);
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 hh, ll;
        split_bf16((v[i][e] - mean) * rstd * g[e] + bb[e], hh, ll);
        hi[e] = hh;
        lo[e] = ll;
      }
      *(bf16x4*)(a_hi + j * kFoldPitch + c * 2) = hi;
      *(bf16x4*)(a_lo + j * kFoldPitch + c * 2) = lo;
    }
  }
  if (16 + w < kTokens) {      // wave-uniform: rows 16..18 on a whole wave each
    const int j = 16 + w;
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) sm += vx[i];
    const float mean = wave_sum(sm) * (1.f / kDim);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) { const float d = vx[i] - mean; sq += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(sq) * (1.f / kDim) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int c = lane + 64 * i;
      __bf16 hh, ll;
      split_bf16((vx[i] - mean) * rstd * ln_w[c] + ln_b[c], hh, ll);
      *(__bf16*)(a_hi + j * kFoldPitch + c * 2) = hh;
      *(__bf16*)(a_lo + j * kFoldPitch + c * 2) = ll;
    }
  }
  CST(3);
  __syncthreads();
  CST(4);
  // ---- scores: this wave's k-steps, two token tiles
  {
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < (KS + 3) / 4; ++i) {
      const int sidx = w + 4 * i;
      if (sidx >= KS) break;                                   // wave-uniform
      bf16x8 uh, ul;
      if constexpr (U24) {      // (lanes without a head hold zeros, which unpack to zeros)
        const f32x4 p0 = uf[i][0], p1 = uf[i][1];
        split8(unpack_f24x4(__float_as_uint(p0[0]), __float_as_uint(p0[1]), __float_as_uint(p0[2])),
               unpack_f24x4(__float_as_uint(p0[3]), __float_as_uint(p1[0]), __float_as_uint(p1[1])), uh, ul);
      } else {
        split8(uf[i][0], uf[i][1], uh, ul);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = 16 * t + fm < kTokens ? 16 * t + fm : kTokens;      // tokens 19..31: the zero row
        const int off = row * kFoldPitch + (32 * sidx + 8 * fg) * 2;
        const bf16x8 bh = *(const bf16x8*)(a_hi + off), bl = *(const bf16x8*)(a_lo + off);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ul, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uh, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uh, bh, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)                                                     // D[row 4 g + e = head][column fm = token]
        if (4 * fg + e < kFoldMaxHeads && 16 * t + fm < 20) s_part[w][4 * fg + e][16 * t + fm] = acc[t][e];
  }
  CST(5);
  __syncthreads();
  CST(6);
  // softmax over the 19 tokens of a head (model_veto.py:91): a half wave per head (lanes 19..31 of a half idle), wave w takes the
  // heads 2 w, 2 w + 1, then 2 w + 8, 2 w + 9; the four partial tiles are summed in a fixed order
  for (int h0 = 2 * w; h0 < heads; h0 += 8) {
    const int sh = h0 + (lane >> 5), sj = lane & 31;
    const bool valid = sh < heads && sj < kTokens;
    const int rh = valid ? sh : 0, rj = valid ? sj : 0;
    const float sc = valid ? ((s_part[0][rh][rj] + s_part[1][rh][rj]) + (s_part[2][rh][rj] + s_part[3][rh][rj])) * scale : -INFINITY;
    float mx = sc;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float ev = valid ? expf(sc - mx) : 0.f;
    float sum = ev;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (valid) p_s[sh][sj] = ev / sum;
  }
  CST(7);
  __syncthreads();
  // ---- abar: this wave's column tiles c = w, w + 4, ...; the probabilities are the A operand (one k-step of 32 token slots)
  CST(8);
  constexpr int NTILE = kDim / 16;                     // 36
  f32x4 out[NTILE / 4];
  {
    bf16x8 ph, pl;
    split8(*(const f32x4*)&p_s[fm][8 * fg], *(const f32x4*)&p_s[fm][8 * fg + 4], ph, pl);
    const int tq = (lane >> 2) & 3, tp = lane & 3;     // transposed read: this lane addresses row 8 g + q (+ 4), 8-byte piece p
    const int r0 = 8 * fg + tq < kTokens ? 8 * fg + tq : kTokens, r1 = 8 * fg + tq + 4 < kTokens ? 8 * fg + tq + 4 : kTokens;
    const int o0 = r0 * kFoldPitch + 8 * tp, o1 = r1 * kFoldPitch + 8 * tp;
    auto frag = [&](const char* plane, int c) {
      const cf_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cf_s16x4*)(plane + o0 + 32 * c));
      const cf_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cf_s16x4*)(plane + o1 + 32 * c));
      return __builtin_bit_cast(bf16x8, (cf_s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (int i = 0; i < NTILE / 4; ++i) {
      const int c = w + 4 * i;
      const bf16x8 bh = frag(a_hi, c), bl = frag(a_lo, c);
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, bh, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, bl, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, bh, o, 0, 0, 0);
      out[i] = o;
    }
  }
  CST(9);
  __syncthreads();                                     // every wave is done with the images: the staging goes over them
  CST(10);
  float* st = (float*)img;                             // [heads][576] fp32
#pragma unroll
  for (int i = 0; i < NTILE / 4; ++i) {
    const int c = w + 4 * i;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * fg + e < heads) st[(4 * fg + e) * kDim + 16 * c + fm] = out[i][e];
  }
  __syncthreads();
  CST(11);
  __bf16* dst = abar + (size_t)pair * (2 * (size_t)heads * kDim);
  for (int e = 4 * tid; e < heads * kDim; e += 4 * 256) {      // split rows (the A operand of the GEMM behind), 8 bytes of hi + 8 of lo per thread
    const f32x4 v = *(const f32x4*)(st + e);
    bf16x4 hi, lo;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      __bf16 hh, ll;
      split_bf16(v[t], hh, ll);
      hi[t] = hh;
      lo[t] = ll;
    }
    __bf16* d = dst + split_index(e);
#if CLS_NT & 2
    __builtin_nontemporal_store(hi, (bf16x4*)d);
    __builtin_nontemporal_store(lo, (bf16x4*)(d + 32));
#else
    *(bf16x4*)d = hi;
    *(bf16x4*)(d + 32) = lo;
#endif
  }
  CST(12);
}

}  // namespace

int cls_fold_max_heads() { return kFoldMaxHeads; }

bool cls_fold_reads_f24() {
  static const bool valu = env_knob_is("VETO_CLS_MFMA", "0");
  static const bool f32 = env_knob_is("VETO_QKV_F24", "0");
  return !valu && !f32;
}

hipError_t launch_cls_fold_attention(const float* x, const float* ln_w, const float* ln_b, const float* u, __bf16* abar, int n_pair, int heads,
                                     hipStream_t s, bool u_f24) {
  if (heads <= 0 || heads > kFoldMaxHeads || kDim % heads != 0 || n_pair <= 0) return hipErrorInvalidValue;
  const float scale = 1.0f / sqrtf((float)(kDim / heads));
  static const bool valu = env_knob_is("VETO_CLS_MFMA", "0");     // A/B knob of the parity tests: the fp32 VALU form
  if (u_f24 && valu) return hipErrorInvalidValue;
  if (valu) VETO_LAUNCH(cls_fold_attention_kernel, dim3(n_pair), dim3(256), 0, s, x, ln_w, ln_b, u, abar, n_pair, heads, scale);
  else if (u_f24) VETO_LAUNCH(cls_fold_attention_mfma_kernel<true>, dim3(n_pair), dim3(256), 0, s, x, ln_w, ln_b, u, abar, n_pair, heads, scale);
  else VETO_LAUNCH(cls_fold_attention_mfma_kernel<false>, dim3(n_pair), dim3(256), 0, s, x, ln_w, ln_b, u, abar, n_pair, heads, scale);
#ifdef VETO_CLS_STAMPS
  {
    static int printed = 0;
    if (printed++ == 3) {
      unsigned long long hst[16];
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(hst, HIP_SYMBOL(g_cls_stamps), sizeof(hst));
      fprintf(stderr, "[cls stamps]");
      for (int k = 1; k <= 12; ++k) fprintf(stderr, " %lld", (long long)(hst[k] - hst[k - 1]));
      fprintf(stderr, "  (request | zero | loads->LN start.. | LN+images | barrier | scores | barrier | softmax | barrier | abar | barrier | staging | stores)\n");
    }
  }
#endif
  return hipGetLastError();
}

bool attention_reads_tables(int heads) {
  if (heads <= 0 || kDim % heads != 0) return false;
  const int dh = kDim / heads;
  return dh == 72 || dh == 96;
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
  if (kDim % a.heads != 0) return hipErrorInvalidValue;
  const int dh = kDim / a.heads;
  if (dh == 72 || dh == 96) {
    const long items = (long)a.n_pair * a.heads;
    unsigned blocks = (unsigned)((items + ATT_WPB - 1) / ATT_WPB);
    if (a.sw) {
      if (!a.ow || !a.stats || !a.vec || !a.subj || !a.obj || a.cls_only) return hipErrorInvalidValue;
      if (dh == 72) VETO_LAUNCH((attention_mfma_kernel<72, true>), dim3(blocks), dim3(64 * ATT_WPB), 0, s, a);
      else VETO_LAUNCH((attention_mfma_kernel<96, true>), dim3(blocks), dim3(64 * ATT_WPB), 0, s, a);
#ifdef VETO_ATT_STAMPS
      {
        static unsigned long long all[8192 * 10];
        unsigned long long host[10] = {0};
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(all, HIP_SYMBOL(g_att_stamps), sizeof(all));
        const long nslots = items < 8192 ? items : 8192;
        for (long i = 0; i < nslots; ++i)
          for (int k = 0; k < 10; ++k) host[k] += all[i * 10 + k];
        const double n = (double)nslots;
        static const char* names[10] = {"issue of the early gathers", "wait for the early gathers", "table arithmetic + Q / K conversion", "issue of the late gathers",
                                        "S^T", "softmax", "wait for the late gathers", "V arithmetic + conversion", "P V + staging", "output conversion + stores"};
        double tot = 0;
        for (int k = 0; k < 10; ++k) tot += host[k] / n;
        fprintf(stderr, "[attention stamps, table form, dh %d, last %.0f items] mean shader-clock cycles per item (s_memtime) and share:\n", dh, n);
        for (int k = 0; k < 10; ++k) fprintf(stderr, "    %-38s %8.1f  %5.1f %%\n", names[k], host[k] / n, 100.0 * host[k] / n / tot);
        fprintf(stderr, "    %-38s %8.1f\n", "sum", tot);
      }
#endif
    } else if (a.qkv_f24) {
      if (dh == 72) VETO_LAUNCH((attention_mfma_kernel<72, false, true>), dim3(blocks), dim3(64 * ATT_WPB), 0, s, a);
      else VETO_LAUNCH((attention_mfma_kernel<96, false, true>), dim3(blocks), dim3(64 * ATT_WPB), 0, s, a);
    } else if (dh == 72) VETO_LAUNCH(attention_mfma_kernel<72>, dim3(blocks), dim3(64 * ATT_WPB), 0, s, a);
    else VETO_LAUNCH(attention_mfma_kernel<96>, dim3(blocks), dim3(64 * ATT_WPB), 0, s, a);
    return hipGetLastError();
  }
  if (a.sw || a.qkv_f24) return hipErrorInvalidValue;
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
