// QKV projection + per-pair multi-head attention of a middle transformer layer in ONE launch (model_veto.py:78-96):
//
//   q | k | v = LN1(x) Wqkv^T            (to_qkv has no bias, model_veto.py:80)
//   out       = softmax(q k^T dh^-0.5) v  per (pair, head) over the pair's 19 tokens, heads merged 'b h n d -> b n (h d)'
//
// q / k / v never exist in memory.  The launch-per-stage form wrote them as a 1.5 GB matrix of 3-byte floats (gemm_split_ps.hip,
// EPI_F24) that the attention launch (attention.hip) read straight back: two launches, each bound by moving those bytes.
//
// Tile = 16 pairs x ONE head: 304 token rows (exactly 19 MFMA row blocks, pair-aligned) x that head's q | k | v columns (3 DH =
// 216 for 8 heads, 288 for 6).  One persistent 8-wave workgroup per CU (two waves per SIMD, <= 256 registers) walks its tiles:
//
//   main loop    18 stages (9 blocks of 64 k's x {fp16 part, e4m3 part} of the VETO_MIXED operands, common.h); a stage is 128 bytes
//                of every row: 38 KiB of activation rows + 27 / 36 KiB of weight rows, LDS-DMA'd (global_load_lds_dwordx4) with the
//                source-side XOR swizzle of gemm_split_ps.hip.  No loader waves (twelve waves would cap the kernel at 168 registers,
//                and the [304 x 3 DH] result needs 140 / 180 accumulator registers per lane): the waves issue the DMA themselves, and
//                SIMD partners take turns -- waves 0-3 issue their share in front of their MFMA groups, waves 4-7 behind them (DMA
//                side, below).  Two stages of 65 KiB are all the LDS holds, and a stage that is issued only when its slot is
//                free gets ONE interval (~1 us) of flight against ~1.6 us from issue to landing with every CU streaming (measured:
//                1 000 cycles of exposed wait per stage).  So the buffers are TWO activation images + THREE weight images, and an
//                activation image is released EARLY: every wave holds its five activation fragments in registers for the whole
//                stage, so behind a second barrier right after those reads the image of stage s is dead and receives stage s + 2
//                while stage s is still being multiplied; a weight image (read fragment by fragment through the stage) is one of
//                three and receives stage s + 2 too.  Everything gets two intervals of flight.  (Six heads: 36 KiB weight images,
//                two of them: W(s + 1) has one interval.)
//                Wave (wm, wn) of 4 x 2 owns row blocks 5 wm .. 5 wm + 4 (block 19 does not exist: wm = 3 multiplies a dummy whose
//                result is never read -- its SIMD would idle otherwise) and column blocks NB wn .. NB wn + NB - 1 (QA_COL_INTERLEAVE: 2 n + wn).
//   attention    behind the last stage the buffers are dead, except the first activation / weight image, which already holds the
//                NEXT tile's first stage where the geometry allows: the waves convert their accumulators to packed fp16 hi / lo once,
//                then twice (pairs in two batches of 8): the owners write the batch's Q / K rows into eight per-pair LDS regions in
//                the image layout of attention_mfma_kernel, barrier, wave w runs that kernel's S^T = K Q^T + softmax on region w,
//                barrier, the owners write the V rows over the Q / K images, barrier, wave w runs P V (V as the B operand through
//                ds_read_b64_tr_b16: token = contraction index), stages the [19 x DH] result in its region and stores it as mixed
//                rows: the operand of the layer tail's out projection (ffn_fused.hip).
//
// Tile order: workgroup b lives on XCD b % 8 (round-robin dispatch; speed only) and in round i takes tile
// ((8 i + b % 8) * 32 + b / 8), head fastest: the 32 workgroups of an XCD work on 4 pair groups x all heads at a time -- every
// activation row is fetched once per XCD round (the other seven heads' workgroups hit L2), and the whole weight matrix (4 MB as
// mixed rows) is what the L2 then holds.
//
// Results depend on the pair alone (fixed k order, no split-K): bit-identical under batch / chunk / permutation changes.
#include "common.h"
#include "kernels.h"

#include <cstdio>
#include <cstdlib>

// timing ablations (tools/variants.sh; results are WRONG with any of them set): 1 = no DMA, 2 = no attention phase (nothing is stored),
// 32 = no scores / softmax, 64 = no P V and no output, 128 = no Q / K / V row writes, 256 = no accumulator conversion (attention-phase ablations),
// 4 = no MFMAs in the main loop, 8 = no output stores, 16 = no fragment reads in the main loop (the MFMAs run on whatever the registers hold)
#ifndef QA_ABLATE
#define QA_ABLATE 0
#endif
// the wait states in front of every inline-asm MFMA (see mma() below); -DQA_NO_PADS builds the kernel WITHOUT them: the negative control of
// tests/test_ffn_asm.py
#ifdef QA_NO_PADS
#define QA_MMA_NOP ""
#endif
#ifndef QA_MMA_NOP
#define QA_MMA_NOP "s_nop 1\n\t"
#endif
// activation pieces the lower half issues beside the weight pieces (three weight images; measured: 4 / 10 / 16 within 1 % of one another,
// 0 -- the upper half issues every activation piece -- 6 % slower)
#ifndef QA_ASPLIT
#define QA_ASPLIT 10
#endif
// weight-fragment buffers of the main loop: fragment n + QA_WBUF - 1 is requested while fragment n is multiplied (3 measured equal to 2:
// 1.315 vs 1.304-1.314 ms; six heads have no registers for a third buffer)
#ifndef QA_WBUF
#define QA_WBUF 2
#endif
// column blocks of the two wave columns: 0 = contiguous halves (wn = 0: q and the first part of k), 1 = interleaved (wave column wn owns the
// blocks 2 n + wn: both columns hold a share of q, k AND v, so the row writes of the attention phase are spread over all eight waves --
// measured 1.340-1.347 against 1.325-1.338 ms: the unbalanced row writes are not what the phase waits for)
// wave priority in the main loop: 1 = the upper half (which multiplies first and issues its DMA share behind its MFMA groups) runs at priority 1,
// i.e. wins the matrix pipe of its SIMD whenever both partners want it: it is through with its MFMAs early and its DMA issue falls under the
// lower half's matrix work instead of behind the stage (1.27-1.29 -> 1.21-1.24 ms; level 3 equal; priority only while it multiplies 1.25; kept
// through the attention phase 1.25); 2 = the lower half raised instead (null: 1.28); 0 = no priorities
#ifndef QA_PRIO
#define QA_PRIO 1
#endif
// row blocks of the four wave rows and the pairs of a batch: 1 = wave row wm owns the blocks wm, wm + 4, ... and batch bt holds the pairs 8 bt .. 8 bt + 7
// (a block's rows then belong to ONE batch, but for the block that holds the batch boundary: 20 block visits of row writes per tile instead of 26, all eight
// waves in every batch); 0 = contiguous blocks 5 wm .., pairs alternating between the batches two by two
#ifndef QA_ROW_INTERLEAVE
#define QA_ROW_INTERLEAVE 1
#endif
#ifndef QA_PITCH_PAD
#define QA_PITCH_PAD 16
#endif
#ifndef QA_COL_INTERLEAVE
#define QA_COL_INTERLEAVE 0
#endif
// TIMING PROBE (results are WRONG): the correction stages as block-scaled fp6 (e2m3) operands -- the K = 128 MFMA in its fp6 form, three
// quarters of the pieces of a correction stage through the LDS-DMA, stand-ins for the block-scale traffic (ffn_fused.hip, FFN_FP6_PROBE)
#ifndef QA_FP6_PROBE
#define QA_FP6_PROBE 0
#endif

namespace veto {

namespace {

constexpr int TP = 16;                     // pairs per tile
constexpr int TM = TP * kTokens;           // 304 rows = 19 row blocks of 16
constexpr int MB = 5;                      // row blocks per wave
constexpr int kRowB = kDim * 4;            // bytes of a mixed row (K = 576)
// (18 stages per tile: kStages inside the kernel)
constexpr int kABytes = TM * 128;          // activation part of a ring slot
constexpr int APIECES = TM / 8;            // 38 LDS-DMA instructions (8 rows x 128 B each) per activation stage
constexpr int kLdsTotal = 163840;
constexpr int VROW = 40;                   // bytes per row of a transposed V image: 20 keys (19 + one zero)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int V> struct Tag { static constexpr int value = V; };

template <int DH>
struct Geo {
  static constexpr int NCOL = 3 * DH;                         // q | k | v columns of one head
  static constexpr int NB = (NCOL + 31) / 32;                 // column blocks per wave (2 waves along N): 7 / 9
  static constexpr int WROWS = 2 * NB * 16;                   // weight rows of a stage image (the last ones may be padding)
  static constexpr int WPIECES = NCOL / 8;                    // 27 / 36
  static constexpr int PM = DH / 8;                           // pieces per matrix (a piece never straddles q / k / v)
  static constexpr int kWBytes = WROWS * 128;
  // buffers: A0 | W0 | A1 | W1 | (W2)
  static constexpr int NWB = (kLdsTotal - 2 * kABytes) / kWBytes >= 3 ? 3 : 2;
  static constexpr int kPair = kABytes + kWBytes;
  static constexpr int a_buf(int i) { return i * kPair; }
  static constexpr int w_buf(int j) { return j < 2 ? kABytes + j * kPair : 2 * kPair; }
  // attention regions (one per wave; two phases as in attention_mfma_kernel): the four Q / K images, then V hi / lo (20 rows: row 19 is
  // zero and stands in for keys 19..31) + fp32 output rows
  // bytes per image row (no contraction padding: masked in registers).  A pitch that is a multiple of 64 bytes puts the 16 rows a half wave
  // writes or reads on only 4 bank groups: six heads (192 B) get 16 bytes of padding (1.35 -> 1.22 ms per launch)
  static constexpr int QP = DH * 2 + (DH % 32 == 0 ? QA_PITCH_PAD : 0);
  static constexpr int PLANE = kTokens * QP;
  static constexpr int VPLANE = (kTokens + 1) * QP;
  static constexpr int PH1 = 4 * PLANE, PH2 = 2 * VPLANE + kTokens * DH * 4;
  static constexpr int REGION = ((PH1 > PH2 ? PH1 : PH2) + 64 + 15) & ~15;      // (+64: over-reads behind the last row of an image stay inside)
  static constexpr int kScratch = (kLdsTotal - 8 * REGION) & ~15;   // the regions end at the end of the LDS
  static constexpr bool kFreeW0 = kScratch >= kPair;          // the first weight image stays free during the attention phase
  static_assert(DH % 8 == 0 && NWB * kWBytes + 2 * kABytes <= kLdsTotal, "LDS budget");
  // pieces (8 rows) of the first activation image that stay free during the attention phase: the next tile's stage 0 lands there meanwhile;
  // the pieces behind them (six heads: the last two, 2 KiB) are issued behind the attention phase like the first weight image
  static constexpr int kAFree = kScratch >= kABytes ? APIECES : kScratch / 1024;
  static_assert(kScratch % 1024 == 0 || kScratch >= kABytes, "the regions start on a piece boundary");
  static_assert(kAFree >= APIECES - 4, "only a few pieces of the next tile's first stage wait for the attention phase");
  static_assert(a_buf(1) + (MB * 4) * 2048 <= kLdsTotal, "the dummy row block of wm = 3 reads inside the LDS");
};

// One LDS-DMA instruction: 64 lanes x 16 bytes from (uniform base + 32-bit lane offset) to LDS address m0 + 16 * lane (inline asm as in
// ffn_fused.hip: no vector address arithmetic, and invisible to the compiler's wait-count pass, which would otherwise drain vmcnt in
// front of every later ds_read of the issuing wave; the kernel counts its own vmcnt).
__device__ __forceinline__ void glds16(const char* base, unsigned voff, unsigned lds_addr) {
  if (QA_ABLATE & 1) return;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// publishes this wave's LDS stores to the workgroup, then waits for everybody's
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wg_barrier();
}

// Four fp32 values -> bf16 hi and bf16 lo (x ~= hi + lo, common.h), both PACKED: {v0 | v1 << 16, v2 | v3 << 16}.  Written on dwords and
// pinned by an empty asm: left to the vector types, the compiler keeps every 16-bit element of the 8 x 35 values in a register of its own
// next to the packed form (the element-wise V^T stores below index them), and spills a few hundred registers.
// Two fp32 values -> packed 16-bit hi and lo parts, x ~= hi + lo (the operands of the attention products: three MFMA terms hi hi + hi lo +
// lo hi).  QA_F16_SPLIT (default): fp16 hi (11 significant bits) + fp16 lo: 22 bits in all, FOUR vector instructions per value pair
// (v_cvt_pk_f16_f32, two v_fma_mix_f32 that read the packed halves in place, v_cvt_pk_f16_f32) -- the attention phase is bound by vector
// instruction issue.  q / k / v are O(10) here (LayerNorm'ed rows times weights), far inside the fp16 range (the conversions saturate at
// 65 504: MODE.FP16_OVFL is set), and a lo part below 2^-14 loses at most what a bf16 lo would have lost.  0: bf16 hi + bf16 lo (16 bits,
// six instructions; the split of attention.hip).  The packed value is pinned by an empty asm: the compiler otherwise converts the first value
// a second time, alone, to get at its half.
#ifndef QA_F16_SPLIT
#define QA_F16_SPLIT 1
#endif
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  const bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
#if QA_F16_SPLIT
  uint32_t h, l;
  float l0, l1;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));      // (volatile: reads MODE, see mixed_pack4 in common.h)
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(b));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(l0), "v"(l1));
  hi = h;
  lo = l;
#else
  uint32_t h = pk_bf16(a, b);
  asm("" : "+v"(h));
  hi = h;
  lo = pk_bf16(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u));
#endif
}
// Four fp32 values -> packed hi and lo: {v0 | v1 << 16, v2 | v3 << 16}.  Written on dwords and pinned by an empty asm: left to the vector
// types, the compiler keeps every 16-bit element of the 8 x 35 values in a register of its own next to the packed form (the element-wise
// stores of an earlier version indexed them), and spills a few hundred registers.
__device__ __forceinline__ void split4(const f32x4& v, u32x2& hi, u32x2& lo) {
  uint32_t h0, l0, h1, l1;
  split2(v[0], v[1], h0, l0);
  split2(v[2], v[3], h1, l1);
  hi = u32x2{h0, h1};
  lo = u32x2{l0, l1};
  asm volatile("" : "+v"(hi), "+v"(lo));
}
// one 32x32x16 MFMA term of the attention products on packed 16-bit operands (8 values = 4 dwords per lane)
typedef _Float16 qa_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(16))) float qa_f32x16;
__device__ __forceinline__ qa_f32x16 mfma16(const u32x4& a, const u32x4& b, const qa_f32x16& c) {
#if QA_F16_SPLIT
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(qa_f16x8, a), __builtin_bit_cast(qa_f16x8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for_impl(F&& f) {
  if constexpr (I < N) {
    f(Tag<I>());
    static_for_impl<I + 1, N>(f);
  }
}
template <int N, class F>
__device__ __forceinline__ void static_for_n(F&& f) { static_for_impl<0, N>(f); }

#ifdef VETO_QA_STAMPS
__device__ unsigned long long g_qa_stamps[256 * 2 * 20];
__device__ __forceinline__ unsigned long long qa_stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define QST(x) x = qa_stamp()
#define QACC(a, t1, t0) a += (t1) - (t0)
// attention sub-phases (accumulated per wave 0 / 7): k = 0 conversion, 1 Q/K rows, 2 barrier, 3 scores + softmax, 4 barrier, 5 V rows, 6 barrier,
// 7 P V + stores, 8 barrier
#define AST(k) do { const unsigned long long t_ = qa_stamp(); a_ph[k] += t_ - t_att; t_att = t_; } while (0)
#else
#define QST(x)
#define QACC(a, t1, t0)
#define AST(k)
#endif

// s_waitcnt vmcnt(n) for a wave-uniform n <= 10 (the instruction takes an immediate)
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
  }
}

typedef short qa_s16x4 __attribute__((ext_vector_type(4)));
typedef short qa_s16x8 __attribute__((ext_vector_type(8)));
// two transposed LDS reads = one MFMA B fragment whose 8 contraction indices are image ROWS (4 + 4 tokens x 16 columns per 16-lane group)
__device__ __forceinline__ u32x4 lds_tr_pair(const char* p0, const char* p1) {
  const qa_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) qa_s16x4*)p0);
  const qa_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) qa_s16x4*)p1);
  const qa_s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(u32x4, v);
}

// FAST (VETO_FAST): the single-pass form -- the fp16 main product only.  The tile's stage stream keeps a length that is a multiple of the
// 2 x 3 image rotation (the next tile's first stage must land in image 0 of both kinds): 9 fp16 stages + 3 EMPTY intervals (a barrier, the
// DMA issue of the stages behind them, no fragment reads, no MFMAs) instead of 18 stages.
template <int DH, bool FAST = false>
__global__ __launch_bounds__(512, 2) void qkv_attn_fused_kernel(QkvAttnArgs g) {
  using G = Geo<DH>;
  constexpr int NB = G::NB, NWB = G::NWB;
  constexpr int kStages = FAST ? 12 : 18;      // intervals of a tile's main loop
  constexpr int kReal = FAST ? 9 : 18;         // ... of which the first kReal multiply: stage s = the 128-byte slice SL * s of every row
  constexpr int SL = FAST ? 2 : 1;
  saturating_conversions_on();   // (the mixed-row output converts without clamps, common.h)
  __shared__ __attribute__((aligned(16))) char smem[kLdsTotal];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w & 3, wn = w >> 2;
  constexpr int kColBase = QA_COL_INTERLEAVE ? 2048 : NB * 2048, kColStep = QA_COL_INTERLEAVE ? 4096 : 2048;   // LDS bytes: wave column, block n -> n + 1
  constexpr int kRowBase = QA_ROW_INTERLEAVE ? 2048 : MB * 2048, kRowStep = QA_ROW_INTERLEAVE ? 4 * 2048 : 2048;   // the same for the wave rows: block m -> m + 1
  const int b = blockIdx.x;
  const int H = g.heads;
  const int groups = (g.n_pair + TP - 1) / TP;
  const int ntiles = groups * H;
  const int per_xcd = gridDim.x >> 3;        // gridDim.x is a multiple of 8 (launcher)
  auto tile_of = [&](int it) { return (it * 8 + (b & 7)) * per_xcd + (b >> 3); };
  int my_tiles = 0;
  while (tile_of(my_tiles) < ntiles) ++my_tiles;
  if (my_tiles == 0) return;

  // ---- DMA side.  SIMD partners (waves i and i + 4) take turns on the two pipes: an LDS-DMA issue blocks its wave for ~130 cycles of
  // back-pressure (the vector-memory path moves the 65 KiB of a stage in ~2 000 of the stage's 2 240 MFMA cycles), and a wave that issues
  // between its MFMA groups stalls the matrix pipe whenever its partner is stalled the same way (measured: 3 300 cycles per stage).  So
  // the LOWER half (waves 0-3) issues its pieces FIRST and multiplies behind them, the UPPER half multiplies first and issues LAST: no
  // barrier between the halves, the matrix pipe of a SIMD always has one wave that is not waiting for the memory pipe, and each half
  // moves half of the bytes.  Three weight images: the lower half moves the weight rows of stage s + 2 (their image is free at the top of
  // the interval) and the first kASplit activation pieces (behind the release barrier), the upper half the other activation pieces;
  // two weight images: the lower half moves the weight rows of stage s + 1 (needed first), the upper half the activation rows of s + 2.
  // A wave's pieces (8 rows x 128 B each) are wl, wl + 4, ... of its range (wl = w & 3).
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int wl = w & 3;
  constexpr bool kRederiveLower = NB * MB * 4 > 160;      // (six heads only: see kRederive below)
  // (a scalar 0 / 1 pinned in a scalar register, compared where it is used: as a plain bool the compiler carries the wave-uniform flag as a
  // lane mask AND its complement, and at 180 accumulators it spilled a vector copy of it into the stage loop)
  int lower_s = __builtin_amdgcn_readfirstlane(w < 4 ? 1 : 0);
  asm volatile("" : "+s"(lower_s));
  auto is_lower = [&]() {      // (laundered at every use: one scalar compare each, no lane mask that lives across the stage loop)
    int t = lower_s;
    if (kRederiveLower) asm volatile("" : "+s"(t));
    return t != 0;
  };
#define lower is_lower()
  constexpr int kASplit = NWB == 3 ? QA_ASPLIT : 0;      // (even: a piece's swizzle parity is its index's)
  const int a_begin = lower ? 0 : kASplit, a_end = lower ? kASplit : APIECES;    // this wave's activation pieces: a_begin + wl + 4 k < a_end
  // Lane-dependent offsets are re-derived from the lane id where they are used (laundered through an empty asm: a few vector instructions
  // per stage) instead of living in registers through the main loop: at 180 accumulators (six heads) the allocator spills exactly those,
  // and a scratch reload inside the main loop waits for vmcnt(0), i.e. drains the wave's DMA queue.
  // (eight heads: 140 accumulators leave room; re-deriving there measured 4 % SLOWER -- the address arithmetic lands in front of the
  // fragment reads of every stage --, so the offsets stay loop-invariant and the compiler keeps them in registers)
  constexpr bool kRederive = NB * MB * 4 > 160;
  auto lane_now = [&]() {
    int l = lane;
    // (re-derived from the hardware, two instructions, rather than laundered from `lane`: at 180 accumulators a register that holds the
    // lane id across the main loop is itself spilled, and its reload sat behind the first barrier of every sixth stage)
    if (kRederive) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  // per-lane source offset of a piece: row (lane >> 3) of its 8, 16-byte slot (lane & 7) XOR-swizzled with the row's position in its 16-row block
  auto piece_voff = [&](int l, int first_row) {
    const int rr = l >> 3;
    const int r16 = ((wl & 1) << 3) + rr;                                // (wl + 4 i keeps the parity)
    const int slot16 = ((l & 7) ^ ((r16 >> 1) & 7)) << 4;                // source-side swizzle (the LDS side is lane-linear)
    return (unsigned)((first_row + rr) * kRowB + slot16);
  };
  constexpr int NPA = ((APIECES - kASplit > kASplit ? APIECES - kASplit : kASplit) + 3) / 4, NPW = (G::WPIECES + 3) / 4;      // pieces per wave at most (either half)
  const int n_acts = a_end - a_begin > wl ? (a_end - a_begin - wl + 3) / 4 : 0;
  const int n_weights = lower ? (G::WPIECES - wl + 3) / 4 : 0;
  struct TileSrc {
    const char* a;      // first activation row of the tile's pair group
    const char* w;      // first weight row of the tile's head: q rows; k rows 576 rows on, v rows 1152 rows on
  };
  auto tile_src = [&](int tile) {
    TileSrc t;
    const int group = tile / H, head = tile - group * H;
    t.a = g.a + (size_t)group * ((size_t)TM * kRowB);
    t.w = g.w + (size_t)(head * DH) * kRowB;
    return t;
  };
  // this wave's share of stage s of a tile: activation rows into image `buf` (0 / 1) / weight rows into image `buf` (0 .. NWB - 1)
  // (p_lo, p_hi: only the pieces p_lo <= p < p_hi -- all of them but for the next tile's stage 0 around the attention phase, see kAFree)
  auto issue_acts = [&](const TileSrc& t, int s, int buf, int p_lo = 0, int p_hi = APIECES) {
    const unsigned voff_a = piece_voff(lane_now(), wl * 8);
    int issued = 0;
    int a_end_s = a_end;
#if QA_FP6_PROBE
    if (s & 1) a_end_s = a_begin + (a_end - a_begin) * 3 / 4;      // three quarters of this half's pieces
#endif
#pragma unroll
    for (int k = 0; k < NPA; ++k)
      if (a_begin + wl + 4 * k < a_end_s && a_begin + wl + 4 * k >= p_lo && a_begin + wl + 4 * k < p_hi) {
        ++issued;
        glds16(t.a + (size_t)(a_begin * 8 + 32 * k) * kRowB + s * (SL * 128), voff_a, lds0 + buf * G::kPair + (a_begin + wl + 4 * k) * 1024);
      }
#if QA_FP6_PROBE
    if ((s & 1) && (w == 0 || w >= 3) && p_lo == 0) {      // stand-ins of the scale DMAs: 5 x 256 B of activation scales, 1 KiB of weight scales
      ++issued;
      if (w == 0) glds16(t.w, (unsigned)(lane_now() * 16), lds0 + buf * G::kPair + 30720);
      else asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dword %1, %2" ::"s"(lds0 + buf * G::kPair + 31744 + (w - 3) * 256), "v"((unsigned)(lane_now() * 4)), "s"(t.a) : "memory");
    }
#endif
    return issued;      // (wave-uniform)
  };
  auto issue_weights = [&](const TileSrc& t, int s, int buf) {
    if (!lower) return 0;
    const unsigned voff_w = piece_voff(lane_now(), 0);
    int issued = 0;
    int wp = G::WPIECES;
#if QA_FP6_PROBE
    if (s & 1) wp = G::WPIECES * 3 / 4;
#endif
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int c = wl + 4 * k;          // piece c of the head's q | k | v rows: matrix c / PM, rows 8 (c % PM) .. of that matrix's head slice
      if (c < wp) {
        ++issued;
        const int mat = c / G::PM, cm = c - mat * G::PM;
        glds16(t.w + (size_t)(mat * kDim + 8 * cm) * kRowB + s * (SL * 128), voff_w, lds0 + (buf < 2 ? kABytes + buf * G::kPair : 2 * G::kPair) + c * 1024);
      }
    }
    return issued;
  };

  // Fragment addresses.  Eight heads: one base register per image and 64-byte half (ten registers), so that every fragment read of the main
  // loop is base + immediate offset (the images lie beyond the 64 KiB an offset field reaches from one base) -- no vector arithmetic in front
  // of the reads that open a stage; the stage loop is unrolled six times (2 activation x 3 weight images) so that the image indices are
  // compile-time.  Six heads: re-derived per stage (see lane_now).
  int ab_[2], ab64_[2], wv_[3], wv64_[3];
  {
    const int fr0 = lane & 15, fq0 = lane >> 4;
    const int fo0 = fr0 * 128 + ((fq0 ^ ((fr0 >> 1) & 7)) << 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ab_[i] = G::a_buf(i) + wm * kRowBase + fo0;
      ab64_[i] = ab_[i] ^ 64;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      wv_[j] = G::w_buf(j < NWB ? j : 0) + wn * kColBase + fo0;
      wv64_[j] = wv_[j] ^ 64;
    }
  }

  // ---- MFMA side ---------------------------------------------------------------------------------------------------------------
  const int mix_scale = (127 - kMixActExp - __builtin_amdgcn_readfirstlane(*g.w_exp)) * 0x01010101;

  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t_begin = 0, s_wait = 0, s_bar1 = 0, s_frag = 0, s_bar2 = 0, s_main = 0, s_att = 0;
  (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)t5; (void)t_begin; (void)s_wait; (void)s_bar1; (void)s_frag; (void)s_bar2; (void)s_main; (void)s_att;
  QST(t_begin);
#ifdef VETO_QA_STAMPS
  unsigned long long a_ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

  // prologue: stages 0 and 1 of the first tile (NWB == 2: the weight rows of stage 1 go out in interval 0 like every W(s + 1))
  TileSrc src_next = tile_src(tile_of(0));
  issue_weights(src_next, 0, 0);
  issue_acts(src_next, 0, 0);
  if (NWB == 3) issue_weights(src_next, 1, 1);
  issue_acts(src_next, 1, 1);
  // DMA instructions of this wave that may still be in flight at the top of the next interval: its pieces of a stage BEHIND the one that
  // interval multiplies (with two weight images the weight pieces are for the very next stage: none of them; they are issued first)
  int younger = n_acts + (NWB == 3 ? n_weights : 0);
  bool stage0_landed = false;    // (behind the attention phase: the stage was waited for in front of it)

  for (int it = 0; it < my_tiles; ++it) {
    const int tile = tile_of(it);
    const int group = tile / H, head = tile - group * H;
    // (pinned in a scalar register: out of scalar registers at 180 accumulators, the compiler otherwise keeps this flag as a 0 / 1 VECTOR
    // value and spills THAT to scratch -- a reload in the stage loop, which waits for vmcnt(0) and with it for the wave's whole DMA queue)
    int has_next_s = __builtin_amdgcn_readfirstlane(it + 1 < my_tiles ? 1 : 0);
    if (kRederive) asm volatile("" : "+s"(has_next_s));
    const bool has_next = has_next_s != 0;
    const TileSrc src = src_next;
    if (has_next) src_next = tile_src(tile_of(it + 1));
    f32x4 acc[NB][MB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (opaque zeros: knowing them, the compiler peels the first two stages into MFMAs with a constant-zero C operand whose results
        // land in fresh registers -- and spills accumulators inside the main loop, where a scratch reload drains the DMA queue)
        asm volatile("" : "+v"(acc[n][m]));
      }

    // The MFMAs are inline asm with the accumulator tied ("+v"), as in ffn_fused.hip: the compiler's own forms rename the accumulators
    // from stage to stage (D != C) and, at 140 / 180 of them, spill inside the main loop -- and a scratch reload there drains the wave's
    // DMA queue (scratch traffic shares vmcnt).  The compiler then pads no MFMA hazard: `s_nop 1` opens every MFMA (operand written by a
    // vector instruction right in front of it), the first non-MFMA readers of the accumulators sit behind mfma_drain() below, and
    // veto_amd/asmcheck.py audits the generated code after every build.
    auto mma = [&](auto kind_tag, f32x4& c, const i32x4& w0, const i32x4& w1, const i32x4& a0, const i32x4& a1, int scale, int scale_b = 0x7f7f7f7f) {
      constexpr int KIND = decltype(kind_tag)::value;
      if (QA_ABLATE & 4) {
        asm volatile("" : "+v"(c) : "v"(w0), "v"(w1), "v"(a0), "v"(a1));
      } else if constexpr (KIND == 0) {
        asm(QA_MMA_NOP "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(w0), "v"(a0));
        asm(QA_MMA_NOP "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(w1), "v"(a1));
      } else if constexpr (QA_FP6_PROBE != 0) {
        typedef int i32x6 __attribute__((ext_vector_type(6)));
        const i32x6 w6 = __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5), a6 = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5);
        asm(QA_MMA_NOP "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(c) : "v"(w6), "v"(a6), "v"(scale), "v"(scale_b));
      } else {
        const i32x8 w8 = __builtin_shufflevector(w0, w1, 0, 1, 2, 3, 4, 5, 6, 7), a8 = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        asm(QA_MMA_NOP "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(w8), "v"(a8), "v"(scale), "v"(0x7f7f7f7f));
      }
    };
    // One stage: this wave's 5 activation fragments are held, its NB weight fragments stream through two buffers; group n = the
    // reads of weight fragment n + 1, the MFMAs of fragment n, then one or two of this wave's DMA instructions.
    auto stage = [&](auto kind_tag, auto ab_tag, auto wb_tag, int s) {
      constexpr int KIND = decltype(kind_tag)::value, AB = decltype(ab_tag)::value, WB = decltype(wb_tag)::value;   // stage kind, s % 2, s % NWB
      QST(t0);
      if (!stage0_landed) wait_vm(younger);              // this wave's pieces of stage s have landed
      stage0_landed = false;
      QST(t3);
      wg_barrier();                                       // everybody's have; every wave is done with stage s - 1
      QST(t1);

      // what this interval issues: the activation rows of stage s + 2 into the image of stage s (released below), and the weight rows
      // of stage s + 2 into the third image (NWB == 3) or of stage s + 1 into the other image (NWB == 2).  Behind the tile's last stage
      // only the next tile's stage 0 may be in flight (the attention regions cover every other image): stage 1 follows the attention.
      const int sa2 = s + 2, sw2 = NWB == 3 ? s + 2 : s + 1;
      const bool real = s < kReal;      // (FAST: the last three intervals of a tile multiply nothing)
      const bool do_a = sa2 < kReal || (sa2 == kStages && has_next);
      const bool do_w = sw2 < kReal || (sw2 == kStages && has_next && G::kFreeW0);
      const TileSrc& ta = sa2 < kStages ? src : src_next;
      const TileSrc& tw = sw2 < kStages ? src : src_next;
      const int st_a = sa2 < kStages ? sa2 : 0, st_w = sw2 < kStages ? sw2 : 0;
      const int a_limit = sa2 < kStages ? APIECES : G::kAFree;      // the next tile's stage 0: only the pieces the attention regions leave free
      int n_issued = 0;                                             // this wave's activation instructions of this interval
      int n_w_issued = 0;
      constexpr int wb2 = NWB == 3 ? (WB == 0 ? 2 : WB - 1) : (WB ^ 1);      // (s + 2) % 3 / (s + 1) % 2
      int a_lo, a_hi, w_lo, w_hi;      // LDS byte offsets of this lane's first activation / weight fragment: first and second 64-byte half
      if constexpr (kRederive) {
        const int lane_s = lane_now();
        const int fr = lane_s & 15, fq = lane_s >> 4;
        const int frag_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
        a_lo = G::a_buf(AB) + wm * kRowBase + frag_off;
        w_lo = G::w_buf(WB) + wn * kColBase + frag_off;
        a_hi = a_lo ^ 64;
        w_hi = w_lo ^ 64;
      } else {
        a_lo = ab_[AB]; a_hi = ab64_[AB]; w_lo = wv_[WB]; w_hi = wv64_[WB];
      }
      constexpr int WBUF = kRederive ? 2 : QA_WBUF;
      i32x4 fa0[MB], fa1[MB], fw0[WBUF], fw1[WBUF];
      // second half of a fragment: 16 bytes, or (fp6 probe, correction stage) 8
      auto half2 = [&](int off) {
        if constexpr (QA_FP6_PROBE != 0 && KIND == 1) {
          const u32x2 v = *(const u32x2*)(smem + off);
          i32x4 r;
          r[0] = (int)v[0]; r[1] = (int)v[1];
          return r;
        } else {
          return *(const i32x4*)(smem + off);
        }
      };
      int sc_w = mix_scale, sc_a = 0x7f7f7f7f;
      if constexpr (QA_FP6_PROBE != 0 && KIND == 1) {      // stand-ins of the block-scale reads
        const u32x2 sa2 = *(const u32x2*)(smem + G::a_buf(AB) + 30720 + ((a_lo * 2) & 0x3f8));
        sc_w = *(const int*)(smem + G::a_buf(AB) + 31744 + (w_lo & 0x3fc));
        sc_a = (int)(sa2[0] ^ sa2[1]);
      }
      if ((QA_ABLATE & 16) || !real) {
#pragma unroll
        for (int m = 0; m < MB; ++m) { fa0[m] = fa1[m] = i32x4{0, 0, 0, 0}; asm volatile("" : "+v"(fa0[m]), "+v"(fa1[m])); }
#pragma unroll
        for (int i = 0; i < WBUF; ++i) { fw0[i] = fw1[i] = i32x4{0, 0, 0, 0}; asm volatile("" : "+v"(fw0[i]), "+v"(fw1[i])); }
      } else {
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        fa0[m] = *(const i32x4*)(smem + a_lo + m * kRowStep);
        fa1[m] = half2(a_hi + m * kRowStep);
      }
#pragma unroll
      for (int i = 0; i < WBUF - 1; ++i) {
        fw0[i] = *(const i32x4*)(smem + w_lo + i * kColStep);
        fw1[i] = half2(w_hi + i * kColStep);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (lower) {
        // early release of the activation image: every wave has its fragments in registers.  (The barrier comes in front of the weight
        // pieces too, which do not need it: the upper half waits at it behind its first MFMA group.)
        if (do_a) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          QST(t4);
          wg_barrier();
          QST(t5);
          QACC(s_frag, t4, t1); QACC(s_bar2, t5, t4);
        }
        if (do_w) n_w_issued = issue_weights(tw, st_w, wb2);
        if (do_a) n_issued = issue_acts(ta, st_a, AB, 0, a_limit);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!real) {
        // an empty interval: no fragments, no MFMAs; the upper half joins the release barrier right away
        if (!lower && do_a) wg_barrier();
      } else
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        if (n + WBUF - 1 < NB && !(QA_ABLATE & 16)) {
          fw0[(n + WBUF - 1) % WBUF] = *(const i32x4*)(smem + w_lo + (n + WBUF - 1) * kColStep);
          fw1[(n + WBUF - 1) % WBUF] = half2(w_hi + (n + WBUF - 1) * kColStep);
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) mma(kind_tag, acc[n][m], fw0[n % WBUF], fw1[n % WBUF], fa0[m], fa1[m], sc_w, sc_a);
        __builtin_amdgcn_sched_barrier(0);
        if (n == 0 && !lower && do_a) {
          // (the upper half joins the release barrier behind its first group: its fragments are in registers by then, and its matrix
          // work starts without waiting for the lower half)
          QST(t4);
          wg_barrier();
          QST(t5);
          QACC(s_bar2, t5, t4);
        }
      }
      if (!lower && do_a) n_issued = issue_acts(ta, st_a, AB, 0, a_limit);
      younger = n_issued + (NWB == 3 && do_w ? n_w_issued : 0);
      QST(t2);
      QACC(s_wait, t3, t0); QACC(s_bar1, t1, t3); QACC(s_main, t2, t1);
    };
    static_assert(kStages % 6 == 0, "the stage loop is unrolled over the 2 x 3 image indices");
#if QA_PRIO == 1
    if (!lower) __builtin_amdgcn_s_setprio(1);
#elif QA_PRIO == 2
    if (lower) __builtin_amdgcn_s_setprio(1);
#endif
    constexpr int K1 = FAST ? 0 : 1;      // kind of the odd stages: the e4m3 part of a block, or (FAST) the next block's fp16 part
    for (int s = 0; s < kStages; s += 6) {
      stage(Tag<0>(), Tag<0>(), Tag<0>(), s);
      stage(Tag<K1>(), Tag<1>(), Tag<1 % NWB>(), s + 1);
      stage(Tag<0>(), Tag<0>(), Tag<2 % NWB>(), s + 2);
      stage(Tag<K1>(), Tag<1>(), Tag<3 % NWB>(), s + 3);
      stage(Tag<0>(), Tag<0>(), Tag<4 % NWB>(), s + 4);
      stage(Tag<K1>(), Tag<1>(), Tag<5 % NWB>(), s + 5);
    }

    // ---- attention phase -----------------------------------------------------------------------------------------------------------
#if QA_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    QST(t0);
#ifdef VETO_QA_STAMPS
    unsigned long long t_att = t0;
#endif
    wait_vm(0);              // the next tile's stage 0 (issued an interval ago) has landed
    wg_barrier();            // every wave is done with the last stage: the images the regions cover are free
    // MFMA result -> vector reader: more than 18 wait states (the MFMA statements are register-only asm: nothing but the scheduling
    // fences orders them against the conversion below)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    u32x2 ch[NB][MB], cl[NB][MB];      // the accumulators as packed bf16 hi / lo (same register count)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        if (QA_ABLATE & 256) { ch[n][m] = u32x2{__float_as_uint(acc[n][m][0]), __float_as_uint(acc[n][m][1])}; cl[n][m] = u32x2{__float_as_uint(acc[n][m][2]), __float_as_uint(acc[n][m][3])}; }
        else split4(acc[n][m], ch[n][m], cl[n][m]);
      }
    AST(0);

    int lane_a = lane;
    asm volatile("" : "+v"(lane_a));           // (see write_rows)
    const int r32 = lane_a & 31, hh = lane_a >> 5;
    const int rq = r32 < kTokens ? r32 : 0;    // lanes beyond the 19 tokens re-read row 0; their outputs are masked
    const float scale_l2 = 1.4426950408889634f / sqrtf((float)DH);      // dh^-0.5 log2(e): the softmax runs on v_exp_f32 (2^x)
    char* const region = smem + G::kScratch + w * G::REGION;
#pragma unroll 1
    for (int bt = 0; bt < ((QA_ABLATE & 2) ? 0 : 2); ++bt) {
      const int my_pair = group * TP + (QA_ROW_INTERLEAVE ? 8 * bt + w : 4 * (w >> 1) + 2 * bt + (w & 1));     // the pair whose attention this wave runs in this batch
      const bool active = my_pair < g.n_pair;
      // -- the owners write the batch's rows: lane (fr, fq) of block (n, m) holds token row 16 (5 wm + m) + fr, columns 16 (NB wn + n) + 4 fq ..
      // of q | k | v.  PHASE 0: the Q / K images (hi, lo planes); PHASE 1: the V images.  Everything but the row part of the address
      // is a compile-time constant per (wn, n): the stores carry it in their offset field.
      auto batch_of = [](int pr) { return QA_ROW_INTERLEAVE ? pr >> 3 : (pr >> 1) & 1; };     // batch of pair pr of the tile
      auto write_rows = [&](auto wn_tag, auto phase_tag) {
        constexpr int WN = decltype(wn_tag)::value, PHASE = decltype(phase_tag)::value;
        constexpr int LO = PHASE == 0 ? G::PLANE : G::VPLANE;                  // distance hi plane -> lo plane
        // (the lane id is laundered through an empty asm: every address below is invariant across batches and tiles, and the compiler
        // would otherwise compute them once, in front of the tile loop, and keep them -- i.e. spill them -- across the main loop)
        int lane_w = lane;
        asm volatile("" : "+v"(lane_w));
        const int fr_w = lane_w & 15, fq_w = lane_w >> 4;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const int r0 = 16 * (QA_ROW_INTERLEAVE ? wm + 4 * m : wm * MB + m);  // wave-uniform: skip blocks without a row of this batch
          const int pa = (r0 * 27) >> 9, pb = ((r0 + 15) * 27) >> 9;           // (r / 19 for r < 513)
          if (r0 >= TM || !(batch_of(pa) == bt || batch_of(pb) == bt)) continue;
          const int r = r0 + fr_w;
          const int p = (r * 27) >> 9, t = r - 19 * p;                         // pair inside the tile, token
          const bool mine = r < TM && batch_of(p) == bt;
          char* const rowp = smem + G::kScratch + (QA_ROW_INTERLEAVE ? p & 7 : ((p >> 2) << 1) | (p & 1)) * G::REGION + t * G::QP + 8 * fq_w;
          if (mine) {
            static_for_n<NB>([&](auto n_tag) {
              constexpr int n = decltype(n_tag)::value;
              constexpr int c0 = 16 * (QA_COL_INTERLEAVE ? 2 * n + WN : WN * NB + n);   // first column of the block; this lane: c0 + 4 fq ..
              constexpr int m_lo = c0 / DH, m_hi = (c0 + 15) / DH;               // matrix of the block's first / last column (3 = padding)
              auto in_phase = [](int mat) { return PHASE == 0 ? mat < 2 : mat == 2; };
              auto off_of = [](int mat) { return (PHASE == 0 ? mat * 2 * G::PLANE : 0) + (c0 - mat * DH) * 2; };   // (+ 8 fq: in rowp)
              if constexpr (m_lo == m_hi) {
                if constexpr (in_phase(m_lo)) {
                  *(u32x2*)(rowp + off_of(m_lo)) = ch[n][m];
                  *(u32x2*)(rowp + off_of(m_lo) + LO) = cl[n][m];
                }
              } else {
                static_assert(m_hi * DH - c0 == 8, "a matrix boundary inside a block lies between lanes fq = 1 and fq = 2");
                if constexpr (in_phase(m_lo) && in_phase(m_hi)) {
                  char* dst = rowp + (fq_w >= 2 ? off_of(m_hi) : off_of(m_lo));
                  *(u32x2*)dst = ch[n][m];
                  *(u32x2*)(dst + LO) = cl[n][m];
                } else if constexpr (in_phase(m_lo)) {
                  if (fq_w < 2) {
                    *(u32x2*)(rowp + off_of(m_lo)) = ch[n][m];
                    *(u32x2*)(rowp + off_of(m_lo) + LO) = cl[n][m];
                  }
                } else if constexpr (in_phase(m_hi)) {
                  if (fq_w >= 2) {
                    *(u32x2*)(rowp + off_of(m_hi)) = ch[n][m];
                    *(u32x2*)(rowp + off_of(m_hi) + LO) = cl[n][m];
                  }
                }
              }
            });
          }
        }
      };
      if (!(QA_ABLATE & 128)) { if (wn == 0) write_rows(Tag<0>(), Tag<0>()); else write_rows(Tag<1>(), Tag<0>()); }
      AST(1);
      lds_barrier();
      AST(2);
      // -- S^T = K Q^T on region w: row = key j, column = query i (attention.hip) ----------------------------------------------------
      u32x4 ph[2], pl[2];
      if (QA_ABLATE & 32) { ph[0] = ph[1] = pl[0] = pl[1] = u32x4{0u, 0u, 0u, 0u}; }
      if (active && !(QA_ABLATE & 32)) {
        const char* q_hi = region;
        const char* q_lo = region + G::PLANE;
        const char* k_hi = region + 2 * G::PLANE;
        const char* k_lo = region + 3 * G::PLANE;
        f32x16 st;
#pragma unroll
        for (int t = 0; t < 16; ++t) st[t] = 0.f;
        constexpr int KS = (DH + 15) / 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int off = rq * G::QP + (16 * s + 8 * hh) * 2;
          u32x4 kh = *(const u32x4*)(k_hi + off), kl = *(const u32x4*)(k_lo + off);
          u32x4 qh = *(const u32x4*)(q_hi + off), ql = *(const u32x4*)(q_lo + off);
          if (16 * s + 8 >= DH) {      // the upper half of the last k-step lies behind the row: zeros (both operands: 0 x NaN is NaN)
            const u32x4 z = {0u, 0u, 0u, 0u};
            if (hh) { kh = z; kl = z; qh = z; ql = z; }
          }
          st = mfma16(kl, qh, st);
          st = mfma16(kh, ql, st);
          st = mfma16(kh, qh, st);
        }
        // softmax over the keys of query (lane & 31): 16 registers here + 16 in lane ^ 32
        float p[16];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int j = (t & 3) + 8 * (t >> 2) + 4 * hh;
          p[t] = j < kTokens ? st[t] * scale_l2 : -INFINITY;
          mx = fmaxf(mx, p[t]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          p[t] = __builtin_amdgcn_exp2f(p[t] - mx);
          sum += p[t];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
          uint32_t h2, l2;
          split2(p[t] * inv, p[t + 1] * inv, h2, l2);
          ph[t >> 3][(t & 7) >> 1] = h2;      // (constant indices: the loop is unrolled)
          pl[t >> 3][(t & 7) >> 1] = l2;
        }
      }
      AST(3);
      lds_barrier();         // every wave has read its Q / K images: the V images go over them
      AST(4);
      if (!(QA_ABLATE & 128)) { if (wn == 0) write_rows(Tag<0>(), Tag<1>()); else write_rows(Tag<1>(), Tag<1>()); }
      if (lane_a < 2 * (G::QP / 16)) {       // image row 19 of this wave's own region: the zero row (keys 19..31)
        const int pln = lane_a >= G::QP / 16;
        *(u32x4*)(region + pln * G::VPLANE + kTokens * G::QP + (lane_a - pln * (G::QP / 16)) * 16) = u32x4{0u, 0u, 0u, 0u};
      }
      AST(5);
      lds_barrier();
      AST(6);
      // -- O = P V: A operand = P^T accumulators, whose element e of k-step s is key 16 s + 8 (e >> 2) + 4 h + (e & 3); the B operand is
      // read in that order from the row-major V images (transposed reads: lane 16 g + 4 q + p supplies row q of the 4, 8-byte piece p)
      if (active && !(QA_ABLATE & 64)) {
        const char* v_hi = region;
        const char* v_lo = region + G::VPLANE;
        float* o_lds = (float*)(region + 2 * G::VPLANE);   // [19][DH] fp32
        const int g4 = lane_a >> 4, tq = (lane_a >> 2) & 3, tp = lane_a & 3;
        int trow[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int ta0 = 16 * s + 4 * hh + tq, ta1 = ta0 + 8;
          trow[s][0] = (ta0 < kTokens ? ta0 : kTokens) * G::QP + (16 * (g4 & 1) + 4 * tp) * 2;
          trow[s][1] = (ta1 < kTokens ? ta1 : kTokens) * G::QP + (16 * (g4 & 1) + 4 * tp) * 2;
        }
        constexpr int NT = (DH + 31) / 32;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int d = 32 * n + r32;
          f32x16 o;
#pragma unroll
          for (int t = 0; t < 16; ++t) o[t] = 0.f;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const u32x4 vh = lds_tr_pair(v_hi + trow[s][0] + 64 * n, v_hi + trow[s][1] + 64 * n);
            const u32x4 vl = lds_tr_pair(v_lo + trow[s][0] + 64 * n, v_lo + trow[s][1] + 64 * n);
            o = mfma16(pl[s], vh, o);
            o = mfma16(ph[s], vl, o);
            o = mfma16(ph[s], vh, o);
          }
          // register t holds query (t & 3) + 8 (t >> 2) + 4 h: t < 8 always a token, t = 8..10 only in the lower half wave (16..18), t >= 11 never
          // (two predicated regions per tile instead of one per register: every `if` is an exec-mask save / restore and a branch)
          if (32 * n + 32 <= DH || d < DH) {
            float* op = o_lds + 4 * hh * DH + d;
#pragma unroll
            for (int t = 0; t < 8; ++t) op[((t & 3) + 8 * (t >> 2)) * DH] = o[t];
            if (!hh) {
#pragma unroll
              for (int t = 8; t < 11; ++t) op[((t & 3) + 8 * (t >> 2)) * DH] = o[t];
            }
          }
        }
        // (a wave reads back only what it wrote itself: LDS operations of one wave complete in order)
        constexpr int CH = DH / 8;
        __bf16* orow0 = (__bf16*)g.o + (size_t)my_pair * kTokens * (2 * kDim);
#pragma unroll
        for (int e0 = 0; e0 < (kTokens * CH + 63) / 64 * 64; e0 += 64) {
          const int e = e0 + lane_a;
          if (e < kTokens * CH) {
            const int i = e / CH, c = e - i * CH;
            const f32x4 v0 = *(const f32x4*)(o_lds + i * DH + c * 8);
            const f32x4 v1 = *(const f32x4*)(o_lds + i * DH + c * 8 + 4);
            if (!(QA_ABLATE & 8)) store_act8_mixed(orow0 + (size_t)i * (2 * kDim), head * DH + c * 8, v0, v1);
          }
        }
      }
      AST(7);
      lds_barrier();     // the regions are free: the second batch's rows, or the next tile's stage 1, go over them
      AST(8);
    }
    if (QA_ABLATE & 2) asm volatile("" :: "v"(ch[0][0]), "v"(cl[NB - 1][MB - 1]));
    if (has_next) {
      // the stages of the next tile that had to wait for the regions: stage 0's weight rows where the regions reach into the first weight
      // image (then interval 0 waits for them), stage 1 (NWB == 2: its weight rows go out in interval 0 like every W(s + 1))
      if (G::kAFree < APIECES) issue_acts(src_next, 0, 0, G::kAFree, APIECES);      // (older than every stage-1 piece below)
      if (!G::kFreeW0) issue_weights(src_next, 0, 0);
      if (NWB == 3) issue_weights(src_next, 1, 1);
      issue_acts(src_next, 1, 1);
      younger = n_acts + (NWB == 3 ? n_weights : 0);
      stage0_landed = G::kFreeW0 && G::kAFree == APIECES;     // (waited for in front of the attention phase; else interval 0 waits with `younger`)
    }
    QST(t1);
    QACC(s_att, t1, t0);
  }
#ifdef VETO_QA_STAMPS
  if ((w == 0 || w == 7) && lane == 0) {
    unsigned long long* o = g_qa_stamps + ((size_t)(b & 255) * 2 + (w == 7)) * 20;
    o[0] = s_wait; o[1] = s_main; o[2] = s_att; o[3] = t1 - t_begin; o[4] = my_tiles; o[5] = s_bar1; o[6] = s_frag; o[7] = s_bar2;
    for (int k = 0; k < 9; ++k) o[8 + k] = a_ph[k];
  }
#endif
}

#undef lower

}  // namespace

bool qkv_attn_fused_supports(int heads) {
  if (heads <= 0 || kDim % heads != 0) return false;
  const int dh = kDim / heads;
  return dh == 72 || dh == 96;
}

// rows the activation buffer must hold (readable) for n_pair pairs: whole tiles of 16 pairs
size_t qkv_attn_rows_padded(int n_pair) { return (size_t)((n_pair + TP - 1) / TP) * TM; }

hipError_t launch_qkv_attn_fused(const QkvAttnArgs& g, hipStream_t s) {
  if (!g.a || !g.w || !g.w_exp || !g.o || g.n_pair <= 0 || !qkv_attn_fused_supports(g.heads)) return hipErrorInvalidValue;
  int num_cu = device_cu_count();
  if (num_cu < 1) return hipErrorInvalidDevice;
  num_cu = num_cu / 8 * 8;
  if (num_cu < 8) num_cu = 8;
  const int ntiles = (g.n_pair + TP - 1) / TP * g.heads;
  int nblocks = num_cu;       // one persistent workgroup per CU (the whole LDS each)
  if (ntiles < nblocks) nblocks = (ntiles + 7) / 8 * 8;
  if (g.fast) {
    if (kDim / g.heads == 72) VETO_LAUNCH((qkv_attn_fused_kernel<72, true>), dim3(nblocks), dim3(512), 0, s, g);
    else VETO_LAUNCH((qkv_attn_fused_kernel<96, true>), dim3(nblocks), dim3(512), 0, s, g);
  } else if (kDim / g.heads == 72) VETO_LAUNCH(qkv_attn_fused_kernel<72>, dim3(nblocks), dim3(512), 0, s, g);
  else VETO_LAUNCH(qkv_attn_fused_kernel<96>, dim3(nblocks), dim3(512), 0, s, g);
  hipError_t rc = hipGetLastError();
#ifdef VETO_QA_STAMPS
  {
    static unsigned long long host[256 * 2 * 20];
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(g_qa_stamps), sizeof(host));
    const int nb = nblocks < 256 ? nblocks : 256;
    for (int wv = 0; wv < 2; ++wv) {
      double sum[20] = {0};
      for (int bb = 0; bb < nb; ++bb)
        for (int k = 0; k < 20; ++k) sum[k] += (double)host[(bb * 2 + wv) * 20 + k];
      fprintf(stderr, "[qkv_attn stamps pairs %d heads %d] wave %d of each workgroup, mean cycles: vm wait %.0f barrier1 %.0f | stage %.0f (of it: fragments %.0f "
              "barrier2 %.0f) | attention %.0f | total %.0f (%.2f tiles)\n", g.n_pair, g.heads, wv * 7, sum[0] / nb, sum[5] / nb, sum[1] / nb, sum[6] / nb,
              sum[7] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb);
      fprintf(stderr, "    attention phases: conversion %.0f | Q/K rows %.0f barrier %.0f | scores+softmax %.0f barrier %.0f | V rows %.0f barrier %.0f | PV+stores %.0f "
              "barrier %.0f\n", sum[8] / nb, sum[9] / nb, sum[10] / nb, sum[11] / nb, sum[12] / nb, sum[13] / nb, sum[14] / nb, sum[15] / nb, sum[16] / nb);
    }
  }
#endif
  return rc;
}

}  // namespace veto
