// The 128-row panel kernel of the VETO relation transformer: everything of a layer behind its attention on full rows.
//
//   MODE 2 (layer tail, the default path)   x1 = x + a Wo^T + bo                (model_veto.py:96 and the residual of :20)
//                                           x  <- x1 + W2 gelu(W1 LN2(x1) + b1) + b2   (:125-132, :137-143, residual of :21)
//                                           + LayerNorm1 of the NEXT layer's attention written as that layer's QKV operand
//   MODE 0 (VETO_TAIL_FUSED=0)              the FeedForward alone:  x <- x + W2 gelu(W1 a + b1) + b2,  a = LayerNorm2(x)
//   MODE 1 (VETO_TAIL_FUSED=0)              the out projection + residual (+ LayerNorm2 rows) alone
//
// in ONE launch: neither x1, nor the LayerNorm statistics, nor the 1152-wide hidden activation leaves the CU (the launch-per-Linear
// form wrote the hidden activation to HBM as a 1.3 GB matrix and read it back, and had two LayerNorm launches per layer).
// VETO_MIXED operands only (fp16 main product + e4m3 correction terms, common.h): a, W1 [1152, 576], W2 [576, 1152], Wo [576, 576]
// as mixed rows.
//
// One persistent workgroup per CU, 8 waves at two waves per SIMD (<= 256 registers), walks 128-row panels.  Per panel the
// [128 x 576] fp32 result stays in registers (144 per lane): it STARTS as the residual rows, the out projection (MODE 1 / 2: 18
// k-slices x 3 column thirds) accumulates on top, the mid-panel epilogue (MODE 2) adds the bias, writes LayerNorm2 of the rows as
// the FeedForward's input and leaves x1 where it is, and the hidden dimension is walked in 6 chunks of 192 columns:
//
//   fc1 phase   18 stages (9 blocks of 64 k's x {fp16 part, e4m3 part}): [128 x 192] += a[128, stage] . W1[chunk, stage]^T,
//               48 accumulator registers per lane
//   fc2 phase   per 64 hidden columns (3 per chunk): bias + GELU + conversion of those columns to the mixed-row format into an
//               LDS image (same layout as a DMA'd activation stage), then 6 sub-stages {fp16, e4m3} x 3 column thirds of
//               W2[third, block]: [128 x 192 of 576] += hidden[128, 64] . W2^T
//
// Wave (wm, wn) = (w >> 1, w & 1) of 4 x 2 owns rows 32 wm .. 32 wm + 31 and, of every 64 weight rows of a stage, the 32 at
// 32 wn: the hidden units 64 j + 32 wn .. of block j of a chunk, so that the hidden block j the fc2 phase consumes is complete as
// soon as every wave has converted ITS part, and lies in natural k order (no weight re-layout: the kernel reads the same mixed
// weight rows as the launch-per-Linear form).
//
// Every stage is 128 bytes of every row, exactly as in gemm_split_ps.hip: LDS-DMA (global_load_lds_dwordx4) with the
// source-side XOR swizzle, conflict-free ds_read_b128 fragments, MFMA issued with the weights as the A operand.  There are
// no loader waves (twelve waves would cap the kernel at 168 registers): every wave issues its share of the DMA of stage T + 2
// during interval T (5 instructions for an fc1 stage: 16 KiB of activations + 24 KiB of W1; 3 for an fc2 sub-stage:
// 24 KiB of W2) into a ring of three 40 KiB slots; one s_barrier per interval.
//
// Results depend on the row alone (fixed k order, no split-K): bit-identical under batch / chunk / permutation changes.
//
// -DVETO_FFN_STAMPS builds a diagnostic copy that accumulates s_memtime deltas per phase; tools/ffn_asm_stats.py and
// tools/audit_ffn_asm.py check the generated code (spills, compiler waits, hazards around the inline-asm MFMAs).
#include "common.h"
#include "kernels.h"

#include <cstdio>
#include <cstdlib>

// timing ablations (tools/ffn_variants.sh; results are WRONG with any of them set): 1 = no DMA, 2 = no MFMA, 4 = no fragment
// reads, 8 = no hidden conversion, 16 = no panel epilogue, 32 = the e4m3 stages move 3/5 (activation + weight slice) or 2/3 (weight slice)
// of their bytes: the DMA volume of 3-byte operand rows (fp16 + ONE e4m3 plane) without their conversion work,
// 256 = no residual-row loads (the accumulators start at zero: what the read burst at the top of a panel costs),
// 128 = the activation half of 32 alone (the LayerNorm2 rows of fc1 as fp16 + one e4m3 plane: upper bound of DESIGN.md section 11 item 1a),
// 64 = every workgroup streams the SAME FeedForward input panel (L2-resident) instead of its own: what the re-reads of the LayerNorm2 rows cost,
// 512 = no workgroup barriers
#ifndef FFN_ABLATE
#define FFN_ABLATE 0
#endif
// micro-variants (A/B with tools/ffn_variants.sh)
// the wait states in front of every inline-asm MFMA (see mma() below).  -DFFN_MMA_NOP='""' builds the kernel WITHOUT them: the
// negative control of tests/test_ffn_asm.py (the audit must then report hazards)
#ifndef FFN_MMA_NOP
#define FFN_MMA_NOP "s_nop 1\n\t"
#endif
// 1: odd hidden chunks walk the 64-k blocks of the panel backwards (see issue_fc1; -1.5 % on the layer tail; 0 = every chunk forwards)
#ifndef FFN_SNAKE
#define FFN_SNAKE 1
#endif
#ifndef FFN_CONV_G0
#define FFN_CONV_G0 0       // first of the three MFMA groups of a sub-stage that carry a piece of the hidden conversion (0..3)
#endif
#ifndef FFN_DMA_EARLY
#define FFN_DMA_EARLY 1     // DMA instructions issued before the first MFMA group of a stage (the rest follow groups 0, 1, ...)
#endif


#ifndef FFN_PRIO
#define FFN_PRIO 0
#endif
#ifndef FFN_NT
#define FFN_NT 0
#endif
#ifndef FFN_PINGPONG
#define FFN_PINGPONG 0
#endif
// 1: the LayerNorm rows of both epilogues leave as ONE 16-byte store per lane and 4 columns instead of two 8-byte stores (fp16 part, e4m3
// part): the lane pair that holds 8 adjacent columns of a row exchanges halves (a quad-permute DPP move in the final epilogue, where the
// pair is adjacent lanes; v_permlane16_swap in the mid-panel one, where it is 16 lanes apart), the even lane stores the 16 bytes of fp16
// values, the odd one the 16 bytes of e4m3 planes -- 36 instead of 72 store instructions per wave and epilogue.  Measured (round 6, same
// box, 287 280 rows): 2.229 / 2.222 / 2.224 ms with 8-byte stores, 2.234 / 2.202 / 2.208 with these: -0.5 % at best -- the panel
// boundary is bound neither by its store-instruction count (this) nor by its bytes (VETO_X_F24).  Kept: fewer instructions, same results
#ifndef FFN_LN16
#define FFN_LN16 1
#endif
// 1 (layer tail on fp32 residual rows; round 6, measured NULL, not the default): the accumulators start at ZERO and the panel's residual rows are
// added during the out projection, a column third at a time: twelve 16-byte loads per lane into the registers the FeedForward's fc1 accumulators
// will use (idle until then), issued at the top of an interval in front of its LDS-DMA instructions -- so the next interval's own vmcnt wait
// covers them -- and added at the top of a later interval of their column third (that third's MFMAs are three intervals old: no hazard).  The
// 36-load burst at the top of a panel becomes three trickles.  2.215 / 2.187 / 2.192 ms with the burst, 2.202 / 2.178 / 2.191 with the trickles
// (same box, parity-green, audit clean, 256 registers): the bytes have to come either way.  0: the accumulators start as the residual rows.
#ifndef FFN_MIDRES
#define FFN_MIDRES 0
#endif
// TIMING PROBE (results are WRONG): the correction stages of the FeedForward (fc1, fc2; not the out projection) as block-scaled fp6
// (e2m3) operands -- the K = 128 MFMA in its fp6 form (6 registers per operand, half the cycles of the e4m3 form), 96 instead of 128 bytes
// of every row per correction stage through the LDS-DMA (12 + 18 instead of 16 + 24 pieces of an fc1 stage, 18 instead of 24 of an fc2
// sub-stage), plus stand-ins for the block-scale traffic (3 / 1 small DMA instructions per stage, three scale reads per stage and wave)
#ifndef FFN_FP6_PROBE
#define FFN_FP6_PROBE 0
#endif

namespace veto {

namespace {

constexpr int FR = 128;                     // rows per panel
constexpr int FC = 192;                     // hidden columns per chunk
constexpr int FH = 2 * kDim;                // hidden width 1152
constexpr int kChunks = FH / FC;            // 6
constexpr int kRow1 = kDim * 4;             // bytes of a mixed row with K = 576 (activation rows, W1 rows)
constexpr int kRow2 = FH * 4;               // ... with K = 1152 (W2 rows)
// (stages per chunk: 18 fc1 stages + 18 fc2 sub-stages on mixed operands, 9 + 9 in the single-pass form: `Stages` inside the kernel)
constexpr int kAB = FR * 128;               // activation part of a ring slot: 16 KiB
constexpr int kWB = FC * 128;               // weight part: 24 KiB
constexpr int kSlot = kAB + kWB;
constexpr int kRing = 3;
constexpr int kHidOff = kRing * kSlot;      // hidden block: fp16 image [128 x 128 B], e4m3 image behind it
constexpr int kB1Off = kHidOff + 2 * kAB;
constexpr int kB2Off = kB1Off + FH * 4;
constexpr int kLds = kB2Off + kDim * 4;     // 162 560 B of the 163 840
static_assert(kLds <= 163840, "LDS budget");

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int V> struct Tag { static constexpr int value = V; };

// One LDS-DMA instruction: 64 lanes x 16 bytes from (uniform base + 32-bit lane offset) to LDS address m0 + 16 * lane.
// Inline asm: the saddr form costs no address arithmetic on the vector side, and the compiler's wait-count pass does not see
// the transfer (it would otherwise put s_waitcnt vmcnt(0) in front of every later ds_read of this wave; the kernel counts
// its own vmcnt).  The s_nop covers the SALU-writes-M0 -> LDS-DMA wait state and the five states between a scalar write of
// the base (the compiler computes it right in front of the statement) and its use by a vector-memory instruction.
// M0 cannot be named as a clobber (the compiler reserves it and warns); it holds nothing of the compiler's in these kernels -- LDS
// instructions need no M0 on gfx9+, and tests/test_ffn_asm.py fails if a compiler-generated instruction ever reads or writes it.
__device__ __forceinline__ void glds16(const char* base, unsigned voff, unsigned lds_addr) {
  if (FFN_ABLATE & 1) return;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}
// the same for rows that are read exactly once (the attention output rows of the out projection): non-temporal
__device__ __forceinline__ void glds16_nt(const char* base, unsigned voff, unsigned lds_addr) {
  if (FFN_ABLATE & 1) return;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory");
}
// a 16-byte global load the compiler does not count (see glds16): the caller's own s_waitcnt vmcnt covers it before the value is used
template <int OFF>
__device__ __forceinline__ void gload16(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  if (!(FFN_ABLATE & 512)) __builtin_amdgcn_s_barrier();      // (ablation 512: no stage barriers -- what the lockstep of the eight waves costs)
  asm volatile("" ::: "memory");
}

#ifdef VETO_FFN_STAMPS
__device__ unsigned long long g_ffn_stamps[256 * 8];
__device__ __forceinline__ unsigned long long stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(x) x = stamp()
#define ACC(a, t1, t0) a += (t1) - (t0)
// timeline of one chunk (workgroup 0, first panel, chunk 2; waves 0 and 4): [wave][position][top, waited, barrier, group 0..5]
__device__ unsigned long long g_ffn_timeline[2 * 36 * 9];
__device__ unsigned long long g_ffn_epi[2 * 12];
#define EP(k) do { if (b == 0 && cur_it == 1 && (w & 3) == 0) g_ffn_epi[(w >> 2) * 12 + (k)] = stamp(); } while (0)
#define TL(P, k) do { if (b == 0 && it == 0 && c == 2 && (w & 3) == 0) g_ffn_timeline[((w >> 2) * 36 + (P)) * 9 + (k)] = stamp(); } while (0)
#else
#define EP(k)
#define TL(P, k)
#define STAMP(x)
#define ACC(a, t1, t0)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(Tag<I>());
    static_for<I + 1, N>(f);
  }
}

// MODE 2: MODE 1's GEMM and MODE 0's FeedForward back to back on one panel -- everything of a layer behind its attention in one
// launch: x1 = x + a Wo^T + bo stays in the accumulators, LayerNorm2(x1) is written as the FeedForward's input rows, fc2 accumulates
// on top of x1 (the residual needs no second read), the final epilogue stores x2 (+ the next layer's LayerNorm1 rows).
// MODE 0: the FeedForward block.  MODE 1: the attention out projection + residual (+ LayerNorm2 epilogue) on the same skeleton: ONE
// GEMM x <- x + a W^T + b with the [128 x 576] result in registers, 18 k-slices x 3 column thirds = 54 positions per panel, the
// activation slice of a k-slice rides with its first third (always ring slot 0) and its fragments are held across the thirds.
// RF24 / OF24 (MODE 2): the residual rows come in / the result rows go out as 3-byte floats (common.h: rows of 576 * 3 bytes at g.resid /
// g.out) -- the residual stream between the layers of the default path, a quarter fewer bytes in the two bursts of a panel boundary
// FAST (VETO_FAST): the single-pass form -- the fp16 main product only.  The correction stage of every 64-k block is skipped (not loaded,
// not multiplied), everything else is the same kernel on the same mixed rows: what this schedule costs with the precision terms free.
template <int MODE, bool RF24 = false, bool OF24 = false, bool FAST = false>
__global__ __launch_bounds__(512, 2) void ffn_fused_kernel(FfnArgs g) {
  constexpr int NK = FAST ? 1 : 2;           // stages per 64-k block: {fp16 part, e4m3 part}, or the fp16 part alone
  constexpr int kS1 = kDim / 64 * NK;        // fc1 stages per chunk: 18 / 9
  constexpr int kS2 = FC / 64 * NK * 3;      // fc2 sub-stages per chunk: 18 / 9
  constexpr int kPer = kS1 + kS2;            // a multiple of the ring length either way
  constexpr int kOutPos = kDim / 64 * NK * 3;   // positions of the out projection: 54 / 27
  constexpr bool kMidRes = FFN_MIDRES && MODE == 2 && !RF24 && !FAST && !(FFN_ABLATE & 256);   // residual rows added during the out projection
  static_assert(kPer % kRing == 0 && kOutPos % kRing == 0, "a position's ring slot is a compile-time constant");
  saturating_conversions_on();   // (the hidden and LayerNorm conversions to mixed rows carry no clamps, common.h)
  __shared__ __attribute__((aligned(16))) char smem[kLds];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // the lane id, recomputed where a late phase needs it (two instructions) instead of a register that lives through the whole
  // kernel: with 192 accumulators the allocator spills such a register and its reload costs a full drain of the wave's DMA queue
  auto lane_now = []() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#if FFN_PRIO == 1
  if (w >= 4) __builtin_amdgcn_s_setprio(1);
#elif FFN_PRIO == 2
  if (w < 4) __builtin_amdgcn_s_setprio(1);
#endif
  const int wm = w >> 1, wn = w & 1;          // 4 x 2 waves: rows 32 wm .., of every 64 weight rows of a stage the 32 at 32 wn
  const int G = gridDim.x, b = blockIdx.x;
  const int my_panels = g.n_panels > b ? (g.n_panels - b + G - 1) / G : 0;   // panels b, b + G, ...
  if (my_panels == 0) return;

  // Equal panels keep the persistent workgroups in lockstep, so all 256 CUs reach their panel epilogue -- 0.6 / 0.9 MB of HBM
  // traffic per workgroup with no matrix work beside it -- at the same moment.  Start offsets per XCD or inside an XCD were
  // measured null (what they save in the epilogues they cost at the end of the launch).  What does pay (a
  // little): the workgroups that have one panel FEWER than the others (n_panels is rarely a multiple of the grid) have a whole
  // panel of slack, so they start `late` microseconds late for free and their epilogues fall into the others' matrix phases:
  // 2.27 -> 2.22-2.25 ms per layer-tail launch with 59 of 256 workgroups shifted by ~90 us.
  if (my_panels >= 2 && my_panels < (g.n_panels + G - 1) / G)
    for (int i = g.late; i > 0; --i) __builtin_amdgcn_s_sleep(32);               // ~1 us per unit
  if (MODE != 1)
    for (int i = tid; i < FH; i += 512) ((float*)(smem + kB1Off))[i] = g.b1[i];
  for (int i = tid; i < kDim; i += 512) ((float*)(smem + kB2Off))[i] = g.b2[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  wg_barrier();

  // ---- DMA side: this wave's pieces (8 rows x 128 B each) are w, w + 8 (, w + 16) of a stage image ----------------------
  // Addresses are (uniform base) + (32-bit lane offset): two registers of lane state in all.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int rr = lane >> 3;
  const int r16 = ((w & 1) << 3) + rr;                                   // row inside its 16-row group (w + 8 k keeps the parity)
  const int slot16 = ((lane & 7) ^ ((r16 >> 1) & 7)) << 4;               // source-side swizzle (the LDS side is lane-linear)
  const unsigned voff1 = (unsigned)((w * 8 + rr) * kRow1 + slot16);
  const unsigned voff2 = (unsigned)((w * 8 + rr) * kRow2 + slot16);
  const char* const w_out = MODE == 2 ? g.wo : g.w2;       // weight rows of the out-projection phase (MODE 1: the only GEMM)
  auto panel_base = [&](int it) { return g.a + (size_t)(b + (size_t)it * G) * ((size_t)FR * kRow1); };
  // MODE 2: the FeedForward phase of a panel reads the LayerNorm2 rows its out-projection phase wrote (ln_out; normally = a)
  auto ffn_base = [&](int it) { return (MODE == 2 ? (const char*)g.ln_out : g.a) + ((FFN_ABLATE & 64) ? (size_t)0 : (size_t)(b + (size_t)it * G) * ((size_t)FR * kRow1)); };
  // instruction k (of 5) of fc1 stage ks of hidden chunk c: 16 KiB of activation rows (k = 0, 1) + 24 KiB of W1 rows (2..4)
  auto issue_fc1 = [&](const char* a_panel, int c, int ks, int slot, int k) {
#if FFN_SNAKE
    // odd chunks walk the 64-k blocks of the panel from the last to the first: the LayerNorm2 rows an XCD's 32 workgroups re-read
    // per hidden chunk (9.4 MB) do not fit its 4 MB L2, but the blocks a chunk reads last are the ones the next chunk then reads first
    if (c & 1) ks = 2 * (8 - (ks >> 1)) + (ks & 1);
#endif
    const unsigned dst = lds0 + slot * kSlot + w * 1024;
#if FFN_FP6_PROBE
    if (ks & 1) {
      if (k == 1 && w >= 4) {      // (waves 4-6: the stand-in of a scale DMA: 2 x 256 B of activation scales, 1 KiB of weight scales)
        if (w < 6) asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dword %1, %2" ::"s"(lds0 + slot * kSlot + 12288 + (w - 4) * 256), "v"((unsigned)(lane_now() * 4)), "s"(a_panel) : "memory");
        else if (w == 6) glds16(g.w1, (unsigned)(lane_now() * 16), lds0 + slot * kSlot + kAB + 18432);
        return;
      }
      if (k == 4 && w >= 2) return;
    }
#endif
    if ((FFN_ABLATE & 32) && (ks & 1) && (k == 1 || k == 4)) return;
    if ((FFN_ABLATE & 128) && (ks & 1) && k == 1) return;   // (the activation half of ablation 32 alone)
    if (k < 2) glds16(a_panel + ks * 128 + k * 64 * kRow1, voff1, dst + k * 8 * 1024);
    else glds16(g.w1 + (size_t)c * ((size_t)FC * kRow1) + ks * 128 + (k - 2) * 64 * kRow1, voff1, dst + kAB + (k - 2) * 8 * 1024);
  };
  // instruction k (of 3) of an fc2 sub-stage: 24 KiB of W2 rows (column third t, 128-byte slice `slice` of the row)
  auto issue_fc2 = [&](int slice, int t, int slot, int k) {
    const unsigned dst = lds0 + slot * kSlot + w * 1024 + kAB;
#if FFN_FP6_PROBE
    if ((slice & 1) && k == 2 && w >= 2) {
      if (w == 7) glds16(g.w2, (unsigned)(lane_now() * 16), lds0 + slot * kSlot + kAB + 18432);
      return;
    }
#endif
    if ((FFN_ABLATE & 32) && (slice & 1) && k == 2) return;
    glds16(g.w2 + (size_t)t * ((size_t)FC * kRow2) + slice * 128 + (size_t)k * 64 * kRow2, voff2, dst + k * 8 * 1024);
  };

  // MODE 1, instruction k of position (ks, t): t == 0: the activation slice (k = 0, 1; ring slot 0) + the weight third (2..4);
  // t > 0: the weight third (k = 0..2).  The weight rows have K = 576 (row pitch kRow1).
  auto issue_out = [&](const char* a_panel, int ks, int t, int slot, int k) {
    const int kw = t == 0 ? k - 2 : k;
    if ((FFN_ABLATE & 32) && (ks & 1) && (t == 0 ? (k == 1 || k == 4) : k == 2)) return;
#if FFN_NT & 8
    if (t == 0 && k < 2) glds16_nt(a_panel + ks * 128 + k * 64 * kRow1, voff1, lds0 + w * 1024 + k * 8 * 1024);
#else
    if (t == 0 && k < 2) glds16(a_panel + ks * 128 + k * 64 * kRow1, voff1, lds0 + w * 1024 + k * 8 * 1024);
#endif
    else if (kw >= 0 && kw < 3)
      glds16(w_out + (size_t)t * ((size_t)FC * kRow1) + ks * 128 + kw * 64 * kRow1, voff1, lds0 + slot * kSlot + kAB + w * 1024 + kw * 8 * 1024);
  };

  // ---- MFMA side ---------------------------------------------------------------------------------------------------------
  // block i = 0..5 of a wave = weight rows (i >> 1) * 64 + wn * 32 + (i & 1) * 16 of the stage; row group m = 0, 1
  const int fr = lane & 15, fq = lane >> 4;
  const int frag_off = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
  const int a_off = (wm * 32) * 128 + frag_off;            // + m * 2048
  const int w_off = kAB + (wn * 32) * 128 + frag_off;      // + (i >> 1) * 8192 + (i & 1) * 2048
  const int sc1 = MODE != 1 ? (127 - kMixActExp - __builtin_amdgcn_readfirstlane(*g.exp1)) * 0x01010101 : 0;
  const int sco = MODE == 2 ? (127 - kMixActExp - __builtin_amdgcn_readfirstlane(*g.expo)) * 0x01010101 : 0;
  const int sc2 = (127 - kMixActExp - __builtin_amdgcn_readfirstlane(*g.exp2)) * 0x01010101;


  // The MFMAs are inline asm with the accumulator tied ("+v"): with 192 of 256 registers in accumulators the compiler's
  // untied forms rename them from stage to stage and spill on the way back (measured: 266 spilled registers; scratch
  // traffic also shares vmcnt with the DMA).  Hazards the compiler no longer pads (it does not know the statement is an MFMA):
  //  * a VALU write of an operand right in front of the MFMA -- the register allocator does insert such copies, e.g. it moved
  //    one accumulator block into its final registers two instructions ahead of its first MFMA, and the last two registers of
  //    the block were read stale (measured) -- : s_nop 1 opens the first MFMA of every accumulator step;
  //  * the first VALU / LDS read of an accumulator behind its last MFMA: mfma_drain*() below, tied to those accumulators;
  //  * an accumulator chain needs none; operands come from ds_read (waited for through the register dependency).
  // tools/audit_ffn_asm.py checks the generated code for compiler instructions that touch accumulator registers near an MFMA.
  auto mma = [&](auto kind_tag, f32x4& acc, const i32x4& fw0, const i32x4& fw1, const i32x4& fa0, const i32x4& fa1, int scale, int scale_b = 0x7f7f7f7f) {
    constexpr int KIND = decltype(kind_tag)::value;
    if (FFN_ABLATE & 2) {
      asm volatile("" : "+v"(acc) : "v"(fw0), "v"(fw1), "v"(fa0), "v"(fa1));
      return;
    }
    if constexpr (KIND == 2) {      // (FFN_FP6_PROBE) the fp6 form: 24 bytes per operand and lane
      typedef int i32x6 __attribute__((ext_vector_type(6)));
      const i32x6 w6 = __builtin_shufflevector(fw0, fw1, 0, 1, 2, 3, 4, 5), a6 = __builtin_shufflevector(fa0, fa1, 0, 1, 2, 3, 4, 5);
      asm(FFN_MMA_NOP "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(acc) : "v"(w6), "v"(a6), "v"(scale), "v"(scale_b));
    } else if constexpr (KIND == 0) {
      asm(FFN_MMA_NOP "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fw0), "v"(fa0));
      asm(FFN_MMA_NOP "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fw1), "v"(fa1));
    } else {
      const i32x8 w8 = __builtin_shufflevector(fw0, fw1, 0, 1, 2, 3, 4, 5, 6, 7), a8 = __builtin_shufflevector(fa0, fa1, 0, 1, 2, 3, 4, 5, 6, 7);
      asm(FFN_MMA_NOP "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(w8), "v"(a8), "v"(scale), "v"(0x7f7f7f7f));
    }
  };
  // MFMA result -> VALU / LDS reader: more than 18 wait states, tied to the accumulators it fences (the MFMA statements are
  // register-only: nothing else orders them against a later statement)
  auto mfma_drain3 = [&](f32x4& a, f32x4& b, f32x4& c) { asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b), "+v"(c)); };
  auto mfma_drain4 = [&](f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  };
  // One stage: the activation image at LDS offset AB (a ring slot's A part, or a hidden image) x the weight part of the ring
  // slot at SB (compile-time constants).  Register budget: 192 accumulators leave ~50 registers: the two activation fragments
  // of the wave are held (16), the six weight fragments stream through two buffers (16), and the four LDS addresses are
  // re-derived per stage from one lane register (laundered through an empty asm: the compiler would otherwise keep a dozen
  // loop-invariant address registers live, and spill).  Group i = the reads of weight fragment i + 1, the MFMAs of fragment
  // i, then one of this wave's DMA instructions for the stage two positions on (`dma(k)`, k < 5): an LDS-DMA issue waits
  // 100-200 cycles in back-pressure, during which the matrix pipe runs this group's MFMAs, then the SIMD partner's.
  // sched_barrier(0) after every group keeps that order.
  typedef __attribute__((address_space(3))) i32x4 lds_i32x4_t;
  typedef const lds_i32x4_t* lds_frag_t;
  auto stage = [&](auto kind_tag, auto ab_tag, auto sb_tag, f32x4 (&acc)[6][2], int scale, auto&& dma, auto reuse_tag, i32x4 (&fa0)[2],
                   i32x4 (&fa1)[2], auto&& valu, auto&& mark) {
    constexpr int AB = decltype(ab_tag)::value, SB = decltype(sb_tag)::value;
    constexpr bool REUSE = decltype(reuse_tag)::value != 0;    // the activation fragments of the previous sub-stage are still valid
    int fo = frag_off;
    asm volatile("" : "+v"(fo));
    const unsigned a0 = lds0 + AB + wm * 4096 + fo, a1 = lds0 + AB + wm * 4096 + (fo ^ 64);
    const unsigned w0 = lds0 + SB + kAB + wn * 4096 + fo, w1 = lds0 + SB + kAB + wn * 4096 + (fo ^ 64);
    i32x4 fw0[2], fw1[2];
    typedef __attribute__((address_space(3))) u32x2 lds_u32x2_t;
    constexpr int KIND_S = decltype(kind_tag)::value;
    // second half of a fragment: 16 bytes, or (fp6 probe) 8
    auto half2 = [&](unsigned addr) {
      if constexpr (KIND_S == 2) {
        const u32x2 v = *(const lds_u32x2_t*)(size_t)addr;
        i32x4 r;
        r[0] = (int)v[0]; r[1] = (int)v[1];
        return r;
      } else {
        return (i32x4)*(lds_frag_t)(size_t)addr;
      }
    };
    int scale_a = 0x7f7f7f7f;
    if constexpr (KIND_S == 2) {     // stand-ins of the block-scale reads: one dword of activation scales, one of weight scales per lane
      typedef __attribute__((address_space(3))) int lds_int_t;
      scale_a = *(const lds_int_t*)(size_t)(lds0 + SB + 12288 + (fo & 0xffc));
      scale = *(const lds_int_t*)(size_t)(lds0 + SB + kAB + 18432 + (fo & 0xffc));
    }
    if (FFN_ABLATE & 4) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        fa0[m] = fa1[m] = fw0[m] = fw1[m] = i32x4{0, 0, 0, 0};
        asm volatile("" : "+v"(fa0[m]), "+v"(fa1[m]), "+v"(fw0[m]), "+v"(fw1[m]));
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int m = 0; m < 2; ++m) mma(kind_tag, acc[i][m], fw0[i & 1], fw1[i & 1], fa0[m], fa1[m], scale);
        if (i == 0) { dma(0); dma(1); }
        dma(i + 2);
        valu(i);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    if constexpr (!REUSE) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        fa0[m] = *(lds_frag_t)(size_t)(a0 + m * 2048);
        fa1[m] = half2(a1 + m * 2048);
      }
    }
    fw0[0] = *(lds_frag_t)(size_t)(w0);
    fw1[0] = half2(w1);
    // the first two DMA instructions go out while the first fragments are on their way from the LDS (their issue back-pressure
    // and the LDS latency overlap instead of adding up); the others follow groups 0, 1, 2
#if FFN_PINGPONG
    // SIMD partners take turns (qkv_attn_fused.hip): waves 0-3 issue their whole DMA share in front of their MFMA groups, waves 4-7 (raised
    // priority, FFN_PRIO 1) multiply first and issue behind them
    if (w < 4) {
#pragma unroll
      for (int k = 0; k < 5; ++k) dma(k);
    }
#else
#pragma unroll
    for (int k = 0; k < FFN_DMA_EARLY; ++k) dma(k);
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i < 5) {
        fw0[(i + 1) & 1] = *(lds_frag_t)(size_t)(w0 + ((i + 1) >> 1) * 8192 + ((i + 1) & 1) * 2048);
        fw1[(i + 1) & 1] = half2(w1 + ((i + 1) >> 1) * 8192 + ((i + 1) & 1) * 2048);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) mma(kind_tag, acc[i][m], fw0[i & 1], fw1[i & 1], fa0[m], fa1[m], scale, scale_a);
#if !FFN_PINGPONG
      dma(i + FFN_DMA_EARLY);
#endif
      valu(i);
      mark(i);
      __builtin_amdgcn_sched_barrier(0);
    }
#if FFN_PINGPONG
    if (w >= 4) {
#pragma unroll
      for (int k = 0; k < 5; ++k) dma(k);
    }
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  // The hidden images of block j of a chunk: blocks 0 and 2 in the dedicated area, block 1 in the activation parts of ring
  // slots 0 and 1 (idle between the last fc1 stage and the prefetch of the next chunk's first stages at positions 34 / 35), so
  // that block j + 1 can be written while block j is being read: no barrier between them, and the conversion runs beside MFMAs.
  auto hid_f16 = [](int j) { return j == 1 ? 0 : kHidOff; };
  auto hid_e4m3 = [](int j) { return j == 1 ? kSlot : kHidOff + kAB; };
  // bias + GELU + mixed-row conversion of hidden block J of chunk c (this wave's 32 columns x 32 rows) into the hidden images
  auto hidden_write = [&](auto j_tag, int c, f32x4 (&acc1)[6][2]) {
    constexpr int J = decltype(j_tag)::value;
    const int lane_h = lane_now();
    const int hr = lane_h & 15, hq = lane_h >> 4;
    // every value is converted first, the eight LDS stores follow in one run behind a scheduling fence: the compiler merges the
    // two row groups of a block into one ds_write2st64_b64 and, left alone, overwrites its last data register in the very next
    // instruction (measured: exactly those two fp16 values of row group 1 reached the LDS corrupted)
    u32x2 hh[2][2];
    u32x2 xy[2][2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const f32x4 bias = *(const f32x4*)(smem + kB1Off + (c * FC + J * 64 + wn * 32 + ib * 16 + hq * 4) * 4);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x4 v = acc1[2 * J + ib][m] + bias;
        const f32x2 g0 = gelu_sigmoid2(f32x2{v[0], v[1]}), g1 = gelu_sigmoid2(f32x2{v[2], v[3]});
        mixed_pack4(f32x4{g0[0], g0[1], g1[0], g1[1]}, hh[ib][m], xy[ib][m]);
      }
    }
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int m = 0; m < 2; ++m) asm volatile("" : "+v"(hh[ib][m]), "+v"(xy[ib][m]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      const int ls = wn * 4 + ib * 2 + (hq >> 1);                       // 16-byte slot of this lane's 4 columns in the 128-byte row
      const int ho = (wm * 32 + hr) * 128 + ((ls ^ ((hr >> 1) & 7)) << 4) + (hq & 1) * 8;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        *(u32x2*)(smem + hid_f16(J) + ho + m * 2048) = hh[ib][m];
        if constexpr (!FAST) *(u32x2*)(smem + hid_e4m3(J) + ho + m * 2048) = xy[ib][m];
      }
    }
  };

  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t_begin = 0, s_wait = 0, s_bar = 0, s_iss = 0, s_cmp = 0, s_hid = 0, s_epi = 0;
  (void)t0; (void)t1; (void)t2; (void)t3; (void)t_begin; (void)s_wait; (void)s_bar; (void)s_iss; (void)s_cmp; (void)s_hid; (void)s_epi;
  STAMP(t_begin);

  // The stream of stages: per chunk 36 positions -- 18 fc1 stages (0..17), then 18 fc2 sub-stages (hidden block j = 0..2 x
  // {fp16, e4m3} x column third 0..2).  Position P lives in ring slot P % 3 (36 is a multiple of 3: every LDS address is a
  // compile-time constant).  The stage two positions on is issued during interval P, into the slot position P - 1 used.
  // (issue_fc1 / issue_out take the 128-byte SLICE of the rows: stage index x 2 in the single-pass form, which walks the fp16 slices only)
  constexpr int SL = FAST ? 2 : 1;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (MODE == 0) issue_fc1(panel_base(0), 0, 0, 0, k); else issue_out(panel_base(0), 0, 0, 0, k);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (MODE == 0) issue_fc1(panel_base(0), 0, 1 * SL, 1, k); else issue_out(panel_base(0), 0, 1, 1, k);
  }
  int skip = 0;                       // intervals whose stage is known to have landed (behind a full drain)

  int cur_it = 0; (void)cur_it;
  // ---- panel epilogue.  MFMA layout: lane 16 q + r holds row r of its 16-row group and 4 consecutive columns at 4 q of every
  // 16-column block, so a row's 576 columns sit in lanes r, r + 16, r + 32, r + 48 of the two waves of its row group.  The
  // accumulators started as the residual rows (panel loop below): acc + bias IS the new residual stream.
  //   MID (the layer tail's mid-panel point): bias = the out projection's, staged with LayerNorm2's gamma | beta (gbv, requested at
  //        position 52); x1 stays in the registers -- fc2 accumulates on top of it -- and LayerNorm2(x1) goes to ln_out;
  //   otherwise: bias = b2 (resident in LDS), the rows are stored to g.out, and with ln_w the LayerNorm of the finished rows
  //        (model_veto.py:125-132 of the NEXT layer's attention PreNorm) is written as mixed rows to ln_out: the standalone
  //        LayerNorm launch (0.66 GB read + 0.66 GB written) disappears.
  // LayerNorm statistics: two passes (mean, then centred squares) like rowq_stats in rowops.hip; the four lanes of a row are
  // summed by lane exchange, the two waves exchange through LDS (the activation part of ring slot 2, idle in every phase that
  // ends in an epilogue); both waves add the same two partial sums: identical statistics in either.  gamma / beta are read from
  // LDS: as global loads inside the normalise loop they would sit behind the loop's own stores in the wave's one in-order memory
  // queue (measured: 23 k cycles for the loop, store count and arithmetic notwithstanding).
  auto epilogue = [&](f32x4 (&acc2)[3][6][2], int panel, auto mid_tag, f32x4 gbv) {
    constexpr bool MID = decltype(mid_tag)::value != 0;
    const int lane_e = lane_now();
    // MID: the MFMA layout as it is.  Otherwise the rows leave through 16-byte stores, and those want every quad of lanes on 64
    // contiguous bytes of one row (four rows per quad cost the final epilogue ~20 % more time, measured): one ds_bpermute per value
    // (lane 4 r + q takes lane 16 q + r, gemm_split_ps.hip) puts row r, columns 4 q.. into lane 4 r + q first.
    const int r = MID ? (lane_e & 15) : (lane_e >> 2), q = MID ? (lane_e >> 4) : (lane_e & 3);
    const int perm_addr = ((q << 4) + r) << 2;
    const int col0 = wn * 32 + q * 4;                    // + t * 192 + (i >> 1) * 64 + (i & 1) * 16
    const int row0 = panel * FR + wm * 32 + r;           // + m * 16
    float* red = (float*)(smem + 2 * kSlot);             // [pass][wave][m][row] floats
    float* gb = (float*)(smem + 2 * kSlot + 4096);       // [gamma 576 | beta 576 | MID: bias 576]
    const bool ln = MID || ((g.ln_out || (MODE == 2 && g.ln1_out)) && g.ln_w);
    const float* bias = (const float*)(smem + kB2Off);
    const int tl = w * 64 + lane_e;                      // (the thread id, re-derived: see lane_now)
    if constexpr (MID) {
      if (tl < 432) ((f32x4*)gb)[tl] = gbv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wg_barrier();
      bias = gb + 2 * kDim;
    } else {
      if (ln && tl < 288) gbv = tl < 144 ? ((const f32x4*)g.ln_w)[tl] : ((const f32x4*)g.ln_b)[tl - 144];   // in flight under the row stores
    }
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int m = u / 6, t = (u % 6) >> 1, i0 = (u & 1) * 3;
      mfma_drain3(acc2[t][i0][m], acc2[t][i0 + 1][m], acc2[t][i0 + 2][m]);
#pragma unroll
      for (int i = i0; i < i0 + 3; ++i) {
        f32x4 tv = acc2[t][i][m];
        if constexpr (!MID) {
#pragma unroll
          for (int e = 0; e < 4; ++e) tv[e] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_addr, __float_as_int(acc2[t][i][m][e])));
        }
        acc2[t][i][m] = tv + *(const f32x4*)(bias + col0 + t * FC + (i >> 1) * 64 + (i & 1) * 16);
        asm volatile("" : "+v"(acc2[t][i][m]));          // pins the group here: register-only instructions carry no order of
      }                                                  // their own before instruction selection
      __builtin_amdgcn_sched_barrier(0);
    }
    EP(1);
    if constexpr (!MID) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int row = row0 + m * 16;
        if (row < g.M) {
          if constexpr (OF24) {
            // four columns = 12 bytes; the quad of lanes of a row stores 48 contiguous bytes.  The registers keep the fp32 values: the
            // LayerNorm below normalises what the layer computed, the next layer's residual is what memory holds (rounded to 16 bits)
            char* op = (char*)g.out + ((size_t)row * kDim + col0) * 3;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
              for (int i = 0; i < 6; ++i) *(u32x3*)(op + (t * FC + (i >> 1) * 64 + (i & 1) * 16) * 3) = pack_f24x4(acc2[t][i][m]);
          } else {
          float* op = g.out + (size_t)row * g.ldo + col0;
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#if FFN_NT & 1
              __builtin_nontemporal_store(acc2[t][i][m], (f32x4*)(op + t * FC + (i >> 1) * 64 + (i & 1) * 16));
#else
              *(f32x4*)(op + t * FC + (i >> 1) * 64 + (i & 1) * 16) = acc2[t][i][m];
#endif
            }
          }
        }
      }
    }
    EP(2);
    if (ln) {
      if (!MID && tl < 288) ((f32x4*)gb)[tl] = gbv;    // (published by the barrier of pass 0)
      float mean[2], rstd[2];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        float part[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float p = 0.f;
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              if (pass == 0) {
                p += (acc2[t][i][m][0] + acc2[t][i][m][1]) + (acc2[t][i][m][2] + acc2[t][i][m][3]);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = acc2[t][i][m][e] - mean[m]; p += d * d; }
              }
            }
          p += __shfl_xor(p, MID ? 16 : 1, 64);
          p += __shfl_xor(p, MID ? 32 : 2, 64);
          part[m] = p;
          if (q == 0) red[((pass * 8 + w) * 2 + m) * 16 + r] = p;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_barrier();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float tot = part[m] + red[((pass * 8 + (w ^ 1)) * 2 + m) * 16 + r];
          if (pass == 0) mean[m] = tot * (1.f / kDim);
          else rstd[m] = 1.f / sqrtf(tot * (1.f / kDim) + 1e-5f);
        }
        EP(3 + pass);
      }
      // (one 64-bit row pointer per row group, computed here: inside the loop the compiler re-derived it per store from a 64-bit
      // multiply; every column offset below is a compile-time constant that folds into the store's immediate)
      typedef __attribute__((address_space(1))) char gchar_t;          // (explicitly global: a pointer that went through an asm is generic)
      typedef __attribute__((address_space(1))) u32x2 gu32x2_t;
      gchar_t* lrow[2];
      f32x2 nmean[2], rstd2[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        nmean[m] = f32x2{-mean[m], -mean[m]};      // both statistics as register PAIRS: operands of the packed instructions as they are
        rstd2[m] = f32x2{rstd[m], rstd[m]};
        asm volatile("" : "+v"(nmean[m]), "+v"(rstd2[m]));   // (also keeps the compiler from carrying the 144 differences x - mean of
                                                             // pass 2 into the loop below: it did, and spilled them)
        const int row = row0 + m * 16;
#if FFN_LN16
        // (even lane of a pair: the fp16 part at its own columns; odd lane: the e4m3 part, starting at its partner's group -- the column
        // offset of a later block adds the same number of bytes to both: mixed_h_offset(cofs) == mixed_x_offset(cofs) - 128 for cofs % 4 == 0)
        const int lbase = (q & 1) ? mixed_x_offset(col0) - 8 : mixed_h_offset(col0);
#else
        const int lbase = mixed_h_offset(col0);
#endif
        lrow[m] = (gchar_t*)((!MID && MODE == 2 && g.ln1_out ? g.ln1_out : g.ln_out) + (size_t)(row < g.M ? row : g.M - 1) * kRow1 + lbase);
        asm volatile("" : "+v"(lrow[m]));
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int cofs = t * FC + (i >> 1) * 64 + (i & 1) * 16;         // column offset: a multiple of 16, so inside a 64-column block
          const int col = col0 + cofs;
          const f32x4 wv = *(const f32x4*)(gb + col), bv = *(const f32x4*)(gb + kDim + col);
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const f32x4 xv = acc2[t][i][m];                                     // (packed fp32 instructions; same operations per element as before)
            const f32x2 ylo = (f32x2{xv[0], xv[1]} + nmean[m]) * rstd2[m] * f32x2{wv[0], wv[1]} + f32x2{bv[0], bv[1]};
            const f32x2 yhi = (f32x2{xv[2], xv[3]} + nmean[m]) * rstd2[m] * f32x2{wv[2], wv[3]} + f32x2{bv[2], bv[3]};
            const f32x4 y = {ylo[0], ylo[1], yhi[0], yhi[1]};
            u32x2 h, xy;
            mixed_pack4(y, h, xy);
#if FFN_LN16
            {
              const bool odd = (q & 1) != 0;
              const uint32_t s0 = odd ? h[0] : xy[0], s1 = odd ? h[1] : xy[1];      // what the partner stores
              uint32_t r0, r1;
              if constexpr (MID) {      // partner = lane ^ 16: rows of 16 lanes swapped pairwise
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                const u2v a0 = __builtin_amdgcn_permlane16_swap(s0, s0, false, false), a1 = __builtin_amdgcn_permlane16_swap(s1, s1, false, false);
                r0 = odd ? a0[0] : a0[1];
                r1 = odd ? a1[0] : a1[1];
              } else {                  // partner = lane ^ 1
                r0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xf, 0xf, true);
                r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xf, 0xf, true);
              }
              const u32x4 v16 = odd ? u32x4{r0, r1, xy[0], xy[1]} : u32x4{h[0], h[1], r0, r1};
              typedef __attribute__((address_space(1))) u32x4 gu32x4_t;
              if (row0 + m * 16 < g.M) *(gu32x4_t*)(lrow[m] + mixed_h_offset(cofs)) = v16;
              continue;
            }
#endif
            if (row0 + m * 16 < g.M) {
#if FFN_NT & 2
              if constexpr (!MID) {
                __builtin_nontemporal_store(h, (gu32x2_t*)(lrow[m] + mixed_h_offset(cofs)));
                __builtin_nontemporal_store(xy, (gu32x2_t*)(lrow[m] + mixed_x_offset(cofs)));
              } else
#endif
              {
              *(gu32x2_t*)(lrow[m] + mixed_h_offset(cofs)) = h;       // (col0 < 48 is a multiple of 4 and cofs % 64 is 0 or 16: both byte
              *(gu32x2_t*)(lrow[m] + mixed_x_offset(cofs)) = xy;      // offsets of column col0 + cofs are those of cofs plus 2 col0)
              }
            }
          }
        }
      EP(5);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the two prefetched stages of the next panel have landed too)
    EP(6);
  };

  for (int it = 0; it < my_panels; ++it) {
    cur_it = it;
    const int panel = b + it * G;
    const char* a_panel = panel_base(it);
    f32x4 acc2[3][6][2];     // [column third t][block i][row group m]
    {
      // The accumulators START as the panel's residual rows, read straight into the MFMA layout: the GEMMs accumulate on top of x, so
      // the new residual stream is complete in registers when the last stage ends -- no residual buffers, no lane transposition,
      // and the rows arrive under the first stages (the compiler sees these loads and waits for each one in front of its first
      // MFMA; the DMA instructions it does not see are all younger or already drained, which only makes its counts conservative).
      const int lane_r = lane_now();
      const int r = lane_r & 15, q = lane_r >> 4;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            int row = panel * FR + wm * 32 + m * 16 + r;
            if (row >= g.M) row = g.M - 1;          // clamp: rows past the end are never stored
            if ((FFN_ABLATE & 256) || kMidRes) {
              acc2[t][i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
              // (opaque zeros: knowing them, the compiler would peel the first stages into MFMAs with a constant C operand in fresh registers)
              if (kMidRes) asm volatile("" : "+v"(acc2[t][i][m]));
            } else if constexpr (RF24) {
              const u32x3 d = *(const u32x3*)((const char*)g.resid + ((size_t)row * kDim + wn * 32 + q * 4 + t * FC + (i >> 1) * 64 + (i & 1) * 16) * 3);
              acc2[t][i][m] = unpack_f24x4(d[0], d[1], d[2]);
            }
#if FFN_NT & 4
            else acc2[t][i][m] = __builtin_nontemporal_load((const f32x4*)(g.resid + (size_t)row * g.ldr + wn * 32 + q * 4 + t * FC + (i >> 1) * 64 + (i & 1) * 16));
#else
            else acc2[t][i][m] = *(const f32x4*)(g.resid + (size_t)row * g.ldr + wn * 32 + q * 4 + t * FC + (i >> 1) * 64 + (i & 1) * 16);
#endif
          }
      if (it == 0) {
        if ((FFN_ABLATE & 256) || kMidRes) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(36)" ::: "memory");   // the prologue's stages 0 and 1 have landed (they are older than the 36 loads)
        skip = 2;
      }
    }
    f32x4 gbv = f32x4{0.f, 0.f, 0.f, 0.f};   // MODE 2: this thread's 16 bytes of LayerNorm2's gamma | beta | the out-projection bias
    (void)gbv;
    if constexpr (MODE != 0) {
      // ---- out projection: 18 k-slices x 3 column thirds.  MODE 2: nothing is prefetched behind position 53 (the FeedForward
      // phase streams the LayerNorm2 rows the epilogue below has yet to write)
      const bool stream_ends = MODE == 2 || it == my_panels - 1;
      const char* a_next = panel_base(it + 1);          // (not dereferenced when the stream ends)
      i32x4 fa0[2], fa1[2];
      // kMidRes: the residual rows of column third t are requested at the top of position kResLoad(t) and added at the top of position
      // kResAdd(t) (a position of that third; the requests of the next third follow the add: one set of 48 registers)
      f32x4 rres[6][2];
      (void)rres;
      auto res_load_pos = [](int t) { return t == 0 ? 0 : t == 1 ? 7 : 17; };
      auto res_add_pos = [](int t) { return t == 0 ? 6 : t == 1 ? 16 : 26; };
      static_for<0, kOutPos>([&](auto p_tag) {
        constexpr int P = decltype(p_tag)::value, KS = P / 3, T = P % 3;
        STAMP(t0);
        if (skip > 0) --skip;
        else if (P == kOutPos - 1) {
          if (stream_ends) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else if ((P + 1) % 3 == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        STAMP(t1);
        wg_barrier();
        STAMP(t2);
        ACC(s_wait, t1, t0); ACC(s_bar, t2, t1);
        if constexpr (kMidRes) {
          static_for<0, 3>([&](auto t_tag) {
            constexpr int TT = decltype(t_tag)::value;
            if constexpr (P == res_add_pos(TT)) {
              static_assert(res_add_pos(TT) % 3 == TT && res_add_pos(TT) >= res_load_pos(TT) + 2, "an interval of the third, two waits behind the requests");
              // (the requests are two of this wave's vmcnt waits old -- each covers everything older than the next stage's DMA instructions --;
              // the launder ties the values to THIS point: the compiler must not move the adds in front of the waits, which it cannot see)
#pragma unroll
              for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                  asm volatile("" : "+v"(rres[i][m]));
                  acc2[TT][i][m] += rres[i][m];
                }
            }
          });
          static_for<0, 3>([&](auto t_tag) {
            constexpr int TT = decltype(t_tag)::value;
            if constexpr (P == res_load_pos(TT)) {
              static_assert(TT == 0 || res_load_pos(TT) > res_add_pos(TT - 1), "the registers are free again");
              const int lane_r = lane_now();
              const int rr_ = lane_r & 15, qq_ = lane_r >> 4;
#pragma unroll
              for (int m = 0; m < 2; ++m) {
                int row = panel * FR + wm * 32 + m * 16 + rr_;
                if (row >= g.M) row = g.M - 1;          // clamp: rows past the end are never stored
                const float* rp = g.resid + (size_t)row * g.ldr + wn * 32 + qq_ * 4 + TT * FC;
                // (inline asm: invisible to the compiler's wait counting, like the LDS-DMA; the kernel's own waits cover them)
                gload16<0>(rres[0][m], rp); gload16<64>(rres[1][m], rp); gload16<256>(rres[2][m], rp);
                gload16<320>(rres[3][m], rp); gload16<512>(rres[4][m], rp); gload16<576>(rres[5][m], rp);
              }
            }
          });
        }
        auto dma = [&](int k) {
          constexpr int P2 = P + 2;
          if constexpr (P2 < kOutPos) issue_out(a_panel, P2 / 3 * SL, P2 % 3, P2 % 3, k);
          else if (!stream_ends) issue_out(a_next, (P2 - kOutPos) / 3 * SL, (P2 - kOutPos) % 3, (P2 - kOutPos) % 3, k);
        };
        stage(Tag<(FAST ? 0 : KS & 1)>(), Tag<0>(), Tag<T * kSlot>(), acc2[T], MODE == 2 ? sco : sc2, dma, Tag<(T > 0)>(), fa0, fa1, [](int) {}, [](int) {});
        if constexpr (MODE == 2 && P == kOutPos - 2) {
          // the phase's last DMA instruction is long out: the parameters of the mid-panel epilogue are requested here and drained by
          // the wait of position 53
          const int tl = w * 64 + lane_now();
          if (tl < 432) gbv = tl < 144 ? ((const f32x4*)g.lnm_w)[tl] : tl < 288 ? ((const f32x4*)g.lnm_b)[tl - 144] : ((const f32x4*)g.bo)[tl - 288];
        }
        STAMP(t0);
        ACC(s_cmp, t0, t2);
      });
    }
    if constexpr (MODE == 2) {
      // ---- x1 = x + a Wo^T + bo stays in the accumulators (the FeedForward's residual needs no second read: fc2 accumulates on top
      // of it); LayerNorm2(x1) goes to ln_out as mixed rows.  Those rows are re-read by this CU's DMA right away, through an L1
      // that may still hold lines of the attention-output rows they replace: stores drained, one L1 invalidate per workgroup.
      STAMP(t0);
      epilogue(acc2, panel, Tag<1>(), gbv);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wg_barrier();
      if (w == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      wg_barrier();
#pragma unroll
      for (int k = 0; k < 5; ++k) issue_fc1(ffn_base(it), 0, 0, 0, k);
#pragma unroll
      for (int k = 0; k < 5; ++k) issue_fc1(ffn_base(it), 0, 1 * SL, 1, k);
      skip = 0;
      STAMP(t1); ACC(s_epi, t1, t0);
    }
    if constexpr (MODE != 1)
    for (int c = 0; c < kChunks; ++c) {
      const bool stream_ends = it == my_panels - 1 && c == kChunks - 1;   // no stage behind this chunk
      const int c_next = c == kChunks - 1 ? 0 : c + 1;
      const char* const f_panel = ffn_base(it);
      const char* a_next = c == kChunks - 1 ? (MODE == 2 ? panel_base(it + 1) : ffn_base(it + 1)) : f_panel;   // (not dereferenced when the stream ends)
      f32x4 acc1[6][2];        // [block i][row group m]
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc1[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};

      i32x4 fa0[2], fa1[2];     // activation fragments: re-read every fc1 stage, held across the three column thirds in fc2
      static_for<0, kPer>([&](auto p_tag) {
        constexpr int P = decltype(p_tag)::value;
        constexpr int Q = P >= kS1 ? P - kS1 : 0;                        // fc2 sub-stage index (P >= kS1)
        constexpr int J = Q / (3 * NK), S = Q % (3 * NK), KIND2 = S / 3, T = S % 3;    // hidden block, sub-stage of the block = kind x third
        if constexpr (P == kS1) {                                         // hidden block 0 of the chunk: nothing to overlap it with
          STAMP(t0);
          mfma_drain4(acc1[0][0], acc1[0][1], acc1[1][0], acc1[1][1]);   // (every fc1 MFMA of the chunk is behind these three)
          mfma_drain4(acc1[2][0], acc1[2][1], acc1[3][0], acc1[3][1]);
          mfma_drain4(acc1[4][0], acc1[4][1], acc1[5][0], acc1[5][1]);
          if (!(FFN_ABLATE & 8)) hidden_write(Tag<0>(), c, acc1);
          STAMP(t1); ACC(s_hid, t1, t0);
        }
        // top of the interval: this wave's part of stage P has landed (all but the instructions of stage P + 1 are done),
        // everybody's after the barrier, and the slot of stage P - 1 is free
        STAMP(t0);
        TL(P, 0);
        if (skip > 0) --skip;
        else if (P == kPer - 1) {
          if (stream_ends) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
#if FFN_FP6_PROBE
        // (this wave's DMA instructions of stage P + 1 -- fewer in a correction stage, and they differ per wave)
        else if (P + 1 < kS1 && ((P + 1) & 1)) {
          if (w < 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
          else if (w < 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else if (P + 1 >= kS1 && ((P + 1 - kS1) % 6) >= 3) {
          if (w < 2 || w == 7) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
#endif
        else if (P + 1 < kS1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (P >= kS1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's hidden-image stores are in the LDS
        STAMP(t1);
        TL(P, 1);
        wg_barrier();
        STAMP(t2);
        TL(P, 2);
        ACC(s_wait, t1, t0); ACC(s_bar, t2, t1);
        auto dma = [&](int k) {       // instruction k of this wave's share of the stage two positions on
          constexpr int P2 = P + 2, S2 = P2 % kRing;
          if constexpr (P2 < kS1) {
            if (k < 5) issue_fc1(f_panel, c, P2 * SL, S2, k);
          } else if constexpr (P2 < kPer) {
            constexpr int Q2 = P2 - kS1;
            if (k < 3) issue_fc2((c * 3 + Q2 / (3 * NK)) * 2 + (Q2 % (3 * NK)) / 3, Q2 % 3, S2, k);
          } else {
            if (k < 5 && !stream_ends) {
              if (MODE == 2 && c == kChunks - 1) issue_out(a_next, 0, P2 - kPer, S2, k);   // the next panel opens with its out projection
              else issue_fc1(a_next, c_next, (P2 - kPer) * SL, S2, k);
            }
          }
        };
        constexpr int SB = (P % kRing) * kSlot;
        auto mark = [&](int i) { (void)i; TL(P, 3 + i); };
        if constexpr (P < kS1) {
          stage(Tag<(FAST ? 0 : (P % 2) * (FFN_FP6_PROBE ? 2 : 1))>(), Tag<SB>(), Tag<SB>(), acc1, sc1, dma, Tag<0>(), fa0, fa1, [](int) {}, mark);
        } else {
          // beside the MFMAs of sub-stages 0..3 of block J: bias + GELU + conversion of one (16 columns x 16 rows) unit of block
          // J + 1, a value per group, packed and stored in group 4
          // (single-pass form: a block has three sub-stages, so two units per sub-stage -- one in groups 0..2, one in groups 3..5 -- in the first two)
          constexpr bool CONV = J < 2 && S < (FAST ? 2 : 4) && !(FFN_ABLATE & 8);
          f32x4 gv, pre;
          int ho = 0;
          auto valu = [&](int i) {
            if constexpr (CONV) {
              const int unit = FAST ? 2 * S + (i >= 3) : S, ii = FAST ? i % 3 : i - FFN_CONV_G0;     // (constants once the group loop is unrolled)
              const int IB = (unit >> 1) & 1, MM = unit & 1;
              if (FAST ? ii == 0 : i == 0) {
                const int lane_h = lane_now();
                const int hr = lane_h & 15, hq = lane_h >> 4;
                const int ls = wn * 4 + IB * 2 + (hq >> 1);
                ho = (wm * 32 + MM * 16 + hr) * 128 + ((ls ^ ((hr >> 1) & 7)) << 4) + (hq & 1) * 8;
                pre = acc1[2 * (J + 1) + IB][MM] + *(const f32x4*)(smem + kB1Off + (c * FC + (J + 1) * 64 + wn * 32 + IB * 16 + hq * 4) * 4);
              }
              if (ii == 0 || ii == 1) {          // two values per group on the packed fp32 instructions
                const int k = 2 * ii;
                const f32x2 g2 = gelu_sigmoid2(f32x2{pre[k], pre[k + 1]});
                gv[k] = g2[0];
                gv[k + 1] = g2[1];
                asm volatile("" : "+v"(gv[k]), "+v"(gv[k + 1]));   // pins the piece in its group (register-only code carries no order of its own)
              }
              if (ii == 2) {
                u32x2 h, xy;
                mixed_pack4(gv, h, xy);
                if constexpr (FAST) {
                  asm volatile("" : "+v"(h));
                  *(u32x2*)(smem + hid_f16(J + 1) + ho) = h;
                } else {
                  asm volatile("" : "+v"(h), "+v"(xy));
                  *(u32x2*)(smem + hid_f16(J + 1) + ho) = h;
                  *(u32x2*)(smem + hid_e4m3(J + 1) + ho) = xy;
                }
              }
            }
          };
          stage(Tag<(FAST ? 0 : KIND2 * (FFN_FP6_PROBE ? 2 : 1))>(), Tag<(KIND2 == 0 ? (J == 1 ? 0 : kHidOff) : (J == 1 ? kSlot : kHidOff + kAB))>(), Tag<SB>(), acc2[T], sc2, dma,
                Tag<(T > 0)>(), fa0, fa1, valu, mark);
        }
        STAMP(t0);
        ACC(s_cmp, t0, t2);
      });
    }

    STAMP(t0);
    EP(0);
    if (!(FFN_ABLATE & 16)) {
      epilogue(acc2, panel, Tag<0>(), f32x4{0.f, 0.f, 0.f, 0.f});
      skip = 2;   // stages 0 and 1 of the next panel landed before the drain inside
    } else {
      asm volatile("" ::"v"(acc2[0][0][0]), "v"(acc2[2][5][1]));
    }
    STAMP(t1); ACC(s_epi, t1, t0);
  }
#ifdef VETO_FFN_STAMPS
  if (w == 0 && lane == 0) {
    unsigned long long* o = g_ffn_stamps + (size_t)(b & 255) * 8;
    o[0] = s_wait; o[1] = s_bar; o[2] = s_iss; o[3] = s_cmp; o[4] = s_hid; o[5] = s_epi; o[6] = t1 - t_begin; o[7] = my_panels;
  }
#endif
}

}  // namespace

int ffn_panel_rows() { return FR; }

namespace {
hipError_t launch_panel(FfnArgs g, int mode, hipStream_t s) {
  if (g.M <= 0 || !g.a || !g.w2 || !g.b2 || !g.resid || !g.out || !g.exp2) return hipErrorInvalidValue;
  if (mode != 1 && (!g.w1 || !g.b1 || !g.exp1)) return hipErrorInvalidValue;
  if (mode == 2 && (!g.wo || !g.bo || !g.expo || !g.lnm_w || !g.lnm_b || !g.ln_out)) return hipErrorInvalidValue;
  if (g.fast && (mode != 2 || g.resid_f24 || g.out_f24)) return hipErrorInvalidValue;      // (the single-pass form: the layer tail on fp32 residual rows)
  if ((g.resid_f24 || g.out_f24) && (mode != 2 || !g.resid_f24)) return hipErrorInvalidValue;      // (3-byte rows: the layer tail only; never f32 in, 3 bytes out)
  if (g.resid_f24 != g.out_f24 && (const void*)g.resid == (const void*)g.out) return hipErrorInvalidValue;   // (rows of different pitch cannot be rewritten in place)
  int num_cu = device_cu_count();
  if (num_cu < 1) return hipErrorInvalidDevice;
  g.n_panels = (g.M + FR - 1) / FR;
  static const int late = env_knob_int("VETO_FFN_LATE", 90);      // (speed only; 0 in one arm of the parity tests)
  g.late = late;
  const int nblocks = g.n_panels < num_cu ? g.n_panels : num_cu;   // one persistent workgroup per CU (LDS: 159 KiB each)
  if (mode == 0) VETO_LAUNCH(ffn_fused_kernel<0>, dim3(nblocks), dim3(512), 0, s, g);
  else if (mode == 1) VETO_LAUNCH(ffn_fused_kernel<1>, dim3(nblocks), dim3(512), 0, s, g);
  else if (g.fast) VETO_LAUNCH((ffn_fused_kernel<2, false, false, true>), dim3(nblocks), dim3(512), 0, s, g);
  else if (g.resid_f24 && g.out_f24) VETO_LAUNCH((ffn_fused_kernel<2, true, true>), dim3(nblocks), dim3(512), 0, s, g);
  else if (g.resid_f24) VETO_LAUNCH((ffn_fused_kernel<2, true, false>), dim3(nblocks), dim3(512), 0, s, g);
  else VETO_LAUNCH(ffn_fused_kernel<2>, dim3(nblocks), dim3(512), 0, s, g);
  hipError_t rc = hipGetLastError();
#ifdef VETO_FFN_STAMPS
  {
    static unsigned long long host[256 * 8];
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ffn_stamps), sizeof(host));
    double sum[8] = {0};
    const int nb = nblocks < 256 ? nblocks : 256;
    for (int bb = 0; bb < nb; ++bb)
      for (int k = 0; k < 8; ++k) sum[k] += (double)host[bb * 8 + k];
    static int printed = 0;
    {
      static unsigned long long ep[2 * 12];
      hipMemcpyFromSymbol(ep, HIP_SYMBOL(g_ffn_epi), sizeof(ep));
      for (int wv = 0; wv < 2; ++wv) {
        fprintf(stderr, "[ffn epilogue] wave %d:", wv * 4);
        for (int k = 1; k < 7; ++k) fprintf(stderr, " %6lld", (long long)(ep[wv * 12 + k] - ep[wv * 12 + k - 1]));
        fprintf(stderr, "  (bias adds | x stores | LN pass 1 | pass 2 | normalise+stores | drain)\n");
      }
    }
    if (mode != 1 && !printed++) {
      static unsigned long long tl[2 * 36 * 9];
      hipMemcpyFromSymbol(tl, HIP_SYMBOL(g_ffn_timeline), sizeof(tl));
      for (int wv = 0; wv < 2; ++wv)
        for (int p = 0; p < 36; ++p) {
          const unsigned long long* r = tl + (wv * 36 + p) * 9;
          const unsigned long long t00 = tl[(wv * 36) * 9];
          fprintf(stderr, "[ffn timeline] wave %d pos %2d: top %7llu | wait %5llu bar %5llu | groups", wv * 4, p, r[0] - t00, r[1] - r[0], r[2] - r[1]);
          for (int k = 0; k < 6; ++k) fprintf(stderr, " %5llu", r[3 + k] - r[2 + k]);
          fprintf(stderr, "\n");
        }
    }
    fprintf(stderr, "[ffn stamps M%d] wave 0 of each workgroup, mean cycles: load wait %.0f barrier %.0f dma issue %.0f mfma %.0f hidden %.0f "
            "epilogue %.0f total %.0f (%.2f panels x %d chunks x %d stages)\n", g.M, sum[0] / nb, sum[1] / nb, sum[2] / nb, sum[3] / nb,
            sum[4] / nb, sum[5] / nb, sum[6] / nb, sum[7] / nb, kChunks, 36);
  }
#endif
  return rc;
}

}  // namespace

hipError_t launch_ffn_fused(FfnArgs g, hipStream_t s) { return launch_panel(g, 0, s); }
// x <- x + a W^T + b (attention out projection + residual) with optional LayerNorm rows; w2 = W [576, 4*576 B] mixed rows,
// exp2 its exponent, b2 the bias; w1 / b1 / exp1 unused
hipError_t launch_out_fused(FfnArgs g, hipStream_t s) { return launch_panel(g, 1, s); }
hipError_t launch_layer_tail(FfnArgs g, hipStream_t s) { return launch_panel(g, 2, s); }

}  // namespace veto
