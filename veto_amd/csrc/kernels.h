// Internal launch interface between the C-ABI host code (veto_abi.hip) and the kernels.
#pragma once
#include "common.h"

namespace veto {

enum Epi { EPI_F32 = 0, EPI_RESID = 1, EPI_GELU_SPLIT = 2, EPI_ATOMIC = 3, EPI_RESID_DROP = 4, EPI_F24 = 5, EPI_SPLIT = 6, EPI_PRE_GELU = 7, EPI_GELU_BWD = 8 };   // EPI_F24: EPI_F32 written as 3-byte floats (common.h); EPI_SPLIT: split rows, no activation; EPI_PRE_GELU (training, split rows): the fp32 pre-activation to c (row stride ldc_f32) AND exact-erf gelu of it as split rows to c_split

// C[M,N] = A[M,K] . W[N,K]^T with A and W in the split-row format (common.h): rows of 2K bf16.
struct GemmArgs {
  const __bf16* a;     // row r at a + r*lda; rows padded to a multiple of 256 when lda == 2K
  const __bf16* w;     // [N, 2K]
  const float* bias;   // [N] or nullptr
  const float* resid;  // EPI_RESID: row r at resid + r*ldr
  float* c;            // EPI_F32 / EPI_RESID: row r at c + r*ldc; EPI_F24: row r at (char*)c + 3*r*ldc, 3 bytes per element
  __bf16* c_split;     // EPI_GELU_SPLIT / EPI_SPLIT: split-row output, row r at c_split + r*ldc (ldc = 2N)
  int M, N, K;
  long lda;            // A row stride in bf16 elements (0 = 2K)
  long ldr;
  long ldc;
  long ldc_f32;        // EPI_PRE_GELU: row stride of c in floats (ldc is c_split's)
  // EPI_GELU_BWD (training, split rows): the input gradient of fc2 times gelu'(pre) -- resid = the fp32 pre-activation (row stride ldr) --
  // written straight as the split rows of fc1's backward (c_split, ldc), with the column sums of every 64-row slice of a tile (the rows of
  // one wave row) to col_partial[(tile_m * 4 + wave row) * N + column]: the bias gradient's partials, folded by launch_column_sums
  float* col_partial;
  int k_splits;          // EPI_ATOMIC: the reduction K is cut into this many equal ranges (K/32 % k_splits == 0), each a
                         // tile of its own that ADDS into c with fp32 atomics (c zero-initialised); 0/1 = one range
  // EPI_RESID_DROP (training only): EPI_RESID with dropout on (A.W^T + bias) before the residual is added; element (row, col) uses index
  // row * N + col of site drop_seed.  drop_thresh = p * 2^24 (0 = off), drop_scale = 1 / (1 - p)
  unsigned long long drop_seed;
  unsigned drop_thresh;
  float drop_scale;
  int tiles_m, tiles_n;  // filled by the launcher
  // tn (EPI_ATOMIC only, weight gradients): C[M, N] += A^T B with BOTH operands row-major in the reduction index, i.e. the
  // split rows the backward already holds: a = dY split rows [k_valid, lda] (M = its column count), w = X split rows
  // [k_valid, ldw] (N = its column count), K = the reduction length rounded up to 32 * k_splits.  Reduction rows >= k_valid
  // are read from `zero` (>= 1 KiB of zeros).  The operands are transposed on their way out of LDS (ds_read_b64_tr_b16).
  // fmt == FMT_MIXED (common.h): a and w are mixed rows (fp16 + e4m3), w_exp points to the weight tensor's e4m3 exponent
  // on the device, and an EPI_GELU_SPLIT output is written as mixed rows too.  Inference forms only (no tn / split-K).
  int fmt;
  // Block-diagonal weights (kb_tiles > 0; split rows, no split-K, not tn): column tile n multiplies only the k-steps
  // [(n / kb_tiles) * kb_steps, + kb_steps) of the rows -- the products of the folded last layer (veto_abi.hip) are block-diagonal
  // over the heads, and the zero blocks are skipped instead of multiplied.
  int kb_tiles, kb_steps;
  const int* w_exp;
  int tn;
  long ldw;
  int k_valid;
  const __bf16* zero;
};

// Environment knobs (veto_abi.hip): every one selects a path that a parity test compares against the default (INTEGRATION.md
// lists them); call sites cache the answer in a function-local static.
bool env_knob_is(const char* name, const char* value);   // the variable is set to exactly `value`
int env_knob_int(const char* name, int dflt);
int gemm_rows_padded(int m);
// compute units of the CURRENT device (hipGetDevice), cached per device ordinal: the persistent kernels size their grids with it
// (a process-wide cache of the first device's count would be wrong on a node with unlike devices); <= 0 on error
int device_cu_count();
// precision: 0 = three split-bf16 terms, 1 = one; g.fmt == FMT_MIXED selects the fp16 + e4m3 kernel regardless
hipError_t launch_gemm_split(GemmArgs g, int epi, int precision, hipStream_t s);
hipError_t launch_gemm_split_ps(GemmArgs g, int epi, int precision, hipStream_t s);

// Fused FeedForward (ffn_fused.hip), VETO_MIXED operands: out = resid + W2 . gelu(W1 . a + b1) + b2 on `M` token rows.
// a: mixed activation rows [rows padded to ffn_panel_rows(), 4*576 B]; w1 [1152, 4*576 B], w2 [576, 4*1152 B]: mixed weight
// rows with their e4m3 exponents exp1 / exp2 (device ints); resid / out fp32 rows (may alias).
struct FfnArgs {
  const char* a;
  const char* w1;
  const char* w2;
  const float* b1;
  const float* b2;
  const float* resid;
  float* out;
  long ldr, ldo;         // row strides of resid / out in floats
  int M;
  const int* exp1;
  const int* exp2;
  // optional: LayerNorm (eps 1e-5) of the result rows with ln_w / ln_b, written as mixed activation rows to ln_out (row pitch
  // 4*576 B; may be the buffer `a` points to: a panel's rows are written after its last read) -- the next layer's PreNorm
  const float* ln_w;
  const float* ln_b;
  char* ln_out;
  char* ln1_out;         // launch_layer_tail only, optional: where the final LayerNorm rows (ln_w / ln_b) go instead of ln_out -- with the
                         // fused QKV + attention launch the attention output (a) and the next layer's LayerNorm1 rows live in different buffers
  // launch_layer_tail only: the attention out projection in front of the FeedForward -- x1 = resid + a . wo^T + bo stays in
  // registers, lnm_w / lnm_b (LayerNorm2) turn it into the FeedForward's input rows, written to ln_out (= a, in place)
  const char* wo;        // [576, 4*576 B] mixed weight rows, exponent expo
  const float* bo;
  const int* expo;
  const float* lnm_w;
  const float* lnm_b;
  // launch_layer_tail only: resid / out are rows of 576 3-byte floats (common.h, pack_f24x4; ldr / ldo unused) instead of fp32 rows.  out_f24
  // needs resid_f24; with different formats the two must be different buffers
  int resid_f24, out_f24;
  int fast;              // launch_layer_tail only (VETO_FAST): the fp16 main product alone -- the correction stages are neither loaded nor multiplied
  int n_panels;          // filled by the launcher
  int late;              // filled by the launcher (speed only): start delay (~us) of the workgroups that have one panel fewer than the others
};
int ffn_panel_rows();
hipError_t launch_ffn_fused(FfnArgs g, hipStream_t s);
// the same skeleton as ONE GEMM: out = resid + a . w2^T + b2 (K = 576: the attention out projection + residual), optional LayerNorm rows
hipError_t launch_out_fused(FfnArgs g, hipStream_t s);
// out projection + residual + LayerNorm2 + FeedForward + residual (+ the next layer's LayerNorm1) in ONE launch: everything of a
// transformer layer behind its attention (model_veto.py:96, :20-21, :125-143).  a = attention output rows (mixed), resid = x in,
// out = x out, ln_out = a (the LayerNorm2 rows, then the next layer's LayerNorm1 rows, in place); wo / bo / expo / lnm_* as above
hipError_t launch_layer_tail(FfnArgs g, hipStream_t s);

// ---- weight preparation (once per weight upload) ---------------------------------------------
// src [rows, K] fp32 -> dst [rows, 2K] split rows
hipError_t launch_split_rows(const float* src, __bf16* dst, size_t rows, int K, hipStream_t s);
// block-form weights of the folded last layer (rowops.hip), fp32: which = 0 Wq padded, 1 Wk^T block-diagonal, 2 Wv block-diagonal, 3 Wo padded
hipError_t launch_fold_blocks(const float* qkv, const float* wo, float* out, int which, int heads, int dhp, hipStream_t s);
// src [rows, K] fp32 -> dst [rows, 4K bytes] mixed WEIGHT rows (common.h); *exp_out (device) receives the tensor's e4m3 exponent e:
// the largest e in [0, 24] with 2^e max|w| <= 448
hipError_t launch_mixed_act_rows(const float* src, __bf16* dst, size_t rows, int K, hipStream_t s);
hipError_t launch_mixed_weight_rows(const float* src, __bf16* dst, size_t rows, int K, int* exp_out, hipStream_t s);
// src [M, ld] fp32 (N columns used) -> dst [N, 2*Mp] split rows of the TRANSPOSE, Mp = M rounded up to 32*mult
// (zero padded): the operand layout of a GEMM that reduces over M (weight gradients)
hipError_t launch_transpose_split(const float* src, long ld, int M, int N, __bf16* dst, int Mp, hipStream_t s);
// W_cat [1152, 2*2048] split rows + bias_cat [1152] from proj_d [512,2048], proj_v [64,2048]
hipError_t launch_build_patch_weight(const float* wd, const float* bd, const float* wv, const float* bv,
                                     __bf16* dst, float* bias_cat, hipStream_t s);
// dst[k][half*576 + j] = src[j][half*kin + k]   (src is [576, 2*kin])
hipError_t launch_transpose_pair_proj(const float* src, float* dst, int kin, hipStream_t s);
// dst[k][c] = src[c][k]  (src [n_out, 576])
hipError_t launch_transpose_head(const float* src, float* dst, int n_out, hipStream_t s);

// ---- per-object stage --------------------------------------------------------------------------
struct ObjPrepArgs {
  const float* boxes;        // [n_obj, 4]
  int box_mode;              // 0 xyxy, 1 xywh
  const int64_t* labels;     // predcls / MEET: embedding lookup
  const float* obj_logits;   // sgcls: softmax(logits) @ E
  const float* embed;        // [num_obj_cls, embed_dim]
  int num_obj_cls, embed_dim;
  const float* bn_w; const float* bn_b; const float* bn_mean; const float* bn_var;  // [4]
  const float* pos_w; const float* pos_b;   // [128,4], [128]
  const float* loc_wt; const float* loc_b;  // [128][1152] transposed, [576]
  const float* cls_wt; const float* cls_b;  // [embed_dim][1152] transposed, [576]
  float* lc;                 // out [n_obj, 2, 1152]: (location | class) x (subj(+bias) | obj)
  float* pos_out;            // optional debug [n_obj,128]
  int n_obj;
  // training only: Dropout(0.1) behind the position embedding's ReLU; element (n, k) -> index n*128 + k
  unsigned long long drop_seed;
  unsigned drop_thresh;
  float drop_scale;
};
hipError_t launch_obj_prep(const ObjPrepArgs& a, hipStream_t s);
// training-mode BatchNorm1d(4) statistics of the box features center_xywh(boxes): out[0..3] mean, [4..7] biased
// variance (what the batch is normalised with), [8..11] unbiased variance (what running_var is updated with)
hipError_t launch_bn_batch_stats(const float* boxes, int box_mode, int n_obj, float* out, hipStream_t s);
// rgb/depth [n_obj, 256, 8, 8] -> patch rows [n_obj*16, 2*2048] split rows (depth features first)
hipError_t launch_patchify(const float* depth, const float* rgb, __bf16* dst, int n_obj, hipStream_t s);

// ---- pair stage --------------------------------------------------------------------------------
hipError_t launch_pair_indices(const int64_t* rel_pairs, const int32_t* img_obj_off,
                               const int32_t* img_pair_off, int n_img, int n_pair, int32_t* subj,
                               int32_t* obj, int64_t* subj64, int64_t* obj64, hipStream_t s);
hipError_t launch_enumerate_pairs(int n, int64_t* out, hipStream_t s);

struct AssembleArgs {
  const float* patch_tab;   // [n_obj*16, 1152]
  const float* lc;          // [n_obj, 2, 1152]
  const float* cls_token;   // [576]
  const float* pos_embedding;  // [576]
  const float* ln_w; const float* ln_b;  // layer-0 attention PreNorm
  const int32_t* subj; const int32_t* obj;  // this chunk's pairs
  float* x;                 // [n_pair*19, 576]
  __bf16* a;                // LN(x), split rows [n_pair*19, 2*576]
  float* stats;             // optional [n_pair*19, 2]: (mean, rstd) of every row; then `a` is written for tokens 17 / 18 only
  int a_fmt;                // ... in this operand format (FMT_SPLIT / FMT_MIXED; with stats only)
  int x_f24;                // (with stats only) x is written as rows of 576 3-byte floats (common.h) instead of fp32
  int n_pair;
  // training only: pos_drop (EMB_DROPOUT) on the assembled tokens; element (row, col) -> index row*576 + col
  unsigned long long drop_seed;
  unsigned drop_thresh;
  float drop_scale;
};
hipError_t launch_assemble(const AssembleArgs& a, hipStream_t s);
// layer 0 in the per-object form (rowops.hip): qkv rows of tokens 0..16 from the per-object tables sw / ow [n_obj*16, 1728],
// the row statistics and the weight-only vectors vec = [c2 | b0 | qkv_cls]; the kernel that builds vec and Wqkv diag(gamma);
// and the row-centred split rows of the two halves of patch_tab [rows, 1152] -> [rows, 2*1152]
hipError_t launch_centre_split(const float* patch_tab, __bf16* dst, int rows, hipStream_t s);
hipError_t launch_qkv0_combine(const float* sw, const float* ow, const float* stats, const float* vec, const int32_t* subj,
                               const int32_t* obj, float* qkv, int n_pair, hipStream_t s);
hipError_t launch_qkv0_consts(const float* wq, const float* gamma, const float* beta, const float* pos, const float* cls, float* wp,
                              float* vec, hipStream_t s);
// y = dropout mask of site `seed` applied to x ([rows, n_cols], index row*n_cols + col), scaled by `scale`; in place allowed
hipError_t launch_dropout_apply(const float* x, float* y, size_t rows, int n_cols, unsigned long long seed, unsigned thresh,
                                float scale, hipStream_t s);

// LayerNorm(eps 1e-5) of `rows` rows (row r at x + r*ldx) -> split rows [rows, 2*576]
// dst row r at dst + r*ldd (bf16 elements; 0 = contiguous rows of 2*576)
// counters[4] += {elements, fp16 values at +-65504, e4m3 value-plane bytes at +-448, e4m3 residual-plane bytes at +-448} of mixed rows
hipError_t launch_count_saturation(const void* base, long stride_bytes, int rows, int K, unsigned long long* counters, hipStream_t s);
// x_f24: the source rows are 3-byte floats (common.h); ldx still counts elements
hipError_t launch_layernorm(const float* x, long ldx, const float* w, const float* b, __bf16* dst, int rows,
                            hipStream_t s, int fmt = FMT_SPLIT, long ldd = 0, bool x_f24 = false);
// n 3-byte floats (n % 4 == 0) -> fp32
hipError_t launch_unpack_f24(const void* src, float* dst, size_t n, hipStream_t s);

struct AttnArgs {
  const float* qkv;        // [n_pair*19, 1728]; qkv_f24: the same matrix as 3-byte floats (rows of 5184 bytes)
  __bf16* o;               // split rows [rows, 2*576]; rows = n_pair*19, or n_pair when cls_only
  int n_pair, heads, cls_only;
  int qkv_f24 = 0;         // MFMA head widths, not the per-object form
  int o_fmt = FMT_SPLIT;   // operand format of o (MFMA head widths; the generic kernel writes split rows only)
  // layer 0, per-object form (rowops.hip): when sw != nullptr the rows of tokens 1..16 are formed on load as
  // rstd * (sw[subj*16 + t-1] + ow[obj*16 + t-1]) + c2, token 0 is the constant row vec + 2*1728, and only the rows of tokens
  // 17 / 18 are read from qkv.  MFMA head widths (72, 96) only: launch_attention returns hipErrorInvalidValue otherwise.
  const float* sw = nullptr;
  const float* ow = nullptr;
  const float* stats = nullptr;   // [n_pair*19, 2]: (mean, rstd)
  const float* vec = nullptr;     // [c2 | b0 | qkv_cls], each [1728]
  const int32_t* subj = nullptr;
  const int32_t* obj = nullptr;
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
bool attention_reads_tables(int heads);   // whether launch_attention takes the per-object form for this head count

// QKV projection + attention of a middle layer in one launch (qkv_attn_fused.hip; VETO_MIXED, head widths 72 / 96): a = LayerNorm1 rows
// (mixed, [rows padded to qkv_attn_rows_padded(n_pair), 4*576 B]), w = Wqkv as mixed weight rows [1728, 4*576 B] with its e4m3 exponent
// w_exp (device int), o = attention output rows (mixed, row = 19 * pair + token; NOT the buffer a points to: every head's tile reads all
// of a pair group's rows)
struct QkvAttnArgs {
  const char* a;
  const char* w;
  const int* w_exp;
  char* o;
  int n_pair, heads;
  int fast;      // VETO_FAST: the fp16 main product alone (the correction stages are neither loaded nor multiplied)
};
bool qkv_attn_fused_supports(int heads);
size_t qkv_attn_rows_padded(int n_pair);
hipError_t launch_qkv_attn_fused(const QkvAttnArgs& g, hipStream_t s);
// Folded CLS-only attention of the last layer: x = residual stream [n_pair*19, 576] (LayerNorm1 with ln_w / ln_b is applied
// inside), u fp32 [n_pair, heads*576] (= LN1(x_0) . Mcat), abar split rows [n_pair, 2*heads*576] (probability-weighted
// token means per head)
int cls_fold_max_heads();
hipError_t launch_cls_fold_attention(const float* x, const float* ln_w, const float* ln_b, const float* u, __bf16* abar, int n_pair, int heads,
                                     hipStream_t s, bool u_f24 = false);
bool cls_fold_reads_f24();   // whether the kernel takes u as 3-byte floats (the MFMA form; VETO_QKV_F24=0 / VETO_CLS_MFMA=0: fp32)

// cls row p at cls + p*ld (ld = 576 for compact CLS rows, 19*576 to read row 0 of every pair of a token matrix)
hipError_t launch_head(const float* cls, const float* wt, const float* bias, float* out, int n_pair,
                       int n_out, hipStream_t s, long ld = kDim);

// ---- post-processing (inference.py:398-453, GT-box branch) -------------------------------------
struct PostArgs {
  const float* rel_logits;       // [n_pair, n_rel_cls]
  const float* obj_logits;       // [n_obj, n_obj_cls]
  const int64_t* rel_pairs;      // [n_pair, 2] image-local
  const int32_t* img_obj_off;    // [n_img + 1]
  const int32_t* img_pair_off;   // [n_img + 1]
  int n_img, n_obj, n_pair, n_rel_cls, n_obj_cls;
  float* obj_scores;             // out [n_obj]
  int64_t* obj_pred;             // out [n_obj]
  float* out_prob;               // out [n_pair, n_rel_cls], sorted
  int64_t* out_pairs;            // out [n_pair, 2], sorted
  int64_t* out_labels;           // out [n_pair], sorted
  float* out_triple;             // optional out [n_pair], sorted
  float* prob_tmp;               // workspace [n_pair, n_rel_cls]
  float* triple;                 // workspace [n_pair]
  int32_t* label_tmp;            // workspace [n_pair]
  int32_t* perm;                 // workspace [n_pair]
  int single_cnt;                // > 0: one segment of this many rows (MEET merge) instead of img_pair_off
  int pair_mod;                  // > 0: row r refers to pair r % pair_mod (MEET: K copies of the pair list)
  int32_t* kept_count;           // optional out [1]: rows of the sorted segment with score >= 0 (expert voting)
};
struct MeetGroup {               // one MEET head, passed by value
  const float* logits;           // [n_pair, width], width = g + 2
  int width;
  int row0;                      // first row of this group in the merged list
  int cols[104];                 // cols[c] = global class of the group's column c (cols[0] = 0), c < width - 1
};
hipError_t launch_postprocess_meet(PostArgs a, const MeetGroup* groups, int n_groups, hipStream_t s);
struct VoteGroup {               // one MEET group with its three expert heads, passed by value
  const float* logits[3];        // each [n_pair, width]
  int width;
  int row0;
  int cols[104];
};
// voting: 0 = consensus ('C', two of three experts agree), 1 = unanimous ('U')
hipError_t launch_postprocess_vote(PostArgs a, const VoteGroup* groups, int n_groups, int voting, hipStream_t s);
int postprocess_max_pairs_per_image();
hipError_t launch_postprocess(const PostArgs& a, hipStream_t s);

// ---- ROI feature extraction (roialign.hip) ----------------------------------------------------------
struct RoiLevel {
  const float* feat;             // [n_img, C, H, W]
  int H, W;
  float scale;
};
struct RoiPoolArgs {
  RoiLevel lv[4];                // FPN levels of the RGB pyramid, finest first
  RoiLevel depth;                // depth map (feat == nullptr: none)
  int n_levels, k_min, k_max;    // LevelMapper range: -log2(first scale) .. -log2(last scale)
  int n_roi, channels, depth_channels;
  int pooled, sampling_ratio;
  const float* rois;             // [n_roi, 5] (image index, x1, y1, x2, y2)
  float* out_rgb;                // [n_roi, channels, pooled, pooled]
  float* out_depth;              // [n_roi, depth_channels, pooled, pooled]
  int32_t* out_levels;           // optional [n_roi]
  // backward only: gradients of the pooled outputs in, gradients of the maps out (accumulated atomically)
  const float* gout_rgb;         // [n_roi, channels, pooled, pooled]
  const float* gout_depth;       // [n_roi, depth_channels, pooled, pooled] or nullptr
  float* lv_grad[4];             // per level, same shape as lv[l].feat, zero-initialised by the caller
  float* depth_grad;
};
hipError_t launch_roi_pool(const RoiPoolArgs& a, hipStream_t s);
hipError_t launch_roi_pool_backward(const RoiPoolArgs& a, hipStream_t s);

// ---- relation evaluators (sgg_eval.hip) ---------------------------------------------------------------
struct SggEvalArgs {
  int n_img, n_rel_cls, n_zeroshot;
  float iou_thres;
  const int32_t* gt_off;         // [n_img + 1] prefix sum of GT relations
  const int32_t* obj_off;        // [n_img + 1] prefix sum of objects
  const int32_t* pair_off;       // [n_img + 1] prefix sum of predicted pairs
  const int64_t* gt_rels;        // [sum G, 3] (subject, object, predicate), image-local indices
  const int64_t* gt_classes;     // [sum N]
  const float* gt_boxes;         // [sum N, 4] xyxy
  const int64_t* pred_pairs;     // [sum P, 2], in ranking order per image
  const float* rel_scores;       // [sum P, n_rel_cls]
  const int64_t* pred_classes;   // [sum N]
  const float* pred_boxes;       // [sum N, 4]
  const float* obj_scores;       // [sum N]
  const int64_t* zeroshot;       // [n_zeroshot, 3] (subject class, object class, predicate)
  int32_t *gc_rank, *ng_rank, *acc_rank, *zeroshot_flag;   // out [sum G]
  int32_t *ng_rows, *ng_cols;    // out [n_img, 100]
  int32_t* ng_count;             // out [n_img]
  double* metrics;               // out [18 + 6 * (n_rel_cls - 1) + 2]
  int32_t *label_tmp, *flag_tmp, *flag_before;             // workspace [sum P]
  float* pair_score;             // workspace [sum P]
  uint32_t* row_key;             // workspace [sum P]: sortable key of the row's largest cell
  int32_t* acc_first;            // workspace [sum G]
  int32_t* cls_table;            // workspace [n_img, 7, n_rel_cls]
};
hipError_t launch_sgg_eval(const SggEvalArgs& a, hipStream_t s);

// ---- training losses / MEET sampling (losses.hip) ----------------------------------------------------------
struct CeLossArgs {
  const float* logits;           // row r at logits + r*ld
  long ld;
  const int64_t* labels;         // [n], aligned with the selected rows
  const float* weight;           // [C] or nullptr
  const int64_t* rows;           // [n] row indices into logits, or nullptr = rows 0..n-1
  int n, C;
  float *lse, *nll_w, *w_row;    // workspace [n] each
  float* inv_wsum;               // workspace [1]
  float* loss;                   // out [1]
  float* grad;                   // optional out [n, C]
};
hipError_t launch_ce_loss(const CeLossArgs& a, hipStream_t s);

struct MeetSampleArgs {
  const int64_t* labels;         // [n] relation labels
  int n, n_groups, n_cls, n_words;
  const uint32_t* words;         // [n_words] raw MT19937 outputs of Python's random, in order
  const int32_t* incre;          // [n_cls] 1-based group of each class
  const int32_t* pos_in_group;   // [n_cls] 1-based position of the class inside its group
  const int32_t* group_size;     // [n_groups]
  const double* rates;           // [n_groups, n_cls] sample_rate_matrix
  int64_t* chosen;               // out [n_groups, n] row indices per group (first counts[k] valid)
  int64_t* group_labels;         // out [n_groups, n] remapped labels of those rows
  int32_t* counts;               // out [n_groups]
  int32_t* words_used;           // out [1]; -1 = the word block was too short
};
hipError_t launch_meet_sample(const MeetSampleArgs& a, hipStream_t s);

// ---- backward building blocks (backward.hip) -----------------------------------------------------------------
// qkv, dqkv: [n_pair*19, 1728]; dout: [n_pair*19, 576] (gradient of the attention output before the out projection)
// exactly one of dqkv (fp32 [n_pair*19, 1728]) and dqkv_split (split rows [n_pair*19, 2*1728]) is written
// cls_only: dout is compact [n_pair, 576] (gradient of the CLS query's output only), q rows 1..18 of qkv are not read
// qkv_f24 (head widths 72 / 96): qkv holds 3-byte floats (common.h), rows of 1728 x 3 bytes
hipError_t launch_attention_backward(const float* qkv, const float* dout, float* dqkv, __bf16* dqkv_split, int n_pair, int heads, int cls_only,
                                     hipStream_t s, bool qkv_f24 = false);
// dx = LayerNorm backward of dy w.r.t. x (+ dres if given); dgamma_dbeta [2, 576]; partial: workspace of
// layernorm_backward_partial_floats(rows) floats
size_t layernorm_backward_partial_floats(int rows);
// split_out / colp (both or neither): also write the result's split rows [rows, 2 * 576] (with the dropout mask of site drop_seed applied
// when drop_thresh != 0: element (row, col) -> index row * 576 + col) and layernorm_backward_col_partials(rows) rows of column sums [*, 576]
// of those rows -- what launch_prep_grad would produce from dx, for the Linear behind this LayerNorm in the backward chain
int layernorm_backward_col_partials(int rows);
hipError_t launch_layernorm_backward(const float* x, const float* dy, const float* gamma, const float* dres, float* dx,
                                     float* dgamma_dbeta, float* partial, int rows, hipStream_t s, __bf16* split_out = nullptr,
                                     float* colp = nullptr, unsigned long long drop_seed = 0, unsigned drop_thresh = 0, float drop_scale = 1.f);
// out[c] = sum_r dy[r][c]; partial: workspace [n_chunks, n_cols]; column_sums_chunks() chunks keep every stage short
int column_sums_chunks();
hipError_t launch_column_sums(const float* dy, long ld, int rows, int n_cols, float* out, float* partial, int n_chunks, hipStream_t s);
hipError_t launch_gelu_backward(const float* pre, const float* dh, float* dpre, size_t n, hipStream_t s);

// ---- training path around the GEMMs (train.hip) ---------------------------------------------------------------
// dY fp32 [M, N] -> split rows [M, 2N] (the operand of BOTH gradient GEMMs) and (optional) per-32-row column sums
// col_partial [Mp/32, N] (Mp = M rounded up to 32, >= M) in one pass.  GradXform: the elementwise backward that precedes this Linear, applied on load
// (dY itself is not rewritten): gelu'(pre) or the forward's dropout mask.
enum GradXformMode { XF_NONE = 0, XF_GELU = 1, XF_DROP = 2 };
struct GradXform {
  int mode = XF_NONE;
  const float* pre = nullptr;           // XF_GELU: pre-activation [M, ld]
  unsigned long long seed = 0;          // XF_DROP
  unsigned thresh = 0;
  float scale = 1.f;
};
hipError_t launch_prep_grad(const float* src, long ld, int M, int N, __bf16* rows_out, int Mp, float* col_partial,
                            const GradXform& xf, hipStream_t s);
hipError_t launch_gelu_split(const float* pre, __bf16* dst, size_t rows, int n_cols, hipStream_t s);
// dx row p = dlogits[p] . W; dw [n_out, 576], db [n_out]; row p of x and of dx at + p * ld floats (the CLS rows)
// partial: workspace of head_backward_partial_floats(n_out) floats
size_t head_backward_partial_floats(int n_out);
hipError_t launch_head_backward(const float* dlogits, const float* w, const float* x, float* dx, float* dw, float* db, float* partial,
                                int n_pair, int n_out, long ld, hipStream_t s);
hipError_t launch_assemble_backward(const float* dx, const int32_t* subj, const int32_t* obj, const float* lc, float* dpatch, float* dlc,
                                    int n_pair, hipStream_t s);
hipError_t launch_sgemm_tn(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int n, int ka, int kb, hipStream_t s);
hipError_t launch_sgemm_nt(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int n, int ki, int kj, hipStream_t s);
hipError_t launch_obj_pos_backward(const float* boxes, int box_mode, const float* stats, const float* bn_w, const float* bn_b,
                                   const float* pos_w, const float* pos_b, const float* dpos, float* dpre, float* xhat, float* bn_out,
                                   float* dbn_out, float* dgamma, float* dbeta, int n_obj, hipStream_t s);
hipError_t launch_sgemm_nn(const float* a, long lda, const float* b, long ldb, float* c, long ldc, int n, int ki, int kj, hipStream_t s);
hipError_t launch_softmax_rows(const float* x, float* y, int n, int c, hipStream_t s);
hipError_t launch_bias_relu(float* x, const float* b, int n, int k, hipStream_t s);
hipError_t launch_gather_rows(const float* table, const int64_t* labels, int dim, float* out, int n, hipStream_t s);
hipError_t launch_scatter_rows(const float* demb, const int64_t* labels, int dim, float* dtable, int n, hipStream_t s);
hipError_t launch_untranspose_pair_proj(const float* dwt, float* dw, int kin, hipStream_t s);
hipError_t launch_patch_weight_grad(const float* dwcat_t, float* dwd, float* dwv, hipStream_t s);
// Input gradient of the patch projection (training): the TRANSPOSE of the combined patch weight as a GEMM weight operand,
// dst [kPatchTRows = 2112, 2*1152] split rows (rows 2048.. are zero: 2112 = 11 x 192 output columns), and the scatter of
// dPA [n_obj*16, ld] fp32 (patch rows x patch features, depth features first) back onto the ROI maps [n_obj, 256, 8, 8]
constexpr int kPatchTRows = 2112;
hipError_t launch_build_patch_weight_t(const float* wd, const float* wv, __bf16* dst, hipStream_t s);
hipError_t launch_unpatchify(const float* dpa, long ld, float* d_depth, float* d_rgb, int n_obj, hipStream_t s);

}  // namespace veto
