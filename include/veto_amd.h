/*
 * veto_amd.h -- C ABI of the MI355X-native VETO relation-prediction hot path.
 *
 * The reference has NO FFI on this path: VETOPredictor is pure Python/PyTorch
 * (pysgg/modeling/roi_heads/relation_head/roi_relation_predictors.py:3997-4139 and
 * model_veto.py:6-146).  The boundary a maintainer binds is therefore the predictor's own
 * forward, flattened to raw device pointers:
 *
 *   veto_create            <- VETOPredictor.__init__            roi_relation_predictors.py:3999-4071
 *   veto_load_weights      <- nn.Module.load_state_dict keys    SURVEY.md section 8(b) key list
 *   veto_forward           <- VETOPredictor.forward (eval)      roi_relation_predictors.py:4074-4139
 *                             Ensemble.forward (MEET, eval)     roi_relation_predictors.py:3752-3853
 *   veto_enumerate_pairs   <- RelationSampling.prepare_test_pairs   sampling.py:31-52 (GT-box branch)
 *
 * Conventions: every pointer marked "device" is a HIP device pointer valid on cfg.device;
 * `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on that
 * stream and nothing synchronises it.  Functions return 0 on success, a negative veto_status
 * otherwise; veto_last_error() gives a thread-local message.  No exceptions cross the boundary.
 * A handle is not thread-safe; use one handle per stream/thread.
 */
#ifndef VETO_AMD_H_
#define VETO_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct veto_handle_s* veto_handle_t;

enum veto_status {
  VETO_OK = 0,
  VETO_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
  VETO_ERR_HIP = -2,          /* a HIP runtime call failed */
  VETO_ERR_WEIGHTS = -3,      /* forward called before every weight was loaded */
  VETO_ERR_WORKSPACE = -4     /* workspace too small */
};

enum veto_precision {
  VETO_PRECISE = 0,  /* 3-term split-bf16 MFMA on every Linear, meets the 1e-3 logit tolerance (2-3e-5 measured).  Range: the Linears'
                        operands keep the fp32 exponent range; the ATTENTION products (every mode, the training path included) split
                        q / k / v and the probabilities into fp16 hi + fp16 lo planes (22 significant bits): |q|, |k|, |v| beyond 65504
                        saturate there and components below ~2^-24 are dropped -- both far outside what LayerNorm'ed rows times
                        weights produce (O(10)); no audit counts them */
  VETO_FAST = 1,     /* single-pass form of VETO_MIXED (round 6; rounds 1-5: a single bf16 pass of every GEMM): the same launches
                        on the same operand rows, but the two fused token-row launches of a layer (QKV + attention, layer tail) skip the
                        correction stages -- the fp16 main product alone, neither loading nor multiplying the e4m3 planes.  The measured
                        floor of the schedule with the precision terms free; logit error ~2e-3 (over the 1e-3 bar): reported, never
                        parity-grade.  Inference only. */
  VETO_MIXED = 2     /* fp16 MFMA main product + e4m3 K=128 MFMA correction terms on the token-row Linears: 2/3 of the
                        matrix-pipe time of VETO_PRECISE, logit error 4-9e-5 measured (1e-3 tolerance); what the Python
                        plugin selects by default (VETO_AMD.PRECISION = "mixed").  Supported activation range: the e4m3
                        planes keep full precision for |activation| <= 448 (an element beyond degrades to the fp16
                        class, 2^-11; fp16 overflows at 65504); 6.3e-5 on trained-like activations
                        (tests/test_gpu_parity.py::test_parity_on_trained_like_activations).  In this mode everything
                        of a full layer behind its attention runs as ONE launch (ffn_fused.hip: out projection, both
                        residuals, both LayerNorms, FeedForward), and q / k / v travel from the QKV projection to the
                        attention kernel as 3-byte floats (a 16-bit significand: what the attention's bf16 hi + lo
                        operands keep anyway); logit error 5.7e-5 measured in round 3.  The inference path uses the
                        one-exponential GELU (|error| <= 1e-6) in every mode; the training path keeps the erf form */
};

/* MODEL.ROI_RELATION_HEAD.VETOTRANSFORMER.* (config/defaults.py:331-338) + class counts. */
typedef struct veto_config {
  int32_t struct_size;     /* sizeof(veto_config_t), for ABI evolution */
  int32_t dim;             /* T_INPUT_DIM; must be 576 (model_veto.py:105-113 force it) */
  int32_t layers;          /* ENC_LAYERS */
  int32_t heads;           /* NHEADS; must divide 576 with 576/heads % 4 == 0 */
  int32_t patch;           /* PATCH_SIZE; must be 2 */
  int32_t channels;        /* ROI channels per modality; must be 256 */
  int32_t resolution;      /* POOLER_RESOLUTION; must be 8 */
  int32_t num_obj_cls;     /* 151 (VG) / 201 (GQA); <= 256 */
  int32_t embed_dim;       /* 200 */
  int32_t num_out;         /* head width: num_rel_cls (51/101) or sum_k (g_k + 2) for MEET */
  int32_t precision;       /* enum veto_precision */
  int32_t device;          /* HIP device ordinal */
  int32_t max_chunk_pairs; /* pairs processed per pass (bounds the workspace); 0 = default */
} veto_config_t;

typedef struct veto_inputs {
  int32_t struct_size;        /* sizeof(veto_inputs_t) */
  int32_t n_obj;              /* total objects over the batch */
  int32_t n_pair;             /* total pairs over the batch */
  int32_t n_img;              /* images in the batch */
  const float* roi_rgb;       /* device [n_obj, 256, 8, 8]  roi_features        (relation_head.py:140-141) */
  const float* roi_depth;     /* device [n_obj, 256, 8, 8]  roi_depth_features                              */
  const float* boxes;         /* device [n_obj, 4]  BoxList.bbox */
  int32_t box_mode;           /* 0 = xyxy, 1 = xywh (BoxList.mode) */
  int32_t reserved0;
  const int64_t* obj_labels;  /* device [n_obj]; predcls GT labels, or MEET sgcls argmax labels; or NULL */
  const float* obj_logits;    /* device [n_obj, num_obj_cls]; vanilla sgcls soft embedding; or NULL */
  const int64_t* rel_pairs;   /* device [n_pair, 2] image-local (subj, obj), images concatenated */
  const int32_t* img_obj_offset;   /* device [n_img + 1] exclusive prefix sum of objects per image */
  const int32_t* img_pair_offset;  /* device [n_img + 1] exclusive prefix sum of pairs per image */
  float* bn_batch_stats;      /* NULL: BatchNorm1d(4) of pos_embed in eval mode (running statistics).  Non-NULL device [12]:
                                 TRAINING-mode BatchNorm (roi_relation_predictors.py:4042-4047, module.train()): the four box
                                 features are normalised with the statistics of THIS batch, which are also written here as
                                 mean[4], biased variance[4], unbiased variance[4] for the caller's running-statistics update */
} veto_inputs_t;

/* Optional extra outputs (all may be NULL); used by the parity tests. */
typedef struct veto_debug_outputs {
  int32_t struct_size;
  int32_t reserved0;
  int64_t* subj_inds;   /* device [n_pair] global subject index  (roi_relation_predictors.py:4112) */
  int64_t* obj_inds;    /* device [n_pair] global object index   (:4113) */
  float* tokens;        /* device [n_pair, 19, 576] transformer input (model_veto.py:52-64) */
  float* cls;           /* device [n_pair, 576] final CLS feature (model_veto.py:23) */
} veto_debug_outputs_t;

const char* veto_last_error(void);
const char* veto_version(void);

int veto_create(const veto_config_t* cfg, veto_handle_t* out);
int veto_destroy(veto_handle_t h);

/* Number of weight tensors the handle expects and the i-th expected name/numel. Names are the
 * reference state-dict keys relative to the predictor, e.g.
 * "fusion_transformer.transformer.layers.0.0.fn.to_qkv.weight"; for MEET the K heads are passed
 * row-concatenated as "rel_out.weight"/"rel_out.bias". */
int veto_num_weights(veto_handle_t h);
int veto_weight_info(veto_handle_t h, int index, const char** name, size_t* numel);
/* Copies `numel` fp32 values from `src` (device or host pointer) on `stream`. */
int veto_load_weights(veto_handle_t h, const char* name, const float* src, size_t numel, void* stream);

size_t veto_workspace_bytes(veto_handle_t h, int32_t n_obj, int32_t n_pair);

/* Eval forward.  out_logits: device [n_pair, num_out] fp32. */
int veto_forward(veto_handle_t h, void* stream, const veto_inputs_t* in, void* workspace,
                 size_t workspace_bytes, float* out_logits, const veto_debug_outputs_t* dbg);

/* ---- saturation audit of the VETO_MIXED operands (diagnostic) ---------------------------------
 * VETO_MIXED stores every activation that feeds a token-row Linear as fp16 + two e4m3 planes (value, residual x 2^11).  The
 * conversions SATURATE (MODE.FP16_OVFL is set in every kernel that writes such rows): |a| > 448 clamps both e4m3 planes of that
 * element at +-448 -- it then carries fp16 precision (2^-11) instead of 2^-16 -- and |a| > 65504 clamps its fp16 at +-65504, i.e.
 * an Inf or an overflow becomes a FINITE wrong value where the fp32 reference would propagate Inf (a NaN stays a NaN).  Both
 * happen silently in veto_forward.  This call makes them visible: it runs the same forward in its launch-per-stage form (every
 * mixed-row operand exists in memory; logits equal to veto_forward's up to the rounding order of the fused kernels) and counts,
 * behind every producer, the elements that sit AT the clamp values.  counts[layer * VETO_SAT_SITES + site] (host memory,
 * capacity >= layers * VETO_SAT_SITES entries); the last layer runs its attention on split-bf16 operands (its qkv_in / attn_out
 * sites report zeros) and its FeedForward on the pairs' CLS rows only -- mixed rows all the same, and the classifier's input: its
 * ffn_in / hidden sites count n_pair rows --, layer 0 feeds only the location / class token rows through a mixed QKV projection.  Non-zero value_saturated / resid_saturated: those elements lost
 * the correction terms (harmless in small numbers: the 1e-3 logit tolerance holds with 3 % of the hidden units at x 30 and some at
 * x 300, tests/test_gpu_parity.py::test_parity_on_trained_like_activations).  Non-zero f16_saturated: the result is wrong; use
 * VETO_PRECISE for this checkpoint. */
enum veto_saturation_site {
  VETO_SAT_QKV_IN = 0,    /* LayerNorm1 rows: the QKV projection's operand (model_veto.py:125-132 -> :85) */
  VETO_SAT_ATTN_OUT = 1,  /* attention output: the out projection's operand (:94-96) */
  VETO_SAT_FFN_IN = 2,    /* LayerNorm2 rows: fc1's operand (:137-139) */
  VETO_SAT_HIDDEN = 3,    /* gelu(fc1): fc2's operand (:140-143) */
  VETO_SAT_SITES = 4
};
typedef struct veto_saturation {
  int64_t elements;          /* operand elements scanned at this site (0: the site does not exist in this layer) */
  /* upper bounds: an encoding at the clamp also holds values that merely round to it, and NaN / Inf encodings are counted too */
  int64_t f16_saturated;     /* fp16 values at or beyond +-65504 (incl. Inf / NaN) */
  int64_t value_saturated;   /* e4m3 value-plane bytes at +-448 (incl. the NaN encoding) */
  int64_t resid_saturated;   /* e4m3 residual-plane bytes at +-448 (incl. the NaN encoding) */
} veto_saturation_t;
int veto_forward_saturation(veto_handle_t h, void* stream, const veto_inputs_t* in, void* workspace, size_t workspace_bytes,
                            float* out_logits, veto_saturation_t* counts, int32_t capacity);

/* out: device [max(n*(n-1), 1), 2] int64, row-major (i, j), i != j; [[0,0]] when n <= 1. */
int veto_enumerate_pairs(void* stream, int32_t n, int64_t* out);

/* ---- relation post-processing ---------------------------------------
 * The vanilla GT-box branch of PostProcessor.forward, pysgg/modeling/roi_heads/relation_head/
 * inference.py:398-453: softmax of object and predicate logits, max over the foreground classes,
 * triple score rel*obj_s*obj_o, descending sort per image (ties: lower original index first), and the
 * pair indices / probabilities / labels emitted in that order. */
typedef struct veto_post_args {
  int32_t struct_size;            /* sizeof(veto_post_args_t) */
  int32_t n_img, n_obj, n_pair;
  int32_t n_rel_cls, n_obj_cls;   /* 51 / 151 (VG) */
  int32_t max_pairs_per_image;    /* host-side maximum of the per-image pair counts; must be <= 4096 */
  int32_t reserved0;
  const float* rel_logits;        /* device [n_pair, n_rel_cls] */
  const float* obj_logits;        /* device [n_obj, n_obj_cls]  (refine logits / predict_logits) */
  const int64_t* rel_pairs;       /* device [n_pair, 2] image-local */
  const int32_t* img_obj_offset;  /* device [n_img + 1] */
  const int32_t* img_pair_offset; /* device [n_img + 1] */
  float* obj_scores;              /* out device [n_obj]   -> BoxList field pred_scores */
  int64_t* obj_pred;              /* out device [n_obj]   -> pred_labels */
  float* rel_prob_sorted;         /* out device [n_pair, n_rel_cls] -> pred_rel_scores */
  int64_t* rel_pairs_sorted;      /* out device [n_pair, 2]         -> rel_pair_idxs */
  int64_t* rel_labels_sorted;     /* out device [n_pair]            -> pred_rel_labels */
  float* triple_sorted;           /* optional out device [n_pair] (the sort keys) */
} veto_post_args_t;

size_t veto_postprocess_workspace_bytes(int32_t n_pair, int32_t n_rel_cls);
int veto_postprocess(void* stream, const veto_post_args_t* args, void* workspace, size_t workspace_bytes);

/* MEET merge branch (ENSEMBLE_LEARNING.ENABLED, EXPERT_GROUP False), inference.py:284-397, for ONE image
 * (the reference zips the group logits with the first image only).  Each of the n_groups heads is
 * soft-maxed over its g_k+2 logits, the last column dropped, the arg-max over columns 1..g_k taken as a
 * GROUP-LOCAL label; all n_groups*n_pair rows are merged and sorted by triple score; row probabilities are
 * scattered into n_rel_cls-wide rows at columns [0] + {c : incre_idx_list[c] == k+1}. */
typedef struct veto_post_meet_args {
  int32_t struct_size;
  int32_t n_obj, n_pair, n_groups;
  int32_t n_rel_cls, n_obj_cls;
  const float* const* group_logits;   /* HOST array of n_groups device pointers [n_pair, group_widths[k]] */
  const int32_t* group_widths;        /* HOST array [n_groups]: g_k + 2 */
  const int32_t* incre_idx_list;      /* HOST array [n_rel_cls]: 1-based group of each class, 0 = background */
  const float* obj_logits;            /* device [n_obj, n_obj_cls] */
  const int64_t* rel_pairs;           /* device [n_pair, 2] */
  float* obj_scores;                  /* out device [n_obj] */
  int64_t* obj_pred;                  /* out device [n_obj] */
  float* rel_prob_sorted;             /* out device [n_groups*n_pair, n_rel_cls] */
  int64_t* rel_pairs_sorted;          /* out device [n_groups*n_pair, 2] */
  int64_t* rel_labels_sorted;         /* out device [n_groups*n_pair] (group-local labels, as the reference) */
  float* triple_sorted;               /* optional out device [n_groups*n_pair] */
} veto_post_meet_args_t;

/* workspace: veto_postprocess_workspace_bytes(n_groups * n_pair, n_rel_cls) */
int veto_postprocess_meet(void* stream, const veto_post_meet_args_t* args, void* workspace, size_t workspace_bytes);

/* EXPERT_GROUP voting branch (ENSEMBLE_LEARNING.EXPERT_GROUP True, the defaults.py:864 default),
 * inference.py:93-283, for ONE image: every group has three expert heads ('group_<k>1..3'); a pair's row
 * for group k is kept when two experts ('C', consensus) or all three ('U', unanimous) pick the same
 * class; kept rows carry the averaged score / probabilities of the agreeing experts.  Outputs are sized
 * for n_groups*n_pair rows; the first *kept_count rows (score order) are the result, the rest is padding. */
typedef struct veto_post_vote_args {
  int32_t struct_size;
  int32_t n_obj, n_pair, n_groups;
  int32_t n_rel_cls, n_obj_cls;
  int32_t voting;                     /* 0 = 'C' (two of three agree), 1 = 'U' (all agree): ENSEMBLE_LEARNING.VOTING */
  int32_t reserved0;
  const float* const* expert_logits;  /* HOST array of 3*n_groups device pointers, [3*k + e] = 'group_<k><e+1>' */
  const int32_t* group_widths;        /* HOST array [n_groups]: g_k + 2 */
  const int32_t* incre_idx_list;      /* HOST array [n_rel_cls] */
  const float* obj_logits;            /* device [n_obj, n_obj_cls] */
  const int64_t* rel_pairs;           /* device [n_pair, 2] */
  float* obj_scores;                  /* out device [n_obj] */
  int64_t* obj_pred;                  /* out device [n_obj] */
  float* rel_prob_sorted;             /* out device [n_groups*n_pair, n_rel_cls] */
  int64_t* rel_pairs_sorted;          /* out device [n_groups*n_pair, 2] */
  int64_t* rel_labels_sorted;         /* out device [n_groups*n_pair] (group-local labels) */
  float* triple_sorted;               /* optional out device [n_groups*n_pair]; -1 marks padding rows */
  int32_t* kept_count;                /* out device [1] */
} veto_post_vote_args_t;

/* workspace: veto_postprocess_workspace_bytes(n_groups * n_pair, n_rel_cls) */
int veto_postprocess_vote(void* stream, const veto_post_vote_args_t* args, void* workspace, size_t workspace_bytes);

/* ---- ROI feature extraction -------------------------------------------
 * VETOFeatureExtractor.forward -> Pooler.forward with cat_all_levels=False
 * (pysgg/modeling/roi_heads/box_head/roi_box_feature_extractors.py:75-121, pysgg/modeling/poolers.py:109-171)
 * over the legacy ROIAlign of pysgg/csrc/cuda/ROIAlign_cuda.cu:65-125 (layers/roi_align.py:12-61):
 * every ROI is pooled from ITS FPN level (LevelMapper, poolers.py:17-43) at that level's scale, the depth
 * map with the fixed pooler of level 2 (poolers.py:144-153; level 0 when there is one level).  One launch. */
typedef struct veto_roi_pool_args {
  int32_t struct_size;
  int32_t n_levels;               /* 1..4 */
  int32_t n_img, n_roi;
  int32_t channels;               /* of the pyramid maps (256) */
  int32_t depth_channels;         /* of the depth map (256); ignored when depth_feat is NULL */
  int32_t pooled;                 /* POOLER_RESOLUTION (8); 1..8 */
  int32_t sampling_ratio;         /* POOLER_SAMPLING_RATIO (2); 1..4 (0 = adaptive is not built) */
  const float* level_feat[4];     /* device [n_img, channels, level_h[l], level_w[l]], finest level first */
  int32_t level_h[4];
  int32_t level_w[4];
  float level_scale[4];           /* POOLER_SCALES, e.g. 1/4, 1/8, 1/16, 1/32 */
  const float* depth_feat;        /* device [n_img, depth_channels, depth_h, depth_w] or NULL */
  int32_t depth_h, depth_w;
  const float* rois;              /* device [n_roi, 5]: image index, x1, y1, x2, y2 (Pooler.convert_to_roi_format) */
  float* out_rgb;                 /* out device [n_roi, channels, pooled, pooled]        -> roi_features */
  float* out_depth;               /* out device [n_roi, depth_channels, pooled, pooled]  -> roi_depth_features */
  int32_t* out_levels;            /* optional out device [n_roi]: the level each ROI was pooled from */
} veto_roi_pool_args_t;

int veto_roi_pool(void* stream, const veto_roi_pool_args_t* args);

/* Backward of veto_roi_pool (pysgg/csrc/cuda/ROIAlign_cuda.cu:178-262 RoIAlignBackwardFeature behind
 * layers/roi_align.py:27-44): `args` describes the forward call (shapes, scales, rois; its feature / output
 * pointers are not read), grad_rgb / grad_depth are the gradients of the two pooled tensors, level_grad[l] /
 * depth_grad receive the map gradients (same shapes as the maps; ZERO-INITIALISED BY THE CALLER, accumulated
 * with atomic adds, so the summation order -- not the result up to rounding -- varies from run to run).
 * grad_depth / depth_grad may be NULL. */
int veto_roi_pool_backward(void* stream, const veto_roi_pool_args_t* args, const float* grad_rgb, const float* grad_depth,
                           float* const* level_grad, float* depth_grad);

/* ---- relation evaluators -----------------------------------------------
 * evaluate_relation_of_one_image (pysgg/data/datasets/evaluation/vg/vg_eval.py:459-566) over the evaluator
 * classes of sgg_eval.py for the GT-box modes: SGRecall (:121-187), SGNoGraphConstraintRecall (:195-255),
 * SGZeroShotRecall (:263-313), SGPairAccuracy (:322-369), SGMeanRecall (:377-466), SGNGMeanRecall (:470-546),
 * and their accumulation over the data set, K = 20 / 50 / 100.  Images are concatenated; the *_off arrays
 * are exclusive prefix sums.  For predcls pass the GT classes / boxes as the predicted ones and obj_scores = 1
 * (vg_eval.py:517-520).  Images without GT relations or without predictions contribute nothing (:474, :544). */
typedef struct veto_sgg_eval_args {
  int32_t struct_size;
  int32_t n_img;
  int32_t n_rel_cls;               /* 51 */
  int32_t n_zeroshot;
  float iou_thres;                 /* TEST.RELATION.IOU_THRESHOLD (0.5) */
  int32_t reserved0;
  const int32_t* gt_offset;        /* device [n_img + 1] GT relations */
  const int32_t* obj_offset;       /* device [n_img + 1] objects */
  const int32_t* pair_offset;      /* device [n_img + 1] predicted pairs */
  const int64_t* gt_rels;          /* device [sum G, 3]: subject, object (image-local), predicate  ('relation_tuple') */
  const int64_t* gt_classes;       /* device [sum N]  ('labels') */
  const float* gt_boxes;           /* device [sum N, 4] xyxy */
  const int64_t* pred_pairs;       /* device [sum P, 2] in ranking order  ('rel_pair_idxs') */
  const float* rel_scores;         /* device [sum P, n_rel_cls]           ('pred_rel_scores') */
  const int64_t* pred_classes;     /* device [sum N]  ('pred_labels') */
  const float* pred_boxes;         /* device [sum N, 4] */
  const float* obj_scores;         /* device [sum N]  ('pred_scores') */
  const int64_t* zeroshot;         /* device [n_zeroshot, 3]: subject class, object class, predicate */
  int32_t* gc_rank;                /* out device [sum G]: index of the first matching prediction, 0x3fffffff = none */
  int32_t* ng_rank;                /* out device [sum G]: the same in the no-graph-constraint top-100 list */
  int32_t* acc_rank;               /* out device [sum G]: the same, counted among the predictions on GT pairs */
  int32_t* zeroshot_flag;          /* out device [sum G] */
  int32_t* ng_rows;                /* out device [n_img, 100]: pair index of the i-th no-graph-constraint entry */
  int32_t* ng_cols;                /* out device [n_img, 100]: its predicate */
  int32_t* ng_count;               /* out device [n_img]: entries in that list (min(100, P * (n_rel_cls - 1))) */
  double* metrics;                 /* out device [18 + 6 * (n_rel_cls - 1) + 2]: R@20/50/100, ngR, zR, A, mR, ng-mR,
                                      per-class recall lists [2 kinds][3 K][n_rel_cls - 1], images evaluated,
                                      images with a zero-shot relation */
} veto_sgg_eval_args_t;

size_t veto_sgg_eval_workspace_bytes(int32_t n_img, int32_t n_pair_total, int32_t n_gt_total, int32_t n_rel_cls);
int veto_sgg_eval(void* stream, const veto_sgg_eval_args_t* args, int32_t n_pair_total, int32_t n_gt_total,
                  void* workspace, size_t workspace_bytes);

/* ---- measurement hooks (bench.py): per-kernel device time from hipEvents on `stream` ---------- */
int veto_profile_enable(veto_handle_t h, int32_t on);
/* Synchronises the recorded events; returns the number of distinct kernels. */
int veto_profile_collect(veto_handle_t h);
int veto_profile_entry(veto_handle_t h, int index, const char** name, double* total_ms, int64_t* launches,
                       double* flops_per_launch, double* bytes_per_launch);
int veto_profile_reset(veto_handle_t h);

/* ---- test hook: C[M,N] = A[M,K] . W[N,K]^T (+bias) through the production split-bf16 GEMM ------ */
int veto_debug_gemm(void* stream, const float* a, const float* w, const float* bias, float* c, int32_t m,
                    int32_t n, int32_t k, int32_t precision, void* workspace, size_t workspace_bytes);
size_t veto_debug_gemm_workspace_bytes(int32_t m, int32_t n, int32_t k);
/* ---- test hook: the same GEMM (3-term split-bf16) in its other forms: block-diagonal weights -- column tile j (192 columns) of C
 * multiplies only the k-steps (32 k's each) [(j / kb_tiles) * kb_steps, + kb_steps) of the rows, the rest of w is ignored;
 * kb_tiles = 0: dense -- and the output forms out_form 0 = fp32 [m, n], 1 = split rows (m x 2n bf16: per 32 columns 32 hi, then
 * 32 lo), 2 = 3-byte floats (m x 3n bytes: the top three bytes of the fp32 rounded to nearest even).  The folded last layer of the
 * predictor is built from these (roi_relation_predictors.py:4118-4131 -> model_veto.py:85-96 for the CLS query). */
int veto_debug_gemm_forms(void* stream, const float* a, const float* w, void* c, int32_t m, int32_t n, int32_t k,
                          int32_t kb_tiles, int32_t kb_steps, int32_t out_form, void* workspace, size_t workspace_bytes);

/* ---- test / measurement hook: the FeedForward block of one layer, x <- x + W2 . gelu(W1 . a + b1) + b2 (model_veto.py:137-143
 * with the residual of :21) on VETO_MIXED operands; mode 0 = two GEMM launches with the hidden activation in HBM, mode 1 = the
 * fused kernel (hidden activation stays on the CU).  a [m, 576] (the LayerNorm'ed rows), w1 [1152, 576], w2 [576, 1152],
 * x [m, 576] in / out.  flags & 1: rebuild the mixed operands in the workspace first.  Runs `reps` times; *ms_per_rep (host,
 * optional) = mean device time of one run.  ln_rows (optional, m x 2304 bytes): LayerNorm(x_out; ln_w, ln_b) as mixed activation
 * rows -- the next layer's PreNorm -- from the fused kernel's epilogue (mode 1) or a LayerNorm launch (mode 0). */
int veto_debug_ffn(void* stream, const float* a, const float* w1, const float* b1, const float* w2, const float* b2,
                   float* x, int32_t m, int32_t mode, int32_t flags, int32_t reps, float* ms_per_rep, void* workspace,
                   size_t workspace_bytes, const float* ln_w, const float* ln_b, void* ln_rows);
size_t veto_debug_ffn_workspace_bytes(int32_t m);

/* ---- test / measurement hook: the attention out projection + residual, x <- x + a W^T + b (model_veto.py:96 `to_out`, :20) on
 * VETO_MIXED operands, optionally followed by LayerNorm rows (ln_rows, m x 2304 bytes of mixed activation rows: the FeedForward
 * PreNorm).  mode 0 = the GEMM launch (+ a LayerNorm launch), mode 1 = the full-row panel kernel (one launch).  a [m, 576],
 * w [576, 576], x [m, 576] in / out. */
int veto_debug_outproj(void* stream, const float* a, const float* w, const float* b, float* x, int32_t m, int32_t mode,
                       int32_t flags, int32_t reps, float* ms_per_rep, void* workspace, size_t workspace_bytes,
                       const float* ln_w, const float* ln_b, void* ln_rows);
size_t veto_debug_outproj_workspace_bytes(int32_t m);

/* ---- test / measurement hook: everything of one layer behind its attention (model_veto.py:96, :20-21, :125-143) on VETO_MIXED
 * operands: x1 = x + a Wo^T + bo, h = LayerNorm(x1; ln2_w, ln2_b), x = x1 + W2 gelu(W1 h + b1) + b2, and optionally ln_rows =
 * LayerNorm(x; ln_w, ln_b) as mixed activation rows (m x 2304 bytes).  mode 0 = the out-projection panel launch + the FeedForward
 * panel launch, mode 1 = one launch.  a [m, 576] (attention output), wo [576, 576], w1 [1152, 576], w2 [576, 1152], x in / out. */
int veto_debug_layer_tail(void* stream, const float* a, const float* wo, const float* bo, const float* ln2_w, const float* ln2_b,
                          const float* w1, const float* b1, const float* w2, const float* b2, float* x, int32_t m, int32_t mode,
                          int32_t reps, float* ms_per_rep, void* workspace, size_t workspace_bytes, const float* ln_w,
                          const float* ln_b, void* ln_rows);
size_t veto_debug_layer_tail_workspace_bytes(int32_t m);

/* ---- test / measurement hook: QKV projection + per-pair attention of a middle layer (model_veto.py:78-96) on VETO_MIXED operands:
 * a [19 n_pair, 576] = LayerNorm1 rows, wqkv [1728, 576] (no bias); out_rows receives the merged-heads attention output as mixed
 * activation rows (19 n_pair x 2304 bytes: the operand of the out projection).  mode 1 = ONE launch (qkv_attn_fused.hip: q / k / v never
 * reach memory), mode 0 = the two launches it replaces (QKV GEMM writing 3-byte q / k / v + the attention launch).  heads 8 or 6. */
int veto_debug_qkv_attn(void* stream, const float* a, const float* wqkv, int32_t n_pair, int32_t heads, int32_t mode, int32_t reps,
                        float* ms_per_rep, void* workspace, size_t workspace_bytes, void* out_rows);
size_t veto_debug_qkv_attn_workspace_bytes(int32_t n_pair);

/* ---- training losses and MEET expert sampling (partial) -------------------------
 * veto_ce_loss: nn.CrossEntropyLoss(weight)(logits[rows], labels), mean reduction -- the relation loss of
 * VETOPredictor.forward (roi_relation_predictors.py:4133, BETA_LOSS weights :4057-4068) and, on a row subset with
 * group-local labels, the per-group losses of Ensemble.forward (:3842-3846).  Writes the loss (device float) and,
 * if grad is non-NULL, d loss / d logits for the selected rows [n, n_cls]. */
size_t veto_ce_loss_workspace_bytes(int32_t n);
int veto_ce_loss(void* stream, const float* logits, int64_t ld, const int64_t* labels, const float* weight,
                 const int64_t* rows, int32_t n, int32_t n_cls, float* loss, float* grad, void* workspace,
                 size_t workspace_bytes);

/* veto_meet_sample: the expert sampling loop of VETOPredictor_MEET.forward in training (:3940-3969,
 * ZERO_LABEL_PADDING_MODE 'rand_insert') followed by the per-group label remap of Ensemble.forward (:3812-3821).
 * `words` are the next raw 32-bit outputs of Python's `random` generator (MT19937), consumed exactly as
 * random.randint / random.random would; *words_used tells the host how far to advance its generator (-1: block too
 * short).  sample_rates is the [n_groups, n_cls] matrix of extra_function_utils.py:185-257 in double precision. */
int veto_meet_sample(void* stream, const int64_t* labels, int32_t n, const uint32_t* words, int32_t n_words,
                     const int32_t* incre_idx_list, const int32_t* pos_in_group, const int32_t* group_size,
                     const double* sample_rates, int32_t n_groups, int32_t n_cls, int64_t* chosen,
                     int64_t* group_labels, int32_t* counts, int32_t* words_used);

/* ---- training path: forward that keeps the activations + backward --------------
 * The backward of VETOPredictor.forward's computation graph (roi_relation_predictors.py:4074-4133, model_veto.py)
 * w.r.t. every parameter, for hard object labels (predcls, MEET), precise mode.  veto_forward_train runs all pairs in one pass, every layer on all 19 tokens, BatchNorm on batch
 * statistics (in->bn_batch_stats is mandatory), and leaves the activations in `workspace`
 * (veto_train_workspace_bytes, ~28 KB per token row and layer); veto_backward takes d loss / d logits
 * [n_pair, num_out] and writes d loss / d parameter for every state-dict tensor into `grads`, a flat float buffer of
 * veto_grad_floats(h) elements in which tensor i starts at veto_weight_offset(h, i) (buffers such as running
 * statistics get zeros).  Both calls must see the same inputs and workspace. */
/* Dropout of the training path (NULL = none): pos_embed's Dropout(0.1) (roi_relation_predictors.py:4042-4047), the
 * transformer's pos_drop (EMB_DROPOUT, model_veto.py:44,63) and the Dropout behind every attention out projection
 * (T_DROPOUT, model_veto.py:80-83).  Masks come from a counter-based hash of (seed, site, element index), recomputed in
 * the backward: the same opts must be given to both calls.  The masks are this library's own (the reference draws from
 * torch's generator), i.e. equal in distribution, not bit for bit. */
typedef struct veto_train_opts {
  int32_t struct_size;
  float p_pos, p_emb, p_attn;
  uint64_t seed;
  /* veto_backward only, optional (NULL = not wanted): gradients of the loss w.r.t. the ROI maps, device [n_obj, 256, 8, 8]
   * fp32 each.  The reference trains its depth backbone through roi_depth_features (tools/relation_train_net.py:166-170). */
  float* d_roi_rgb;
  float* d_roi_depth;
} veto_train_opts_t;

size_t veto_train_workspace_bytes(veto_handle_t h, int32_t n_obj, int32_t n_pair);
size_t veto_grad_floats(veto_handle_t h);
int veto_weight_offset(veto_handle_t h, int index, size_t* offset_floats);
int veto_forward_train(veto_handle_t h, void* stream, const veto_inputs_t* in, const veto_train_opts_t* opts, void* workspace,
                       size_t workspace_bytes, float* out_logits);
int veto_backward(veto_handle_t h, void* stream, const veto_inputs_t* in, const veto_train_opts_t* opts, void* workspace,
                  size_t workspace_bytes, const float* dlogits, float* grads);

/* ---- test hook: dw[N,K] = dy[M,N]^T . x[M,K], the weight-gradient GEMM (reduction over the M rows) through the
 * production split-bf16 kernel in its split-K / atomic-add form.  k must be a multiple of 192; k_splits 0 = auto. */
int veto_debug_wgrad(void* stream, const float* dy, const float* x, float* dw, int32_t m, int32_t n, int32_t k,
                     int32_t k_splits, void* workspace, size_t workspace_bytes);
size_t veto_debug_wgrad_workspace_bytes(int32_t m, int32_t n, int32_t k, int32_t k_splits);

/* ---- test hooks: backward building blocks of the transformer (chained by veto_backward) ----------------------
 * veto_debug_attention_backward: qkv, dqkv device [n_pair*19, 1728], dout device [n_pair*19, 576]   (model_veto.py:85-96)
 * veto_debug_layernorm_backward: x, dy, dx device [rows, 576] (dres optional, added to dx), gamma [576],
 *                                dgamma_dbeta device [2, 576]; workspace of veto_debug_layernorm_backward_workspace_bytes
 * veto_debug_gelu_backward:      dpre = dh * gelu'(pre), n elements (n % 4 == 0)                       (model_veto.py:140)
 * veto_debug_column_sums:        out[c] = sum_r dy[r, c] (bias gradients); workspace 256 * n_cols floats */
int veto_debug_attention_backward(void* stream, const float* qkv, const float* dout, float* dqkv, int32_t n_pair, int32_t heads);
size_t veto_debug_layernorm_backward_workspace_bytes(int32_t rows);
int veto_debug_layernorm_backward(void* stream, const float* x, const float* dy, const float* gamma, const float* dres,
                                  float* dx, float* dgamma_dbeta, int32_t rows, void* workspace, size_t workspace_bytes);
int veto_debug_gelu_backward(void* stream, const float* pre, const float* dh, float* dpre, size_t n);
int veto_debug_column_sums(void* stream, const float* dy, int64_t ld, int32_t rows, int32_t n_cols, float* out,
                           void* workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* VETO_AMD_H_ */
