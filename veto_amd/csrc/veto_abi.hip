// Host side of the C ABI (include/veto_amd.h): weight store, workspace carving and the launch
// sequence of one eval forward of VETOPredictor / the MEET Ensemble trunk
// (roi_relation_predictors.py:4074-4139, :3752-3853; model_veto.py:15-26).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "../../include/veto_amd.h"
#include "kernels.h"

using namespace veto;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess) return fail(VETO_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

struct Param {
  std::string name;
  size_t numel = 0;
  size_t offset = 0;  // floats into the raw arena
  bool loaded = false;
};

typedef __bf16* SplitW;  // [N, 2K] split rows (common.h)

struct LayerW {
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *out_b, *fc1_b, *fc2_b;
  SplitW qkv = nullptr, out = nullptr, fc1 = nullptr, fc2 = nullptr;
  // VETO_MIXED: the same four weights as mixed rows (common.h) and their e4m3 exponents (device ints: qkv, out, fc1, fc2)
  SplitW qkv_m = nullptr, out_m = nullptr, fc1_m = nullptr, fc2_m = nullptr;
  int* exp_m = nullptr;
};

struct ProfRec {
  int name_id;
  hipEvent_t start, stop;
  double flops, bytes;
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

namespace veto {
bool env_knob_is(const char* name, const char* value) {
  const char* v = getenv(name);
  return v && !strcmp(v, value);
}
int device_cu_count() {
  static std::atomic<int> cache[64];      // per device ordinal; 0 = not queried yet (a benign race: every thread stores the same value)
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return -1;
  if (dev < 64 && (cus = cache[dev].load(std::memory_order_relaxed)) > 0) return cus;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return -1;
  if (dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}
int env_knob_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
}  // namespace veto

// Padded head width of the block form of the folded last layer, or 0 when the products are used instead: the head width rounded up
// to 32 k's must divide a 192-column GEMM tile (heads 12 / 8 / 6 / 3 of the 576 columns), and the form can be switched off
// (VETO_FOLD_BLOCKS=0, A/B knob).
static int fold_block_width(int heads) {
  static const bool off = env_knob_is("VETO_FOLD_BLOCKS", "0");
  if (off || heads <= 0 || kDim % heads != 0) return 0;
  const int dhp = (kDim / heads + 31) / 32 * 32;
  return 192 % dhp == 0 && (heads * dhp) % 192 == 0 ? dhp : 0;
}

struct veto_handle_s {
  veto_config_t cfg;
  int dh = 0;
  int chunk = 0;
  std::vector<Param> params;
  std::map<std::string, int> index;
  float* raw = nullptr;      // fp32 copies of every state-dict tensor
  char* derived = nullptr;   // split planes, transposes, folded tables
  bool dirty = true;         // a weight was uploaded since the derived operands were built
  bool infer_dirty = false;  // ... the training-side operands are current, the inference-only ones (mixed rows, layer-0 tables, folded last layer) are not
  // generation of the derived weight operands (bumped by every finalize_weights) and, per training workspace, the generation its
  // saved activations were computed with: veto_backward refuses a workspace whose forward saw other weights
  uint64_t weight_gen = 0;
  // (a handful of workspaces at most: a training loop re-uses one; the table is cleared by every weight upload and bounded besides)
  static constexpr size_t kMaxTrainWorkspaces = 8;
  std::map<const void*, uint64_t> train_gen;
  void stamp_train_workspace(const void* ws) {
    if (train_gen.size() >= kMaxTrainWorkspaces && !train_gen.count(ws)) {   // evict the entry of the oldest generation
      auto old = train_gen.begin();
      for (auto it = train_gen.begin(); it != train_gen.end(); ++it)
        if (it->second < old->second) old = it;
      train_gen.erase(old);
    }
    train_gen[ws] = weight_gen;
  }
  std::vector<LayerW> layers;
  SplitW patch_w = nullptr;
  // last layer, folded CLS attention (attention.hip): Mcat [heads*576, 2*576], Ncat [576, 2*heads*576], and their fp32 staging
  SplitW fold_m = nullptr, fold_n = nullptr;
  float* fold_tmp = nullptr;
  // ... in block form (fold_dhp > 0: the head width padded to 32 k's divides a 192-column tile): Mcat = Wq^T Wk and Ncat = Wo Wv
  // are products of rank dh per head, so u = (a0 Wq_pad^T) . blockdiag(Wk) and out = (abar . blockdiag(Wv)^T) Wo_pad^T take four
  // GEMMs of 67 GF in all (zero blocks skipped: GemmArgs::kb_tiles) instead of two of 80 GF each
  SplitW fold_q = nullptr, fold_k = nullptr, fold_v = nullptr, fold_o = nullptr;
  int fold_dhp = 0;
  // layer 0, per-object form of LayerNorm + QKV (rowops.hip): Wqkv diag(gamma) as a GEMM operand, vec = [c2 | b0 | qkv_cls]
  SplitW q0_w = nullptr;
  float* q0_vec = nullptr;
  float* patch_bias = nullptr;
  float* loc_wt = nullptr;
  float* cls_wt = nullptr;
  float* head_wt = nullptr;
  // veto_forward_saturation: device counters [layers][VETO_SAT_SITES][4], allocated by veto_create (VETO_MIXED handles).  The call
  // hands them to forward_impl as an argument -- no handle state changes, so a concurrent veto_forward on the same handle is unaffected --
  // and forward_impl then takes the launch-per-stage form of the mixed path (every mixed-row operand exists in memory) and counts
  // behind every producer
  unsigned long long* sat_buf = nullptr;
  // profiling
  bool prof_on = false;
  std::vector<std::string> prof_names;
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> event_pool;
  struct Agg { double ms = 0; int64_t n = 0; double flops = 0, bytes = 0; };
  std::vector<Agg> prof_agg;

  const float* p(const std::string& name) const { return raw + params[index.at(name)].offset; }
  void add(const std::string& name, size_t numel) {
    Param q;
    q.name = name;
    q.numel = numel;
    index[name] = (int)params.size();
    params.push_back(q);
  }
};

namespace {

const char* kT = "fusion_transformer.transformer.";

std::string lname(int l, const char* rest) {
  char buf[160];
  snprintf(buf, sizeof(buf), "%slayers.%d.%s", kT, l, rest);
  return buf;
}

int prof_id(veto_handle_t h, const char* name) {
  for (size_t i = 0; i < h->prof_names.size(); ++i)
    if (h->prof_names[i] == name) return (int)i;
  h->prof_names.push_back(name);
  h->prof_agg.emplace_back();
  return (int)h->prof_names.size() - 1;
}

struct ProfScope {
  veto_handle_t h;
  hipStream_t s;
  int rec = -1;
  ProfScope(veto_handle_t h_, hipStream_t s_, const char* name, double flops, double bytes) : h(h_), s(s_) {
    if (!h || !h->prof_on) return;
    ProfRec r;
    r.name_id = prof_id(h, name);
    r.flops = flops;
    r.bytes = bytes;
    for (hipEvent_t* e : {&r.start, &r.stop}) {
      if (!h->event_pool.empty()) { *e = h->event_pool.back(); h->event_pool.pop_back(); }
      else if (hipEventCreate(e) != hipSuccess) return;
    }
    (void)hipEventRecord(r.start, s);      // profiling is best effort: a failed record shows up as a missing timing
    h->prof_recs.push_back(r);
    rec = (int)h->prof_recs.size() - 1;
  }
  ~ProfScope() {
    if (rec >= 0) (void)hipEventRecord(h->prof_recs[rec].stop, s);
  }
};

// train_only: just the operands the training path reads (split rows of the Linears, the patch / pair-projection / head re-layouts).  A
// training loop uploads every weight after every optimizer step; the mixed rows (a max-|w| reduction per tensor), the table form of
// layer 0 and the folded last layer are the inference path's, and are built when an inference forward next needs them (infer_dirty).
int finalize_weights(veto_handle_t h, hipStream_t s, bool train_only = false) {
  for (const Param& q : h->params)
    if (!q.loaded) return fail(VETO_ERR_WEIGHTS, "weight '%s' was never loaded", q.name.c_str());
  const int L = h->cfg.layers;
  const bool base = h->dirty;      // (false: only the inference-side operands are missing)
  for (int l = 0; l < L; ++l) {
    LayerW& w = h->layers[l];
    if (base) {
    HIP_TRY(launch_split_rows(h->p(lname(l, "0.fn.to_qkv.weight")), w.qkv, 3 * kDim, kDim, s));
    HIP_TRY(launch_split_rows(h->p(lname(l, "0.fn.to_out.0.weight")), w.out, kDim, kDim, s));
    HIP_TRY(launch_split_rows(h->p(lname(l, "1.fn.net.0.weight")), w.fc1, 2 * kDim, kDim, s));
    HIP_TRY(launch_split_rows(h->p(lname(l, "1.fn.net.3.weight")), w.fc2, kDim, 2 * kDim, s));
    }
    if (h->cfg.precision != VETO_PRECISE && !train_only) {      // (VETO_FAST runs VETO_MIXED's launches, with the correction stages of the fused ones skipped)
      HIP_TRY(launch_mixed_weight_rows(h->p(lname(l, "0.fn.to_qkv.weight")), w.qkv_m, 3 * kDim, kDim, w.exp_m + 0, s));
      HIP_TRY(launch_mixed_weight_rows(h->p(lname(l, "0.fn.to_out.0.weight")), w.out_m, kDim, kDim, w.exp_m + 1, s));
      HIP_TRY(launch_mixed_weight_rows(h->p(lname(l, "1.fn.net.0.weight")), w.fc1_m, 2 * kDim, kDim, w.exp_m + 2, s));
      HIP_TRY(launch_mixed_weight_rows(h->p(lname(l, "1.fn.net.3.weight")), w.fc2_m, kDim, 2 * kDim, w.exp_m + 3, s));
    }
  }
  if (base) {
  const std::string pe = std::string(kT) + "patch_embed.";
  HIP_TRY(launch_build_patch_weight(h->p(pe + "proj_d.weight"), h->p(pe + "proj_d.bias"), h->p(pe + "proj_v.weight"),
                                    h->p(pe + "proj_v.bias"), h->patch_w, h->patch_bias, s));
  HIP_TRY(launch_transpose_pair_proj(h->p("location_projection.0.weight"), h->loc_wt, kPosDim, s));
  HIP_TRY(launch_transpose_pair_proj(h->p("class_projection.0.weight"), h->cls_wt, h->cfg.embed_dim, s));
  HIP_TRY(launch_transpose_head(h->p("rel_out.weight"), h->head_wt, h->cfg.num_out, s));
  }
  if (train_only) {
    h->dirty = false;
    h->infer_dirty = true;
    ++h->weight_gen;
    return VETO_OK;
  }
  {   // layer 0: Wqkv diag(gamma) and the weight-only vectors of the per-object form (fold_tmp holds >= 1728 x 576 floats)
    const std::string T0 = kT;
    HIP_TRY(launch_qkv0_consts(h->p(lname(0, "0.fn.to_qkv.weight")), h->layers[0].ln1_w, h->layers[0].ln1_b, h->p(T0 + "pos_embedding"),
                               h->p(T0 + "cls_token"), h->fold_tmp, h->q0_vec, s));
    HIP_TRY(launch_split_rows(h->fold_tmp, h->q0_w, 3 * kDim, kDim, s));
  }
  if (h->cfg.heads <= cls_fold_max_heads()) {
    // last layer: M_h = W_q,h^T W_k,h and N_h = W_o,h W_v,h (products over the head width, fp32), as GEMM weight operands
    const int H = h->cfg.heads, dh = kDim / H;
    const float* qkv = h->p(lname(L - 1, "0.fn.to_qkv.weight"));      // [1728, 576]: q rows, k rows, v rows
    const float* wo = h->p(lname(L - 1, "0.fn.to_out.0.weight"));     // [576, 576]
    if (h->fold_dhp > 0) {   // block form: the four factors themselves, padded / block-diagonal, as split rows
      const int np = H * h->fold_dhp;
      HIP_TRY(launch_fold_blocks(qkv, wo, h->fold_tmp, 0, H, h->fold_dhp, s));
      HIP_TRY(launch_split_rows(h->fold_tmp, h->fold_q, (size_t)np, kDim, s));
      HIP_TRY(launch_fold_blocks(qkv, wo, h->fold_tmp, 1, H, h->fold_dhp, s));
      HIP_TRY(launch_split_rows(h->fold_tmp, h->fold_k, (size_t)H * kDim, np, s));
      HIP_TRY(launch_fold_blocks(qkv, wo, h->fold_tmp, 2, H, h->fold_dhp, s));
      HIP_TRY(launch_split_rows(h->fold_tmp, h->fold_v, (size_t)np, H * kDim, s));
      HIP_TRY(launch_fold_blocks(qkv, wo, h->fold_tmp, 3, H, h->fold_dhp, s));
      HIP_TRY(launch_split_rows(h->fold_tmp, h->fold_o, (size_t)kDim, np, s));
    } else {
    for (int hd = 0; hd < H; ++hd)   // Mcat row (hd, c), column c' = sum_d Wk[hd*dh + d, c] Wq[hd*dh + d, c']
      HIP_TRY(launch_sgemm_tn(qkv + ((size_t)kDim + hd * dh) * kDim, kDim, qkv + (size_t)hd * dh * kDim, kDim,
                              h->fold_tmp + (size_t)hd * kDim * kDim, kDim, dh, kDim, kDim, s));
    HIP_TRY(launch_split_rows(h->fold_tmp, h->fold_m, (size_t)H * kDim, kDim, s));
    for (int hd = 0; hd < H; ++hd)   // Ncat row r, column (hd, c) = sum_d Wo[r, hd*dh + d] Wv[hd*dh + d, c]
      HIP_TRY(launch_sgemm_nn(wo + hd * dh, kDim, qkv + ((size_t)2 * kDim + hd * dh) * kDim, kDim, h->fold_tmp + (size_t)hd * kDim,
                              (long)H * kDim, kDim, kDim, dh, s));
    HIP_TRY(launch_split_rows(h->fold_tmp, h->fold_n, (size_t)kDim, H * kDim, s));
    }
  }
  if (base) ++h->weight_gen;      // (completing a training-side upload changes no operand a training workspace was computed with)
  h->dirty = false;
  h->infer_dirty = false;
  return VETO_OK;
}

struct Workspace {
  int32_t *subj, *obj;
  float* lc;
  __bf16* pa;       // patch rows, split [prow, 2*2048]
  float* patch_tab;
  float* x;
  __bf16* a;        // LN(x) / attention output, split [mpad, 2*576]
  char* big;        // qkv fp32 [mpad,1728]; later the MLP hidden, split [mpad, 2*1152]
  float* xc;
  __bf16* ptab_split;   // layer 0, per-object form: patch_tab as split rows [prow, 2*1152]
  float *sw, *ow;       // ... its products with Wqkv diag(gamma), fp32 [prow, 1728] each
  float* stats;         // ... (mean, rstd) of the layer-0 token rows [mpad, 2]
  __bf16 *ac, *hc;  // CLS-compact operands, split [cpad, 2*576] / [cpad, 2*1152]
  size_t total;
};

// Carves (or, with base == nullptr, just sizes) the workspace.
Workspace carve(char* base, int n_obj, int n_pair, int chunk) {
  Workspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return ptr;
  };
  const size_t prow = (size_t)gemm_rows_padded(n_obj * 16);
  // (the fused QKV + attention launch reads whole tiles of 16 pairs: the activation rows are padded to those too)
  const size_t tile_rows = qkv_attn_rows_padded(chunk);
  const size_t mpad = (size_t)gemm_rows_padded((int)(tile_rows > (size_t)chunk * kTokens ? tile_rows : (size_t)chunk * kTokens));
  const size_t cpad = (size_t)gemm_rows_padded(chunk);
  w.subj = (int32_t*)take((size_t)n_pair * 4);
  w.obj = (int32_t*)take((size_t)n_pair * 4);
  w.lc = (float*)take((size_t)n_obj * 2 * 2 * kDim * 4);
  w.pa = (__bf16*)take(prow * 2 * 2048 * 2);
  w.patch_tab = (float*)take((size_t)n_obj * 16 * 2 * kDim * 4);
  w.x = (float*)take(mpad * kDim * 4);
  w.a = (__bf16*)take(mpad * 2 * kDim * 2);
  {   // qkv of a chunk; in the last layer instead u [cpad, H*576] fp32 + abar [cpad, 2*H*576] split (folded CLS attention)
    const size_t qkv_bytes = mpad * 3 * kDim * 4, fold_bytes = 2 * (align_up(cpad * (size_t)cls_fold_max_heads() * kDim * 4, 256));
    w.big = take(qkv_bytes > fold_bytes ? qkv_bytes : fold_bytes);
  }
  w.xc = (float*)take(cpad * kDim * 4);
  w.ptab_split = (__bf16*)take(prow * 2 * 2 * kDim * 2);
  w.sw = (float*)take(prow * 3 * kDim * 4);
  w.ow = (float*)take(prow * 3 * kDim * 4);
  w.stats = (float*)take(mpad * 2 * 4);
  w.ac = (__bf16*)take(cpad * 2 * kDim * 2);
  w.hc = (__bf16*)take(cpad * 4 * kDim * 2);
  w.total = off;
  return w;
}

// lda / ldc of split operands are in bf16 elements (2K / 2N for contiguous rows).
struct DropSite {   // one dropout site of the training path: threshold p * 2^24 (0 = off), scale 1 / (1 - p)
  unsigned long long seed = 0;
  unsigned thresh = 0;
  float scale = 1.f;
};

int run_gemm(veto_handle_t h, hipStream_t s, const char* name, const __bf16* a, SplitW w, const float* bias,
             const float* resid, long ldr, float* c, __bf16* c_split, long ldc, int M, int N, int K, int epi,
             long lda = 0, int w_row0 = 0, DropSite drop = DropSite(), const int* w_exp = nullptr, int kb_tiles = 0, int kb_steps = 0) {
  GemmArgs g{};
  g.kb_tiles = kb_tiles; g.kb_steps = kb_steps;   // block-diagonal weights (kernels.h): flops / bytes below count the blocks only
  if (w_exp) { g.fmt = FMT_MIXED; g.w_exp = w_exp; }   // a, w (and an EPI_GELU_SPLIT output) are mixed rows
  if (drop.thresh) {
    if (epi != EPI_RESID) return fail(VETO_ERR_INVALID, "dropout is fused into the residual epilogue only");
    epi = EPI_RESID_DROP;
    g.drop_seed = drop.seed; g.drop_thresh = drop.thresh; g.drop_scale = drop.scale;
  }
  g.a = a; g.lda = lda;
  g.w = w + (size_t)w_row0 * 2 * K;
  g.bias = bias; g.resid = resid; g.c = c; g.c_split = c_split;
  g.M = M; g.N = N; g.K = K; g.ldr = ldr; g.ldc = ldc;
  if (epi == EPI_PRE_GELU) g.ldc_f32 = N;      // (c = the fp32 pre-activation rows, contiguous; ldc is the split rows')
  const double kk = kb_tiles > 0 ? 32.0 * kb_steps : (double)K;
  const double flops = 2.0 * M * (double)N * kk;
  const double bytes = 4.0 * ((double)M * K + (double)N * kk) + (double)M * N * (epi == EPI_RESID ? 8.0 : epi == EPI_F24 ? 3.0 : 4.0);
  ProfScope ps(h, s, name, flops, bytes);
  HIP_TRY(launch_gemm_split(g, epi, 0, s));      // (the per-object and CLS-row GEMMs of a VETO_FAST handle are VETO_MIXED's)
  return VETO_OK;
}

}  // namespace

extern "C" {

const char* veto_last_error(void) { return g_err.c_str(); }
const char* veto_version(void) { return "veto_amd 0.1 (gfx950)"; }

int veto_create(const veto_config_t* cfg, veto_handle_t* out) {
  if (!cfg || !out) return fail(VETO_ERR_INVALID, "null argument");
  if (cfg->struct_size != (int32_t)sizeof(veto_config_t)) return fail(VETO_ERR_INVALID, "veto_config_t size mismatch");
  if (cfg->dim != kDim) return fail(VETO_ERR_INVALID, "T_INPUT_DIM must be 576 (proj_d 512 + proj_v 64), got %d", cfg->dim);
  if (cfg->patch != 2 || cfg->channels != 256 || cfg->resolution != 8)
    return fail(VETO_ERR_INVALID, "only PATCH_SIZE 2, 256 channels, POOLER_RESOLUTION 8 are supported");
  if (cfg->layers < 1 || cfg->layers > 64) return fail(VETO_ERR_INVALID, "bad ENC_LAYERS %d", cfg->layers);
  if (cfg->heads < 1 || kDim % cfg->heads != 0 || (kDim / cfg->heads) % 4 != 0)
    return fail(VETO_ERR_INVALID, "NHEADS %d must divide 576 with head dim %% 4 == 0", cfg->heads);
  if (cfg->num_obj_cls < 2 || cfg->num_obj_cls > 256 || cfg->embed_dim < 1 || cfg->embed_dim > 256)
    return fail(VETO_ERR_INVALID, "num_obj_cls/embed_dim out of range");
  if (cfg->num_out < 1 || cfg->num_out > 4096) return fail(VETO_ERR_INVALID, "bad num_out %d", cfg->num_out);
  if (cfg->precision != VETO_PRECISE && cfg->precision != VETO_FAST && cfg->precision != VETO_MIXED) return fail(VETO_ERR_INVALID, "bad precision");
  HIP_TRY(hipSetDevice(cfg->device));

  veto_handle_t h = new veto_handle_s();
  h->cfg = *cfg;
  h->dh = kDim / cfg->heads;
  h->chunk = cfg->max_chunk_pairs > 0 ? cfg->max_chunk_pairs : 32768;
  const int E = cfg->embed_dim, L = cfg->layers;
  h->add("obj_embed.weight", (size_t)cfg->num_obj_cls * E);
  h->add("class_projection.0.weight", (size_t)kDim * 2 * E);
  h->add("class_projection.0.bias", kDim);
  h->add("pos_embed.0.weight", 4);
  h->add("pos_embed.0.bias", 4);
  h->add("pos_embed.0.running_mean", 4);
  h->add("pos_embed.0.running_var", 4);
  h->add("pos_embed.1.weight", (size_t)kPosDim * 4);
  h->add("pos_embed.1.bias", kPosDim);
  h->add("location_projection.0.weight", (size_t)kDim * 2 * kPosDim);
  h->add("location_projection.0.bias", kDim);
  h->add(std::string(kT) + "cls_token", kDim);
  h->add(std::string(kT) + "pos_embedding", kDim);
  h->add(std::string(kT) + "patch_embed.proj_d.weight", (size_t)512 * 2048);
  h->add(std::string(kT) + "patch_embed.proj_d.bias", 512);
  h->add(std::string(kT) + "patch_embed.proj_v.weight", (size_t)64 * 2048);
  h->add(std::string(kT) + "patch_embed.proj_v.bias", 64);
  for (int l = 0; l < L; ++l) {
    h->add(lname(l, "0.norm.weight"), kDim);
    h->add(lname(l, "0.norm.bias"), kDim);
    h->add(lname(l, "0.fn.to_qkv.weight"), (size_t)3 * kDim * kDim);
    h->add(lname(l, "0.fn.to_out.0.weight"), (size_t)kDim * kDim);
    h->add(lname(l, "0.fn.to_out.0.bias"), kDim);
    h->add(lname(l, "1.norm.weight"), kDim);
    h->add(lname(l, "1.norm.bias"), kDim);
    h->add(lname(l, "1.fn.net.0.weight"), (size_t)2 * kDim * kDim);
    h->add(lname(l, "1.fn.net.0.bias"), 2 * kDim);
    h->add(lname(l, "1.fn.net.3.weight"), (size_t)2 * kDim * kDim);
    h->add(lname(l, "1.fn.net.3.bias"), kDim);
  }
  h->add("rel_out.weight", (size_t)cfg->num_out * kDim);
  h->add("rel_out.bias", cfg->num_out);
  size_t off = 0;
  for (Param& q : h->params) {
    q.offset = off;
    off += align_up(q.numel, 64);
  }
  hipError_t e = hipMalloc((void**)&h->raw, off * sizeof(float));
  if (e != hipSuccess) { delete h; return fail(VETO_ERR_HIP, "hipMalloc(raw weights): %s", hipGetErrorString(e)); }

  // derived weights
  size_t doff = 0;
  auto dtake = [&](size_t bytes) { size_t o = doff; doff += align_up(bytes, 256); return o; };
  std::vector<size_t> lo_(L * 8);
  for (int l = 0; l < L; ++l) {
    lo_[l * 8 + 0] = dtake((size_t)3 * kDim * kDim * 4);
    lo_[l * 8 + 2] = dtake((size_t)kDim * kDim * 4);
    lo_[l * 8 + 4] = dtake((size_t)2 * kDim * kDim * 4);
    lo_[l * 8 + 6] = dtake((size_t)2 * kDim * kDim * 4);
    if (cfg->precision != VETO_PRECISE) {
      lo_[l * 8 + 1] = dtake((size_t)3 * kDim * kDim * 4);
      lo_[l * 8 + 3] = dtake((size_t)kDim * kDim * 4);
      lo_[l * 8 + 5] = dtake((size_t)2 * kDim * kDim * 4);
      lo_[l * 8 + 7] = dtake((size_t)2 * kDim * kDim * 4);
    }
  }
  const size_t o_exp = dtake((size_t)L * 4 * sizeof(int));
  const size_t o_pw = dtake((size_t)2 * kDim * 2048 * 4);
  const size_t o_pb = dtake((size_t)2 * kDim * 4);
  const size_t o_loc = dtake((size_t)kPosDim * 2 * kDim * 4);
  const size_t o_cls = dtake((size_t)E * 2 * kDim * 4);
  const size_t o_head = dtake((size_t)kDim * cfg->num_out * 4);
  const int fold_dhp = fold_block_width(cfg->heads);
  const size_t fold_np = (size_t)cfg->heads * (fold_dhp > 0 ? fold_dhp : 0);
  const size_t fold_el = fold_dhp > 0 ? (size_t)cfg->heads * kDim * fold_np : (size_t)cfg->heads * kDim * kDim;   // largest staged matrix
  const size_t o_fm = dtake(fold_dhp > 0 ? 256 : fold_el * 4), o_fn = dtake(fold_dhp > 0 ? 256 : fold_el * 4), o_ft = dtake(fold_el * 4);
  const size_t o_fq = dtake(fold_np * kDim * 4 + 256), o_fk = dtake((size_t)cfg->heads * kDim * fold_np * 4 + 256),
               o_fv = dtake((size_t)cfg->heads * kDim * fold_np * 4 + 256), o_fo = dtake(fold_np * kDim * 4 + 256);
  const size_t o_q0w = dtake((size_t)3 * kDim * kDim * 4), o_q0v = dtake((size_t)3 * 3 * kDim * 4);
  e = hipMalloc((void**)&h->derived, doff);
  if (e != hipSuccess) { (void)hipFree(h->raw); delete h; return fail(VETO_ERR_HIP, "hipMalloc(derived weights): %s", hipGetErrorString(e)); }
  h->layers.resize(L);
  for (int l = 0; l < L; ++l) {
    LayerW& w = h->layers[l];
    char* d = h->derived;
    w.qkv = (__bf16*)(d + lo_[l * 8 + 0]);
    w.out = (__bf16*)(d + lo_[l * 8 + 2]);
    w.fc1 = (__bf16*)(d + lo_[l * 8 + 4]);
    w.fc2 = (__bf16*)(d + lo_[l * 8 + 6]);
    if (cfg->precision != VETO_PRECISE) {
      w.qkv_m = (__bf16*)(d + lo_[l * 8 + 1]);
      w.out_m = (__bf16*)(d + lo_[l * 8 + 3]);
      w.fc1_m = (__bf16*)(d + lo_[l * 8 + 5]);
      w.fc2_m = (__bf16*)(d + lo_[l * 8 + 7]);
      w.exp_m = (int*)(d + o_exp) + 4 * l;
    }
    w.ln1_w = h->p(lname(l, "0.norm.weight")); w.ln1_b = h->p(lname(l, "0.norm.bias"));
    w.ln2_w = h->p(lname(l, "1.norm.weight")); w.ln2_b = h->p(lname(l, "1.norm.bias"));
    w.out_b = h->p(lname(l, "0.fn.to_out.0.bias"));
    w.fc1_b = h->p(lname(l, "1.fn.net.0.bias"));
    w.fc2_b = h->p(lname(l, "1.fn.net.3.bias"));
  }
  h->patch_w = (__bf16*)(h->derived + o_pw);
  h->patch_bias = (float*)(h->derived + o_pb);
  h->loc_wt = (float*)(h->derived + o_loc);
  h->cls_wt = (float*)(h->derived + o_cls);
  h->head_wt = (float*)(h->derived + o_head);
  h->fold_m = (__bf16*)(h->derived + o_fm);
  h->fold_n = (__bf16*)(h->derived + o_fn);
  h->fold_tmp = (float*)(h->derived + o_ft);
  h->fold_q = (__bf16*)(h->derived + o_fq);
  h->fold_k = (__bf16*)(h->derived + o_fk);
  h->fold_v = (__bf16*)(h->derived + o_fv);
  h->fold_o = (__bf16*)(h->derived + o_fo);
  h->fold_dhp = fold_dhp;
  h->q0_w = (__bf16*)(h->derived + o_q0w);
  h->q0_vec = (float*)(h->derived + o_q0v);
  if (cfg->precision == VETO_MIXED) {
    e = hipMalloc((void**)&h->sat_buf, (size_t)L * VETO_SAT_SITES * 4 * sizeof(unsigned long long));
    if (e != hipSuccess) { (void)hipFree(h->raw); (void)hipFree(h->derived); delete h; return fail(VETO_ERR_HIP, "hipMalloc(saturation counters): %s", hipGetErrorString(e)); }
  }
  *out = h;
  return VETO_OK;
}

int veto_destroy(veto_handle_t h) {
  if (!h) return VETO_OK;
  // teardown: nothing useful can be done with a failure here
  for (ProfRec& r : h->prof_recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
  (void)hipFree(h->raw);
  (void)hipFree(h->derived);
  if (h->sat_buf) (void)hipFree(h->sat_buf);
  delete h;
  return VETO_OK;
}

int veto_num_weights(veto_handle_t h) { return h ? (int)h->params.size() : fail(VETO_ERR_INVALID, "null handle"); }

int veto_weight_info(veto_handle_t h, int index, const char** name, size_t* numel) {
  if (!h || index < 0 || index >= (int)h->params.size()) return fail(VETO_ERR_INVALID, "bad weight index");
  if (name) *name = h->params[index].name.c_str();
  if (numel) *numel = h->params[index].numel;
  return VETO_OK;
}

int veto_load_weights(veto_handle_t h, const char* name, const float* src, size_t numel, void* stream) {
  if (!h || !name || !src) return fail(VETO_ERR_INVALID, "null argument");
  auto it = h->index.find(name);
  if (it == h->index.end()) return fail(VETO_ERR_INVALID, "unknown weight '%s'", name);
  Param& q = h->params[it->second];
  if (q.numel != numel) return fail(VETO_ERR_INVALID, "weight '%s': expected %zu elements, got %zu", name, q.numel, numel);
  HIP_TRY(hipMemcpyAsync(h->raw + q.offset, src, numel * sizeof(float), hipMemcpyDefault, (hipStream_t)stream));
  q.loaded = true;
  h->dirty = true;
  h->train_gen.clear();   // (no workspace's saved activations match the weights any more; also bounds the map)
  return VETO_OK;
}

size_t veto_workspace_bytes(veto_handle_t h, int32_t n_obj, int32_t n_pair) {
  if (!h || n_obj <= 0 || n_pair <= 0) return 0;
  const int chunk = n_pair < h->chunk ? n_pair : h->chunk;
  return carve(nullptr, n_obj, n_pair, chunk).total;
}

static int forward_impl(veto_handle_t h, void* stream, const veto_inputs_t* in, void* workspace, size_t workspace_bytes, float* out_logits,
                        const veto_debug_outputs_t* dbg, unsigned long long* sat);

int veto_forward(veto_handle_t h, void* stream, const veto_inputs_t* in, void* workspace,
                 size_t workspace_bytes, float* out_logits, const veto_debug_outputs_t* dbg) {
  return forward_impl(h, stream, in, workspace, workspace_bytes, out_logits, dbg, nullptr);
}

// sat != nullptr (veto_forward_saturation): device counters [layers][VETO_SAT_SITES][4]
static int forward_impl(veto_handle_t h, void* stream, const veto_inputs_t* in, void* workspace, size_t workspace_bytes, float* out_logits,
                        const veto_debug_outputs_t* dbg, unsigned long long* sat) {
  if (!h || !in || !out_logits) return fail(VETO_ERR_INVALID, "null argument");
  if (in->struct_size != (int32_t)sizeof(veto_inputs_t)) return fail(VETO_ERR_INVALID, "veto_inputs_t size mismatch");
  if (dbg && dbg->struct_size != (int32_t)sizeof(veto_debug_outputs_t)) return fail(VETO_ERR_INVALID, "veto_debug_outputs_t size mismatch");
  if (in->n_obj <= 0 || in->n_pair <= 0 || in->n_img <= 0) return fail(VETO_ERR_INVALID, "empty batch (n_obj=%d n_pair=%d n_img=%d)", in->n_obj, in->n_pair, in->n_img);
  if (!in->roi_rgb || !in->roi_depth || !in->boxes || !in->rel_pairs || !in->img_obj_offset || !in->img_pair_offset)
    return fail(VETO_ERR_INVALID, "missing input pointer");
  if (!in->obj_labels && !in->obj_logits) return fail(VETO_ERR_INVALID, "need obj_labels or obj_logits");
  if ((size_t)in->n_obj * 16 > (size_t)1 << 30 || (size_t)in->n_pair * kTokens > (size_t)1 << 30)
    return fail(VETO_ERR_INVALID, "batch too large");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (h->dirty || h->infer_dirty) {
    int rc = finalize_weights(h, s);
    if (rc != VETO_OK) return rc;
  }
  const int n_obj = in->n_obj, n_pair = in->n_pair;
  const int chunk = n_pair < h->chunk ? n_pair : h->chunk;
  const size_t need = carve(nullptr, n_obj, n_pair, chunk).total;
  if (!workspace || workspace_bytes < need) return fail(VETO_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  if (((uintptr_t)workspace & 255) != 0) return fail(VETO_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  Workspace ws = carve((char*)workspace, n_obj, n_pair, chunk);
  const int L = h->cfg.layers, H = h->cfg.heads, n_out = h->cfg.num_out;
  const std::string T = kT;

  // ---- stage 0: pair indices + per-object partial products ------------------------------------
  {
    ProfScope ps(h, s, "pair_indices", 0, (double)n_pair * 24);
    HIP_TRY(launch_pair_indices(in->rel_pairs, in->img_obj_offset, in->img_pair_offset, in->n_img, n_pair, ws.subj,
                                ws.obj, dbg ? dbg->subj_inds : nullptr, dbg ? dbg->obj_inds : nullptr, s));
  }
  {
    ObjPrepArgs a{};
    a.boxes = in->boxes; a.box_mode = in->box_mode; a.labels = in->obj_labels; a.obj_logits = in->obj_logits;
    a.embed = h->p("obj_embed.weight"); a.num_obj_cls = h->cfg.num_obj_cls; a.embed_dim = h->cfg.embed_dim;
    a.bn_w = h->p("pos_embed.0.weight"); a.bn_b = h->p("pos_embed.0.bias");
    a.bn_mean = h->p("pos_embed.0.running_mean"); a.bn_var = h->p("pos_embed.0.running_var");
    if (in->bn_batch_stats) {   // training-mode BatchNorm: this batch's statistics (biased variance)
      HIP_TRY(launch_bn_batch_stats(in->boxes, in->box_mode, n_obj, in->bn_batch_stats, s));
      a.bn_mean = in->bn_batch_stats; a.bn_var = in->bn_batch_stats + 4;
    }
    a.pos_w = h->p("pos_embed.1.weight"); a.pos_b = h->p("pos_embed.1.bias");
    a.loc_wt = h->loc_wt; a.loc_b = h->p("location_projection.0.bias");
    a.cls_wt = h->cls_wt; a.cls_b = h->p("class_projection.0.bias");
    a.lc = ws.lc; a.pos_out = nullptr; a.n_obj = n_obj;
    ProfScope ps(h, s, "obj_prep", 2.0 * n_obj * 2 * kDim * (kPosDim + h->cfg.embed_dim), (double)n_obj * 2 * 2 * kDim * 4);
    HIP_TRY(launch_obj_prep(a, s));
  }
  {
    ProfScope ps(h, s, "patchify", 0, (double)n_obj * 2 * 256 * 64 * (4 + 4));
    HIP_TRY(launch_patchify(in->roi_depth, in->roi_rgb, ws.pa, n_obj, s));
  }
  {
    int rc = run_gemm(h, s, "gemm_patch", ws.pa, h->patch_w, h->patch_bias, nullptr, 0, ws.patch_tab, nullptr, 2 * kDim,
                      n_obj * 16, 2 * kDim, 2048, EPI_F32);
    if (rc) return rc;
  }
  // Layer 0 in the per-object form: LayerNorm + QKV of the 16 patch tokens of every pair come from two
  // per-object tables SW = S W'^T, OW = O W'^T (S | O = the halves of patch_tab, W' = Wqkv diag(gamma)) -- a GEMM over the
  // n_obj*16 object rows instead of the n_pair*19 token rows.  Needs a layer behind it that reads LN1 rows as usual (L >= 2).
  static const bool tables_off = env_knob_is("VETO_QKV0_TABLES", "0");   // A/B knob of the parity tests
  const bool qkv0_tables = L >= 2 && !tables_off;
  static const bool fold_off = env_knob_is("VETO_CLS_FOLD", "0");         // A/B knob of the parity tests
  const bool fold_last = !fold_off && H <= cls_fold_max_heads();   // last layer in the folded CLS form (attention.hip)
  // VETO_MIXED: the four token-row Linears of every layer but the last, and layer 0's QKV launches of the location / class token
  // rows, take fp16 + e4m3 operands (common.h); the other per-object and
  // CLS-row GEMMs (8 % of the GEMM work) stay on split-bf16 operands.  The out projection only behind the MFMA attention kernel.
  // VETO_FAST = VETO_MIXED's launches with the correction stages of the two fused token-row launches skipped (fp16 main product only:
  // what the schedule costs with the precision terms free; logit error ~2e-3, reported, never parity-grade)
  const bool mixed = h->cfg.precision != VETO_PRECISE;
  const bool fast = h->cfg.precision == VETO_FAST;
  const bool mixed_out = mixed && attention_reads_tables(H);
#ifndef VETO_CLS_FFN_MIXED
#define VETO_CLS_FFN_MIXED 1
#endif
  // the CLS rows' FeedForward of the last layer on mixed operands too (round 5; rounds 2-4 kept it on split-bf16 because those rows ARE the
  // classifier's input): 0.141 -> 0.119 ms for the two launches, logit error against the CPU oracle 6.2e-5 -> 6.6e-5 on the bench batch -- inside
  // the 3e-4 the parity tests hold the mode to (-DVETO_CLS_FFN_MIXED=0: the split-bf16 form)
  const bool cls_mixed = mixed && VETO_CLS_FFN_MIXED;
  // VETO_MIXED runs everything of a layer behind its attention as ONE panel launch (ffn_fused.hip MODE 2); VETO_TAIL_FUSED=0 (a knob
  // the parity tests compare against) splits it into the out projection + LayerNorm2 launch and the FeedForward + LayerNorm1 launch
  // of the same kernel.  VETO_PRECISE / VETO_FAST take the launch-per-Linear GEMMs and LayerNorm launches.
  static const bool tail_off = env_knob_is("VETO_TAIL_FUSED", "0");
  const bool tail_fused = !tail_off && !sat;
  const bool panel = !sat;        // (the saturation audit needs the LayerNorm2 rows and the hidden activation in memory)
  auto count_sat = [&](int layer, int site, const void* rows, long stride_bytes, int n_rows, int K) -> hipError_t {
    if (!sat) return hipSuccess;
    return launch_count_saturation(rows, stride_bytes, n_rows, K, sat + ((size_t)layer * VETO_SAT_SITES + site) * 4, s);
  };
  static const bool qkv_f24_off = env_knob_is("VETO_QKV_F24", "0");
  // middle layers: QKV projection + attention as ONE launch (qkv_attn_fused.hip): q / k / v never reach memory.  The attention output
  // then lives in ws.big (every head's tile reads all rows of ws.a), the layer tail takes it from there and writes the next layer's
  // LayerNorm1 rows back to ws.a.  VETO_QKV_ATTN_FUSED=0 (a knob the parity tests compare against): the two launches.
  static const bool qa_off = env_knob_is("VETO_QKV_ATTN_FUSED", "0");
  const bool qa_fused = mixed && mixed_out && tail_fused && !qa_off && qkv_attn_fused_supports(H);
  // VETO_X_F24=1 (off by default; round 6, measured null): the residual stream BETWEEN the layers as 3-byte floats (common.h: a 16-bit
  // significand) -- token assembly writes them, every layer tail reads them, every tail but the last writes them: a quarter fewer bytes in
  // the two bursts of a panel boundary.  The last tail writes fp32 rows for the last layer's CLS-row kernels, into the buffer that held
  // the LayerNorm1 rows (dead by then: rows of another pitch cannot go over 3-byte rows that other workgroups have yet to read; the
  // folded last layer keeps its own operands in ws.big), so the form needs a fused QKV + attention launch in front of the last tail
  // (L >= 3).  Same-box A/B on the bench batch: 10.99 / 11.03 ms with, 10.98 / 11.02 ms without (profiles/r06_tail_variants.txt) -- the
  // bursts are bound by their request count (16 row pieces per wave instruction either way), not by their bytes -- at a logit error of
  // 1.0-1.4e-4 instead of 5-7e-5.  Kept as a tested variant, not as the default.
  static const bool x_f24_on = env_knob_is("VETO_X_F24", "1");
  const bool x_f24 = x_f24_on && !fast && mixed && mixed_out && tail_fused && fold_last && qkv0_tables && qa_fused && L >= 3;
  if (qkv0_tables) {
    const int R = n_obj * 16;
    HIP_TRY(launch_centre_split(ws.patch_tab, ws.ptab_split, R, s));
    int rc = run_gemm(h, s, "gemm_qkv0_tab", ws.ptab_split, h->q0_w, h->q0_vec + 3 * kDim, nullptr, 0, ws.sw, nullptr, 3 * kDim, R, 3 * kDim, kDim,
                      EPI_F32, (long)2 * 2 * kDim, 0);
    if (rc) return rc;
    rc = run_gemm(h, s, "gemm_qkv0_tab", ws.ptab_split + 2 * kDim, h->q0_w, nullptr, nullptr, 0, ws.ow, nullptr, 3 * kDim, R, 3 * kDim, kDim,
                  EPI_F32, (long)2 * 2 * kDim, 0);
    if (rc) return rc;
  }

  // ---- pairs, in chunks that bound the workspace ----------------------------------------------
  for (int c0 = 0; c0 < n_pair; c0 += chunk) {
    const int np = (n_pair - c0 < chunk) ? n_pair - c0 : chunk;
    const int M = np * kTokens;
    float* qkv = (float*)ws.big;
    __bf16* hid = (__bf16*)ws.big;  // MLP hidden, split rows [M, 2*1152] (qkv is dead by then)
    {
      AssembleArgs a{};
      a.patch_tab = ws.patch_tab; a.lc = ws.lc; a.cls_token = h->p(T + "cls_token");
      a.pos_embedding = h->p(T + "pos_embedding");
      a.ln_w = h->layers[0].ln1_w; a.ln_b = h->layers[0].ln1_b;
      a.subj = ws.subj + c0; a.obj = ws.obj + c0; a.x = ws.x; a.a = ws.a; a.n_pair = np;
      a.stats = qkv0_tables ? ws.stats : nullptr;
      a.x_f24 = x_f24 ? 1 : 0;
      a.a_fmt = mixed && qkv0_tables ? FMT_MIXED : FMT_SPLIT;   // (the rows of tokens 17 / 18: the A operand of gemm_qkv0_lc below)
      // bytes = what the kernel WRITES (its HBM stream; the per-object rows it gathers are cache-resident): the fp32 token rows
      // plus either their LayerNorm'ed split copy, or -- per-object layer 0 -- the row statistics and the split rows of tokens 17, 18
      ProfScope ps(h, s, "assemble_tokens", 0, (double)M * kDim * (x_f24 ? 3 : 4) + (qkv0_tables ? (double)M * 8 + 2.0 * np * kDim * 4 : (double)M * kDim * 4));
      HIP_TRY(launch_assemble(a, s));
    }
    if (dbg && dbg->tokens) {
      if (x_f24) HIP_TRY(launch_unpack_f24(ws.x, dbg->tokens + (size_t)c0 * kTokens * kDim, (size_t)M * kDim, s));
      else HIP_TRY(hipMemcpyAsync(dbg->tokens + (size_t)c0 * kTokens * kDim, ws.x, (size_t)M * kDim * 4, hipMemcpyDeviceToDevice, s));
    }
    const float* xlast = ws.x;      // the residual stream the last layer reads (x_f24: the fp32 rows the last tail wrote)
    for (int l = 0; l < L; ++l) {
      const LayerW& w = h->layers[l];
      const bool last = (l == L - 1);
      int rc;
      bool qkv_f24 = false;
      bool attn_in_big = false;      // this layer's attention output is in ws.big (fused QKV + attention launch)
      const bool fold = last && fold_last;
      if (fold) {
        // last layer, folded (attention.hip): u = a_0 . Mcat on the CLS rows, per-pair scores / softmax / weighted token means,
        // then out = abar . Ncat^T + b_o + x_0 -- no key / value projection of the 19 tokens
        float* u = (float*)ws.big;
        __bf16* abar = (__bf16*)(ws.big + align_up((size_t)gemm_rows_padded(np) * H * kDim * 4, 256));
        {   // LayerNorm1 of the CLS rows (row p*19 of x -> compact split row p): the A operand of the u GEMM
          ProfScope ps(h, s, "layernorm_cls", 0, (double)np * kDim * 8);
          HIP_TRY(launch_layernorm(xlast, (long)kTokens * kDim, w.ln1_w, w.ln1_b, ws.ac, np, s));
        }
        const int dhp = h->fold_dhp, npad = H * dhp;   // block form: padded width of the per-head q / v rows
        bool u24 = false;
        if (dhp > 0) {
          // q0 = a0 Wq_pad^T as split rows (every head's dh columns padded to dhp), then u = q0 . blockdiag(Wk): column tile n of u
          // belongs to head n / 3 and multiplies that head's dhp / 32 k-steps only
          rc = run_gemm(h, s, "gemm_q_cls", ws.ac, h->fold_q, nullptr, nullptr, 0, nullptr, ws.hc, 2L * npad, np, npad, kDim, EPI_SPLIT);
          if (rc) return rc;
          u24 = cls_fold_reads_f24();      // u as 3-byte floats: its consumer splits it into bf16 hi + lo, a 16-bit significand
          rc = run_gemm(h, s, "gemm_u_cls", ws.hc, h->fold_k, nullptr, nullptr, 0, u, nullptr, (long)H * kDim, np, H * kDim, npad,
                        u24 ? EPI_F24 : EPI_F32, 0, 0, DropSite(), nullptr, 3, dhp / 32);
        } else {
          rc = run_gemm(h, s, "gemm_u_cls", ws.ac, h->fold_m, nullptr, nullptr, 0, u, nullptr, (long)H * kDim, np, H * kDim, kDim, EPI_F32);
        }
        if (rc) return rc;
        {
          ProfScope ps(h, s, "attention_cls", 4.0 * np * H * kTokens * kDim, (double)M * kDim * 4 + (double)np * H * kDim * 8);
          HIP_TRY(launch_cls_fold_attention(xlast, w.ln1_w, w.ln1_b, u, abar, np, H, s, u24));
        }
        if (dhp > 0) {
          // vbar = abar . blockdiag(Wv)^T as split rows (column tile n covers the 192 / dhp heads whose 576-wide k blocks it needs),
          // then out = vbar Wo_pad^T + b_o + x_0
          const int hpt = 192 / dhp;
          rc = run_gemm(h, s, "gemm_v_cls", abar, h->fold_v, nullptr, nullptr, 0, nullptr, ws.hc, 2L * npad, np, npad, H * kDim, EPI_SPLIT,
                        0, 0, DropSite(), nullptr, 1, hpt * kDim / 32);
          if (rc) return rc;
          rc = run_gemm(h, s, "gemm_out_cls", ws.hc, h->fold_o, w.out_b, xlast, (long)kTokens * kDim, ws.xc, nullptr, kDim, np, kDim, npad,
                        EPI_RESID);
        } else {
          rc = run_gemm(h, s, "gemm_out_cls", abar, h->fold_n, w.out_b, xlast, (long)kTokens * kDim, ws.xc, nullptr, kDim, np, kDim,
                        H * kDim, EPI_RESID);
        }
        if (rc) return rc;
      } else if (l == 0 && qkv0_tables) {
        if (!attention_reads_tables(H)) {   // head widths without an MFMA attention: materialise the rows of tokens 0..16
          ProfScope ps(h, s, "qkv0_combine", 0, (double)np * 17 * 3 * kDim * 4 * 3);
          HIP_TRY(launch_qkv0_combine(ws.sw, ws.ow, ws.stats, h->q0_vec, ws.subj + c0, ws.obj + c0, qkv, np, s));
        }
        for (int t = kTokens - 2; t < kTokens; ++t) {   // the ReLU'd location / class rows: LayerNorm'ed rows x Wqkv as usual (VETO_MIXED: mixed operands)
          if (mixed) HIP_TRY(count_sat(l, VETO_SAT_QKV_IN, ws.a + (size_t)t * 2 * kDim, (long)kTokens * kDim * 4, np, kDim));
          rc = run_gemm(h, s, "gemm_qkv0_lc", ws.a + (size_t)t * 2 * kDim, mixed ? w.qkv_m : w.qkv, nullptr, nullptr, 0,
                        qkv + (size_t)t * 3 * kDim, nullptr, (long)kTokens * 3 * kDim, np, 3 * kDim, kDim, EPI_F32, (long)kTokens * 2 * kDim, 0,
                        DropSite(), mixed ? w.exp_m + 0 : nullptr);
          if (rc) return rc;
        }
      } else if (!last && qa_fused && l > 0) {
        QkvAttnArgs q{};
        q.a = (const char*)ws.a; q.w = (const char*)w.qkv_m; q.w_exp = w.exp_m + 0; q.o = ws.big; q.n_pair = np; q.heads = H; q.fast = fast ? 1 : 0;
        ProfScope ps(h, s, "qkv_attn_fused", 2.0 * M * 3.0 * kDim * kDim + 4.0 * np * kTokens * kTokens * kDim,
                     (double)M * kDim * 8 + 3.0 * kDim * kDim * 4);
        HIP_TRY(launch_qkv_attn_fused(q, s));
        attn_in_big = true;
      } else if (!last) {
        const bool mq = mixed && l > 0;   // layer 0's LayerNorm'ed rows come from token assembly (split rows)
        // q / k / v as 3-byte floats between this GEMM and the attention kernel (common.h; VETO_QKV_F24=0: fp32)
        qkv_f24 = mq && !qkv_f24_off && attention_reads_tables(H);
        if (mq) HIP_TRY(count_sat(l, VETO_SAT_QKV_IN, ws.a, (long)kDim * 4, M, kDim));
        rc = run_gemm(h, s, "gemm_qkv", ws.a, mq ? w.qkv_m : w.qkv, nullptr, nullptr, 0, qkv, nullptr, 3 * kDim, M, 3 * kDim, kDim,
                      qkv_f24 ? EPI_F24 : EPI_F32, 0, 0, DropSite(), mq ? w.exp_m + 0 : nullptr);
        if (rc) return rc;
      } else {
        // last layer: keys/values for all 19 tokens, the query for the CLS row of each pair only
        rc = run_gemm(h, s, "gemm_kv_last", ws.a, w.qkv, nullptr, nullptr, 0, qkv + kDim, nullptr, 3 * kDim, M, 2 * kDim,
                      kDim, EPI_F32, 0, kDim);
        if (rc) return rc;
        rc = run_gemm(h, s, "gemm_q_cls", ws.a, w.qkv, nullptr, nullptr, 0, qkv, nullptr, (long)kTokens * 3 * kDim, np, kDim,
                      kDim, EPI_F32, (long)kTokens * 2 * kDim, 0);
        if (rc) return rc;
      }
      if (!fold && !attn_in_big) {
        AttnArgs a{};
        a.qkv = qkv; a.n_pair = np; a.heads = H; a.cls_only = last ? 1 : 0;
        a.qkv_f24 = qkv_f24 ? 1 : 0;
        a.o = last ? ws.ac : ws.a;
        a.o_fmt = (!last && mixed_out) ? FMT_MIXED : FMT_SPLIT;
        if (l == 0 && qkv0_tables && attention_reads_tables(H)) {   // q / k / v of the patch tokens are formed on load
          a.sw = ws.sw; a.ow = ws.ow; a.stats = ws.stats; a.vec = h->q0_vec; a.subj = ws.subj + c0; a.obj = ws.obj + c0;
        }
        const double nq = last ? 1 : kTokens;
        ProfScope ps(h, s, last ? "attention_cls" : "attention", 4.0 * np * nq * kTokens * kDim,
                     (double)M * 3 * kDim * (qkv_f24 ? 3 : 4) + (double)np * nq * kDim * 4);
        HIP_TRY(launch_attention(a, s));
        if (a.o_fmt == FMT_MIXED) HIP_TRY(count_sat(l, VETO_SAT_ATTN_OUT, ws.a, (long)kDim * 4, M, kDim));
      }
      if (!last) {
        // ... and the LayerNorm in front of the next layer's QKV GEMM in the FeedForward epilogue, when that GEMM takes mixed rows
        const bool ffn_ln_next = mixed && panel && l + 1 < L - 1;
        if (mixed_out && tail_fused) {
          // everything of the layer behind its attention in ONE launch (ffn_fused.hip, MODE 2): x1 = x + a Wo^T + bo stays in
          // registers, LayerNorm2(x1) is written in place over the attention output and streamed back as the FeedForward's
          // input, fc2 accumulates on top of x1, the epilogue stores x (and the next layer's LayerNorm1 rows)
          FfnArgs f{};
          f.a = (const char*)ws.a; f.wo = (const char*)w.out_m; f.bo = w.out_b; f.expo = w.exp_m + 1; f.lnm_w = w.ln2_w; f.lnm_b = w.ln2_b;
          f.w1 = (const char*)w.fc1_m; f.w2 = (const char*)w.fc2_m; f.b1 = w.fc1_b; f.b2 = w.fc2_b; f.exp1 = w.exp_m + 2; f.exp2 = w.exp_m + 3;
          f.resid = ws.x; f.out = ws.x; f.ldr = kDim; f.ldo = kDim; f.M = M; f.ln_out = (char*)ws.a;
          if (attn_in_big) { f.a = ws.big; f.ln_out = ws.big; f.ln1_out = (char*)ws.a; }
          if (ffn_ln_next) { f.ln_w = h->layers[l + 1].ln1_w; f.ln_b = h->layers[l + 1].ln1_b; }
          f.fast = fast ? 1 : 0;
          if (x_f24) {
            f.resid_f24 = 1;
            if (l + 1 < L - 1) f.out_f24 = 1;
            else {      // the last tail: fp32 rows for the folded last layer, into a buffer that is dead by now (rows of another pitch
                        // cannot go over the 3-byte rows other workgroups have yet to read)
              if (!attn_in_big) return fail(VETO_ERR_INVALID, "internal: 3-byte residual rows without a free buffer for the last tail's fp32 rows");
              f.out = (float*)ws.a;
              xlast = (const float*)ws.a;
            }
          }
          ProfScope ps(h, s, "layer_tail_fused", 2.0 * M * (double)kDim * kDim + 2.0 * 2.0 * M * (double)kDim * 2 * kDim,
                       (double)M * kDim * (x_f24 ? (f.out_f24 ? 14 : 15) : 16) + 5.0 * kDim * kDim * 4);
          HIP_TRY(launch_layer_tail(f, s));
        } else {
        if (mixed_out && panel) {
          // out projection + residual + LayerNorm2 in one launch on full rows (ffn_fused.hip, MODE 1): x <- x + a Wo^T + bo, then
          // a <- LayerNorm2(x) as mixed rows in place over the attention output
          FfnArgs f{};
          f.a = (const char*)ws.a; f.w2 = (const char*)w.out_m; f.b2 = w.out_b; f.resid = ws.x; f.out = ws.x; f.ldr = kDim; f.ldo = kDim;
          f.M = M; f.exp2 = w.exp_m + 1; f.ln_w = w.ln2_w; f.ln_b = w.ln2_b; f.ln_out = (char*)ws.a;
          ProfScope ps(h, s, "out_ln_fused", 2.0 * M * (double)kDim * kDim, (double)M * kDim * 16 + (double)kDim * kDim * 4);
          HIP_TRY(launch_out_fused(f, s));
        } else {
          rc = run_gemm(h, s, "gemm_out", ws.a, mixed_out ? w.out_m : w.out, w.out_b, ws.x, kDim, ws.x, nullptr, kDim, M, kDim, kDim, EPI_RESID,
                        0, 0, DropSite(), mixed_out ? w.exp_m + 1 : nullptr);
          if (rc) return rc;
          {
            ProfScope ps(h, s, "layernorm", 0, (double)M * kDim * 8);
            HIP_TRY(launch_layernorm(ws.x, kDim, w.ln2_w, w.ln2_b, ws.a, M, s, mixed ? FMT_MIXED : FMT_SPLIT));
          }
          if (mixed) HIP_TRY(count_sat(l, VETO_SAT_FFN_IN, ws.a, (long)kDim * 4, M, kDim));
        }
        if (mixed && panel) {
          // FeedForward in one launch (ffn_fused.hip): the hidden activation never leaves the CU
          FfnArgs f{};
          f.a = (const char*)ws.a; f.w1 = (const char*)w.fc1_m; f.w2 = (const char*)w.fc2_m; f.b1 = w.fc1_b; f.b2 = w.fc2_b;
          f.resid = ws.x; f.out = ws.x; f.ldr = kDim; f.ldo = kDim; f.M = M; f.exp1 = w.exp_m + 2; f.exp2 = w.exp_m + 3;
          if (ffn_ln_next) {   // the next layer's LayerNorm1 in the epilogue (mixed rows, in place over this launch's input rows)
            f.ln_w = h->layers[l + 1].ln1_w; f.ln_b = h->layers[l + 1].ln1_b; f.ln_out = (char*)ws.a;
          }
          // bytes: the LayerNorm'ed rows in, the residual stream in and out, the two weight matrices
          ProfScope ps(h, s, "ffn_fused", 2.0 * 2.0 * M * (double)kDim * 2 * kDim, (double)M * kDim * 12 + 2.0 * 2 * kDim * kDim * 4);
          HIP_TRY(launch_ffn_fused(f, s));
        } else {
          rc = run_gemm(h, s, "gemm_fc1", ws.a, mixed ? w.fc1_m : w.fc1, w.fc1_b, nullptr, 0, nullptr, hid, 4 * kDim, M, 2 * kDim, kDim,
                        EPI_GELU_SPLIT, 0, 0, DropSite(), mixed ? w.exp_m + 2 : nullptr);
          if (rc) return rc;
          if (mixed) HIP_TRY(count_sat(l, VETO_SAT_HIDDEN, hid, (long)2 * kDim * 4, M, 2 * kDim));
          rc = run_gemm(h, s, "gemm_fc2", hid, mixed ? w.fc2_m : w.fc2, w.fc2_b, ws.x, kDim, ws.x, nullptr, kDim, M, kDim, 2 * kDim, EPI_RESID,
                        0, 0, DropSite(), mixed ? w.exp_m + 3 : nullptr);
          if (rc) return rc;
        }
        }
        {
          const LayerW& nx = h->layers[l + 1];
          if (l + 1 == L - 1 && fold_last) {
            // the folded last layer LayerNorms its token rows itself
          } else if (ffn_ln_next) {
            // written by the fused FeedForward launch
          } else {
            ProfScope ps(h, s, "layernorm", 0, (double)M * kDim * 8);
            // the next layer's QKV GEMM takes mixed rows unless it is the (unfolded) last layer
            HIP_TRY(launch_layernorm(ws.x, kDim, nx.ln1_w, nx.ln1_b, ws.a, M, s, (mixed && l + 1 < L - 1) ? FMT_MIXED : FMT_SPLIT));
          }
        }
      } else {
        // Only x[:, 0] of the last layer is consumed (model_veto.py:23): out-proj, FeedForward and
        // both residuals run on the CLS row of each pair (row p*19 of x -> compact row p).
        if (!fold) {
          rc = run_gemm(h, s, "gemm_out_cls", ws.ac, w.out, w.out_b, ws.x, (long)kTokens * kDim, ws.xc, nullptr, kDim, np,
                        kDim, kDim, EPI_RESID);
          if (rc) return rc;
        }
        {
          ProfScope ps(h, s, "layernorm_cls", 0, (double)np * kDim * 8);
          HIP_TRY(launch_layernorm(ws.xc, kDim, w.ln2_w, w.ln2_b, ws.ac, np, s, cls_mixed ? FMT_MIXED : FMT_SPLIT));
        }
        // (the CLS rows' FeedForward takes mixed operands too, and those rows are the classifier's input: audited like every other site)
        if (cls_mixed) HIP_TRY(count_sat(l, VETO_SAT_FFN_IN, ws.ac, (long)kDim * 4, np, kDim));
        rc = run_gemm(h, s, "gemm_fc1_cls", ws.ac, cls_mixed ? w.fc1_m : w.fc1, w.fc1_b, nullptr, 0, nullptr, ws.hc, 4 * kDim, np, 2 * kDim, kDim,
                      EPI_GELU_SPLIT, 0, 0, DropSite(), cls_mixed ? w.exp_m + 2 : nullptr);
        if (rc) return rc;
        if (cls_mixed) HIP_TRY(count_sat(l, VETO_SAT_HIDDEN, ws.hc, (long)2 * kDim * 4, np, 2 * kDim));
        rc = run_gemm(h, s, "gemm_fc2_cls", ws.hc, cls_mixed ? w.fc2_m : w.fc2, w.fc2_b, ws.xc, kDim, ws.xc, nullptr, kDim, np, kDim, 2 * kDim,
                      EPI_RESID, 0, 0, DropSite(), cls_mixed ? w.exp_m + 3 : nullptr);
        if (rc) return rc;
      }
    }
    {
      ProfScope ps(h, s, "head", 2.0 * np * kDim * n_out, (double)np * (kDim + n_out) * 4);
      HIP_TRY(launch_head(ws.xc, h->head_wt, h->p("rel_out.bias"), out_logits + (size_t)c0 * n_out, np, n_out, s));
    }
    if (dbg && dbg->cls)
      HIP_TRY(hipMemcpyAsync(dbg->cls + (size_t)c0 * kDim, ws.xc, (size_t)np * kDim * 4, hipMemcpyDeviceToDevice, s));
  }
  return VETO_OK;
}

int veto_forward_saturation(veto_handle_t h, void* stream, const veto_inputs_t* in, void* workspace, size_t workspace_bytes,
                            float* out_logits, veto_saturation_t* counts, int32_t capacity) {
  if (!h || !counts) return fail(VETO_ERR_INVALID, "null argument");
  if (h->cfg.precision != VETO_MIXED) return fail(VETO_ERR_INVALID, "veto_forward_saturation audits the VETO_MIXED operands; this handle computes in another mode");
  const int n = h->cfg.layers * VETO_SAT_SITES;
  if (capacity < n) return fail(VETO_ERR_INVALID, "counts holds %d entries, need layers * VETO_SAT_SITES = %d", capacity, n);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (!h->sat_buf) return fail(VETO_ERR_INVALID, "no saturation counters (handle not created in VETO_MIXED)");
  HIP_TRY(hipMemsetAsync(h->sat_buf, 0, (size_t)n * 4 * sizeof(unsigned long long), s));
  const int rc = forward_impl(h, stream, in, workspace, workspace_bytes, out_logits, nullptr, h->sat_buf);
  if (rc != VETO_OK) return rc;
  std::vector<unsigned long long> host((size_t)n * 4);
  HIP_TRY(hipMemcpyAsync(host.data(), h->sat_buf, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) {
    counts[i].elements = (int64_t)host[i * 4 + 0];
    counts[i].f16_saturated = (int64_t)host[i * 4 + 1];
    counts[i].value_saturated = (int64_t)host[i * 4 + 2];
    counts[i].resid_saturated = (int64_t)host[i * 4 + 3];
  }
  return VETO_OK;
}

}  // extern "C"

// ================================================================================================================
// Training path: forward that keeps every activation the backward needs, and the
// backward itself.  One pass over all pairs (no chunking), precise mode, no dropout (the caller refuses p > 0).
// Every layer runs on all 19 tokens (the CLS-only shortcut of the inference path would complicate the backward).
// ================================================================================================================
extern "C" size_t veto_train_workspace_bytes(veto_handle_t h, int32_t n_obj, int32_t n_pair);

namespace {

struct TrainLayer {
  float* xin;    // [mpad, 576]   input of the layer (residual stream)
  __bf16* a1;    // split LN1(xin)
  float* qkv;    // [mpad, 1728]
  __bf16* ao;    // split attention output
  float* xmid;   // [mpad, 576]   after the attention residual
  __bf16* a2;    // split LN2(xmid)
  float* pre;    // [mpad, 1152]  fc1 pre-activation
  __bf16* hid;   // split gelu(pre)
};

struct TrainWs {
  int32_t *subj, *obj;
  float *lc, *patch_tab, *xout;
  __bf16* pa;
  std::vector<TrainLayer> layers;
  // backward scratch
  float *dx, *dmid, *dtmp, *dbig;
  __bf16 *dsplit, *wdg, *zero;
  __bf16* dsplit_b;   // the last two thirds of dsplit: rows of 2 * 1152 bf16 (the fc2 input-gradient epilogue writes fc1's gradient rows there
                      // while its own operand, rows of 2 * 576 bf16, occupies the first third)
  float *ln_partial, *col_partial, *colp, *dgb, *head_partial;
  float *dpatch, *dlc, *dpos, *dpre, *xhat, *bn_out, *dbn_out, *emb, *demb, *prob, *dloc_wt, *dcls_wt, *dwcat_t;
  __bf16* wcat_t;   // Wcat^T [2112, 2*1152] split rows: weight operand of the patch projection's input gradient
  float* dpa;       // dPA [prow, 2112] fp32: gradient w.r.t. the patch rows
  size_t mp2;    // padded reduction length of the weight-gradient GEMMs
  size_t total;
};

// q / k / v of the training path as 3-byte floats (common.h) between the QKV projection, the attention and the attention backward: all
// three split them into 16-bit hi + lo parts anyway, so the 16-bit significand in memory is what they would have kept -- 1.7 KB per token
// row and layer less to keep and to move (round 6).  MFMA head widths only; VETO_TRAIN_QKV_F24=0: fp32.
bool train_qkv_f24(veto_handle_t h) {
  static const bool off = env_knob_is("VETO_TRAIN_QKV_F24", "0");
  return !off && (h->dh == 72 || h->dh == 96);
}

// VETO_TRAIN_RECOMPUTE=1 (off by default): the LayerNorm1 / LayerNorm2 rows and the GELU rows of a layer are not kept for the backward but
// recomputed there from the residual rows and the pre-activation that ARE kept (the same kernels on the same inputs: bit-identical operands,
// bit-identical gradients): 8.6 GB less workspace at cfg-2 for three more passes per layer (~3 ms per step).  With 288 GB of HBM the default keeps them.
bool train_recompute() {
  static const bool on = env_knob_is("VETO_TRAIN_RECOMPUTE", "1");
  return on;
}

// dpre = dh * gelu'(pre) in the epilogue of fc2's input-gradient GEMM (round 6); VETO_TRAIN_GELU_EPI=0: inside the operand preparation of
// the fc1 backward, a pass of its own over an fp32 copy of dh (rounds 1-5)
bool train_gelu_epilogue() {
  static const bool off = env_knob_is("VETO_TRAIN_GELU_EPI", "0");
  return !off;
}

// VETO_TRAIN_LN_SPLIT=1 (off by default; round 6, measured SLOWER): the LayerNorm backward kernels emit the split rows and bias partials of the
// Linear behind them instead of a preparation pass per Linear reading the fp32 gradient rows back.  63.0 ms per step with, 61.6 without on one
// box: the kernel is at its register limit (two statistics passes over 18 values per lane and tensor) and spills with the extra column
// sums, split conversion and dropout hash, and its 4-byte split stores are slower than the pass they replace.  Kept as a tested variant.
bool train_ln_emits_split() {
  static const bool on = env_knob_is("VETO_TRAIN_LN_SPLIT", "1");
  return on;
}

TrainWs carve_train(char* base, veto_handle_t h, int n_obj, int n_pair) {
  TrainWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += align_up(bytes, 256);
    return ptr;
  };
  const int L = h->cfg.layers, E = h->cfg.embed_dim;
  const size_t M = (size_t)n_pair * kTokens;
  const size_t mpad = (size_t)gemm_rows_padded((int)M);
  const size_t prow = (size_t)gemm_rows_padded(n_obj * 16);
  w.subj = (int32_t*)take((size_t)n_pair * 4);
  w.obj = (int32_t*)take((size_t)n_pair * 4);
  w.lc = (float*)take((size_t)n_obj * 2 * 2 * kDim * 4);
  w.pa = (__bf16*)take(prow * 2 * 2048 * 2);
  w.patch_tab = (float*)take((size_t)n_obj * 16 * 2 * kDim * 4);
  w.layers.resize(L);
  const size_t cpad = (size_t)gemm_rows_padded(n_pair);
  const bool recompute = train_recompute();
  __bf16* a_scr = recompute ? (__bf16*)take(mpad * 2 * kDim * 2) : nullptr;      // LayerNorm rows of whichever Linear is next (forward) / being differentiated (backward)
  __bf16* hid_scr = recompute ? (__bf16*)take(mpad * 4 * kDim * 2) : nullptr;    // the same for the GELU rows
  for (int l = 0; l < L; ++l) {
    TrainLayer& t = w.layers[l];
    // (the last layer runs on the pairs' CLS rows behind its attention: those buffers are compact, one row per pair -- sized for every token
    // row until round 6, 4.4 GB too many at cfg-2)
    const size_t rows = l == L - 1 ? cpad : mpad;
    t.xin = (float*)take(mpad * kDim * 4);
    t.a1 = recompute ? a_scr : (__bf16*)take(mpad * 2 * kDim * 2);
    t.qkv = (float*)take(mpad * 3 * kDim * (train_qkv_f24(h) ? 3 : 4));
    t.ao = (__bf16*)take(rows * 2 * kDim * 2);
    t.xmid = (float*)take(rows * kDim * 4);
    t.a2 = recompute ? a_scr : (__bf16*)take(rows * 2 * kDim * 2);
    t.pre = (float*)take(rows * 2 * kDim * 4);
    t.hid = recompute ? hid_scr : (__bf16*)take(rows * 4 * kDim * 2);
  }
  w.xout = (float*)take(cpad * kDim * 4);      // (compact: the last layer's CLS rows)
  w.dx = (float*)take(mpad * kDim * 4);
  w.dmid = (float*)take(mpad * kDim * 4);
  w.dtmp = (float*)take(mpad * kDim * 4);
  // (fc2's input gradient as fp32 rows [M, 1152]: only the VETO_TRAIN_GELU_EPI=0 form writes it; sized [M, 1728] until round 6, 2 GB at cfg-2)
  w.dbig = (float*)take(train_gelu_epilogue() ? 256 : mpad * 2 * kDim * 4);
  {   // the token-row gradients (6 * 576 bf16 per row) and, for the input gradient of the patch projection, the split rows of
      // dpatch (prow rows x 2 * 1152 bf16) share this buffer: size it for the larger (n_obj * 16 can exceed 19 * n_pair / 1.5)
    const size_t tok = mpad * 6 * kDim * 2, obj = prow * 4 * kDim * 2;
    w.dsplit = (__bf16*)take(tok > obj ? tok : obj);
    w.dsplit_b = (__bf16*)((char*)w.dsplit + mpad * 2 * kDim * 2);
  }
  w.mp2 = (M + 32 * 64 + 31) / 32 * 32;   // room for any split count up to 64
  w.zero = (__bf16*)take(1024);           // what the weight-gradient GEMM reads for reduction rows past the last one
  w.wdg = (__bf16*)take((size_t)3 * kDim * kDim * 4);
  w.ln_partial = (float*)take(layernorm_backward_partial_floats((int)M) * 4);
  w.col_partial = (float*)take((size_t)column_sums_chunks() * 3 * kDim * 4);
  w.colp = (float*)take((w.mp2 / 32) * 3 * kDim * 4);
  w.dgb = (float*)take(2 * kDim * 4);
  w.head_partial = (float*)take(head_backward_partial_floats(h->cfg.num_out) * 4);
  w.dpatch = (float*)take((size_t)n_obj * 16 * 2 * kDim * 4);
  w.dlc = (float*)take((size_t)n_obj * 2 * 2 * kDim * 4);
  w.dpos = (float*)take((size_t)n_obj * kPosDim * 4);
  w.dpre = (float*)take((size_t)n_obj * kPosDim * 4);
  w.xhat = (float*)take((size_t)n_obj * 4 * 4);
  w.bn_out = (float*)take((size_t)n_obj * 4 * 4);
  w.dbn_out = (float*)take((size_t)n_obj * 4 * 4);
  w.emb = (float*)take((size_t)n_obj * E * 4);
  w.demb = (float*)take((size_t)n_obj * E * 4);
  w.prob = (float*)take((size_t)n_obj * 256 * 4);
  w.dloc_wt = (float*)take((size_t)kPosDim * 2 * kDim * 4);
  w.dcls_wt = (float*)take((size_t)E * 2 * kDim * 4);
  w.dwcat_t = (float*)take((size_t)2048 * 2 * kDim * 4);
  w.wcat_t = (__bf16*)take((size_t)kPatchTRows * 2 * 2 * kDim * 2);
  w.dpa = (float*)take((size_t)gemm_rows_padded(n_obj * 16) * kPatchTRows * 4);
  w.total = off;
  return w;
}

// Backward of y = x W^T (+ b) over the token rows: dW[N, K] = dY^T x, db[N] = column sums of dY (if db), dX[M, K] = dY W.
// One pass over dY (prep_grad_kernel) produces its split rows -- the A operand of the input-gradient GEMM AND, read through
// transposing LDS loads, of the weight-gradient GEMM (GemmArgs::tn; the saved activation x_split is its other operand as it
// is) -- and the bias partials.  No transposed copies of dY or x exist.
// dy == nullptr: the producer (attention backward) has already written the split rows into w.dsplit; no bias then.
// presplit / presplit_partials: the producer wrote the split rows itself (to `presplit`; nullptr = w.dsplit) together with
// `presplit_partials` rows of column sums in w.colp (0 = none: no bias gradient then).
// gelu_pre (fc2 only): the input gradient is not written as fp32 rows: the GEMM's epilogue multiplies it by gelu'(gelu_pre) and writes the
// split rows of fc1's backward to w.dsplit_b and their column-sum partials to w.colp (*next_partials rows): the pass that read dH and the
// pre-activation back (0.9 ms per layer) is gone.
int run_linear_backward(veto_handle_t h, hipStream_t s, const TrainWs& w, const float* dy, int M, int N, const __bf16* x_split, int K,
                        const float* weight, float* dw, float* db, float* dx, const GradXform& xf = GradXform(),
                        const __bf16* presplit = nullptr, int presplit_partials = 0, const float* gelu_pre = nullptr, int* next_partials = nullptr) {
  if (!dy && db && presplit_partials <= 0) return fail(VETO_ERR_INVALID, "a pre-split gradient without column partials cannot feed a bias gradient");
  const __bf16* gsplit = !dy && presplit ? presplit : w.dsplit;
  const int out_tiles = ((N + 255) / 256) * (K / 192);
  int ks = 2 * 256 / out_tiles;
  const int max_ks = (M + 32 * 64 - 1) / (32 * 64);
  if (ks > max_ks) ks = max_ks;
  if (ks > 64) ks = 64;
  if (ks < 1) ks = 1;
  const size_t mp = ((size_t)M + 32 * (size_t)ks - 1) / (32 * (size_t)ks) * 32 * (size_t)ks;
  if (mp > w.mp2) return fail(VETO_ERR_WORKSPACE, "weight-gradient partial buffer too small");
  if (dy) HIP_TRY(launch_prep_grad(dy, N, M, N, w.dsplit, (int)mp, db ? w.colp : nullptr, xf, s));
  if (db) HIP_TRY(launch_column_sums(w.colp, N, dy ? (int)(mp / 32) : presplit_partials, N, db, w.col_partial, column_sums_chunks(), s));
  HIP_TRY(hipMemsetAsync(dw, 0, (size_t)N * K * 4, s));
  {
    GemmArgs g{};
    g.a = gsplit; g.w = x_split; g.c = dw;
    g.M = N; g.N = K; g.K = (int)mp; g.ldc = K; g.k_splits = ks;
    g.tn = 1; g.lda = 2 * (long)N; g.ldw = 2 * (long)K; g.k_valid = M; g.zero = w.zero;
    ProfScope ps(h, s, "bwd_wgrad", 2.0 * M * (double)N * K, 0);
    HIP_TRY(launch_gemm_split(g, EPI_ATOMIC, 0, s));
  }
  HIP_TRY(launch_transpose_split(weight, K, N, K, w.wdg, N, s));     // W [N, K] -> W^T split rows [K, 2N]
  {
    GemmArgs g{};
    g.a = gsplit; g.w = w.wdg; g.c = dx;
    g.M = M; g.N = K; g.K = N; g.ldc = K;
    ProfScope ps(h, s, "bwd_dgrad", 2.0 * M * (double)N * K, 0);
    if (gelu_pre) {
      if (gsplit == w.dsplit_b || !next_partials) return fail(VETO_ERR_INVALID, "internal: the fused gelu' epilogue writes w.dsplit_b");
      g.c = nullptr; g.c_split = w.dsplit_b; g.ldc = 2L * K; g.resid = gelu_pre; g.ldr = K; g.col_partial = w.colp;
      *next_partials = (M + 255) / 256 * 4;      // one partial row per 64-row slice of every 256-row tile
      HIP_TRY(launch_gemm_split(g, EPI_GELU_BWD, 0, s));
    } else {
      HIP_TRY(launch_gemm_split(g, EPI_F32, 0, s));
    }
  }
  return VETO_OK;
}

// dropout sites of the training path: 1 = pos_embed Dropout(0.1), 2 = pos_drop on the tokens, 3 + l = to_out of layer l
DropSite drop_site(const veto_train_opts_t* o, int site) {
  DropSite d;
  if (!o) return d;
  const float p = site == 1 ? o->p_pos : site == 2 ? o->p_emb : o->p_attn;
  if (!(p > 0.f)) return d;
  d.seed = o->seed + (unsigned long long)site * 0x632BE59BD9B4E019ull;
  d.thresh = (unsigned)(p * 16777216.0f);
  d.scale = 1.f / (1.f - p);
  return d;
}

int check_train_opts(const veto_train_opts_t* o) {
  if (!o) return VETO_OK;
  if (o->struct_size != (int32_t)sizeof(veto_train_opts_t)) return fail(VETO_ERR_INVALID, "veto_train_opts_t size mismatch");
  for (float p : {o->p_pos, o->p_emb, o->p_attn})
    if (!(p >= 0.f && p < 1.f)) return fail(VETO_ERR_INVALID, "dropout probabilities must be in [0, 1)");
  return VETO_OK;
}

int check_train_inputs(veto_handle_t h, const veto_inputs_t* in, void* workspace, size_t workspace_bytes) {
  if (!h || !in || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (in->struct_size != (int32_t)sizeof(veto_inputs_t)) return fail(VETO_ERR_INVALID, "veto_inputs_t size mismatch");
  if (in->n_obj <= 0 || in->n_pair <= 0 || in->n_img <= 0) return fail(VETO_ERR_INVALID, "empty batch");
  if (!in->roi_rgb || !in->roi_depth || !in->boxes || !in->rel_pairs || !in->img_obj_offset || !in->img_pair_offset)
    return fail(VETO_ERR_INVALID, "missing input pointer");
  if (!in->obj_labels && !in->obj_logits) return fail(VETO_ERR_INVALID, "neither obj_labels nor obj_logits given");
  if (!in->bn_batch_stats) return fail(VETO_ERR_INVALID, "the training path needs bn_batch_stats (training-mode BatchNorm)");
  if (h->cfg.precision == VETO_FAST) return fail(VETO_ERR_INVALID, "the training path runs on split-bf16 operands (precise / mixed handles) only");
  if (workspace_bytes < veto_train_workspace_bytes(h, in->n_obj, in->n_pair)) return fail(VETO_ERR_WORKSPACE, "training workspace too small");
  return VETO_OK;
}

}  // namespace

extern "C" {

size_t veto_train_workspace_bytes(veto_handle_t h, int32_t n_obj, int32_t n_pair) {
  if (!h || n_obj <= 0 || n_pair <= 0) return 0;
  return carve_train(nullptr, h, n_obj, n_pair).total;
}

size_t veto_grad_floats(veto_handle_t h) {
  if (!h || h->params.empty()) return 0;
  const Param& q = h->params.back();
  return q.offset + align_up(q.numel, 64);
}

int veto_weight_offset(veto_handle_t h, int index, size_t* offset_floats) {
  if (!h || index < 0 || index >= (int)h->params.size() || !offset_floats) return fail(VETO_ERR_INVALID, "bad weight index");
  *offset_floats = h->params[index].offset;
  return VETO_OK;
}

int veto_forward_train(veto_handle_t h, void* stream, const veto_inputs_t* in, const veto_train_opts_t* opts, void* workspace,
                       size_t workspace_bytes, float* out_logits) {
  int rc = check_train_inputs(h, in, workspace, workspace_bytes);
  if (rc) return rc;
  if ((rc = check_train_opts(opts))) return rc;
  if (!out_logits) return fail(VETO_ERR_INVALID, "null out_logits");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (h->dirty) { rc = finalize_weights(h, s, true); if (rc) return rc; }
  h->train_gen.erase(workspace);     // (stamped at the end: a failed forward leaves no workspace that veto_backward would accept)
  const int n_obj = in->n_obj, n_pair = in->n_pair, L = h->cfg.layers, H = h->cfg.heads, n_out = h->cfg.num_out;
  const int M = n_pair * kTokens;
  TrainWs ws = carve_train((char*)workspace, h, n_obj, n_pair);
  const std::string T = kT;
  HIP_TRY(launch_pair_indices(in->rel_pairs, in->img_obj_offset, in->img_pair_offset, in->n_img, n_pair, ws.subj, ws.obj, nullptr,
                              nullptr, s));
  {
    ObjPrepArgs a{};
    a.boxes = in->boxes; a.box_mode = in->box_mode; a.labels = in->obj_labels; a.obj_logits = in->obj_logits;
    a.embed = h->p("obj_embed.weight"); a.num_obj_cls = h->cfg.num_obj_cls; a.embed_dim = h->cfg.embed_dim;
    a.bn_w = h->p("pos_embed.0.weight"); a.bn_b = h->p("pos_embed.0.bias");
    HIP_TRY(launch_bn_batch_stats(in->boxes, in->box_mode, n_obj, in->bn_batch_stats, s));
    a.bn_mean = in->bn_batch_stats; a.bn_var = in->bn_batch_stats + 4;
    a.pos_w = h->p("pos_embed.1.weight"); a.pos_b = h->p("pos_embed.1.bias");
    a.loc_wt = h->loc_wt; a.loc_b = h->p("location_projection.0.bias");
    a.cls_wt = h->cls_wt; a.cls_b = h->p("class_projection.0.bias");
    a.lc = ws.lc; a.pos_out = nullptr; a.n_obj = n_obj;
    const DropSite d = drop_site(opts, 1);
    a.drop_seed = d.seed; a.drop_thresh = d.thresh; a.drop_scale = d.scale;
    HIP_TRY(launch_obj_prep(a, s));
  }
  HIP_TRY(launch_patchify(in->roi_depth, in->roi_rgb, ws.pa, n_obj, s));
  rc = run_gemm(h, s, "gemm_patch", ws.pa, h->patch_w, h->patch_bias, nullptr, 0, ws.patch_tab, nullptr, 2 * kDim, n_obj * 16,
                2 * kDim, 2048, EPI_F32);
  if (rc) return rc;
  {
    AssembleArgs a{};
    a.patch_tab = ws.patch_tab; a.lc = ws.lc; a.cls_token = h->p(T + "cls_token");
    a.pos_embedding = h->p(T + "pos_embedding");
    a.ln_w = h->layers[0].ln1_w; a.ln_b = h->layers[0].ln1_b;
    a.subj = ws.subj; a.obj = ws.obj; a.x = ws.layers[0].xin; a.a = ws.layers[0].a1; a.n_pair = n_pair;
    const DropSite d = drop_site(opts, 2);
    a.drop_seed = d.seed; a.drop_thresh = d.thresh; a.drop_scale = d.scale;
    HIP_TRY(launch_assemble(a, s));
  }
  const bool q24 = train_qkv_f24(h);
  const int qepi = q24 ? EPI_F24 : EPI_F32;
  for (int l = 0; l < L; ++l) {
    const LayerW& w = h->layers[l];
    TrainLayer& t = ws.layers[l];
    if (l == L - 1) {
      // Last layer: only x[:, 0] reaches the loss (model_veto.py:23), so -- as at inference -- keys / values cover the 19
      // tokens, everything behind the attention runs on the CLS row of each pair.  The saved activations of this layer
      // (ao, xmid, a2, pre, hid, ws.xout) are COMPACT: row p = pair p.  The backward mirrors this.
      rc = run_gemm(h, s, "gemm_kv_last", t.a1, w.qkv, nullptr, nullptr, 0, q24 ? (float*)((char*)t.qkv + 3 * kDim) : t.qkv + kDim, nullptr,
                    3 * kDim, M, 2 * kDim, kDim, qepi, 0, kDim);
      if (rc) return rc;
      rc = run_gemm(h, s, "gemm_q_cls", t.a1, w.qkv, nullptr, nullptr, 0, t.qkv, nullptr, (long)kTokens * 3 * kDim, n_pair, kDim, kDim,
                    qepi, (long)kTokens * 2 * kDim, 0);
      if (rc) return rc;
      {
        AttnArgs a{};
        a.qkv = t.qkv; a.n_pair = n_pair; a.heads = H; a.cls_only = 1; a.o = t.ao; a.qkv_f24 = q24 ? 1 : 0;
        HIP_TRY(launch_attention(a, s));
      }
      rc = run_gemm(h, s, "gemm_out_cls", t.ao, w.out, w.out_b, t.xin, (long)kTokens * kDim, t.xmid, nullptr, kDim, n_pair, kDim, kDim,
                    EPI_RESID, 0, 0, drop_site(opts, 3 + l));
      if (rc) return rc;
      HIP_TRY(launch_layernorm(t.xmid, kDim, w.ln2_w, w.ln2_b, t.a2, n_pair, s));
      rc = run_gemm(h, s, "gemm_fc1_cls", t.a2, w.fc1, w.fc1_b, nullptr, 0, t.pre, t.hid, 4 * kDim, n_pair, 2 * kDim, kDim, EPI_PRE_GELU);
      if (rc) return rc;
      rc = run_gemm(h, s, "gemm_fc2_cls", t.hid, w.fc2, w.fc2_b, t.xmid, kDim, ws.xout, nullptr, kDim, n_pair, kDim, 2 * kDim, EPI_RESID);
      if (rc) return rc;
      break;
    }
    float* xnext = ws.layers[l + 1].xin;
    rc = run_gemm(h, s, "gemm_qkv", t.a1, w.qkv, nullptr, nullptr, 0, t.qkv, nullptr, 3 * kDim, M, 3 * kDim, kDim, qepi);
    if (rc) return rc;
    {
      AttnArgs a{};
      a.qkv = t.qkv; a.n_pair = n_pair; a.heads = H; a.cls_only = 0; a.o = t.ao; a.qkv_f24 = q24 ? 1 : 0;
      HIP_TRY(launch_attention(a, s));
    }
    rc = run_gemm(h, s, "gemm_out", t.ao, w.out, w.out_b, t.xin, kDim, t.xmid, nullptr, kDim, M, kDim, kDim, EPI_RESID, 0, 0,
                  drop_site(opts, 3 + l));
    if (rc) return rc;
    HIP_TRY(launch_layernorm(t.xmid, kDim, w.ln2_w, w.ln2_b, t.a2, M, s));
    // (the epilogue writes the fp32 pre-activation -- gelu' needs it -- AND its exact-erf GELU as fc2's split rows: round 6; before, a pass
    // of its own read the pre-activation back, 0.55 ms per layer)
    rc = run_gemm(h, s, "gemm_fc1", t.a2, w.fc1, w.fc1_b, nullptr, 0, t.pre, t.hid, 4 * kDim, M, 2 * kDim, kDim, EPI_PRE_GELU);
    if (rc) return rc;
    rc = run_gemm(h, s, "gemm_fc2", t.hid, w.fc2, w.fc2_b, t.xmid, kDim, xnext, nullptr, kDim, M, kDim, 2 * kDim, EPI_RESID);
    if (rc) return rc;
    HIP_TRY(launch_layernorm(xnext, kDim, h->layers[l + 1].ln1_w, h->layers[l + 1].ln1_b, ws.layers[l + 1].a1, M, s));
  }
  HIP_TRY(launch_head(ws.xout, h->head_wt, h->p("rel_out.bias"), out_logits, n_pair, n_out, s, (long)kDim));
  h->stamp_train_workspace(workspace);
  return VETO_OK;
}

int veto_backward(veto_handle_t h, void* stream, const veto_inputs_t* in, const veto_train_opts_t* opts, void* workspace,
                  size_t workspace_bytes, const float* dlogits, float* grads) {
  int rc = check_train_inputs(h, in, workspace, workspace_bytes);
  if (rc) return rc;
  if ((rc = check_train_opts(opts))) return rc;
  if (!dlogits || !grads) return fail(VETO_ERR_INVALID, "null gradient pointer");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipSetDevice(h->cfg.device));
  // the backward reads the derived operands the matching forward used: a weight upload in between would pair this
  // workspace's activations with different weights
  if (h->dirty) return fail(VETO_ERR_WEIGHTS, "weights were reloaded between veto_forward_train and veto_backward");
  {
    auto it = h->train_gen.find(workspace);
    if (it == h->train_gen.end()) return fail(VETO_ERR_INVALID, "veto_backward: this workspace holds no veto_forward_train activations");
    if (it->second != h->weight_gen)
      return fail(VETO_ERR_WEIGHTS, "veto_backward: the weights were refreshed (by a later forward) since this workspace's veto_forward_train");
  }
  const int n_obj = in->n_obj, n_pair = in->n_pair, L = h->cfg.layers, H = h->cfg.heads, n_out = h->cfg.num_out, E = h->cfg.embed_dim;
  const int M = n_pair * kTokens;
  TrainWs ws = carve_train((char*)workspace, h, n_obj, n_pair);
  const std::string T = kT;
  auto G = [&](const std::string& name) { return grads + h->params[h->index.at(name)].offset; };
  HIP_TRY(hipMemsetAsync(grads, 0, veto_grad_floats(h) * 4, s));
  HIP_TRY(hipMemsetAsync(ws.zero, 0, 1024, s));

  // ---- classifier head: gradient of the compact CLS rows ------------------------------------------------------------
  HIP_TRY(launch_head_backward(dlogits, h->p("rel_out.weight"), ws.xout, ws.dx, G("rel_out.weight"), G("rel_out.bias"), ws.head_partial,
                               n_pair, n_out, (long)kDim, s));

  // ---- transformer layers, last to first -------------------------------------------------------------------------
  int dx_presplit = 0;      // > 0: ws.dsplit / ws.colp already hold the split rows / that many rows of column partials of ws.dx
  for (int l = L - 1; l >= 0; --l) {
    const LayerW& w = h->layers[l];
    TrainLayer& t = ws.layers[l];
    const bool last = l == L - 1;
    const int R = last ? n_pair : M;     // rows behind the attention: the CLS rows only in the last layer (compact buffers)
    // x_out = x_mid + gelu(LN2(x_mid) W1^T + b1) W2^T + b2
    // dpre = dh * gelu'(pre): in the epilogue of fc2's input-gradient GEMM (round 6; VETO_TRAIN_GELU_EPI=0: a pass of its own inside the
    // operand preparation of the fc1 backward, rounds 1-5)
    // (the gradient of this layer's output: compact CLS rows from the head in the last layer, else the rows the LayerNorm1 backward of the
    // layer above left -- with their split rows and bias partials when it emitted them)
    const float* dy2 = dx_presplit ? nullptr : ws.dx;
    const bool recompute = train_recompute();
    if (recompute) HIP_TRY(launch_gelu_split(t.pre, t.hid, (size_t)R, 2 * kDim, s));      // gelu(pre): fc2's operand, as the forward's epilogue wrote it
    if (train_gelu_epilogue()) {
      int partials = 0;
      rc = run_linear_backward(h, s, ws, dy2, R, kDim, t.hid, 2 * kDim, h->p(lname(l, "1.fn.net.3.weight")),
                               G(lname(l, "1.fn.net.3.weight")), G(lname(l, "1.fn.net.3.bias")), nullptr, GradXform(), nullptr, dx_presplit, t.pre, &partials);
      if (rc) return rc;
      if (recompute) HIP_TRY(launch_layernorm(t.xmid, kDim, w.ln2_w, w.ln2_b, t.a2, R, s));      // LayerNorm2 rows: fc1's operand
      rc = run_linear_backward(h, s, ws, nullptr, R, 2 * kDim, t.a2, kDim, h->p(lname(l, "1.fn.net.0.weight")),
                               G(lname(l, "1.fn.net.0.weight")), G(lname(l, "1.fn.net.0.bias")), ws.dtmp, GradXform(), ws.dsplit_b, partials);
      if (rc) return rc;
    } else {
    rc = run_linear_backward(h, s, ws, dy2, R, kDim, t.hid, 2 * kDim, h->p(lname(l, "1.fn.net.3.weight")),
                             G(lname(l, "1.fn.net.3.weight")), G(lname(l, "1.fn.net.3.bias")), ws.dbig, GradXform(), nullptr, dx_presplit);
    if (rc) return rc;
    GradXform gelu;              // dpre = dh * gelu'(pre), folded into the operand preparation of the fc1 backward
    gelu.mode = XF_GELU;
    gelu.pre = t.pre;
    if (recompute) HIP_TRY(launch_layernorm(t.xmid, kDim, w.ln2_w, w.ln2_b, t.a2, R, s));
    rc = run_linear_backward(h, s, ws, ws.dbig, R, 2 * kDim, t.a2, kDim, h->p(lname(l, "1.fn.net.0.weight")),
                             G(lname(l, "1.fn.net.0.weight")), G(lname(l, "1.fn.net.0.bias")), ws.dtmp, gelu);
    if (rc) return rc;
    }
    // x_mid = x_in + dropout(attention(LN1(x_in) Wqkv^T) Wo^T + bo): the projection sees the masked gradient
    const DropSite dsite = drop_site(opts, 3 + l);
    const bool ln_split = train_ln_emits_split();
    if (ln_split)
      HIP_TRY(launch_layernorm_backward(t.xmid, ws.dtmp, w.ln2_w, ws.dx, ws.dmid, ws.dgb, ws.ln_partial, R, s, ws.dsplit, ws.colp,
                                        dsite.seed, dsite.thresh, dsite.scale));
    else
      HIP_TRY(launch_layernorm_backward(t.xmid, ws.dtmp, w.ln2_w, ws.dx, ws.dmid, ws.dgb, ws.ln_partial, R, s));
    HIP_TRY(hipMemcpyAsync(G(lname(l, "1.norm.weight")), ws.dgb, kDim * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(G(lname(l, "1.norm.bias")), ws.dgb + kDim, kDim * 4, hipMemcpyDeviceToDevice, s));
    GradXform drop;
    if (dsite.thresh) {
      drop.mode = XF_DROP;
      drop.seed = dsite.seed;
      drop.thresh = dsite.thresh;
      drop.scale = dsite.scale;
    }
    rc = run_linear_backward(h, s, ws, ln_split ? nullptr : ws.dmid, R, kDim, t.ao, kDim, h->p(lname(l, "0.fn.to_out.0.weight")),
                             G(lname(l, "0.fn.to_out.0.weight")), G(lname(l, "0.fn.to_out.0.bias")), ws.dtmp, drop, nullptr,
                             ln_split ? layernorm_backward_col_partials(R) : 0);
    if (rc) return rc;
    HIP_TRY(launch_attention_backward(t.qkv, ws.dtmp, nullptr, ws.dsplit, n_pair, H, last ? 1 : 0, s, train_qkv_f24(h)));
    const float* dres = ws.dmid;
    if (last) {
      // the residual gradient d x_mid lives on the CLS rows only: spread the compact rows over a zeroed token matrix
      HIP_TRY(hipMemsetAsync(ws.dx, 0, (size_t)M * kDim * 4, s));
      HIP_TRY(hipMemcpy2DAsync(ws.dx, (size_t)kTokens * kDim * 4, ws.dmid, (size_t)kDim * 4, (size_t)kDim * 4, (size_t)n_pair,
                               hipMemcpyDeviceToDevice, s));
      dres = ws.dx;
    }
    if (recompute) HIP_TRY(launch_layernorm(t.xin, kDim, w.ln1_w, w.ln1_b, t.a1, M, s));      // LayerNorm1 rows: the QKV projection's operand
    rc = run_linear_backward(h, s, ws, nullptr, M, 3 * kDim, t.a1, kDim, h->p(lname(l, "0.fn.to_qkv.weight")),
                             G(lname(l, "0.fn.to_qkv.weight")), nullptr, ws.dtmp);
    if (rc) return rc;
    // (in the last layer dres == ws.dx is also the output: every element is read and written by the same thread)
    if (ln_split && l > 0) {      // (its result is the gradient matrix of fc2 of the layer below)
      HIP_TRY(launch_layernorm_backward(t.xin, ws.dtmp, w.ln1_w, dres, ws.dx, ws.dgb, ws.ln_partial, M, s, ws.dsplit, ws.colp));
      dx_presplit = layernorm_backward_col_partials(M);
    } else {
      HIP_TRY(launch_layernorm_backward(t.xin, ws.dtmp, w.ln1_w, dres, ws.dx, ws.dgb, ws.ln_partial, M, s));
      dx_presplit = 0;
    }
    HIP_TRY(hipMemcpyAsync(G(lname(l, "0.norm.weight")), ws.dgb, kDim * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(G(lname(l, "0.norm.bias")), ws.dgb + kDim, kDim * 4, hipMemcpyDeviceToDevice, s));
  }

  // ---- token assembly: cls_token, pos_embedding, per-object tables -----------------------------------------------
  // x0[p, t] = token + pos_embedding (ONE 576-vector broadcast over all tokens, model_veto.py:43,62): its gradient is
  // the sum over every token row; the cls_token's is the sum over row 0 of every pair
  {
    const DropSite d = drop_site(opts, 2);   // pos_drop sits between (token + pos_embedding) and the first layer
    if (d.thresh) HIP_TRY(launch_dropout_apply(ws.dx, ws.dx, (size_t)M, kDim, d.seed, d.thresh, d.scale, s));
  }
  HIP_TRY(launch_column_sums(ws.dx, kDim, M, kDim, G(T + "pos_embedding"), ws.col_partial, column_sums_chunks(), s));
  HIP_TRY(launch_column_sums(ws.dx, (long)kTokens * kDim, n_pair, kDim, G(T + "cls_token"), ws.col_partial, column_sums_chunks(), s));
  HIP_TRY(hipMemsetAsync(ws.dpatch, 0, (size_t)n_obj * 16 * 2 * kDim * 4, s));
  HIP_TRY(hipMemsetAsync(ws.dlc, 0, (size_t)n_obj * 2 * 2 * kDim * 4, s));
  HIP_TRY(launch_assemble_backward(ws.dx, ws.subj, ws.obj, ws.lc, ws.dpatch, ws.dlc, n_pair, s));

  // ---- patch projection: patch_tab = PA . Wcat^T + bias_cat (bias only on the subject half) ------------------------
  {
    const std::string pe = T + "patch_embed.";
    const int R = n_obj * 16;
    // dWcat^T [2048, 1152] = PA^T . dpatch: "dy" = PA (split rows), "x" = dpatch (fp32) in run_wgrad's roles swapped,
    // so that the output width (1152) is a multiple of 192
    {
      const int N = 2048, K = 2 * kDim;
      const int out_tiles = ((N + 255) / 256) * (K / 192);
      int ks = 2 * 256 / out_tiles;
      const int max_ks = (R + 32 * 8 - 1) / (32 * 8);
      if (ks > max_ks) ks = max_ks;
      if (ks < 1) ks = 1;
      const size_t mp = ((size_t)R + 32 * (size_t)ks - 1) / (32 * (size_t)ks) * 32 * (size_t)ks;
      HIP_TRY(launch_split_rows(ws.dpatch, ws.dsplit, (size_t)R, K, s));
      HIP_TRY(hipMemsetAsync(ws.dwcat_t, 0, (size_t)N * K * 4, s));
      GemmArgs g{};
      g.a = ws.pa; g.w = ws.dsplit; g.c = ws.dwcat_t;
      g.M = N; g.N = K; g.K = (int)mp; g.ldc = K; g.k_splits = ks;
      g.tn = 1; g.lda = 2 * (long)N; g.ldw = 2 * (long)K; g.k_valid = R; g.zero = ws.zero;
      HIP_TRY(launch_gemm_split(g, EPI_ATOMIC, 0, s));
    }
    HIP_TRY(launch_patch_weight_grad(ws.dwcat_t, G(pe + "proj_d.weight"), G(pe + "proj_v.weight"), s));
    // biases: column sums of the subject half of dpatch: columns 0..511 -> proj_d.bias, 512..575 -> proj_v.bias
    HIP_TRY(launch_column_sums(ws.dpatch, 2 * kDim, R, kDim, ws.dgb, ws.col_partial, column_sums_chunks(), s));
    HIP_TRY(hipMemcpyAsync(G(pe + "proj_d.bias"), ws.dgb, 512 * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(G(pe + "proj_v.bias"), ws.dgb + 512, 64 * 4, hipMemcpyDeviceToDevice, s));
    // input gradient (optional): dPA = dpatch . Wcat, i.e. the forward GEMM kernel on split(dpatch) (still in ws.dsplit) and
    // Wcat^T as the weight operand; then the inverse of patchify onto the ROI maps.  The reference's depth backbone is
    // trained through roi_depth_features (tools/relation_train_net.py:166-170).
    if (opts && (opts->d_roi_rgb || opts->d_roi_depth)) {
      HIP_TRY(launch_build_patch_weight_t(h->p(pe + "proj_d.weight"), h->p(pe + "proj_v.weight"), ws.wcat_t, s));
      GemmArgs g{};
      g.a = ws.dsplit; g.w = ws.wcat_t; g.c = ws.dpa;
      g.M = R; g.N = kPatchTRows; g.K = 2 * kDim; g.ldc = kPatchTRows;
      HIP_TRY(launch_gemm_split(g, EPI_F32, 0, s));
      HIP_TRY(launch_unpatchify(ws.dpa, kPatchTRows, opts->d_roi_depth, opts->d_roi_rgb, n_obj, s));
    }
  }

  // ---- location / class projections and what feeds them -----------------------------------------------------------
  {
    const long ldlc = 2 * 2 * kDim;   // dlc row n = [location 1152 | class 1152]
    // biases sit on the subject half
    HIP_TRY(launch_column_sums(ws.dlc, ldlc, n_obj, kDim, G("location_projection.0.bias"), ws.col_partial, column_sums_chunks(), s));
    HIP_TRY(launch_column_sums(ws.dlc + 2 * kDim, ldlc, n_obj, kDim, G("class_projection.0.bias"), ws.col_partial, column_sums_chunks(), s));
    // position branch: recompute pos (post-ReLU) through the masked gradient path
    //   dpos = dlc_loc . loc_wt^T ; obj_pos_backward -> dpre (ReLU'), BatchNorm affine gradients
    HIP_TRY(launch_sgemm_nt(ws.dlc, ldlc, h->loc_wt, 2 * kDim, ws.dpos, kPosDim, n_obj, kPosDim, 2 * kDim, s));
    const DropSite dpos_site = drop_site(opts, 1);   // Dropout(0.1) behind the ReLU of pos_embed
    if (dpos_site.thresh) HIP_TRY(launch_dropout_apply(ws.dpos, ws.dpos, (size_t)n_obj, kPosDim, dpos_site.seed, dpos_site.thresh, dpos_site.scale, s));
    HIP_TRY(launch_obj_pos_backward(in->boxes, in->box_mode, in->bn_batch_stats, h->p("pos_embed.0.weight"), h->p("pos_embed.0.bias"),
                                    h->p("pos_embed.1.weight"), h->p("pos_embed.1.bias"), ws.dpos, ws.dpre, ws.xhat, ws.bn_out, ws.dbn_out,
                                    G("pos_embed.0.weight"), G("pos_embed.0.bias"), n_obj, s));
    // pos_embed.1: Linear(4, 128): dW[k, c] = sum_n dpre[n, k] bn_out[n, c]; db = column sums of dpre
    HIP_TRY(launch_sgemm_tn(ws.dpre, kPosDim, ws.bn_out, 4, G("pos_embed.1.weight"), 4, n_obj, kPosDim, 4, s));
    HIP_TRY(launch_column_sums(ws.dpre, kPosDim, n_obj, kPosDim, G("pos_embed.1.bias"), ws.col_partial, column_sums_chunks(), s));
    // location_projection weight: d loc_wt[k, j] = sum_n pos[n, k] dlc_loc[n, j], pos = relu(pre): recompute pos = the
    // forward's value; it equals (dpre != 0 ? ... ) no -- recompute it from bn_out
    // pos[n, k] = relu(pos_b[k] + sum_c pos_w[k, c] bn_out[n, c]): one small product + ReLU, done by reusing dpos as storage
    HIP_TRY(launch_sgemm_nt(ws.bn_out, 4, h->p("pos_embed.1.weight"), 4, ws.dpos, kPosDim, n_obj, kPosDim, 4, s));
    HIP_TRY(launch_bias_relu(ws.dpos, h->p("pos_embed.1.bias"), n_obj, kPosDim, s));
    if (dpos_site.thresh) HIP_TRY(launch_dropout_apply(ws.dpos, ws.dpos, (size_t)n_obj, kPosDim, dpos_site.seed, dpos_site.thresh, dpos_site.scale, s));
    HIP_TRY(launch_sgemm_tn(ws.dpos, kPosDim, ws.dlc, ldlc, ws.dloc_wt, 2 * kDim, n_obj, kPosDim, 2 * kDim, s));
    HIP_TRY(launch_untranspose_pair_proj(ws.dloc_wt, G("location_projection.0.weight"), kPosDim, s));
    // class branch: emb = E[label] (hard labels) or softmax(logits) . E (sgcls, :4092-4095);
    // d cls_wt[k, j] = sum_n emb[n, k] dlc_cls[n, j]; demb = dlc_cls . cls_wt^T
    const int C_obj = h->cfg.num_obj_cls;
    if (in->obj_logits) {
      HIP_TRY(launch_softmax_rows(in->obj_logits, ws.prob, n_obj, C_obj, s));
      HIP_TRY(launch_sgemm_nn(ws.prob, C_obj, h->p("obj_embed.weight"), E, ws.emb, E, n_obj, E, C_obj, s));
    } else {
      HIP_TRY(launch_gather_rows(h->p("obj_embed.weight"), in->obj_labels, E, ws.emb, n_obj, s));
    }
    HIP_TRY(launch_sgemm_tn(ws.emb, E, ws.dlc + 2 * kDim, ldlc, ws.dcls_wt, 2 * kDim, n_obj, E, 2 * kDim, s));
    HIP_TRY(launch_untranspose_pair_proj(ws.dcls_wt, G("class_projection.0.weight"), E, s));
    HIP_TRY(launch_sgemm_nt(ws.dlc + 2 * kDim, ldlc, h->cls_wt, 2 * kDim, ws.demb, E, n_obj, E, 2 * kDim, s));
    if (in->obj_logits) HIP_TRY(launch_sgemm_tn(ws.prob, C_obj, ws.demb, E, G("obj_embed.weight"), E, n_obj, C_obj, E, s));
    else HIP_TRY(launch_scatter_rows(ws.demb, in->obj_labels, E, G("obj_embed.weight"), n_obj, s));
  }
  return VETO_OK;
}

int veto_enumerate_pairs(void* stream, int32_t n, int64_t* out) {
  if (n < 0 || !out) return fail(VETO_ERR_INVALID, "bad argument");
  HIP_TRY(launch_enumerate_pairs(n, out, (hipStream_t)stream));
  return VETO_OK;
}

size_t veto_postprocess_workspace_bytes(int32_t n_pair, int32_t n_rel_cls) {
  if (n_pair <= 0 || n_rel_cls <= 0) return 0;
  return align_up((size_t)n_pair * n_rel_cls * 4, 256) + 3 * align_up((size_t)n_pair * 4, 256);
}

int veto_postprocess(void* stream, const veto_post_args_t* a, void* workspace, size_t workspace_bytes) {
  if (!a || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (a->struct_size != (int32_t)sizeof(veto_post_args_t)) return fail(VETO_ERR_INVALID, "veto_post_args_t size mismatch");
  if (a->n_img <= 0 || a->n_obj <= 0 || a->n_pair <= 0 || a->n_rel_cls < 2 || a->n_obj_cls < 2)
    return fail(VETO_ERR_INVALID, "bad sizes");
  if (!a->rel_logits || !a->obj_logits || !a->rel_pairs || !a->img_obj_offset || !a->img_pair_offset || !a->obj_scores ||
      !a->obj_pred || !a->rel_prob_sorted || !a->rel_pairs_sorted || !a->rel_labels_sorted)
    return fail(VETO_ERR_INVALID, "missing pointer");
  if (a->max_pairs_per_image < 1 || a->max_pairs_per_image > postprocess_max_pairs_per_image())
    return fail(VETO_ERR_INVALID, "max_pairs_per_image %d outside 1..%d (MAX_PROPOSAL_PAIR is 2048 at test time)",
                a->max_pairs_per_image, postprocess_max_pairs_per_image());
  if (workspace_bytes < veto_postprocess_workspace_bytes(a->n_pair, a->n_rel_cls)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  char* base = (char*)workspace;
  PostArgs p{};
  p.rel_logits = a->rel_logits; p.obj_logits = a->obj_logits; p.rel_pairs = a->rel_pairs;
  p.img_obj_off = a->img_obj_offset; p.img_pair_off = a->img_pair_offset;
  p.n_img = a->n_img; p.n_obj = a->n_obj; p.n_pair = a->n_pair; p.n_rel_cls = a->n_rel_cls; p.n_obj_cls = a->n_obj_cls;
  p.obj_scores = a->obj_scores; p.obj_pred = a->obj_pred; p.out_prob = a->rel_prob_sorted;
  p.out_pairs = a->rel_pairs_sorted; p.out_labels = a->rel_labels_sorted; p.out_triple = a->triple_sorted;
  p.prob_tmp = (float*)base;
  base += align_up((size_t)a->n_pair * a->n_rel_cls * 4, 256);
  p.triple = (float*)base; base += align_up((size_t)a->n_pair * 4, 256);
  p.label_tmp = (int32_t*)base; base += align_up((size_t)a->n_pair * 4, 256);
  p.perm = (int32_t*)base;
  HIP_TRY(launch_postprocess(p, (hipStream_t)stream));
  return VETO_OK;
}

int veto_postprocess_meet(void* stream, const veto_post_meet_args_t* a, void* workspace, size_t workspace_bytes) {
  if (!a || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (a->struct_size != (int32_t)sizeof(veto_post_meet_args_t)) return fail(VETO_ERR_INVALID, "veto_post_meet_args_t size mismatch");
  if (a->n_obj <= 0 || a->n_pair <= 0 || a->n_groups <= 0 || a->n_groups > 16 || a->n_rel_cls < 2 || a->n_obj_cls < 2)
    return fail(VETO_ERR_INVALID, "bad sizes");
  if (!a->group_logits || !a->group_widths || !a->incre_idx_list || !a->obj_logits || !a->rel_pairs || !a->obj_scores ||
      !a->obj_pred || !a->rel_prob_sorted || !a->rel_pairs_sorted || !a->rel_labels_sorted)
    return fail(VETO_ERR_INVALID, "missing pointer");
  const long total = (long)a->n_groups * a->n_pair;
  if (total > postprocess_max_pairs_per_image())
    return fail(VETO_ERR_INVALID, "n_groups * n_pair = %ld exceeds %d", total, postprocess_max_pairs_per_image());
  if (workspace_bytes < veto_postprocess_workspace_bytes((int32_t)total, a->n_rel_cls)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  std::vector<MeetGroup> groups(a->n_groups);
  for (int k = 0; k < a->n_groups; ++k) {
    MeetGroup& g = groups[k];
    g.logits = a->group_logits[k];
    g.width = a->group_widths[k];
    g.row0 = k * a->n_pair;
    if (!g.logits || g.width < 3 || g.width - 1 > 104) return fail(VETO_ERR_INVALID, "bad group %d (width %d)", k, g.width);
    int n = 0;
    g.cols[n++] = 0;
    for (int c = 0; c < a->n_rel_cls; ++c)
      if (a->incre_idx_list[c] == k + 1) {
        if (n >= g.width - 1) return fail(VETO_ERR_INVALID, "group %d has more classes than its head is wide", k);
        g.cols[n++] = c;
      }
    if (n != g.width - 1) return fail(VETO_ERR_INVALID, "group %d: %d classes but head width %d", k, n - 1, g.width);
  }
  char* base = (char*)workspace;
  PostArgs p{};
  p.obj_logits = a->obj_logits; p.rel_pairs = a->rel_pairs;
  p.n_img = 1; p.n_obj = a->n_obj; p.n_pair = a->n_pair; p.n_rel_cls = a->n_rel_cls; p.n_obj_cls = a->n_obj_cls;
  p.obj_scores = a->obj_scores; p.obj_pred = a->obj_pred; p.out_prob = a->rel_prob_sorted;
  p.out_pairs = a->rel_pairs_sorted; p.out_labels = a->rel_labels_sorted; p.out_triple = a->triple_sorted;
  p.prob_tmp = (float*)base;
  base += align_up((size_t)total * a->n_rel_cls * 4, 256);
  p.triple = (float*)base; base += align_up((size_t)total * 4, 256);
  p.label_tmp = (int32_t*)base; base += align_up((size_t)total * 4, 256);
  p.perm = (int32_t*)base;
  HIP_TRY(launch_postprocess_meet(p, groups.data(), a->n_groups, (hipStream_t)stream));
  return VETO_OK;
}

int veto_postprocess_vote(void* stream, const veto_post_vote_args_t* a, void* workspace, size_t workspace_bytes) {
  if (!a || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (a->struct_size != (int32_t)sizeof(veto_post_vote_args_t)) return fail(VETO_ERR_INVALID, "veto_post_vote_args_t size mismatch");
  if (a->n_obj <= 0 || a->n_pair <= 0 || a->n_groups <= 0 || a->n_groups > 16 || a->n_rel_cls < 2 || a->n_obj_cls < 2)
    return fail(VETO_ERR_INVALID, "bad sizes");
  if (a->voting != 0 && a->voting != 1) return fail(VETO_ERR_INVALID, "voting must be 0 ('C') or 1 ('U')");
  if (!a->expert_logits || !a->group_widths || !a->incre_idx_list || !a->obj_logits || !a->rel_pairs || !a->obj_scores ||
      !a->obj_pred || !a->rel_prob_sorted || !a->rel_pairs_sorted || !a->rel_labels_sorted || !a->kept_count)
    return fail(VETO_ERR_INVALID, "missing pointer");
  const long total = (long)a->n_groups * a->n_pair;
  if (total > postprocess_max_pairs_per_image())
    return fail(VETO_ERR_INVALID, "n_groups * n_pair = %ld exceeds %d", total, postprocess_max_pairs_per_image());
  if (workspace_bytes < veto_postprocess_workspace_bytes((int32_t)total, a->n_rel_cls)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  std::vector<VoteGroup> groups(a->n_groups);
  for (int k = 0; k < a->n_groups; ++k) {
    VoteGroup& g = groups[k];
    for (int e = 0; e < 3; ++e) {
      g.logits[e] = a->expert_logits[3 * k + e];
      if (!g.logits[e]) return fail(VETO_ERR_INVALID, "group %d expert %d: null logits", k, e + 1);
    }
    g.width = a->group_widths[k];
    g.row0 = k * a->n_pair;
    if (g.width < 3 || g.width - 1 > 104) return fail(VETO_ERR_INVALID, "bad group %d (width %d)", k, g.width);
    int n = 0;
    g.cols[n++] = 0;
    for (int c = 0; c < a->n_rel_cls; ++c)
      if (a->incre_idx_list[c] == k + 1) {
        if (n >= g.width - 1) return fail(VETO_ERR_INVALID, "group %d has more classes than its head is wide", k);
        g.cols[n++] = c;
      }
    if (n != g.width - 1) return fail(VETO_ERR_INVALID, "group %d: %d classes but head width %d", k, n - 1, g.width);
  }
  char* base = (char*)workspace;
  PostArgs p{};
  p.obj_logits = a->obj_logits; p.rel_pairs = a->rel_pairs;
  p.n_img = 1; p.n_obj = a->n_obj; p.n_pair = a->n_pair; p.n_rel_cls = a->n_rel_cls; p.n_obj_cls = a->n_obj_cls;
  p.obj_scores = a->obj_scores; p.obj_pred = a->obj_pred; p.out_prob = a->rel_prob_sorted;
  p.out_pairs = a->rel_pairs_sorted; p.out_labels = a->rel_labels_sorted; p.out_triple = a->triple_sorted;
  p.kept_count = a->kept_count;
  p.prob_tmp = (float*)base;
  base += align_up((size_t)total * a->n_rel_cls * 4, 256);
  p.triple = (float*)base; base += align_up((size_t)total * 4, 256);
  p.label_tmp = (int32_t*)base; base += align_up((size_t)total * 4, 256);
  p.perm = (int32_t*)base;
  HIP_TRY(launch_postprocess_vote(p, groups.data(), a->n_groups, a->voting, (hipStream_t)stream));
  return VETO_OK;
}

// shared argument check / conversion of veto_roi_pool and veto_roi_pool_backward
static int roi_pool_args(const veto_roi_pool_args_t* a, bool forward, RoiPoolArgs* out) {
  if (!a) return fail(VETO_ERR_INVALID, "null argument");
  if (a->struct_size != (int32_t)sizeof(veto_roi_pool_args_t)) return fail(VETO_ERR_INVALID, "veto_roi_pool_args_t size mismatch");
  if (a->n_levels < 1 || a->n_levels > 4) return fail(VETO_ERR_INVALID, "n_levels must be 1..4, got %d", a->n_levels);
  if (a->n_img <= 0 || a->n_roi <= 0 || a->channels <= 0) return fail(VETO_ERR_INVALID, "bad sizes (n_img %d, n_roi %d, channels %d)", a->n_img, a->n_roi, a->channels);
  if (a->pooled < 1 || a->pooled > 8) return fail(VETO_ERR_INVALID, "pooled must be 1..8, got %d", a->pooled);
  if (a->sampling_ratio < 1 || a->sampling_ratio > 4)
    return fail(VETO_ERR_INVALID, "sampling_ratio must be 1..4 (adaptive sampling is not built), got %d", a->sampling_ratio);
  if (!a->rois || (forward && !a->out_rgb)) return fail(VETO_ERR_INVALID, "missing pointer");
  const bool has_depth = forward ? a->depth_feat != nullptr : a->depth_h > 0;
  if (has_depth && ((forward && !a->out_depth) || a->depth_channels <= 0 || a->depth_h <= 0 || a->depth_w <= 0))
    return fail(VETO_ERR_INVALID, "depth map given without out_depth / sizes");
  RoiPoolArgs p{};
  for (int l = 0; l < a->n_levels; ++l) {
    if ((forward && !a->level_feat[l]) || a->level_h[l] <= 0 || a->level_w[l] <= 0 || !(a->level_scale[l] > 0.f))
      return fail(VETO_ERR_INVALID, "bad pyramid level %d", l);
    p.lv[l] = RoiLevel{a->level_feat[l], a->level_h[l], a->level_w[l], a->level_scale[l]};
  }
  p.n_levels = a->n_levels;
  // poolers.py:86-88: the level range follows from the first and last scale
  p.k_min = (int)lroundf(-log2f(a->level_scale[0]));
  p.k_max = (int)lroundf(-log2f(a->level_scale[a->n_levels - 1]));
  if (has_depth) {
    const int dl = a->n_levels > 1 ? 2 : 0;  // poolers.py:146-149
    if (dl >= a->n_levels) return fail(VETO_ERR_INVALID, "the depth pooler is level 2; need at least 3 levels or exactly 1");
    p.depth = RoiLevel{a->depth_feat, a->depth_h, a->depth_w, a->level_scale[dl]};
  }
  p.n_roi = a->n_roi; p.channels = a->channels; p.depth_channels = has_depth ? a->depth_channels : 0;
  p.pooled = a->pooled; p.sampling_ratio = a->sampling_ratio;
  p.rois = a->rois; p.out_rgb = a->out_rgb; p.out_depth = a->out_depth; p.out_levels = a->out_levels;
  *out = p;
  return VETO_OK;
}

int veto_roi_pool(void* stream, const veto_roi_pool_args_t* a) {
  RoiPoolArgs p{};
  const int rc = roi_pool_args(a, true, &p);
  if (rc != VETO_OK) return rc;
  HIP_TRY(launch_roi_pool(p, (hipStream_t)stream));
  return VETO_OK;
}

int veto_roi_pool_backward(void* stream, const veto_roi_pool_args_t* a, const float* grad_rgb, const float* grad_depth,
                           float* const* level_grad, float* depth_grad) {
  RoiPoolArgs p{};
  const int rc = roi_pool_args(a, false, &p);
  if (rc != VETO_OK) return rc;
  if (!grad_rgb || !level_grad) return fail(VETO_ERR_INVALID, "missing gradient pointer");
  for (int l = 0; l < p.n_levels; ++l) {
    if (!level_grad[l]) return fail(VETO_ERR_INVALID, "level_grad[%d] is null", l);
    p.lv_grad[l] = level_grad[l];
  }
  if ((grad_depth != nullptr) != (depth_grad != nullptr)) return fail(VETO_ERR_INVALID, "grad_depth and depth_grad go together");
  if (grad_depth && p.depth.H <= 0) return fail(VETO_ERR_INVALID, "depth gradient without depth sizes in args");
  if (grad_depth) p.depth.feat = grad_depth;   // only its non-null-ness and the sizes are used
  else p.depth.feat = nullptr;
  p.gout_rgb = grad_rgb; p.gout_depth = grad_depth; p.depth_grad = depth_grad;
  HIP_TRY(launch_roi_pool_backward(p, (hipStream_t)stream));
  return VETO_OK;
}

size_t veto_sgg_eval_workspace_bytes(int32_t n_img, int32_t n_pair_total, int32_t n_gt_total, int32_t n_rel_cls) {
  if (n_img <= 0 || n_pair_total < 0 || n_gt_total < 0 || n_rel_cls < 2) return 0;
  return 5 * align_up((size_t)n_pair_total * 4 + 4, 256) + align_up((size_t)n_gt_total * 4 + 4, 256) +
         align_up((size_t)n_img * 7 * n_rel_cls * 4, 256);
}

int veto_sgg_eval(void* stream, const veto_sgg_eval_args_t* a, int32_t n_pair_total, int32_t n_gt_total, void* workspace,
                  size_t workspace_bytes) {
  if (!a || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (a->struct_size != (int32_t)sizeof(veto_sgg_eval_args_t)) return fail(VETO_ERR_INVALID, "veto_sgg_eval_args_t size mismatch");
  if (a->n_img <= 0 || a->n_rel_cls < 2 || a->n_rel_cls > 4096 || a->n_zeroshot < 0 || n_pair_total < 0 || n_gt_total < 0)
    return fail(VETO_ERR_INVALID, "bad sizes");
  if (!(a->iou_thres >= 0.f && a->iou_thres <= 1.f)) return fail(VETO_ERR_INVALID, "iou_thres must be in [0, 1]");
  if (!a->gt_offset || !a->obj_offset || !a->pair_offset || !a->gt_rels || !a->gt_classes || !a->gt_boxes || !a->pred_pairs ||
      !a->rel_scores || !a->pred_classes || !a->pred_boxes || !a->obj_scores || (a->n_zeroshot > 0 && !a->zeroshot) ||
      !a->gc_rank || !a->ng_rank || !a->acc_rank || !a->zeroshot_flag || !a->ng_rows || !a->ng_cols || !a->ng_count || !a->metrics)
    return fail(VETO_ERR_INVALID, "missing pointer");
  if (workspace_bytes < veto_sgg_eval_workspace_bytes(a->n_img, n_pair_total, n_gt_total, a->n_rel_cls))
    return fail(VETO_ERR_WORKSPACE, "workspace too small");
  SggEvalArgs p{};
  p.n_img = a->n_img; p.n_rel_cls = a->n_rel_cls; p.n_zeroshot = a->n_zeroshot; p.iou_thres = a->iou_thres;
  p.gt_off = a->gt_offset; p.obj_off = a->obj_offset; p.pair_off = a->pair_offset;
  p.gt_rels = a->gt_rels; p.gt_classes = a->gt_classes; p.gt_boxes = a->gt_boxes;
  p.pred_pairs = a->pred_pairs; p.rel_scores = a->rel_scores; p.pred_classes = a->pred_classes;
  p.pred_boxes = a->pred_boxes; p.obj_scores = a->obj_scores; p.zeroshot = a->zeroshot;
  p.gc_rank = a->gc_rank; p.ng_rank = a->ng_rank; p.acc_rank = a->acc_rank; p.zeroshot_flag = a->zeroshot_flag;
  p.ng_rows = a->ng_rows; p.ng_cols = a->ng_cols; p.ng_count = a->ng_count; p.metrics = a->metrics;
  char* base = (char*)workspace;
  const size_t per_pair = align_up((size_t)n_pair_total * 4 + 4, 256);
  p.label_tmp = (int32_t*)base; base += per_pair;
  p.flag_tmp = (int32_t*)base; base += per_pair;
  p.flag_before = (int32_t*)base; base += per_pair;
  p.pair_score = (float*)base; base += per_pair;
  p.row_key = (uint32_t*)base; base += per_pair;
  p.acc_first = (int32_t*)base; base += align_up((size_t)n_gt_total * 4 + 4, 256);
  p.cls_table = (int32_t*)base;
  HIP_TRY(launch_sgg_eval(p, (hipStream_t)stream));
  return VETO_OK;
}

int veto_profile_enable(veto_handle_t h, int32_t on) {
  if (!h) return fail(VETO_ERR_INVALID, "null handle");
  h->prof_on = on != 0;
  return VETO_OK;
}

int veto_profile_collect(veto_handle_t h) {
  if (!h) return fail(VETO_ERR_INVALID, "null handle");
  for (ProfRec& r : h->prof_recs) {
    HIP_TRY(hipEventSynchronize(r.stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.start, r.stop));
    auto& a = h->prof_agg[r.name_id];
    a.ms += ms;
    a.n += 1;
    a.flops = r.flops;
    a.bytes = r.bytes;
    h->event_pool.push_back(r.start);
    h->event_pool.push_back(r.stop);
  }
  h->prof_recs.clear();
  return (int)h->prof_names.size();
}

int veto_profile_entry(veto_handle_t h, int index, const char** name, double* total_ms, int64_t* launches,
                       double* flops_per_launch, double* bytes_per_launch) {
  if (!h || index < 0 || index >= (int)h->prof_names.size()) return fail(VETO_ERR_INVALID, "bad profile index");
  if (name) *name = h->prof_names[index].c_str();
  if (total_ms) *total_ms = h->prof_agg[index].ms;
  if (launches) *launches = h->prof_agg[index].n;
  if (flops_per_launch) *flops_per_launch = h->prof_agg[index].flops;
  if (bytes_per_launch) *bytes_per_launch = h->prof_agg[index].bytes;
  return VETO_OK;
}

int veto_profile_reset(veto_handle_t h) {
  if (!h) return fail(VETO_ERR_INVALID, "null handle");
  for (ProfRec& r : h->prof_recs) { h->event_pool.push_back(r.start); h->event_pool.push_back(r.stop); }
  h->prof_recs.clear();
  for (auto& a : h->prof_agg) a = veto_handle_s::Agg();
  return VETO_OK;
}

size_t veto_debug_gemm_workspace_bytes(int32_t m, int32_t n, int32_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const size_t mp = (size_t)gemm_rows_padded(m);
  return align_up(mp * k * 4, 256) + align_up((size_t)n * k * 4, 256) + 256;
}

int veto_debug_gemm(void* stream, const float* a, const float* w, const float* bias, float* c, int32_t m, int32_t n,
                    int32_t k, int32_t precision, void* workspace, size_t workspace_bytes) {
  if (!a || !w || !c || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (workspace_bytes < veto_debug_gemm_workspace_bytes(m, n, k)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t mp = (size_t)gemm_rows_padded(m);
  char* base = (char*)workspace;
  __bf16* a_s = (__bf16*)base;
  __bf16* w_s = (__bf16*)(base + align_up(mp * k * 4, 256));
  int* w_exp = (int*)(base + align_up(mp * k * 4, 256) + align_up((size_t)n * k * 4, 256));
  HIP_TRY(hipMemsetAsync(base, 0, align_up(mp * k * 4, 256), s));
  GemmArgs g{};
  if (precision == VETO_MIXED) {
    if (k % 64 != 0) return fail(VETO_ERR_INVALID, "mixed rows need K %% 64 == 0");
    HIP_TRY(launch_mixed_act_rows(a, a_s, (size_t)m, k, s));
    HIP_TRY(launch_mixed_weight_rows(w, w_s, (size_t)n, k, w_exp, s));
    g.fmt = FMT_MIXED; g.w_exp = w_exp;
  } else {
    HIP_TRY(launch_split_rows(a, a_s, (size_t)m, k, s));
    HIP_TRY(launch_split_rows(w, w_s, (size_t)n, k, s));
  }
  g.a = a_s; g.w = w_s; g.bias = bias; g.c = c;
  g.M = m; g.N = n; g.K = k; g.ldc = n;
  hipError_t e = launch_gemm_split(g, EPI_F32, precision == VETO_FAST ? 1 : 0, s);
  if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? VETO_ERR_INVALID : VETO_ERR_HIP,
                                   "gemm launch failed (N must be a multiple of 192, K of 32): %s", hipGetErrorString(e));
  return VETO_OK;
}

// Test hook of the round-3 forms of the split-row GEMM: block-diagonal weights (kb_tiles > 0: column tile n multiplies only the
// k-steps [(n / kb_tiles) * kb_steps, + kb_steps) -- everything outside those blocks of w is ignored) and the output forms:
// out_form 0 = fp32 rows [m, n], 1 = split rows (m x 2n bf16: per 32 columns 32 hi then 32 lo), 2 = 3-byte floats (m x 3n bytes).
// Workspace as for veto_debug_gemm.
int veto_debug_gemm_forms(void* stream, const float* a, const float* w, void* c, int32_t m, int32_t n, int32_t k, int32_t kb_tiles,
                          int32_t kb_steps, int32_t out_form, void* workspace, size_t workspace_bytes) {
  if (!a || !w || !c || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (out_form < 0 || out_form > 2) return fail(VETO_ERR_INVALID, "out_form must be 0, 1 or 2");
  if (workspace_bytes < veto_debug_gemm_workspace_bytes(m, n, k)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t mp = (size_t)gemm_rows_padded(m);
  char* base = (char*)workspace;
  __bf16* a_s = (__bf16*)base;
  __bf16* w_s = (__bf16*)(base + align_up(mp * k * 4, 256));
  HIP_TRY(hipMemsetAsync(base, 0, align_up(mp * k * 4, 256), s));
  HIP_TRY(launch_split_rows(a, a_s, (size_t)m, k, s));
  HIP_TRY(launch_split_rows(w, w_s, (size_t)n, k, s));
  GemmArgs g{};
  g.a = a_s; g.w = w_s; g.M = m; g.N = n; g.K = k;
  g.kb_tiles = kb_tiles; g.kb_steps = kb_steps;
  if (out_form == 1) { g.c_split = (__bf16*)c; g.ldc = 2L * n; }
  else { g.c = (float*)c; g.ldc = n; }
  hipError_t e = launch_gemm_split(g, out_form == 1 ? EPI_SPLIT : out_form == 2 ? EPI_F24 : EPI_F32, 0, s);
  if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? VETO_ERR_INVALID : VETO_ERR_HIP,
                                   "gemm launch failed (N a multiple of 192, K of 32, the blocks must tile K): %s", hipGetErrorString(e));
  return VETO_OK;
}

// Test / measurement hook of the FeedForward block (model_veto.py:137-143 + the residual of :21) on VETO_MIXED operands:
// x <- x + W2 . gelu(W1 . a + b1) + b2 for m token rows.  mode 0 = the two GEMM launches (fc1 with the GELU epilogue writing
// the hidden activation as mixed rows, fc2 with the residual epilogue), mode 1 = the fused kernel (ffn_fused.hip).
// flags & 1: (re)build the mixed operands from a / w1 / w2 first.  The block runs `reps` times (x accumulates: timing only when
// reps > 1); *ms_per_rep (host, optional) receives the mean device time of one run from hipEvents on `stream`.
size_t veto_debug_ffn_workspace_bytes(int32_t m) {
  if (m <= 0) return 0;
  const size_t mp = (size_t)gemm_rows_padded(m);
  return align_up(mp * kDim * 4, 256) + align_up(mp * 2 * kDim * 4, 256) + 2 * align_up((size_t)2 * kDim * kDim * 4, 256) + 256;
}

int veto_debug_ffn(void* stream, const float* a, const float* w1, const float* b1, const float* w2, const float* b2, float* x,
                   int32_t m, int32_t mode, int32_t flags, int32_t reps, float* ms_per_rep, void* workspace, size_t workspace_bytes,
                   const float* ln_w, const float* ln_b, void* ln_rows) {
  if (!a || !w1 || !b1 || !w2 || !b2 || !x || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (m <= 0 || reps <= 0 || (mode != 0 && mode != 1)) return fail(VETO_ERR_INVALID, "bad m / reps / mode");
  if (ln_rows && (!ln_w || !ln_b)) return fail(VETO_ERR_INVALID, "ln_rows needs ln_w and ln_b");
  if (workspace_bytes < veto_debug_ffn_workspace_bytes(m)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t mp = (size_t)gemm_rows_padded(m);
  char* base = (char*)workspace;
  __bf16* a_m = (__bf16*)base;
  __bf16* hid = (__bf16*)(base + align_up(mp * kDim * 4, 256));
  __bf16* w1_m = (__bf16*)((char*)hid + align_up(mp * 2 * kDim * 4, 256));
  __bf16* w2_m = (__bf16*)((char*)w1_m + align_up((size_t)2 * kDim * kDim * 4, 256));
  int* exps = (int*)((char*)w2_m + align_up((size_t)2 * kDim * kDim * 4, 256));
  if (flags & 1) {
    HIP_TRY(hipMemsetAsync(a_m, 0, mp * kDim * 4, s));
    HIP_TRY(launch_mixed_act_rows(a, a_m, (size_t)m, kDim, s));
    HIP_TRY(launch_mixed_weight_rows(w1, w1_m, (size_t)2 * kDim, kDim, exps + 0, s));
    HIP_TRY(launch_mixed_weight_rows(w2, w2_m, (size_t)kDim, 2 * kDim, exps + 1, s));
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ms_per_rep) {
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, s));
  }
  for (int r = 0; r < reps; ++r) {
    if (mode == 1) {
      FfnArgs f{};
      f.a = (const char*)a_m; f.w1 = (const char*)w1_m; f.w2 = (const char*)w2_m; f.b1 = b1; f.b2 = b2;
      f.resid = x; f.out = x; f.ldr = kDim; f.ldo = kDim; f.M = m; f.exp1 = exps + 0; f.exp2 = exps + 1;
      if (ln_rows) { f.ln_w = ln_w; f.ln_b = ln_b; f.ln_out = (char*)ln_rows; }
      HIP_TRY(launch_ffn_fused(f, s));
    } else {
      GemmArgs g1{};
      g1.fmt = FMT_MIXED; g1.w_exp = exps + 0; g1.a = a_m; g1.w = w1_m; g1.bias = b1; g1.c_split = hid;
      g1.M = m; g1.N = 2 * kDim; g1.K = kDim; g1.ldc = 4 * kDim;
      HIP_TRY(launch_gemm_split(g1, EPI_GELU_SPLIT, 0, s));
      GemmArgs g2{};
      g2.fmt = FMT_MIXED; g2.w_exp = exps + 1; g2.a = hid; g2.w = w2_m; g2.bias = b2; g2.resid = x; g2.c = x;
      g2.M = m; g2.N = kDim; g2.K = 2 * kDim; g2.ldr = kDim; g2.ldc = kDim;
      HIP_TRY(launch_gemm_split(g2, EPI_RESID, 0, s));
      if (ln_rows) HIP_TRY(launch_layernorm(x, kDim, ln_w, ln_b, (__bf16*)ln_rows, m, s, FMT_MIXED));
    }
  }
  if (ms_per_rep) {
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_rep = ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  return VETO_OK;
}

// Test / measurement hook of the attention out projection + residual (model_veto.py:96 `to_out`, :20) on VETO_MIXED operands:
// x <- x + a W^T + b for m token rows, optionally followed by LayerNorm rows (the FeedForward PreNorm, :125-132).  mode 0 = the GEMM
// launch with the residual epilogue (+ a LayerNorm launch), mode 1 = the full-row panel kernel (ffn_fused.hip, MODE 1).
size_t veto_debug_outproj_workspace_bytes(int32_t m) {
  if (m <= 0) return 0;
  const size_t mp = (size_t)gemm_rows_padded(m);
  return align_up(mp * kDim * 4, 256) + align_up((size_t)kDim * kDim * 4, 256) + 256;
}

int veto_debug_outproj(void* stream, const float* a, const float* w, const float* b, float* x, int32_t m, int32_t mode, int32_t flags,
                       int32_t reps, float* ms_per_rep, void* workspace, size_t workspace_bytes, const float* ln_w, const float* ln_b,
                       void* ln_rows) {
  if (!a || !w || !b || !x || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (m <= 0 || reps <= 0 || (mode != 0 && mode != 1)) return fail(VETO_ERR_INVALID, "bad m / reps / mode");
  if (ln_rows && (!ln_w || !ln_b)) return fail(VETO_ERR_INVALID, "ln_rows needs ln_w and ln_b");
  if (workspace_bytes < veto_debug_outproj_workspace_bytes(m)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t mp = (size_t)gemm_rows_padded(m);
  char* base = (char*)workspace;
  __bf16* a_m = (__bf16*)base;
  __bf16* w_m = (__bf16*)(base + align_up(mp * kDim * 4, 256));
  int* exps = (int*)((char*)w_m + align_up((size_t)kDim * kDim * 4, 256));
  if (flags & 1) HIP_TRY(launch_mixed_weight_rows(w, w_m, (size_t)kDim, kDim, exps, s));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ms_per_rep) {
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
  }
  float total = 0.f;
  for (int r = 0; r < reps; ++r) {
    // the fused form writes its LayerNorm rows over its input rows: rebuild them for every run (outside the timed span)
    HIP_TRY(hipMemsetAsync(a_m, 0, mp * kDim * 4, s));
    HIP_TRY(launch_mixed_act_rows(a, a_m, (size_t)m, kDim, s));
    if (ms_per_rep) HIP_TRY(hipEventRecord(e0, s));
    if (mode == 1) {
      FfnArgs f{};
      f.a = (const char*)a_m; f.w2 = (const char*)w_m; f.b2 = b; f.resid = x; f.out = x; f.ldr = kDim; f.ldo = kDim; f.M = m; f.exp2 = exps;
      if (ln_rows) { f.ln_w = ln_w; f.ln_b = ln_b; f.ln_out = (char*)ln_rows; }
      HIP_TRY(launch_out_fused(f, s));
    } else {
      GemmArgs g{};
      g.fmt = FMT_MIXED; g.w_exp = exps; g.a = a_m; g.w = w_m; g.bias = b; g.resid = x; g.c = x;
      g.M = m; g.N = kDim; g.K = kDim; g.ldr = kDim; g.ldc = kDim;
      HIP_TRY(launch_gemm_split(g, EPI_RESID, 0, s));
      if (ln_rows) HIP_TRY(launch_layernorm(x, kDim, ln_w, ln_b, (__bf16*)ln_rows, m, s, FMT_MIXED));
    }
    if (ms_per_rep) {
      HIP_TRY(hipEventRecord(e1, s));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      total += ms;
    }
  }
  if (ms_per_rep) {
    *ms_per_rep = total / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  return VETO_OK;
}

// Test / measurement hook of everything behind the attention of one layer (model_veto.py:96, :20-21, :125-143): x1 = x + a Wo^T +
// bo, h = LayerNorm2(x1), x2 = x1 + W2 gelu(W1 h + b1) + b2 (+ LayerNorm rows of x2) on VETO_MIXED operands.  mode 0 = the
// out-projection panel launch + the FeedForward panel launch, mode 1 = ONE launch (ffn_fused.hip MODE 2).
size_t veto_debug_layer_tail_workspace_bytes(int32_t m) {
  if (m <= 0) return 0;
  const size_t mp = (size_t)gemm_rows_padded(m);
  return align_up(mp * kDim * 4, 256) + align_up((size_t)kDim * kDim * 4, 256) + 2 * align_up((size_t)2 * kDim * kDim * 4, 256) + 256;
}

int veto_debug_layer_tail(void* stream, const float* a, const float* wo, const float* bo, const float* ln2_w, const float* ln2_b,
                          const float* w1, const float* b1, const float* w2, const float* b2, float* x, int32_t m, int32_t mode,
                          int32_t reps, float* ms_per_rep, void* workspace, size_t workspace_bytes, const float* ln_w, const float* ln_b,
                          void* ln_rows) {
  if (!a || !wo || !bo || !ln2_w || !ln2_b || !w1 || !b1 || !w2 || !b2 || !x || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (m <= 0 || reps <= 0 || (mode != 0 && mode != 1)) return fail(VETO_ERR_INVALID, "bad m / reps / mode");
  if (ln_rows && (!ln_w || !ln_b)) return fail(VETO_ERR_INVALID, "ln_rows needs ln_w and ln_b");
  if (workspace_bytes < veto_debug_layer_tail_workspace_bytes(m)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t mp = (size_t)gemm_rows_padded(m);
  char* base = (char*)workspace;
  __bf16* a_m = (__bf16*)base;
  __bf16* wo_m = (__bf16*)(base + align_up(mp * kDim * 4, 256));
  __bf16* w1_m = (__bf16*)((char*)wo_m + align_up((size_t)kDim * kDim * 4, 256));
  __bf16* w2_m = (__bf16*)((char*)w1_m + align_up((size_t)2 * kDim * kDim * 4, 256));
  int* exps = (int*)((char*)w2_m + align_up((size_t)2 * kDim * kDim * 4, 256));
  HIP_TRY(launch_mixed_weight_rows(wo, wo_m, (size_t)kDim, kDim, exps + 0, s));
  HIP_TRY(launch_mixed_weight_rows(w1, w1_m, (size_t)2 * kDim, kDim, exps + 1, s));
  HIP_TRY(launch_mixed_weight_rows(w2, w2_m, (size_t)kDim, 2 * kDim, exps + 2, s));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ms_per_rep) {
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
  }
  float total = 0.f;
  for (int r = 0; r < reps; ++r) {
    HIP_TRY(hipMemsetAsync(a_m, 0, mp * kDim * 4, s));       // the activation rows are overwritten in place: rebuilt per run, untimed
    HIP_TRY(launch_mixed_act_rows(a, a_m, (size_t)m, kDim, s));
    if (ms_per_rep) HIP_TRY(hipEventRecord(e0, s));
    FfnArgs f{};
    f.a = (const char*)a_m; f.resid = x; f.out = x; f.ldr = kDim; f.ldo = kDim; f.M = m; f.ln_out = (char*)a_m;
    if (mode == 1) {
      f.wo = (const char*)wo_m; f.bo = bo; f.expo = exps + 0; f.lnm_w = ln2_w; f.lnm_b = ln2_b;
      f.w1 = (const char*)w1_m; f.w2 = (const char*)w2_m; f.b1 = b1; f.b2 = b2; f.exp1 = exps + 1; f.exp2 = exps + 2;
      if (ln_rows) { f.ln_w = ln_w; f.ln_b = ln_b; }
      HIP_TRY(launch_layer_tail(f, s));
    } else {
      FfnArgs o = f;
      o.w2 = (const char*)wo_m; o.b2 = bo; o.exp2 = exps + 0; o.ln_w = ln2_w; o.ln_b = ln2_b;
      HIP_TRY(launch_out_fused(o, s));
      f.w1 = (const char*)w1_m; f.w2 = (const char*)w2_m; f.b1 = b1; f.b2 = b2; f.exp1 = exps + 1; f.exp2 = exps + 2;
      if (ln_rows) { f.ln_w = ln_w; f.ln_b = ln_b; } else f.ln_out = nullptr;
      HIP_TRY(launch_ffn_fused(f, s));
    }
    if (ms_per_rep) {
      HIP_TRY(hipEventRecord(e1, s));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      total += ms;
    }
  }
  if (ln_rows) HIP_TRY(hipMemcpyAsync(ln_rows, a_m, (size_t)m * kDim * 4, hipMemcpyDeviceToDevice, s));
  if (ms_per_rep) {
    *ms_per_rep = total / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  return VETO_OK;
}

// Test / bench hook of the fused QKV + attention launch: a = LayerNorm1 rows fp32 [19 n_pair, 576], wqkv fp32 [1728, 576]; out_rows receives
// the attention output as mixed rows [19 n_pair, 4*576 B].  mode 1 = qkv_attn_fused.hip, mode 0 = the two launches it replaces (QKV GEMM
// with 3-byte q / k / v + attention_mfma_kernel).
size_t veto_debug_qkv_attn_workspace_bytes(int32_t n_pair) {
  if (n_pair <= 0) return 0;
  const size_t tile_rows = qkv_attn_rows_padded(n_pair), m = (size_t)n_pair * kTokens;
  const size_t mp = (size_t)gemm_rows_padded((int)(tile_rows > m ? tile_rows : m));
  return 2 * align_up(mp * kDim * 4, 256) + align_up(mp * 3 * kDim * 3, 256) + align_up((size_t)3 * kDim * kDim * 4, 256) + 256;
}

int veto_debug_qkv_attn(void* stream, const float* a, const float* wqkv, int32_t n_pair, int32_t heads, int32_t mode, int32_t reps,
                        float* ms_per_rep, void* workspace, size_t workspace_bytes, void* out_rows) {
  if (!a || !wqkv || !workspace || !out_rows) return fail(VETO_ERR_INVALID, "null argument");
  if (n_pair <= 0 || reps <= 0 || (mode != 0 && mode != 1) || !qkv_attn_fused_supports(heads)) return fail(VETO_ERR_INVALID, "bad n_pair / reps / mode / heads");
  if (workspace_bytes < veto_debug_qkv_attn_workspace_bytes(n_pair)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t tile_rows = qkv_attn_rows_padded(n_pair), m = (size_t)n_pair * kTokens;
  const size_t mp = (size_t)gemm_rows_padded((int)(tile_rows > m ? tile_rows : m));
  char* base = (char*)workspace;
  __bf16* a_m = (__bf16*)base;
  char* o_m = base + align_up(mp * kDim * 4, 256);
  char* qkv = o_m + align_up(mp * kDim * 4, 256);
  __bf16* w_m = (__bf16*)(qkv + align_up(mp * 3 * kDim * 3, 256));
  int* exps = (int*)((char*)w_m + align_up((size_t)3 * kDim * kDim * 4, 256));
  HIP_TRY(hipMemsetAsync(a_m, 0, mp * kDim * 4, s));
  HIP_TRY(launch_mixed_act_rows(a, a_m, m, kDim, s));
  HIP_TRY(launch_mixed_weight_rows(wqkv, w_m, (size_t)3 * kDim, kDim, exps, s));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ms_per_rep) {
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, s));
  }
  for (int r = 0; r < reps; ++r) {
    if (mode == 1) {
      QkvAttnArgs q{};
      q.a = (const char*)a_m; q.w = (const char*)w_m; q.w_exp = exps; q.o = o_m; q.n_pair = n_pair; q.heads = heads;
      HIP_TRY(launch_qkv_attn_fused(q, s));
    } else {
      GemmArgs g{};
      g.fmt = FMT_MIXED; g.w_exp = exps; g.a = a_m; g.w = w_m; g.c = (float*)qkv; g.M = (int)m; g.N = 3 * kDim; g.K = kDim; g.ldc = 3 * kDim;
      HIP_TRY(launch_gemm_split(g, EPI_F24, 0, s));
      AttnArgs t{};
      t.qkv = (const float*)qkv; t.o = (__bf16*)o_m; t.n_pair = n_pair; t.heads = heads; t.cls_only = 0; t.qkv_f24 = 1; t.o_fmt = FMT_MIXED;
      HIP_TRY(launch_attention(t, s));
    }
  }
  if (ms_per_rep) {
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_rep = ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  HIP_TRY(hipMemcpyAsync(out_rows, o_m, m * kDim * 4, hipMemcpyDeviceToDevice, s));
  return VETO_OK;
}

size_t veto_ce_loss_workspace_bytes(int32_t n) { return n > 0 ? 3 * align_up((size_t)n * 4, 256) + 256 : 0; }

int veto_ce_loss(void* stream, const float* logits, int64_t ld, const int64_t* labels, const float* weight,
                 const int64_t* rows, int32_t n, int32_t n_cls, float* loss, float* grad, void* workspace,
                 size_t workspace_bytes) {
  if (!logits || !labels || !loss || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (n <= 0 || n_cls < 2 || ld < n_cls) return fail(VETO_ERR_INVALID, "bad sizes (n %d, n_cls %d, ld %lld)", n, n_cls, (long long)ld);
  if (workspace_bytes < veto_ce_loss_workspace_bytes(n)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  CeLossArgs a{};
  a.logits = logits; a.ld = ld; a.labels = labels; a.weight = weight; a.rows = rows; a.n = n; a.C = n_cls;
  char* base = (char*)workspace;
  const size_t per = align_up((size_t)n * 4, 256);
  a.lse = (float*)base; a.nll_w = (float*)(base + per); a.w_row = (float*)(base + 2 * per); a.inv_wsum = (float*)(base + 3 * per);
  a.loss = loss; a.grad = grad;
  HIP_TRY(launch_ce_loss(a, (hipStream_t)stream));
  return VETO_OK;
}

int veto_meet_sample(void* stream, const int64_t* labels, int32_t n, const uint32_t* words, int32_t n_words,
                     const int32_t* incre_idx_list, const int32_t* pos_in_group, const int32_t* group_size,
                     const double* sample_rates, int32_t n_groups, int32_t n_cls, int64_t* chosen,
                     int64_t* group_labels, int32_t* counts, int32_t* words_used) {
  if (!labels || !words || !incre_idx_list || !pos_in_group || !group_size || !sample_rates || !chosen || !group_labels ||
      !counts || !words_used)
    return fail(VETO_ERR_INVALID, "null argument");
  if (n <= 0 || n_words <= 0 || n_groups < 1 || n_groups > 64 || n_cls < 2) return fail(VETO_ERR_INVALID, "bad sizes");
  MeetSampleArgs a{};
  a.labels = labels; a.n = n; a.n_groups = n_groups; a.n_cls = n_cls; a.n_words = n_words; a.words = words;
  a.incre = incre_idx_list; a.pos_in_group = pos_in_group; a.group_size = group_size; a.rates = sample_rates;
  a.chosen = chosen; a.group_labels = group_labels; a.counts = counts; a.words_used = words_used;
  HIP_TRY(launch_meet_sample(a, (hipStream_t)stream));
  return VETO_OK;
}

// dw[N,K] = dy[M,N]^T . x[M,K]: the weight-gradient shape (reduction over the M rows).  Both operands are
// transposed into split rows ([N, 2*Mp] and [K, 2*Mp]) and the persistent GEMM runs split-K with atomic adds.
static int wgrad_splits(int n, int k, int m, int k_splits) {
  if (k_splits > 0) return k_splits;
  const int out_tiles = ((n + 255) / 256) * (k / 192);
  int ks = 2 * 256 / out_tiles;                         // just under 2 full rounds of the 256 persistent workgroups
  const int max_ks = (m + 32 * 64 - 1) / (32 * 64);     // at least 64 k-steps per tile
  if (ks > max_ks) ks = max_ks;
  return ks < 1 ? 1 : ks;
}

static size_t wgrad_mp(int m, int ks) { return ((size_t)m + 32 * (size_t)ks - 1) / (32 * (size_t)ks) * 32 * (size_t)ks; }

// n % 32 == 0 (every Linear of the transformer): the row-major form, both operands as the split rows the backward holds
// anyway, transposed on their way out of LDS (GemmArgs::tn).  Otherwise: explicit transposed split copies.
static size_t wgrad_tn_bytes(int m, int n, int k) {
  return align_up(((size_t)m + 1) * n * 4, 256) + align_up(((size_t)m + 1) * k * 4, 256) + 1024;
}

size_t veto_debug_wgrad_workspace_bytes(int32_t m, int32_t n, int32_t k, int32_t k_splits) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  if (n % 32 == 0) return wgrad_tn_bytes(m, n, k);
  const size_t mp = wgrad_mp(m, wgrad_splits(n, k, m, k_splits));
  return align_up((size_t)gemm_rows_padded(n) * mp * 4, 256) + align_up((size_t)k * mp * 4, 256);
}

int veto_debug_wgrad(void* stream, const float* dy, const float* x, float* dw, int32_t m, int32_t n, int32_t k,
                     int32_t k_splits, void* workspace, size_t workspace_bytes) {
  if (!dy || !x || !dw || !workspace) return fail(VETO_ERR_INVALID, "null argument");
  if (m <= 0 || n <= 0 || k <= 0 || k % 192 != 0) return fail(VETO_ERR_INVALID, "k must be a positive multiple of 192");
  if (workspace_bytes < veto_debug_wgrad_workspace_bytes(m, n, k, k_splits)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int ks = wgrad_splits(n, k, m, k_splits);
  const size_t mp = wgrad_mp(m, ks);
  char* base = (char*)workspace;
  HIP_TRY(hipMemsetAsync(dw, 0, (size_t)n * k * 4, s));
  GemmArgs g{};
  g.c = dw;
  g.M = n; g.N = k; g.K = (int)mp; g.ldc = k; g.k_splits = ks;
  if (n % 32 == 0) {
    const size_t a_bytes = align_up(((size_t)m + 1) * n * 4, 256), w_bytes = align_up(((size_t)m + 1) * k * 4, 256);
    __bf16* a_s = (__bf16*)base;
    __bf16* w_s = (__bf16*)(base + a_bytes);
    HIP_TRY(hipMemsetAsync(base, 0, a_bytes + w_bytes + 1024, s));   // the row behind the last one is read (never used) by partial tiles
    HIP_TRY(launch_split_rows(dy, a_s, (size_t)m, n, s));
    HIP_TRY(launch_split_rows(x, w_s, (size_t)m, k, s));
    g.a = a_s; g.w = w_s;
    g.tn = 1; g.lda = 2 * (long)n; g.ldw = 2 * (long)k; g.k_valid = m;
    g.zero = (const __bf16*)(base + a_bytes + w_bytes);
  } else {
    const size_t a_bytes = align_up((size_t)gemm_rows_padded(n) * mp * 4, 256);
    __bf16* a_s = (__bf16*)base;
    __bf16* w_s = (__bf16*)(base + a_bytes);
    HIP_TRY(hipMemsetAsync(base, 0, a_bytes, s));   // rows n..padded stay zero
    HIP_TRY(launch_transpose_split(dy, n, m, n, a_s, (int)mp, s));
    HIP_TRY(launch_transpose_split(x, k, m, k, w_s, (int)mp, s));
    g.a = a_s; g.w = w_s;
  }
  hipError_t e = launch_gemm_split(g, EPI_ATOMIC, 0, s);
  if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? VETO_ERR_INVALID : VETO_ERR_HIP, "wgrad gemm launch failed: %s", hipGetErrorString(e));
  return VETO_OK;
}

int veto_debug_attention_backward(void* stream, const float* qkv, const float* dout, float* dqkv, int32_t n_pair, int32_t heads) {
  if (!qkv || !dout || !dqkv || n_pair <= 0) return fail(VETO_ERR_INVALID, "bad argument");
  hipError_t e = launch_attention_backward(qkv, dout, dqkv, nullptr, n_pair, heads, 0, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e == hipErrorInvalidValue ? VETO_ERR_INVALID : VETO_ERR_HIP, "attention backward: %s (heads must give a head width of 72, 96 or 144)", hipGetErrorString(e));
  return VETO_OK;
}

size_t veto_debug_layernorm_backward_workspace_bytes(int32_t rows) { return rows > 0 ? layernorm_backward_partial_floats(rows) * 4 : 0; }

int veto_debug_layernorm_backward(void* stream, const float* x, const float* dy, const float* gamma, const float* dres,
                                  float* dx, float* dgamma_dbeta, int32_t rows, void* workspace, size_t workspace_bytes) {
  if (!x || !dy || !gamma || !dx || !dgamma_dbeta || !workspace || rows <= 0) return fail(VETO_ERR_INVALID, "bad argument");
  if (workspace_bytes < veto_debug_layernorm_backward_workspace_bytes(rows)) return fail(VETO_ERR_WORKSPACE, "workspace too small");
  HIP_TRY(launch_layernorm_backward(x, dy, gamma, dres, dx, dgamma_dbeta, (float*)workspace, rows, (hipStream_t)stream));
  return VETO_OK;
}

int veto_debug_gelu_backward(void* stream, const float* pre, const float* dh, float* dpre, size_t n) {
  if (!pre || !dh || !dpre || n == 0 || n % 4 != 0) return fail(VETO_ERR_INVALID, "bad argument (n must be a positive multiple of 4)");
  HIP_TRY(launch_gelu_backward(pre, dh, dpre, n, (hipStream_t)stream));
  return VETO_OK;
}

int veto_debug_column_sums(void* stream, const float* dy, int64_t ld, int32_t rows, int32_t n_cols, float* out, void* workspace,
                           size_t workspace_bytes) {
  if (!dy || !out || !workspace || rows <= 0 || n_cols <= 0 || ld < n_cols) return fail(VETO_ERR_INVALID, "bad argument");
  if (workspace_bytes < (size_t)column_sums_chunks() * n_cols * 4) return fail(VETO_ERR_WORKSPACE, "workspace too small (256 * n_cols floats)");
  HIP_TRY(launch_column_sums(dy, ld, rows, n_cols, out, (float*)workspace, column_sums_chunks(), (hipStream_t)stream));
  return VETO_OK;
}

}  // extern "C"
