// Relation evaluators on the device: what the reference computes per image on the
// host in numpy after pulling every BoxList back (pysgg/data/datasets/evaluation/vg/vg_eval.py:459-566 driving
// sgg_eval.py): R@K, no-graph-constraint R@K, zero-shot R@K, GT-pair accuracy A@K, mean recall and
// no-graph-constraint mean recall, K = 20 / 50 / 100, for the GT-box modes.
//
// All of them are functions of, per GT relation g, the position of the FIRST prediction that matches it
// (sgg_eval.py:78-118 builds the inverse map pred -> [gt]; `reduce(np.union1d, pred_to_gt[:k])` contains g iff
// that position is < k):
//   gc_rank   in the prediction list as given (one predicate per pair, :146-166)
//   acc_rank  the same match, counted among the predictions that sit on a GT pair only (:338-366)
//   ng_rank   in the top-100 of obj_s * obj_o * rel_scores[:, 1:] over all (pair, predicate) cells (:221-229)
// One workgroup per image computes the three ranks and the zero-shot flag of every GT relation plus the image's
// per-predicate (count, hits@K) table; a second, single-workgroup kernel folds the images into the final
// numbers in a fixed order (double precision, bit-reproducible).
//
// The top-100 selection works on a unique 64-bit key (score bits, then inverted flat index), i.e. on the total
// order (score descending, flat index ascending).  Fast path: the 100-th largest ROW maximum bounds the 100-th
// largest cell from below, so a radix select over the P row maxima plus a scan of the ~100 rows that reach the
// bound leaves a few hundred candidate cells for an LDS bitonic sort.  General path (fewer than 100 rows, or
// more than 2048 cells above the bound): radix select over all P x (C-1) cells, six histogram passes.
#include "common.h"
#include "kernels.h"

// IoU thresholds are compared against numpy's separately rounded float32 arithmetic: no FMA contraction here
#pragma clang fp contract(off)

namespace veto {

namespace {

constexpr int kNoMatch = 0x3fffffff;
constexpr int kTop = 100;
constexpr int kThreads = 1024;
constexpr int kBig = 2048;   // cells above the row-maximum bound that the pruned path can hold

__device__ __forceinline__ float iou_plus1(const float* a, const float* b) {
  // structures/boxlist_ops.py:54-90, float32, +1 pixel convention
  const float area_a = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  const float area_b = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
  const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + 1.f, 0.f);
  const float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + 1.f, 0.f);
  const float inter = w * h;
  return inter / (area_a + area_b - inter);
}

// monotone map float -> uint32 (larger float = larger key), valid for all non-NaN floats
__device__ __forceinline__ uint32_t float_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct Sel {            // radix-select state shared by the workgroup
  unsigned long long prefix;   // bits of the 64-bit key decided so far (aligned to the top)
  int need;                    // how many keys with this prefix are still wanted
};

__global__ __launch_bounds__(kThreads) void sgg_eval_image_kernel(SggEvalArgs a) {
  __shared__ int s_hist[4096];
  __shared__ int s_scan[kThreads];
  __shared__ Sel s_sel;
  __shared__ unsigned long long s_cand[128];
  __shared__ unsigned long long s_big[kBig];
  __shared__ int s_ncand;
  const int img = blockIdx.x, tid = threadIdx.x;
  const int g0 = a.gt_off[img], G = a.gt_off[img + 1] - g0;
  const int o0 = a.obj_off[img];
  const int p0 = a.pair_off[img], P = a.pair_off[img + 1] - p0;
  const int C = a.n_rel_cls, Cf = C - 1;
  int* cls_tab = a.cls_table + (size_t)img * 7 * C;   // [count | hits gc@20,50,100 | hits ng@20,50,100][C]
  for (int i = tid; i < 7 * C; i += kThreads) cls_tab[i] = 0;
  for (int g = tid; g < G; g += kThreads) {
    a.gc_rank[g0 + g] = kNoMatch;
    a.ng_rank[g0 + g] = kNoMatch;
    a.acc_first[g0 + g] = kNoMatch;
  }
  if (tid == 0) a.ng_count[img] = 0;
  if (G == 0 || P == 0) return;   // vg_eval.py:474-475 / :544-545: the image contributes nothing
  const int64_t* pairs = a.pred_pairs + 2 * (size_t)p0;
  const float* scores = a.rel_scores + (size_t)p0 * C;
  const int64_t* gt = a.gt_rels + 3 * (size_t)g0;

  // ---- per prediction: graph-constraint label, pair score, "sits on a GT pair" flag ---------------------
  for (int p = tid >> 6; p < P; p += kThreads >> 6) {   // one wave per prediction row
    const int ln = tid & 63;
    const float* row = scores + (size_t)p * C;
    float best = -INFINITY;
    int lab = 0x7fffffff;
    for (int c = 1 + ln; c < C; c += 64)
      if (row[c] > best) { best = row[c]; lab = c; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                     // first maximum, as numpy argmax: ties -> lower class
      const float ob = __shfl_xor(best, o, 64);
      const int ol = __shfl_xor(lab, o, 64);
      if (ob > best || (ob == best && ol < lab)) { best = ob; lab = ol; }
    }
    const int s = (int)pairs[2 * p], o = (int)pairs[2 * p + 1];
    int flag = 0;
    for (int g = ln; g < G; g += 64) flag |= (gt[3 * g] == s && gt[3 * g + 1] == o);
    flag = __any(flag);
    if (ln == 0) {
      const float ps = a.obj_scores[o0 + s] * a.obj_scores[o0 + o];
      a.label_tmp[p0 + p] = lab;
      a.pair_score[p0 + p] = ps;
      a.row_key[p0 + p] = float_key(ps * best);   // the row's largest cell (ps >= 0: the product is monotone)
      a.flag_tmp[p0 + p] = flag ? 1 : 0;
    }
  }
  __syncthreads();
  // exclusive prefix count of the flags, in prediction order (chunks of kThreads with a carry)
  {
    __shared__ int s_carry;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < P; base += kThreads) {
      const int p = base + tid;
      const int v = p < P ? a.flag_tmp[p0 + p] : 0;
      s_scan[tid] = v;
      __syncthreads();
      for (int d = 1; d < kThreads; d <<= 1) {
        const int t = tid >= d ? s_scan[tid - d] : 0;
        __syncthreads();
        s_scan[tid] += t;
        __syncthreads();
      }
      if (p < P) a.flag_before[p0 + p] = s_carry + s_scan[tid] - v;
      __syncthreads();
      if (tid == kThreads - 1) s_carry += s_scan[tid];
      __syncthreads();
    }
  }

  // ---- no-graph-constraint list: the kTop largest cells under (score desc, flat index asc) ---------------
  const long M = (long)P * Cf;
  // cell (p, c), c = 1..C-1, has flat index i = p * (C-1) + (c-1) as in numpy's ravel of rel_scores[:, 1:]
  auto key_of = [&](int p, int c) -> unsigned long long {
    const float sc = a.pair_score[p0 + p] * scores[(size_t)p * C + c];
    return ((unsigned long long)float_key(sc) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(p * Cf + (c - 1)));
  };
  const int lane = tid & 63, wave = tid >> 6, n_wave = kThreads >> 6;
  int n_list = -1;
  if (M <= kTop) {
    for (int i = tid; i < (int)M; i += kThreads) s_cand[i] = key_of(i / Cf, i % Cf + 1);
    n_list = (int)M;
  } else if (P >= kTop) {
    // Pruning: the kTop-th largest ROW MAXIMUM B is a lower bound of the kTop-th largest cell (there are kTop
    // cells >= B), so only cells >= B can be in the list and only rows whose maximum is >= B hold any.  A radix
    // select over the P row keys (3 digits of the 32-bit score key) finds B; then ~kTop rows are scanned.
    if (tid == 0) { s_sel.prefix = 0ull; s_sel.need = kTop; s_ncand = 0; }
    __syncthreads();
    const int shifts[3] = {20, 8, 0};
    const int nbits[3] = {12, 12, 8};
    int done_bits = 0;
    for (int lvl = 0; lvl < 3; ++lvl) {
      const int nb = 1 << nbits[lvl];
      for (int i = tid; i < nb; i += kThreads) s_hist[i] = 0;
      __syncthreads();
      const uint32_t prefix = (uint32_t)s_sel.prefix;
      const uint32_t pmask = done_bits ? ~0u << (32 - done_bits) : 0u;
      for (int p = tid; p < P; p += kThreads) {
        const uint32_t k = a.row_key[p0 + p];
        if ((k & pmask) == prefix) atomicAdd(&s_hist[(int)((k >> shifts[lvl]) & (uint32_t)(nb - 1))], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int need = s_sel.need, d = nb - 1;
        for (; d > 0; --d) {
          if (s_hist[d] >= need) break;
          need -= s_hist[d];
        }
        s_sel.prefix = (unsigned long long)(prefix | ((uint32_t)d << shifts[lvl]));
        s_sel.need = need;
      }
      __syncthreads();
      done_bits += nbits[lvl];
    }
    const uint32_t bound = (uint32_t)s_sel.prefix;
    for (int p = wave; p < P; p += n_wave) {
      if (a.row_key[p0 + p] < bound) continue;       // wave-uniform: the whole row lies below the bound
      for (int c = 1 + lane; c < C; c += 64) {
        const unsigned long long k = key_of(p, c);
        if ((uint32_t)(k >> 32) >= bound) {
          const int slot = atomicAdd(&s_ncand, 1);
          if (slot < kBig) s_big[slot] = k;
        }
      }
    }
    __syncthreads();
    const int n_big = s_ncand;
    __syncthreads();
    if (n_big <= kBig) {
      int n2 = 128;
      while (n2 < n_big) n2 <<= 1;
      for (int i = tid; i < n2; i += kThreads)
        if (i >= n_big) s_big[i] = 0ull;
      __syncthreads();
      for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < n2; i += kThreads) {
            const int l = i ^ j;
            if (l > i) {
              const unsigned long long x = s_big[i], y = s_big[l];
              const bool desc_block = (i & k) == 0;
              if (desc_block ? x < y : x > y) { s_big[i] = y; s_big[l] = x; }
            }
          }
          __syncthreads();
        }
      n_list = n_big < kTop ? n_big : kTop;
      if (tid < 128) s_cand[tid] = tid < n_list ? s_big[tid] : 0ull;
      __syncthreads();
    }
  }
  if (n_list < 0) {   // fewer than kTop rows, or too many cells above the bound: select over all cells
    if (tid == 0) { s_sel.prefix = 0ull; s_sel.need = kTop; s_ncand = 0; }
    __syncthreads();
    // six digits, most significant first, cover all 64 bits: after the last one `prefix` IS the kTop-th largest
    // key (keys are unique), so `key >= prefix` selects exactly kTop cells
    const int shifts[6] = {52, 40, 32, 22, 12, 0};
    const int nbits[6] = {12, 12, 8, 10, 10, 12};
    int done_bits = 0;
    for (int lvl = 0; lvl < 6; ++lvl) {
      const int nb = 1 << nbits[lvl];
      for (int i = tid; i < nb; i += kThreads) s_hist[i] = 0;
      __syncthreads();
      const unsigned long long prefix = s_sel.prefix;
      const unsigned long long pmask = done_bits ? ~0ull << (64 - done_bits) : 0ull;
      for (int p = wave; p < P; p += n_wave)       // a wave walks one row of scores: coalesced, no division
        for (int c = 1 + lane; c < C; c += 64) {
          const unsigned long long k = key_of(p, c);
          if ((k & pmask) == prefix) atomicAdd(&s_hist[(int)((k >> shifts[lvl]) & (unsigned long long)(nb - 1))], 1);
        }
      __syncthreads();
      if (tid == 0) {   // walk the digits from the top until `need` keys are covered
        int need = s_sel.need, d = nb - 1;
        for (; d > 0; --d) {
          if (s_hist[d] >= need) break;
          need -= s_hist[d];
        }
        s_sel.prefix = prefix | ((unsigned long long)d << shifts[lvl]);
        s_sel.need = need;   // keys still wanted inside digit d
      }
      __syncthreads();
      done_bits += nbits[lvl];
    }
    const unsigned long long thr = s_sel.prefix;
    for (int p = wave; p < P; p += n_wave)
      for (int c = 1 + lane; c < C; c += 64) {
        const unsigned long long k = key_of(p, c);
        if (k >= thr) {
          const int slot = atomicAdd(&s_ncand, 1);
          if (slot < 128) s_cand[slot] = k;
        }
      }
    __syncthreads();
    n_list = s_ncand < kTop ? s_ncand : kTop;
  }
  __syncthreads();
  // sort the candidates descending (bitonic over 128 slots, padding = 0 sorts last)
  if (tid < 128 && tid >= n_list) s_cand[tid] = 0ull;
  __syncthreads();
  for (int k = 2; k <= 128; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (tid < 128) {
        const int l = tid ^ j;
        if (l > tid) {
          const unsigned long long x = s_cand[tid], y = s_cand[l];
          const bool desc_block = (tid & k) == 0;
          if (desc_block ? x < y : x > y) { s_cand[tid] = y; s_cand[l] = x; }
        }
      }
      __syncthreads();
    }
  if (tid < n_list) {
    const long i = (long)(0xffffffffu - (uint32_t)(s_cand[tid] & 0xffffffffull));
    a.ng_rows[(size_t)img * kTop + tid] = (int)(i / Cf);
    a.ng_cols[(size_t)img * kTop + tid] = (int)(i % Cf) + 1;
  }
  if (tid == 0) a.ng_count[img] = n_list;
  __syncthreads();

  // ---- matching: first matching prediction per GT relation in each list ------------------------------------
  const float thr_iou = a.iou_thres;
  for (long w = tid; w < (long)G * P; w += kThreads) {
    const int g = (int)(w / P), p = (int)(w % P);
    const int gs = (int)gt[3 * g], go = (int)gt[3 * g + 1], gr = (int)gt[3 * g + 2];
    if (a.label_tmp[p0 + p] != gr) continue;
    const int s = (int)pairs[2 * p], o = (int)pairs[2 * p + 1];
    if (a.pred_classes[o0 + s] != a.gt_classes[o0 + gs] || a.pred_classes[o0 + o] != a.gt_classes[o0 + go]) continue;
    if (iou_plus1(a.gt_boxes + 4 * (size_t)(o0 + gs), a.pred_boxes + 4 * (size_t)(o0 + s)) < thr_iou) continue;
    if (iou_plus1(a.gt_boxes + 4 * (size_t)(o0 + go), a.pred_boxes + 4 * (size_t)(o0 + o)) < thr_iou) continue;
    atomicMin(&a.gc_rank[g0 + g], p);
    if (a.flag_tmp[p0 + p]) atomicMin(&a.acc_first[g0 + g], p);
  }
  for (int w = tid; w < G * n_list; w += kThreads) {
    const int g = w / n_list, j = w % n_list;
    const int gs = (int)gt[3 * g], go = (int)gt[3 * g + 1], gr = (int)gt[3 * g + 2];
    if (a.ng_cols[(size_t)img * kTop + j] != gr) continue;
    const int p = a.ng_rows[(size_t)img * kTop + j];
    const int s = (int)pairs[2 * p], o = (int)pairs[2 * p + 1];
    if (a.pred_classes[o0 + s] != a.gt_classes[o0 + gs] || a.pred_classes[o0 + o] != a.gt_classes[o0 + go]) continue;
    if (iou_plus1(a.gt_boxes + 4 * (size_t)(o0 + gs), a.pred_boxes + 4 * (size_t)(o0 + s)) < thr_iou) continue;
    if (iou_plus1(a.gt_boxes + 4 * (size_t)(o0 + go), a.pred_boxes + 4 * (size_t)(o0 + o)) < thr_iou) continue;
    atomicMin(&a.ng_rank[g0 + g], j);
  }
  // zero-shot flag (:279-291): the GT (subject class, object class, predicate) occurs in the table
  for (int g = tid; g < G; g += kThreads) a.zeroshot_flag[g0 + g] = 0;
  __syncthreads();
  for (long w = tid; w < (long)G * a.n_zeroshot; w += kThreads) {
    const int g = (int)(w / a.n_zeroshot);
    const int64_t* z = a.zeroshot + 3 * (w % a.n_zeroshot);
    if (z[0] == a.gt_classes[o0 + (int)gt[3 * g]] && z[1] == a.gt_classes[o0 + (int)gt[3 * g + 1]] && z[2] == gt[3 * g + 2])
      a.zeroshot_flag[g0 + g] = 1;
  }
  __syncthreads();
  // acc_rank = number of GT-pair predictions in front of the first GT-pair match; per-class table
  for (int g = tid; g < G; g += kThreads) {
    const int f = a.acc_first[g0 + g];
    a.acc_rank[g0 + g] = f < kNoMatch ? a.flag_before[p0 + f] : kNoMatch;
    const int r = (int)gt[3 * g + 2];
    if (r > 0 && r < C) {
      atomicAdd(&cls_tab[r], 1);
      const int gc = a.gc_rank[g0 + g], ng = a.ng_rank[g0 + g];
      if (gc < 20) atomicAdd(&cls_tab[1 * C + r], 1);
      if (gc < 50) atomicAdd(&cls_tab[2 * C + r], 1);
      if (gc < 100) atomicAdd(&cls_tab[3 * C + r], 1);
      if (ng < 20) atomicAdd(&cls_tab[4 * C + r], 1);
      if (ng < 50) atomicAdd(&cls_tab[5 * C + r], 1);
      if (ng < 100) atomicAdd(&cls_tab[6 * C + r], 1);
    }
  }
}

// Dataset-level numbers, the reference's accumulation (sgg_eval.py:133-136, :209, :331-336, :420-466):
//   out[0..2] R@K, [3..5] ngR@K, [6..8] zR@K, [9..11] A@K, [12..14] mR@K, [15..17] ng-mR@K,
//   [18 + (kind*3 + k)*(C-1) + n] per-class recall lists (kind 0 = graph constraint, 1 = no graph constraint),
//   then [.. + 0] images evaluated, [.. + 1] images with a zero-shot relation.
// One thread per output quantity, images visited in index order: deterministic.
__global__ __launch_bounds__(256) void sgg_eval_reduce_kernel(SggEvalArgs a) {
  const int C = a.n_rel_cls, Cf = C - 1, tid = threadIdx.x;
  const int ks[3] = {20, 50, 100};
  const int n_scalar = 12, n_cls = 6 * Cf;
  double* out = a.metrics;
  for (int q = tid; q < n_scalar + n_cls + 2; q += blockDim.x) {
    if (q < n_scalar) {
      const int kind = q / 3, k = ks[q % 3];
      double sum = 0.0, hit_sum = 0.0, cnt_sum = 0.0;
      long n = 0;
      for (int img = 0; img < a.n_img; ++img) {
        const int g0 = a.gt_off[img], G = a.gt_off[img + 1] - g0;
        const int P = a.pair_off[img + 1] - a.pair_off[img];
        if (G == 0 || P == 0) continue;
        int hits = 0, zs = 0;
        for (int g = 0; g < G; ++g) {
          const int r = kind == 1 ? a.ng_rank[g0 + g] : kind == 3 ? a.acc_rank[g0 + g] : a.gc_rank[g0 + g];
          const int z = a.zeroshot_flag[g0 + g];
          zs += z;
          if (kind == 2) hits += (r < k) && z; else hits += r < k;
        }
        if (kind == 2) {
          if (zs > 0) { sum += (double)hits / (double)zs; ++n; }
        } else if (kind == 3) {
          hit_sum += (double)hits; cnt_sum += (double)G; ++n;
        } else {
          sum += (double)hits / (double)G; ++n;
        }
      }
      out[q] = kind == 3 ? (n ? (hit_sum / (double)n) / (cnt_sum / (double)n) : __builtin_nan("")) : (n ? sum / (double)n : __builtin_nan(""));
    } else if (q < n_scalar + n_cls) {
      const int e = q - n_scalar, kind = e / (3 * Cf), ki = (e / Cf) % 3, cls = e % Cf + 1;
      double sum = 0.0;
      long n = 0;
      for (int img = 0; img < a.n_img; ++img) {
        const int* tab = a.cls_table + (size_t)img * 7 * C;
        const int cnt = tab[cls];
        if (cnt > 0) { sum += (double)tab[(1 + kind * 3 + ki) * C + cls] / (double)cnt; ++n; }
      }
      out[18 + e] = n ? sum / (double)n : 0.0;
    } else {
      long n = 0;
      for (int img = 0; img < a.n_img; ++img) {
        const int g0 = a.gt_off[img], G = a.gt_off[img + 1] - g0;
        const int P = a.pair_off[img + 1] - a.pair_off[img];
        if (G == 0 || P == 0) continue;
        if (q == n_scalar + n_cls) { ++n; continue; }
        int zs = 0;
        for (int g = 0; g < G; ++g) zs += a.zeroshot_flag[g0 + g];
        n += zs > 0;
      }
      out[18 + n_cls + (q - n_scalar - n_cls)] = (double)n;
    }
  }
  __syncthreads();
  // mean recall = sum of the per-class recalls / (C - 1), classes in index order (:452-466)
  if (tid < 6) {
    double s = 0.0;
    for (int n = 0; n < Cf; ++n) s += out[18 + tid * Cf + n];
    out[12 + tid] = s / (double)Cf;
  }
}

}  // namespace

hipError_t launch_sgg_eval(const SggEvalArgs& a, hipStream_t s) {
  VETO_LAUNCH(sgg_eval_image_kernel, dim3(a.n_img), dim3(kThreads), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  VETO_LAUNCH(sgg_eval_reduce_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace veto
