// Relation post-processing, the step right after the predicate logits:
// the vanilla, GT-box branch of PostProcessor.forward, pysgg/modeling/roi_heads/relation_head/
// inference.py:398-453.  Per image:
//   obj_prob = softmax(obj_logit); obj_prob[:, 0] = 0; obj_score, obj_pred = max over classes 1..   (:405-412)
//   rel_prob = softmax(rel_logit); rel_score, rel_class = max over classes 1..                       (:440-442)
//   triple   = rel_score * obj_score[subj] * obj_score[obj]; descending sort                         (:444-445)
//   emit rel_pair_idx, rel_prob, rel_class in that order                                              (:446-452)
// Kernels: per-object softmax/max, per-pair softmax/max/triple score (one wave per pair), one
// workgroup per image for the segmented sort (bitonic in LDS over (score desc, index asc) -- a total
// order, so the result is deterministic; torch.sort leaves the order of exact ties unspecified),
// and a row gather into sorted order.
#include "common.h"
#include "kernels.h"

namespace veto {

namespace {

// one wave per row: softmax over `ncls` logits, then max over classes 1..ncls-1
__device__ __forceinline__ void wave_softmax_fgmax(const float* __restrict__ logit, int ncls, int lane,
                                                   float* __restrict__ prob_out, float& best, int& best_cls) {
  float mx = -INFINITY;
  for (int c = lane; c < ncls; c += 64) mx = fmaxf(mx, logit[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int c = lane; c < ncls; c += 64) sum += expf(logit[c] - mx);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  best = -1.f;
  best_cls = 0x7fffffff;
  for (int c = lane; c < ncls; c += 64) {
    const float p = expf(logit[c] - mx) * inv;
    if (prob_out) prob_out[c] = p;
    if (c >= 1 && p > best) { best = p; best_cls = c; }  // first maximum inside the lane (ascending c)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {  // arg-max across lanes; ties -> the lower class index (torch.max)
    const float ob = __shfl_xor(best, o, 64);
    const int oc = __shfl_xor(best_cls, o, 64);
    if (ob > best || (ob == best && oc < best_cls)) { best = ob; best_cls = oc; }
  }
}

__global__ __launch_bounds__(256) void obj_score_kernel(const float* __restrict__ obj_logits, int n_obj, int ncls,
                                                        float* __restrict__ obj_scores, int64_t* __restrict__ obj_pred) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= n_obj) return;
  float best;
  int cls;
  wave_softmax_fgmax(obj_logits + (size_t)n * ncls, ncls, lane, nullptr, best, cls);
  if (lane == 0) { obj_scores[n] = best; obj_pred[n] = cls; }
}

__global__ __launch_bounds__(256) void rel_score_kernel(PostArgs a) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= a.n_pair) return;
  float best;
  int cls;
  wave_softmax_fgmax(a.rel_logits + (size_t)p * a.n_rel_cls, a.n_rel_cls, lane, a.prob_tmp + (size_t)p * a.n_rel_cls, best, cls);
  if (lane == 0) {
    int lo = 0, hi = a.n_img - 1;  // image of this pair: largest i with img_pair_off[i] <= p
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (a.img_pair_off[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const int off = a.img_obj_off[lo];
    const float s0 = a.obj_scores[off + a.rel_pairs[2 * (size_t)p]];
    const float s1 = a.obj_scores[off + a.rel_pairs[2 * (size_t)p + 1]];
    a.triple[p] = best * s0 * s1;  // same association as the reference: (rel * obj0) * obj1
    a.label_tmp[p] = cls;
  }
}

constexpr int kSortMax = 16384;  // 128 KiB of LDS: 5 MEET groups x MAX_PROPOSAL_PAIR (2048) pairs fit

// one workgroup per image: sort its pairs by (score descending, original index ascending)
__global__ __launch_bounds__(1024) void segment_sort_kernel(PostArgs a) {
  __shared__ float s_key[kSortMax];
  __shared__ int s_idx[kSortMax];
  const int img = blockIdx.x;
  const int p0 = a.single_cnt > 0 ? 0 : a.img_pair_off[img];
  const int cnt = a.single_cnt > 0 ? a.single_cnt : a.img_pair_off[img + 1] - p0;
  int n2 = 1;
  while (n2 < cnt) n2 <<= 1;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) {
    s_key[i] = i < cnt ? a.triple[p0 + i] : -INFINITY;
    s_idx[i] = i < cnt ? i : 0x7fffffff;  // padding sorts last
  }
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const float ki = s_key[i], kl = s_key[l];
          const int ii = s_idx[i], il = s_idx[l];
          // "i before l" in the final order: higher score first, then lower index (NaN-free inputs)
          const bool i_first = ki > kl || (ki == kl && ii < il);
          const bool ascending_block = (i & k) == 0;
          if (ascending_block ? !i_first : i_first) {
            s_key[i] = kl; s_key[l] = ki;
            s_idx[i] = il; s_idx[l] = ii;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) a.perm[p0 + i] = p0 + s_idx[i];
  if (a.kept_count) {  // voted-out rows carry score -1 and sort last: the kept rows are a prefix
    __shared__ int s_kept;
    if (threadIdx.x == 0) s_kept = 0;
    __syncthreads();
    int mine = 0;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) mine += s_key[i] >= 0.f;
    if (mine) atomicAdd(&s_kept, mine);
    __syncthreads();
    if (threadIdx.x == 0) a.kept_count[img] = s_kept;
  }
}

__global__ __launch_bounds__(256) void gather_sorted_kernel(PostArgs a) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= a.n_pair) return;
  const int src = a.perm[p];
  for (int c = lane; c < a.n_rel_cls; c += 64) a.out_prob[(size_t)p * a.n_rel_cls + c] = a.prob_tmp[(size_t)src * a.n_rel_cls + c];
  if (lane == 0) {
    const int psrc = a.pair_mod > 0 ? src % a.pair_mod : src;  // MEET: K copies of the same pair list
    a.out_pairs[2 * (size_t)p] = a.rel_pairs[2 * (size_t)psrc];
    a.out_pairs[2 * (size_t)p + 1] = a.rel_pairs[2 * (size_t)psrc + 1];
    a.out_labels[p] = a.label_tmp[src];
    if (a.out_triple) a.out_triple[p] = a.triple[src];
  }
}

// MEET group head (inference.py:346-372): softmax over the group's g+2 logits, the last ("other
// group") column dropped, arg-max over columns 1..g (a group-local label), the g+1 probabilities
// scattered into a zero num_rel_cls-wide row at columns [0] + the group's own classes.
__global__ __launch_bounds__(256) void meet_rel_score_kernel(PostArgs a, MeetGroup grp) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= a.n_pair) return;
  const float* logit = grp.logits + (size_t)p * grp.width;
  const size_t out_row = (size_t)grp.row0 + p;
  float* prob = a.prob_tmp + out_row * a.n_rel_cls;
  for (int c = lane; c < a.n_rel_cls; c += 64) prob[c] = 0.f;
  float mx = -INFINITY;
  for (int c = lane; c < grp.width; c += 64) mx = fmaxf(mx, logit[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int c = lane; c < grp.width; c += 64) sum += expf(logit[c] - mx);
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  float best = -1.f;
  int best_cls = 0x7fffffff;
  for (int c = lane; c < grp.width - 1; c += 64) {  // the last column is dropped
    const float pr = expf(logit[c] - mx) * inv;
    prob[grp.cols[c]] = pr;
    if (c >= 1 && pr > best) { best = pr; best_cls = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oc = __shfl_xor(best_cls, o, 64);
    if (ob > best || (ob == best && oc < best_cls)) { best = ob; best_cls = oc; }
  }
  if (lane == 0) {
    const float s0 = a.obj_scores[a.rel_pairs[2 * (size_t)p]];
    const float s1 = a.obj_scores[a.rel_pairs[2 * (size_t)p + 1]];
    a.triple[out_row] = best * s0 * s1;
    a.label_tmp[out_row] = best_cls;
  }
}

// EXPERT_GROUP voting (inference.py:93-283): three expert heads per group.  One wave per (pair, group);
// a lane owns columns lane and lane + 64 of the group's g+1 kept columns (g + 2 <= 105).
//   per expert e: p_e = softmax(logit_e) without its last column, (score_e, cls_e) = max over columns 1..,
//                 t_e = score_e * obj_s * obj_o                                                   (:178-190)
//   agree_ab = cls_a == cls_b for (0,1), (1,2), (0,2)                                              (:192-199)
//   consensus: pair means t_ab = (t_a + t_b)/2, p_ab = (p_a + p_b)/2 with p_12 = p_1 (the reference
//              averages expert 1 with itself, :224-226); row = sum over agreeing pairs / their count (:211-255)
//   unanimous: all three agree; row = three-way mean                                                (:257-261)
// A voted-out row gets score -1 (it sorts behind every kept row, whose scores are >= 0) and label 0.
__global__ __launch_bounds__(256) void vote_rel_score_kernel(PostArgs a, VoteGroup grp, int voting) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= a.n_pair) return;
  const size_t out_row = (size_t)grp.row0 + p;
  float* prob = a.prob_tmp + out_row * a.n_rel_cls;
  for (int c = lane; c < a.n_rel_cls; c += 64) prob[c] = 0.f;
  const float s0 = a.obj_scores[a.rel_pairs[2 * (size_t)p]];
  const float s1 = a.obj_scores[a.rel_pairs[2 * (size_t)p + 1]];
  const int keepw = grp.width - 1;  // the last column is dropped
  float pr[3][2], t[3];
  int cls[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const float* logit = grp.logits[e] + (size_t)p * grp.width;
    float mx = -INFINITY;
    for (int c = lane; c < grp.width; c += 64) mx = fmaxf(mx, logit[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int c = lane; c < grp.width; c += 64) sum += expf(logit[c] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    float best = -1.f;
    int best_cls = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = lane + 64 * j;
      pr[e][j] = c < keepw ? expf(logit[c] - mx) * inv : 0.f;
      if (c >= 1 && c < keepw && pr[e][j] > best) { best = pr[e][j]; best_cls = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oc = __shfl_xor(best_cls, o, 64);
      if (ob > best || (ob == best && oc < best_cls)) { best = ob; best_cls = oc; }
    }
    t[e] = best * s0 * s1;
    cls[e] = best_cls;
  }
  const bool ag01 = cls[0] == cls[1], ag12 = cls[1] == cls[2], ag02 = cls[0] == cls[2];
  float triple, out[2];
  int label;
  bool keep;
  if (voting == 0) {
    const int cnt = (int)ag01 + (int)ag12 + (int)ag02;
    keep = cnt > 0;
    const float fc = (float)cnt;
    triple = (((ag01 ? (t[0] + t[1]) * 0.5f : 0.f) + (ag12 ? (t[1] + t[2]) * 0.5f : 0.f)) + (ag02 ? (t[0] + t[2]) * 0.5f : 0.f)) / fc;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      out[j] = (((ag01 ? (pr[0][j] + pr[1][j]) * 0.5f : 0.f) + (ag12 ? (pr[1][j] + pr[1][j]) * 0.5f : 0.f)) +
                (ag02 ? (pr[0][j] + pr[2][j]) * 0.5f : 0.f)) / fc;
    label = ag02 ? cls[2] : ag12 ? cls[1] : ag01 ? cls[0] : 0;  // later assignments override (:250-251)
  } else {
    keep = ag01 && ag12 && ag02;
    triple = ((t[0] + t[1]) + t[2]) / 3.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) out[j] = ((pr[0][j] + pr[1][j]) + pr[2][j]) / 3.f;
    label = cls[2];
  }
  if (keep) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = lane + 64 * j;
      if (c < keepw) prob[grp.cols[c]] = out[j];
    }
  }
  if (lane == 0) {
    a.triple[out_row] = keep ? triple : -1.f;
    a.label_tmp[out_row] = keep ? label : 0;
  }
}

}  // namespace

hipError_t launch_postprocess_vote(PostArgs a, const VoteGroup* groups, int n_groups, int voting, hipStream_t s) {
  VETO_LAUNCH(obj_score_kernel, dim3((a.n_obj + 3) / 4), dim3(256), 0, s, a.obj_logits, a.n_obj, a.n_obj_cls, a.obj_scores, a.obj_pred);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  for (int k = 0; k < n_groups; ++k) {
    VETO_LAUNCH(vote_rel_score_kernel, dim3((a.n_pair + 3) / 4), dim3(256), 0, s, a, groups[k], voting);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  PostArgs m = a;
  m.single_cnt = n_groups * a.n_pair;
  m.pair_mod = a.n_pair;
  m.n_pair = m.single_cnt;
  VETO_LAUNCH(segment_sort_kernel, dim3(1), dim3(1024), 0, s, m);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  VETO_LAUNCH(gather_sorted_kernel, dim3((m.n_pair + 3) / 4), dim3(256), 0, s, m);
  return hipGetLastError();
}

hipError_t launch_postprocess_meet(PostArgs a, const MeetGroup* groups, int n_groups, hipStream_t s) {
  VETO_LAUNCH(obj_score_kernel, dim3((a.n_obj + 3) / 4), dim3(256), 0, s, a.obj_logits, a.n_obj, a.n_obj_cls, a.obj_scores, a.obj_pred);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  for (int k = 0; k < n_groups; ++k) {
    VETO_LAUNCH(meet_rel_score_kernel, dim3((a.n_pair + 3) / 4), dim3(256), 0, s, a, groups[k]);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  PostArgs m = a;  // the merged list: one "image" of n_groups * n_pair rows
  m.single_cnt = n_groups * a.n_pair;
  m.pair_mod = a.n_pair;
  m.n_pair = m.single_cnt;
  VETO_LAUNCH(segment_sort_kernel, dim3(1), dim3(1024), 0, s, m);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  VETO_LAUNCH(gather_sorted_kernel, dim3((m.n_pair + 3) / 4), dim3(256), 0, s, m);
  return hipGetLastError();
}

int postprocess_max_pairs_per_image() { return kSortMax; }

hipError_t launch_postprocess(const PostArgs& a, hipStream_t s) {
  VETO_LAUNCH(obj_score_kernel, dim3((a.n_obj + 3) / 4), dim3(256), 0, s, a.obj_logits, a.n_obj, a.n_obj_cls, a.obj_scores, a.obj_pred);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  VETO_LAUNCH(rel_score_kernel, dim3((a.n_pair + 3) / 4), dim3(256), 0, s, a);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  VETO_LAUNCH(segment_sort_kernel, dim3(a.n_img), dim3(1024), 0, s, a);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  VETO_LAUNCH(gather_sorted_kernel, dim3((a.n_pair + 3) / 4), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace veto
