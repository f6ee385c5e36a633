// Training-time losses and MEET expert sampling (the parts named there).
//   * weighted cross entropy with mean reduction, nn.CrossEntropyLoss(weight=w): the relation loss of
//     VETOPredictor (roi_relation_predictors.py:4067-4068,4133; BETA_LOSS class-balanced weights :4057-4066) and,
//     unweighted on a row subset with remapped labels, the per-group losses of Ensemble.forward (:3842-3846).
//     loss = sum_i w[y_i] (logsumexp(z_i) - z_i[y_i]) / sum_i w[y_i]; d loss / d z_i = w[y_i] (softmax(z_i) - e_{y_i}) / sum w.
//     One wave per row for the log-sum-exp, a fixed-order double-precision fold (deterministic), one wave per row
//     for the gradient.
//   * the expert sampling loop of VETOPredictor_MEET.forward (:3940-3969), which the reference runs in Python with one
//     .item() per relation: here one wave walks the relations in order, consuming the SAME random stream (raw MT19937
//     words of Python's `random`, prepared by the host: random() = two words, randint = rejection on the top bits),
//     and appends each relation to its groups; a second kernel remaps the labels per group (:3812-3821).
#include "common.h"
#include "kernels.h"

namespace veto {

namespace {

__global__ __launch_bounds__(256) void ce_rows_kernel(CeLossArgs a) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= a.n) return;
  const long r = a.rows ? a.rows[i] : i;
  const float* z = a.logits + r * a.ld;
  float mx = -INFINITY;
  for (int c = lane; c < a.C; c += 64) mx = fmaxf(mx, z[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int c = lane; c < a.C; c += 64) sum += expf(z[c] - mx);
  sum = wave_sum(sum);
  if (lane == 0) {
    // a NEGATIVE label -- nn.CrossEntropyLoss's ignore_index (-100) -- is an ignored row: weight 0.  A label >= C is a bug in the
    // caller's class mapping (torch raises): the row poisons the loss with NaN instead of silently dropping out of it.
    const long yl = a.labels[i];
    const bool valid = yl >= 0 && yl < a.C, bad = yl >= a.C;
    const int y = valid ? (int)yl : 0;
    const float lse = mx + logf(sum);
    const float w = valid ? (a.weight ? a.weight[y] : 1.f) : 0.f;
    a.lse[i] = lse;
    a.nll_w[i] = bad ? __builtin_nanf("") : valid ? w * (lse - z[y]) : 0.f;
    a.w_row[i] = w;
  }
}

// loss = sum nll_w / sum w, rows folded in index order in double precision (one workgroup, deterministic)
__global__ __launch_bounds__(256) void ce_reduce_kernel(CeLossArgs a) {
  __shared__ double s_num[256], s_den[256];
  const int tid = threadIdx.x;
  const int per = (a.n + 255) / 256;
  double num = 0.0, den = 0.0;
  for (int i = tid * per; i < (tid + 1) * per && i < a.n; ++i) { num += (double)a.nll_w[i]; den += (double)a.w_row[i]; }
  s_num[tid] = num;
  s_den[tid] = den;
  __syncthreads();
  if (tid == 0) {
    double n2 = 0.0, d2 = 0.0;
    for (int t = 0; t < 256; ++t) { n2 += s_num[t]; d2 += s_den[t]; }
    a.loss[0] = (float)(n2 / d2);
    a.inv_wsum[0] = (float)(1.0 / d2);
  }
}

__global__ __launch_bounds__(256) void ce_grad_kernel(CeLossArgs a) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= a.n) return;
  const long r = a.rows ? a.rows[i] : i;
  const float* z = a.logits + r * a.ld;
  const long yl = a.labels[i];
  const int y = (yl >= 0 && yl < a.C) ? (int)yl : -1;      // ignored rows have w_row = 0: zero gradient
  // a label >= C made the loss NaN (above); its gradient row is NaN too, so that an optimizer step cannot proceed on a loss that
  // never was one (torch raises in this case)
  const float scale = yl >= a.C ? __builtin_nanf("") : a.w_row[i] * a.inv_wsum[0], lse = a.lse[i];
  float* g = a.grad + (size_t)i * a.C;
  for (int c = lane; c < a.C; c += 64) g[c] = (expf(z[c] - lse) - (c == y ? 1.f : 0.f)) * scale;
}

// ---- MEET expert sampling --------------------------------------------------------------------------------
// Python's random on raw words: random() = ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53; randint(0, G-1) = rejection
// sampling on the top bit_length(G) bits of one word per attempt (CPython _randbelow_with_getrandbits).
__global__ void meet_sample_kernel(MeetSampleArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int pos = 0;
  const int G = a.n_groups;
  int kbits = 0;
  while ((G >> kbits) != 0) ++kbits;
  for (int k = 0; k < G; ++k) a.counts[k] = 0;
  int overflow = 0;
  for (int i = 0; i < a.n; ++i) {
    const int lab = (int)a.labels[i];
    int upto;         // the relation goes to groups 0 .. upto-1, or to the single group `single`
    int single = -1;
    if (lab == 0) {
      if (pos >= a.n_words) { overflow = 1; break; }
      unsigned r = a.words[pos++] >> (32 - kbits);
      while (r >= (unsigned)G) {
        if (pos >= a.n_words) { overflow = 1; break; }
        r = a.words[pos++] >> (32 - kbits);
      }
      if (overflow) break;
      single = (int)r;
      upto = 0;
    } else {
      if (pos + 2 > a.n_words) { overflow = 1; break; }
      const unsigned w0 = a.words[pos] >> 5, w1 = a.words[pos + 1] >> 6;
      pos += 2;
      const double u = ((double)w0 * 67108864.0 + (double)w1) * (1.0 / 9007199254740992.0);
      const int g = a.incre[lab];
      upto = 0;
      for (int act = G; act >= 1; --act)
        if (u <= a.rates[(size_t)(act - 1) * a.n_cls + lab] || act < g) { upto = act; break; }
    }
    if (single >= 0) {
      a.chosen[(size_t)single * a.n + a.counts[single]++] = i;
    } else {
      for (int k = 0; k < upto; ++k) a.chosen[(size_t)k * a.n + a.counts[k]++] = i;
    }
  }
  a.words_used[0] = overflow ? -1 : pos;
}

// group-local labels of the chosen rows (:3812-3821): own classes -> 1.., other foreground -> size + 1, background 0
__global__ __launch_bounds__(256) void meet_labels_kernel(MeetSampleArgs a) {
  const int k = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.counts[k]) return;
  const int lab = (int)a.labels[a.chosen[(size_t)k * a.n + i]];
  int out = 0;
  if (lab != 0) out = a.incre[lab] == k + 1 ? a.pos_in_group[lab] : a.group_size[k] + 1;
  a.group_labels[(size_t)k * a.n + i] = out;
}

}  // namespace

hipError_t launch_ce_loss(const CeLossArgs& a, hipStream_t s) {
  VETO_LAUNCH(ce_rows_kernel, dim3((a.n + 3) / 4), dim3(256), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  VETO_LAUNCH(ce_reduce_kernel, dim3(1), dim3(256), 0, s, a);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (a.grad) {
    VETO_LAUNCH(ce_grad_kernel, dim3((a.n + 3) / 4), dim3(256), 0, s, a);
    e = hipGetLastError();
  }
  return e;
}

hipError_t launch_meet_sample(const MeetSampleArgs& a, hipStream_t s) {
  VETO_LAUNCH(meet_sample_kernel, dim3(1), dim3(64), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  VETO_LAUNCH(meet_labels_kernel, dim3((a.n + 255) / 256, a.n_groups), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace veto
