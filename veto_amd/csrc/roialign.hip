// ROI feature extraction in front of the predictor: the legacy ("aligned =
// False") ROIAlign of pysgg/csrc/cuda/ROIAlign_cuda.cu:16-125 with the FPN level choice of
// LevelMapper (pysgg/modeling/poolers.py:17-43) and Pooler.forward's cat_all_levels=False dispatch
// (poolers.py:109-171) folded into ONE launch: blockIdx.z = 0 pools every ROI from its own FPN level,
// blockIdx.z = 1 pools the depth map with the fixed 1/16 pooler.
//
// Mapping: one workgroup per (ROI, 32-channel slab).  The bilinear weights are separable, so the ROI's
// sample table is 2 x (pooled * grid) axis entries (low/high index, low/high weight, validity) built once
// in LDS and shared by every channel -- the reference recomputes it per output element.  A wave then
// owns one channel plane at a time: its 64 lanes are the 64 output bins, so the 16 taps of a lane fall
// in the ROI's footprint of ONE plane (L1/L2 resident after the first touch) and the store is one
// contiguous 256-byte row of the [R, C, 8, 8] output.
//
// Arithmetic: float32 in the reference's operation order with FMA contraction switched off, so the result
// is bit-identical to the CPU restatement in oracle/roi_align_oracle.py.
#include "common.h"
#include "kernels.h"

// The compiler contracts a*b + c into an FMA by default; the oracle (numpy) rounds every product.  Switch
// contraction off for the code of this file and use plain operators: hip's rounded-op intrinsics are inline
// header functions compiled under the DEFAULT contraction mode, so their results still get fused.
#pragma clang fp contract(off)

namespace veto {

namespace {

constexpr int kMaxAxis = 32;   // pooled * sampling_ratio <= 32
constexpr int kSlab = 32;      // channels per workgroup

struct AxisTable {
  int lo[kMaxAxis], hi[kMaxAxis];
  float l[kMaxAxis], h[kMaxAxis];
  int valid[kMaxAxis];
};

// one axis entry, ROIAlign_cuda.cu:21-57 restricted to one coordinate
__device__ __forceinline__ void axis_entry(AxisTable& t, int k, float start, float bin, int p, int i, int grid, int size) {
  float v = (start + (float)p * bin) + (((float)i + 0.5f) * bin) / (float)grid;
  const bool valid = !(v < -1.0f || v > (float)size);
  if (v <= 0.f) v = 0.f;
  int lo = (int)v, hi;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    v = (float)lo;
  } else {
    hi = lo + 1;
  }
  const float l = v - (float)lo;
  t.lo[k] = valid ? lo : 0;
  t.hi[k] = valid ? hi : 0;
  t.l[k] = l;
  t.h[k] = 1.0f - l;
  t.valid[k] = valid;
}

template <int G>  // G = sampling ratio (samples per bin and axis)
__global__ __launch_bounds__(256) void roi_pool_kernel(RoiPoolArgs a) {
  __shared__ AxisTable ty, tx;
  __shared__ int s_level;
  const int r = blockIdx.x, tid = threadIdx.x;
  const bool depth = blockIdx.z == 1;
  const float* roi = a.rois + (size_t)r * 5;
  const int C = depth ? a.depth_channels : a.channels;
  const int c0 = blockIdx.y * kSlab;
  if (c0 >= C) return;

  if (tid == 0) {
    int lvl = 0;
    if (a.n_levels > 1) {
      // LevelMapper: floor(4 + log2(sqrt(area) / 224 + 1e-6)) clamped to [k_min, k_max], minus k_min; the
      // area uses the +1 pixel convention of BoxList.area() (bounding_box.py:249-259)
      const float area = ((roi[3] - roi[1]) + 1.f) * ((roi[4] - roi[2]) + 1.f);
      const float s = sqrtf(area);
      float t = floorf(4.f + log2f(s / 224.f + 1e-6f));
      t = fminf(fmaxf(t, (float)a.k_min), (float)a.k_max);
      lvl = (int)t - a.k_min;
    }
    s_level = lvl;
    if (a.out_levels && !depth && blockIdx.y == 0) a.out_levels[r] = lvl;
  }
  __syncthreads();
  const RoiLevel L = depth ? a.depth : a.lv[s_level];
  const int P = a.pooled, n_axis = P * G;
  if (tid < 2 * n_axis) {
    const bool is_y = tid < n_axis;
    const int k = is_y ? tid : tid - n_axis;
    // :84-98 no rounding of the scaled box; malformed ROIs become 1x1
    const float lo_c = (is_y ? roi[2] : roi[1]) * L.scale;
    const float hi_c = (is_y ? roi[4] : roi[3]) * L.scale;
    const float len = fmaxf(hi_c - lo_c, 1.0f);
    const float bin = len / (float)P;
    axis_entry(is_y ? ty : tx, k, lo_c, bin, k / G, k % G, G, is_y ? L.H : L.W);
  }
  __syncthreads();

  const int b = (int)roi[0];
  const int bin_id = tid & 63, wv = tid >> 6;
  const int ph = bin_id / P, pw = bin_id % P;
  if (bin_id >= P * P) return;
  // this lane's G x G samples: 4 tap offsets and 4 weights each, fixed for every channel
  int off[G * G][4];
  float wgt[G * G][4];
  bool ok[G * G];
#pragma unroll
  for (int iy = 0; iy < G; ++iy)
#pragma unroll
    for (int ix = 0; ix < G; ++ix) {
      const int ky = ph * G + iy, kx = pw * G + ix, q = iy * G + ix;
      ok[q] = ty.valid[ky] && tx.valid[kx];  // otherwise bilinear_interpolate returns 0 (:26-29)
      const float hy = ty.h[ky], ly = ty.l[ky], hx = tx.h[kx], lx = tx.l[kx];
      off[q][0] = ty.lo[ky] * L.W + tx.lo[kx];
      off[q][1] = ty.lo[ky] * L.W + tx.hi[kx];
      off[q][2] = ty.hi[ky] * L.W + tx.lo[kx];
      off[q][3] = ty.hi[ky] * L.W + tx.hi[kx];
      wgt[q][0] = hy * hx;  // :61
      wgt[q][1] = hy * lx;
      wgt[q][2] = ly * hx;
      wgt[q][3] = ly * lx;
    }
  const float count = (float)(G * G);
  const size_t plane_sz = (size_t)L.H * L.W;
  float* out = (depth ? a.out_depth : a.out_rgb) + (size_t)r * C * P * P;
#pragma unroll 4
  for (int c = c0 + wv; c < c0 + kSlab && c < C; c += 4) {
    const float* plane = L.feat + ((size_t)b * C + c) * plane_sz;
    float v[G * G][4];
#pragma unroll
    for (int q = 0; q < G * G; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) v[q][t] = plane[off[q][t]];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < G * G; ++q) {
      const float val = ((wgt[q][0] * v[q][0] + wgt[q][1] * v[q][1]) + wgt[q][2] * v[q][2]) + wgt[q][3] * v[q][3];  // :63
      acc = acc + (ok[q] ? val : 0.f);
    }
    out[(size_t)c * P * P + bin_id] = acc / count;
  }
}

// Backward (ROIAlign_cuda.cu:178-262): every output gradient is spread over the 4 taps of each of its G x G
// samples as top_diff * w / count, accumulated into the map gradient with atomic adds (several ROIs and bins
// touch the same pixel).  Same mapping and the same LDS sample table as the forward kernel.
template <int G>
__global__ __launch_bounds__(256) void roi_pool_backward_kernel(RoiPoolArgs a) {
  __shared__ AxisTable ty, tx;
  __shared__ int s_level;
  const int r = blockIdx.x, tid = threadIdx.x;
  const bool depth = blockIdx.z == 1;
  const float* roi = a.rois + (size_t)r * 5;
  const int C = depth ? a.depth_channels : a.channels;
  const int c0 = blockIdx.y * kSlab;
  if (c0 >= C) return;
  if (tid == 0) {
    int lvl = 0;
    if (a.n_levels > 1) {
      const float area = ((roi[3] - roi[1]) + 1.f) * ((roi[4] - roi[2]) + 1.f);
      const float s = sqrtf(area);
      float t = floorf(4.f + log2f(s / 224.f + 1e-6f));
      t = fminf(fmaxf(t, (float)a.k_min), (float)a.k_max);
      lvl = (int)t - a.k_min;
    }
    s_level = lvl;
  }
  __syncthreads();
  const RoiLevel L = depth ? a.depth : a.lv[s_level];
  float* grad_map = depth ? a.depth_grad : a.lv_grad[s_level];
  const int P = a.pooled, n_axis = P * G;
  if (tid < 2 * n_axis) {
    const bool is_y = tid < n_axis;
    const int k = is_y ? tid : tid - n_axis;
    const float lo_c = (is_y ? roi[2] : roi[1]) * L.scale;
    const float hi_c = (is_y ? roi[4] : roi[3]) * L.scale;
    const float len = fmaxf(hi_c - lo_c, 1.0f);
    const float bin = len / (float)P;
    axis_entry(is_y ? ty : tx, k, lo_c, bin, k / G, k % G, G, is_y ? L.H : L.W);
  }
  __syncthreads();
  const int b = (int)roi[0];
  const int bin_id = tid & 63, wv = tid >> 6;
  const int ph = bin_id / P, pw = bin_id % P;
  if (bin_id >= P * P) return;
  int off[G * G][4];
  float wgt[G * G][4];
  bool ok[G * G];
#pragma unroll
  for (int iy = 0; iy < G; ++iy)
#pragma unroll
    for (int ix = 0; ix < G; ++ix) {
      const int ky = ph * G + iy, kx = pw * G + ix, q = iy * G + ix;
      ok[q] = ty.valid[ky] && tx.valid[kx];
      const float hy = ty.h[ky], ly = ty.l[ky], hx = tx.h[kx], lx = tx.l[kx];
      off[q][0] = ty.lo[ky] * L.W + tx.lo[kx];
      off[q][1] = ty.lo[ky] * L.W + tx.hi[kx];
      off[q][2] = ty.hi[ky] * L.W + tx.lo[kx];
      off[q][3] = ty.hi[ky] * L.W + tx.hi[kx];
      wgt[q][0] = hy * hx;
      wgt[q][1] = hy * lx;
      wgt[q][2] = ly * hx;
      wgt[q][3] = ly * lx;
    }
  const float count = (float)(G * G);
  const size_t plane_sz = (size_t)L.H * L.W;
  const float* gout = (depth ? a.gout_depth : a.gout_rgb) + (size_t)r * C * P * P;
  for (int c = c0 + wv; c < c0 + kSlab && c < C; c += 4) {
    float* plane = grad_map + ((size_t)b * C + c) * plane_sz;
    const float g = gout[(size_t)c * P * P + bin_id];
#pragma unroll
    for (int q = 0; q < G * G; ++q)
      if (ok[q]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) atomicAdd(plane + off[q][t], g * wgt[q][t] / count);   // :240-243
      }
  }
}

}  // namespace

hipError_t launch_roi_pool(const RoiPoolArgs& a, hipStream_t s) {
  if (a.pooled < 1 || a.pooled > 8 || a.sampling_ratio < 1 || a.sampling_ratio > 4) return hipErrorInvalidValue;
  const int cmax = a.depth.feat && a.depth_channels > a.channels ? a.depth_channels : a.channels;
  dim3 grid(a.n_roi, (cmax + kSlab - 1) / kSlab, a.depth.feat ? 2 : 1);
  switch (a.sampling_ratio) {
    case 1: VETO_LAUNCH(roi_pool_kernel<1>, grid, dim3(256), 0, s, a); break;
    case 2: VETO_LAUNCH(roi_pool_kernel<2>, grid, dim3(256), 0, s, a); break;
    case 3: VETO_LAUNCH(roi_pool_kernel<3>, grid, dim3(256), 0, s, a); break;
    case 4: VETO_LAUNCH(roi_pool_kernel<4>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_roi_pool_backward(const RoiPoolArgs& a, hipStream_t s) {
  if (a.pooled < 1 || a.pooled > 8 || a.sampling_ratio < 1 || a.sampling_ratio > 4) return hipErrorInvalidValue;
  const bool with_depth = a.depth.feat && a.gout_depth && a.depth_grad;
  const int cmax = with_depth && a.depth_channels > a.channels ? a.depth_channels : a.channels;
  dim3 grid(a.n_roi, (cmax + kSlab - 1) / kSlab, with_depth ? 2 : 1);
  switch (a.sampling_ratio) {
    case 1: VETO_LAUNCH(roi_pool_backward_kernel<1>, grid, dim3(256), 0, s, a); break;
    case 2: VETO_LAUNCH(roi_pool_backward_kernel<2>, grid, dim3(256), 0, s, a); break;
    case 3: VETO_LAUNCH(roi_pool_backward_kernel<3>, grid, dim3(256), 0, s, a); break;
    case 4: VETO_LAUNCH(roi_pool_backward_kernel<4>, grid, dim3(256), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace veto
