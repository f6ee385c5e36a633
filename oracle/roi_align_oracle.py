"""CPU restatement of the ROI feature extraction next to the VETO hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else; the product path is
veto_amd/csrc/roialign.hip behind veto_roi_pool / veto_roi_pool_backward.

PARITY PINNED to the executed reference (round 5): oracle/build_ref.sh compiles the reference's own CPU kernel templates
(pysgg/csrc/cpu/ROIAlign_cpu.cpp:1-219, unmodified, against the installed torch headers; only the 35-line ATen wrapper behind
them targets torch 1.4 and is replaced by a ctypes binding) into oracle/_ref/libroialign_ref.so, and
tests/golden/make_golden.py::run_roialign drives it through the reference's own ROIAlign layer (layers/roi_align.py) and Pooler
(modeling/poolers.py: LevelMapper, convert_to_roi_format, per-level dispatch, fixed 1/16 depth pooler) ->
tests/golden/roialign_single.npz / roialign_pooler.npz.  tests/test_roi_align.py checks this file against them BIT FOR BIT
(five (pooled, sampling ratio) shapes incl. the adaptive grid, out-of-map / sub-pixel / malformed ROIs, all four FPN levels).
The backward exists in the reference only as CUDA (cuda/ROIAlign_cuda.cu:178-262; ROIAlign.h:44 "Not implemented on the CPU"):
its restatement below is pinned by the adjoint identity <pool(f), g> = <f, pool_backward(g)> against the pinned forward.
Kept next to the goldens:
  * analytic known answers in tests/test_roi_align.py: on an affine feature map f(y, x) = a*y + b*x + c
    bilinear sampling is exact, so every interior bin must equal f at the bin centre; constant maps,
    hand-computed border / out-of-map samples, and the FPN level boundaries of the LevelMapper formula;
  * the one published known-answer vector of this kernel lineage (maskrcnn-benchmark's legacy ROIAlign, kept by detectron2 as
    `aligned=False` and pinned there as `old_results`: arange(25) as 5 x 5, box (1, 1, 3, 3), 4 x 4 bins), reproduced exactly.

All arithmetic is float32 in the reference's operation order (numpy does not fuse multiply-adds),
so the HIP kernel, which uses explicitly rounded mul/add, is compared bit for bit.
"""
import numpy as np

F = np.float32


def _axis_samples(start, bin_size, pooled, grid, size):
    """Per-axis sample table of ROIAlign_cuda.cu:21-57 / ROIAlign_cpu.cpp:36-91 for one ROI axis:
    returns (valid[pooled*grid], low, high, l, h) for the sample coordinate
    start + p*bin + (i + .5)*bin/grid, p major, i minor (:104,:108)."""
    n = pooled * grid
    valid = np.zeros(n, dtype=bool)
    low = np.zeros(n, dtype=np.int64)
    high = np.zeros(n, dtype=np.int64)
    l = np.zeros(n, dtype=F)
    h = np.zeros(n, dtype=F)
    for p in range(pooled):
        for i in range(grid):
            v = F(F(start + F(F(p) * bin_size)) + F(F(F(i + 0.5) * bin_size) / F(grid)))
            k = p * grid + i
            if v < F(-1.0) or v > F(size):      # :26-29 the sample lies outside the map: contributes 0
                continue
            valid[k] = True
            if v <= 0:                          # :31-32
                v = F(0)
            lo = int(v)
            if lo >= size - 1:                  # :39-44 the far edge snaps low = high = size-1
                hi = lo = size - 1
                v = F(lo)
            else:
                hi = lo + 1
            low[k], high[k] = lo, hi
            l[k] = F(v - F(lo))                 # :53-55
            h[k] = F(F(1.0) - l[k])
    return valid, low, high, l, h


def roi_align(feat, rois, spatial_scale, pooled=8, sampling_ratio=2):
    """Legacy ROIAlign forward (ROIAlign_cuda.cu:65-125; layers/roi_align.py:12-61).
    feat: [B, C, H, W] float32; rois: [R, 5] float32 rows (batch index, x1, y1, x2, y2) as built by
    Pooler.convert_to_roi_format (poolers.py:96-107).  Returns [R, C, pooled, pooled] float32."""
    feat = np.ascontiguousarray(feat, dtype=F)
    rois = np.asarray(rois, dtype=F)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, pooled, pooled), dtype=F)
    scale = F(spatial_scale)
    for r in range(R):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = (F(rois[r, 1] * scale), F(rois[r, 2] * scale), F(rois[r, 3] * scale), F(rois[r, 4] * scale))  # :84-87
        roi_w = max(F(x2 - x1), F(1.0))          # :95-96 malformed ROIs become 1x1
        roi_h = max(F(y2 - y1), F(1.0))
        bin_h = F(roi_h / F(pooled))             # :97-98
        bin_w = F(roi_w / F(pooled))
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(roi_h / F(pooled)))  # :103-104
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(roi_w / F(pooled)))
        count = F(gh * gw)
        vy, ylo, yhi, ly, hy = _axis_samples(y1, bin_h, pooled, gh, H)
        vx, xlo, xhi, lx, hx = _axis_samples(x1, bin_w, pooled, gw, W)
        plane = feat[b]                          # [C, H, W]
        for ph in range(pooled):
            for pw in range(pooled):
                acc = np.zeros(C, dtype=F)
                for iy in range(gh):
                    ky = ph * gh + iy
                    for ix in range(gw):
                        kx = pw * gw + ix
                        if not (vy[ky] and vx[kx]):
                            continue             # bilinear_interpolate returned 0: acc += 0
                        w1, w2 = F(hy[ky] * hx[kx]), F(hy[ky] * lx[kx])   # :61
                        w3, w4 = F(ly[ky] * hx[kx]), F(ly[ky] * lx[kx])
                        v1 = plane[:, ylo[ky], xlo[kx]]
                        v2 = plane[:, ylo[ky], xhi[kx]]
                        v3 = plane[:, yhi[ky], xlo[kx]]
                        v4 = plane[:, yhi[ky], xhi[kx]]
                        val = ((w1 * v1 + w2 * v2) + w3 * v3) + w4 * v4   # :63, left to right, fp32
                        acc = acc + val          # :115
                out[r, :, ph, pw] = acc / count  # :118
    return out


def roi_align_backward(grad_out, rois, spatial_scale, feat_shape, pooled=8, sampling_ratio=2):
    """RoIAlignBackwardFeature (ROIAlign_cuda.cu:178-262): every output gradient goes to the 4 taps of each of
    its samples as top_diff * w / count (:236-243).  The reference accumulates with float32 atomics in arbitrary
    order; here the products are float32 as there and the accumulation is float64 (the order-free value)."""
    grad_out = np.asarray(grad_out, dtype=F)
    rois = np.asarray(rois, dtype=F)
    B, C, H, W = feat_shape
    grad = np.zeros((B, C, H, W), dtype=np.float64)
    scale = F(spatial_scale)
    for r in range(rois.shape[0]):
        b = int(rois[r, 0])
        x1, y1, x2, y2 = (F(rois[r, 1] * scale), F(rois[r, 2] * scale), F(rois[r, 3] * scale), F(rois[r, 4] * scale))
        roi_w = max(F(x2 - x1), F(1.0))
        roi_h = max(F(y2 - y1), F(1.0))
        bin_h, bin_w = F(roi_h / F(pooled)), F(roi_w / F(pooled))
        gh = gw = sampling_ratio
        count = F(gh * gw)
        vy, ylo, yhi, ly, hy = _axis_samples(y1, bin_h, pooled, gh, H)
        vx, xlo, xhi, lx, hx = _axis_samples(x1, bin_w, pooled, gw, W)
        for ph in range(pooled):
            for pw in range(pooled):
                g = grad_out[r, :, ph, pw]
                for iy in range(gh):
                    ky = ph * gh + iy
                    for ix in range(gw):
                        kx = pw * gw + ix
                        if not (vy[ky] and vx[kx]):
                            continue
                        w = (F(hy[ky] * hx[kx]), F(hy[ky] * lx[kx]), F(ly[ky] * hx[kx]), F(ly[ky] * lx[kx]))
                        taps = ((ylo[ky], xlo[kx]), (ylo[ky], xhi[kx]), (yhi[ky], xlo[kx]), (yhi[ky], xhi[kx]))
                        for wk, (yy, xx) in zip(w, taps):
                            grad[b, :, yy, xx] += (g * wk / count).astype(np.float64)
    return grad


def box_area(boxes):
    """BoxList.area() for xyxy boxes, pysgg/structures/bounding_box.py:249-259: the +1 pixel convention."""
    b = np.asarray(boxes, dtype=F)
    return F(F(b[:, 2] - b[:, 0]) + F(1)) * F(F(b[:, 3] - b[:, 1]) + F(1))


def map_levels(boxes, k_min=2, k_max=5, s0=224, lvl0=4, eps=1e-6):
    """LevelMapper.__call__, pysgg/modeling/poolers.py:17-43: floor(lvl0 + log2(sqrt(area)/s0 + eps))
    clamped to [k_min, k_max], minus k_min.  float32 throughout, as torch computes it."""
    s = np.sqrt(box_area(boxes)).astype(F)
    lv = np.floor(F(lvl0) + np.log2((s / F(s0) + F(eps)).astype(F)).astype(F)).astype(F)
    lv = np.clip(lv, k_min, k_max)
    return lv.astype(np.int64) - k_min


def to_rois(boxes_per_image):
    """Pooler.convert_to_roi_format, poolers.py:96-107."""
    rows = []
    for i, b in enumerate(boxes_per_image):
        b = np.asarray(b, dtype=F)
        rows.append(np.concatenate([np.full((len(b), 1), i, dtype=F), b], axis=1))
    return np.concatenate(rows, axis=0)


def pooler_forward(features, boxes_per_image, depth_features=None, scales=(0.25, 0.125, 0.0625, 0.03125), pooled=8,
                   sampling_ratio=2):
    """Pooler.forward with cat_all_levels=False (poolers.py:109-171), the way VETOFeatureExtractor calls it
    (roi_box_feature_extractors.py:75-121; relation_head.py:53 builds it without cat_all_levels).
    features: list of per-level [B, C, H_l, W_l]; every ROI is pooled from ITS level (LevelMapper) with that
    level's scale; the depth map is always pooled with poolers[2] (scale 1/16, :144-153) when there are
    several levels, with poolers[0] otherwise.  Returns (rgb [R, C, 8, 8], depth [R, Cd, 8, 8] or None)."""
    rois = to_rois(boxes_per_image)
    if len(scales) == 1:
        rgb = roi_align(features[0], rois, scales[0], pooled, sampling_ratio)
        dep = roi_align(depth_features, rois, scales[0], pooled, sampling_ratio) if depth_features is not None else None
        return rgb, dep
    k_min = int(round(-np.log2(scales[0])))
    k_max = int(round(-np.log2(scales[-1])))
    levels = map_levels(np.concatenate([np.asarray(b, dtype=F) for b in boxes_per_image]), k_min, k_max)
    C = features[0].shape[1]
    rgb = np.zeros((len(rois), C, pooled, pooled), dtype=F)
    for lvl, (feat, sc) in enumerate(zip(features, scales)):
        idx = np.nonzero(levels == lvl)[0]
        if len(idx):
            rgb[idx] = roi_align(feat, rois[idx], sc, pooled, sampling_ratio)
    dep = None
    if depth_features is not None:
        dep = roi_align(depth_features, rois, scales[2], pooled, sampling_ratio)
    return rgb, dep
