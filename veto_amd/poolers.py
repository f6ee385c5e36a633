"""MI355X-native ROI feature extraction for the VETO relation head.

Host-side mirror of the reference's
  * `ROIAlign` module          pysgg/layers/roi_align.py:49-68
  * `LevelMapper`, `Pooler`    pysgg/modeling/poolers.py:12-171 (cat_all_levels=False, the way VETO builds it:
                               relation_head.py:53)
  * `VETOFeatureExtractor`     pysgg/modeling/roi_heads/box_head/roi_box_feature_extractors.py:75-121
with the same constructor arguments, call signatures and return values.  All arithmetic runs in
libveto_amd.so (veto_roi_pool): one launch pools every ROI from its own FPN level and the depth map with
the fixed 1/16 pooler; differentiable w.r.t. the maps (veto_roi_pool_backward, the atomic-add scatter of
ROIAlign_cuda.cu:178-262).  Union pooling
(7x7, Pooler.forward(union=True)) and cat_all_levels=True are not used by VETO and raise."""
import ctypes
import math

import torch
from torch import nn

from . import native


def _fill_args(feats, scales, rois, n_img, pooled, sampling_ratio, depth_shape):
    a = native.VetoRoiPoolArgs()
    a.struct_size = ctypes.sizeof(native.VetoRoiPoolArgs)
    a.n_levels, a.n_img, a.n_roi, a.channels = len(feats), n_img, rois.shape[0], feats[0][1]
    a.pooled, a.sampling_ratio = pooled, sampling_ratio
    for l, (shape, sc) in enumerate(zip(feats, scales)):
        a.level_h[l], a.level_w[l], a.level_scale[l] = shape[2], shape[3], float(sc)
    if depth_shape is not None:
        a.depth_channels, a.depth_h, a.depth_w = depth_shape[1], depth_shape[2], depth_shape[3]
    a.rois = rois.data_ptr()
    return a


def _roi_pool(level_feats, scales, rois, n_img, pooled, sampling_ratio, depth=None, want_levels=False):
    lib = native.load_library()
    device = rois.device
    if device.type != "cuda":
        raise RuntimeError("veto_amd ROI pooling runs only on a HIP device (got %s)" % device)
    f32 = dict(device=device, dtype=torch.float32)
    feats = [f.detach().to(**f32).contiguous() for f in level_feats]
    rois = rois.detach().to(**f32).contiguous()
    n_roi, C = rois.shape[0], feats[0].shape[1]
    for l, f in enumerate(feats):
        if f.shape[0] != n_img or f.shape[1] != C:
            raise ValueError("pyramid level %d has shape %s, expected [%d, %d, H, W]" % (l, tuple(f.shape), n_img, C))
    if depth is not None:
        depth = depth.detach().to(**f32).contiguous()
    a = _fill_args([tuple(f.shape) for f in feats], scales, rois, n_img, pooled, sampling_ratio,
                   tuple(depth.shape) if depth is not None else None)
    for l, f in enumerate(feats):
        a.level_feat[l] = f.data_ptr()
    out_rgb = torch.empty((n_roi, C, pooled, pooled), **f32)
    out_depth = None
    if depth is not None:
        a.depth_feat = depth.data_ptr()
        out_depth = torch.empty((n_roi, depth.shape[1], pooled, pooled), **f32)
        a.out_depth = out_depth.data_ptr()
    levels = torch.empty(n_roi, dtype=torch.int32, device=device) if want_levels else None
    a.out_rgb = out_rgb.data_ptr()
    a.out_levels = levels.data_ptr() if want_levels else None
    stream = torch.cuda.current_stream(device)
    native.check(lib.veto_roi_pool(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(a)))
    for t in feats + [rois] + ([depth] if depth is not None else []):
        t.record_stream(stream)
    return out_rgb, out_depth, levels


def _roi_pool_backward(shapes, scales, rois, n_img, pooled, sampling_ratio, grad_rgb, depth_shape, grad_depth):
    """Map gradients of one veto_roi_pool call: (list of per-level gradients, depth gradient or None)."""
    lib = native.load_library()
    device = rois.device
    f32 = dict(device=device, dtype=torch.float32)
    rois = rois.detach().to(**f32).contiguous()
    a = _fill_args(shapes, scales, rois, n_img, pooled, sampling_ratio, depth_shape if grad_depth is not None else None)
    grad_rgb = grad_rgb.detach().to(**f32).contiguous()
    level_grads = [torch.zeros(shape, **f32) for shape in shapes]
    ptrs = (ctypes.c_void_p * 4)(*[g.data_ptr() for g in level_grads])
    depth_grad = None
    if grad_depth is not None:
        grad_depth = grad_depth.detach().to(**f32).contiguous()
        depth_grad = torch.zeros(depth_shape, **f32)
    stream = torch.cuda.current_stream(device)
    native.check(lib.veto_roi_pool_backward(
        ctypes.c_void_p(stream.cuda_stream), ctypes.byref(a), ctypes.c_void_p(grad_rgb.data_ptr()),
        ctypes.c_void_p(grad_depth.data_ptr()) if grad_depth is not None else None, ptrs,
        ctypes.c_void_p(depth_grad.data_ptr()) if depth_grad is not None else None))
    for t in [rois, grad_rgb] + ([grad_depth] if grad_depth is not None else []):
        t.record_stream(stream)
    return level_grads, depth_grad


class _RoiPoolFn(torch.autograd.Function):
    """layers/roi_align.py:12-44 (_ROIAlign) for the fused multi-level + depth call: differentiable w.r.t. the maps."""

    @staticmethod
    def forward(ctx, rois, scales, n_img, pooled, sampling_ratio, want_levels, n_levels, *maps):
        feats, depth = list(maps[:n_levels]), (maps[n_levels] if len(maps) > n_levels else None)
        rgb, dep, levels = _roi_pool(feats, scales, rois, n_img, pooled, sampling_ratio, depth=depth, want_levels=want_levels)
        ctx.save_for_backward(rois)
        ctx.meta = ([tuple(f.shape) for f in feats], list(scales), n_img, pooled, sampling_ratio,
                    tuple(depth.shape) if depth is not None else None)
        ctx.mark_non_differentiable(*([levels] if levels is not None else []))
        outs = (rgb,) + ((dep,) if dep is not None else ()) + ((levels,) if levels is not None else ())
        return outs

    @staticmethod
    def backward(ctx, *grads):
        (rois,) = ctx.saved_tensors
        shapes, scales, n_img, pooled, sampling_ratio, depth_shape = ctx.meta
        grad_rgb = grads[0]
        grad_dep = grads[1] if depth_shape is not None else None
        if grad_rgb is None:
            grad_rgb = torch.zeros((rois.shape[0], shapes[0][1], pooled, pooled), device=rois.device)
        level_grads, depth_grad = _roi_pool_backward(shapes, scales, rois, n_img, pooled, sampling_ratio, grad_rgb,
                                                     depth_shape, grad_dep)
        return (None,) * 7 + tuple(level_grads) + ((depth_grad,) if depth_shape is not None else ())


def _roi_pool_autograd(level_feats, scales, rois, n_img, pooled, sampling_ratio, depth=None, want_levels=False):
    """_roi_pool, recorded on the autograd tape when a map requires grad."""
    maps = list(level_feats) + ([depth] if depth is not None else [])
    if not (torch.is_grad_enabled() and any(m.requires_grad for m in maps)):
        return _roi_pool(level_feats, scales, rois, n_img, pooled, sampling_ratio, depth=depth, want_levels=want_levels)
    outs = _RoiPoolFn.apply(rois, tuple(scales), n_img, pooled, sampling_ratio, want_levels, len(level_feats), *maps)
    rgb = outs[0]
    dep = outs[1] if depth is not None else None
    levels = outs[-1] if want_levels else None
    return rgb, dep, levels


class ROIAlign(nn.Module):
    """layers/roi_align.py:49-68: ROIAlign(output_size, spatial_scale, sampling_ratio)(input, rois)."""

    def __init__(self, output_size, spatial_scale, sampling_ratio):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        if self.output_size[0] != self.output_size[1]:
            raise NotImplementedError("veto_amd ROIAlign: square outputs only (VETO pools 8x8)")

    def forward(self, input, rois):
        return _roi_pool_autograd([input], [self.spatial_scale], rois, input.shape[0], self.output_size[0], self.sampling_ratio)[0]

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)


class LevelMapper(object):
    """poolers.py:12-43.  Kept for callers that want the levels; Pooler.forward computes them in the kernel."""

    def __init__(self, k_min, k_max, canonical_scale=224, canonical_level=4, eps=1e-6):
        self.k_min, self.k_max, self.s0, self.lvl0, self.eps = k_min, k_max, canonical_scale, canonical_level, eps

    def __call__(self, boxlists):
        s = torch.sqrt(torch.cat([b.area() for b in boxlists]))
        lv = torch.floor(self.lvl0 + torch.log2(s / self.s0 + self.eps))
        return torch.clamp(lv, min=self.k_min, max=self.k_max).to(torch.int64) - self.k_min


class Pooler(nn.Module):
    """poolers.py:46-171 with cat_all_levels=False."""

    def __init__(self, output_size, scales, sampling_ratio, in_channels=512, cat_all_levels=False):
        super().__init__()
        if cat_all_levels:
            raise NotImplementedError("veto_amd Pooler: cat_all_levels=True (reduce_channel conv) is not the VETO path "
                                      "(relation_head.py:53 builds the extractor without it)")
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.scales = [float(s) for s in scales]
        self.sampling_ratio = sampling_ratio
        self.cat_all_levels = False
        self.poolers = nn.ModuleList(ROIAlign(self.output_size, s, sampling_ratio) for s in self.scales)
        lvl_min = -math.log2(self.scales[0])
        lvl_max = -math.log2(self.scales[-1])
        self.map_levels = LevelMapper(lvl_min, lvl_max)
        if not 1 <= len(self.scales) <= 4:
            raise ValueError("1 to 4 pooler scales are supported, got %d" % len(self.scales))

    def convert_to_roi_format(self, boxes):
        """poolers.py:96-107: rows (image index, x1, y1, x2, y2)."""
        concat = torch.cat([b.bbox for b in boxes], dim=0)
        ids = torch.cat([torch.full((len(b), 1), i, dtype=concat.dtype, device=concat.device) for i, b in enumerate(boxes)], dim=0)
        return torch.cat([ids, concat], dim=1)

    def forward(self, x, boxes, depth_features=None, union=False):
        if union:
            raise NotImplementedError("veto_amd Pooler: union pooling (7x7) is not used by the VETO predictors")
        for b in boxes:
            if b.mode != "xyxy":
                raise ValueError("Pooler expects xyxy boxes (poolers.py:96-107 passes BoxList.bbox through), got %s" % b.mode)
        rois = self.convert_to_roi_format(boxes)
        assert rois.size(0) > 0
        if len(x) != len(self.scales):
            raise ValueError("%d feature levels for %d pooler scales" % (len(x), len(self.scales)))
        rgb, depth, levels = _roi_pool_autograd(list(x), self.scales, rois, x[0].shape[0], self.output_size[0],
                                                self.sampling_ratio, depth=depth_features,
                                                want_levels=getattr(self, "keep_levels", False))
        self.last_levels = levels
        if depth_features is not None:
            return rgb, depth
        return rgb


class VETOFeatureExtractor(nn.Module):
    """roi_box_feature_extractors.py:75-121: forward(x, proposals, depth_features) ->
    (x_2d, d_2d, None, None), the pooled [sum N, 256, 8, 8] RGB / depth ROI maps."""

    def __init__(self, cfg, in_channels, half_out=False, cat_all_levels=False, for_relation=False):
        super().__init__()
        resolution = cfg.MODEL.ROI_RELATION_HEAD.POOLER_RESOLUTION
        scales = cfg.MODEL.ROI_BOX_HEAD.POOLER_SCALES
        sampling_ratio = cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO
        self.pooler = Pooler(output_size=(resolution, resolution), scales=scales, sampling_ratio=sampling_ratio,
                             in_channels=in_channels, cat_all_levels=cat_all_levels)
        self.out_channels = 256

    def forward(self, x, proposals, depth_features=None):
        x_2d = d_2d = None
        if depth_features is not None:
            x_2d, d_2d = self.pooler(x, proposals, depth_features=depth_features)
        else:
            x = self.pooler(x, proposals, depth_features=depth_features)   # :112 (result unused by the reference too)
        return x_2d, d_2d, None, None


def make_roi_box_feature_extractor(cfg, in_channels, half_out=False, cat_all_levels=False, for_relation=False):
    """roi_box_feature_extractors.py:315-323: the relation head asks with for_relation=True and gets
    ROI_RELATION_HEAD.FEATURE_EXTRACTOR_MINI ("VETOFeatureExtractor", VETO_final.yaml:71) when depth is on."""
    if for_relation and getattr(cfg.MODEL.ROI_RELATION_HEAD, "FEATURE_EXTRACTOR_MINI", None) is not None \
            and cfg.DATASETS.USE_DEPTH:
        name = cfg.MODEL.ROI_RELATION_HEAD.FEATURE_EXTRACTOR_MINI
    else:
        name = cfg.MODEL.ROI_BOX_HEAD.FEATURE_EXTRACTOR
    if name != "VETOFeatureExtractor":
        raise ValueError("veto_amd only provides VETOFeatureExtractor, got %r" % (name,))
    return VETOFeatureExtractor(cfg, in_channels, half_out, cat_all_levels, for_relation)
