"""Eval-time glue of the relation head around the HIP predictor, for the VETO predictors.

Mirrors the caller of the hot path, `ROIRelationHead.forward`
(pysgg/modeling/roi_heads/relation_head/relation_head.py:90-248), for the branch VETO takes in
test mode with GT boxes:
  :104-111  predcls: `predict_logits` overloaded with +-1000 one-hots of the GT labels
            (utils_motifs.py:92-107), `pred_scores` = 1, `pred_labels` = labels
  :134      rel_pair_idxs = samp_processor.prepare_test_pairs(device, proposals)
  :140-141  roi_features, d_2d, _, _ = box_feature_extractor(features, proposals, depth_features=...)
            (VETOFeatureExtractor -> Pooler -> ROIAlign, veto_amd/poolers.py -> veto_roi_pool)
  :196-203  predictor(proposals, rel_pair_idxs, rel_labels, logger, roi_features=, roi_depth_features=)
  :229-230  object refine logits = the proposals' `predict_logits`
  :241-243  post_processor((relation_logits, obj_refine_logits), rel_pair_idxs, proposals, incre_idx_list=...)
Everything numeric runs in libveto_amd.so; this file only moves fields around."""
import torch
from torch import nn

from . import registry
from .pairs import prepare_test_pairs
from .poolers import make_roi_box_feature_extractor
from .postprocess import make_roi_relation_post_processor


def to_onehot(vec, num_classes, fill=1000.0):
    """[n, num_classes] float tensor with +fill at vec[i] and -fill elsewhere (utils_motifs.py:92-107)."""
    out = torch.full((vec.shape[0], num_classes), -fill, dtype=torch.float32, device=vec.device)
    out[torch.arange(vec.shape[0], device=vec.device), vec.long()] = fill
    return out


def _host_samp_processor(cfg):
    """The host code base's own relation sampler, built the way the reference's head builds it (relation_head.py:66-67:
    `self.samp_processor = make_roi_relation_samp_processor(cfg)`), or None when this process has no pysgg to import."""
    try:
        from pysgg.modeling.roi_heads.relation_head.sampling import make_roi_relation_samp_processor
    except Exception:     # not installed / its own imports fail outside the host code base
        return None
    return make_roi_relation_samp_processor(cfg)


class VETORelationHead(nn.Module):
    def __init__(self, cfg, in_channels=512, samp_processor=None):
        """samp_processor: the relation sampler; only the training branch uses it (gtbox_relsample).  Default (None): the host code
        base's own, built from cfg exactly as the reference does (relation_head.py:66-67), so `VETORelationHead(cfg, in_channels)`
        trains inside pysgg like the head it replaces; outside pysgg pass one explicitly (pair sampling is not part of this package)."""
        super().__init__()
        self.cfg = cfg
        rh = cfg.MODEL.ROI_RELATION_HEAD
        if rh.PREDICTOR not in ("VETOPredictor", "VETOPredictor_MEET"):
            raise ValueError("VETORelationHead only drives the VETO predictors, got %r" % (rh.PREDICTOR,))
        if not rh.USE_GT_BOX:
            raise NotImplementedError("sgdet (detected boxes) is outside the built path")
        self.mode = "predcls" if rh.USE_GT_OBJECT_LABEL else "sgcls"
        self.box_feature_extractor = make_roi_box_feature_extractor(cfg, in_channels, for_relation=True)  # :53
        self.predictor = registry.make_roi_relation_predictor(cfg, in_channels)
        self.post_processor = make_roi_relation_post_processor(cfg)
        self.samp_processor = samp_processor if samp_processor is not None else _host_samp_processor(cfg)
        self.num_obj_cls = self.predictor.num_obj_cls
        self.max_proposal_pairs = int(getattr(rh, "MAX_PROPOSAL_PAIR", 2048))

    def forward(self, features, proposals, depth_features=None, targets=None, logger=None, x=None):
        """The reference's signature, parameter for parameter (relation_head.py:90; called as `self.relation(features,
        detections, targets=..., depth_features=..., logger=..., x=x)`, roi_heads.py:69): features = list of FPN maps
        [B, 256, H_l, W_l], depth_features = [B, 256, H/16, W/16], proposals = list[BoxList] (xyxy) on the HIP device; `x` (the
        box head's pooled features, unused by the VETO branch of the reference as well) is accepted and ignored.
        Returns (roi_features, result, {}) like the reference's test branch (:243)."""
        if depth_features is None:
            raise ValueError("the VETO predictors need depth_features (relation_head.py:141)")
        if self.training:
            # :112-121 GT-box relation sampling, :140-141 ROI features, :196-203 predictor -> losses, :247 return
            if targets is None:
                raise ValueError("training needs the targets (GT BoxLists with a 'relation' matrix)")
            if self.samp_processor is None:
                raise ValueError("training needs a relation sampler: pysgg's make_roi_relation_samp_processor(cfg) could not be "
                                 "imported in this process; pass VETORelationHead(cfg, in_channels, samp_processor=...)")
            self._overload_predcls_fields(proposals, features[0].device)
            with torch.no_grad():
                proposals, rel_labels, rel_pair_idxs, _ = self.samp_processor.gtbox_relsample(proposals, targets)
            roi_features, d_2d, _, _ = self.box_feature_extractor(features, proposals, depth_features=depth_features)
            _, _, add_losses, _, _, _ = self.predictor(proposals, rel_pair_idxs, rel_labels, logger, roi_features=roi_features,
                                                       roi_depth_features=d_2d)
            return roi_features, proposals, add_losses
        roi_features, d_2d, _, _ = self.box_feature_extractor(features, proposals, depth_features=depth_features)
        return self.forward_pooled(proposals, roi_features, d_2d, logger)

    def _overload_predcls_fields(self, proposals, device):
        if self.mode != "predcls":
            return
        n_objs = [len(p) for p in proposals]   # :104-111, one batched one-hot instead of one per image
        labels = torch.cat([p.get_field("labels") for p in proposals]).to(device)
        onehot = to_onehot(labels, self.num_obj_cls).split(n_objs)
        ones = torch.ones(labels.shape[0], device=device).split(n_objs)
        for p, oh, sc, lab in zip(proposals, onehot, ones, labels.split(n_objs)):
            p.add_field("predict_logits", oh)
            p.add_field("pred_scores", sc)
            p.add_field("pred_labels", lab)

    def forward_pooled(self, proposals, roi_features, roi_depth_features, logger=None):
        """Same, from already pooled ROI maps [sum N, 256, 8, 8] (the rest of :104-243)."""
        if self.training:
            raise NotImplementedError("forward_pooled is the test-time tail; call forward(features, proposals, targets, ...) to train")
        device = roi_features.device
        self._overload_predcls_fields(proposals, device)
        rel_pair_idxs = prepare_test_pairs(device, proposals, self.max_proposal_pairs)
        obj_dists, relation_logits, add_losses, incre_idx_list, _, _ = self.predictor(
            proposals, rel_pair_idxs, None, logger, roi_features=roi_features, roi_depth_features=roi_depth_features)
        obj_refine_logits = [p.get_field("predict_logits") for p in proposals]
        result = self.post_processor((relation_logits, obj_refine_logits), rel_pair_idxs, proposals,
                                     incre_idx_list=incre_idx_list, ensemble=isinstance(relation_logits, dict))
        return roi_features, result, {}
