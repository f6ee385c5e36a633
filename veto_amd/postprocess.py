"""MI355X-native relation PostProcessor.

Mirrors the vanilla, GT-box branch of the reference's PostProcessor
(pysgg/modeling/roi_heads/relation_head/inference.py:9-92,398-453): same constructor arguments, same
`forward(x, rel_pair_idxs, boxes)` call, and the same BoxList fields on the results
(`pred_labels`, `pred_scores`, `rel_pair_idxs`, `pred_rel_scores`, `pred_rel_labels`).  The
arithmetic (softmax, foreground max, triple score, per-image descending sort, gather) runs in
libveto_amd.so (veto_postprocess).  When the relation logits are the MEET dict of group heads
(ENSEMBLE_LEARNING.ENABLED with EXPERT_GROUP False), the MEET merge branch (inference.py:284-397) runs
through veto_postprocess_meet with the reference's quirks kept: one image per call, group-local
labels, float pair indices.  With EXPERT_GROUP True (three expert heads per group, keys 'group_<k><e>') the
voting branch (inference.py:93-283, ENSEMBLE_LEARNING.VOTING 'C' or 'U') runs through veto_postprocess_vote;
its row count is data dependent, so that branch reads one int32 back from the device.  sgdet decoding
(per-class NMS, inference.py:413-418) and attributes are not built; they raise."""
import ctypes

import torch
from torch import nn

from . import native
from .predictor import cached_offsets


class PostProcessor(nn.Module):
    def __init__(self, attribute_on, use_gt_box=False, later_nms_pred_thres=0.3, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.attribute_on = attribute_on
        self.use_gt_box = use_gt_box
        self.later_nms_pred_thres = later_nms_pred_thres
        self._workspace = None

    def forward(self, x, rel_pair_idxs, boxes, custom_rel_labels=None, cur_chosen_matrix=None, incre_idx_list=None,
                ensemble=False):
        relation_logits, refine_logits = x
        if self.attribute_on:
            raise NotImplementedError("veto_amd.PostProcessor: attribute head is outside the VETO path")
        if not self.use_gt_box:
            raise NotImplementedError("veto_amd.PostProcessor: sgdet decoding (per-class NMS) is not built")
        if isinstance(relation_logits, dict):
            if "group_01" in relation_logits:   # inference.py:93: three experts per group
                return self._forward_vote(relation_logits, refine_logits, rel_pair_idxs, boxes, incre_idx_list)
            return self._forward_meet(relation_logits, refine_logits, rel_pair_idxs, boxes, incre_idx_list)
        rel = torch.cat(list(relation_logits), 0) if isinstance(relation_logits, (list, tuple)) else relation_logits
        obj = torch.cat(list(refine_logits), 0) if isinstance(refine_logits, (list, tuple)) else refine_logits
        device = rel.device
        if device.type != "cuda":
            raise RuntimeError("veto_amd.PostProcessor runs only on a HIP device (got %s)" % device)
        lib = native.load_library()
        n_objs = [len(b) for b in boxes]
        n_pairs = [int(p.shape[0]) for p in rel_pair_idxs]
        n_obj, n_pair = sum(n_objs), sum(n_pairs)
        f32 = dict(device=device, dtype=torch.float32)
        rel = rel.detach().to(**f32).contiguous()
        obj = obj.detach().to(**f32).contiguous()
        if rel.shape[0] != n_pair or obj.shape[0] != n_obj:
            raise ValueError("logit rows (%d, %d) do not match pairs/objects (%d, %d)" % (rel.shape[0], obj.shape[0], n_pair, n_obj))
        pairs = torch.cat([p.reshape(-1, 2) for p in rel_pair_idxs], 0).to(device=device, dtype=torch.int64).contiguous()
        obj_off, pair_off = cached_offsets(n_objs, n_pairs, device)   # no host-blocking H2D copy in the steady state
        out = {
            "obj_scores": torch.empty(n_obj, **f32), "obj_pred": torch.empty(n_obj, dtype=torch.int64, device=device),
            "prob": torch.empty((n_pair, rel.shape[1]), **f32),
            "pairs": torch.empty((n_pair, 2), dtype=torch.int64, device=device),
            "labels": torch.empty(n_pair, dtype=torch.int64, device=device), "triple": torch.empty(n_pair, **f32),
        }
        need = lib.veto_postprocess_workspace_bytes(n_pair, rel.shape[1])
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != device:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        a = native.VetoPostArgs()
        a.struct_size = ctypes.sizeof(native.VetoPostArgs)
        a.n_img, a.n_obj, a.n_pair = len(boxes), n_obj, n_pair
        a.n_rel_cls, a.n_obj_cls, a.max_pairs_per_image = rel.shape[1], obj.shape[1], max(n_pairs)
        a.rel_logits, a.obj_logits, a.rel_pairs = rel.data_ptr(), obj.data_ptr(), pairs.data_ptr()
        a.img_obj_offset, a.img_pair_offset = obj_off.data_ptr(), pair_off.data_ptr()
        a.obj_scores, a.obj_pred = out["obj_scores"].data_ptr(), out["obj_pred"].data_ptr()
        a.rel_prob_sorted, a.rel_pairs_sorted = out["prob"].data_ptr(), out["pairs"].data_ptr()
        a.rel_labels_sorted, a.triple_sorted = out["labels"].data_ptr(), out["triple"].data_ptr()
        stream = torch.cuda.current_stream(device)
        native.check(lib.veto_postprocess(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(a),
                                          ctypes.c_void_p(self._workspace.data_ptr()), self._workspace.numel()))
        for t in (rel, obj, pairs, obj_off, pair_off):
            t.record_stream(stream)
        self.last_triple_scores = out["triple"].split(n_pairs)
        results = []
        for box, sc, pr, prob, pidx, lab in zip(boxes, out["obj_scores"].split(n_objs), out["obj_pred"].split(n_objs),
                                                out["prob"].split(n_pairs), out["pairs"].split(n_pairs),
                                                out["labels"].split(n_pairs)):
            box.add_field("pred_labels", pr)       # inference.py:431-432 (the GT-box branch re-uses `box`)
            box.add_field("pred_scores", sc)
            box.add_field("rel_pair_idxs", pidx)   # :450-452
            box.add_field("pred_rel_scores", prob)
            box.add_field("pred_rel_labels", lab)
            results.append(box)
        return results


    def _forward_meet(self, relation_logits, refine_logits, rel_pair_idxs, boxes, incre_idx_list):
        if incre_idx_list is None:
            raise ValueError("the MEET merge needs incre_idx_list (4th element of the predictor's return tuple)")
        if len(boxes) != 1:
            raise ValueError("the MEET merge (inference.py:303-306) pairs the batch-wide group logits with the first "
                             "image only; call it with one image per batch, got %d" % len(boxes))
        lib = native.load_library()
        keys = ["group_%d" % k for k in range(len(relation_logits))]
        device = relation_logits[keys[0]].device
        if device.type != "cuda":
            raise RuntimeError("veto_amd.PostProcessor runs only on a HIP device (got %s)" % device)
        f32 = dict(device=device, dtype=torch.float32)
        groups = [relation_logits[k].detach().to(**f32).contiguous() for k in keys]
        obj = (refine_logits[0] if isinstance(refine_logits, (list, tuple)) else refine_logits).detach().to(**f32).contiguous()
        pairs = rel_pair_idxs[0].reshape(-1, 2).to(device=device, dtype=torch.int64).contiguous()
        n_obj, n_pair, K, n_rel = obj.shape[0], pairs.shape[0], len(groups), len(incre_idx_list)
        total = K * n_pair
        out = {"obj_scores": torch.empty(n_obj, **f32), "obj_pred": torch.empty(n_obj, dtype=torch.int64, device=device),
               "prob": torch.empty((total, n_rel), **f32), "pairs": torch.empty((total, 2), dtype=torch.int64, device=device),
               "labels": torch.empty(total, dtype=torch.int64, device=device), "triple": torch.empty(total, **f32)}
        need = lib.veto_postprocess_workspace_bytes(total, n_rel)
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != device:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        ptrs = (ctypes.c_void_p * K)(*[g.data_ptr() for g in groups])
        widths = (ctypes.c_int32 * K)(*[g.shape[1] for g in groups])
        incre = (ctypes.c_int32 * n_rel)(*[int(x) for x in incre_idx_list])
        a = native.VetoPostMeetArgs()
        a.struct_size = ctypes.sizeof(native.VetoPostMeetArgs)
        a.n_obj, a.n_pair, a.n_groups, a.n_rel_cls, a.n_obj_cls = n_obj, n_pair, K, n_rel, obj.shape[1]
        a.group_logits = ctypes.cast(ptrs, ctypes.c_void_p)
        a.group_widths = ctypes.cast(widths, ctypes.c_void_p)
        a.incre_idx_list = ctypes.cast(incre, ctypes.c_void_p)
        a.obj_logits, a.rel_pairs = obj.data_ptr(), pairs.data_ptr()
        a.obj_scores, a.obj_pred = out["obj_scores"].data_ptr(), out["obj_pred"].data_ptr()
        a.rel_prob_sorted, a.rel_pairs_sorted = out["prob"].data_ptr(), out["pairs"].data_ptr()
        a.rel_labels_sorted, a.triple_sorted = out["labels"].data_ptr(), out["triple"].data_ptr()
        stream = torch.cuda.current_stream(device)
        native.check(lib.veto_postprocess_meet(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(a),
                                               ctypes.c_void_p(self._workspace.data_ptr()), self._workspace.numel()))
        for t in groups + [obj, pairs]:
            t.record_stream(stream)
        self.last_triple_scores = [out["triple"]]
        box = boxes[0]
        box.add_field("pred_labels", out["obj_pred"])
        box.add_field("pred_scores", out["obj_scores"])
        box.add_field("rel_pair_idxs", out["pairs"].to(torch.float32))  # torch.zeros(total, 2) in the reference (:381)
        box.add_field("pred_rel_scores", out["prob"])
        box.add_field("pred_rel_labels", out["labels"])                 # group-local labels, as the reference (:388)
        return [box]


    def _forward_vote(self, relation_logits, refine_logits, rel_pair_idxs, boxes, incre_idx_list):
        if incre_idx_list is None:
            raise ValueError("expert voting needs incre_idx_list (4th element of the predictor's return tuple)")
        if len(boxes) != 1:
            raise ValueError("the EXPERT_GROUP branch (inference.py:114-116) uses the first image only; call it with "
                             "one image per batch, got %d" % len(boxes))
        voting = str(self.cfg.ENSEMBLE_LEARNING.VOTING) if self.cfg is not None else "C"
        if voting not in ("C", "U"):
            raise ValueError("ENSEMBLE_LEARNING.VOTING must be 'C' or 'U', got %r" % voting)
        if len(relation_logits) % 3:
            raise ValueError("expected three expert heads per group, got %d heads" % len(relation_logits))
        lib = native.load_library()
        K = len(relation_logits) // 3
        keys = ["group_%d%d" % (k, e + 1) for k in range(K) for e in range(3)]
        device = relation_logits[keys[0]].device
        if device.type != "cuda":
            raise RuntimeError("veto_amd.PostProcessor runs only on a HIP device (got %s)" % device)
        f32 = dict(device=device, dtype=torch.float32)
        heads = [relation_logits[k].detach().to(**f32).contiguous() for k in keys]
        obj = (refine_logits[0] if isinstance(refine_logits, (list, tuple)) else refine_logits).detach().to(**f32).contiguous()
        pairs = rel_pair_idxs[0].reshape(-1, 2).to(device=device, dtype=torch.int64).contiguous()
        n_obj, n_pair, n_rel = obj.shape[0], pairs.shape[0], len(incre_idx_list)
        total = K * n_pair
        out = {"obj_scores": torch.empty(n_obj, **f32), "obj_pred": torch.empty(n_obj, dtype=torch.int64, device=device),
               "prob": torch.empty((total, n_rel), **f32), "pairs": torch.empty((total, 2), dtype=torch.int64, device=device),
               "labels": torch.empty(total, dtype=torch.int64, device=device), "triple": torch.empty(total, **f32),
               "kept": torch.zeros(1, dtype=torch.int32, device=device)}
        need = lib.veto_postprocess_workspace_bytes(total, n_rel)
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != device:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        ptrs = (ctypes.c_void_p * (3 * K))(*[h.data_ptr() for h in heads])
        widths = (ctypes.c_int32 * K)(*[heads[3 * k].shape[1] for k in range(K)])
        incre = (ctypes.c_int32 * n_rel)(*[int(x) for x in incre_idx_list])
        a = native.VetoPostVoteArgs()
        a.struct_size = ctypes.sizeof(native.VetoPostVoteArgs)
        a.n_obj, a.n_pair, a.n_groups, a.n_rel_cls, a.n_obj_cls = n_obj, n_pair, K, n_rel, obj.shape[1]
        a.voting = 0 if voting == "C" else 1
        a.expert_logits = ctypes.cast(ptrs, ctypes.c_void_p)
        a.group_widths = ctypes.cast(widths, ctypes.c_void_p)
        a.incre_idx_list = ctypes.cast(incre, ctypes.c_void_p)
        a.obj_logits, a.rel_pairs = obj.data_ptr(), pairs.data_ptr()
        a.obj_scores, a.obj_pred = out["obj_scores"].data_ptr(), out["obj_pred"].data_ptr()
        a.rel_prob_sorted, a.rel_pairs_sorted = out["prob"].data_ptr(), out["pairs"].data_ptr()
        a.rel_labels_sorted, a.triple_sorted = out["labels"].data_ptr(), out["triple"].data_ptr()
        a.kept_count = out["kept"].data_ptr()
        stream = torch.cuda.current_stream(device)
        native.check(lib.veto_postprocess_vote(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(a),
                                               ctypes.c_void_p(self._workspace.data_ptr()), self._workspace.numel()))
        for t in heads + [obj, pairs]:
            t.record_stream(stream)
        kept = int(out["kept"].item())   # the one device read-back: the result's row count is data dependent
        self.last_triple_scores = [out["triple"][:kept]]
        box = boxes[0]
        box.add_field("pred_labels", out["obj_pred"])
        box.add_field("pred_scores", out["obj_scores"])
        box.add_field("rel_pair_idxs", out["pairs"][:kept].to(torch.float32))  # float, as the reference (:267)
        box.add_field("pred_rel_scores", out["prob"][:kept])
        box.add_field("pred_rel_labels", out["labels"][:kept])                 # group-local labels
        return [box]


def make_roi_relation_post_processor(cfg):
    """inference.py:456-468."""
    return PostProcessor(getattr(cfg.MODEL, "ATTRIBUTE_ON", False), cfg.MODEL.ROI_RELATION_HEAD.USE_GT_BOX,
                         cfg.TEST.RELATION.LATER_NMS_PREDICTION_THRES, cfg)
