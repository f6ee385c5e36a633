"""MI355X-native relation evaluation.

Host-side counterpart of the reference's per-image routine `evaluate_relation_of_one_image`
(pysgg/data/datasets/evaluation/vg/vg_eval.py:459-566) and of the evaluator classes it drives
(sgg_eval.py: SGRecall, SGNoGraphConstraintRecall, SGZeroShotRecall, SGPairAccuracy, SGMeanRecall,
SGNGMeanRecall) for the GT-box modes predcls / sgcls.  The reference pulls every BoxList to the host and loops
in numpy; here the sorted predictions stay on the device, one launch (veto_sgg_eval) scores all images of the
batch, and only the final numbers (and, for inspection, the per-relation match ranks) come back.

`SGGEvaluator.evaluate(images)` takes per-image dicts with the reference's local_container names
(gt_rels, gt_classes, gt_boxes, pred_rel_inds, rel_scores, pred_classes, pred_boxes, obj_scores);
`evaluate_boxlists(groundtruths, predictions)` takes the BoxLists the reference's loop is fed with.
The result dict follows the reference's result_dict ('<mode>_recall' -> {20, 50, 100}, ...)."""
import ctypes

import numpy as np
import torch

from . import native
from .predictor import cached_offsets

KS = (20, 50, 100)
NO_MATCH = 0x3fffffff


def _t(x, dtype, device):
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x))
    return x.to(device=device, dtype=dtype)


class SGGEvaluator:
    def __init__(self, mode, num_rel_category, zeroshot_triplet, iou_thres=0.5, device="cuda"):
        if mode not in ("predcls", "sgcls"):
            raise NotImplementedError("veto_amd.SGGEvaluator covers the GT-box modes predcls / sgcls, got %r" % (mode,))
        self.mode, self.num_rel, self.iou_thres = mode, int(num_rel_category), float(iou_thres)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("veto_amd.SGGEvaluator runs only on a HIP device (got %s)" % self.device)
        self.zeroshot = _t(zeroshot_triplet, torch.int64, self.device).reshape(-1, 3).contiguous()
        self._workspace = None
        self.reset()

    # ---- dataset-level accumulation (the reference's evaluators keep per-image lists over the WHOLE split and fold them once,
    # sgg_eval.py:133-136,209,331-336,420-466; per-batch results cannot be averaged into mR@K / A@K afterwards) --------------
    def reset(self):
        """Forgets everything accumulated by update()."""
        self._acc = []        # per evaluated image: (gc_rank, ng_rank, acc_rank, zeroshot flags, GT predicate of every GT relation)

    def update(self, images):
        """Scores one batch on the device (same launch as evaluate()) and keeps its per-relation match ranks for finalize().
        Returns the batch's own result dict."""
        res = self.evaluate(images)
        for im, r in zip(images, res["per_image"]):
            if r is None:
                continue
            gt_pred = _t(im["gt_rels"], torch.int64, "cpu").reshape(-1, 3)[:, 2].numpy()
            self._acc.append((r["gc_rank"], r["ng_rank"], r["acc_rank"], r["zeroshot"], gt_pred))
        return res

    def update_boxlists(self, groundtruths, predictions):
        res = self.evaluate_boxlists(groundtruths, predictions)
        for gt, r in zip(groundtruths, res["per_image"]):
            if r is None:
                continue
            gt_pred = _t(gt.get_field("relation_tuple"), torch.int64, "cpu").reshape(-1, 3)[:, 2].numpy()
            self._acc.append((r["gc_rank"], r["ng_rank"], r["acc_rank"], r["zeroshot"], gt_pred))
        return res

    def finalize(self):
        """Folds every image seen by update() the way the reference does at the end of the split: np.mean of the per-image
        recalls, A@K = mean(hits) / mean(counts), mean recall from the per-image per-class hit ratios."""
        C = self.num_rel
        res = {"images_evaluated": len(self._acc)}
        lists = {n: {k: [] for k in KS} for n in ("recall", "recall_nogc", "zeroshot_recall", "acc_hit", "acc_cnt")}
        coll = {n: {k: [[] for _ in range(C)] for k in KS} for n in ("mean_recall", "ng_mean_recall")}
        n_zs = 0
        for gc, ng, ac, zs, gt_pred in self._acc:
            G = len(gc)
            n_zs += bool(zs.any())
            for k in KS:
                hit, nghit = gc < k, ng < k
                lists["recall"][k].append(float(hit.sum()) / float(G))
                lists["recall_nogc"][k].append(float(nghit.sum()) / float(G))
                if zs.any():
                    lists["zeroshot_recall"][k].append(float((hit & zs).sum()) / float(zs.sum()))
                lists["acc_hit"][k].append(float((ac < k).sum()))
                lists["acc_cnt"][k].append(float(G))
                for name, h in (("mean_recall", hit), ("ng_mean_recall", nghit)):
                    cnt = np.bincount(gt_pred, minlength=C)
                    hc = np.bincount(gt_pred[h], minlength=C)
                    for n in np.nonzero(cnt[1:])[0] + 1:
                        coll[name][k][n].append(float(hc[n]) / float(cnt[n]))
        res["images_with_zeroshot"] = n_zs
        for name in ("recall", "recall_nogc", "zeroshot_recall"):
            res[name] = {k: (float(np.mean(v)) if len(v) else float("nan")) for k, v in lists[name].items()}
            res[name + "_list"] = lists[name]
        res["accuracy"] = {k: (float(np.mean(lists["acc_hit"][k]) / np.mean(lists["acc_cnt"][k])) if lists["acc_cnt"][k] else float("nan"))
                           for k in KS}
        for name in ("mean_recall", "ng_mean_recall"):
            res[name + "_list"] = {k: [float(np.mean(coll[name][k][n + 1])) if coll[name][k][n + 1] else 0.0 for n in range(C - 1)]
                                   for k in KS}
            res[name] = {k: sum(res[name + "_list"][k]) / float(C - 1) for k in KS}
        return res

    def evaluate_boxlists(self, groundtruths, predictions):
        """vg_eval.py:470-498: unpack the fields of the GT / prediction BoxLists."""
        images = []
        for gt, pr in zip(groundtruths, predictions):
            images.append({
                "gt_rels": gt.get_field("relation_tuple"), "gt_classes": gt.get_field("labels"), "gt_boxes": gt.convert("xyxy").bbox,
                "pred_rel_inds": pr.get_field("rel_pair_idxs"), "rel_scores": pr.get_field("pred_rel_scores"),
                "pred_classes": pr.get_field("pred_labels"), "pred_boxes": pr.convert("xyxy").bbox,
                "obj_scores": pr.get_field("pred_scores")})
        return self.evaluate(images)

    def evaluate(self, images):
        dev, lib = self.device, native.load_library()
        i64, f32, i32 = torch.int64, torch.float32, torch.int32
        def cat(key, dtype, shape):
            t = torch.cat([_t(im[key], dtype, dev).reshape(shape) for im in images], 0).contiguous()
            # an all-empty field still needs a valid device pointer for the ABI's argument check
            return t if t.numel() else torch.zeros((1,) + tuple(abs(d) for d in shape[1:]), dtype=dtype, device=dev)
        gt_rels = cat("gt_rels", i64, (-1, 3))
        gt_classes, gt_boxes = cat("gt_classes", i64, (-1,)), cat("gt_boxes", f32, (-1, 4))
        pred_pairs, rel_scores = cat("pred_rel_inds", i64, (-1, 2)), cat("rel_scores", f32, (-1, self.num_rel))
        if self.mode == "predcls":   # vg_eval.py:517-520
            pred_classes, pred_boxes = gt_classes, gt_boxes
            obj_scores = torch.ones(gt_classes.shape[0], dtype=f32, device=dev)
        else:
            pred_classes, pred_boxes = cat("pred_classes", i64, (-1,)), cat("pred_boxes", f32, (-1, 4))
            obj_scores = cat("obj_scores", f32, (-1,))
            if pred_boxes.shape[0] != gt_boxes.shape[0]:
                raise ValueError("sgcls: the number of predicted boxes must equal the number of GT boxes")
        rows = lambda x, width: int(x.numel() // width) if isinstance(x, torch.Tensor) else int(np.asarray(x).size // width)
        n_g = [rows(im["gt_rels"], 3) for im in images]
        n_o = [int(len(im["gt_classes"])) for im in images]
        n_p = [rows(im["pred_rel_inds"], 2) for im in images]
        obj_off, pair_off = cached_offsets(n_o, n_p, dev)
        gt_off = cached_offsets(n_g, n_g, dev)[0]
        n_img, sum_g, sum_p, C = len(images), sum(n_g), sum(n_p), self.num_rel
        out = {k: torch.empty(max(sum_g, 1), dtype=i32, device=dev) for k in ("gc_rank", "ng_rank", "acc_rank", "zeroshot_flag")}
        ng_rows = torch.zeros((n_img, 100), dtype=i32, device=dev)
        ng_cols = torch.zeros((n_img, 100), dtype=i32, device=dev)
        ng_count = torch.zeros(n_img, dtype=i32, device=dev)
        metrics = torch.empty(18 + 6 * (C - 1) + 2, dtype=torch.float64, device=dev)
        need = lib.veto_sgg_eval_workspace_bytes(n_img, sum_p, sum_g, C)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        a = native.VetoSggEvalArgs()
        a.struct_size = ctypes.sizeof(native.VetoSggEvalArgs)
        a.n_img, a.n_rel_cls, a.n_zeroshot, a.iou_thres = n_img, C, int(self.zeroshot.shape[0]), self.iou_thres
        a.gt_offset, a.obj_offset, a.pair_offset = gt_off.data_ptr(), obj_off.data_ptr(), pair_off.data_ptr()
        a.gt_rels, a.gt_classes, a.gt_boxes = gt_rels.data_ptr(), gt_classes.data_ptr(), gt_boxes.data_ptr()
        a.pred_pairs, a.rel_scores = pred_pairs.data_ptr(), rel_scores.data_ptr()
        a.pred_classes, a.pred_boxes, a.obj_scores = pred_classes.data_ptr(), pred_boxes.data_ptr(), obj_scores.data_ptr()
        a.zeroshot = self.zeroshot.data_ptr() if self.zeroshot.shape[0] else None
        a.gc_rank, a.ng_rank, a.acc_rank = out["gc_rank"].data_ptr(), out["ng_rank"].data_ptr(), out["acc_rank"].data_ptr()
        a.zeroshot_flag = out["zeroshot_flag"].data_ptr()
        a.ng_rows, a.ng_cols, a.ng_count, a.metrics = ng_rows.data_ptr(), ng_cols.data_ptr(), ng_count.data_ptr(), metrics.data_ptr()
        stream = torch.cuda.current_stream(dev)
        native.check(lib.veto_sgg_eval(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(a), sum_p, sum_g,
                                       ctypes.c_void_p(self._workspace.data_ptr()), self._workspace.numel()))
        m = metrics.cpu().numpy()
        Cf = C - 1
        res = {"images_evaluated": int(m[18 + 6 * Cf]), "images_with_zeroshot": int(m[18 + 6 * Cf + 1])}
        for j, name in enumerate(("recall", "recall_nogc", "zeroshot_recall", "accuracy", "mean_recall", "ng_mean_recall")):
            res[name] = {k: float(m[3 * j + i]) for i, k in enumerate(KS)}
        for kind, name in enumerate(("mean_recall_list", "ng_mean_recall_list")):
            res[name] = {k: m[18 + (kind * 3 + i) * Cf: 18 + (kind * 3 + i + 1) * Cf].tolist() for i, k in enumerate(KS)}
        # per-image views (what the reference keeps as lists in its result_dict), from the match ranks
        host = {k: v.cpu().numpy() for k, v in out.items()}
        ngc, ngr, ngk = ng_count.cpu().numpy(), ng_rows.cpu().numpy(), ng_cols.cpu().numpy()
        res["per_image"] = []
        lists = {n: {k: [] for k in KS} for n in ("recall_list", "recall_nogc_list", "zeroshot_recall_list")}
        g0 = 0
        for i in range(n_img):
            G = n_g[i]
            if G == 0 or n_p[i] == 0:
                res["per_image"].append(None)
                g0 += G
                continue
            sl = slice(g0, g0 + G)
            zs = host["zeroshot_flag"][sl].astype(bool)
            res["per_image"].append({"gc_rank": host["gc_rank"][sl].astype(np.int64), "ng_rank": host["ng_rank"][sl].astype(np.int64),
                                     "acc_rank": host["acc_rank"][sl].astype(np.int64), "zeroshot": zs,
                                     "ng_rows": ngr[i, :ngc[i]], "ng_cols": ngk[i, :ngc[i]]})
            for k in KS:
                hit = host["gc_rank"][sl] < k
                lists["recall_list"][k].append(float(hit.sum()) / float(G))
                lists["recall_nogc_list"][k].append(float((host["ng_rank"][sl] < k).sum()) / float(G))
                if zs.any():
                    lists["zeroshot_recall_list"][k].append(float((hit & zs).sum()) / float(zs.sum()))
            g0 += G
        res.update(lists)
        return res

    def generate_print_string(self, res):
        """The lines the reference logs (sgg_eval.py generate_print_string of the six evaluators)."""
        fmt = lambda tag, d: "".join(" %s @ %d: %.4f; " % (tag, k, d[k]) for k in KS)
        m = self.mode
        return ("SGG eval: " + fmt(" R", res["recall"]) + " for mode=%s, type=Recall(Main).\n" % m +
                "SGG eval: " + fmt("ngR", res["recall_nogc"]) + " for mode=%s, type=No Graph Constraint Recall(Main).\n" % m +
                "SGG eval: " + fmt(" zR", res["zeroshot_recall"]) + " for mode=%s, type=Zero Shot Recall.\n" % m +
                "SGG eval: " + fmt(" mR", res["mean_recall"]) + " for mode=%s, type=Mean Recall.\n" % m +
                "SGG eval: " + fmt("ng-mR", res["ng_mean_recall"]) + " for mode=%s, type=No Graph Constraint Mean Recall.\n" % m +
                "SGG eval: " + fmt("  A", res["accuracy"]) + " for mode=%s, type=TopK Accuracy.\n" % m)
