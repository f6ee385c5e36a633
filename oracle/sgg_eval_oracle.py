"""CPU restatement of the reference's relation evaluators.

TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else; the product path is
veto_amd/csrc/sgg_eval.hip behind veto_sgg_eval.

PARITY PINNED: tests/golden/sggeval_*.npz hold the result_dict the reference's own evaluator classes
(pysgg/data/datasets/evaluation/vg/sgg_eval.py, driven by vg_eval.py:459-566
`evaluate_relation_of_one_image`) produce for veto_amd.synth.synthetic_eval_images; see
tests/golden/make_golden.py::run_sgg_eval and tests/test_sgg_eval.py.

Restated (GT-box modes predcls / sgcls; numpy, the reference's arithmetic):
  SGRecall                   sgg_eval.py:121-187   R@K    (graph constraint: one predicate per pair)
  SGNoGraphConstraintRecall  :195-255              ngR@K  (top-100 (pair, predicate) scores)
  SGZeroShotRecall           :263-313              zR@K
  SGPairAccuracy             :322-369              A@K    (only predictions on GT pairs)
  SGMeanRecall               :377-466              mR@K
  SGNGMeanRecall             :470-546              ng-mR@K
built on _triplet (:44-75), _compute_pred_matches (:78-118), intersect_2d / argsort_desc
(utils/miscellaneous.py:47-70) and boxlist_iou with the +1 pixel convention (structures/boxlist_ops.py:54-90).
"""
import numpy as np

KS = (20, 50, 100)
NO_MATCH = 0x3fffffff


def box_iou(a, b):
    """boxlist_iou (boxlist_ops.py:54-90) for one box against many, float32, +1 convention."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    one = np.float32(1)
    area_a = (a[2] - a[0] + one) * (a[3] - a[1] + one)
    area_b = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    lt = np.maximum(a[None, :2], b[:, :2])
    rb = np.minimum(a[None, 2:], b[:, 2:])
    wh = np.clip(rb - lt + one, 0, None)
    inter = wh[:, 0] * wh[:, 1]
    return inter / (area_a + area_b - inter)


def first_match_ranks(gt_triplets, gt_boxes8, pred_triplets, pred_boxes8, iou_thres):
    """_compute_pred_matches (:78-118) turned around: for every GT relation the index of the FIRST prediction
    that matches it (same (subject class, predicate, object class) and both box IoUs >= iou_thres), or
    NO_MATCH.  `reduce(np.union1d, pred_to_gt[:k])` (:171) contains GT g iff that index is < k."""
    ranks = np.full(len(gt_triplets), NO_MATCH, dtype=np.int64)
    for g in range(len(gt_triplets)):
        same = np.nonzero((pred_triplets == gt_triplets[g][None]).all(1))[0]
        if len(same) == 0:
            continue
        ok = (box_iou(gt_boxes8[g, :4], pred_boxes8[same, :4]) >= iou_thres) & \
             (box_iou(gt_boxes8[g, 4:], pred_boxes8[same, 4:]) >= iou_thres)
        if ok.any():
            ranks[g] = same[ok][0]
    return ranks


def evaluate_image(img, mode, zeroshot, iou_thres=0.5):
    """vg_eval.py:459-566 for one image.  Returns None when the image has no GT relation (:474-475), else a
    dict with, per GT relation: gc_rank, ng_rank (first matching prediction in the two ranked lists),
    acc_rank (first match counted among the predictions that sit on a GT pair) and the zero-shot flag."""
    gt_rels = np.asarray(img["gt_rels"], dtype=np.int64)
    if len(gt_rels) == 0:
        return None
    gt_classes, gt_boxes = np.asarray(img["gt_classes"]), np.asarray(img["gt_boxes"], dtype=np.float32)
    pred_rel_inds, rel_scores = np.asarray(img["pred_rel_inds"], dtype=np.int64), np.asarray(img["rel_scores"])
    if mode == "predcls":     # :517-520
        pred_boxes, pred_classes, obj_scores = gt_boxes, gt_classes, np.ones(len(gt_classes))
    else:
        pred_boxes, pred_classes = np.asarray(img["pred_boxes"], dtype=np.float32), np.asarray(img["pred_classes"])
        obj_scores = np.asarray(img["obj_scores"])
    # SGPairAccuracy.prepare_gtpair (:338-346) and SGZeroShotRecall.prepare_zeroshot (:279-291)
    pred_pair_in_gt = ((pred_rel_inds[:, 0] * 1024 + pred_rel_inds[:, 1])[:, None] ==
                       (gt_rels[:, 0] * 1024 + gt_rels[:, 1])[None]).any(1)
    gt_soc = np.column_stack([gt_classes[gt_rels[:, 0]], gt_classes[gt_rels[:, 1]], gt_rels[:, 2]])
    zs = (gt_soc[:, None, :] == np.asarray(zeroshot)[None]).all(2).any(1)
    if len(pred_rel_inds) == 0:   # :544-545
        return None
    gt_triplets = np.column_stack([gt_classes[gt_rels[:, 0]], gt_rels[:, 2], gt_classes[gt_rels[:, 1]]])   # :60
    gt_boxes8 = np.column_stack([gt_boxes[gt_rels[:, 0]], gt_boxes[gt_rels[:, 1]]])
    # graph constraint (:146-166): one label per pair, the list order is the prediction order
    labels = 1 + rel_scores[:, 1:].argmax(1)
    pred_triplets = np.column_stack([pred_classes[pred_rel_inds[:, 0]], labels, pred_classes[pred_rel_inds[:, 1]]])
    pred_boxes8 = np.column_stack([pred_boxes[pred_rel_inds[:, 0]], pred_boxes[pred_rel_inds[:, 1]]])
    gc_rank = first_match_ranks(gt_triplets, gt_boxes8, pred_triplets, pred_boxes8, iou_thres)
    # pair accuracy (:357-366): the same matches, ranked within the predictions that sit on a GT pair
    flagged_before = np.concatenate([[0], np.cumsum(pred_pair_in_gt)[:-1]])
    acc_rank = np.full(len(gt_rels), NO_MATCH, dtype=np.int64)
    keep = np.nonzero(pred_pair_in_gt)[0]
    if len(keep):
        r = first_match_ranks(gt_triplets, gt_boxes8, pred_triplets[keep], pred_boxes8[keep], iou_thres)
        acc_rank = np.where(r < NO_MATCH, flagged_before[keep[np.minimum(r, len(keep) - 1)]], NO_MATCH)
    # no graph constraint (:221-229): top 100 of obj_s * obj_o * rel_scores[:, 1:] over (pair, predicate)
    per_rel = obj_scores[pred_rel_inds].prod(1)
    overall = per_rel[:, None] * rel_scores[:, 1:]
    flat = np.argsort(-overall.ravel(), kind="stable")[:100]      # the reference's quicksort leaves exact ties open
    rows, cols = np.unravel_index(flat, overall.shape)
    ng_triplets = np.column_stack([pred_classes[pred_rel_inds[rows, 0]], cols + 1, pred_classes[pred_rel_inds[rows, 1]]])
    ng_boxes8 = np.column_stack([pred_boxes[pred_rel_inds[rows, 0]], pred_boxes[pred_rel_inds[rows, 1]]])
    ng_rank = first_match_ranks(gt_triplets, gt_boxes8, ng_triplets, ng_boxes8, iou_thres)
    return {"gc_rank": gc_rank, "ng_rank": ng_rank, "acc_rank": acc_rank, "zeroshot": zs, "gt_pred": gt_rels[:, 2],
            "ng_rows": rows, "ng_cols": cols + 1}


def evaluate(images, mode, zeroshot, num_rel, iou_thres=0.5):
    """The reference's accumulation: per-image lists averaged with np.mean (:133-136, :209), mean recall from the
    per-image per-class hit ratios (:420-466), A@K = mean(hits) / mean(counts) (:331-336)."""
    res = {"recall": {k: [] for k in KS}, "recall_nogc": {k: [] for k in KS}, "zeroshot_recall": {k: [] for k in KS},
           "accuracy_hit": {k: [] for k in KS}, "accuracy_count": {k: [] for k in KS}}
    collect = {k: [[] for _ in range(num_rel)] for k in KS}
    ng_collect = {k: [[] for _ in range(num_rel)] for k in KS}
    per_image = []
    for img in images:
        r = evaluate_image(img, mode, zeroshot, iou_thres)
        per_image.append(r)
        if r is None:
            continue
        G = len(r["gc_rank"])
        for k in KS:
            hit = r["gc_rank"] < k
            res["recall"][k].append(float(hit.sum()) / float(G))
            res["recall_nogc"][k].append(float((r["ng_rank"] < k).sum()) / float(G))
            if r["zeroshot"].any():
                res["zeroshot_recall"][k].append(float((hit & r["zeroshot"]).sum()) / float(r["zeroshot"].sum()))
            res["accuracy_hit"][k].append(float((r["acc_rank"] < k).sum()))
            res["accuracy_count"][k].append(float(G))
            for coll, h in ((collect, hit), (ng_collect, r["ng_rank"] < k)):
                cnt = np.bincount(r["gt_pred"], minlength=num_rel)
                hc = np.bincount(r["gt_pred"][h], minlength=num_rel)
                cnt[0], hc[0] = G, h.sum()
                for n in range(num_rel):
                    if cnt[n] > 0:
                        coll[k][n].append(float(hc[n] / cnt[n]))
    out = {"per_image": per_image}
    for name in ("recall", "recall_nogc", "zeroshot_recall"):
        out[name] = {k: (float(np.mean(v)) if len(v) else float("nan")) for k, v in res[name].items()}
        out[name + "_list"] = res[name]
    out["accuracy"] = {k: float(np.mean(res["accuracy_hit"][k]) / np.mean(res["accuracy_count"][k])) for k in KS}
    for name, coll in (("mean_recall", collect), ("ng_mean_recall", ng_collect)):
        out[name], out[name + "_list"] = {}, {}
        for k in KS:
            per_cls = [float(np.mean(coll[k][n + 1])) if len(coll[k][n + 1]) else 0.0 for n in range(num_rel - 1)]
            out[name + "_list"][k] = per_cls
            out[name][k] = sum(per_cls) / float(num_rel - 1)
    return out
