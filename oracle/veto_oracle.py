"""CPU oracle for the VETO pairwise relation-prediction path.  TEST INFRASTRUCTURE ONLY.

A from-scratch restatement, in plain torch-CPU tensor algebra, of what the
reference computes on this path, written in the *reference formulation*
(materialised cat(f[s], f[o]) gather, per-pair patch embedding, every layer on
all 19 tokens), so that timing it is timing the reference's own cost.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  The product (veto_amd/) never does.

Parity status: PINNED.  The reference holds no tests/golden vectors for this
path, so the oracle is pinned against outputs of the
reference itself: tests/golden/*.npz were produced in the build container by
tests/golden/make_golden.py, which imports the real VETOPredictor /
VETOPredictor_MEET from /root/reference; tests/test_oracle_golden.py checks this
file against every one of those fixtures (<= 2e-5 max-abs on logits).

All file:line citations are relative to /root/reference/pysgg/modeling/roi_heads/relation_head/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(x, dtype):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(dtype) if x.is_floating_point() else x


class OracleConfig:
    def __init__(self, layers=6, heads=6, dim=576, patch=2, mode="predcls", meet_groups=None,
                 prefix="", experts=0):
        self.layers, self.heads, self.dim, self.patch = layers, heads, dim, patch
        self.mode = mode
        self.meet_groups = meet_groups  # list of group sizes, or None for the vanilla head
        self.experts = experts          # 3 with ENSEMBLE_LEARNING.EXPERT_GROUP (:3717-3723), else 0
        self.prefix = prefix            # "model." for VETOPredictor_MEET state dicts


def enumerate_test_pairs(n):
    """sampling.py:31-52 (GT boxes): nonzero(ones - eye) in row-major order, or the
    [[0, 0]] placeholder when there is no candidate pair (n == 1)."""
    if n <= 1:
        return np.zeros((1, 2), dtype=np.int64)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    keep = i != j
    return np.stack([i[keep], j[keep]], axis=1).astype(np.int64)


def build_pair_indices(rel_pair_idxs, num_objs):
    """roi_relation_predictors.py:4104-4115: per-image object offset added to both columns."""
    subj, obj, off = [], [], 0
    for pairs, n in zip(rel_pair_idxs, num_objs):
        pairs = np.asarray(pairs)
        subj.append(pairs[:, 0] + off)
        obj.append(pairs[:, 1] + off)
        off += n
    return np.concatenate(subj).astype(np.int64), np.concatenate(obj).astype(np.int64)


def center_xywh_from_xyxy(boxes):
    """structures/bounding_box.py:60-78 (xyxy -> xywh with the +1 convention) followed by
    model_mpv2.py:341-345 (centre = corner + 0.5 * size)."""
    x1, y1, x2, y2 = boxes.unbind(-1)
    w = x2 - x1 + 1
    h = y2 - y1 + 1
    return torch.stack([x1 + 0.5 * w, y1 + 0.5 * h, w, h], dim=-1)


def object_embeddings(sd, cfg, labels, predict_logits, pred_labels, dtype):
    """roi_relation_predictors.py:4086-4095 (vanilla) / :3769-3784 (MEET Ensemble)."""
    E = _t(sd[cfg.prefix + "obj_embed.weight"], dtype)
    num_cls = E.shape[0]
    if cfg.mode == "predcls":
        lab = torch.as_tensor(labels).long()
        return E[lab], F.one_hot(lab, num_cls).to(dtype)
    lab = torch.as_tensor(pred_labels).long()
    obj_dists = F.one_hot(lab, num_cls).to(dtype)
    if cfg.meet_groups is None:
        emb = torch.softmax(_t(predict_logits, dtype), dim=1) @ E          # :4095
    else:
        preds = obj_dists[:, 1:].max(1)[1] + 1                            # :3783
        emb = E[preds]                                                    # :3784
    return emb, obj_dists


def position_embedding(sd, cfg, boxes_xyxy, dtype):
    """pos_embed = BatchNorm1d(4) (eval: running stats, eps 1e-5) -> Linear(4,128) -> ReLU;
    roi_relation_predictors.py:4042-4047,4097-4102."""
    p = cfg.prefix + "pos_embed."
    x = center_xywh_from_xyxy(_t(boxes_xyxy, dtype))
    mean, var = _t(sd[p + "0.running_mean"], dtype), _t(sd[p + "0.running_var"], dtype)
    x = (x - mean) / torch.sqrt(var + 1e-5) * _t(sd[p + "0.weight"], dtype) + _t(sd[p + "0.bias"], dtype)
    x = x @ _t(sd[p + "1.weight"], dtype).t() + _t(sd[p + "1.bias"], dtype)
    return torch.relu(x)


def patchify(x, p):
    """einops 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' of model_veto.py:109-110, spelled out."""
    b, c, H, W = x.shape
    h, w = H // p, W // p
    x = x.reshape(b, c, h, p, w, p)            # b c h p1 w p2
    x = x.permute(0, 2, 4, 3, 5, 1)            # b h w p1 p2 c
    return x.reshape(b, h * w, p * p * c)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def build_tokens(sd, cfg, rel_depth, rel_visual, rel_location, rel_class, dtype):
    """PatchEmbed.forward + Transformer.forward, model_veto.py:52-64,108-115.
    NB the crossed naming: the FIRST argument (depth) goes through proj_d (512 wide),
    the second (rgb) through proj_v (64 wide); roi_relation_predictors.py:4124."""
    t = cfg.prefix + "fusion_transformer.transformer."
    d = patchify(rel_depth, cfg.patch)
    v = patchify(rel_visual, cfg.patch)
    d = d @ _t(sd[t + "patch_embed.proj_d.weight"], dtype).t() + _t(sd[t + "patch_embed.proj_d.bias"], dtype)
    v = v @ _t(sd[t + "patch_embed.proj_v.weight"], dtype).t() + _t(sd[t + "patch_embed.proj_v.bias"], dtype)
    x = torch.cat([d, v], dim=2)
    cls = _t(sd[t + "cls_token"], dtype).expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x, rel_location.unsqueeze(1), rel_class.unsqueeze(1)], dim=1)
    return x + _t(sd[t + "pos_embedding"], dtype)


def encoder_layer(sd, cfg, x, l, dtype):
    """One (PreNorm Attention + residual, PreNorm FeedForward + residual) block;
    model_veto.py:18-21,85-96,125-146."""
    t = cfg.prefix + "fusion_transformer.transformer.layers.%d." % l
    H = cfg.heads
    b, n, D = x.shape
    dh = D // H
    y = layer_norm(x, _t(sd[t + "0.norm.weight"], dtype), _t(sd[t + "0.norm.bias"], dtype))
    qkv = y @ _t(sd[t + "0.fn.to_qkv.weight"], dtype).t()
    q, k, v = [z.reshape(b, n, H, dh).permute(0, 2, 1, 3) for z in qkv.chunk(3, dim=-1)]
    dots = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
    attn = torch.softmax(dots, dim=-1)
    out = (attn @ v).permute(0, 2, 1, 3).reshape(b, n, D)
    out = out @ _t(sd[t + "0.fn.to_out.0.weight"], dtype).t() + _t(sd[t + "0.fn.to_out.0.bias"], dtype)
    x = out + x
    y = layer_norm(x, _t(sd[t + "1.norm.weight"], dtype), _t(sd[t + "1.norm.bias"], dtype))
    h = gelu_erf(y @ _t(sd[t + "1.fn.net.0.weight"], dtype).t() + _t(sd[t + "1.fn.net.0.bias"], dtype))
    y = h @ _t(sd[t + "1.fn.net.3.weight"], dtype).t() + _t(sd[t + "1.fn.net.3.bias"], dtype)
    return y + x


def forward(sd, cfg, batch, rel_pair_idxs=None, dtype=torch.float32, return_intermediates=False,
            pair_chunk=4096):
    """Eval forward of VETOPredictor (roi_relation_predictors.py:4074-4139) or, with
    cfg.meet_groups, of VETOPredictor_MEET/Ensemble (:3752-3853,:3909-3995).

    batch: dict from veto_amd.synth.synthetic_batch (numpy arrays).
    Returns (logits [P, n_out], subj_inds, obj_inds[, intermediates]).
    For MEET, logits is the column-concatenation of the K group heads.
    `pair_chunk` only bounds memory (pairs are independent); it does not change results.
    """
    num_objs = batch["num_objs"]
    if rel_pair_idxs is None:
        rel_pair_idxs = [enumerate_test_pairs(n) for n in num_objs]
    subj, obj = build_pair_indices(rel_pair_idxs, num_objs)
    subj_t, obj_t = torch.from_numpy(subj), torch.from_numpy(obj)
    pre = cfg.prefix

    emb, obj_dists = object_embeddings(sd, cfg, batch["labels"], batch.get("predict_logits"),
                                       batch.get("pred_labels"), dtype)
    pos = position_embedding(sd, cfg, batch["boxes"], dtype)
    rgb = _t(batch["roi_features"], dtype)
    dep = _t(batch["roi_depth_features"], dtype)

    if cfg.meet_groups is None:
        Wh = _t(sd[pre + "rel_out.weight"], dtype)
        bh = _t(sd[pre + "rel_out.bias"], dtype)
    elif cfg.experts:   # :3833-3837 rel_out_group[j][k]; columns expert-major, then group
        names = [pre + "rel_out_group.%d.%d" % (e, k) for e in range(cfg.experts) for k in range(len(cfg.meet_groups))]
        Wh = torch.cat([_t(sd[n + ".weight"], dtype) for n in names])
        bh = torch.cat([_t(sd[n + ".bias"], dtype) for n in names])
    else:
        Wh = torch.cat([_t(sd[pre + "rel_out.%d.weight" % k], dtype) for k in range(len(cfg.meet_groups))])
        bh = torch.cat([_t(sd[pre + "rel_out.%d.bias" % k], dtype) for k in range(len(cfg.meet_groups))])

    logits, inter = [], {"tokens": [], "cls": []}
    for c0 in range(0, len(subj), pair_chunk):
        s, o = subj_t[c0:c0 + pair_chunk], obj_t[c0:c0 + pair_chunk]
        # :4118-4123 -- the materialised pair gathers
        rel_location = torch.cat([pos[s], pos[o]], dim=1)
        rel_location = torch.relu(rel_location @ _t(sd[pre + "location_projection.0.weight"], dtype).t()
                                  + _t(sd[pre + "location_projection.0.bias"], dtype))
        rel_class = torch.cat([emb[s], emb[o]], dim=1)
        rel_class = torch.relu(rel_class @ _t(sd[pre + "class_projection.0.weight"], dtype).t()
                               + _t(sd[pre + "class_projection.0.bias"], dtype))
        rel_visual = torch.cat([rgb[s], rgb[o]], dim=1)
        rel_depth = torch.cat([dep[s], dep[o]], dim=1)
        x = build_tokens(sd, cfg, rel_depth, rel_visual, rel_location, rel_class, dtype)
        if return_intermediates:
            inter["tokens"].append(x)
        for l in range(cfg.layers):
            x = encoder_layer(sd, cfg, x, l, dtype)
        cls = x[:, 0]                                   # model_veto.py:23 (no final LayerNorm)
        if return_intermediates:
            inter["cls"].append(cls)
        logits.append(cls @ Wh.t() + bh)                # :4125 / :3842-3843
    logits = torch.cat(logits)
    if return_intermediates:
        inter = {k: torch.cat(v) for k, v in inter.items()}
        inter["obj_dists"] = obj_dists
        inter["pos_embed"] = pos
        inter["obj_embed"] = emb
        return logits, subj, obj, inter
    return logits, subj, obj


def postprocess(rel_logits, obj_logits, rel_pair_idxs, num_objs, dtype=torch.float32):
    """Vanilla GT-box branch of PostProcessor.forward, inference.py:398-453.
    rel_logits [sum P, C], obj_logits [sum N, Co], rel_pair_idxs list of [P_i, 2] (image-local).
    Returns one dict per image: pred_scores, pred_labels, rel_pair_idxs, pred_rel_scores,
    pred_rel_labels, triple_scores (sorted, descending; exact ties keep the lower index first)."""
    rel_logits, obj_logits = _t(rel_logits, dtype), _t(obj_logits, dtype)
    out, o0, p0 = [], 0, 0
    for pairs, n in zip(rel_pair_idxs, num_objs):
        pairs = torch.as_tensor(np.asarray(pairs)).long()
        P = pairs.shape[0]
        obj_prob = torch.softmax(obj_logits[o0:o0 + n], -1)           # :405
        obj_prob[:, 0] = 0                                            # :406
        obj_scores, obj_pred = obj_prob[:, 1:].max(dim=1)             # :411
        obj_pred = obj_pred + 1                                       # :412
        rel_prob = torch.softmax(rel_logits[p0:p0 + P], -1)           # :440
        rel_scores, rel_class = rel_prob[:, 1:].max(dim=1)            # :441
        rel_class = rel_class + 1                                     # :442
        triple = rel_scores * obj_scores[pairs[:, 0]] * obj_scores[pairs[:, 1]]   # :444
        order = torch.sort(triple, descending=True, stable=True)[1]  # :445 (stable: defined tie order)
        out.append({"pred_scores": obj_scores, "pred_labels": obj_pred, "rel_pair_idxs": pairs[order],
                    "pred_rel_scores": rel_prob[order], "pred_rel_labels": rel_class[order],
                    "triple_scores": triple[order]})
        o0 += n
        p0 += P
    return out


def postprocess_meet(rel_logits_by_group, obj_logits, rel_pair_idx, incre_idx_list, num_rel_cls, dtype=torch.float32):
    """MEET merge branch of PostProcessor.forward (ENSEMBLE_LEARNING.ENABLED, EXPERT_GROUP False),
    inference.py:284-397, one image (the reference zips the batch-wide group logits with the FIRST
    image only, so it is only meaningful for one image per batch).
    Per group i: softmax over its g_i+2 logits, DROP the last ("other group") column (:350-351),
    rel_class = argmax over columns 1.. (+1) -- a GROUP-LOCAL label, kept as such (:352-353,:388) --,
    triple score as in the vanilla branch; every pair is kept (:358-372); the group's g_i+1
    probabilities are scattered into a zero [*, num_rel_cls] row at columns [0] + its own classes
    (:354-357,:387); the K*P rows are sorted by triple score, descending (:390).  The pair indices come
    out as a FLOAT tensor (torch.zeros(total, 2), :381)."""
    obj_logits = _t(obj_logits, dtype)
    pairs = torch.as_tensor(np.asarray(rel_pair_idx)).long()
    obj_prob = torch.softmax(obj_logits, -1)
    obj_prob[:, 0] = 0
    obj_scores, obj_pred = obj_prob[:, 1:].max(dim=1)
    obj_pred = obj_pred + 1
    K = len(rel_logits_by_group)
    triples, labels, probs, prs = [], [], [], []
    for i in range(K):
        logit = _t(rel_logits_by_group["group_%d" % i], dtype)
        prob = torch.softmax(logit, -1)[:, :-1]
        rel_scores, rel_class = prob[:, 1:].max(dim=1)
        rel_class = rel_class + 1
        cols = [0] + [c for c, x in enumerate(incre_idx_list) if x == i + 1]
        full = torch.zeros(prob.shape[0], num_rel_cls, dtype=dtype)
        full[:, cols] = prob
        triples.append(rel_scores * obj_scores[pairs[:, 0]] * obj_scores[pairs[:, 1]])
        labels.append(rel_class)
        probs.append(full)
        prs.append(pairs)
    triple, label, prob, pr = torch.cat(triples), torch.cat(labels), torch.cat(probs), torch.cat(prs)
    order = torch.sort(triple, descending=True, stable=True)[1]
    return {"pred_scores": obj_scores, "pred_labels": obj_pred, "rel_pair_idxs": pr[order].to(torch.float32),
            "pred_rel_scores": prob[order], "pred_rel_labels": label[order], "triple_scores": triple[order]}


def postprocess_vote(rel_logits, obj_logits, rel_pair_idx, incre_idx_list, num_rel_cls, voting="C",
                     dtype=torch.float32):
    """EXPERT_GROUP voting branch of PostProcessor.forward, inference.py:93-283, one image.
    rel_logits: dict 'group_<k><e>' (k = group 0.., e = expert 1..3) of [P, g_k+2] logits.
    Per group, per expert e: prob_e = softmax(logit_e)[:, :-1]; (score_e, cls_e) = max over columns 1..
    (+1); t_e = score_e * obj_s * obj_o (:178-190; `chosen_idx_bool` is always true once the last column
    is dropped).  agree_ab = cls_a == cls_b for the expert pairs (0,1), (1,2), (0,2) (:192-199).
      'C' (two agree, :211-255): pair means t_ab = (t_a + t_b)/2 and p_ab = (p_a + p_b)/2 -- EXCEPT that
          the (1,2) probability mean is built from expert 1 twice (:224-226), i.e. p_12 = p_1 --; a row is
          kept if any pair agrees; its score / probabilities are the sum over the agreeing pairs divided
          by their count, its label the agreed class.
      'U' (all agree, :257-261): kept if all three pairs agree; score / probabilities are the three-way
          means (sum / 3), label = the experts' common class.
    Kept rows of all groups are concatenated in (group, pair) order, each group's g+1 probabilities
    scattered to columns [0] + its own classes of a zero [*, num_rel_cls] row, and sorted by score
    descending (:263-283).  Pair indices come out as a FLOAT tensor, labels are group-local."""
    assert voting in ("C", "U")
    obj_logits = _t(obj_logits, dtype)
    pairs = torch.as_tensor(np.asarray(rel_pair_idx)).long()
    obj_prob = torch.softmax(obj_logits, -1)
    obj_prob[:, 0] = 0
    obj_scores, obj_pred = obj_prob[:, 1:].max(dim=1)
    obj_pred = obj_pred + 1
    s0, s1 = obj_scores[pairs[:, 0]], obj_scores[pairs[:, 1]]
    K = len(rel_logits) // 3
    triples, labels, probs, prs = [], [], [], []
    for k in range(K):
        p, c, t = [], [], []
        for e in range(3):
            prob = torch.softmax(_t(rel_logits["group_%d%d" % (k, e + 1)], dtype), -1)[:, :-1]
            sc, cl = prob[:, 1:].max(dim=1)
            p.append(prob)
            c.append(cl + 1)
            t.append(sc * s0 * s1)
        agree = [c[0] == c[1], c[1] == c[2], c[0] == c[2]]
        if voting == "C":
            t_pair = torch.stack([(t[0] + t[1]) / 2, (t[1] + t[2]) / 2, (t[0] + t[2]) / 2], 1)
            p_pair = torch.stack([(p[0] + p[1]) / 2, (p[1] + p[1]) / 2, (p[0] + p[2]) / 2], 1)
            mask = torch.stack(agree, 1)
            cnt = mask.sum(1)
            keep = cnt > 0
            triple = (t_pair * mask).sum(1) / cnt
            prob = (p_pair * mask.unsqueeze(2)).sum(1) / cnt.unsqueeze(1)
            label = torch.zeros_like(c[0])
            for cl, ag in zip(c, agree):
                label[ag] = cl[ag]
        else:
            keep = agree[0] & agree[1] & agree[2]
            triple = (t[0] + t[1] + t[2]) / 3
            prob = (p[0] + p[1] + p[2]) / 3
            label = c[2]
        cols = [0] + [cc for cc, x in enumerate(incre_idx_list) if x == k + 1]
        full = torch.zeros(int(keep.sum()), num_rel_cls, dtype=dtype)
        full[:, cols] = prob[keep]
        triples.append(triple[keep])
        labels.append(label[keep])
        probs.append(full)
        prs.append(pairs[keep])
    triple, label, prob, pr = torch.cat(triples), torch.cat(labels), torch.cat(probs), torch.cat(prs)
    order = torch.sort(triple, descending=True, stable=True)[1]
    return {"pred_scores": obj_scores, "pred_labels": obj_pred, "rel_pair_idxs": pr[order].to(torch.float32),
            "pred_rel_scores": prob[order], "pred_rel_labels": label[order], "triple_scores": triple[order]}


def meet_incre_idx_list(group_sizes):
    """SHA_GCL_extra/extra_function_utils.py:39-77 (first return value) for the contiguous
    splits of group_chosen_function.py:6-94: class c (1-based, frequency order) belongs to
    group g+1 where g is the first group whose cumulative size reaches c; background -> 0."""
    out = [0]
    for g, size in enumerate(group_sizes):
        out += [g + 1] * size
    return out


def class_balanced_weights(counts, beta=0.999):
    """BETA_LOSS weights, roi_relation_predictors.py:4058-4066: counts sorted descending in
    place, w = (1-beta)/(1-beta**n), renormalised to sum to the number of classes."""
    c = np.sort(np.asarray(counts, dtype=np.float64))[::-1]
    w = (1.0 - beta) / (1.0 - beta ** c)
    return (w * (len(c) / w.sum())).astype(np.float32)
