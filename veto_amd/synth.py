"""Portable synthetic inputs and weights for the VETO relation-prediction path.

Everything here is produced by a counter-based integer hash (splitmix64) with
exact IEEE arithmetic on top, so the same tensors are regenerated bit-for-bit on
any machine (this container, the GPU box) from a (seed, name) pair alone.  The
golden fixtures in tests/golden/ only hold the *outputs* the reference produced
for these tensors; the 70 MB of weights never need to be committed.

Shapes and ranges follow SURVEY.md section 8(d) ("Synthetic inputs").
"""
import hashlib
import math

import numpy as np

_MASK = (1 << 64) - 1


def _name_key(seed, name):
    h = hashlib.sha256(("%d/%s" % (seed, name)).encode()).digest()
    return int.from_bytes(h[:8], "little")


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, name, n, stream=0):
    """n doubles in [0,1): (hash >> 11) * 2**-53, exact."""
    key = np.uint64((_name_key(seed, name) + 0x632BE59BD9B4E019 * stream) & _MASK)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key
    bits = _splitmix64(ctr)
    return (bits >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def uniform(seed, name, shape, lo, hi):
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, name, shape, mean=0.0, std=1.0):
    """Approximate N(mean, std): centred sum of four uniforms (Irwin-Hall), which
    needs no libm call and is therefore bit-portable."""
    n = int(np.prod(shape)) if len(shape) else 1
    s = np.zeros(n, dtype=np.float64)
    for k in range(4):
        s += uniform01(seed, name, n, stream=k + 1)
    z = (s - 2.0) * 1.7320508075688772  # var(sum of 4 U) = 1/3
    return (mean + std * z).astype(np.float32).reshape(shape)


def integers(seed, name, shape, lo, hi):
    """Integers in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64).reshape(shape)


# ---------------------------------------------------------------------------
# Weights: one entry per state-dict key of the reference VETOPredictor
# (roi_relation_predictors.py:3999-4071; key list in SURVEY.md section 8b).
# ---------------------------------------------------------------------------

def _linear(sd, seed, prefix, out_f, in_f, bias=True):
    b = 1.0 / math.sqrt(in_f)
    sd[prefix + ".weight"] = uniform(seed, prefix + ".weight", (out_f, in_f), -b, b)
    if bias:
        sd[prefix + ".bias"] = uniform(seed, prefix + ".bias", (out_f,), -b, b)


def transformer_state_dict(seed, prefix, dim=576, layers=6, in_channels=256, patch=2):
    """Keys of VETOTransformer (model_veto.py:6-146) under `prefix`."""
    sd = {}
    t = prefix + "transformer."
    sd[t + "cls_token"] = normal(seed, t + "cls_token", (1, 1, dim))
    sd[t + "pos_embedding"] = normal(seed, t + "pos_embedding", (1, 1, dim))
    pdim = in_channels * 2 * patch * patch
    _linear(sd, seed, t + "patch_embed.proj_d", 512, pdim)
    _linear(sd, seed, t + "patch_embed.proj_v", 64, pdim)
    for l in range(layers):
        a = t + "layers.%d.0." % l
        f = t + "layers.%d.1." % l
        sd[a + "norm.weight"] = normal(seed, a + "norm.weight", (dim,), 1.0, 0.1)
        sd[a + "norm.bias"] = normal(seed, a + "norm.bias", (dim,), 0.0, 0.1)
        _linear(sd, seed, a + "fn.to_qkv", 3 * dim, dim, bias=False)
        _linear(sd, seed, a + "fn.to_out.0", dim, dim)
        sd[f + "norm.weight"] = normal(seed, f + "norm.weight", (dim,), 1.0, 0.1)
        sd[f + "norm.bias"] = normal(seed, f + "norm.bias", (dim,), 0.0, 0.1)
        _linear(sd, seed, f + "fn.net.0", 2 * dim, dim)
        _linear(sd, seed, f + "fn.net.3", dim, 2 * dim)
    return sd


def trunk_state_dict(seed, prefix="", dim=576, layers=6, num_obj_cls=151, embed_dim=200):
    """Everything VETOPredictor and the MEET Ensemble share (all keys but rel_out)."""
    sd = {}
    sd[prefix + "obj_embed2.weight"] = normal(seed, prefix + "obj_embed2.weight", (num_obj_cls, embed_dim))
    sd[prefix + "obj_embed.weight"] = normal(seed, prefix + "obj_embed.weight", (num_obj_cls, embed_dim), 0.0, 0.5)
    _linear(sd, seed, prefix + "class_projection.0", dim, 2 * embed_dim)
    _linear(sd, seed, prefix + "bbox_embed.0", 32, 9)
    _linear(sd, seed, prefix + "bbox_embed.3", 128, 32)
    p = prefix + "pos_embed.0."
    sd[p + "weight"] = normal(seed, p + "weight", (4,), 1.0, 0.1)
    sd[p + "bias"] = normal(seed, p + "bias", (4,), 0.0, 0.1)
    # running statistics in pixel units (boxes are un-normalised)
    sd[p + "running_mean"] = (np.array([250.0, 200.0, 110.0, 110.0], dtype=np.float32)
                              + normal(seed, p + "running_mean", (4,), 0.0, 5.0))
    sd[p + "running_var"] = (np.array([20000.0, 15000.0, 3500.0, 3500.0], dtype=np.float32)
                             * uniform(seed, p + "running_var", (4,), 0.8, 1.2))
    sd[p + "num_batches_tracked"] = np.array(1000, dtype=np.int64)
    _linear(sd, seed, prefix + "pos_embed.1", 128, 4)
    _linear(sd, seed, prefix + "location_projection.0", dim, 256)
    sd.update(transformer_state_dict(seed, prefix + "fusion_transformer.", dim=dim, layers=layers))
    return sd


def predictor_state_dict(seed, dim=576, layers=6, num_obj_cls=151, num_rel_cls=51):
    """Full state dict of the vanilla VETOPredictor."""
    sd = trunk_state_dict(seed, "", dim, layers, num_obj_cls)
    std = math.sqrt(2.0 / (dim + num_rel_cls))  # xavier_normal_, utils/miscellaneous.py
    sd["rel_out.weight"] = normal(seed, "rel_out.weight", (num_rel_cls, dim), 0.0, std)
    sd["rel_out.bias"] = uniform(seed, "rel_out.bias", (num_rel_cls,), -0.04, 0.04)
    sd["criterion_loss_rel.weight"] = np.ones((num_rel_cls,), dtype=np.float32)
    return sd


def meet_state_dict(seed, group_sizes, dim=576, layers=6, num_obj_cls=151, experts=0):
    """State dict of VETOPredictor_MEET (everything lives under `model.`).  experts=3 gives the
    ENSEMBLE_LEARNING.EXPERT_GROUP layout: `model.rel_out_group.{e}.{k}` for 3 experts x K groups, with
    `model.rel_out.{k}` the same tensors as the LAST expert's (roi_relation_predictors.py:3717-3723
    leaves `self.rel_out` bound to the last list it appended)."""
    sd = trunk_state_dict(seed, "model.", dim, layers, num_obj_cls)
    del sd["model.obj_embed2.weight"]  # Ensemble has no obj_embed2 (roi_relation_predictors.py:3676)

    def head(name, g):
        std = math.sqrt(2.0 / (dim + g + 2))
        sd[name + ".weight"] = normal(seed, name + ".weight", (g + 2, dim), 0.0, std)
        sd[name + ".bias"] = uniform(seed, name + ".bias", (g + 2,), -0.04, 0.04)

    if experts:
        for e in range(experts):
            for k, g in enumerate(group_sizes):
                head("model.rel_out_group.%d.%d" % (e, k), g)
        for k in range(len(group_sizes)):
            for part in (".weight", ".bias"):
                sd["model.rel_out.%d%s" % (k, part)] = sd["model.rel_out_group.%d.%d%s" % (experts - 1, k, part)]
    else:
        for k, g in enumerate(group_sizes):
            head("model.rel_out.%d" % k, g)
    return sd


# ---------------------------------------------------------------------------
# Inputs: B images x N boxes.
# ---------------------------------------------------------------------------

def synthetic_batch(seed, num_images, num_objs, num_obj_cls=151, channels=256, res=8,
                    relu_like=False):
    """Returns a dict of numpy arrays; `num_objs` may be an int or a per-image list."""
    if isinstance(num_objs, int):
        num_objs = [num_objs] * num_images
    total = int(sum(num_objs))
    xy = uniform(seed, "boxes.xy", (total, 2), 0.0, 400.0)
    wh = uniform(seed, "boxes.wh", (total, 2), 10.0, 210.0)
    boxes = np.concatenate([xy, xy + wh], axis=1).astype(np.float32)  # xyxy
    labels = integers(seed, "labels", (total,), 1, num_obj_cls)
    pred_labels = integers(seed, "pred_labels", (total,), 1, num_obj_cls)
    predict_logits = normal(seed, "predict_logits", (total, num_obj_cls))
    rgb = normal(seed, "roi_features", (total, channels, res, res))
    depth = normal(seed, "roi_depth_features", (total, channels, res, res))
    if relu_like:
        rgb = (0.5 * np.abs(rgb)).astype(np.float32)
        depth = (0.5 * np.abs(depth)).astype(np.float32)
    return {
        "num_objs": list(num_objs), "image_size": (800, 600), "boxes": boxes, "labels": labels,
        "pred_labels": pred_labels, "predict_logits": predict_logits,
        "roi_features": rgb, "roi_depth_features": depth,
    }


# ---------------------------------------------------------------------------
# Evaluation inputs: ground truth + sorted predictions per image.
# ---------------------------------------------------------------------------

def synthetic_eval_images(seed, num_objs, mode="predcls", num_rel_cls=51, num_obj_cls=21):
    """Per image: GT boxes / classes / relation tuples and a PostProcessor-shaped prediction (pairs sorted by
    triple score, [P, num_rel_cls] probabilities, object labels / scores).  Division-only arithmetic (no
    libm), so the arrays are bit-portable.  Near-duplicate boxes with equal classes make some predictions
    match a GT relation through a DIFFERENT box index (IoU >= 0.5), and a fraction of the GT predicates is
    boosted so that recall is neither 0 nor 1.  Also returns a zero-shot triplet table [Z, 3]
    (subject class, object class, predicate) that holds part of the GT triplets."""
    images, zs_rows = [], []
    for i, n in enumerate(num_objs):
        tag = "eval.%d." % i
        xy = uniform(seed, tag + "xy", (n, 2), 0.0, 300.0)
        wh = uniform(seed, tag + "wh", (n, 2), 30.0, 160.0)
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        classes = integers(seed, tag + "cls", (n,), 1, num_obj_cls)
        dup = uniform01(seed, tag + "dup", n)
        src = integers(seed, tag + "dupsrc", (n,), 0, max(n, 1))
        jit = uniform(seed, tag + "jit", (n, 4), -6.0, 6.0)
        for k in range(1, n):                     # ~30 %: a jittered copy of an earlier box, same class
            if dup[k] < 0.3:
                j = int(src[k]) % k
                boxes[k] = boxes[j] + jit[k]
                classes[k] = classes[j]
        pairs = np.array([(a, b) for a in range(n) for b in range(n) if a != b], dtype=np.int64).reshape(-1, 2)
        P = len(pairs)
        n_gt = int(min(P, 3 + integers(seed, tag + "ngt", (1,), 0, 22)[0] + (40 if n >= 15 else 0)))
        order = np.argsort(uniform01(seed, tag + "gtsel", P), kind="stable")[:n_gt]
        u = uniform01(seed, tag + "gtpred", n_gt)
        gt_pred = (1 + np.floor((u * u * np.sqrt(u)) * (num_rel_cls - 1))).astype(np.int64)   # skewed towards the head classes
        gt_rels = np.concatenate([pairs[order], gt_pred[:, None]], 1)
        if n_gt > 2:                              # a repeated pair with another predicate
            extra = gt_rels[:1].copy()
            extra[0, 2] = 1 + (extra[0, 2] % (num_rel_cls - 1))
            gt_rels = np.concatenate([gt_rels, extra], 0)
        w = uniform01(seed, tag + "w", P * num_rel_cls).reshape(P, num_rel_cls)
        w = w * w * w
        boost = uniform01(seed, tag + "boost", len(gt_rels))
        row_of = {(int(a), int(b)): r for r, (a, b) in enumerate(pairs)}
        for g, (s, o, r) in enumerate(gt_rels):
            if boost[g] < 0.65:
                w[row_of[(int(s), int(o))], int(r)] += 1.0 + 2.0 * boost[g]
        rel_scores = (w / w.sum(1, keepdims=True)).astype(np.float32)
        if mode == "predcls":
            pred_classes = classes.copy()
            obj_scores = np.ones(n, dtype=np.float32)
        else:
            flip = uniform01(seed, tag + "flip", n) < 0.2
            alt = integers(seed, tag + "alt", (n,), 1, num_obj_cls)
            pred_classes = np.where(flip, alt, classes)
            obj_scores = uniform(seed, tag + "objs", (n,), 0.3, 1.0)
        triple = rel_scores[:, 1:].max(1) * obj_scores[pairs[:, 0]] * obj_scores[pairs[:, 1]]
        srt = np.argsort(-triple, kind="stable")
        images.append({
            "gt_rels": gt_rels.astype(np.int64), "gt_classes": classes.astype(np.int64), "gt_boxes": boxes,
            "pred_rel_inds": pairs[srt], "rel_scores": rel_scores[srt], "pred_classes": pred_classes.astype(np.int64),
            "pred_boxes": boxes.copy(), "obj_scores": obj_scores,
        })
        zsel = uniform01(seed, tag + "zs", len(gt_rels)) < 0.35
        for (s, o, r) in gt_rels[zsel]:
            zs_rows.append((classes[s], classes[o], r))
    filler = integers(seed, "eval.zs_filler", (40, 3), 1, num_obj_cls)
    filler[:, 2] = 1 + filler[:, 2] % (num_rel_cls - 1)
    zeroshot = np.concatenate([np.array(zs_rows, dtype=np.int64).reshape(-1, 3), filler.astype(np.int64)], 0)
    return images, zeroshot


def synthetic_relation_targets(seed=41, num_objs=(6, 40, 3, 1), num_rel_cls=51):
    """(boxes [n, 4], relation matrix [n, n]) per image for the training-time relation sampler: random predicate matrices,
    one image with more foreground pairs than the positive budget, one with a single object (no candidate pair)."""
    out = []
    for i, n in enumerate(num_objs):
        boxes = uniform(seed, "rs.boxes.%d" % i, (n, 4), 0.0, 300.0)
        u = uniform01(seed, "rs.rel.%d" % i, n * n).reshape(n, n)
        lab = integers(seed, "rs.lab.%d" % i, (n, n), 1, num_rel_cls)
        rel = np.where(u < (0.5 if n >= 30 else 0.15), lab, 0)
        np.fill_diagonal(rel, 0)
        out.append((boxes, rel.astype(np.int64)))
    return out


# ---- ROI feature extraction fixtures ------------------------------------------------
def roi_test_boxes(rng, n, W, H):
    """n xyxy boxes on a W x H image: log-uniform sizes, some starting outside, and four fixed corner cases
    (the whole image, mostly outside, sub-pixel, malformed x2 < x1)."""
    xy = rng.uniform(-20, [W * 0.9, H * 0.9], size=(n, 2))
    wh = np.exp(rng.uniform(np.log(2), np.log(max(W, H) * 1.2), size=(n, 2)))
    b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    b[0] = [0, 0, W - 1, H - 1]
    b[1] = [W - 3, H - 3, W + 40, H + 40]
    b[2] = [30.2, 40.7, 30.3, 40.8]
    b[3] = [50, 60, 40, 30]
    return b


ROI_SINGLE_CASES = [(8, 2), (7, 2), (8, 1), (4, 4), (6, 0)]   # (pooled, sampling_ratio); 0 = adaptive grid (oracle only)


def synthetic_roi_single(pooled, ratio, channels=37):
    """One feature map [2, channels, 50, 84] at scale 1/16 and 23 ROI rows (image index, x1, y1, x2, y2)."""
    rng = np.random.RandomState(pooled * 10 + ratio)
    feat = rng.randn(2, channels, 50, 84).astype(np.float32)
    boxes = roi_test_boxes(rng, 23, 84 * 16, 50 * 16)
    rois = np.concatenate([rng.randint(0, 2, size=(23, 1)).astype(np.float32), boxes], 1)
    return feat, rois


def synthetic_roi_pyramid(channels=256, num_objs=(9, 17, 5), W=1024, H=640, seed=11):
    """Four FPN levels (strides 4..32) + a stride-16 depth map for 3 images, and per-image xyxy boxes that cover all four levels."""
    rng = np.random.RandomState(seed)
    feats = [rng.randn(len(num_objs), channels, H >> (2 + l), W >> (2 + l)).astype(np.float32) for l in range(4)]
    depth = rng.randn(len(num_objs), channels, H >> 4, W >> 4).astype(np.float32)
    boxes = [roi_test_boxes(rng, n, W, H) for n in num_objs]
    return feats, depth, boxes, (W, H)
