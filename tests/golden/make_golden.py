#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REAL reference.

Runs only in the build container (needs /root/reference); the GPU box never runs
it.  Nothing of the reference is copied: the script imports
`pysgg.modeling.roi_heads.relation_head.roi_relation_predictors` from
/root/reference with stand-ins for third-party modules that are absent here
(yacs, ipdb, h5py, cv2, torchvision, apex, ...), loads the
portable-RNG weights of `veto_amd.synth`, runs VETOPredictor /
VETOPredictor_MEET in eval mode on CPU fp32 and stores inputs-by-recipe +
outputs-by-value as .npz.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Cases (weights seed 0, data seed 7):
    predcls_n10_l6h6, predcls_n36_l6h6, predcls_n10_l4h8, predcls_n36_l4h8,
    sgcls_n10_l6h6, ragged_l4h8 (3 images of 5/1/9 boxes, the 1-box image using
    the [[0,0]] placeholder pair of sampling.py:50-51), meet_n10_l6h6,
    meet_sgcls_n10_l6h6, gqa-sized meet head list, train losses for predcls.
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from veto_amd import synth  # noqa: E402


# ---------------------------------------------------------------------------
# Import the reference with stand-ins for absent third-party modules.
# ---------------------------------------------------------------------------

class _CN(dict):
    """Minimal stand-in for yacs.config.CfgNode (attribute access over a dict)."""

    def __init__(self, init=None, **kw):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass


def import_reference():
    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = _CN
    yacs.config = yacs_config
    sys.modules["yacs"] = yacs
    sys.modules["yacs.config"] = yacs_config
    for name in [
        "ipdb", "h5py", "cv2", "pycocotools", "pycocotools.mask", "pycocotools.coco",
        "pycocotools.cocoeval", "torchvision", "torchvision.ops", "torchvision.transforms",
        "torchvision.transforms.functional", "torchvision.models", "torchvision.models.resnet",
        "torchvision.datasets", "torchvision.datasets.coco", "pysgg._C", "graphviz", "apex",
        "apex.amp", "gpustat", "tensorboardX", "termcolor", "overrides", "torchvision.ops.boxes",
        "torchvision.ops.misc", "torchvision.models.detection", "matplotlib", "matplotlib.pyplot",
        "seaborn", "PIL", "PIL.Image", "PIL.ImageDraw",
    ]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = mock.MagicMock(name=name)
    if not hasattr(torch, "_six"):
        six = types.ModuleType("torch._six")
        six.PY37 = True
        six.PY3 = True
        six.string_classes = (str,)
        six.int_classes = (int,)
        torch._six = six
        sys.modules["torch._six"] = six
    # apex is absent: its amp.float_function decorator (pysgg/layers/roi_align.py:57) only pins the op to fp32
    for m in ("apex", "apex.amp"):
        if isinstance(sys.modules[m], mock.MagicMock):
            sys.modules[m].float_function = lambda f: f
    if isinstance(sys.modules["apex"], mock.MagicMock):
        sys.modules["apex"].amp = sys.modules["apex.amp"]
    sys.path.insert(0, REF)
    import pysgg.modeling.roi_heads.relation_head.roi_relation_predictors as P
    from pysgg.config import cfg
    from pysgg.structures.bounding_box import BoxList
    return P, cfg, BoxList


def configure(P, cfg, mode, layers, heads, predictor="VETOPredictor", dataset="VG"):
    rh = cfg.MODEL.ROI_RELATION_HEAD
    rh.PREDICTOR = predictor
    rh.USE_GT_BOX = True
    rh.USE_GT_OBJECT_LABEL = (mode == "predcls")
    rh.VETOTRANSFORMER.ENC_LAYERS = layers
    rh.VETOTRANSFORMER.NHEADS = heads
    rh.VETOTRANSFORMER.T_INPUT_DIM = 576
    cfg.GLOBAL_SETTING.DATASET_CHOICE = dataset
    cfg.GLOBAL_SETTING.BETA_LOSS = False
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = False
    cfg.ENSEMBLE_LEARNING.TYPE = "group"
    n_obj, n_rel = (151, 51) if dataset == "VG" else (201, 101)
    stats = {"obj_classes": ["c%d" % i for i in range(n_obj)],
             "rel_classes": ["r%d" % i for i in range(n_rel)]}
    P.get_dataset_statistics = lambda c: stats
    P.obj_edge_vectors = lambda names, wv_dir, wv_dim: torch.zeros(len(names), wv_dim)
    return n_obj, n_rel


def make_proposals(BoxList, batch, mode):
    props, start = [], 0
    for n in batch["num_objs"]:
        sl = slice(start, start + n)
        b = BoxList(torch.from_numpy(batch["boxes"][sl]), batch["image_size"], mode="xyxy")
        b.add_field("labels", torch.from_numpy(batch["labels"][sl]))
        if mode != "predcls":
            b.add_field("predict_logits", torch.from_numpy(batch["predict_logits"][sl]))
            b.add_field("pred_labels", torch.from_numpy(batch["pred_labels"][sl]))
        props.append(b)
        start += n
    return props


def test_pairs(num_objs):
    """prepare_test_pairs of sampling.py:31-52 (GT-box branch), run verbatim through torch."""
    out = []
    for n in num_objs:
        cand = torch.ones((n, n)) - torch.eye(n)
        idxs = torch.nonzero(cand).view(-1, 2)
        out.append(idxs if len(idxs) > 0 else torch.zeros((1, 2), dtype=torch.int64))
    return out


def load_sd(module, sd_np):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    # Only buffers we do not synthesise may be missing.
    assert not unexpected, unexpected
    assert all("criterion" in m or "CE_loss" in m for m in missing), missing


def reference_test_pairs(props, num_objs):
    """The reference's own RelationSampling.prepare_test_pairs (sampling.py:31-52) with the GT-box settings and
    MAX_PROPOSAL_PAIR = 2048 (defaults.py): images with more candidate pairs keep the 2048 best by pred_scores product.
    `pred_scores` comes from the portable RNG, one stream per image."""
    from pysgg.modeling.roi_heads.relation_head.sampling import RelationSampling
    for i, (p, n) in enumerate(zip(props, num_objs)):
        p.add_field("pred_scores", torch.from_numpy(synth.uniform01(7, "pred_scores.%d" % i, n).astype(np.float32)))
    samp = RelationSampling(0.5, False, 4, 1024, 0.25, 2048, True, False)   # max_proposal_pairs 2048, use_gt_box True
    return samp.prepare_test_pairs(torch.device("cpu"), props)


def run_case(P, cfg, BoxList, name, mode, layers, heads, num_objs, meet=False, dataset="VG",
             train=False, experts=False, capped_pairs=False):
    n_obj, n_rel = configure(P, cfg, mode, layers, heads,
                             "VETOPredictor_MEET" if meet else "VETOPredictor", dataset)
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = bool(experts)   # defaults.py:864 default True: 3 experts per group
    torch.manual_seed(0)
    if meet:
        model = P.VETOPredictor_MEET(cfg, 512)
        groups = list(model.max_group_element_number_list)
        sd = synth.meet_state_dict(0, groups, layers=layers, num_obj_cls=n_obj, experts=3 if experts else 0)
    else:
        model = P.VETOPredictor(cfg, 512)
        sd = synth.predictor_state_dict(0, layers=layers, num_obj_cls=n_obj, num_rel_cls=n_rel)
    load_sd(model, sd)
    model.eval()
    batch = synth.synthetic_batch(7, len(num_objs), list(num_objs), num_obj_cls=n_obj)
    props = make_proposals(BoxList, batch, mode)
    pairs = reference_test_pairs(props, batch["num_objs"]) if capped_pairs else test_pairs(batch["num_objs"])
    rgb = torch.from_numpy(batch["roi_features"])
    dep = torch.from_numpy(batch["roi_depth_features"])
    out = {"layers": layers, "heads": heads, "num_objs": np.array(batch["num_objs"]),
           "mode": mode, "dataset": dataset, "meet": int(meet), "experts": int(bool(experts)),
           "capped_pairs": int(bool(capped_pairs)), "pair_counts": np.array([len(p) for p in pairs]),
           # the checkpoint contract: every key of the reference module's state dict
           "state_dict_keys": np.array(sorted(model.state_dict().keys()))}
    with torch.no_grad():
        res = model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)
    obj_dists, rel_dists = res[0], res[1]
    out["pair_idx"] = torch.cat(pairs, 0).numpy()
    out["obj_dists_argmax"] = torch.cat([o.argmax(1) for o in obj_dists]).numpy()
    if meet:
        for k, v in rel_dists.items():
            out["rel_" + k] = v.numpy()
        out["incre_idx_list"] = np.array(res[3])
        out["group_sizes"] = np.array(groups)
    else:
        out["rel_dists"] = torch.cat(list(rel_dists), 0).numpy()
        # intermediate probes through the reference's own sub-modules
        with torch.no_grad():
            tr = model.fusion_transformer
            hooks = {}
            h = tr.transformer.register_forward_hook(lambda m, i, o: hooks.__setitem__("tokens", o))
            model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)
            h.remove()
        tok = hooks["tokens"]
        step = 7 if tok.shape[0] < 200 else 97
        out["tokens_sample"] = tok[::step, :, ::9].numpy()  # strided sample of [P,19,576]
        out["tokens_step"] = step
    if train and not meet:
        model.train()
        # dropout off so the loss is deterministic; BN keeps batch statistics (train semantics)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        P_tot = sum(len(p) for p in pairs)
        rel_labels = torch.from_numpy(synth.integers(7, "rel_labels", (P_tot,), 0, n_rel))
        splits = [len(p) for p in pairs]
        res = model(props, pairs, list(rel_labels.split(splits)), None, roi_features=rgb,
                    roi_depth_features=dep)
        out["train_rel_loss"] = np.array(res[2]["rel_loss"].item(), dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    sz = os.path.getsize(path)
    key = "rel_dists" if not meet else ("rel_group_01" if experts else "rel_group_0")
    print("%-24s %s %s  |max|=%.3f  %d bytes" % (name, mode, out[key].shape, np.abs(out[key]).max(), sz))


def run_postprocessor(cfg, BoxList, name, num_objs, onehot):
    """The reference's own PostProcessor (inference.py:9-92,398-453) on portable-RNG logits."""
    from pysgg.modeling.roi_heads.relation_head.inference import make_roi_relation_post_processor
    cfg.MODEL.ROI_RELATION_HEAD.USE_GT_BOX = True
    cfg.ENSEMBLE_LEARNING.ENABLED = False
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = False   # configs/VETO_final.yaml:154; inference.py:93 tests this flag alone
    cfg.MODEL.ATTRIBUTE_ON = False
    post = make_roi_relation_post_processor(cfg).eval()
    n_obj, P_list = sum(num_objs), [max(n * (n - 1), 1) for n in num_objs]
    rel_logits = torch.from_numpy(synth.normal(21, "post.rel_logits", (sum(P_list), 51), 0.0, 2.0))
    if onehot:   # predcls: relation_head.py:109 overloads predict_logits with +-1000 one-hots
        lab = torch.from_numpy(synth.integers(21, "post.labels", (n_obj,), 1, 151))
        obj_logits = torch.full((n_obj, 151), -1000.0)
        obj_logits[torch.arange(n_obj), lab] = 1000.0
    else:
        obj_logits = torch.from_numpy(synth.normal(21, "post.obj_logits", (n_obj, 151), 0.0, 3.0))
    pairs = test_pairs(num_objs)
    boxes = [BoxList(torch.zeros(n, 4), (800, 600), mode="xyxy") for n in num_objs]
    with torch.no_grad():
        res = post((list(rel_logits.split(P_list)), list(obj_logits.split(list(num_objs)))), pairs, boxes)
    out = {"num_objs": np.array(num_objs), "onehot": int(onehot)}
    for i, r in enumerate(res):
        for f in ("pred_labels", "pred_scores", "rel_pair_idxs", "pred_rel_scores", "pred_rel_labels"):
            out["%s_%d" % (f, i)] = r.get_field(f).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-24s postprocessor %s images, %d pairs" % (name, len(res), sum(P_list)))


def run_postprocessor_meet(cfg, BoxList, name, n, dataset):
    """The reference's MEET merge branch (inference.py:284-397) on portable-RNG group logits.  The
    branch calls .cuda() unconditionally; in this CPU-only container Tensor.cuda is patched to the
    identity for the duration of the call (a stand-in for the absent device, not for reference code)."""
    from pysgg.modeling.roi_heads.relation_head.inference import make_roi_relation_post_processor
    from SHA_GCL_extra.group_chosen_function import get_group_splits
    from SHA_GCL_extra.extra_function_utils import get_current_predicate_idx
    cfg.MODEL.ROI_RELATION_HEAD.USE_GT_BOX = True
    cfg.ENSEMBLE_LEARNING.ENABLED = True
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = False
    cfg.ENSEMBLE_LEARNING.TYPE = ["group"]
    cfg.GLOBAL_SETTING.DATASET_CHOICE = dataset
    cfg.MODEL.ATTRIBUTE_ON = False
    post = make_roi_relation_post_processor(cfg).eval()
    stage_list, sizes = get_group_splits(dataset, "divide4")
    incre_idx_list = get_current_predicate_idx(stage_list, 0.1, dataset)[0]
    n_objc = 151 if dataset == "VG" else 201
    P_ = n * (n - 1)
    rel = {"group_%d" % k: torch.from_numpy(synth.normal(23, "meet.group_%d" % k, (P_, g + 2), 0.0, 2.0))
           for k, g in enumerate(sizes)}
    obj_logits = torch.from_numpy(synth.normal(23, "meet.obj_logits", (n, n_objc), 0.0, 3.0))
    pairs = test_pairs([n])
    boxes = [BoxList(torch.zeros(n, 4), (800, 600), mode="xyxy")]
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            res = post((rel, [obj_logits]), pairs, boxes, incre_idx_list=incre_idx_list, ensemble=True)
    finally:
        torch.Tensor.cuda = orig
    r = res[0]
    out = {"n": n, "dataset": dataset, "group_sizes": np.array(sizes), "incre_idx_list": np.array(incre_idx_list)}
    for f in ("pred_labels", "pred_scores", "rel_pair_idxs", "pred_rel_scores", "pred_rel_labels"):
        out[f] = r.get_field(f).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-24s MEET merge: %d rows, pair dtype %s" % (name, out["rel_pair_idxs"].shape[0], out["rel_pair_idxs"].dtype))


def run_postprocessor_vote(cfg, BoxList, name, n, dataset, voting):
    """The reference's EXPERT_GROUP voting branch (inference.py:93-283): three experts per group,
    VOTING 'C' (two of three agree) or 'U' (all agree), on portable-RNG logits.  Expert logits share a
    common component so that all agreement patterns (none / one pair / all) occur.  Tensor.cuda is
    patched to the identity as in run_postprocessor_meet."""
    from pysgg.modeling.roi_heads.relation_head.inference import make_roi_relation_post_processor
    from SHA_GCL_extra.group_chosen_function import get_group_splits
    from SHA_GCL_extra.extra_function_utils import get_current_predicate_idx
    cfg.MODEL.ROI_RELATION_HEAD.USE_GT_BOX = True
    cfg.ENSEMBLE_LEARNING.ENABLED = True
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = True
    cfg.ENSEMBLE_LEARNING.VOTING = voting
    cfg.ENSEMBLE_LEARNING.TYPE = ["group"]
    cfg.GLOBAL_SETTING.DATASET_CHOICE = dataset
    cfg.MODEL.ATTRIBUTE_ON = False
    post = make_roi_relation_post_processor(cfg).eval()
    stage_list, sizes = get_group_splits(dataset, "divide4")
    incre_idx_list = get_current_predicate_idx(stage_list, 0.1, dataset)[0]
    n_objc = 151 if dataset == "VG" else 201
    P_ = n * (n - 1)
    rel = {}
    for k, g in enumerate(sizes):
        base = synth.normal(29, "vote.base_%d" % k, (P_, g + 2), 0.0, 1.5)
        for e in range(3):
            own = synth.normal(29, "vote.group_%d%d" % (k, e + 1), (P_, g + 2), 0.0, 1.0)
            rel["group_%d%d" % (k, e + 1)] = torch.from_numpy((base + own).astype(np.float32))
    obj_logits = torch.from_numpy(synth.normal(29, "vote.obj_logits", (n, n_objc), 0.0, 3.0))
    pairs = test_pairs([n])
    boxes = [BoxList(torch.zeros(n, 4), (800, 600), mode="xyxy")]
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.no_grad():
            res = post((rel, [obj_logits]), pairs, boxes, incre_idx_list=incre_idx_list, ensemble=True)
    finally:
        torch.Tensor.cuda = orig
    r = res[0]
    out = {"n": n, "dataset": dataset, "voting": voting, "group_sizes": np.array(sizes),
           "incre_idx_list": np.array(incre_idx_list)}
    for f in ("pred_labels", "pred_scores", "rel_pair_idxs", "pred_rel_scores", "pred_rel_labels"):
        out[f] = r.get_field(f).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-24s expert voting %s: %d of %d rows kept" % (name, voting, out["rel_pair_idxs"].shape[0], len(sizes) * P_))


def run_train_losses(P, cfg, BoxList, name, meet, beta_loss=False, dataset="VG", mode="predcls", experts=False, layers=2, n_heads=8,
                     num_objs=(7, 5, 9)):
    """Training-mode forward of the reference predictor (dropout off, so it is deterministic) on a small batch with
    random relation labels: stores the classifier logits it produced (forward hooks on the rel_out modules), the
    labels, and the losses it returned.  MEET: also the expert sampling (`cur_chosen_matrix`), which the reference
    draws from Python's `random` (seeded with 1, tools/relation_train_net.py:44-50), and its sample_rate_matrix."""
    import random
    n_obj, n_rel = configure(P, cfg, mode, layers, n_heads, "VETOPredictor_MEET" if meet else "VETOPredictor", dataset)
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = bool(experts)
    cfg.GLOBAL_SETTING.BETA_LOSS = False
    torch.manual_seed(0)
    num_objs = list(num_objs)
    if meet:
        model = P.VETOPredictor_MEET(cfg, 512)
        groups = list(model.max_group_element_number_list)
        sd = synth.meet_state_dict(0, groups, layers=layers, num_obj_cls=n_obj, experts=3 if experts else 0)
    else:
        model = P.VETOPredictor(cfg, 512)
        sd = synth.predictor_state_dict(0, layers=layers, num_obj_cls=n_obj, num_rel_cls=n_rel)
    load_sd(model, sd)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    batch = synth.synthetic_batch(7, len(num_objs), num_objs, num_obj_cls=n_obj)
    props = make_proposals(BoxList, batch, mode)
    pairs = test_pairs(num_objs)
    P_tot = sum(len(p) for p in pairs)
    u = synth.uniform01(9, "train.labels", P_tot)
    labels = np.where(u < 0.55, 0, 1 + np.floor((u - 0.55) / 0.45 * (n_rel - 1))).astype(np.int64)   # ~55 % background
    labels = np.minimum(labels, n_rel - 1)
    rel_labels = torch.from_numpy(labels)
    captured = {}
    heads = ([h for lst in model.model.rel_out_group for h in lst] if experts else list(model.model.rel_out)) if meet else [model.rel_out]
    hooks = [h.register_forward_hook(lambda mod, i, o, k=k: captured.__setitem__(k, o.detach().clone())) for k, h in enumerate(heads)]
    if beta_loss:   # roi_relation_predictors.py:4057-4066 with the counts of pred_counts.pkl (the path there is absolute)
        import pickle
        with open(os.path.join(REF, "pred_counts.pkl"), "rb") as f:
            counts = np.asarray(pickle.load(f), dtype=np.float64)
        counts[::-1].sort()
        w = (1.0 - 0.999) / (1 - (0.999 ** counts))
        w *= float(n_rel) / np.sum(w)
        model.criterion_loss_rel = torch.nn.CrossEntropyLoss(weight=torch.FloatTensor(w))
    random.seed(1)
    # the ROI maps are leaves of the autograd graph too: the reference trains its depth backbone through roi_depth_features
    # (tools/relation_train_net.py:166-170), so their gradients are part of the training contract
    roi_in = {"roi_features": torch.from_numpy(batch["roi_features"]).requires_grad_(True),
              "roi_depth_features": torch.from_numpy(batch["roi_depth_features"]).requires_grad_(True)}
    res = model(props, pairs, list(rel_labels.split([len(p) for p in pairs])), None,
                roi_features=roi_in["roi_features"], roi_depth_features=roi_in["roi_depth_features"])
    for h in hooks:
        h.remove()
    out = {"labels": labels, "meet": int(meet), "dataset": dataset, "beta_loss": int(beta_loss), "num_objs": np.array(num_objs),
           "mode": mode, "experts": int(bool(experts)), "layers": layers, "heads": n_heads}
    # gradients of the summed losses w.r.t. every parameter (the reference's training loop sums the loss dict,
    # engine/trainer + tools/relation_train_net.py:297): stored as norm + a strided sample (full tensor when small)
    sum(res[2].values()).backward()
    for pname, prm in model.named_parameters():
        if prm.grad is None:
            continue
        gflat = prm.grad.detach().reshape(-1).numpy()
        step = max(1, gflat.size // 512)
        out["gradnorm_" + pname] = np.array(float(np.linalg.norm(gflat.astype(np.float64))))
        out["gradsample_" + pname] = gflat[::step].copy()
        out["gradstep_" + pname] = np.array(step)
    for iname, t in roi_in.items():
        gflat = t.grad.detach().reshape(-1).numpy()
        step = max(1, gflat.size // 2048)
        out["inputgradnorm_" + iname] = np.array(float(np.linalg.norm(gflat.astype(np.float64))))
        out["inputgradsample_" + iname] = gflat[::step].copy()
        out["inputgradstep_" + iname] = np.array(step)
    for k, v in captured.items():
        out["logits_%d" % k] = v.numpy()
    for k, v in res[2].items():
        out["loss_" + k] = np.array(float(v.item()), dtype=np.float64)
    if beta_loss:
        out["class_weights"] = w.astype(np.float32)
    if meet:
        chosen = res[4][0]          # expert_dist[0] is cur_chosen_matrix itself (appended by reference)
        for k, rows in enumerate(chosen):
            out["chosen_%d" % k] = np.array(rows, dtype=np.int64)
        out["sample_rate_matrix"] = np.array(model.sample_rate_matrix, dtype=np.float64)
        out["incre_idx_list"] = np.array(model.incre_idx_list)
        out["group_sizes"] = np.array(groups)
        out["random_after"] = np.array([random.random()], dtype=np.float64)   # the next draw: pins how much was consumed
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-24s losses %s" % (name, {k: round(float(v.item()), 5) for k, v in res[2].items()}))


def relsample_inputs(BoxList_cls):
    props, targets = [], []
    for boxes, rel in synth.synthetic_relation_targets():
        b = torch.from_numpy(boxes)
        t = BoxList_cls(b.clone(), (800, 600), mode="xyxy")
        t.add_field("relation", torch.from_numpy(rel))
        props.append(BoxList_cls(b, (800, 600), mode="xyxy"))
        targets.append(t)
    return props, targets


def run_relsample(BoxList, name):
    """The reference's RelationSampling.gtbox_relsample (sampling.py:54-107) with VETO_final.yaml's budget (1024 pairs per
    image, a quarter of them foreground at most), torch seeded with 0."""
    from pysgg.modeling.roi_heads.relation_head.sampling import RelationSampling
    samp = RelationSampling(0.5, False, 4, 1024, 0.25, 2048, True, False)
    props, targets = relsample_inputs(BoxList)
    torch.manual_seed(0)
    _, labels, pairs, binaries = samp.gtbox_relsample(props, targets)
    out = {}
    for i, (l, p, b) in enumerate(zip(labels, pairs, binaries)):
        out["labels_%d" % i], out["pairs_%d" % i], out["binary_%d" % i] = l.numpy(), p.numpy(), b.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-24s %s pairs per image, foreground %s" % (name, [len(p) for p in pairs], [int((l > 0).sum()) for l in labels]))


SGG_EVAL_CASES = {"sggeval_predcls": (31, [6, 9, 12, 3, 15, 20, 2, 8], "predcls"),
                  "sggeval_sgcls": (32, [6, 9, 12, 3, 15, 20, 2, 8], "sgcls")}


def run_sgg_eval(cfg, BoxList, name):
    """The reference's relation evaluators (sgg_eval.py) driven by its own per-image routine
    (vg_eval.py:459-566 evaluate_relation_of_one_image) exactly as do_vg_evaluation does (:330-420), on
    veto_amd.synth.synthetic_eval_images.  Stores the result_dict entries."""
    import pysgg.data.datasets.evaluation.vg.vg_eval as ve
    from pysgg.data.datasets.evaluation.vg.sgg_eval import (SGMeanRecall, SGNGMeanRecall, SGNoGraphConstraintRecall,
                                                             SGPairAccuracy, SGRecall, SGZeroShotRecall)
    seed, num_objs, mode = SGG_EVAL_CASES[name]
    num_rel = 51
    images, zeroshot = synth.synthetic_eval_images(seed, num_objs, mode, num_rel_cls=num_rel)
    rd = {}
    names = ["r%d" % i for i in range(num_rel)]
    evaluator = {"eval_recall": SGRecall(rd), "eval_nog_recall": SGNoGraphConstraintRecall(rd),
                 "eval_zeroshot_recall": SGZeroShotRecall(rd), "eval_pair_accuracy": SGPairAccuracy(rd),
                 "eval_mean_recall": SGMeanRecall(rd, num_rel, names, print_detail=True),
                 "eval_ng_mean_recall": SGNGMeanRecall(rd, num_rel, names, print_detail=True)}
    for e in evaluator.values():
        e.register_container(mode)
    gc = {"zeroshot_triplet": zeroshot, "result_dict": rd, "mode": mode, "multiple_preds": False,
          "num_rel_category": num_rel, "iou_thres": 0.5, "attribute_on": False, "num_attributes": 201}
    for img in images:
        gt = BoxList(torch.from_numpy(img["gt_boxes"]), (800, 600), mode="xyxy")
        gt.add_field("relation_tuple", torch.from_numpy(img["gt_rels"]))
        gt.add_field("labels", torch.from_numpy(img["gt_classes"]))
        pr = BoxList(torch.from_numpy(img["pred_boxes"]), (800, 600), mode="xyxy")
        pr.add_field("rel_pair_idxs", torch.from_numpy(img["pred_rel_inds"]))
        pr.add_field("pred_rel_scores", torch.from_numpy(img["rel_scores"]))
        pr.add_field("pred_labels", torch.from_numpy(img["pred_classes"]))
        pr.add_field("pred_scores", torch.from_numpy(img["obj_scores"]))
        ve.evaluate_relation_of_one_image(gt, pr, gc, evaluator)
    evaluator["eval_mean_recall"].calculate_mean_recall(mode)
    evaluator["eval_ng_mean_recall"].calculate_mean_recall(mode)
    out = {"mode": mode, "seed": seed, "num_objs": np.array(num_objs), "num_rel": num_rel}
    for key in ("recall", "recall_nogc", "zeroshot_recall", "accuracy_hit", "accuracy_count"):
        for k in (20, 50, 100):
            out["%s_%d" % (key, k)] = np.array(rd["%s_%s" % (mode, key)][k], dtype=np.float64)
    for key in ("mean_recall", "ng_mean_recall"):
        for k in (20, 50, 100):
            out["%s_%d" % (key, k)] = np.array(rd["%s_%s" % (mode, key)][k], dtype=np.float64)
            out["%s_list_%d" % (key, k)] = np.array(rd["%s_%s_list" % (mode, key)][k], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-24s %s R@20/50/100 = %s  mR@100 = %.4f  ngR@100 = %.4f" % (
        name, mode, [round(float(np.mean(out["recall_%d" % k])), 4) for k in (20, 50, 100)],
        float(out["mean_recall_100"]), float(np.mean(out["recall_nogc_100"]))))
    print(evaluator["eval_recall"].generate_print_string(mode) + evaluator["eval_pair_accuracy"].generate_print_string(mode), end="")


# ---------------------------------------------------------------------------
# ROI feature extraction: the reference's OWN ROIAlign arithmetic and Pooler.
# ---------------------------------------------------------------------------
def _reference_roi_align_forward():
    """The binding the reference's C++ extension exposes as `_C.roi_align_forward` (csrc/ROIAlign.h:10-24 ->
    cpu/ROIAlign_cpu.cpp:221-257).  That 35-line ATen wrapper does not build against torch 2.10; the kernel templates it
    calls (ROIAlign_cpu.cpp:1-219) do, unmodified: oracle/build_ref.sh compiles them into oracle/_ref/libroialign_ref.so and
    this function is the wrapper's equivalent over ctypes (same argument order, same output allocation)."""
    import ctypes
    import subprocess
    path = os.path.join(REPO, "oracle", "_ref", "libroialign_ref.so")
    if not os.path.exists(path):
        subprocess.check_call([os.path.join(REPO, "oracle", "build_ref.sh")])
    lib = ctypes.CDLL(path)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.veto_ref_roi_align_forward.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_float,
                                               ctypes.c_int, ctypes.c_int, fp]
    lib.veto_ref_roi_align_forward.restype = None

    def forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio):
        assert pooled_height == pooled_width and input.dtype == torch.float32 and not input.is_cuda
        inp, r = input.contiguous(), rois.contiguous().float()
        out = torch.empty((r.shape[0], inp.shape[1], pooled_height, pooled_width), dtype=torch.float32)
        if out.numel():
            lib.veto_ref_roi_align_forward(ctypes.cast(inp.data_ptr(), fp), inp.shape[1], inp.shape[2], inp.shape[3],
                                           ctypes.cast(r.data_ptr(), fp), r.shape[0], float(spatial_scale), pooled_height,
                                           sampling_ratio, ctypes.cast(out.data_ptr(), fp))
        return out
    return forward


ROI_KEEP_CHANNELS = 6      # ROIAlign treats channels independently: the fixtures keep the first few


def run_roialign(BoxList):
    """roialign_single.npz: the reference's ROIAlign layer (layers/roi_align.py:50-61 -> ROIAlign_cpu.cpp) on
    veto_amd.synth.synthetic_roi_single; roialign_pooler.npz: the reference's Pooler (poolers.py:45-171, cat_all_levels False,
    the way VETOFeatureExtractor builds it) -- LevelMapper, convert_to_roi_format, per-level dispatch, the fixed 1/16 depth
    pooler -- on veto_amd.synth.synthetic_roi_pyramid."""
    sys.modules["pysgg._C"].roi_align_forward = _reference_roi_align_forward()
    import pysgg
    pysgg._C = sys.modules["pysgg._C"]
    from pysgg.layers.roi_align import ROIAlign
    from pysgg.modeling.poolers import LevelMapper, Pooler
    out = {}
    for pooled, ratio in synth.ROI_SINGLE_CASES:
        feat, rois = synth.synthetic_roi_single(pooled, ratio, channels=ROI_KEEP_CHANNELS)
        y = ROIAlign((pooled, pooled), 1.0 / 16, ratio)(torch.from_numpy(feat), torch.from_numpy(rois))
        out["rois_p%d_r%d" % (pooled, ratio)] = rois
        out["out_p%d_r%d" % (pooled, ratio)] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "roialign_single.npz"), **out)
    print("%-24s %s" % ("roialign_single", {k: v.shape for k, v in out.items() if k.startswith("out")}))

    feats, depth, boxes, size = synth.synthetic_roi_pyramid(channels=ROI_KEEP_CHANNELS)
    props = [BoxList(torch.from_numpy(b), size, mode="xyxy") for b in boxes]
    pooler = Pooler((8, 8), (0.25, 0.125, 0.0625, 0.03125), 2, in_channels=ROI_KEEP_CHANNELS, cat_all_levels=False)
    rgb, dep = pooler([torch.from_numpy(f) for f in feats], props, depth_features=torch.from_numpy(depth))
    levels = pooler.map_levels(props)
    rois = pooler.convert_to_roi_format(props)
    # the level boundaries of test_roi_align.py::test_level_mapper_boundaries through the reference's LevelMapper
    def box(w, h):
        return [10.0, 20.0, 10.0 + w - 1, 20.0 + h - 1]
    bnd = np.array([box(8, 8), box(111, 111), box(112, 112), box(223, 223), box(224, 224), box(447, 447), box(448, 448),
                    box(2000, 2000), box(56, 224)], dtype=np.float32)
    bnd_levels = LevelMapper(2, 5)([BoxList(torch.from_numpy(bnd), (4000, 4000), mode="xyxy")])
    np.savez_compressed(os.path.join(HERE, "roialign_pooler.npz"), rgb=rgb.numpy(), depth=dep.numpy(), levels=levels.numpy(),
                        rois=rois.numpy(), boundary_boxes=bnd, boundary_levels=bnd_levels.numpy())
    print("%-24s rgb %s depth %s levels %s" % ("roialign_pooler", tuple(rgb.shape), tuple(dep.shape), np.bincount(levels.numpy().astype(np.int64)).tolist()))


# Full-size cases (round 2): the BASELINE.json workloads that round 1 only exercised at <= 14 objects per image.
RAGGED12 = [1, 2, 64, 36, 7, 50, 13, 3, 46, 20, 36, 5]     # 64 / 50 / 46 objects exceed MAX_PROPOSAL_PAIR = 2048 candidates


def predictor_cases(P, cfg, BoxList):
    run_case(P, cfg, BoxList, "predcls_n10_l6h6", "predcls", 6, 6, [10], train=True)
    run_case(P, cfg, BoxList, "predcls_n10_l4h8", "predcls", 4, 8, [10])
    run_case(P, cfg, BoxList, "predcls_n36_l6h6", "predcls", 6, 6, [36])
    run_case(P, cfg, BoxList, "predcls_n36_l4h8", "predcls", 4, 8, [36])
    run_case(P, cfg, BoxList, "sgcls_n10_l6h6", "sgcls", 6, 6, [10])
    run_case(P, cfg, BoxList, "ragged_l4h8", "predcls", 4, 8, [5, 1, 9])
    run_case(P, cfg, BoxList, "meet_n10_l6h6", "predcls", 6, 6, [10], meet=True)
    run_case(P, cfg, BoxList, "meet_sgcls_n10_l6h6", "sgcls", 6, 6, [10], meet=True)
    run_case(P, cfg, BoxList, "meet_gqa_n6_l4h8", "predcls", 4, 8, [6], meet=True, dataset="GQA")
    run_case(P, cfg, BoxList, "meetx_n10_l4h8", "predcls", 4, 8, [10], meet=True, experts=True)
    # cfg-4 per-GPU workload: sgcls, 12 images x 36 objects
    run_case(P, cfg, BoxList, "sgcls_b12_n36_l4h8", "sgcls", 4, 8, [36] * 12)
    # the shipped architecture (configs/VETO_final.yaml: 6 layers x 6 heads) on the 12 x 36 batch
    run_case(P, cfg, BoxList, "predcls_b12_n36_l6h6", "predcls", 6, 6, [36] * 12)
    # cfg-5 heads at 36 objects: VG (5 group heads) and GQA (4 group heads)
    run_case(P, cfg, BoxList, "meet_n36_l6h6", "predcls", 6, 6, [36], meet=True)
    run_case(P, cfg, BoxList, "meet_gqa_n36_l4h8", "predcls", 4, 8, [36], meet=True, dataset="GQA")
    # one ragged 12-image batch, 1 .. 64 objects, with the reference's own 2048-pair cap on three images
    run_case(P, cfg, BoxList, "ragged12_capped_l4h8", "predcls", 4, 8, RAGGED12, capped_pairs=True)


def main():
    torch.set_num_threads(8)
    P, cfg, BoxList = import_reference()
    if os.environ.get("GOLDEN_ONLY") == "roialign":    # regenerate only the ROI feature extraction fixtures
        run_roialign(BoxList)
        return
    if os.environ.get("GOLDEN_ONLY") == "relsample":
        run_relsample(BoxList, "relsample_gtbox")
        return
    if os.environ.get("GOLDEN_ONLY") == "train":   # regenerate only the training-loss fixtures
        run_train_losses(P, cfg, BoxList, "train_vanilla", meet=False)
        run_train_losses(P, cfg, BoxList, "train_vanilla_beta", meet=False, beta_loss=True)
        run_train_losses(P, cfg, BoxList, "train_meet_vg", meet=True)
        run_train_losses(P, cfg, BoxList, "train_meet_gqa", meet=True, dataset="GQA")
        run_train_losses(P, cfg, BoxList, "train_vanilla_sgcls", meet=False, mode="sgcls")
        run_train_losses(P, cfg, BoxList, "train_meet_sgcls", meet=True, mode="sgcls")
        run_train_losses(P, cfg, BoxList, "train_meet_experts", meet=True, experts=True)
        run_train_losses(P, cfg, BoxList, "train_vanilla_l1h6_ragged", meet=False, layers=1, n_heads=6, num_objs=(2, 6, 3))
        run_train_losses(P, cfg, BoxList, "train_meet_l3h4", meet=True, layers=3, n_heads=4, num_objs=(4, 5))
        return
    if os.environ.get("GOLDEN_ONLY") == "sggeval":   # regenerate only the evaluator fixtures
        for name in SGG_EVAL_CASES:
            run_sgg_eval(cfg, BoxList, name)
        return
    if os.environ.get("GOLDEN_ONLY") == "predictor":   # regenerate only the predictor forward fixtures
        predictor_cases(P, cfg, BoxList)
        return
    if os.environ.get("GOLDEN_ONLY") == "experts":   # regenerate only the EXPERT_GROUP fixtures
        run_postprocessor_vote(cfg, BoxList, "postvote_vg_c_n10", 10, "VG", "C")
        run_postprocessor_vote(cfg, BoxList, "postvote_vg_u_n10", 10, "VG", "U")
        run_postprocessor_vote(cfg, BoxList, "postvote_gqa_c_n7", 7, "GQA", "C")
        run_case(P, cfg, BoxList, "meetx_n10_l4h8", "predcls", 4, 8, [10], meet=True, experts=True)
        return
    run_roialign(BoxList)
    run_relsample(BoxList, "relsample_gtbox")
    run_train_losses(P, cfg, BoxList, "train_vanilla", meet=False)
    run_train_losses(P, cfg, BoxList, "train_vanilla_beta", meet=False, beta_loss=True)
    run_train_losses(P, cfg, BoxList, "train_meet_vg", meet=True)
    run_train_losses(P, cfg, BoxList, "train_meet_gqa", meet=True, dataset="GQA")
    run_train_losses(P, cfg, BoxList, "train_vanilla_sgcls", meet=False, mode="sgcls")
    run_train_losses(P, cfg, BoxList, "train_meet_sgcls", meet=True, mode="sgcls")
    run_train_losses(P, cfg, BoxList, "train_meet_experts", meet=True, experts=True)
    run_train_losses(P, cfg, BoxList, "train_vanilla_l1h6_ragged", meet=False, layers=1, n_heads=6, num_objs=(2, 6, 3))
    run_train_losses(P, cfg, BoxList, "train_meet_l3h4", meet=True, layers=3, n_heads=4, num_objs=(4, 5))
    for name in SGG_EVAL_CASES:
        run_sgg_eval(cfg, BoxList, name)
    run_postprocessor_vote(cfg, BoxList, "postvote_vg_c_n10", 10, "VG", "C")
    run_postprocessor_vote(cfg, BoxList, "postvote_vg_u_n10", 10, "VG", "U")
    run_postprocessor_vote(cfg, BoxList, "postvote_gqa_c_n7", 7, "GQA", "C")
    run_postprocessor_meet(cfg, BoxList, "postmeet_vg_n10", 10, "VG")
    run_postprocessor_meet(cfg, BoxList, "postmeet_gqa_n7", 7, "GQA")
    run_postprocessor(cfg, BoxList, "post_sgcls_ragged", [5, 1, 9], onehot=False)
    run_postprocessor(cfg, BoxList, "post_predcls_n36", [36], onehot=True)
    predictor_cases(P, cfg, BoxList)
    # BETA_LOSS data (roi_relation_predictors.py:4058-4066 reads this pickle): 51 predicate counts
    import pickle
    with open(os.path.join(REF, "pred_counts.pkl"), "rb") as f:
        counts = np.asarray(pickle.load(f), dtype=np.float64)
    np.savetxt(os.path.join(HERE, "pred_counts.txt"), counts, fmt="%.1f")


if __name__ == "__main__":
    main()
