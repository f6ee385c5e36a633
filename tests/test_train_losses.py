"""Training-time losses and MEET expert sampling (partial): the oracle restatement is
pinned to what the reference predictor produced in training mode (tests/golden/train_*.npz); the HIP kernels are
compared with both."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import train_oracle as to
from veto_amd import meet_tables

# *_l1h6_ragged: one layer (first == last), 96-wide heads, images of 2 / 6 / 3 objects; *_l3h4: three layers, 144-wide heads
CASES_VANILLA = ["train_vanilla", "train_vanilla_beta", "train_vanilla_sgcls", "train_vanilla_l1h6_ragged"]
CASES_MEET = ["train_meet_vg", "train_meet_gqa", "train_meet_sgcls", "train_meet_l3h4"]
CASES_EXPERTS = ["train_meet_experts"]     # EXPERT_GROUP: 3 experts per group, 15 heads


def _load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def _words_after_seed(seed, n):
    """n raw MT19937 words of Python's `random` after random.seed(seed), through numpy's identical generator."""
    random.seed(seed)
    st = random.getstate()
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": np.array(st[1][:624], dtype=np.uint32), "pos": st[1][624]}}
    return bg.random_raw(n).astype(np.uint32), st


@pytest.mark.parametrize("name", CASES_VANILLA)
def test_oracle_weighted_ce_matches_reference_loss(name):
    g = _load(name)
    w = g["class_weights"] if int(g["beta_loss"]) else None
    loss, grad = to.weighted_ce(g["logits_0"], g["labels"], w)
    assert abs(loss - float(g["loss_rel_loss"])) < 2e-6
    z = torch.from_numpy(g["logits_0"]).double().requires_grad_(True)
    crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(w).double() if w is not None else None)
    crit(z, torch.from_numpy(g["labels"])).backward()
    assert np.abs(grad - z.grad.numpy()).max() < 1e-12
    if w is not None:   # the BETA_LOSS weights themselves (roi_relation_predictors.py:4058-4066)
        from oracle import veto_oracle as vo
        counts = np.loadtxt(os.path.join(GOLDEN_DIR, "pred_counts.txt"))
        assert np.allclose(vo.class_balanced_weights(counts), w, rtol=1e-6)


@pytest.mark.parametrize("name", CASES_MEET)
def test_oracle_meet_sampling_and_group_losses_match_reference(name):
    g = _load(name)
    sizes = [int(x) for x in g["group_sizes"]]
    incre = [int(x) for x in g["incre_idx_list"]]
    assert meet_tables.incre_idx_list(sizes) == incre
    srm = meet_tables.sample_rate_matrix(str(g["dataset"]), sizes)
    assert np.array_equal(np.array(srm), g["sample_rate_matrix"])
    words, st = _words_after_seed(1, 4 * len(g["labels"]) + 64)
    stream = to.PyRandomStream(words)
    chosen = to.meet_sampling(g["labels"], incre, srm, len(sizes), stream)
    for k in range(len(sizes)):
        assert np.array_equal(np.array(chosen[k], dtype=np.int64), g["chosen_%d" % k]), k
    # the stream position is exactly where Python's generator stood afterwards: the next random() agrees
    assert stream.random() == float(g["random_after"][0])
    for k in range(len(sizes)):
        lab = to.meet_group_labels(g["labels"], chosen[k], incre, k)
        loss, _ = to.weighted_ce(g["logits_%d" % k][chosen[k]], lab)
        assert abs(loss - float(g["loss_group_%d_CE_loss" % k])) < 2e-6, k


# ----------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES_VANILLA)
def test_hip_ce_loss_matches_reference_and_autograd(name):
    from veto_amd.losses import relation_ce_loss
    g = _load(name)
    dev = torch.device("cuda:0")
    w = torch.from_numpy(g["class_weights"]).to(dev) if int(g["beta_loss"]) else None
    logits = torch.from_numpy(g["logits_0"]).to(dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    loss, grad = relation_ce_loss(logits, labels, weight=w, want_grad=True)
    assert abs(float(loss) - float(g["loss_rel_loss"])) < 5e-6
    _, ref_grad = to.weighted_ce(g["logits_0"], g["labels"], g["class_weights"] if w is not None else None)
    assert np.abs(grad.cpu().numpy() - ref_grad).max() < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES_MEET)
def test_hip_meet_sampling_and_group_losses_match_reference(name):
    from veto_amd.losses import MeetTrainingSampler, relation_ce_loss
    g = _load(name)
    dev = torch.device("cuda:0")
    sizes = [int(x) for x in g["group_sizes"]]
    sampler = MeetTrainingSampler(str(g["dataset"]), sizes, device=dev)
    labels = torch.from_numpy(g["labels"]).to(dev)
    random.seed(1)
    chosen, group_labels = sampler.sample(labels)
    assert random.random() == float(g["random_after"][0])      # Python's generator was advanced by exactly what was used
    incre = [int(x) for x in g["incre_idx_list"]]
    for k in range(len(sizes)):
        assert np.array_equal(chosen[k].cpu().numpy(), g["chosen_%d" % k]), k
        assert np.array_equal(group_labels[k].cpu().numpy(), to.meet_group_labels(g["labels"], g["chosen_%d" % k], incre, k))
        loss, _ = relation_ce_loss(torch.from_numpy(g["logits_%d" % k]).to(dev), group_labels[k], rows=chosen[k])
        assert abs(float(loss) - float(g["loss_group_%d_CE_loss" % k])) < 5e-6, k


def _train_setup(name, meet, dev, forward_only=True):
    from veto_amd import synth, testing
    g = _load(name)
    dataset = str(g["dataset"])
    n_obj_cls = 151 if dataset == "VG" else 201
    num_objs = [int(x) for x in g["num_objs"]]
    layers, heads = int(g.get("layers", 2)), int(g.get("heads", 8))
    cfg = testing.make_config(layers, heads, str(g["mode"]), meet, dataset)
    cfg.VETO_AMD.TRAIN_FORWARD_ONLY = forward_only
    if int(g["beta_loss"]):
        cfg.GLOBAL_SETTING.BETA_LOSS = True
        cfg.GLOBAL_SETTING.REL_COUNTS = np.loadtxt(os.path.join(GOLDEN_DIR, "pred_counts.txt")).tolist()
    experts = bool(int(g.get("experts", 0)))
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = experts
    if meet:
        sd = synth.meet_state_dict(0, [int(x) for x in g["group_sizes"]], layers=layers, num_obj_cls=n_obj_cls, experts=3 if experts else 0)
    else:
        sd = synth.predictor_state_dict(0, layers=layers, num_obj_cls=n_obj_cls, num_rel_cls=51 if dataset == "VG" else 101)
    if int(g["beta_loss"]):
        sd.pop("criterion_loss_rel.weight")       # the synthetic checkpoint stores all-ones class weights
    model = testing.make_predictor(cfg, sd, dev)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    batch = synth.synthetic_batch(7, len(num_objs), num_objs, num_obj_cls=n_obj_cls)
    return g, model, batch, num_objs


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES_VANILLA + CASES_MEET + CASES_EXPERTS)
def test_training_mode_forward_reproduces_reference_losses(name):
    """The predictor in .train() (forward + losses only, VETO_AMD.TRAIN_FORWARD_ONLY, dropout off): BatchNorm on batch
    statistics, relation loss / MEET group losses as the reference returned them for the same weights, inputs, labels
    and Python random seed; the BatchNorm running statistics move the way nn.BatchNorm1d(momentum=0.001) moves them."""
    from veto_amd import testing
    from veto_amd.pairs import prepare_test_pairs
    dev = torch.device("cuda:0")
    meet = name in CASES_MEET + CASES_EXPERTS
    g, model, batch, num_objs = _train_setup(name, meet, dev)
    props = testing.make_proposals(batch, str(g["mode"]), dev)
    pairs = prepare_test_pairs(dev, props)
    rel_labels = list(torch.from_numpy(g["labels"]).to(dev).split([int(p.shape[0]) for p in pairs]))
    bn = (model.model if meet else model).pos_embed[0]
    rm0, rv0, nb0 = bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)
    random.seed(1)
    out = model(props, pairs, rel_labels, None, roi_features=torch.from_numpy(batch["roi_features"]).to(dev),
                roi_depth_features=torch.from_numpy(batch["roi_depth_features"]).to(dev))
    assert out[0] is None and out[1] is None
    for key, val in out[2].items():
        ref = float(g["loss_" + key])
        assert abs(float(val) - ref) < 2e-4 * max(1.0, abs(ref)), (key, float(val), ref)
    assert set(out[2]) == {k[5:] for k in g if k.startswith("loss_")}
    if meet:
        assert random.random() == float(g["random_after"][0])
        for k in range(len(g["group_sizes"])):
            assert np.array_equal(out[4][0][k].cpu().numpy(), g["chosen_%d" % k])
    # running statistics: (1 - m) * old + m * batch statistic (unbiased variance), m = 0.001
    boxes = torch.from_numpy(batch["boxes"]).double()
    wh = boxes[:, 2:] - boxes[:, :2] + 1
    feat = torch.cat([boxes[:, :2] + 0.5 * wh, wh], 1)
    assert torch.allclose(bn.running_mean.cpu().double(), 0.999 * rm0.cpu().double() + 0.001 * feat.mean(0), rtol=1e-5)
    assert torch.allclose(bn.running_var.cpu().double(), 0.999 * rv0.cpu().double() + 0.001 * feat.var(0, unbiased=True), rtol=1e-5)
    assert int(bn.num_batches_tracked) == nb0 + 1


def _small_train_call(dev, layers=2, seed=3):
    from veto_amd import synth, testing
    from veto_amd.pairs import prepare_test_pairs
    cfg = testing.make_config(layers, 8)
    model = testing.make_predictor(cfg, synth.predictor_state_dict(seed, layers=layers), dev).train()
    batch = synth.synthetic_batch(11, 2, [6, 5])
    props = testing.make_proposals(batch, "predcls", dev)
    pairs = prepare_test_pairs(dev, props)
    n = sum(int(p.shape[0]) for p in pairs)
    labels = torch.from_numpy(synth.integers(5, "fd.labels", (n,), 0, 51)).to(dev)
    rel_labels = list(labels.split([int(p.shape[0]) for p in pairs]))
    kw = dict(roi_features=torch.from_numpy(batch["roi_features"]).to(dev), roi_depth_features=torch.from_numpy(batch["roi_depth_features"]).to(dev))
    return model, (props, pairs, rel_labels, None), kw


@pytest.mark.gpu
def test_training_dropout_masks_are_seeded_and_shared_by_forward_and_backward():
    """With the reference's dropout rates (0.1 / 0.35 / 0.35) the loss is a deterministic function of torch's seed, differs
    between seeds, and the backward uses the forward's masks: a central finite difference of the loss along a random
    direction in parameter space (same seed on both sides) matches <grad, direction>."""
    dev = torch.device("cuda:0")
    model, args, kw = _small_train_call(dev)
    drops = {n: m.p for n, m in model.named_modules() if isinstance(m, torch.nn.Dropout)}
    assert drops["pos_embed.3"] == 0.1 and drops["fusion_transformer.transformer.pos_drop"] == 0.35

    def loss_at(seed):
        torch.manual_seed(seed)
        return model(*args, **kw)[2]["rel_loss"]

    l1, l1b, l2 = loss_at(7), loss_at(7), loss_at(8)
    assert float(l1.detach()) == float(l1b.detach()) and abs(float(l1.detach()) - float(l2.detach())) > 1e-4
    model.eval()
    with torch.no_grad():
        ev = model(args[0], args[1], None, None, **kw)[1]
    model.train()
    # finite differences along a random direction over a few parameter tensors
    names = ["rel_out.bias", "fusion_transformer.transformer.layers.0.1.fn.net.0.bias", "fusion_transformer.transformer.pos_embedding",
             "fusion_transformer.transformer.layers.1.0.fn.to_out.0.bias", "location_projection.0.bias", "pos_embed.1.bias"]
    params = dict(model.named_parameters())
    for p in params.values():
        p.grad = None
    loss_at(7).backward()
    gen = torch.Generator().manual_seed(0)
    dirs = {n: torch.randn(params[n].shape, generator=gen).to(dev) for n in names}
    analytic = sum(float((params[n].grad * dirs[n]).sum()) for n in names)
    eps = 2e-2
    with torch.no_grad():
        for n in names:
            params[n].add_(dirs[n], alpha=eps)
        lp = float(loss_at(7).detach())
        for n in names:
            params[n].add_(dirs[n], alpha=-2 * eps)
        lm = float(loss_at(7).detach())
        for n in names:
            params[n].add_(dirs[n], alpha=eps)
    numeric = (lp - lm) / (2 * eps)
    assert abs(numeric - analytic) < 2e-2 * max(1.0, abs(analytic)), (numeric, analytic)
    assert len(ev) == 2      # eval still runs after training calls (weights re-uploaded, inference path untouched)


@pytest.mark.gpu
def test_training_steps_reduce_the_loss():
    """A few SGD steps on one batch through the HIP forward / backward (dropout off so that the loss is comparable)."""
    dev = torch.device("cuda:0")
    model, args, kw = _small_train_call(dev, layers=1)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = model(*args, **kw)[2]["rel_loss"]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] - 0.3 and all(b < a for a, b in zip(losses, losses[1:])), losses


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES_VANILLA + CASES_MEET + CASES_EXPERTS)
def test_training_backward_matches_reference_gradients(name):
    """loss.backward() through the HIP training path (veto_forward_train / veto_backward / veto_ce_loss) against the
    gradients the reference's autograd produced for the same weights, inputs, labels and random seed: every parameter's
    gradient norm and a strided sample of its entries."""
    from veto_amd import testing
    from veto_amd.pairs import prepare_test_pairs
    dev = torch.device("cuda:0")
    meet = name in CASES_MEET + CASES_EXPERTS
    g, model, batch, num_objs = _train_setup(name, meet, dev, forward_only=False)
    props = testing.make_proposals(batch, str(g["mode"]), dev)
    pairs = prepare_test_pairs(dev, props)
    rel_labels = list(torch.from_numpy(g["labels"]).to(dev).split([int(p.shape[0]) for p in pairs]))
    random.seed(1)
    roi_in = {"roi_features": torch.from_numpy(batch["roi_features"]).to(dev).requires_grad_(True),
              "roi_depth_features": torch.from_numpy(batch["roi_depth_features"]).to(dev).requires_grad_(True)}
    out = model(props, pairs, rel_labels, None, roi_features=roi_in["roi_features"], roi_depth_features=roi_in["roi_depth_features"])
    for key, val in out[2].items():
        ref = float(g["loss_" + key])
        assert abs(float(val.detach()) - ref) < 2e-4 * max(1.0, abs(ref)), (key, float(val.detach()), ref)
    sum(out[2].values()).backward()
    torch.cuda.synchronize()
    # input gradients: the ROI maps are leaves of the reference's graph (its depth backbone trains through them)
    for iname, t in roi_in.items():
        assert t.grad is not None and t.grad.shape == t.shape, iname
        got = t.grad.detach().reshape(-1).cpu().numpy().astype(np.float64)
        ref_norm, step = float(g["inputgradnorm_" + iname]), int(g["inputgradstep_" + iname])
        ref_s = g["inputgradsample_" + iname].astype(np.float64)
        scale = max(np.abs(ref_s).max(), ref_norm / np.sqrt(got.size), 1e-12)
        err = np.abs(got[::step] - ref_s).max() / scale
        nerr = abs(np.linalg.norm(got) - ref_norm) / max(ref_norm, 1e-12)
        assert err < 2e-3 and nerr < 2e-3, (iname, err, nerr, ref_norm)
    params = dict(model.named_parameters(remove_duplicate=False))     # EXPERT_GROUP: rel_out aliases the last expert's heads
    names = [k[9:] for k in g if k.startswith("gradnorm_")]
    assert names
    worst = 0.0
    for pname in names:
        prm = params[pname]
        assert prm.grad is not None, pname
        got = prm.grad.detach().reshape(-1).cpu().numpy().astype(np.float64)
        ref_norm = float(g["gradnorm_" + pname])
        step = int(g["gradstep_" + pname])
        ref_s = g["gradsample_" + pname].astype(np.float64)
        scale = max(np.abs(ref_s).max(), ref_norm / np.sqrt(got.size), 1e-8)
        err = np.abs(got[::step] - ref_s).max() / scale
        nerr = abs(np.linalg.norm(got) - ref_norm) / max(ref_norm, 1e-8)
        worst = max(worst, err, nerr)
        assert err < 2e-3 and nerr < 2e-3, (pname, err, nerr, ref_norm)
    # parameters the loss does not depend on stay without gradient (obj_embed2 in predcls, BatchNorm statistics)
    used = {id(params[n]) for n in names}
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in params.values() if id(p) not in used)
    print("%s: worst relative gradient error %.2e over %d parameters" % (name, worst, len(names)))


@pytest.mark.gpu
def test_relation_head_training_branch_end_to_end():
    """ROIRelationHead.forward in training mode on the device: GT-box relation sampling (budget 1024 / 25 % foreground),
    ROI pooling from FPN + depth maps, the predictor's loss with an autograd graph whose backward fills the predictor's
    parameter gradients AND, through the ROI maps and the differentiable ROI pooling, reaches the feature maps: the reference
    trains its depth backbone this way (tools/relation_train_net.py:166-170)."""
    from veto_amd import synth, testing
    from veto_amd.relation_head import VETORelationHead
    from veto_amd.structures import BoxList
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(21)
    W, H = 512, 384
    feats = [torch.from_numpy((0.5 * rng.randn(2, 256, H >> (2 + l), W >> (2 + l))).astype(np.float32)).to(dev) for l in range(4)]
    depth = torch.from_numpy((0.5 * rng.randn(2, 256, H >> 4, W >> 4)).astype(np.float32)).to(dev).requires_grad_(True)
    feats[1].requires_grad_(True)
    cfg = testing.make_config(2, 8)
    from relation_sampling import make_roi_relation_samp_processor    # tests/: a stand-in for the host code base's sampler
    head = VETORelationHead(cfg, samp_processor=make_roi_relation_samp_processor(cfg))
    head.predictor = testing.make_predictor(cfg, synth.predictor_state_dict(3, layers=2), dev)
    head.train()
    props, targets = [], []
    for i, (boxes, rel) in enumerate(synth.synthetic_relation_targets(num_objs=(7, 5))):
        b = torch.from_numpy(boxes)
        b[:, 2:] = b[:, :2] + b[:, 2:].abs() * 0.5 + 8
        labels = torch.from_numpy(synth.integers(3, "head.labels.%d" % i, (len(boxes),), 1, 151))
        p = BoxList(b, (W, H)).to(dev)
        p.add_field("labels", labels.to(dev))
        t = BoxList(b.clone(), (W, H)).to(dev)
        t.add_field("relation", torch.from_numpy(rel).to(dev))
        t.add_field("labels", labels.to(dev))
        props.append(p)
        targets.append(t)
    torch.manual_seed(5)
    roi, out_props, losses = head(feats, props, targets=targets, depth_features=depth, logger=None, x=None)
    assert set(losses) == {"rel_loss"} and losses["rel_loss"].requires_grad and roi.shape == (12, 256, 8, 8)
    assert all(p.has_field("locating_match") if hasattr(p, "has_field") else "locating_match" in p.extra_fields for p in out_props)
    losses["rel_loss"].backward()
    grads = [p.grad for p in head.predictor.parameters() if p.grad is not None]
    assert len(grads) >= 30 and all(torch.isfinite(g).all() for g in grads)
    assert float(head.predictor.rel_out.weight.grad.abs().max()) > 0
    # the chain veto_backward -> d roi maps -> veto_roi_pool_backward -> feature maps is connected
    assert depth.grad is not None and torch.isfinite(depth.grad).all() and float(depth.grad.abs().max()) > 0
    assert feats[1].grad is not None and torch.isfinite(feats[1].grad).all()


@pytest.mark.gpu
def test_ce_loss_edge_cases_follow_torch():
    """No rows (a MEET tail group that received no sampled relation): NaN like the reference's CE over nothing, empty gradient,
    no launch.  NEGATIVE labels -- nn.CrossEntropyLoss's ignore_index -100 -- are ignored rows: same loss and gradients as torch's
    own criterion.  A label >= C (a class-mapping bug: torch raises) poisons the loss with NaN instead of dropping the row."""
    from veto_amd.losses import ce_loss, relation_ce_loss
    dev = torch.device("cuda:0")
    loss, grad = relation_ce_loss(torch.zeros((0, 7), device=dev), torch.zeros(0, dtype=torch.int64, device=dev), want_grad=True)
    assert torch.isnan(loss).all() and grad.shape == (0, 7)
    rows = torch.zeros(0, dtype=torch.int64, device=dev)
    loss, grad = relation_ce_loss(torch.randn(5, 7, device=dev), torch.zeros(0, dtype=torch.int64, device=dev), rows=rows, want_grad=True)
    assert torch.isnan(loss).all() and grad.shape == (0, 7)
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(40, 11, generator=g)
    labels = torch.randint(0, 11, (40,), generator=g)
    labels[[3, 17, 18]] = -100
    w = torch.rand(11, generator=g) + 0.1
    ref_in = logits.clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss(weight=w)(ref_in, labels)
    ref.backward()
    got_in = logits.to(dev).requires_grad_(True)
    got = ce_loss(got_in, labels.to(dev), weight=w.to(dev))
    got.backward()
    assert abs(float(got.detach()) - float(ref.detach())) < 1e-6
    assert (got_in.grad.cpu() - ref_in.grad).abs().max() < 1e-7
    assert float(got_in.grad[[3, 17, 18]].abs().max()) == 0.0
    bad = labels.clone()
    bad[5] = 11
    assert torch.isnan(ce_loss(logits.to(dev), bad.to(dev), weight=w.to(dev)).detach()).all()
    # ... and its gradient row too (and, through the NaN weight sum, nothing else can be trusted either): no optimizer step on it
    loss, grad = relation_ce_loss(logits.to(dev), bad.to(dev), weight=w.to(dev), want_grad=True)
    assert torch.isnan(loss).all() and torch.isnan(grad[5]).all()


@pytest.mark.gpu
def test_training_workspace_is_released_without_a_backward_and_weights_refresh():
    """(1) A training-mode forward that never gets a backward (torch.no_grad(), a validation pass) must not pin the cached
    36 GB-class workspace: the next step reuses it.  (2) A write through `.data` is invisible to the version stamp in eval mode:
    refresh_weights() makes it take effect; load_state_dict and training mode refresh by themselves."""
    from veto_amd import testing
    from veto_amd.pairs import prepare_test_pairs
    dev = torch.device("cuda:0")
    g, model, batch, num_objs = _train_setup("train_vanilla", False, dev, forward_only=False)
    props = testing.make_proposals(batch, "predcls", dev)
    pairs = prepare_test_pairs(dev, props)
    rel_labels = list(torch.from_numpy(g["labels"]).to(dev).split([int(p.shape[0]) for p in pairs]))
    rgb, dep = torch.from_numpy(batch["roi_features"]).to(dev), torch.from_numpy(batch["roi_depth_features"]).to(dev)
    with torch.no_grad():
        model(props, pairs, rel_labels, None, roi_features=rgb, roi_depth_features=dep)
    ws0 = model.__dict__["_train_ws"]
    out = model(props, pairs, rel_labels, None, roi_features=rgb, roi_depth_features=dep)     # would allocate a second one if pinned
    assert model.__dict__["_train_ws"] is ws0
    holder = model.__dict__["_train_ws_owner"]()
    assert holder is not None and not holder.done
    sum(out[2].values()).backward()
    assert holder.done
    # (2)
    model.eval()
    with torch.no_grad():
        a = torch.cat(list(model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)[1]))
        model.rel_out.bias.data.add_(1.0)                      # does not bump Tensor._version
        b = torch.cat(list(model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)[1]))
        assert torch.equal(a, b)                               # the documented blind spot ...
        model.refresh_weights()
        c = torch.cat(list(model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)[1]))
        assert (c - a - 1.0).abs().max() < 1e-5                # ... and its remedy
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        sd["rel_out.bias"] -= 1.0
        model.load_state_dict(sd)
        d = torch.cat(list(model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)[1]))
        assert (d - a).abs().max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"VETO_TRAIN_LN_SPLIT": "1"}, {"VETO_TRAIN_GELU_EPI": "0", "VETO_TRAIN_QKV_F24": "0"}, {"VETO_TRAIN_RECOMPUTE": "1"}],
                         ids=["ln-backward-emits-split-rows", "round5-forms", "recompute-instead-of-keeping"])
def test_training_variants_behind_the_knobs_match_reference_gradients(env):
    """The forms of the training path that a knob selects (read once per process, hence a child process): the LayerNorm backward that emits the
    split rows of the Linear behind it (round 6: measured slower, off by default), and round 5's forms of what round 6 changed (gelu' as a pass
    of its own, fp32 q / k / v), and the form that recomputes the LayerNorm / GELU rows in the backward instead of keeping them.  Same gradient and
    finite-difference tests as the default path."""
    import subprocess
    import sys
    picked = "(test_training_backward_matches_reference_gradients or test_training_dropout_masks_are_seeded) and (train_vanilla- or train_meet_vg or seeded)"
    subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", picked, "-p", "no:cacheprovider"],
                   env=dict(os.environ, **env), check=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
