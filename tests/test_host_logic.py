"""Host-side logic that needs no GPU: registry protocol, config slice, BoxList, MEET tables,
state-dict key compatibility, error behaviour, synthetic-data determinism."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from veto_amd import meet_tables, predictor, registry, synth, testing
from veto_amd.config import default_config
from veto_amd.structures import BoxList


def test_registry_protocol_matches_reference():
    r = registry.Registry()

    @r.register("a")
    def a():
        return 1

    r.register("b", a)
    assert r["a"] is a and r["b"] is a
    with pytest.raises(AssertionError):       # utils/registry.py:4-6
        r.register("a", a)
    assert set(registry.ROI_RELATION_PREDICTOR) >= {"VETOPredictor", "VETOPredictor_MEET"}


def test_install_overwrites_reference_entries():
    target = registry.Registry({"VETOPredictor": object(), "VETOPredictor_MEET": object(), "MotifPredictor": 7})
    registry.install(target)
    assert target["VETOPredictor"] is predictor.VETOPredictor
    assert target["VETOPredictor_MEET"] is predictor.VETOPredictor_MEET
    assert target["MotifPredictor"] == 7
    cfg = testing.make_config(2, 8)
    predictor.set_statistics_provider(None)
    predictor.set_embedding_provider(lambda names, d, k: torch.zeros(len(names), k))
    m = registry.make_roi_relation_predictor(cfg, 512)
    assert isinstance(m, predictor.VETOPredictor) and m.mode == "predcls"


def test_state_dict_keys_equal_the_reference_names():
    # synth.*_state_dict key lists were checked against the real reference's load_state_dict
    # (tests/golden/make_golden.py: no unexpected keys, only criterion buffers missing)
    predictor.set_embedding_provider(lambda names, d, k: torch.zeros(len(names), k))
    m = predictor.VETOPredictor(testing.make_config(6, 6), 512)
    assert set(m.state_dict()) == set(synth.predictor_state_dict(0, layers=6))
    mm = predictor.VETOPredictor_MEET(testing.make_config(6, 6, meet=True), 512)
    assert set(mm.state_dict()) == set(synth.meet_state_dict(0, [4, 6, 9, 19, 12], layers=6))
    assert mm.max_group_element_number_list == [4, 6, 9, 19, 12]
    g, _, _ = load_golden("meet_n10_l6h6")
    assert mm.incre_idx_list == [int(x) for x in g["incre_idx_list"]]
    gq = predictor.VETOPredictor_MEET(testing.make_config(4, 8, meet=True, dataset="GQA"), 512)
    g, _, _ = load_golden("meet_gqa_n6_l4h8")
    assert gq.incre_idx_list == [int(x) for x in g["incre_idx_list"]]
    assert [h.out_features for h in gq.model.rel_out] == [int(x) + 2 for x in g["group_sizes"]]


@pytest.mark.parametrize("name", ["predcls_n36_l6h6", "predcls_n36_l4h8", "sgcls_n10_l6h6", "meet_n36_l6h6", "meet_gqa_n36_l4h8", "meetx_n10_l4h8"])
def test_state_dict_keys_equal_the_reference_modules_own_key_list(name):
    """The checkpoint contract, pinned directly: every predictor fixture stores sorted(model.state_dict()) of the REAL
    reference module it was generated from; the mirror module must have exactly those keys."""
    g, _, _ = load_golden(name)
    predictor.set_embedding_provider(lambda names, d, k: torch.zeros(len(names), k))
    meet = bool(int(g["meet"]))
    cfg = testing.make_config(g["_layers"], g["_heads"], str(g["mode"]), meet, str(g["dataset"]))
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = bool(int(g["experts"]))
    m = (predictor.VETOPredictor_MEET if meet else predictor.VETOPredictor)(cfg, 512)
    ref_keys = [str(k) for k in g["state_dict_keys"]]
    assert sorted(m.state_dict().keys()) == ref_keys, sorted(set(m.state_dict()) ^ set(ref_keys))


def test_every_predictor_fixture_carries_the_round2_keys():
    from conftest import golden_names
    for name in golden_names():
        g, _, _ = load_golden(name)
        for k in ("experts", "capped_pairs", "pair_counts", "state_dict_keys"):
            assert k in g, (name, k)
        assert int(g["pair_counts"].sum()) == g["pair_idx"].shape[0]


def test_modes_and_constructor_errors():
    cfg = default_config()
    cfg.MODEL.ROI_RELATION_HEAD.USE_GT_OBJECT_LABEL = False
    predictor.set_embedding_provider(lambda names, d, k: torch.zeros(len(names), k))
    assert predictor.VETOPredictor(cfg, 512).mode == "sgcls"
    cfg.MODEL.ROI_RELATION_HEAD.USE_GT_BOX = False
    assert predictor.VETOPredictor(cfg, 512).mode == "sgdet"
    cfg.MODEL.ROI_RELATION_HEAD.VETOTRANSFORMER.T_INPUT_DIM = 512
    with pytest.raises(ValueError):
        predictor.VETOPredictor(cfg, 512)
    cfg = testing.make_config(2, 8, meet=True)
    cfg.ENSEMBLE_LEARNING.EXPERT_GROUP = True      # 3 experts x 5 groups, `rel_out` aliases the last expert
    m = predictor.VETOPredictor_MEET(cfg, 512)
    want = set(synth.meet_state_dict(0, m.max_group_element_number_list, layers=2, experts=3))
    got = set(m.state_dict())
    assert got == want and m._num_out == 3 * sum(g + 2 for g in m.max_group_element_number_list)
    assert m.model.rel_out[0].weight is m.model.rel_out_group[2][0].weight
    cfg = testing.make_config(2, 8)
    cfg.VETO_AMD.PRECISION = "int4"
    with pytest.raises(ValueError):
        predictor.VETOPredictor(cfg, 512)


def test_no_cpu_path_and_no_training_path():
    predictor.set_embedding_provider(lambda names, d, k: torch.zeros(len(names), k))
    m = predictor.VETOPredictor(testing.make_config(1, 8), 512).eval()
    batch = synth.synthetic_batch(7, 1, 3)
    props = testing.make_proposals(batch, "predcls", "cpu")
    pairs = [torch.tensor([[0, 1], [1, 0]])]
    with pytest.raises(RuntimeError, match="HIP device"):
        m(props, pairs, None, None, roi_features=torch.from_numpy(batch["roi_features"]),
          roi_depth_features=torch.from_numpy(batch["roi_depth_features"]))
    with pytest.raises(RuntimeError, match="HIP device"):      # training mode has no CPU path either
        m.train()(props, pairs, [torch.tensor([1, 0])], None, roi_features=torch.from_numpy(batch["roi_features"]),
                  roi_depth_features=torch.from_numpy(batch["roi_depth_features"]))


def test_beta_loss_weights_from_reference_counts():
    import os
    from conftest import GOLDEN_DIR
    counts = np.loadtxt(os.path.join(GOLDEN_DIR, "pred_counts.txt"))
    w = predictor.class_balanced_weights(counts, 51)
    assert abs(float(w.sum()) - 51) < 1e-3 and abs(float(w[0]) - 0.14453256) < 1e-6 and abs(float(w[50]) - 28.964382) < 1e-3
    cfg = testing.make_config(1, 8)
    cfg.GLOBAL_SETTING.BETA_LOSS = True
    with pytest.raises(ValueError):
        predictor.VETOPredictor(cfg, 512)
    cfg.GLOBAL_SETTING.REL_COUNTS = counts.tolist()
    m = predictor.VETOPredictor(cfg, 512)
    assert torch.allclose(m.criterion_loss_rel.weight, w)


def test_boxlist_convert_uses_the_plus_one_convention():
    b = BoxList(torch.tensor([[10.0, 20.0, 29.0, 59.0]]), (800, 600), "xyxy")
    b.add_field("labels", torch.tensor([3]))
    w = b.convert("xywh")
    assert w.bbox.tolist() == [[10.0, 20.0, 20.0, 40.0]] and w.get_field("labels").item() == 3
    assert w.convert("xyxy").bbox.tolist() == b.bbox.tolist() and len(b) == 1
    with pytest.raises(ValueError):
        b.convert("cxcywh")


def test_meet_tables():
    for (ds, split), sizes in meet_tables.GROUP_SIZES.items():
        assert sum(sizes) == meet_tables.NUM_CLASSES[ds][1] - 1
        idx = meet_tables.incre_idx_list(sizes)
        assert len(idx) == meet_tables.NUM_CLASSES[ds][1] and idx[0] == 0 and idx[-1] == len(sizes)
    with pytest.raises(ValueError):
        meet_tables.group_sizes("VG", "nope")


def test_synthetic_generators_are_bit_stable():
    # the golden fixtures depend on these streams: a changed digest means every golden must be regenerated
    sd = synth.predictor_state_dict(0, layers=1)
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    b = synth.synthetic_batch(7, 1, 4)
    for k in ("boxes", "labels", "roi_features", "predict_logits"):
        h.update(np.ascontiguousarray(b[k]).tobytes())
    assert h.hexdigest() == "5fffec42c9ef2181a42859be88e635aa1c1ff2db4c7890dd2ddc2d1cba19197e"


def test_gtbox_relsample_matches_reference_sampler():
    """The tests' stand-in sampler (tests/relation_sampling.py, used to drive VETORelationHead in training mode) against the
    reference's own function run with the same torch seed (tests/golden/relsample_gtbox.npz): same pairs, labels, order."""
    import os
    from conftest import GOLDEN_DIR
    from relation_sampling import make_roi_relation_samp_processor
    props, targets = [], []
    for boxes, rel in synth.synthetic_relation_targets():
        props.append(BoxList(torch.from_numpy(boxes), (800, 600)))
        t = BoxList(torch.from_numpy(boxes.copy()), (800, 600))
        t.add_field("relation", torch.from_numpy(rel))
        targets.append(t)
    g = np.load(os.path.join(GOLDEN_DIR, "relsample_gtbox.npz"))
    samp = make_roi_relation_samp_processor(testing.make_config(1, 8))
    torch.manual_seed(0)
    out_props, labels, pairs, binaries = samp.gtbox_relsample(props, targets)
    assert out_props is props and all(p.get_field("locating_match").sum() == len(p) for p in props)
    for i in range(len(props)):
        assert np.array_equal(pairs[i].numpy(), g["pairs_%d" % i]) and np.array_equal(labels[i].numpy(), g["labels_%d" % i])
        assert np.array_equal(binaries[i].numpy(), g["binary_%d" % i])
    assert len(pairs[1]) == 1024 and int((labels[1] > 0).sum()) == 256 and len(pairs[3]) == 0     # budget hit; single object


def test_relation_head_forward_has_the_reference_signature():
    """ROIRelationHead.forward(features, proposals, depth_features=None, targets=None, logger=None, x=None)
    (relation_head.py:90), called by the reference as self.relation(features, detections, targets=..., depth_features=...,
    logger=..., x=x) (roi_heads.py:69): same parameter names, order and defaults, and that keyword set binds."""
    import inspect
    from veto_amd.relation_head import VETORelationHead
    sig = inspect.signature(VETORelationHead.forward)
    assert list(sig.parameters) == ["self", "features", "proposals", "depth_features", "targets", "logger", "x"]
    assert all(sig.parameters[k].default is None for k in ("depth_features", "targets", "logger", "x"))
    head = VETORelationHead(testing.make_config(2, 8))
    with pytest.raises(ValueError, match="depth_features"):     # binds, then fails on the missing depth maps (not a TypeError)
        head(features=[torch.zeros(1, 256, 8, 8)], proposals=[], targets=None, depth_features=None, logger=None, x=None)
    head.train()
    with pytest.raises(ValueError, match="sampler"):            # pair sampling belongs to the host code base
        head([torch.zeros(1, 256, 8, 8)], [], torch.zeros(1, 256, 2, 2), targets=[object()])


def test_relation_head_builds_the_host_sampler_like_the_reference(monkeypatch):
    """ROIRelationHead.__init__ builds its sampler from cfg (relation_head.py:66-67); VETORelationHead(cfg, in_channels) does the same
    when the host code base is importable, keeps an explicit one, and stays None (raising only on a training forward) otherwise."""
    import sys
    import types
    from veto_amd import relation_head, testing
    cfg = testing.make_config(1, 8)
    assert relation_head._host_samp_processor(cfg) is None            # no pysgg in this process
    seen = []
    mod = types.ModuleType("pysgg.modeling.roi_heads.relation_head.sampling")
    mod.make_roi_relation_samp_processor = lambda c: seen.append(c) or "host-sampler"
    for name in ("pysgg", "pysgg.modeling", "pysgg.modeling.roi_heads", "pysgg.modeling.roi_heads.relation_head"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    monkeypatch.setitem(sys.modules, mod.__name__, mod)
    assert relation_head._host_samp_processor(cfg) == "host-sampler" and seen == [cfg]


def test_bench_reports_the_floors_of_the_layer_tail():
    """bench.py's `floors` object: the four lower bounds of one layer-tail launch at the headline size, and the
    fraction of the tightest one that the measured time reaches."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = 15120 * 19
    flops = 2.0 * rows * 576 * 576 + 4.0 * rows * 576 * 1152
    f = bench.layer_tail_floors(rows, {"flops_per_launch": flops, "bytes_per_launch": rows * 576 * 16.0}, 7.56e9, 2.25)
    assert abs(f["matrix_ms"] - 0.662) < 0.01 and abs(f["l2_to_lds_ms"] - 0.838) < 0.01 and abs(f["l2_miss_ms"] - 0.945) < 0.01
    assert 0.5 < f["hbm_ms"] < 0.6 and abs(f["frac_of_tightest_floor"] - 0.945 / 2.25) < 0.01
    assert bench.layer_tail_floors(rows, {"flops_per_launch": flops, "bytes_per_launch": 1.0}, None, 2.25)["l2_miss_ms"] is None
