import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")
VG_MEET_GROUPS = [4, 6, 9, 19, 12]
GQA_MEET_GROUPS = [5, 10, 20, 65]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """Predictor fixtures (the post_* files are PostProcessor fixtures, see post_golden_names)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and not f.startswith(("post", "sggeval", "train_", "relsample", "roialign_")))


def post_golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("post_"))


def postmeet_golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("postmeet_"))


def postvote_golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("postvote_"))


def load_postvote_golden(name):
    """Returns (fixture, {group_<k><e>: logits}, obj_logits, pairs, n) with the inputs regenerated."""
    from oracle import veto_oracle as vo
    from veto_amd import synth
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    n = int(g["n"])
    sizes = [int(x) for x in g["group_sizes"]]
    n_objc = 151 if str(g["dataset"]) == "VG" else 201
    P = n * (n - 1)
    rel = {}
    for k, gk in enumerate(sizes):
        base = synth.normal(29, "vote.base_%d" % k, (P, gk + 2), 0.0, 1.5)
        for e in range(3):
            own = synth.normal(29, "vote.group_%d%d" % (k, e + 1), (P, gk + 2), 0.0, 1.0)
            rel["group_%d%d" % (k, e + 1)] = (base + own).astype(np.float32)
    obj_logits = synth.normal(29, "vote.obj_logits", (n, n_objc), 0.0, 3.0)
    return g, rel, obj_logits, vo.enumerate_test_pairs(n), n


def load_postmeet_golden(name):
    """Returns (fixture, {group_k: logits}, obj_logits, pairs, n) with the inputs regenerated."""
    from oracle import veto_oracle as vo
    from veto_amd import synth
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    n = int(g["n"])
    sizes = [int(x) for x in g["group_sizes"]]
    n_objc = 151 if str(g["dataset"]) == "VG" else 201
    P = n * (n - 1)
    rel = {"group_%d" % k: synth.normal(23, "meet.group_%d" % k, (P, gk + 2), 0.0, 2.0) for k, gk in enumerate(sizes)}
    obj_logits = synth.normal(23, "meet.obj_logits", (n, n_objc), 0.0, 3.0)
    return g, rel, obj_logits, vo.enumerate_test_pairs(n), n


def load_post_golden(name):
    """Returns (fixture, rel_logits, obj_logits, pair list, num_objs) with the inputs regenerated."""
    import torch
    from oracle import veto_oracle as vo
    from veto_amd import synth
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    num_objs = [int(x) for x in g["num_objs"]]
    pairs = [vo.enumerate_test_pairs(n) for n in num_objs]
    n_obj, n_pair = sum(num_objs), sum(len(p) for p in pairs)
    rel_logits = synth.normal(21, "post.rel_logits", (n_pair, 51), 0.0, 2.0)
    if int(g["onehot"]):
        lab = synth.integers(21, "post.labels", (n_obj,), 1, 151)
        obj_logits = np.full((n_obj, 151), -1000.0, dtype=np.float32)
        obj_logits[np.arange(n_obj), lab] = 1000.0
    else:
        obj_logits = synth.normal(21, "post.obj_logits", (n_obj, 151), 0.0, 3.0)
    return g, rel_logits, obj_logits, pairs, num_objs


def load_golden(name):
    """Returns (fixture dict, regenerated state dict, regenerated input batch)."""
    from veto_amd import synth
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    layers, heads = int(g["layers"]), int(g["heads"])
    dataset = str(g["dataset"])
    n_obj, n_rel = (151, 51) if dataset == "VG" else (201, 101)
    num_objs = [int(x) for x in g["num_objs"]]
    if int(g["meet"]):
        groups = [int(x) for x in g["group_sizes"]]
        sd = synth.meet_state_dict(0, groups, layers=layers, num_obj_cls=n_obj,
                                   experts=3 if int(g.get("experts", 0)) else 0)
    else:
        sd = synth.predictor_state_dict(0, layers=layers, num_obj_cls=n_obj, num_rel_cls=n_rel)
    batch = synth.synthetic_batch(7, len(num_objs), num_objs, num_obj_cls=n_obj)
    g["_layers"], g["_heads"], g["_n_obj"], g["_n_rel"] = layers, heads, n_obj, n_rel
    return g, sd, batch


def golden_pairs(g):
    """The fixture's pair lists, one [P_i, 2] int64 array per image (the capped fixtures carry the reference's own
    top-2048 selection, whose tie order a re-implementation need not reproduce: feed these to the predictor)."""
    counts = [int(x) for x in g["pair_counts"]]
    return np.split(g["pair_idx"], np.cumsum(counts)[:-1])


def subset_images(g, batch, idxs):
    """(batch, pair lists, row index into the fixture's logits) restricted to the images `idxs`: bounds the CPU time of the
    oracle checks on the full-size fixtures (images are independent, eval-mode BatchNorm uses running statistics)."""
    num_objs = [int(x) for x in batch["num_objs"]]
    o0 = np.concatenate([[0], np.cumsum(num_objs)])
    pairs = golden_pairs(g)
    p0 = np.concatenate([[0], np.cumsum([len(p) for p in pairs])])
    sub = {"num_objs": [num_objs[i] for i in idxs], "image_size": batch["image_size"]}
    for k, v in batch.items():
        if isinstance(v, np.ndarray) and v.shape[:1] == (o0[-1],):
            sub[k] = np.concatenate([v[o0[i]:o0[i + 1]] for i in idxs])
    rows = np.concatenate([np.arange(p0[i], p0[i + 1]) for i in idxs])
    return sub, [pairs[i] for i in idxs], rows


# images of the full-size fixtures the CPU oracle check walks (the GPU parity test takes all of them)
ORACLE_IMAGE_SUBSET = {"sgcls_b12_n36_l4h8": [0, 11], "predcls_b12_n36_l6h6": [5], "ragged12_capped_l4h8": [0, 1, 2, 4, 7, 11]}


@pytest.fixture(scope="session")
def repo_root():
    return REPO
