"""Helpers shared by tests/, bench.py and __graft_entry__.py: build a predictor from the portable
synthetic weights and wrap a synthetic batch in BoxLists.  No oracle code lives here."""
import numpy as np
import torch

from . import meet_tables, predictor
from .config import default_config
from .structures import BoxList


def make_config(layers, heads, mode="predcls", meet=False, dataset="VG", precision="mixed", max_chunk_pairs=0):
    cfg = default_config()
    rh = cfg.MODEL.ROI_RELATION_HEAD
    rh.PREDICTOR = "VETOPredictor_MEET" if meet else "VETOPredictor"
    rh.USE_GT_BOX = True
    rh.USE_GT_OBJECT_LABEL = (mode == "predcls")
    rh.VETOTRANSFORMER.ENC_LAYERS = layers
    rh.VETOTRANSFORMER.NHEADS = heads
    cfg.GLOBAL_SETTING.DATASET_CHOICE = dataset
    cfg.VETO_AMD.PRECISION = precision
    cfg.VETO_AMD.MAX_CHUNK_PAIRS = max_chunk_pairs
    return cfg


def make_predictor(cfg, state_dict_np, device="cuda"):
    n_obj, n_rel = meet_tables.NUM_CLASSES[cfg.GLOBAL_SETTING.DATASET_CHOICE]
    predictor.set_statistics_provider(lambda c: {"obj_classes": ["o%d" % i for i in range(n_obj)],
                                                 "rel_classes": ["r%d" % i for i in range(n_rel)]})
    predictor.set_embedding_provider(lambda names, wv_dir, wv_dim: torch.zeros(len(names), wv_dim))
    cls = predictor.VETOPredictor_MEET if cfg.MODEL.ROI_RELATION_HEAD.PREDICTOR == "VETOPredictor_MEET" \
        else predictor.VETOPredictor
    model = cls(cfg, 512)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state_dict_np.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("criterion" in m or "CE_loss" in m for m in missing), missing
    return model.to(device).eval()


def make_proposals(batch, mode, device):
    props, start = [], 0
    for n in batch["num_objs"]:
        sl = slice(start, start + n)
        b = BoxList(torch.from_numpy(batch["boxes"][sl]).to(device), batch["image_size"], mode="xyxy")
        b.add_field("labels", torch.from_numpy(batch["labels"][sl]).to(device))
        if mode != "predcls":
            b.add_field("predict_logits", torch.from_numpy(batch["predict_logits"][sl]).to(device))
            b.add_field("pred_labels", torch.from_numpy(batch["pred_labels"][sl]).to(device))
        props.append(b)
        start += n
    return props
