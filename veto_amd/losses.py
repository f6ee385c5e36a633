"""Training-time losses and MEET expert sampling on the device (partial: the forward
and backward of the transformer itself are not built, so the predictors still refuse training mode).

  relation_ce_loss      nn.CrossEntropyLoss(weight)(logits[rows], labels) and its gradient w.r.t. the logits
                        (roi_relation_predictors.py:4067-4068,4133; per MEET group :3842-3846)        -> veto_ce_loss
  MeetTrainingSampler   the expert sampling of VETOPredictor_MEET.forward (:3940-3969) + the per-group label remap
                        (:3812-3821).  The reference draws from Python's `random` once per relation, in a Python loop
                        with one .item() per relation; here the host hands the device the next raw words of the SAME
                        generator, the device consumes them exactly as randint / random would, and the host generator
                        is advanced by what was used, so a run stays in lock-step with the reference's stream.
"""
import ctypes
import random

import numpy as np
import torch

from . import meet_tables, native


def relation_ce_loss(logits, labels, weight=None, rows=None, want_grad=False):
    """Returns (loss [1] float32 device tensor, grad [n, C] or None)."""
    lib = native.load_library()
    dev = logits.device
    if dev.type != "cuda":
        raise RuntimeError("veto_amd losses run only on a HIP device (got %s)" % dev)
    logits = logits.detach().to(torch.float32).contiguous()
    labels = labels.to(device=dev, dtype=torch.int64).contiguous()
    n, C = int(labels.shape[0]), int(logits.shape[1])
    if rows is not None:
        rows = rows.to(device=dev, dtype=torch.int64).contiguous()
        if rows.shape[0] != n:
            raise ValueError("rows and labels must have the same length")
    elif logits.shape[0] != n:
        raise ValueError("labels must have one entry per logit row")
    if weight is not None:
        weight = weight.to(device=dev, dtype=torch.float32).contiguous()
    if n == 0:
        # no sampled relation for this head (a MEET tail group on a small batch): the reference's CE_loss over zero rows is
        # the mean of nothing, NaN (roi_relation_predictors.py:3842-3846); same value here, an empty gradient, no launch
        return (torch.full((1,), float("nan"), dtype=torch.float32, device=dev),
                torch.zeros((0, C), dtype=torch.float32, device=dev) if want_grad else None)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    grad = torch.empty((n, C), dtype=torch.float32, device=dev) if want_grad else None
    ws = torch.empty(lib.veto_ce_loss_workspace_bytes(n), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    native.check(lib.veto_ce_loss(
        ctypes.c_void_p(stream.cuda_stream), logits.data_ptr(), logits.stride(0), labels.data_ptr(),
        weight.data_ptr() if weight is not None else None, rows.data_ptr() if rows is not None else None, n, C,
        loss.data_ptr(), grad.data_ptr() if grad is not None else None, ws.data_ptr(), ws.numel()))
    for t in (logits, labels, ws):
        t.record_stream(stream)
    return loss, grad


class _CELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight, rows):
        loss, grad = relation_ce_loss(logits, labels, weight=weight, rows=rows, want_grad=True)
        ctx.save_for_backward(grad)
        ctx.rows, ctx.shape = rows, tuple(logits.shape)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        grad = grad * g
        if ctx.rows is not None:     # gradient rows back to their places (tiny: [n, <= 21] per MEET group)
            full = torch.zeros(ctx.shape, dtype=grad.dtype, device=grad.device)
            full.index_add_(0, ctx.rows.to(grad.device), grad)
            grad = full
        return grad, None, None, None


def ce_loss(logits, labels, weight=None, rows=None):
    """relation_ce_loss as a differentiable scalar: the value and d loss / d logits both come from veto_ce_loss."""
    return _CELossFn.apply(logits, labels, weight, rows)


def _numpy_generator_from_python_random():
    st = random.getstate()
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": np.array(st[1][:624], dtype=np.uint32), "pos": int(st[1][624])}}
    return bg, st


class MeetTrainingSampler:
    def __init__(self, dataset, group_sizes, device="cuda", zero_label_padding_mode="rand_insert"):
        if zero_label_padding_mode != "rand_insert":
            raise NotImplementedError("only GCL_SETTING.ZERO_LABEL_PADDING_MODE = 'rand_insert' (VETO_final.yaml) is built")
        self.device = torch.device(device)
        self.sizes = [int(x) for x in group_sizes]
        incre = meet_tables.incre_idx_list(self.sizes)
        pos, seen = [0] * len(incre), {}
        for c, g in enumerate(incre):
            if g:
                seen[g] = seen.get(g, 0) + 1
                pos[c] = seen[g]
        i32 = dict(dtype=torch.int32, device=self.device)
        self.n_cls = len(incre)
        self.incre = torch.tensor(incre, **i32)
        self.pos_in_group = torch.tensor(pos, **i32)
        self.group_size = torch.tensor(self.sizes, **i32)
        self.rates = torch.tensor(meet_tables.sample_rate_matrix(dataset, self.sizes), dtype=torch.float64, device=self.device)

    def sample(self, rel_labels):
        """rel_labels: [n] int64.  Returns (chosen, group_labels): per group the selected row indices (the reference's
        cur_chosen_matrix[0][k]) and their group-local labels.  Advances Python's `random` exactly as the reference's
        loop would have."""
        lib = native.load_library()
        dev, G = self.device, len(self.sizes)
        labels = rel_labels.to(device=dev, dtype=torch.int64).contiguous()
        n = int(labels.shape[0])
        n_words = 4 * n + 64      # 2 words per foreground relation; background: 1 + rejections (p < 1/2 each)
        while True:
            bg, st = _numpy_generator_from_python_random()
            words = torch.from_numpy(bg.random_raw(n_words).astype(np.uint32).view(np.int32)).to(dev)
            chosen = torch.empty((G, n), dtype=torch.int64, device=dev)
            glabels = torch.empty((G, n), dtype=torch.int64, device=dev)
            meta = torch.empty(G + 1, dtype=torch.int32, device=dev)
            stream = torch.cuda.current_stream(dev)
            native.check(lib.veto_meet_sample(
                ctypes.c_void_p(stream.cuda_stream), labels.data_ptr(), n, words.data_ptr(), n_words, self.incre.data_ptr(),
                self.pos_in_group.data_ptr(), self.group_size.data_ptr(), self.rates.data_ptr(), G, self.n_cls,
                chosen.data_ptr(), glabels.data_ptr(), meta.data_ptr(), meta[G:].data_ptr()))
            host = meta.cpu().tolist()       # the list lengths are data dependent: one small read-back
            if host[G] >= 0:
                break
            n_words *= 2                      # astronomically unlikely: the rejection loop outran the block
        used = host[G]
        bg2, st = _numpy_generator_from_python_random()
        if used:
            bg2.random_raw(used)
        s2 = bg2.state["state"]
        random.setstate((st[0], tuple(int(x) for x in s2["key"]) + (int(s2["pos"]),), st[2]))
        return [chosen[k, :host[k]] for k in range(G)], [glabels[k, :host[k]] for k in range(G)]
