"""Test-time pair enumeration, the caller-side contract of the hot path.

Mirrors `RelationSampling.prepare_test_pairs` (sampling.py:31-52) for the GT-box modes the
predictor is benchmarked in (predcls / sgcls): every ordered pair (i, j), i != j, in the row-major
order of `torch.nonzero(ones - eye)`, or the `[[0, 0]]` placeholder when an image has no candidate
pair.  The enumeration runs on the device through the C ABI (veto_enumerate_pairs)."""
import ctypes

import torch

from . import native


def prepare_test_pairs(device, proposals, max_proposal_pairs=2048):
    lib = native.load_library()
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("veto_amd.prepare_test_pairs runs on a HIP device only (got %s)" % device)
    stream = torch.cuda.current_stream(device).cuda_stream
    out = []
    for p in proposals:
        n = len(p)
        total = n * (n - 1) if n > 1 else 1
        idxs = torch.empty((total, 2), dtype=torch.int64, device=device)
        native.check(lib.veto_enumerate_pairs(ctypes.c_void_p(stream), n, ctypes.c_void_p(idxs.data_ptr())))
        if total > max_proposal_pairs:
            # sampling.py:41-45: keep the MAX_PROPOSAL_PAIR best pairs by pred_scores product
            q = p.get_field("pred_scores").to(device)
            q = q[idxs[:, 0]] * q[idxs[:, 1]]
            idxs = idxs[torch.sort(q, descending=True)[1][:max_proposal_pairs]]
        out.append(idxs)
    return out
