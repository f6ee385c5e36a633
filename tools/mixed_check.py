"""GPU check of the VETO_MIXED GEMM and forward: (1) veto_debug_gemm in the three precision modes against fp64,
(2) golden parity of the forward in mixed vs precise mode, (3) GEMM time per mode at the four transformer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from veto_amd import native, testing
lib = native.load_library()
dev = torch.device("cuda:0")

for (m, n, k) in [(300, 384, 576), (1000, 1728, 576), (777, 576, 1152)]:
    g = torch.Generator().manual_seed(m)
    a = torch.randn(m, k, generator=g).to(dev); w = (torch.randn(n, k, generator=g) * 0.04).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    ref = a.double() @ w.double().t() + bias.double()
    scale = (a.abs().double() @ w.abs().double().t()).clamp_min(1e-6)
    ws = torch.empty(lib.veto_debug_gemm_workspace_bytes(m, n, k), dtype=torch.uint8, device=dev)
    for prec in (0, 2, 1):
        c = torch.full((m, n), float("nan"), device=dev)
        native.check(lib.veto_debug_gemm(None, a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, prec, ws.data_ptr(), ws.numel()))
        torch.cuda.synchronize()
        print("gemm M%d N%d K%d precision %d: max rel err %.3e, max abs err %.3e" % (m, n, k, prec, ((c.double() - ref).abs() / scale).max().item(), (c.double() - ref).abs().max().item()))

from conftest import load_golden
from veto_amd.pairs import prepare_test_pairs
for name in ["predcls_n10_l4h8", "predcls_n36_l4h8", "predcls_n36_l6h6", "ragged_l4h8", "sgcls_n10_l6h6"]:
    g, sd, batch = load_golden(name)
    for prec in ("precise", "mixed"):
        cfg = testing.make_config(g["_layers"], g["_heads"], str(g["mode"]), False, str(g["dataset"]), precision=prec)
        model = testing.make_predictor(cfg, sd, dev)
        props = testing.make_proposals(batch, str(g["mode"]), dev)
        pairs = prepare_test_pairs(dev, props)
        with torch.no_grad():
            out = model(props, pairs, None, None, roi_features=torch.from_numpy(batch["roi_features"]).to(dev),
                        roi_depth_features=torch.from_numpy(batch["roi_depth_features"]).to(dev))
        got = torch.cat(list(out[1])).cpu().numpy()
        print("%s %s: logit max-abs-err %.3e" % (name, prec, np.abs(got - g["rel_dists"]).max()))

for (m, n, k) in [(287280, 1728, 576), (287280, 576, 576), (287280, 1152, 576), (287280, 576, 1152)]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05
    c = torch.empty(m, n, device=dev)
    ws = torch.empty(lib.veto_debug_gemm_workspace_bytes(m, n, k), dtype=torch.uint8, device=dev)
    for prec in (0, 2):
        run = lambda: native.check(lib.veto_debug_gemm(None, a.data_ptr(), w.data_ptr(), None, c.data_ptr(), m, n, k, prec, ws.data_ptr(), ws.numel()))
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        print("M%d N%d K%d precision %d: %.3f ms per call incl. operand conversion" % (m, n, k, prec, e0.elapsed_time(e1) / 5))
