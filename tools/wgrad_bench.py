"""Times the weight-gradient GEMM dw[N,K] = dy[M,N]^T . x[M,K] (split-K + atomic adds through the production
kernel, operands transposed into split rows first) on the transformer's four Linear shapes at cfg-2 size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veto_amd import native
lib = native.load_library()
dev = torch.device("cuda:0")
M = 287280
for (n, k) in [(1728, 576), (576, 576), (1152, 576), (576, 1152)]:
    dy = torch.randn(M, n, device=dev); x = torch.randn(M, k, device=dev)
    dw = torch.empty(n, k, device=dev)
    ws = torch.empty(lib.veto_debug_wgrad_workspace_bytes(M, n, k, 0), dtype=torch.uint8, device=dev)
    run = lambda: native.check(lib.veto_debug_wgrad(None, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), M, n, k, 0, ws.data_ptr(), ws.numel()))
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("dW[%d,%d] over M=%d: %.3f ms per call incl. transposes -> %.0f TF alg." % (n, k, M, ms, 2.0 * M * n * k / (ms * 1e-3) / 1e12))
    del dy, x, ws
