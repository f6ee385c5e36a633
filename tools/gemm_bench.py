"""Times the production GEMM (through veto_debug_gemm) on the four transformer shapes of cfg-2.
usage: python tools/gemm_bench.py [n_shapes] [precision: 0 precise | 1 fast | 2 mixed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veto_amd import native
lib = native.load_library()
dev = torch.device("cuda:0")
shapes = [(287280, 1728, 576), (287280, 576, 576), (287280, 1152, 576), (287280, 576, 1152)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for (m, n, k) in shapes:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05
    c = torch.empty(m, n, device=dev)
    ws = torch.empty(lib.veto_debug_gemm_workspace_bytes(m, n, k), dtype=torch.uint8, device=dev)
    # veto_debug_gemm = split kernels + gemm; time the gemm alone via the difference to a split-only call is
    # fiddly, so time whole calls and subtract the measured split time (M=1 rows of N... negligible gemm)
    def run():
        native.check(lib.veto_debug_gemm(None, a.data_ptr(), w.data_ptr(), None, c.data_ptr(), m, n, k, prec,
                                         ws.data_ptr(), ws.numel()))
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("M%d N%d K%d: %.3f ms per call (incl. ~%.2f ms of operand splitting) -> <=%.0f TF" %
          (m, n, k, ms, (m * k * 8 + 2 * m * k * 2) / 4.5e9, 2.0 * m * n * k / (ms * 1e-3) / 1e12))
