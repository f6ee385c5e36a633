"""Attention out projection + residual (+ LayerNorm rows) through veto_debug_outproj: the full-row panel kernel (mode 1) against the
GEMM launch + LayerNorm launch (mode 0), timed with hipEvents inside the library.  usage: python tools/outproj_bench.py [rows]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from veto_amd import native

lib = native.load_library()
dev = torch.device("cuda:0")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 287280
a = torch.randn(m, 576, device=dev)
x = torch.randn(m, 576, device=dev)
w = torch.randn(576, 576, device=dev) * 0.05
b = torch.randn(576, device=dev) * 0.1
lw = torch.ones(576, device=dev)
lb = torch.zeros(576, device=dev)
rows = torch.zeros(m, 2304, dtype=torch.uint8, device=dev)
ws = torch.empty(lib.veto_debug_outproj_workspace_bytes(m), dtype=torch.uint8, device=dev)
for ln in (0, 1):
    for mode, name in ((0, "GEMM launch%s" % (" + LayerNorm launch" if ln else "")), (1, "panel kernel")):
        ms = ctypes.c_float(0)
        for it in range(2):
            native.check(lib.veto_debug_outproj(None, a.data_ptr(), w.data_ptr(), b.data_ptr(), x.data_ptr(), m, mode, 1, 5, ctypes.byref(ms),
                                                ws.data_ptr(), ws.numel(), lw.data_ptr() if ln else None, lb.data_ptr() if ln else None,
                                                rows.data_ptr() if ln else None))
        print("M=%d layernorm=%d %-34s %.3f ms" % (m, ln, name, ms.value), flush=True)
