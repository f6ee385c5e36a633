#!/usr/bin/env python3
"""Same-box A/B of the fused QKV + attention launch against the two launches it replaces (veto_debug_qkv_attn).
usage: qkv_attn_bench.py [n_pair] [heads] [reps]"""
import ctypes
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from veto_amd import native  # noqa: E402

n_pair = int(sys.argv[1]) if len(sys.argv) > 1 else 15120
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = native.load_library()
dev = torch.device("cuda:0")
a = torch.randn(n_pair * 19, 576, device=dev)
w = torch.randn(1728, 576, device=dev) * 0.05
ws = torch.empty(lib.veto_debug_qkv_attn_workspace_bytes(n_pair), dtype=torch.uint8, device=dev)
rows = torch.zeros(n_pair * 19, 4 * 576, dtype=torch.uint8, device=dev)
for rnd in range(3):
    for mode in (0, 1):
        ms = ctypes.c_float(0)
        native.check(lib.veto_debug_qkv_attn(None, a.data_ptr(), w.data_ptr(), n_pair, heads, mode, reps, ctypes.byref(ms), ws.data_ptr(),
                                             ws.numel(), rows.data_ptr()))
        torch.cuda.synchronize()
        flops = 2.0 * n_pair * 19 * 1728 * 576
        print("round %d  %-12s %.3f ms  (%.0f TFLOP/s algorithmic on the projection)" % (rnd, "fused" if mode else "two launches", ms.value,
                                                                                        flops / ms.value / 1e9), flush=True)
