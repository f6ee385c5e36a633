"""Everything of a layer behind its attention through veto_debug_layer_tail: ONE launch (mode 1) against the out-projection panel
launch + the FeedForward panel launch (mode 0).  usage: python tools/layer_tail_bench.py [rows]"""
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
wo = torch.randn(576, 576, device=dev) * 0.05
bo = torch.randn(576, device=dev) * 0.1
w1 = torch.randn(1152, 576, device=dev) * 0.04
b1 = torch.randn(1152, device=dev) * 0.1
w2 = torch.randn(576, 1152, device=dev) * 0.03
b2 = torch.randn(576, device=dev) * 0.1
lw = torch.ones(576, device=dev)
lb = torch.zeros(576, device=dev)
rows = torch.zeros(m, 2304, dtype=torch.uint8, device=dev)
ws = torch.empty(lib.veto_debug_layer_tail_workspace_bytes(m), dtype=torch.uint8, device=dev)
for mode, name in ((0, "out-projection launch + FeedForward launch"), (1, "one launch")):
    ms = ctypes.c_float(0)
    for it in range(2):
        native.check(lib.veto_debug_layer_tail(None, a.data_ptr(), wo.data_ptr(), bo.data_ptr(), lw.data_ptr(), lb.data_ptr(), w1.data_ptr(),
                                               b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), m, mode, 5, ctypes.byref(ms),
                                               ws.data_ptr(), ws.numel(), lw.data_ptr(), lb.data_ptr(), rows.data_ptr()))
    print("M=%d %-44s %.3f ms" % (m, name, ms.value), flush=True)
