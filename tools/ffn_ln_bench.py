import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from veto_amd import native
lib = native.load_library(); dev = torch.device("cuda:0")
m = 287280
a = torch.randn(m, 576, device=dev); x = torch.randn(m, 576, device=dev)
w1 = torch.randn(1152, 576, device=dev) * 0.04; b1 = torch.randn(1152, device=dev) * 0.1
w2 = torch.randn(576, 1152, device=dev) * 0.03; b2 = torch.randn(576, device=dev) * 0.1
lw = torch.ones(576, device=dev); lb = torch.zeros(576, device=dev)
rows = torch.zeros(m, 2304, dtype=torch.uint8, device=dev)
ws = torch.empty(lib.veto_debug_ffn_workspace_bytes(m), dtype=torch.uint8, device=dev)
for ln in (0, 1):
    ms = ctypes.c_float(0)
    for it in range(3):
        native.check(lib.veto_debug_ffn(None, a.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(), m, 1, 1 if it == 0 else 0, 5,
                                        ctypes.byref(ms), ws.data_ptr(), ws.numel(), lw.data_ptr() if ln else None, lb.data_ptr() if ln else None, rows.data_ptr() if ln else None))
    print("ln=%d: %.3f ms" % (ln, ms.value), flush=True)
