"""Runs veto_debug_gemm once per (shape, precision) on a -DVETO_GEMM_STAMPS build of the library (VETO_AMD_LIB), which prints
the mean s_memtime cycles per phase of the persistent GEMM (consumer: barrier wait / MFMA phase / epilogue; loader: vmcnt wait /
barrier / issue).  usage: VETO_AMD_LIB=build/libveto_stamps.so python tools/gemm_stamps.py [precisions...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veto_amd import native
lib = native.load_library()
dev = torch.device("cuda:0")
precs = [int(x) for x in sys.argv[1:]] or [0, 2]
for (m, n, k) in [(287280, 1728, 576), (287280, 576, 576), (287280, 1152, 576), (287280, 576, 1152)]:
    a = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05
    c = torch.empty(m, n, device=dev)
    ws = torch.empty(lib.veto_debug_gemm_workspace_bytes(m, n, k), dtype=torch.uint8, device=dev)
    for prec in precs:
        for rep in range(2):
            sys.stderr.write("precision %d rep %d: " % (prec, rep)); sys.stderr.flush()
            native.check(lib.veto_debug_gemm(None, a.data_ptr(), w.data_ptr(), None, c.data_ptr(), m, n, k, prec, ws.data_ptr(), ws.numel()))
            torch.cuda.synchronize()
