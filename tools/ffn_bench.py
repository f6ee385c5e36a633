"""FeedForward block (fc1 -> GELU -> fc2 + residual) through veto_debug_ffn: the fused kernel (mode 1) against the two-launch form
(mode 0), both checked against an fp64 reference and timed with hipEvents inside the library.
usage: python tools/ffn_bench.py [rows ...]        (VETO_AMD_LIB=build/libveto_ffn_stamps.so for the in-kernel stamps)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from veto_amd import native

lib = native.load_library()
dev = torch.device("cuda:0")
rows = [int(x) for x in sys.argv[1:]] or [1000, 287280]
reps = int(os.environ.get("FFN_REPS", "10"))
fast = os.environ.get("FFN_FAST") == "1"     # timing of the fused kernel only (variant libraries: no reference, no two-launch form)
for m in rows:
    g = torch.Generator(device="cpu").manual_seed(m)
    a = torch.randn(m, 576, generator=g).to(dev)
    x0 = torch.randn(m, 576, generator=g).to(dev)
    w1 = (torch.randn(1152, 576, generator=g) * 0.04).to(dev)
    b1 = (torch.randn(1152, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(576, 1152, generator=g) * 0.03).to(dev)
    b2 = (torch.randn(576, generator=g) * 0.1).to(dev)
    if not fast:
        hid = torch.nn.functional.gelu(a.double() @ w1.double().t() + b1.double())
        ref = x0.double() + hid @ w2.double().t() + b2.double()
        scale = (hid.abs() @ w2.double().abs().t()).clamp_min(1e-6)
        del hid
    ws = torch.empty(lib.veto_debug_ffn_workspace_bytes(m), dtype=torch.uint8, device=dev)
    for mode, name in (((1, "fused"),) if fast else ((0, "two launches"), (1, "fused"))):
        x = x0.clone()
        native.check(lib.veto_debug_ffn(None, a.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(),
                                        m, mode, 1, 1, None, ws.data_ptr(), ws.numel(), None, None, None))
        torch.cuda.synchronize()
        if fast:
            msg = "M=%d %-12s %s" % (m, name, os.path.basename(os.environ.get("VETO_AMD_LIB", "libveto_amd.so")))
        else:
            err = (x.double() - ref).abs()
            bad = int((~torch.isfinite(x)).sum().item())
            msg = "M=%d %-12s max-abs-err %.3e  rel-to-sum|h||w| %.3e  non-finite %d" % (m, name, err.max().item(), (err / scale).max().item(), bad)
        ms = ctypes.c_float(0)
        for _ in range(2):   # the first timed batch warms the clocks
            native.check(lib.veto_debug_ffn(None, a.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(),
                                            m, mode, 0, reps, ctypes.byref(ms), ws.data_ptr(), ws.numel(), None, None, None))
        flops = 2.0 * m * 576 * 1152 * 2
        print("%s  %.3f ms  %.0f TFLOP/s alg." % (msg, ms.value, flops / (ms.value * 1e-3) / 1e12), flush=True)
    if not fast:
        del ref, scale
