"""Context for the split GEMM's issued MFMA rate: what the vendor bf16 GEMM (torch.matmul -> hipBLASLt) reaches on this box at the
same output shapes -- once with the true K (one bf16 pass: NOT parity-capable) and once with 3 K, i.e. the
same number of MFMA flops the 3-term split kernel issues for that Linear."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
M = 287280
for name, n, k in (("qkv", 1728, 576), ("out", 576, 576), ("fc1", 1152, 576), ("fc2", 576, 1152)):
    for mult in (1, 3):
        a = torch.randn(M, k * mult, device=dev, dtype=torch.bfloat16)
        w = torch.randn(n, k * mult, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            c = a @ w.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c = a @ w.t()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-3s M %d N %4d K %4d (x%d): %.3f ms, %.0f TFLOP/s of bf16 MFMA (bf16 output)" % (name, M, n, k * mult, mult, ms, 2.0 * M * n * k * mult / (ms * 1e-3) / 1e12))
        del a, w, c
