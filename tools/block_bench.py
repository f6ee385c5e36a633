"""Time single building-block kernels of the training path through their test hooks (attention backward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veto_amd import native

def main():
    n_pair, heads = 15120, 8
    lib = native.load_library()
    dev = torch.device("cuda:0")
    qkv = torch.randn(n_pair * 19, 1728, device=dev)
    dout = torch.randn(n_pair * 19, 576, device=dev)
    dqkv = torch.empty_like(qkv)
    run = lambda: native.check(lib.veto_debug_attention_backward(None, qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), n_pair, heads))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record(torch.cuda.default_stream(dev))
    for _ in range(n):
        run()
    e1.record(torch.cuda.default_stream(dev))
    torch.cuda.synchronize()
    print("attention backward: %.1f us per launch (%d pairs, %d heads)" % (e0.elapsed_time(e1) * 1e3 / n, n_pair, heads))

if __name__ == "__main__":
    main()
