"""Diagnostic cases for the fused FeedForward kernel (veto_debug_ffn mode 1): structured operands that isolate the epilogue, the
fc2 phase and the fc1 phase, with a map of where the errors sit.  usage: python tools/ffn_diag.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from veto_amd import native

lib = native.load_library()
dev = torch.device("cuda:0")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def run(a, w1, b1, w2, b2, x0, mode=1):
    ws = torch.empty(lib.veto_debug_ffn_workspace_bytes(m), dtype=torch.uint8, device=dev)
    x = x0.clone()
    native.check(lib.veto_debug_ffn(None, a.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), x.data_ptr(),
                                    m, mode, 1, 1, None, ws.data_ptr(), ws.numel(), None, None, None))
    torch.cuda.synchronize()
    return x


def ref(a, w1, b1, w2, b2, x0):
    hid = torch.nn.functional.gelu(a.double() @ w1.double().t() + b1.double())
    return x0.double() + hid @ w2.double().t() + b2.double()


def report(name, got, want):
    err = (got.double() - want).abs()
    bad = err > 1e-3 * (1 + want.abs())
    print("%-34s max err %.3e, bad %d of %d" % (name, err.max().item(), int(bad.sum()), bad.numel()))
    if bad.any():
        rows = bad.any(dim=1).nonzero().flatten()
        cols = bad.any(dim=0).nonzero().flatten()
        print("   bad rows: %d (first %s) rows%%128 set %s" % (rows.numel(), rows[:8].tolist(), sorted(set((rows % 128).tolist()))[:40]))
        print("   bad cols: %d (first %s) cols%%64 set %s" % (cols.numel(), cols[:8].tolist(), sorted(set((cols % 64).tolist()))[:70]))
        r, c = int(rows[0]), int(cols[0])
        print("   sample got %s want %s" % (got[r, c:c + 4].tolist(), want[r, c:c + 4].tolist()))


g = torch.Generator(device="cpu").manual_seed(1)
A = torch.randn(m, 576, generator=g).to(dev)
X = torch.randn(m, 576, generator=g).to(dev)
W1 = (torch.randn(1152, 576, generator=g) * 0.04).to(dev)
B1 = (torch.randn(1152, generator=g) * 0.1).to(dev)
W2 = (torch.randn(576, 1152, generator=g) * 0.03).to(dev)
B2 = (torch.randn(576, generator=g) * 0.1).to(dev)
Z1, Z2 = torch.zeros_like(W1), torch.zeros_like(W2)
zb1, zb2 = torch.zeros_like(B1), torch.zeros_like(B2)

cases = [
    ("epilogue only (w2 = 0)", (A, W1, B1, Z2, B2, X)),
    ("fc2 only (w1 = 0, b1 = 1)", (A, Z1, torch.ones_like(B1), W2, zb2, X)),
    ("fc2 only, b1 random", (A, Z1, B1 * 10, W2, zb2, X)),
    ("fc1 -> one hidden unit per out", None),
    ("full", (A, W1, B1, W2, B2, X)),
]
for name, args in cases:
    if args is None:
        # w2 picks hidden unit j for output column j % 576 ... : out[:, n] = gelu(fc1)[:, n] (n < 576) + gelu(fc1)[:, n + 576]
        w2 = torch.zeros(576, 1152, device=dev)
        idx = torch.arange(576, device=dev)
        w2[idx, idx] = 1.0
        args = (A, W1, B1, w2, zb2, torch.zeros_like(X))
    want = ref(*args)
    report(name + " [fused]", run(*args, mode=1), want)
    report(name + " [two launches]", run(*args, mode=0), want)

# ---- decode test: hidden[row][u] = 8 + (row % 128) / 256 + u / 4096 (gelu is the identity there), out col n picks hidden unit
# n + 576 * half: a wrong entry tells which (row, unit) it really came from
print("decode test")
a = torch.zeros(m, 576, device=dev)
a[:, 0] = (torch.arange(m, device=dev) % 128).float() / 256
w1 = torch.zeros(1152, 576, device=dev)
w1[:, 0] = 1.0
b1 = 8 + torch.arange(1152, device=dev).float() / 4096
for half in (0, 1):
    w2 = torch.zeros(576, 1152, device=dev)
    idx = torch.arange(576, device=dev)
    w2[idx, idx + 576 * half] = 1.0
    args = (a, w1, b1, w2, zb2, torch.zeros_like(X))
    got = run(*args, mode=1)
    want = ref(*args)
    err = (got.double() - want).abs()
    bad = (err > 2e-3).nonzero()
    print(" half %d: bad %d" % (half, bad.shape[0]))
    for r, c in bad[:24].tolist():
        v = got[r, c].item() - 8
        # v = row'/256 + u'/4096 with row' < 128, u' < 1152: u'/4096 < 0.2813, row'/256 multiples of 1/256 = 16/4096
        q = round(v * 4096)
        print("   out[%d][%d] (unit %d): got %.5f want %.5f  -> code %d (want %d) diff %d" % (r, c, c + 576 * half, got[r, c].item(), want[r, c].item(), q,
              round((want[r, c].item() - 8) * 4096), q - round((want[r, c].item() - 8) * 4096)))

# ---- probe: hidden = 8 everywhere (w1 = 0, b1 = 8), W2[n][k] = 1/64 on a chosen set of k: which k positions go wrong?
print("probe: constant hidden, W2 = 1/64 on selected k")
w1 = torch.zeros(1152, 576, device=dev)
b1 = torch.full((1152,), 8.0, device=dev)
a = torch.randn(m, 576, device=dev)
def probe(sel, label):
    w2 = torch.zeros(576, 1152, device=dev)
    w2[:, sel] = 1.0 / 64
    args = (a, w1, b1, w2, zb2, torch.zeros_like(X))
    got = run(*args, mode=1)
    want = ref(*args)
    err = (got.double() - want).abs()
    bad = err > 1e-3
    if bad.any():
        rows = bad.any(dim=1).nonzero().flatten()
        cols = bad.any(dim=0).nonzero().flatten()
        print("  %-22s bad %6d max err %.4f rows%%32 %s cols %d (first %s) got %.4f want %.4f" % (label, int(bad.sum()), err.max().item(),
              sorted(set((rows % 32).tolist())), cols.numel(), cols[:6].tolist(), got[rows[0], cols[0]].item(), want[rows[0], cols[0]].item()))
    return bool(bad.any())
ks = torch.arange(1152, device=dev)
probe(ks >= 0, "all k")
bad_kk = [kk for kk in range(64) if probe(ks % 64 == kk, "k%%64 == %d" % kk)]
print("  bad k%64:", bad_kk)
bad_kb = [kb for kb in range(18) if probe(ks // 64 == kb, "k//64 == %d" % kb)]
print("  bad k//64:", bad_kb)
