"""Backward building blocks of the relation transformer (groundwork) against torch
autograd in float64 on the CPU -- autograd over these standard ops IS what the reference's backward computes
(model_veto.py uses nn.LayerNorm, nn.GELU, softmax attention)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from veto_amd import native
    return native, native.load_library()


@pytest.mark.parametrize("heads,n_pair", [(8, 5), (6, 3), (4, 2)])
def test_attention_backward_against_autograd(heads, n_pair):
    native, lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(heads * 10 + n_pair)
    dh = 576 // heads
    qkv = torch.randn(n_pair * 19, 1728, generator=g)
    dout = torch.randn(n_pair * 19, 576, generator=g)
    x = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(n_pair, 19, heads, dh).transpose(1, 2) for t in x.split(576, dim=1)]      # 'b n (h d) -> b h n d'
    attn = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1)                                  # model_veto.py:88-92
    out = (attn @ v).transpose(1, 2).reshape(n_pair * 19, 576)
    (out * dout.double()).sum().backward()
    dqkv = torch.full_like(qkv, float("nan")).to(dev)
    qkv_d, dout_d = qkv.to(dev), dout.to(dev)      # keep the device copies alive: a temporary's block would be reused
    native.check(lib.veto_debug_attention_backward(None, qkv_d.data_ptr(), dout_d.data_ptr(), dqkv.data_ptr(), n_pair, heads))
    torch.cuda.synchronize()
    err = (dqkv.cpu().double() - x.grad).abs().max().item()
    assert err < 2e-5 * max(1.0, x.grad.abs().max().item()), err


@pytest.mark.parametrize("rows,with_res", [(37, False), (1000, True)])
def test_layernorm_backward_against_autograd(rows, with_res):
    native, lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, 576, generator=g) * 2 + 0.3
    dy = torch.randn(rows, 576, generator=g)
    gamma = torch.randn(576, generator=g)
    beta = torch.randn(576, generator=g)
    dres = torch.randn(rows, 576, generator=g) if with_res else None
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xd, (576,), gd, bd, 1e-5)
    (y * dy.double()).sum().backward()
    want_dx = xd.grad + (dres.double() if with_res else 0)
    dx = torch.empty(rows, 576, device=dev)
    dgb = torch.empty(2, 576, device=dev)
    ws = torch.empty(lib.veto_debug_layernorm_backward_workspace_bytes(rows), dtype=torch.uint8, device=dev)
    x_d, dy_d, gamma_d, dres_d = x.to(dev), dy.to(dev), gamma.to(dev), (dres.to(dev) if with_res else None)
    native.check(lib.veto_debug_layernorm_backward(None, x_d.data_ptr(), dy_d.data_ptr(), gamma_d.data_ptr(),
                                                   dres_d.data_ptr() if with_res else None, dx.data_ptr(), dgb.data_ptr(),
                                                   rows, ws.data_ptr(), ws.numel()))
    torch.cuda.synchronize()
    assert (dx.cpu().double() - want_dx).abs().max().item() < 2e-5
    assert (dgb[0].cpu().double() - gd.grad).abs().max().item() < 1e-4 * max(1.0, gd.grad.abs().max().item())
    assert (dgb[1].cpu().double() - bd.grad).abs().max().item() < 1e-4 * max(1.0, bd.grad.abs().max().item())


def test_gelu_backward_and_column_sums_against_autograd():
    native, lib = _lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    pre = torch.randn(513, 1152, generator=g) * 2
    dh = torch.randn(513, 1152, generator=g)
    pd = pre.double().requires_grad_(True)
    (torch.nn.functional.gelu(pd) * dh.double()).sum().backward()
    dpre = torch.empty_like(pre).to(dev)
    pre_d, dh_d = pre.to(dev), dh.to(dev)
    native.check(lib.veto_debug_gelu_backward(None, pre_d.data_ptr(), dh_d.data_ptr(), dpre.data_ptr(), pre.numel()))
    out = torch.empty(1152, device=dev)
    ws = torch.empty(1024 * 1152 * 4, dtype=torch.uint8, device=dev)
    dyd = dh.to(dev)
    native.check(lib.veto_debug_column_sums(None, dyd.data_ptr(), 1152, 513, 1152, out.data_ptr(), ws.data_ptr(), ws.numel()))
    torch.cuda.synchronize()
    assert (dpre.cpu().double() - pd.grad).abs().max().item() < 2e-6
    assert (out.cpu().double() - dh.double().sum(0)).abs().max().item() < 1e-4
