"""Linear kernel (vca_linear_bf16: nn.Linear and 1x1 / stride-1 convolutions, forward and input gradient) against
torch fp32 operators: skinny and tiny shapes, ragged row / column / reduction tails, bias, fused residual, and the
strided views the attention projections use."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(autouse=True)
def _reset_precision():
    yield
    from vcagan_amd import ops
    ops.set_precision('fp32')


SHAPES = [(2400, 256, 2560), (32, 1, 512), (33, 321, 256), (4800, 1280, 256), (70, 24, 40), (2400, 1536, 1024),
          (1, 8, 8), (640, 512, 5120), (100, 7, 136)]


@gpu
@pytest.mark.parametrize('kernel', ['skinny', 'tiled'])
@pytest.mark.parametrize('prec,tol', [('bf16x3', 1e-4), ('bf16', 2e-2)])
@pytest.mark.parametrize('M,N,K', SHAPES)
def test_linear_forward_and_gradients(M, N, K, prec, tol, kernel, monkeypatch):
    from vcagan_amd import ops
    ops.set_precision(prec)
    # every shape through each of the two kernels (by default the product's size decides)
    monkeypatch.setattr(ops, 'LINEAR_MAX_FLOP', 1e30 if kernel == 'skinny' else 0.0)
    monkeypatch.setattr(ops, '_tiled_wins', lambda rows, n, k: kernel == 'tiled')
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.cuda().requires_grad_(True)
    wd, bd = torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda())
    y = ops.linear(xd, wd, bd)
    assert rel_err(y, yr) < tol
    y.backward(gy.cuda())
    assert rel_err(xd.grad, xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol and rel_err(bd.grad, br.grad) < 1e-5


@gpu
def test_pointwise_conv_input_gradient_with_fused_residual():
    """_conv_dgrad_raw on a 1x1 convolution adds `res` in the linear kernel's epilogue."""
    from vcagan_amd import ops
    ops.set_precision('bf16x3')
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(3, 10, 37, 128, generator=g).cuda()
    w = (torch.randn(128, 64, 1, 1, generator=g) * 0.1).cuda()
    res = torch.randn(3, 10, 37, 64, generator=g).cuda()
    cfg = ops.ConvCfg(w.shape, (3, 10, 37, 64), 1, (0, 0), 0)
    got = ops._conv_dgrad_raw(gy, w, cfg, res)
    want = torch.einsum('nhwo,oi->nhwi', gy.double(), w[:, :, 0, 0].double()) + res.double()
    assert rel_err(got, want) < 1e-4


@gpu
def test_linear_strided_views():
    """A as a (B, F, T, C) channels-last map read as rows (b, t) with reduction index (f, c); C written as the map
    (B, F, T, C') from columns (f, c'): the layouts of the attention query / output projections (generator.py:159,169)."""
    from vcagan_amd import _lib, ops
    ops.set_precision('bf16x3')
    B, Fq, T, C, N, Co = 3, 20, 37, 64, 256, 32
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, Fq, T, C, generator=g).cuda()
    w = (torch.randn(N, Fq * C, generator=g) / (Fq * C) ** 0.5).cuda()       # columns ordered (f, c)
    bias = torch.randn(N, generator=g).cuda()
    out = torch.empty(B * T, N, device='cuda')
    want = x.permute(0, 2, 1, 3).reshape(B * T, Fq * C).double() @ w.double().t() + bias.double()
    w2 = (torch.randn(Fq * Co, N, generator=g) / N ** 0.5).cuda()
    res = torch.randn(B, Fq, T, Co, generator=g).cuda()
    for tiled in (False, True):
        out = torch.empty(B * T, N, device='cuda')
        ops._linear_raw(x, ops._pack(w, 10), bias, None, out, B * T, N, Fq * C, 3,
                        a_view=(T, Fq * T * C, C, C, T * C), tiled=tiled)
        assert rel_err(out, want) < 1e-4, tiled
        # output projection: rows (b, t), columns (f, c') scattered into the map (B, F, T, C'), residual laid out alike
        ymap = torch.zeros(B, Fq, T, Co, device='cuda')
        ops._linear_raw(out, ops._pack(w2, 10), None, res, ymap, B * T, Fq * Co, N, 3,
                        c_view=(T, Fq * T * Co, Co, Co, T * Co), tiled=tiled)
        want2 = (out.double() @ w2.double().t()).view(B, T, Fq, Co).permute(0, 2, 1, 3) + res.double()
        assert rel_err(ymap, want2) < 1e-4, tiled


def test_linear_rejects_descriptors_it_cannot_take():
    """Host-side argument checks (no GPU call): reductions that are not a multiple of 8, misaligned strides, a_ck that
    would let a 32-chunk straddle two groups."""
    import __graft_entry__ as ge
    ge.build()
    from vcagan_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    f = L.vca_linear_bf16
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.POINTER(_lib.LinDesc), ctypes.c_int] + [ctypes.c_void_p] * 6
    ok = dict(M=64, N=64, K=64, a_tm=64, a_ck=64, a_bs=0, a_ts=64, a_ks=0, c_tm=64, c_cn=64, c_bs=0, c_ts=64, c_ns=0)
    for bad in (dict(K=60), dict(a_ts=62), dict(a_ck=48, K=96), dict(M=0), dict(a_tm=0)):
        d = _lib.LinDesc(**dict(ok, **bad))
        assert f(ctypes.byref(d), 3, 16, 16, None, None, 16, None) == -22, bad
    d = _lib.LinDesc(**ok)
    assert f(ctypes.byref(d), 2, 16, 16, None, None, 16, None) == -22       # nsplit must be 1 or 3
    assert f(ctypes.byref(d), 3, 8, 16, None, None, 16, None) == -22        # misaligned A
