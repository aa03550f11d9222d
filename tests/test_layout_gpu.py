"""Data-movement kernels (concatenation with broadcast, window means, dropout, mel pyramid) through the C ABI vs the
torch operators the reference uses.  fp32, exact or 1e-6."""
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu


@gpu
def test_concat2_broadcast_forms():
    """generator.py:249-250 (phonemes tiled over frequency | noise), :257 (map | map), :312-314 (map | vector)."""
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(0)
    n, h, w = 2, 5, 7
    full = torch.randn(n, h, w, 12, generator=g)
    overh = torch.randn(n, w, 8, generator=g)
    vec = torch.randn(n, 6, generator=g)
    cases = [(overh, full, overh[:, None].expand(n, h, w, 8)), (full, full * 2, None), (full, vec, None)]
    for x1, x2, _ in cases:
        e1 = x1 if x1.dim() == 4 else (x1[:, None].expand(n, h, w, -1) if x1.dim() == 3 else x1[:, None, None].expand(n, h, w, -1))
        e2 = x2 if x2.dim() == 4 else (x2[:, None].expand(n, h, w, -1) if x2.dim() == 3 else x2[:, None, None].expand(n, h, w, -1))
        r1, r2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        er1 = r1 if r1.dim() == 4 else (r1[:, None].expand(n, h, w, -1) if r1.dim() == 3 else r1[:, None, None].expand(n, h, w, -1))
        er2 = r2 if r2.dim() == 4 else (r2[:, None].expand(n, h, w, -1) if r2.dim() == 3 else r2[:, None, None].expand(n, h, w, -1))
        ref = torch.cat([er1, er2], 3)
        gy = torch.randn(ref.shape, generator=g)
        ref.backward(gy)
        a, b = x1.cuda().requires_grad_(True), x2.cuda().requires_grad_(True)
        y = ops.concat2(a, b, n, h, w)
        assert torch.equal(y.cpu(), torch.cat([e1, e2], 3))
        y.backward(gy.cuda())
        assert torch.allclose(a.grad.cpu(), r1.grad, atol=1e-5) and torch.allclose(b.grad.cpu(), r2.grad, atol=1e-5)


@gpu
def test_window_mean_and_double_backward():
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(3, 5, 6, 8, generator=g)
    for wh, ww in ((4, 4), (5, 6)):
        x = x0.cuda().requires_grad_(True)
        y = ops.WindowMeanFn.apply(x, wh, ww)
        assert torch.allclose(y.cpu(), x0[:, :wh, :ww].mean((1, 2)), atol=1e-6)
        # R1-style: gradient of a function of the mean w.r.t. x, then differentiate its squared norm again
        gx, = torch.autograd.grad((y ** 2).sum(), x, create_graph=True)
        gx.pow(2).sum().backward()
        xr = x0.clone().requires_grad_(True)
        yr = xr[:, :wh, :ww].mean((1, 2))
        gr, = torch.autograd.grad((yr ** 2).sum(), xr, create_graph=True)
        gr.pow(2).sum().backward()
        assert torch.allclose(gx.detach().cpu(), gr.detach(), atol=1e-6)
        assert torch.allclose(x.grad.cpu(), xr.grad, atol=1e-6)


@gpu
def test_dropout_mask_statistics_and_streams():
    from vcagan_amd import ops
    dev = torch.device('cuda', 0)
    x = torch.ones(400, 512, device=dev, requires_grad=True)
    ops.rng_advance(dev)
    y = ops.dropout(x, 0.3, True)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.7) < 0.01, keep
    assert torch.allclose(y[y != 0], torch.full((1,), 1 / 0.7, device=dev))
    y.sum().backward()
    assert torch.equal(x.grad != 0, y != 0) and torch.allclose(x.grad[x.grad != 0], torch.full((1,), 1 / 0.7, device=dev))
    y2 = ops.dropout(x, 0.3, True)                       # second call site in the same step: another stream
    ops.rng_advance(dev)
    y3 = ops.dropout(x, 0.3, True)                       # same call site, next step: another stream
    for other in (y2, y3):
        agree = ((other != 0) == (y != 0)).float().mean().item()
        assert 0.5 < agree < 0.66, agree                 # independent masks agree on 0.7^2 + 0.3^2 = 0.58
    assert ops.dropout(x, 0.3, False) is x and ops.dropout(x, 0.0, True) is x


@gpu
@pytest.mark.parametrize('factor', [2, 4])
def test_mel_pyramid_matches_interpolate(factor):
    from vcagan_amd import ops
    mel = torch.rand(3, 1, 80, 300, generator=torch.Generator().manual_seed(2)) * 2 - 1
    ref = F.interpolate(mel, scale_factor=1.0 / factor, mode='bilinear', align_corners=False)
    out = ops.mel_down(mel.cuda(), factor)
    assert out.shape == ref.shape and torch.allclose(out.cpu(), ref, atol=1e-6)
