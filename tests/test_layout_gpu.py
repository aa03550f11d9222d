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


@gpu
def test_hband_expand_matches_rowwise_sum():
    """y[b,h,w,c] = base + sum_{0 <= h+kh-pad < H} p[b,w,kh*C+c] and its adjoint, vs a torch loop."""
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(3)
    for (B, H, W, C, K, pad) in [(2, 20, 7, 8, 5, 2), (3, 4, 5, 12, 1, 0), (1, 3, 6, 4, 5, 2)]:
        p0 = torch.randn(B, W, K * C, generator=g)
        base0 = torch.randn(B, H, W, C, generator=g)
        gy = torch.randn(B, H, W, C, generator=g)
        pr, br = p0.clone().requires_grad_(True), base0.clone().requires_grad_(True)
        rows = []
        for h in range(H):
            acc = br[:, h]
            for kh in range(K):
                if 0 <= h + kh - pad < H:
                    acc = acc + pr[:, :, kh * C:(kh + 1) * C]
            rows.append(acc)
        ref = torch.stack(rows, 1)
        ref.backward(gy)
        p, base = p0.cuda().requires_grad_(True), base0.cuda().requires_grad_(True)
        y = ops.hband_expand(p, H, K, pad, base=base)
        assert torch.allclose(y.cpu(), ref.detach(), atol=1e-6)
        y.backward(gy.cuda())
        assert torch.allclose(p.grad.cpu(), pr.grad, atol=1e-5) and torch.allclose(base.grad.cpu(), br.grad, atol=1e-6)
        y2 = ops.hband_expand(p0.cuda(), H, K, pad)
        assert torch.allclose(y2.cpu(), (ref - br).detach(), atol=1e-5)


@gpu
@pytest.mark.parametrize('train', [True, False])
def test_tiled_decoder_input_equals_explicit_concat(train):
    """decode[0] on (phonemes tiled over the 20 rows | noise) computed from the single phoneme row
    (GenResBlk.forward_tiled) equals the block applied to the explicit concatenation (generator.py:249-251):
    output, running statistics, input gradient and every parameter gradient.  fp32 kernels, 2e-4 relative."""
    import copy
    from vcagan_amd import ops
    from vcagan_amd.models import generator as G
    ops.set_precision('fp32')
    torch.manual_seed(11)
    blk = G.GenResBlk(64 + 16, 32).cuda().train(train)
    with torch.no_grad():
        for prm in blk.parameters():
            prm.add_(torch.randn_like(prm) * 0.05)
        blk.norm1.running_var.uniform_(0.5, 2.0)
        blk.norm1.running_mean.normal_()
    blk2 = copy.deepcopy(blk)
    B, H, W = 3, 20, 9
    x0 = torch.randn(B, W, 64, device='cuda')
    nz = torch.randn(B, H, W, 16, device='cuda')
    gy = torch.randn(B, H, W, 32, device='cuda')
    xa = x0.clone().requires_grad_(True)
    ya = blk(ops.concat2(xa, nz, B, H, W))
    ya.backward(gy)
    xb = x0.clone().requires_grad_(True)
    yb = blk2.forward_tiled(xb, nz)
    yb.backward(gy)

    def close(a, b, what):
        err = float((a - b).detach().abs().max())
        # 2e-5 absolute floor: conv1.bias feeds a batch-norm, its exact gradient is 0 and both sides hold rounding noise
        assert err <= 2e-4 * float(a.detach().abs().max()) + 2e-5, (what, err, float(a.detach().abs().max()))

    close(ya, yb, 'output')
    close(xa.grad, xb.grad, 'input gradient')
    for (n, pa), (_, pb) in zip(blk.named_parameters(), blk2.named_parameters()):
        close(pa.grad, pb.grad, n)
    for (n, ba), (_, bb) in zip(blk.named_buffers(), blk2.named_buffers()):
        close(ba.float(), bb.float(), n)


@gpu
def test_untiled_condition_head_equals_explicit_concat():
    """Discriminator.cond[1] on (features | sentence vector tiled over the 5 x T/4 map) computed from the vector
    (Discriminator._cond_conv) equals the convolution of the explicit concatenation (generator.py:312-315): logits,
    input gradient, every parameter gradient.  fp32 kernels."""
    from vcagan_amd import ops
    from vcagan_amd.models import generator as G
    ops.set_precision('fp32')
    torch.manual_seed(12)
    dis = G.Discriminator(phase='1').cuda()
    mel = torch.randn(3, 1, 20, 24, device='cuda')
    c = torch.randn(3, 512, 24, device='cuda')
    res = {}
    for tiled in (False, True):
        G.TILED_INPUT = tiled
        try:
            dis.zero_grad()
            m = mel.clone().requires_grad_(True)
            u, o = dis(m, c, 24)
            (u.sum() + (o * torch.arange(1, 4, device='cuda').view(3, 1)).sum()).backward()
            res[tiled] = [u.detach(), o.detach(), m.grad] + [p.grad.clone() for p in dis.parameters()]
        finally:
            G.TILED_INPUT = True
    for i, (a, b) in enumerate(zip(res[False], res[True])):
        err = float((a - b).abs().max())
        assert err <= 2e-4 * float(a.abs().max()) + 1e-6, (i, err, float(a.abs().max()))


@gpu
def test_subpixel_upsample_block_equals_upsampled_convolution():
    """GenResBlk(upsample=True) in sub-pixel form (four tap-merged 3x3 convolutions of the source map) equals the
    5x5 convolution of the nearest x2 upsampled map (generator.py:112-126): output, running statistics, input
    gradient and every parameter gradient.  fp32 kernels."""
    import copy
    from vcagan_amd import ops
    from vcagan_amd.models import generator as G
    ops.set_precision('fp32')
    torch.manual_seed(13)
    blk = G.GenResBlk(32, 16, upsample=True).cuda().train()
    with torch.no_grad():
        for prm in blk.parameters():
            prm.add_(torch.randn_like(prm) * 0.05)
    blk2 = copy.deepcopy(blk)
    x0 = torch.randn(2, 6, 9, 32, device='cuda')
    gy = torch.randn(2, 12, 18, 16, device='cuda')
    res = []
    for b, sub in ((blk, False), (blk2, True)):
        G.SUBPIXEL_UPSAMPLE = sub
        try:
            x = x0.clone().requires_grad_(True)
            y = b(x)
            y.backward(gy)
        finally:
            G.SUBPIXEL_UPSAMPLE = True
        res.append([y.detach(), x.grad] + [p.grad for p in b.parameters()] + [t.float() for t in b.buffers()])
    names = ['output', 'input gradient'] + [n for n, _ in blk.named_parameters()] + [n for n, _ in blk.named_buffers()]
    for n, u, v in zip(names, res[0], res[1]):
        err = float((u - v).abs().max())
        assert err <= 2e-4 * float(u.abs().max()) + 2e-5, (n, err, float(u.abs().max()))
