"""Per-channel / pooling / loss / Adam HIP kernels (through the C ABI) vs torch CPU operators.  fp32; tolerance 1e-5
relative to the tensor's max magnitude unless noted."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

gpu = pytest.mark.gpu


def rel_err(a, b):
    return (a.detach().cpu().double() - b.detach().double()).abs().max().item() / (b.detach().abs().max().item() + 1e-12)


@gpu
@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('act', ['prelu', 'leaky', 'relu', None])
@pytest.mark.parametrize('with_res', [False, True])
def test_bn_act(train, act, with_res):
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(1)
    N, H, W, C = 3, 7, 5, 64
    x = torch.randn(N, H, W, C, generator=g) * 2 + 0.5
    res = torch.randn(N, H, W, C, generator=g) if with_res else None
    gy = torch.randn(N, H, W, C, generator=g)

    def mk():
        bn = nn.BatchNorm2d(C)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(2)) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(3)))
            bn.running_mean.copy_(torch.randn(C, generator=torch.Generator().manual_seed(4)) * 0.1)
            bn.running_var.copy_(torch.rand(C, generator=torch.Generator().manual_seed(5)) + 0.5)
        a = None
        if act == 'prelu':
            a = nn.PReLU(C)
            with torch.no_grad():
                a.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(6)) * 0.5)
        elif act == 'relu':
            a = nn.ReLU()
        return bn.train(train), a

    bn_r, a_r = mk()
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    rr = res.permute(0, 3, 1, 2).clone().requires_grad_(True) if with_res else None
    yr = bn_r(xr)
    if with_res:
        yr = yr + rr
    yr = F.leaky_relu(yr, 0.2) if act == 'leaky' else (a_r(yr) if a_r is not None else yr)
    yr.backward(gy.permute(0, 3, 1, 2))

    bn_d, a_d = mk()
    bn_d.cuda()
    if isinstance(a_d, nn.PReLU):
        a_d.cuda()
    xd = x.cuda().requires_grad_(True)
    rd = res.cuda().requires_grad_(True) if with_res else None
    y = ops.bn_act(xd, bn_d, ops.LEAKY if act == 'leaky' else a_d, rd)
    y.backward(gy.cuda())
    assert rel_err(y, yr.permute(0, 2, 3, 1)) < 1e-5
    assert rel_err(xd.grad, xr.grad.permute(0, 2, 3, 1)) < 2e-5
    assert rel_err(bn_d.weight.grad, bn_r.weight.grad) < 2e-5
    assert rel_err(bn_d.bias.grad, bn_r.bias.grad) < 2e-5
    if with_res:
        assert rel_err(rd.grad, rr.grad.permute(0, 2, 3, 1)) < 1e-5
    if act == 'prelu':
        assert rel_err(a_d.weight.grad, a_r.weight.grad) < 2e-5
    assert rel_err(bn_d.running_mean, bn_r.running_mean) < 1e-5
    assert rel_err(bn_d.running_var, bn_r.running_var) < 1e-5
    assert int(bn_d.num_batches_tracked) == int(bn_r.num_batches_tracked)


@gpu
@pytest.mark.parametrize('act', ['prelu', 'leaky'])
def test_bn_act_skip_alias_and_in_place_parameter_gradients(act):
    """The training step's form of the same operator: x also feeds an identity shortcut taken from the alias output
    (its gradient is added by the backward kernel), the parameter gradients are added into existing .grad storage by
    the kernels (ops.direct_param_grad), num_batches_tracked is bumped when the step ends.  Run twice, so that the
    gradients accumulate, against torch CPU autograd doing the same."""
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(11)
    N, H, W, C = 4, 9, 6, 128
    x = torch.randn(N, H, W, C, generator=g) + 0.3
    gy = torch.randn(N, H, W, C, generator=g)

    def mk():
        bn = nn.BatchNorm2d(C)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(2)) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(3)))
        a = nn.PReLU(C) if act == 'prelu' else None
        if a is not None:
            with torch.no_grad():
                a.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(6)) * 0.5)
        return bn.train(), a

    bn_r, a_r = mk()
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    for _ in range(2):
        yr = bn_r(xr)
        yr = F.leaky_relu(yr, 0.2) if a_r is None else a_r(yr)
        ((yr * 0.7 + xr * 1.3) * gy.permute(0, 3, 1, 2)).sum().backward()

    bn_d, a_d = mk()
    bn_d.cuda()
    params = [bn_d.weight, bn_d.bias]
    if a_d is not None:
        a_d.cuda()
        params.append(a_d.weight)
    for p in params:
        p.grad = torch.zeros_like(p)
    xd = x.cuda().requires_grad_(True)
    dev = xd.device
    for _ in range(2):
        ops.step_begin(dev)
        y, xa = ops.bn_act(xd, bn_d, ops.LEAKY if a_d is None else a_d, skip=True)
        out = ((y * 0.7 + xa * 1.3) * gy.cuda()).sum()
        with ops.direct_param_grad():
            out.backward()
        ops.step_end(dev)
    assert rel_err(y, yr.permute(0, 2, 3, 1)) < 1e-5
    assert rel_err(xd.grad, xr.grad.permute(0, 2, 3, 1)) < 2e-5
    assert rel_err(bn_d.weight.grad, bn_r.weight.grad) < 2e-5
    assert rel_err(bn_d.bias.grad, bn_r.bias.grad) < 2e-5
    if a_d is not None:
        assert rel_err(a_d.weight.grad, a_r.weight.grad) < 2e-5
    assert int(bn_d.num_batches_tracked) == int(bn_r.num_batches_tracked) == 2


@gpu
def test_wide_and_1d_bn():
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(9)
    for shape in [(2, 20, 9, 640), (3, 50, 128), (1000, 1280)]:
        C = shape[-1]
        x = torch.randn(*shape, generator=g) + 1.0
        bn = nn.BatchNorm1d(C).train()
        ref = bn(x.reshape(-1, C)).reshape(shape)
        bn2 = nn.BatchNorm1d(C).train().cuda()
        y = ops.bn_act(x.cuda(), bn2, None)
        assert rel_err(y, ref) < 1e-5


@gpu
def test_act_double_backward_and_pools():
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 9, 6, 32, generator=g)
    w = torch.randn(2, 4, 3, 32, generator=g)

    def f(leaky, pool, xx):
        h = pool(leaky(xx))
        out = (h * h * w.to(h.device)).sum()
        gr, = torch.autograd.grad(out, xx, create_graph=True)
        return (gr ** 2).sum() + out

    xr = x.clone().requires_grad_(True)
    pr = f(lambda t: F.leaky_relu(t, 0.2),
           lambda t: F.avg_pool2d(t.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1), xr)
    pr.backward()
    xd = x.cuda().requires_grad_(True)
    pd = f(ops.leaky, ops.avgpool2, xd)
    pd.backward()
    assert abs(pd.item() - pr.item()) < 1e-5 * abs(pr.item())
    assert rel_err(xd.grad, xr.grad) < 1e-5
    # max pool 3x3 s2 p1
    x = torch.randn(3, 11, 12, 64, generator=g)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.cuda().requires_grad_(True)
    y = ops.maxpool3x3s2(xd)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    assert rel_err(y, yr.permute(0, 2, 3, 1)) == 0
    assert rel_err(xd.grad, xr.grad.permute(0, 2, 3, 1)) < 1e-6
    # residual merge and tanh on a single-channel map
    a, b = torch.randn(2, 5, 4, 32, generator=g), torch.randn(2, 5, 4, 32, generator=g)
    ad, bd = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.add_scale(ad, bd, 0.70710678)
    y.sum().backward()
    assert rel_err(y, (a + b) * 0.70710678) < 1e-6 and rel_err(ad.grad, torch.full_like(a, 0.70710678)) < 1e-6
    t = torch.randn(2, 20, 10, 1, generator=g)
    td = t.cuda().requires_grad_(True)
    y = ops.tanh(td)
    y.sum().backward()
    assert rel_err(y, torch.tanh(t)) < 1e-6 and rel_err(td.grad, 1 - torch.tanh(t) ** 2) < 1e-5


@gpu
def test_losses_and_adam():
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(32, 1, generator=g) * 3
    for sign in (1.0, -1.0):
        xr = x.clone().requires_grad_(True)
        lr_ = F.softplus(sign * xr).mean(); lr_.backward()
        xd = x.cuda().requires_grad_(True)
        ld = ops.softplus_mean(xd, sign); ld.backward()
        assert abs(ld.item() - lr_.item()) < 1e-6 and rel_err(xd.grad, xr.grad) < 1e-5
    a, b = torch.randn(2, 1, 80, 40, generator=g), torch.randn(2, 1, 80, 40, generator=g)
    ar = a.clone().requires_grad_(True)
    lr_ = F.l1_loss(ar * 5.7 + 1, b * 5.7 + 1); lr_.backward()
    ad = a.cuda().requires_grad_(True)
    ld = ops.l1_mean(ad, b.cuda(), 5.7); ld.backward()
    assert abs(ld.item() - lr_.item()) < 1e-5 * lr_.item() and rel_err(ad.grad, ar.grad) < 1e-5
    # fused Adam (amsgrad on/off) against torch.optim.Adam for 3 steps
    for ams in (True, False):
        lin_r = nn.Sequential(nn.Linear(37, 19), nn.Linear(19, 5))
        lin_d = nn.Sequential(nn.Linear(37, 19), nn.Linear(19, 5))
        lin_d.load_state_dict(lin_r.state_dict())
        lin_d.cuda()
        opt_r = torch.optim.Adam(lin_r.parameters(), lr=1e-2, weight_decay=1e-3, amsgrad=ams)
        flat = ops.flatten_module(lin_d)
        opt_d = ops.FusedAdam([flat], lr=1e-2, weight_decay=1e-3, amsgrad=ams)
        for it in range(3):
            xin = torch.randn(11, 37, generator=g)
            opt_r.zero_grad(); lin_r(xin).square().sum().backward(); opt_r.step()
            opt_d.zero_grad(); lin_d(xin.cuda()).square().sum().backward(); opt_d.step()
        for pr, pd in zip(lin_r.parameters(), lin_d.parameters()):
            assert rel_err(pd, pr) < 1e-5


@gpu
@pytest.mark.parametrize('prec,tol', [('fp32', 2e-5), ('bf16x3', 2e-4)])
def test_gru_matches_torch(prec, tol):
    from vcagan_amd import ops
    ops.set_precision(prec)
    try:
        torch.manual_seed(3)
        ref = nn.GRU(48, 32, 2, bidirectional=True)
        mod = nn.GRU(48, 32, 2, bidirectional=True)
        mod.load_state_dict(ref.state_dict())
        mod.cuda()
        x = torch.randn(5, 9, 48)             # (B, T, I)
        xr = x.clone().requires_grad_(True)
        yr, _ = ref(xr.transpose(0, 1))
        yr = yr.transpose(0, 1)
        gy = torch.randn(yr.shape)
        yr.backward(gy)
        xd = x.cuda().requires_grad_(True)
        y = ops.gru(xd, mod, False)
        y.backward(gy.cuda())
        assert rel_err(y, yr) < tol
        assert rel_err(xd.grad, xr.grad) < tol
        for (n, pr), (_, pd) in zip(ref.named_parameters(), mod.named_parameters()):
            assert rel_err(pd.grad, pr.grad) < tol * 2, n
    finally:
        ops.set_precision('fp32')


@gpu
@pytest.mark.parametrize('B,Tq,S,lens', [(2, 20, 20, [20, 13]), (3, 150, 75, [75, 40, 1]), (2, 300, 150, [150, 77])])
def test_attention_core(B, Tq, S, lens):
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(B, n, 256, generator=g) for n in (Tq, S, S))
    gy = torch.randn(B, Tq, 256, generator=g)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    att = torch.bmm(qr, kr.transpose(1, 2)) / 16.0
    for i, l in enumerate(lens):
        att[i, :, l:] = float('-inf')
    out_r = torch.bmm(att.softmax(2), vr)
    out_r.backward(gy)
    qd, kd, vd = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = ops.masked_attention(qd, kd, vd, ops.as_lens(lens, qd.device), 1.0 / 16.0)
    out.backward(gy.cuda())
    assert rel_err(out, out_r) < 1e-5
    assert rel_err(qd.grad, qr.grad) < 2e-5 and rel_err(kd.grad, kr.grad) < 2e-5 and rel_err(vd.grad, vr.grad) < 2e-5


@gpu
@pytest.mark.parametrize('B,T,C,temp', [(3, 20, 512, 1.0), (2, 75, 512, 0.5), (2, 150, 96, 2.0)])
def test_sync_loss_heads(B, T, C, temp):
    """vca_sync_nce_* / vca_sync_cos_* vs the torch operators of generator.py:347-361 (fp32, 2e-5 relative)."""
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(3)
    v0 = torch.randn(B, T, C, generator=g)
    a0 = torch.randn(B, T, C, generator=g) + 0.3 * v0     # correlated, so the diagonal stands out
    w = torch.randn(B, generator=g)

    def ref(gen):
        v, a = v0.clone().requires_grad_(True), a0.clone().requires_grad_(True)
        if gen:
            loss = 5.0 - F.cosine_similarity(v, a, 2).abs().mean(1)
        else:
            sim = torch.bmm(F.normalize(v, dim=2), F.normalize(a, dim=2).transpose(1, 2)) / temp
            d_va = torch.diagonal(F.log_softmax(sim, 2), dim1=-2, dim2=-1).mean(1)
            d_av = torch.diagonal(F.log_softmax(sim, 1), dim1=-2, dim2=-1).mean(1)
            loss = -0.5 * (d_va + d_av)
        (loss * w).sum().backward()
        return loss, v.grad, a.grad

    for gen in (False, True):
        rl, rv, ra = ref(gen)
        v, a = v0.cuda().requires_grad_(True), a0.cuda().requires_grad_(True)
        loss = ops.sync_loss(v, a, temp, gen)
        (loss * w.cuda()).sum().backward()
        assert rel_err(loss, rl) < 2e-5, (gen, rel_err(loss, rl))
        assert rel_err(v.grad, rv) < 5e-5, (gen, rel_err(v.grad, rv))
        assert rel_err(a.grad, ra) < 5e-5, (gen, rel_err(a.grad, ra))
    # only the audio side needs a gradient in the generator phase (train.py:221: v_feat is detached)
    a = a0.cuda().requires_grad_(True)
    ops.sync_loss(v0.cuda(), a, temp, True).sum().backward()
    assert a.grad is not None and torch.isfinite(a.grad).all()


@gpu
@pytest.mark.parametrize('train', [True, False])
def test_bn_act_maxpool3_fused_equals_separate_ops(train):
    """BatchNorm -> PReLU -> MaxPool(3x3, s2, p1) as one kernel pair (visual_front.py:12-14) vs the separate ops:
    output, running statistics, input gradient, gamma/beta/alpha gradients."""
    import copy
    from vcagan_amd import ops
    torch.manual_seed(21)
    bn = torch.nn.BatchNorm2d(16).cuda().train(train)
    pr = torch.nn.PReLU(16).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
        pr.weight.uniform_(0.1, 0.4)
    bn2, pr2 = copy.deepcopy(bn), copy.deepcopy(pr)
    x0 = torch.randn(3, 10, 14, 16, device='cuda')
    gy = torch.randn(3, 5, 7, 16, device='cuda')
    xa = x0.clone().requires_grad_(True)
    ya = ops.maxpool3x3s2(ops.bn_act(xa, bn, pr))
    ya.backward(gy)
    xb = x0.clone().requires_grad_(True)
    yb = ops.bn_act_maxpool3(xb, bn2, pr2)
    yb.backward(gy)
    assert torch.allclose(ya, yb, atol=1e-6)
    assert torch.allclose(xa.grad, xb.grad, atol=2e-6, rtol=1e-5)
    for u, v in ((bn.weight, bn2.weight), (bn.bias, bn2.bias), (pr.weight, pr2.weight)):
        assert torch.allclose(u.grad, v.grad, atol=1e-4, rtol=1e-5), float((u.grad - v.grad).abs().max())
    assert torch.allclose(bn.running_mean, bn2.running_mean) and torch.allclose(bn.running_var, bn2.running_var)


@gpu
@pytest.mark.parametrize('shape', [(2, 8, 10, 32), (3, 9, 7, 64), (2, 37, 10, 256), (1, 2, 3, 4)])
def test_leaky_avgpool2_fused(shape):
    """LeakyReLU + 2x2 average of one map in one pass (the entry of a down-sampling discriminator block,
    generator.py:71-83) against torch: both outputs, the joined input gradient, and the gradient of that gradient
    (the R1 penalty differentiates the blocks' backward pass), on even and odd map sizes."""
    import torch.nn.functional as F
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    ga, ggx = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    n, h, w, c = shape
    gp = torch.randn(n, h // 2, w // 2, c, generator=g)

    def run(fn, dev):
        xx = x.to(dev).requires_grad_(True)
        ya, yp = fn(xx)
        a, p = ga.to(dev).requires_grad_(True), gp.to(dev).requires_grad_(True)
        gx, = torch.autograd.grad([ya, yp], xx, [a, p], create_graph=True)
        da, dp = torch.autograd.grad(gx, [a, p], ggx.to(dev))
        gx_a, = torch.autograd.grad(ya, xx, a.detach(), retain_graph=True)       # one branch only
        gx_p, = torch.autograd.grad(yp, xx, p.detach())
        return [t.detach().cpu() for t in (ya, yp, gx, da, dp, gx_a, gx_p)]

    def ref(xx):
        return F.leaky_relu(xx, 0.2), F.avg_pool2d(xx.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)

    want = run(ref, 'cpu')
    got = run(ops.leaky_avgpool2, 'cuda')
    for name, a, b in zip(('act', 'pool', 'gx', 'd/dga', 'd/dgp', 'gx(act only)', 'gx(pool only)'), got, want):
        assert a.shape == b.shape, name
        assert float((a - b).abs().max()) < 1e-6, name
