"""HIP implicit-GEMM convolution (through the C ABI) vs torch's CPU convolution: forward, input gradient, weight
gradient, bias gradient and the grad-of-grad the R1 penalty needs.  fp32 tolerance: 2e-5 relative to the tensor's
max magnitude (the MFMA accumulates in a different order than the CPU kernels)."""
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu


def rel_err(a, b):
    return (a.detach().cpu().double() - b.detach().double()).abs().max().item() / (b.detach().abs().max().item() + 1e-12)


CASES = [
    # N, H, W, Cin, Cout, K, stride, pad, ups
    (2, 9, 11, 32, 48, 5, 1, 2, 0),
    (3, 12, 12, 64, 128, 3, 2, 1, 0),      # ResNet stride-2 block conv
    (3, 12, 12, 64, 128, 1, 2, 0, 0),      # 1x1 stride-2 shortcut
    (2, 5, 18, 48, 24, 5, 1, 0, 0),        # valid 5x5 head conv
    (2, 10, 7, 16, 32, 5, 1, 2, 1),        # conv over a nearest x2 upsample
    (2, 20, 13, 1, 32, 5, 1, 2, 0),        # single-channel input (discriminator stem)
    (2, 20, 14, 1, 128, 3, 2, 1, 0),       # sync discriminator stem
    (8, 80, 37, 1, 16, 5, 1, 2, 0),        # single-channel input: more output rows than partial-sum slabs
    (3, 21, 15, 1, 40, 3, 2, 1, 0),        # single-channel input, stride 2 on odd sizes, Cout not a multiple of 32
    (2, 9, 10, 1, 8, 3, 1, 0, 0),          # single-channel input, no padding
    (16, 130, 21, 1, 32, 5, 1, 2, 0),      # single-channel input: groups of 2 output rows
    (40, 70, 9, 1, 32, 5, 1, 2, 0),        # single-channel input: groups of 4 output rows with a tail of 2
    (24, 131, 20, 1, 24, 3, 2, 1, 0),      # single-channel input: groups of 2 output rows, stride 2, odd height
    (2, 8, 9, 32, 1, 1, 1, 0, 0),          # to_mel 1x1 to one channel
    (1, 6, 5, 80, 321, 1, 1, 0, 0),        # odd Cout
    (1, 7, 6, 321, 16, 1, 1, 0, 0),        # odd Cin
    (2, 6, 6, 256, 256, 3, 1, 1, 0),
    (2, 13, 9, 24, 40, 3, 2, 1, 0),        # stride 2 on odd map sizes (sub-pixel input gradient drops the last row/column)
]


@pytest.fixture(autouse=True)
def _reset_precision():
    yield
    try:
        from vcagan_amd import ops
        ops.set_precision('fp32')
    except Exception:
        pass


BIG_CASES = [
    (2, 20, 75, 64, 128, 5, 1, 2, 0),      # decoder-shaped map: 5x25 patches
    (5, 24, 24, 64, 64, 3, 1, 1, 0),       # ResNet layer1
    (30, 6, 6, 128, 256, 3, 1, 1, 0),      # small maps: several images per patch
    (40, 3, 3, 256, 512, 3, 1, 1, 0),
    (9, 12, 12, 64, 128, 3, 2, 1, 0),
    (2, 40, 30, 32, 32, 5, 1, 2, 1),
    (3, 5, 18, 160, 128, 5, 1, 0, 0),
    (200, 1, 1, 512, 96, 1, 1, 0, 0),      # linear layer
]


@gpu
@pytest.mark.parametrize('prec,tol', [('fp32', 2e-5), ('bf16x3', 1e-4), ('bf16', 3e-2)])
@pytest.mark.parametrize('case', CASES + BIG_CASES)
def test_conv2d_fwd_bwd(case, prec, tol):
    from vcagan_amd import ops
    ops.set_precision(prec)
    N, H, W, Cin, Cout, K, s, p, ups = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode='nearest') if ups else xr
    yr = F.conv2d(xin, wr, br, s, p)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xd, wd, bd, s, p, ups)
    assert y.shape == (N,) + tuple(yr.shape[2:]) + (Cout,)
    assert rel_err(y, yr.permute(0, 2, 3, 1)) < tol
    y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    assert rel_err(xd.grad, xr.grad.permute(0, 2, 3, 1)) < tol
    assert rel_err(wd.grad, wr.grad) < tol
    assert rel_err(bd.grad, br.grad) < 2e-5


@gpu
@pytest.mark.parametrize('case', [(2, 12, 10, 1, 32, 5, 1, 2), (2, 6, 9, 32, 64, 5, 1, 2), (2, 5, 8, 64, 64, 5, 1, 0)])
def test_conv2d_double_backward(case):
    """d/dw of ||d y.sum / d x||^2 (the R1 penalty structure, train.py:188-194)."""
    from vcagan_amd import ops
    N, H, W, Cin, Cout, K, s, p = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, H, W, Cin, generator=g)
    w1 = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    w2 = torch.randn(Cout, Cout, 3, 3, generator=g) / (Cout * 9) ** 0.5

    def penalty(conv, act, xx, a, b, nchw):
        h = act(conv(xx, a, None, s, p))
        out = conv(h, b, None, 1, 1)
        gr, = torch.autograd.grad(out.sum(), xx, create_graph=True)
        return (gr ** 2).sum()

    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    a, b = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    pr = penalty(F.conv2d, lambda t: F.leaky_relu(t, 0.2), xr, a, b, True)
    pr.backward()

    xd = x.cuda().requires_grad_(True)
    ad, bd = w1.cuda().requires_grad_(True), w2.cuda().requires_grad_(True)
    pd = penalty(lambda t, ww, bb, ss, pp: ops.conv2d(t, ww, bb, ss, pp), lambda t: F.leaky_relu(t, 0.2), xd, ad, bd, False)
    pd.backward()
    assert abs(pd.item() - pr.item()) < 1e-4 * abs(pr.item())
    assert rel_err(ad.grad, a.grad) < 1e-4
    assert rel_err(bd.grad, b.grad) < 1e-4


@gpu
def test_conv1d_linear_stem():
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(3)
    # conv1d (Postnet, generator.py:178)
    x = torch.randn(2, 37, 80, generator=g); w = torch.randn(128, 80, 7, generator=g) * 0.05; b = torch.randn(128, generator=g)
    y = ops.conv1d(x.cuda(), w.cuda(), b.cuda(), 3)
    assert rel_err(y, F.conv1d(x.transpose(1, 2), w, b, 1, 3).transpose(1, 2)) < 2e-5
    # linear
    x = torch.randn(3, 17, 512, generator=g); w = torch.randn(256, 512, generator=g) * 0.05; b = torch.randn(256, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.linear(xr, wr, b); yr.square().sum().backward()
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    y = ops.linear(xd, wd, b.cuda()); y.square().sum().backward()
    assert rel_err(y, yr) < 2e-5 and rel_err(xd.grad, xr.grad) < 2e-5 and rel_err(wd.grad, wr.grad) < 2e-5
    # 3-D stem (visual_front.py:11)
    x = torch.randn(2, 1, 6, 20, 24, generator=g); w = torch.randn(64, 1, 5, 7, 7, generator=g) * 0.05
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(x, wr, None, (1, 2, 2), (2, 3, 3))
    gy = torch.randn(yr.shape, generator=g); yr.backward(gy)
    wd = w.cuda().requires_grad_(True)
    y, _ = ops.conv3d_stem(x.permute(0, 2, 3, 4, 1).contiguous().cuda(), wd)
    assert rel_err(y, yr.permute(0, 2, 3, 4, 1)) < 2e-5
    y.backward(gy.permute(0, 2, 3, 4, 1).contiguous().cuda())
    assert rel_err(wd.grad, wr.grad) < 2e-5


@gpu
@pytest.mark.parametrize('prec,tol', [('bf16x3', 1e-4), ('bf16', 3e-2)])
# (1, 3, 21, 22): W % 4 != 0; (1, 17, 20, 24) and (2, 33, 40, 36): clips longer than one 15-frame chunk of a workgroup's walk
@pytest.mark.parametrize('shape', [(2, 6, 20, 24), (1, 3, 96, 96), (2, 2, 112, 104), (1, 3, 21, 22), (1, 17, 20, 24),
                                   (2, 33, 40, 36), (1, 16, 70, 66)])
def test_stem_kernel(shape, prec, tol):
    """Dedicated 5x7x7 stem kernel (visual_front.py:11) + its fused batch-norm statistics."""
    from vcagan_amd import ops
    ops.set_precision(prec)
    B, T, H, W = shape
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B, 1, T, H, W, generator=g)
    w = torch.randn(64, 1, 5, 7, 7, generator=g) * 0.06
    wr = w.clone().requires_grad_(True)
    yr = F.conv3d(x, wr, None, (1, 2, 2), (2, 3, 3))
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    wd = torch.nn.Parameter(w.cuda())
    y, stats = ops.conv3d_stem(x.reshape(B, T, H, W, 1).cuda(), wd)
    yl = yr.permute(0, 2, 3, 4, 1)
    assert y.shape == yl.shape
    assert rel_err(y, yl) < tol
    assert stats.dim() == 3                      # partial buffers, totalled by the batch-norm finalize
    stats = stats.sum(0)
    assert rel_err(stats[0], yl.reshape(-1, 64).sum(0)) < max(tol, 1e-4)
    assert rel_err(stats[1], yl.reshape(-1, 64).pow(2).sum(0)) < max(tol, 1e-4)
    y.backward(gy.permute(0, 2, 3, 4, 1).contiguous().cuda())
    # the weight gradient runs at the same matrix-core precision as the forward pass (patch form of the clip)
    assert rel_err(wd.grad, wr.grad) < (2e-5 if prec == 'bf16x3' else tol)


@gpu
@pytest.mark.parametrize('prec,tol', [('bf16x3', 1e-4), ('mixed', 1.5e-2)])
@pytest.mark.parametrize('case', [(3, 9, 11, 32, 48, 5, 2), (4, 12, 12, 64, 100, 3, 1), (2, 6, 7, 96, 33, 1, 0),
                                  (2, 20, 13, 64, 64, 5, 2),
                                  (3, 20, 31, 1, 32, 5, 2), (2, 9, 14, 1, 40, 3, 1)])    # single-channel input
def test_conv2d_gradients_into_sinks(case, prec, tol):
    """Parameters that own a preallocated gradient (the flat buffers of a training step) receive the weight gradient
    through the per-parameter workspace and the bias gradient from the weight-gradient kernel itself (column sums of
    the output-gradient tiles it stages), both ADDED to what the buffers already hold."""
    from vcagan_amd import ops
    ops.set_precision(prec)
    N, H, W, ci, co, k, pad = case
    g = torch.Generator(device='cpu').manual_seed(11)
    x = torch.randn(N, H, W, ci, generator=g)
    w = torch.randn(co, ci, k, k, generator=g) * 0.1
    b = torch.randn(co, generator=g)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, 1, pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    wd = torch.nn.Parameter(w.cuda())
    bd = torch.nn.Parameter(b.cuda())
    w0, b0 = torch.randn(w.shape, generator=g), torch.randn(co, generator=g)
    wd.grad, bd.grad = w0.cuda(), b0.cuda()
    xd = x.cuda().requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, 1, pad)
    with ops.direct_param_grad():
        y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    torch.cuda.synchronize()
    assert rel_err(bd.grad - b0.cuda(), br.grad) < max(tol / 50, 2e-5)      # fp32 sums whatever the product precision
    assert rel_err(wd.grad - w0.cuda(), wr.grad) < tol
    assert rel_err(xd.grad.permute(0, 3, 1, 2), xr.grad) < tol


@gpu
@pytest.mark.parametrize('prec,tol', [('bf16x3', 1e-4), ('mixed', 1.5e-2)])
@pytest.mark.parametrize('case', [(3, 12, 12, 64, 128), (2, 13, 9, 24, 40), (5, 7, 16, 136, 72), (70, 6, 6, 256, 512)])
def test_stride2_input_gradient_subpixel_form(case, prec, tol):
    """Input gradient of a 3x3 / stride 2 / padding 1 convolution (resnet.py:5-9) as a 2x2 convolution of the
    undilated output gradient with a depth-to-space epilogue, against torch and against the zero-dilated form,
    with and without a tensor added in the epilogue."""
    from vcagan_amd import ops
    ops.set_precision(prec)
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    cfg = ops.ConvCfg(w.shape, (N, H, W, Cin), 2, (1, 1), 0)
    assert cfg.subpixel_dgrad()
    gy = torch.randn(cfg.yshape, generator=g)
    res = torch.randn(N, H, W, Cin, generator=g)
    want = torch.nn.grad.conv2d_input((N, Cin, H, W), w, gy.permute(0, 3, 1, 2), 2, 1).permute(0, 2, 3, 1)
    wd, gyd, resd = w.cuda(), gy.cuda(), res.cuda()
    got = ops._conv_dgrad_raw(gyd, wd, cfg)
    got_res = ops._conv_dgrad_raw(gyd, wd, cfg, res=resd)
    assert rel_err(got, want) < tol
    assert rel_err(got_res, want + res) < tol
    ops.SUBPIXEL_DGRAD = False
    try:
        assert not cfg.subpixel_dgrad()
        dilated = ops._conv_dgrad_raw(gyd, wd, cfg)
    finally:
        ops.SUBPIXEL_DGRAD = True
    assert rel_err(got, dilated.cpu()) < (1e-5 if prec == 'bf16x3' else 1e-2)


@gpu
@pytest.mark.parametrize('ns', [3, 1])
@pytest.mark.parametrize('case', [(8, 5, 18, 256, 256, 5, 0), (8, 5, 18, 160, 136, 5, 2), (6, 10, 37, 256, 96, 3, 1)])
def test_split_reduction_workspace_and_atomics_agree(case, ns):
    """Few output tiles, deep reduction: the channel slices are split over workgroups.  The workspace form (partial
    sums in slabs + fold kernel) and the atomics form of the same launch must both match the CPU convolution; the
    workspace's previous contents must not matter and a too small workspace must fall back to atomics."""
    from vcagan_amd import _lib, ops
    from vcagan_amd._lib import ptr, stream, check
    N, H, W, Cin, Cout, K, p = case
    g = torch.Generator().manual_seed(sum(case) + ns)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    yr = F.conv2d(x.permute(0, 3, 1, 2), w, b, 1, p).permute(0, 2, 3, 1)
    ops.set_precision('bf16x3')
    cfg = ops.ConvCfg(w.shape, x.shape, 1, (p, p), 0)
    d = cfg.desc()
    L = _lib.lib()
    nws = int(L.vca_conv_fwd_bf16_ws_floats(d))
    assert nws > 0, 'the case is meant to split its reduction'
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    wp = ops._pack(wd, 10)
    tol = 1e-4 if ns == 3 else 3e-2
    outs = []
    for ws in (None, torch.full((nws,), float('nan'), device='cuda'), torch.full((nws // 2,), float('nan'), device='cuda')):
        y = torch.full(cfg.yshape, float('nan'), device='cuda')
        check(L.vca_conv_fwd_bf16_ws(d, ns, ptr(xd), wp.data_ptr(), ptr(bd), None, None, None, None, ptr(y), None,
                                     ptr(ws), 0 if ws is None else ws.numel(), stream()), 'conv_fwd_bf16_ws')
        torch.cuda.synchronize()
        assert rel_err(y, yr) < tol
        outs.append(y)
    # one summation order per output element in the workspace form: bit-identical from run to run
    y2 = torch.empty(cfg.yshape, device='cuda')
    ws = torch.zeros(nws, device='cuda')
    check(L.vca_conv_fwd_bf16_ws(d, ns, ptr(xd), wp.data_ptr(), ptr(bd), None, None, None, None, ptr(y2), None,
                                 ptr(ws), nws, stream()), 'conv_fwd_bf16_ws')
    assert torch.equal(y2, outs[1])
