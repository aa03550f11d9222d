"""The IEEE-half build of the library (libvca_hip_f16.so: the same kernels with half instead of bf16 matrix operands)
and the dynamic loss scale its gradients need (BASELINE configs[4]: "fp16 with loss-scaling").  The reference trains in
fp32 only, so there is no reference half-precision output to compare with: the half modes are held to the same fixtures
and bounds as their bf16 counterparts (parity against a half-precision reference: unpinned)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures
from parity_util import check_d_side, check_step, load, measure_step

gpu = pytest.mark.gpu
D = ('dis1', 'dis2', 'dis3', 's_dis')


@pytest.fixture(autouse=True)
def _reset_precision():
    yield
    from vcagan_amd import ops
    ops.set_precision('bf16x3')       # back to the bf16 build for whoever runs next
    ops.set_precision('fp32')


def rel_err(a, b):
    return (a.detach().cpu().double() - b.detach().double()).abs().max().item() / (b.detach().abs().max().item() + 1e-12)


CASES = [(2, 9, 11, 32, 48, 5, 1, 2), (3, 12, 12, 64, 128, 3, 2, 1), (2, 5, 18, 48, 24, 5, 1, 0),
         (2, 20, 75, 64, 128, 5, 1, 2), (5, 24, 24, 64, 64, 3, 1, 1), (200, 1, 1, 512, 96, 1, 1, 0)]


@gpu
@pytest.mark.parametrize('prec,tol', [('fp16x3', 1e-4), ('fp16', 3e-3)])
@pytest.mark.parametrize('N,H,W,Cin,Cout,K,stride,pad', CASES)
def test_conv_on_half_operands(N, H, W, Cin, Cout, K, stride, pad, prec, tol):
    from vcagan_amd import _lib, ops
    ops.set_precision(prec)
    assert _lib.ELEM[0] == 'fp16' and _lib.lib().vca_elem_type() == 1 and ops.get_precision() == prec
    g = torch.Generator().manual_seed(N * 7 + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride, pad)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    wd, bd = torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda())
    y = ops.conv2d(xd, wd, bd, stride, pad)
    assert rel_err(y.permute(0, 3, 1, 2), yr) < tol
    y.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
    assert rel_err(xd.grad.permute(0, 3, 1, 2), xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol


def _step(prec, variant='grid', scale=None, **kw):
    from vcagan_amd import ops, train_step
    ops.set_precision(prec)
    g = load('step_' + variant)
    m = fixtures.fill_models(train_step.build_models(), int(g['wseed']))
    for k in m:
        m[k].cuda().train()
    opts = train_step.build_optimizers(m, amsgrad=(variant == 'grid'))
    step = train_step.GanStep(m, opts, variant, **kw)
    assert step.scaler is not None            # created by default in the half modes
    if scale is not None:
        step.scaler.set_scale(scale)
    b = fixtures.make_batch(int(g['B']), int(g['T']), int(g['H']), int(g['seed']), list(g['lens']))
    b = {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}
    return g, m, opts, step, b


@gpu
@pytest.mark.parametrize('variant', ['grid', 'lrs'])
def test_half_step_matches_reference(variant):
    """'fp16-mixed' against the reference's fp32 step fixture, held to the bounds of 'mixed': the forward pass is the
    three-product split (half hi/lo), the gradients are single half products under the loss scale, which the
    optimizers divide out again."""
    g, m, opts, step, b = _step('fp16-mixed', variant)
    s = step.scaler.value
    assert s == 65536.0
    step.d_phase(b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'])
    for o in opts:                       # the flat buffers hold scale x gradient
        o.all_grads.mul_(1.0 / s)
    mt = check_d_side(m, step._st['phon_d'].grad / s, g, 0, 3e-2)
    print('fp16-mixed', variant, 'D side', {k: round(v, 4) for k, v in mt.items()})
    for o in opts:
        o.all_grads.mul_(s)
    step.g_phase()
    res = step.finish()
    assert step.scaler.skipped == 0 and step.scaler.value == s
    for o in opts:
        o.all_grads.mul_(1.0 / s)
    res = {k: (v.item() if v.numel() == 1 else v.cpu()) for k, v in res.items()}
    res['grad_p'] = res['grad_p'].numpy()
    check_step(m, res, g, 0, loss_rtol=5e-4, out_atol=3e-3, grad_rel=0.15, param_p90=2.1e-4, skip_grads=D)
    print('fp16-mixed', variant, {k: float('%.3g' % v) for k, v in measure_step(m, res, g, 0, D).items()})


@gpu
def test_gradients_do_not_depend_on_the_scale():
    """Scaling by a power of two is exact: with two scales that both keep the gradients inside half's normal range
    the unscaled gradients agree to rounding noise; and Adam's update, which divides the scale out, is the same."""
    got = {}
    for s in (1024.0, 65536.0):
        g, m, opts, step, b = _step('fp16-mixed', scale=s)
        step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
        got[s] = ([o.all_grads.clone() / s for o in opts],
                  torch.cat([p.detach().reshape(-1) for k in sorted(m) for p in m[k].parameters()]))
    for a, c in zip(got[1024.0][0], got[65536.0][0]):
        assert float((a - c).norm() / c.norm()) < 2e-2       # run-to-run noise of this network is 1e-2 (atomics order)
    assert float((got[1024.0][1] - got[65536.0][1]).abs().max()) <= 2.1e-4   # one Adam step moves a weight by <= lr


@gpu
def test_overflow_skips_the_step_and_halves_the_scale():
    g, m, opts, step, b = _step('fp16-mixed', scale=2.0 ** 60)
    before = [o.groups[0][0].clone() for o in opts]
    t0 = [int(o.t_dev) for o in opts]
    res = step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
    assert np.isfinite(float(res['gen_loss'])) and np.isfinite(float(res['dis_loss']))   # losses are unscaled
    assert step.scaler.skipped == 1 and step.scaler.value == 2.0 ** 59
    assert int(step.scaler.state[1]) == 0 and int(step.scaler.state[2]) == 0                # flags cleared
    for o, p0, t in zip(opts, before, t0):
        assert torch.equal(o.groups[0][0], p0) and int(o.t_dev) == t                        # nothing moved
        assert all(float(st['m'].abs().max()) == 0.0 for st in o.state)
    # a sane scale again: the next step goes through
    step.scaler.set_scale(65536.0)
    step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
    assert step.scaler.skipped == 1 and all(int(o.t_dev) == t + 1 for o, t in zip(opts, t0))
    assert not torch.equal(opts[0].groups[0][0], before[0])


@gpu
def test_scale_adapts_inside_a_captured_step():
    """The scaler lives in device words: a replayed graph grows the scale after `interval` clean steps and every
    replay uses the scale of the moment."""
    from vcagan_amd import ops, train_step
    g, m, opts, step, b = _step('fp16', scaler=None)
    step.scaler.interval = 1
    step.scaler.set_scale(256.0)
    gs = train_step.GraphedStep(step, b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'], warmup=1)
    s0 = step.scaler.value                                   # after the warm-up step and the capture's own step
    for i in range(3):
        res = gs(b['mel'], b['spec'], b['vid'], None, b['noise'])
        assert np.isfinite(float(res['gen_loss']))
    torch.cuda.synchronize()
    assert step.scaler.value == s0 * 8 and step.scaler.skipped == 0
    for k in m:
        assert all(torch.isfinite(p).all() for p in m[k].parameters()), k
