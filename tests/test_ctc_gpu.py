"""CTC loss kernels (through the C ABI) against torch.nn.functional.ctc_loss on the CPU: per-sample losses and the
gradient with respect to the logits.  The reference has no CTC training code (its ASR scorer is inference only,
ASR_model/GRID/test.py), so torch's operator is the stated semantics; it is evaluated in float64, because a log-space
recursion in fp32 (torch's own float path) drifts by about |log alpha| * 6e-8 per frame: with log-likelihoods of a
few hundred over 150-300 frames that is a few 1e-4 in the gradients.  The kernels run their recursions in fp64 too.
Tolerance: 2e-5 relative on the losses; 1e-5 of the largest gradient entry on the gradients."""
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu


def _ref(logits, targets, in_len, tgt_len, zero_infinity):
    lg = logits.double().requires_grad_(True)
    lp = F.log_softmax(lg, 2).transpose(0, 1)                     # (T, B, V)
    nll = F.ctc_loss(lp, targets, in_len, tgt_len, blank=0, reduction='none', zero_infinity=zero_infinity)
    return lg, nll


def _case(B, T, V, Lmax, seed, repeats=False):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, T, V, generator=g) * 2.0
    tgt_len = torch.randint(1, Lmax + 1, (B,), generator=g)
    in_len = torch.randint(max(2 * Lmax + 1, T // 2), T + 1, (B,), generator=g)
    targets = torch.randint(1, V, (B, Lmax), generator=g)
    if repeats:                                                    # doubled letters need a blank between them
        targets[:, 1::2] = targets[:, 0:-1:2] if Lmax % 2 == 0 else targets[:, 1::2]
    in_len[0], tgt_len[0] = T, Lmax                                # the full-size corner
    return logits, targets, in_len, tgt_len


@gpu
@pytest.mark.parametrize('B,T,V,Lmax,repeats', [(4, 38, 28, 12, False), (32, 150, 28, 40, True), (3, 75, 28, 30, True),
                                                (2, 40, 50, 6, False), (5, 300, 28, 140, False)])
def test_ctc_matches_torch(B, T, V, Lmax, repeats):
    from vcagan_amd import ops
    logits, targets, in_len, tgt_len = _case(B, T, V, Lmax, 7 + T, repeats)
    w = torch.rand(B, generator=torch.Generator().manual_seed(3)) + 0.5
    lg, want = _ref(logits, targets, in_len, tgt_len, True)
    (want * w.double()).sum().backward()
    x = logits.cuda().requires_grad_(True)
    got = ops.ctc_nll(x, targets.cuda(), in_len.cuda(), tgt_len.cuda())
    (got * w.cuda()).sum().backward()
    assert torch.isfinite(want).all()
    assert ((got.cpu().double() - want).abs() / want.abs().clamp(min=1.0)).max().item() < 2e-5
    gmax = lg.grad.abs().max().item()
    dev = (x.grad.cpu().double() - lg.grad).abs().max().item()
    print('ctc gradient deviation %.2e (largest entry %.2f)' % (dev, gmax))
    assert dev < 1e-5 * max(gmax, 1.0)
    # frames behind a sample's length take no gradient
    for b in range(B):
        assert x.grad[b, int(in_len[b]):].abs().max().item() == 0.0 if int(in_len[b]) < T else True
    # the class sums use a fixed order: a second run gives the same bits
    x2 = logits.cuda().requires_grad_(True)
    got2 = ops.ctc_nll(x2, targets.cuda(), in_len.cuda(), tgt_len.cuda())
    (got2 * w.cuda()).sum().backward()
    assert torch.equal(got, got2) and torch.equal(x.grad, x2.grad)


@gpu
def test_ctc_edge_cases():
    """Empty target, a target that does not fit its frames (zero_infinity on and off), zero frames."""
    from vcagan_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, V, Lmax = 4, 10, 28, 8
    logits = torch.randn(B, T, V, generator=g)
    targets = torch.randint(1, V, (B, Lmax), generator=g)
    targets[1, :] = 5                                   # 8 equal letters need 15 frames: impossible in 10
    in_len = torch.tensor([10, 10, 6, 10])
    tgt_len = torch.tensor([0, 8, 3, 5])
    for zi in (True, False):
        lg, want = _ref(logits, targets, in_len, tgt_len, zi)
        want.sum().backward()
        x = logits.cuda().requires_grad_(True)
        got = ops.ctc_nll(x, targets.cuda(), in_len.cuda(), tgt_len.cuda(), zero_infinity=zi)
        got.sum().backward()
        gc, xg = got.cpu(), x.grad.cpu()
        if zi:
            assert (gc.double() - want).abs().max().item() < 2e-5 and gc[1].item() == 0.0
            assert (xg.double() - lg.grad).abs().max().item() < 2e-5
            assert xg[1].abs().max().item() == 0.0
        else:
            assert torch.isinf(gc[1]) and torch.isinf(want[1])
            keep = [0, 2, 3]
            assert (gc[keep].double() - want[keep]).abs().max().item() < 2e-5
            assert (xg[keep].double() - lg.grad[keep]).abs().max().item() < 2e-5
            assert torch.isnan(xg[1, :10]).all()
    # a sample without frames: loss 0 for the empty target, gradient 0
    x = logits.cuda().requires_grad_(True)
    got = ops.ctc_nll(x, targets.cuda(), torch.tensor([0, 10, 6, 10]).cuda(), tgt_len.cuda())
    got.sum().backward()
    assert got[0].item() == 0.0 and x.grad[0].abs().max().item() == 0.0


@gpu
# bounds as in tests/test_configs_gpu.py; the split-bf16 loss bound is wider here (measured 9.3e-4 on a generator-side
# loss, which is taken after the discriminators' first Adam update and amplifies last-bit differences of the D phase)
@pytest.mark.parametrize('prec,atol,lrt,grel', [('fp32', 4e-4, 2e-4, 5e-2), ('mixed', 3e-3, 2e-3, 0.15)])
def test_lrs_step_with_ctc_auxiliary_loss(prec, atol, lrt, grel):
    """The multi-task configuration (BASELINE configs[4]: train_LRS.py step + CTC loss of the frozen ASR scorer on the
    generated mel) against the oracle's statement of the same step (oracle/cpu_ref.gan_step(ctc=...), built on
    torch's ctc_loss and the oracle's ASR restatement): every loss, the outputs and the generator-side gradients,
    which now carry the CTC term back through the scorer.  Also replayed from a captured graph."""
    from oracle import asr_ref, cpu_ref, fixtures
    from test_configs_gpu import _check_step, _dev, _pair
    from vcagan_amd import ops, train_step
    from vcagan_amd.models import asr
    ops.set_precision(prec)
    try:
        ref, hip = _pair(910)
        B, T, H = 2, 24, 112
        b = fixtures.make_batch(B, T, H, 53, [24, 17])
        g = torch.Generator().manual_seed(77)
        b['text'] = torch.randint(1, 28, (B, 9), generator=g)
        b['text_len'] = torch.tensor([9, 6])
        scorers = []
        for mod in (asr_ref, asr):
            front, back = mod.Audio_front(), mod.Backend()
            fixtures.det_fill(front, 21)
            fixtures.det_fill(back, 22)
            scorers.append(dict(front=front.eval(), back=back.eval(), weight=0.5))
        scorers[1]['front'].cuda(), scorers[1]['back'].cuda()
        want = cpu_ref.gan_step(ref, cpu_ref.build_optimizers(ref, amsgrad=False), b, variant='lrs', noise=b['noise'],
                                ctc=scorers[0])
        step = train_step.GanStep(hip, train_step.build_optimizers(hip, amsgrad=False), 'lrs', ctc=scorers[1])
        d = _dev(b)
        got = step(d['mel'], d['spec'], d['vid'], d['vid_len'], noise=d['noise'], text=d['text'],
                   text_len=d['text_len'])
        _check_step(ref, hip, want, got, lrt, atol, grel)
        assert want['ctc_loss'] > 0.5            # the term is not negligible in gen_loss
        assert abs(float(got['ctc_loss']) - want['ctc_loss']) <= max(lrt, 1e-4) * max(1.0, want['ctc_loss'])
        # the same configuration under graph replay: finite, and close to the eager losses of the following step
        gs = train_step.GraphedStep(step, d['mel'], d['spec'], d['vid'], d['vid_len'], d['noise'], warmup=1,
                                    text=d['text'], text_len=d['text_len'])
        res = gs(d['mel'], d['spec'], d['vid'], None, d['noise'], text=d['text'], text_len=d['text_len'])
        assert all(bool(torch.isfinite(res[k])) for k in ('gen_loss', 'dis_loss', 'ctc_loss'))
    finally:
        ops.set_precision('fp32')
