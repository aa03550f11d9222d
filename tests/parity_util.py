"""Shared comparison helpers for the golden fixtures written by oracle/make_goldens.py."""
import os

import numpy as np
import torch

from oracle import fixtures

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
LOSS_KEYS = ('dis_loss', 'gen_loss', 'sync_loss', 'g_sync_loss', 'recon_loss', 'g_loss')


def load(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def to_np(a):
    return a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)


def _sampled(a, g):
    """a as the fixture stores tensors of its size: whole, or every out_stride-th value (benchmark-sized fixture)."""
    stride = int(g['out_stride']) if 'out_stride' in g.files else 0
    a = to_np(a)
    return a.reshape(-1)[::stride] if stride else a


def close(a, b, atol=1e-5, rtol=1e-5):
    np.testing.assert_allclose(to_np(a), b, atol=atol, rtol=rtol)


def check_step(models, res, g, s, loss_rtol, out_atol, grad_rel, param_p90, skip_grads=()):
    """Compare one GAN step with the golden.

    Gradients of this network are discontinuous in the activations (LeakyReLU/PReLU kinks, max-pool argmax, L1
    sign): a 1e-6 relative change of the input video moves individual weight gradients by ~0.5 % at step 0 and
    ~10 % at step 1 (measured on the CPU oracle against itself), so gradients are compared as a relative L2 error
    over each module's sampled entries, and parameters after Adam - whose first steps move every weight by
    lr*sign(grad) - by the distribution of absolute differences.

    skip_grads: modules whose .grad is not compared.  After a step the reference's discriminators hold the
    gradient of gen_loss, which it never uses (d_optimizer.zero_grad() runs before the next D backward,
    train.py:209); the HIP step does not compute it.  Their parameters after Adam are still compared.
    """
    mt = measure_step(models, res, g, s, skip_grads)
    for k in LOSS_KEYS:
        assert mt['loss:' + k] <= loss_rtol, (s, k, mt['loss:' + k])
    np.testing.assert_allclose(res['grad_p'], g['s%d:grad_p' % s], rtol=max(loss_rtol * 5, 1e-3))
    for k in [k[8:] for k in mt if k.startswith('out_max:')]:
        assert mt['out_max:' + k] <= out_atol, (s, k, mt['out_max:' + k])
        assert mt['out_l1:' + k] <= max(out_atol / 10, 3e-5), (s, k, mt['out_l1:' + k])    # mel-L1
    for mod in sorted(models):
        if mod in skip_grads:
            continue
        assert mt['grad_rel:' + mod] <= grad_rel, (s, mod, mt['grad_rel:' + mod])
        assert mt['gradnorm_rel:' + mod] <= grad_rel + 1e-6, (s, mod, mt['gradnorm_rel:' + mod])
    assert mt['param_max'] <= 2.2e-4 * (s + 1)
    assert mt['param_p90'] <= param_p90, mt['param_p90']


def measure_step(models, res, g, s, skip_grads=()):
    """The deviations check_step bounds, as numbers: relative loss errors, max / mean output errors, per-module
    relative L2 error of the sampled gradients and of the gradient norm, and the parameter differences after Adam."""
    mt = {}
    for k in LOSS_KEYS:
        want = float(g['s%d:%s' % (s, k)])
        mt['loss:' + k] = abs(res[k] - want) / max(1.0, abs(want))
    for k in ('g1', 'g2', 'g3') + (('gs',) if 's%d:gs' % s in g.files else ()):
        d = np.abs(_sampled(res[k], g) - g['s%d:%s' % (s, k)])
        mt['out_max:' + k], mt['out_l1:' + k] = float(d.max()), float(d.mean())
    pdiff = []
    for mod in sorted(models):
        got, want, gn = [], [], 0.0
        for pname, p in models[mod].named_parameters():
            idx = fixtures.sample_idx(p.numel(), 8, mod + '.' + pname)
            pdiff.append(np.abs(to_np(p).reshape(-1)[idx] - g['s%d:param:%s:%s' % (s, mod, pname)]))
            if p.grad is not None and mod not in skip_grads:
                gn += float(p.grad.double().pow(2).sum())
                got.append(to_np(p.grad).reshape(-1)[idx])
                want.append(g['s%d:grad:%s:%s' % (s, mod, pname)])
        if mod in skip_grads:
            continue
        got, want = np.concatenate(got), np.concatenate(want)
        mt['grad_rel:' + mod] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        wn = float(g['s%d:gradnorm:%s' % (s, mod)])
        mt['gradnorm_rel:' + mod] = abs(gn ** 0.5 - wn) / wn
    pdiff = np.concatenate(pdiff)
    mt['param_max'], mt['param_p90'] = float(pdiff.max()), float(np.percentile(pdiff, 90))
    return mt


D_NAMES = ('dis1', 'dis2', 'dis3', 's_dis')


def measure_d_side(models, dphon, g, s, names=D_NAMES):
    """Deviations of the gradients d_optimizer.step() consumes (read between dis_loss.backward() and the optimizer
    step, train.py:210-211) from the fixture's 'dgrad' keys: per network the relative L2 error over the sampled
    entries and of the gradient norm, plus the relative L2 error of d dis_loss / d phon (the sync loss's way into
    v_front).  The discriminators' gradients carry the R1 penalty's double backward."""
    mt = {}
    for mod in names:
        got, want, gn = [], [], 0.0
        for pname, p in models[mod].named_parameters():
            key = 's%d:dgrad:%s:%s' % (s, mod, pname)
            if key not in g.files:
                continue
            assert p.grad is not None, (mod, pname)
            idx = fixtures.sample_idx(p.numel(), 8, mod + '.' + pname)
            got.append(to_np(p.grad).reshape(-1)[idx])
            want.append(g[key])
            gn += float(p.grad.double().pow(2).sum())
        got, want = np.concatenate(got), np.concatenate(want)
        mt['dgrad_rel:' + mod] = float(np.linalg.norm(got - want) / np.linalg.norm(want))
        wn = float(g['s%d:dgradnorm:%s' % (s, mod)])
        mt['dgradnorm_rel:' + mod] = abs(gn ** 0.5 - wn) / wn
    if dphon is not None:
        want = g['s%d:dphon' % s]
        mt['dphon_rel'] = float(np.linalg.norm(_sampled(dphon, g) - want) / np.linalg.norm(want))
    return mt


def check_d_side(models, dphon, g, s, grad_rel, names=D_NAMES):
    mt = measure_d_side(models, dphon, g, s, names)
    for k, v in mt.items():
        assert v <= grad_rel, (s, k, v, mt)
    return mt


TRAJ_KEYS = ('dis_loss', 'gen_loss', 'recon_loss', 'sync_loss')


def traj_deviation(run_step, g, nsteps=None, to_device=None):
    """Drive run_step(batch) -> result dict over the trajectory fixture's batches.  Returns, per step, a dict
    key -> (deviation from the reference's run a, envelope): relative loss deviations for TRAJ_KEYS and the mel-L1 of
    g3 under 'g3_l1'.  The envelope of a quantity at step s is the largest distance between the reference's run a and
    any of its runs b_i (clips 1e-6 apart) over steps 0..s: what the network's own sensitivity does to last-bit
    differences."""
    B, T, H = int(g['B']), int(g['T']), int(g['H'])
    nper = int(g['nper'])
    env = {k: 0.0 for k in TRAJ_KEYS + ('g3_l1',)}
    out = []
    for s in range(int(g['nsteps']) if nsteps is None else nsteps):
        b = fixtures.make_batch(B, T, H, int(g['seed']) + s, list(g['lens']))
        if to_device is not None:
            b = to_device(b)
        res = run_step(b)
        d = {}
        for k in TRAJ_KEYS:
            a = float(g['a:%d:%s' % (s, k)])
            for i in range(nper):
                env[k] = max(env[k], abs(a - float(g['b%d:%d:%s' % (i, s, k)])) / max(1.0, abs(a)))
            d[k] = (abs(float(res[k]) - a) / max(1.0, abs(a)), env[k])
        env['g3_l1'] = max([env['g3_l1']] + [float(g['b%d:%d:g3_l1' % (i, s)]) for i in range(nper)])
        d['g3_l1'] = (float(np.abs(to_np(res['g3']) - g['a:%d:g3' % s]).mean()), env['g3_l1'])
        out.append(d)
    return out
