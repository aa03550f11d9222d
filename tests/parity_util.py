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
    for k in LOSS_KEYS:
        want = float(g['s%d:%s' % (s, k)])
        assert abs(res[k] - want) <= loss_rtol * max(1.0, abs(want)), (s, k, res[k], want)
    np.testing.assert_allclose(res['grad_p'], g['s%d:grad_p' % s], rtol=max(loss_rtol * 5, 1e-3))
    for k in ('g1', 'g2', 'g3'):
        close(res[k], g['s%d:%s' % (s, k)], atol=out_atol, rtol=0)
        assert np.abs(to_np(res[k]) - g['s%d:%s' % (s, k)]).mean() <= max(out_atol / 10, 3e-5), (s, k)    # mel-L1
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
        rel = np.linalg.norm(got - want) / np.linalg.norm(want)
        assert rel <= grad_rel, (s, mod, rel)
        wn = float(g['s%d:gradnorm:%s' % (s, mod)])
        assert abs(gn ** 0.5 - wn) <= grad_rel * wn + 1e-6, (s, mod, gn ** 0.5, wn)
    pdiff = np.concatenate(pdiff)
    assert pdiff.max() <= 2.2e-4 * (s + 1)
    assert np.percentile(pdiff, 90) <= param_p90, np.percentile(pdiff, 90)
