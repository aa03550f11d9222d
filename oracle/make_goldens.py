"""Generate tests/golden/*.npz by running the REFERENCE's src/models on CPU.

Development-container script (needs /root/reference; never runs on the GPU box).
Only inputs' seeds and the reference's outputs are written - no reference source.
The reference modules are imported unmodified; two runtime shims make them
runnable without a GPU: Tensor.cuda becomes the identity, and while the
reference Decoder runs, torch.randn returns the fixture's noise tensor so the
draw at generator.py:248 is reproducible on any platform.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens.py
"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = os.environ.get('VCA_REFERENCE', '/root/reference')

from oracle import cpu_ref, fixtures  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def import_reference():
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from src.models import generator as rg
    from src.models import visual_front as rv
    return rg, rv


@contextlib.contextmanager
def fixed_randn(noise):
    orig = torch.randn

    def fake(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        assert shape == tuple(noise.shape), (shape, noise.shape)
        return noise.clone()
    torch.randn = fake
    try:
        yield
    finally:
        torch.randn = orig


class RefGen(torch.nn.Module):
    """Adapter giving the reference Decoder the oracle's noise= keyword."""

    def __init__(self, gen):
        super().__init__()
        self.inner = gen

    def forward(self, s, x, lens, noise=None):
        with fixed_randn(noise):
            return self.inner(s, x, lens)


def build_reference(rg, rv, seed):
    m = dict(v_front=rv.Visual_front(in_channels=1), gen=rg.Decoder(), post=rg.Postnet(),
             dis1=rg.Discriminator(phase='1'), dis2=rg.Discriminator(phase='2'),
             dis3=rg.Discriminator(phase='3'), s_dis=rg.sync_Discriminator(temp=1.0))
    return fixtures.fill_models(m, seed)


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def forward_case(rg, rv, name, train, B=2, T=20, H=112, lens=(20, 13), seed=11):
    torch.manual_seed(0)
    m = build_reference(rg, rv, 100)
    for k in m:
        m[k].train(train)
    b = fixtures.make_batch(B, T, H, seed, list(lens))
    out = {}
    with torch.no_grad():
        phon, sent = m['v_front'](b['vid'])
        with fixed_randn(b['noise']):
            g1, g2, g3 = m['gen'](sent, phon, b['vid_len'])
        gs = m['post'](g3)
        mel = b['mel']
        mel1 = torch.nn.functional.interpolate(mel, scale_factor=0.25, mode='bilinear')
        mel2 = torch.nn.functional.interpolate(mel, scale_factor=0.5, mode='bilinear')
        for i, (d, r, g) in enumerate(zip(('dis1', 'dis2', 'dis3'), (mel1, mel2, mel), (g1, g2, g3)), 1):
            u, c = m[d](r, sent, T)
            out['real_u%d' % i], out['real_c%d' % i] = np32(u), np32(c)
            u, c = m[d](g, sent, T)
            out['fake_u%d' % i], out['fake_c%d' % i] = np32(u), np32(c)
        out['sync_d'] = np32(m['s_dis'](phon, mel))
        out['sync_g'] = np32(m['s_dis'](phon, g3, True))
    out.update(phon=np32(phon), sent=np32(sent), g1=np32(g1), g2=np32(g2), g3=np32(g3), gs=np32(gs))
    if train:   # running statistics after one train-mode forward (a few representative BN layers)
        for mod, key in (('v_front', 'frontend.1.running_mean'), ('v_front', 'frontend.1.running_var'),
                         ('v_front', 'resnet.layer4.1.bn2.running_var'), ('gen', 'decode.0.norm1.running_mean'),
                         ('gen', 'g3.2.norm2.running_var'), ('post', 'postnet.1.running_var'),
                         ('s_dis', 'frontend.4.running_mean')):
            out['stat:%s:%s' % (mod, key)] = np32(m[mod].state_dict()[key])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), B=B, T=T, H=H, lens=np.array(lens), seed=seed,
                        wseed=100, **out)
    print(name, {k: v.shape for k, v in out.items() if not k.startswith('stat')})


def vfront_case(rg, rv, name, H, B=1, T=6, seed=21):
    m = build_reference(rg, rv, 100)
    vf = m['v_front'].eval()
    b = fixtures.make_batch(B, T, H, seed)
    with torch.no_grad():
        phon, sent = vf(b['vid'])
    np.savez_compressed(os.path.join(OUT, name + '.npz'), B=B, T=T, H=H, seed=seed, wseed=100,
                        phon=np32(phon), sent=np32(sent))
    print(name, phon.shape, sent.shape)


def step_case(rg, rv, name, variant, B=2, T=20, H=112, lens=(20, 13), nsteps=2, seed=31, out_stride=None):
    """out_stride: None stores the generated mels whole; an integer stores every out_stride-th value of g1 / g2 / g3
    and of the postnet's spectrogram ('gs') instead, for the benchmark-sized case whose outputs run to megabytes."""
    m = build_reference(rg, rv, 200)
    for k in m:
        m[k].train()
    amsgrad = variant == 'grid'
    opts = cpu_ref.build_optimizers(m, amsgrad=amsgrad)
    drive = dict(m)
    drive['gen'] = RefGen(m['gen'])          # same parameters; adapter only injects the noise
    out = {}
    for s in range(nsteps):
        b = fixtures.make_batch(B, T, H, seed + s, list(lens))

        def d_side(mods, phon, s=s):
            # What d_optimizer.step() consumes (train.py:210-211): the discriminators' gradients of dis_loss (R1
            # double backward included), v_front's sync-loss contribution, and d dis_loss / d phon.
            for mod in cpu_ref.D_NAMES + ('v_front',):
                gn = 0.0
                for pname, p in m[mod].named_parameters():
                    if p.grad is None:
                        continue
                    idx = fixtures.sample_idx(p.numel(), 8, mod + '.' + pname)
                    out['s%d:dgrad:%s:%s' % (s, mod, pname)] = np32(p.grad).reshape(-1)[idx]
                    gn += float(p.grad.double().pow(2).sum())
                out['s%d:dgradnorm:%s' % (s, mod)] = np.float64(gn ** 0.5)
            dp = np32(phon.grad)
            out['s%d:dphon' % s] = dp if not out_stride else dp.reshape(-1)[::out_stride].copy()
        res = cpu_ref.gan_step(drive, opts, b, variant=variant, noise=b['noise'], on_d_backward=d_side)
        for k in ('dis_loss', 'gen_loss', 'sync_loss', 'g_sync_loss', 'recon_loss', 'g_loss'):
            out['s%d:%s' % (s, k)] = np.float64(res[k])
        out['s%d:grad_p' % s] = np.array(res['grad_p'], dtype=np.float64)
        for k in ('g1', 'g2', 'g3') + (('gs',) if out_stride else ()):
            v = np32(res[k])
            out['s%d:%s' % (s, k)] = v if not out_stride else v.reshape(-1)[::out_stride].copy()
        for mod in sorted(m):
            gn, pn = 0.0, 0.0
            for pname, p in m[mod].named_parameters():
                idx = fixtures.sample_idx(p.numel(), 8, mod + '.' + pname)
                out['s%d:param:%s:%s' % (s, mod, pname)] = np32(p).reshape(-1)[idx]
                if p.grad is not None:
                    out['s%d:grad:%s:%s' % (s, mod, pname)] = np32(p.grad).reshape(-1)[idx]
                    gn += float(p.grad.double().pow(2).sum())
                pn += float(p.detach().double().pow(2).sum())
            out['s%d:gradnorm:%s' % (s, mod)] = np.float64(gn ** 0.5)
            out['s%d:paramnorm:%s' % (s, mod)] = np.float64(pn ** 0.5)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), B=B, T=T, H=H, lens=np.array(lens), seed=seed,
                        wseed=200, nsteps=nsteps, variant=variant, out_stride=out_stride or 0, **out)
    print(name, {k: float(v) for k, v in out.items() if k.endswith('_loss')})


def traj_case(rg, rv, name, variant='grid', B=2, T=20, H=112, lens=(20, 13), nsteps=8, seed=41, eps=1e-6, nper=4):
    """Loss / output trajectory of the reference over several steps: run a on the fixture clips, runs b0..b{nper-1} on
    clips moved by eps * N(0,1) (a different draw per run).  How far the b runs drift from a is the envelope the
    network's own sensitivity allows a reimplementation whose arithmetic differs in the last bits; a test bounds its
    own drift from a by a multiple of the widest of them."""
    out = {}
    for tag in ['a'] + ['b%d' % i for i in range(nper)]:
        m = build_reference(rg, rv, 200)
        for k in m:
            m[k].train()
        opts = cpu_ref.build_optimizers(m, amsgrad=(variant == 'grid'))
        drive = dict(m)
        drive['gen'] = RefGen(m['gen'])
        for s in range(nsteps):
            b = fixtures.make_batch(B, T, H, seed + s, list(lens))
            if tag != 'a':
                g = torch.Generator().manual_seed(seed + 1000 * (1 + int(tag[1:])) + s)
                b['vid'] = b['vid'] + eps * torch.randn(b['vid'].shape, generator=g)
            res = cpu_ref.gan_step(drive, opts, b, variant=variant, noise=b['noise'])
            for k in ('dis_loss', 'gen_loss', 'sync_loss', 'g_sync_loss', 'recon_loss', 'g_loss'):
                out['%s:%d:%s' % (tag, s, k)] = np.float64(res[k])
            if tag == 'a':
                out['a:%d:g3' % s] = np32(res['g3'])
            else:
                out['%s:%d:g3_l1' % (tag, s)] = np.float64(np.abs(np32(res['g3']) - out['a:%d:g3' % s]).mean())
            print(name, tag, s, {k: round(float(res[k]), 6) for k in ('dis_loss', 'gen_loss', 'recon_loss')}, flush=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), B=B, T=T, H=H, lens=np.array(lens), seed=seed,
                        wseed=200, nsteps=nsteps, variant=variant, eps=eps, nper=nper, **out)


def full_case(rg, rv):
    """One reference step at the benchmark's batch and clip length (BASELINE configs[1]: batch 32, 75 frames,
    full-length clips) on 112 x 112 crops, the size the reference trains on: its ResNet ends in AvgPool2d(4)
    (resnet.py:82), which rejects the 3 x 3 maps that 96 x 96 crops leave, so the reference cannot run the
    benchmark's 96 x 96 itself.  Needs about 55 GB of host memory and two minutes of CPU time; not part of the
    default run."""
    step_case(rg, rv, 'step_full', 'grid', B=32, T=75, H=112, lens=(75,) * 32, nsteps=1, seed=51, out_stride=13)
    # BASELINE configs[3]: the train_LRS.py step on a batch of 16 variable-length clips padded to 150 frames
    lens = (150, 143, 131, 127, 118, 109, 101, 96, 88, 80, 73, 67, 59, 52, 46, 40)
    step_case(rg, rv, 'step_full_lrs', 'lrs', B=16, T=150, H=112, lens=lens, nsteps=1, seed=52, out_stride=13)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    rg, rv = import_reference()
    only = sys.argv[1:]
    if only:            # e.g. `make_goldens.py step traj`: regenerate the step and trajectory fixtures only
        if 'step' in only:
            step_case(rg, rv, 'step_grid', 'grid')
            step_case(rg, rv, 'step_lrs', 'lrs')
        if 'traj' in only:
            traj_case(rg, rv, 'traj_grid')
        if 'full' in only:
            full_case(rg, rv)
        return
    forward_case(rg, rv, 'fwd_eval', train=False)
    forward_case(rg, rv, 'fwd_train', train=True)
    vfront_case(rg, rv, 'vfront_h104', 104)
    vfront_case(rg, rv, 'vfront_h128', 128)
    step_case(rg, rv, 'step_grid', 'grid')
    step_case(rg, rv, 'step_lrs', 'lrs')
    traj_case(rg, rv, 'traj_grid')


if __name__ == '__main__':
    main()
