"""Product modules (HIP path, on the GPU) vs the golden fixtures generated from the reference.

Tolerances: fp32 kernels accumulate in a different order than the CPU reference, and 60+ layers of train-mode
batch-norm amplify that: activations are compared at 2e-4 abs (mel outputs are in [-1,1]; the BASELINE gate is
mel-L1 <= 1e-3), step gradients statistically (see parity_util.check_step)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures
from parity_util import check_d_side, check_step, close, load, traj_deviation

gpu = pytest.mark.gpu
D = ('dis1', 'dis2', 'dis3', 's_dis')


def _models(seed, train):
    from vcagan_amd import train_step
    m = fixtures.fill_models(train_step.build_models(), seed)
    for k in m:
        m[k].cuda().train(train)
    return m


def _cuda_batch(b):
    return {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}


@gpu
def test_state_dict_keys_match_oracle():
    from oracle import cpu_ref
    from vcagan_amd import train_step
    a, b = train_step.build_models(), cpu_ref.build_models()
    for k in a:
        sa, sb = a[k].state_dict(), b[k].state_dict()
        assert list(sa.keys()) == list(sb.keys()), k
        assert all(sa[n].shape == sb[n].shape for n in sa), k


@pytest.fixture(autouse=True)
def _reset_precision():
    yield
    try:
        from vcagan_amd import ops
        ops.set_precision('fp32')
    except Exception:
        pass


# (precision, activation atol, mel-L1 bound).  The north-star gate is mel-L1 <= 1e-3 against the reference.
PRECISIONS = [('fp32', 2e-4, 2e-5), ('bf16x3', 3e-3, 3e-4)]


@gpu
@pytest.mark.parametrize('prec,atol,l1', PRECISIONS)
@pytest.mark.parametrize('train', [False, True])
def test_forward_matches_reference(train, prec, atol, l1):
    from vcagan_amd import ops
    ops.set_precision(prec)
    g = load('fwd_train' if train else 'fwd_eval')
    B, T, H = int(g['B']), int(g['T']), int(g['H'])
    m = _models(int(g['wseed']), train)
    b = _cuda_batch(fixtures.make_batch(B, T, H, int(g['seed']), list(g['lens'])))
    tol = dict(atol=atol, rtol=0)
    with torch.no_grad():
        phon, sent = m['v_front'](b['vid'])
        close(phon, g['phon'], **tol); close(sent, g['sent'], **tol)
        g1, g2, g3 = m['gen'](sent, phon, b['vid_len'], noise=b['noise'])
        gs = m['post'](g3)
        for name, t in (('g1', g1), ('g2', g2), ('g3', g3), ('gs', gs)):
            close(t, g[name], **tol)
            assert np.abs(t.cpu().numpy() - g[name]).mean() < l1, name      # mel-L1 vs reference
        mel = b['mel']
        mel1 = F.interpolate(mel, scale_factor=0.25, mode='bilinear')
        mel2 = F.interpolate(mel, scale_factor=0.5, mode='bilinear')
        for i, (d, r, gg) in enumerate(zip(('dis1', 'dis2', 'dis3'), (mel1, mel2, mel), (g1, g2, g3)), 1):
            u, c = m[d](r, sent, T)
            close(u, g['real_u%d' % i], **tol); close(c, g['real_c%d' % i], **tol)
            u, c = m[d](gg, sent, T)
            close(u, g['fake_u%d' % i], **tol); close(c, g['fake_c%d' % i], **tol)
        close(m['s_dis'](phon, mel), g['sync_d'], **tol)
        close(m['s_dis'](phon, g3, True), g['sync_g'], **tol)
    if train:
        for key in g.files:
            if key.startswith('stat:'):
                _, mod, name = key.split(':')
                close(m[mod].state_dict()[name], g[key], atol=1e-4, rtol=1e-4)


@gpu
@pytest.mark.parametrize('name', ['vfront_h104', 'vfront_h128'])
def test_avgpool_window_rule(name):
    g = load(name)
    m = _models(int(g['wseed']), False)
    b = fixtures.make_batch(int(g['B']), int(g['T']), int(g['H']), int(g['seed']))
    with torch.no_grad():
        phon, sent = m['v_front'](b['vid'].cuda())
    close(phon, g['phon'], atol=2e-4, rtol=0); close(sent, g['sent'], atol=2e-4, rtol=0)


@gpu
@pytest.mark.parametrize('prec', ['fp32', 'bf16x3', 'mixed'])
@pytest.mark.parametrize('variant', ['grid', 'lrs'])
def test_step_matches_reference(variant, prec):
    from vcagan_amd import ops, train_step
    ops.set_precision(prec)
    g = load('step_' + variant)
    B, T, H = int(g['B']), int(g['T']), int(g['H'])
    m = _models(int(g['wseed']), True)
    opts = train_step.build_optimizers(m, amsgrad=(variant == 'grid'))
    step = train_step.GanStep(m, opts, variant)
    for s in range(int(g['nsteps'])):
        b = _cuda_batch(fixtures.make_batch(B, T, H, int(g['seed']) + s, list(g['lens'])))
        res = step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
        res = {k: (v.item() if v.numel() == 1 else v.cpu()) for k, v in res.items()}
        res['grad_p'] = res['grad_p'].numpy()
        if s == 0 and prec == 'fp32':
            check_step(m, res, g, s, loss_rtol=5e-5, out_atol=2e-4, grad_rel=5e-2, param_p90=2e-5, skip_grads=D)
        elif s == 0:
            check_step(m, res, g, s, loss_rtol=5e-4, out_atol=3e-3, grad_rel=0.15, param_p90=2.1e-4, skip_grads=D)
        elif prec == 'fp32':
            # Second step: the first Adam update moves every weight by ~lr * sign(grad), so last-bit differences in
            # step 0's gradients flip signs of small entries and the two runs part (the CPU oracle run twice with a
            # 1e-6 input perturbation parts as far).  Measured here (tools/step_tolerance.py): loss 1.1e-4 / 7.0e-4,
            # outputs 3.9e-2, mel-L1 4.5e-3, gradients 0.17 / 0.20 (GRID / LRS); bounds = about twice that.
            check_step(m, res, g, s, loss_rtol=1.5e-3, out_atol=8e-2, grad_rel=0.4, param_p90=2.1e-4, skip_grads=D)
        else:
            # bf16x3 / mixed measured over several runs: loss 3.0e-4 .. 1.7e-3, outputs 6.4e-2 .. 8.3e-2,
            # mel-L1 8.6e-3 .. 1.0e-2, gradients 0.19 .. 0.29
            check_step(m, res, g, s, loss_rtol=3e-3, out_atol=0.17, grad_rel=0.55, param_p90=2.1e-4, skip_grads=D)


# Bounds of the D-side gradients (relative L2 over each network's sampled entries, gradient norms, d dis_loss/d phon),
# per precision: (step 0, step 1).  Step 0 is the parity statement; step 1 sits behind the first, sign-like Adam update.
# Measured (tools/step_tolerance.py, worst network, GRID / LRS): step 0 fp32 0.0014, bf16x3 0.0054, mixed 0.0075;
# step 1 fp32 0.039, bf16x3 0.063, mixed 0.067.  Bounds = 3-4x that.
D_SIDE_BOUNDS = {'fp32': (6e-3, 0.15), 'bf16x3': (2e-2, 0.2), 'mixed': (3e-2, 0.2)}


@gpu
@pytest.mark.parametrize('prec', ['fp32', 'bf16x3', 'mixed'])
@pytest.mark.parametrize('variant', ['grid', 'lrs'])
def test_d_side_gradients_match_reference(variant, prec):
    """The gradients d_optimizer.step() consumes (train.py:209-211): dis1/dis2/dis3 (softplus heads + the R1 penalty's
    double backward), s_dis, and the sync loss's gradient w.r.t. the phoneme features (its only way into v_front),
    read between the D phase's backward pass and the optimizer step and compared with what the reference held at
    that moment ('dgrad' keys of the step fixtures)."""
    from vcagan_amd import ops, train_step
    ops.set_precision(prec)
    g = load('step_' + variant)
    B, T, H = int(g['B']), int(g['T']), int(g['H'])
    m = _models(int(g['wseed']), True)
    step = train_step.GanStep(m, train_step.build_optimizers(m, amsgrad=(variant == 'grid')), variant)
    for s in range(int(g['nsteps'])):
        b = _cuda_batch(fixtures.make_batch(B, T, H, int(g['seed']) + s, list(g['lens'])))
        step.d_phase(b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'])
        torch.cuda.synchronize()
        mt = check_d_side(m, step._st['phon_d'].grad, g, s, D_SIDE_BOUNDS[prec][min(s, 1)])
        print(prec, variant, s, {k: round(v, 4) for k, v in mt.items()})
        step.d_exchange.reduce()
        step.g_phase()
        step.g_exchange.reduce()
        step.finish()


# (precision, multiple of the reference's own envelope for the losses / for the mel-L1, floors):
# |hip - a| <= k * envelope + floor per step, where the envelope is the running maximum of |a - b_i| between the
# reference's run a and its four runs b_i (clips 1e-6 apart).  The floor is the precision's step-0 deviation (losses
# relative, mel-L1 absolute), which the envelope (7e-6 at step 0) does not cover.  The four b runs agree on the mel-L1
# drift within 10 % at every step (it is set by the sign-like Adam updates, not by the size of the perturbation), so
# that bound is the tight one; the loss drifts scatter by an order of magnitude between the b runs themselves.
# Measured (worst ratio to the envelope over the 8 steps): fp32 losses 2.8x, mel-L1 1.25x; mixed losses 10x at step 1
# (2e-5 against an envelope of 2e-6: covered by the floor), 2x afterwards, mel-L1 3.0x at step 1, 1.3x at the end.
TRAJ_BOUNDS = [('fp32', 5.0, 2.0, 1e-4, 3e-5), ('bf16x3', 5.0, 5.0, 1e-3, 3e-4), ('mixed', 5.0, 5.0, 1e-3, 3e-4),
               ('fp16-mixed', 5.0, 2.0, 1e-4, 3e-5)]     # IEEE-half operands under the loss scale: held to fp32's bounds


@gpu
@pytest.mark.parametrize('prec,k_loss,k_l1,loss_floor,l1_floor', TRAJ_BOUNDS)
def test_trajectory_stays_inside_reference_envelope(prec, k_loss, k_l1, loss_floor, l1_floor):
    """Eight consecutive training steps against the reference's trajectory (tests/golden/traj_grid.npz): dis_loss,
    gen_loss, recon_loss, sync_loss and the mel-L1 of g3 stay within a small multiple of the distance the reference
    itself shows between runs whose clips differ by 1e-6.  A biased gradient would walk out of that envelope;
    last-bit noise does not."""
    from vcagan_amd import ops, train_step
    ops.set_precision(prec)
    g = load('traj_grid')
    m = _models(int(g['wseed']), True)
    step = train_step.GanStep(m, train_step.build_optimizers(m), 'grid')

    def run(b):
        res = step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
        return {k_: (v.item() if v.numel() == 1 else v.cpu()) for k_, v in res.items()}
    dev = traj_deviation(run, g, to_device=_cuda_batch)
    for s, d in enumerate(dev):
        print(prec, s, {k_: ('%.2e' % a, '%.2e' % e) for k_, (a, e) in d.items()})
    for s, d in enumerate(dev):
        for key, (got, env) in d.items():
            k, floor = (k_l1, l1_floor) if key == 'g3_l1' else (k_loss, loss_floor)
            assert got <= k * env + floor, (prec, s, key, got, env)


@gpu
def test_graph_replay_matches_eager():
    """A step captured into one HIP graph, or into the three per-phase graphs the data-parallel path replays around
    its gradient all-reduces, gives the same losses as issuing the kernels from Python
    (fp32 atomics make any two runs differ in the last bits and the sign-like first Adam steps amplify that: two
    eager runs of this case differ by ~0.5 % in gen_loss after 4 steps, so the bound here is 3 %)."""
    from vcagan_amd import ops, train_step
    ops.set_precision('bf16x3')
    B, T, H = 2, 20, 112
    b = _cuda_batch(fixtures.make_batch(B, T, H, 5, [20, 13]))
    out = {}
    for mode in ('eager', 'graph', 'segmented'):
        m = _models(300, True)
        opts = train_step.build_optimizers(m)
        step = train_step.GanStep(m, opts, 'grid')
        if mode == 'eager':
            for _ in range(4):
                res = step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
        else:
            gs = train_step.GraphedStep(step, b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'], warmup=2,
                                        segmented=(mode == 'segmented'))
            assert len(gs.graphs) == (3 if mode == 'segmented' else 1)
            for _ in range(2):
                res = gs(b['mel'], b['spec'], b['vid'], None, b['noise'])
                float(res['gen_loss'])          # what a training loop that logs its losses does between replays
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        out[mode] = {k: float(res[k]) for k in ('dis_loss', 'gen_loss', 'recon_loss', 'sync_loss')}
        assert opts[0].t_dev.item() == 4 and opts[1].t_dev.item() == 4
    for mode in ('graph', 'segmented'):
        for k in out['eager']:
            assert abs(out['eager'][k] - out[mode][k]) <= 3e-2 * abs(out['eager'][k]), (mode, k, out)


@gpu
def test_graph_replay_is_stable_under_host_reads():
    """Replays of a captured forward pass give the same result whatever the host does between them.  The
    split-reduction convolutions (few output tiles, deep reduction: the discriminators' heads) accumulate into a
    zero-filled output; with that fill issued as hipMemsetAsync the graph held memset nodes, and a device-to-host
    read plus a device synchronisation between two replays made later replays accumulate into stale values."""
    from vcagan_amd import ops
    ops.set_precision('bf16x3')
    m = _models(301, True)
    dis = m['dis3']
    g = torch.Generator(device='cpu').manual_seed(7)
    mel = (torch.rand(32, 1, 80, 300, generator=g) * 2 - 1).cuda()
    sent = torch.randn(32, 512, 75, generator=g).cuda()
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eager = [o.clone() for o in dis(mel, sent, 75)]
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = dis(mel, sent, 75)
    for i in range(4):
        graph.replay()
        outs[0].flatten()[0].item()
        torch.cuda.synchronize()
        for a, b in zip(eager, outs):
            assert torch.isfinite(b).all(), i
            assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-6, (i, float((a - b).abs().max()))
