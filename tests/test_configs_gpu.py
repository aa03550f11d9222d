"""The benchmark's own configurations on the GPU against the CPU oracle computed inside the test.

The golden fixtures are taken at 112x112 (the only frame size the reference itself runs: its nn.AvgPool2d(4),
src/models/resnet.py:82, raises on the 3x3 / 2x2 maps that 96x96 / 48x48 clips produce).  The benchmark shapes
(BASELINE.json configs[0] 48x48 and configs[1] 96x96) go through the global-mean extension of the pooling head,
which the oracle implements too (oracle/cpu_ref.py ResNet.head_pool), so they are compared with the oracle run live
on the same seeded weights and inputs: parity with the restatement, not with the reference (which cannot run them).
configs[3] (train_LRS.py step, clips up to T=150 with ragged lengths) is compared the same way at 112x112.
"""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, fixtures

gpu = pytest.mark.gpu
G = ('v_front', 'gen', 'post')
LOSSES = ('dis_loss', 'gen_loss', 'sync_loss', 'g_sync_loss', 'recon_loss', 'g_loss')


@pytest.fixture(autouse=True)
def _reset_precision():
    yield
    from vcagan_amd import ops
    ops.set_precision('fp32')


def _pair(seed, train=True):
    """The oracle's modules and the HIP modules with the same seeded weights."""
    from vcagan_amd import train_step
    ref = fixtures.fill_models(cpu_ref.build_models(), seed)
    hip = fixtures.fill_models(train_step.build_models(), seed)
    for k in ref:
        ref[k].train(train)
        hip[k].cuda().train(train)
    return ref, hip


def _dev(b):
    return {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _check_step(ref, hip, want, got, loss_rtol, out_atol, grad_rel):
    for k in LOSSES:
        w, g = float(want[k]), float(got[k])
        assert abs(g - w) <= loss_rtol * max(1.0, abs(w)), (k, g, w)
    np.testing.assert_allclose(got['grad_p'].cpu().numpy(), np.asarray(want['grad_p']), rtol=max(5 * loss_rtol, 1e-3))
    for k in ('g1', 'g2', 'g3', 'gs'):
        d = (got[k].cpu() - want[k]).abs()
        assert float(d.max()) <= out_atol, (k, float(d.max()))
        assert float(d.mean()) <= 1e-3, (k, float(d.mean()))            # the mel-L1 gate
    for mod in G:
        num = den = 0.0
        for (n, p), (_, q) in zip(hip[mod].named_parameters(), ref[mod].named_parameters()):
            num += float((p.grad.cpu().double() - q.grad.double()).pow(2).sum())
            den += float(q.grad.double().pow(2).sum())
        assert (num / den) ** 0.5 <= grad_rel, (mod, (num / den) ** 0.5)


# (precision, activation atol, loss rtol, gradient relative L2).  The generator-side losses are taken after the
# discriminators' first Adam update (~lr * sign(grad) per weight), which amplifies last-bit differences of the
# D-phase gradients: measured up to 1.2e-4 relative on g_loss in fp32 at T = 150, so the fp32 loss bound is 2e-4.
# Activations: the maximum runs over up to 96 000 mel bins per output here (T = 150) against 3 200 in the golden
# cases, and fp32 sums are accumulated in a different order than on the CPU: measured up to 2.2e-4, bound 4e-4.
PREC = [('fp32', 4e-4, 2e-4, 5e-2), ('bf16x3', 3e-3, 5e-4, 0.15)]


@gpu
@pytest.mark.parametrize('H', [96, 48])
@pytest.mark.parametrize('prec,atol,_l,_g', PREC)
@pytest.mark.parametrize('train', [False, True])
def test_visual_front_small_frames(H, train, prec, atol, _l, _g):
    """96x96 and 48x48 clips (3x3 and 2x2 maps in front of the pooling head) against the oracle."""
    from vcagan_amd import ops
    ops.set_precision(prec)
    ref, hip = _pair(500 + H, train)
    b = fixtures.make_batch(2, 12, H, 31)
    with torch.no_grad():
        wp, ws = ref['v_front'](b['vid'])
        gp, gs = hip['v_front'](b['vid'].cuda())
    assert float((gp.cpu() - wp).abs().max()) <= atol
    assert float((gs.cpu() - ws).abs().max()) <= atol
    if train:   # running statistics of the stem's and the trunk's batch-norms moved the same way
        rs, hs = ref['v_front'].state_dict(), hip['v_front'].state_dict()
        for k in rs:
            if k.endswith('running_mean') or k.endswith('running_var'):
                np.testing.assert_allclose(hs[k].cpu().numpy(), rs[k].numpy(), atol=1e-4, rtol=1e-4)


@gpu
@pytest.mark.parametrize('B,T,H', [(2, 75, 48), (2, 20, 96)])
@pytest.mark.parametrize('prec,atol,lrt,grel', PREC + [('mixed', 3e-3, 5e-4, 0.15)])
def test_grid_step_small_frames(B, T, H, prec, atol, lrt, grel):
    """One full GRID step (train.py:166-237) at BASELINE configs[0] (B2 x 75 x 48x48) and at the benchmark's frame
    size (96x96) against oracle/cpu_ref.gan_step."""
    from vcagan_amd import ops, train_step
    ops.set_precision(prec)
    ref, hip = _pair(600 + H)
    b = fixtures.make_batch(B, T, H, 41, [T, T - 7])
    want = cpu_ref.gan_step(ref, cpu_ref.build_optimizers(ref), b, noise=b['noise'])
    step = train_step.GanStep(hip, train_step.build_optimizers(hip), 'grid')
    d = _dev(b)
    got = step(d['mel'], d['spec'], d['vid'], d['vid_len'], noise=d['noise'])
    _check_step(ref, hip, want, got, lrt, atol, grel)


@gpu
@pytest.mark.parametrize('prec,atol,lrt,grel', PREC)
def test_lrs_step_long_ragged(prec, atol, lrt, grel):
    """train_LRS.py:179-243 at the longest clip length of BASELINE configs[3] (T = 150, mel 600 frames) with ragged
    valid lengths (the key-padding mask of generator.py:162-163 is the only place they reach) against the oracle."""
    from vcagan_amd import ops, train_step
    ops.set_precision(prec)
    ref, hip = _pair(700)
    B, T, H = 2, 150, 112
    b = fixtures.make_batch(B, T, H, 43, [150, 97])
    want = cpu_ref.gan_step(ref, cpu_ref.build_optimizers(ref, amsgrad=False), b, variant='lrs', noise=b['noise'])
    step = train_step.GanStep(hip, train_step.build_optimizers(hip, amsgrad=False), 'lrs')
    d = _dev(b)
    got = step(d['mel'], d['spec'], d['vid'], d['vid_len'], noise=d['noise'])
    _check_step(ref, hip, want, got, lrt, atol, grel)


def test_bucket_batches_group_by_padded_length():
    """Host-side batch sampler of the length buckets (no GPU): every batch comes from one bucket, every index is used
    at most once, clips are padded to the smallest bucket that holds them."""
    from vcagan_amd import train_step as ts
    lens = [50, 51, 75, 76, 149, 150, 33, 100, 101, 125, 60, 70]
    assert [ts.bucket_of(n) for n in (1, 50, 51, 150)] == [50, 50, 75, 150]
    with pytest.raises(ValueError):
        ts.bucket_of(151)
    bb = ts.bucket_batches(lens, 2, seed=3)
    seen = [i for _, idx in bb for i in idx]
    assert len(seen) == len(set(seen))
    for T, idx in bb:
        assert len(idx) == 2 and all(ts.bucket_of(lens[i]) == T for i in idx)
    full = ts.bucket_batches(lens, 2, seed=3, drop_last=False)
    assert sorted(i for _, idx in full for i in idx) == list(range(len(lens)))


@gpu
def test_length_buckets_replay_matches_eager():
    """train_LRS.py steps on clips padded to two different bucket lengths, alternating (BASELINE configs[3]
    'bucketing'): BucketedStep (one captured step per bucket in a shared memory pool, first batch of a bucket eager)
    against the same sequence issued kernel by kernel.  Both runs draw the same device-side noise and dropout masks
    (same seed, same step counter)."""
    from vcagan_amd import ops, train_step as ts
    ops.set_precision('mixed')
    B, H = 2, 96
    seq = [50, 75, 50, 75, 75, 50]
    batches = {T: _dev(fixtures.make_batch(B, T, H, 60 + T, [T, T - 11])) for T in (50, 75)}
    out = {}
    for mode in ('eager', 'bucketed'):
        m = fixtures.fill_models(ts.build_models(), 900)
        for k in m:
            m[k].cuda().train()
        step = ts.GanStep(m, ts.build_optimizers(m, amsgrad=False), 'lrs')
        ops.set_seed(77)
        if torch.device('cuda', 0) in ops._rng_step:
            ops._rng_step[torch.device('cuda', 0)].zero_()
        run = ts.BucketedStep(step) if mode == 'bucketed' else \
            (lambda mel, spec, vid, vl: step(mel, spec, vid, vl, noise='device'))
        res = []
        for T in seq:
            b = batches[T]
            r = run(b['mel'], b['spec'], b['vid'], b['vid_len'])
            res.append({k: float(r[k]) for k in LOSSES} | {'g3': tuple(r['g3'].shape)})
        torch.cuda.synchronize()
        out[mode] = res
        if mode == 'bucketed':
            assert sorted(run.graphed) == [50, 75] and all(len(g.graphs) == 1 for g in run.graphed.values())
        del m, step, run
        torch.cuda.empty_cache()
    for i, (a, b) in enumerate(zip(out['eager'], out['bucketed'])):
        assert a['g3'] == b['g3'] == (B, 1, 80, 4 * seq[i])
        # Step 0 is the same eager launch sequence in both runs.  fp32 atomics make two runs differ in the last bits and
        # every sign-like Adam update amplifies that (same bounds as the other replay tests): step 1 (eager in both,
        # behind one update), then replays against eager launches.
        tol = (2e-5, 3e-3)[i] if i < 2 else 3e-2
        for k in LOSSES:
            assert abs(a[k] - b[k]) <= tol * max(1.0, abs(a[k])), (i, k, a[k], b[k])


@gpu
def test_full_size_step_eager_vs_replay():
    """The benchmark's own shape and precision, B32 x T75 x 96x96 in 'mixed' (BASELINE configs[1]): every loss is
    finite, a step replayed from the captured graph gives the losses of the same step issued kernel by kernel, and
    the first step's losses are those of 'bf16x3' (whose forward arithmetic 'mixed' shares and which is the mode
    compared with the oracle at small sizes) - so the headline's forward pass is the parity-tested one at full size.
    (Too large for the CPU oracle inside a test: size-independent properties instead.  fp32 atomics make two runs
    differ in the last bits.)"""
    from vcagan_amd import ops, train_step
    B, T, H = 32, 75, 96
    b = _dev(fixtures.make_batch(B, T, H, 47))
    out, first = {}, {}
    for mode in ('eager', 'graph', 'bf16x3'):
        from vcagan_amd import train_step as ts
        ops.set_precision('bf16x3' if mode == 'bf16x3' else 'mixed')
        m = fixtures.fill_models(ts.build_models(), 800)
        for k in m:
            m[k].cuda().train()
        step = ts.GanStep(m, ts.build_optimizers(m), 'grid')
        if mode == 'graph':
            gs = ts.GraphedStep(step, b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'], warmup=2)
            res = gs(b['mel'], b['spec'], b['vid'], None, b['noise'])
        else:
            for i in range(1 if mode == 'bf16x3' else 3):
                res = step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
                if i == 0:
                    first[mode] = {k: float(res[k]) for k in LOSSES}
        torch.cuda.synchronize()
        out[mode] = {k: float(res[k]) for k in LOSSES}
        assert all(np.isfinite(v) for v in out[mode].values()), out[mode]
        assert torch.isfinite(res['g3']).all() and torch.isfinite(res['grad_p']).all()
        for p in m['v_front'].parameters():
            assert torch.isfinite(p).all()
        del m, step, res
        torch.cuda.empty_cache()
    for k in LOSSES:
        assert abs(first['eager'][k] - first['bf16x3'][k]) <= 2e-4 * max(1.0, abs(first['eager'][k])), (k, first)
    for k in LOSSES:
        assert abs(out['eager'][k] - out['graph'][k]) <= 3e-2 * max(1.0, abs(out['eager'][k])), (k, out)


@gpu
@pytest.mark.parametrize('prec', ['mixed', 'bf16x3', 'fp16-mixed'])
@pytest.mark.parametrize('name,variant,shape', [('step_full', 'grid', (32, 75, 112)),
                                                ('step_full_lrs', 'lrs', (16, 150, 112))])
def test_benchmark_sized_step_matches_reference(name, variant, shape, prec):
    """One step at the benchmark's batch and clip length (32 clips of 75 frames, train.py's step; BASELINE
    configs[1]) and one of train_LRS.py's step on 16 clips of 40-150 valid frames padded to 150 (configs[3]) against
    fixtures written from the reference itself (oracle/make_goldens.py full_case), in the benchmark's precision and in
    the three-product one: losses, R1 terms, sampled generator / postnet outputs (the mel-L1 gate is 1e-3), the
    gradients both optimizers consume and the parameters after Adam.  The crops are 112 x 112, the reference's training
    size: its ResNet ends in AvgPool2d(4) (resnet.py:82) and cannot take the 3 x 3 maps of the benchmark's 96 x 96
    crops, so no reference output exists at that size."""
    from parity_util import check_d_side, check_step, load, measure_step
    from vcagan_amd import ops, train_step
    ops.set_precision(prec)
    g = load(name)
    B, T, H = int(g['B']), int(g['T']), int(g['H'])
    assert (B, T, H) == shape and str(g['variant']) == variant
    m = fixtures.fill_models(train_step.build_models(), int(g['wseed']))
    for k in m:
        m[k].cuda().train()
    step = train_step.GanStep(m, train_step.build_optimizers(m, amsgrad=(variant == 'grid')), variant)
    b = _dev(fixtures.make_batch(B, T, H, int(g['seed']), list(g['lens'])))
    opts = (step.g_opt, step.d_opt)
    sc = step.scaler.value if step.scaler is not None else 1.0     # the half modes' gradients carry the loss scale
    step.d_phase(b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'])
    for o in opts:
        o.all_grads.mul_(1.0 / sc)
    mt = check_d_side(m, step._st['phon_d'].grad / sc, g, 0, 3e-2)
    print(prec, variant, 'D side', {k: round(v, 4) for k, v in mt.items()})
    for o in opts:
        o.all_grads.mul_(sc)
    step.g_phase()
    res = step.finish()
    for o in opts:
        o.all_grads.mul_(1.0 / sc)
    res = {k: (v.item() if v.numel() == 1 else v.cpu()) for k, v in res.items()}
    res['grad_p'] = res['grad_p'].numpy()
    check_step(m, res, g, 0, loss_rtol=5e-4, out_atol=3e-3, grad_rel=0.15, param_p90=2.1e-4,
               skip_grads=('dis1', 'dis2', 'dis3', 's_dis'))
    mt = measure_step(m, res, g, 0, ('dis1', 'dis2', 'dis3', 's_dis'))
    print(prec, variant, {k: float('%.3g' % v) for k, v in mt.items()})
    assert max(mt['out_l1:' + k] for k in ('g1', 'g2', 'g3', 'gs')) <= 3e-4        # a third of the 1e-3 gate


@gpu
def test_one_rank_rccl_segmented_replay():
    """The data-parallel code path with a live RCCL process group (one rank, in this process): the per-phase graphs
    with the gradient all-reduces issued between them (three graphs and two blocking collectives, or five graphs with
    the collectives started slice by slice beside the phases that do not need them) give the losses of the
    single-graph replay, and the overlapped schedule leaves the same gradients in the flat buffers as the plain one
    (the reference's counterpart is nn.DataParallel, train.py:112-119)."""
    import os
    import torch.distributed as dist
    from vcagan_amd import ops, train_step
    ops.set_precision('bf16x3')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        b = _dev(fixtures.make_batch(2, 20, 112, 5, [20, 13]))
        out, grads = {}, {}
        for mode in ('single', 'segmented', 'overlapped'):
            m = fixtures.fill_models(train_step.build_models(), 300)
            for k in m:
                m[k].cuda().train()
            seg = mode != 'single'
            step = train_step.GanStep(m, train_step.build_optimizers(m), 'grid', force_exchange=seg,
                                      overlap=(mode == 'overlapped'))
            assert step.d_exchange.active == seg and step.g_exchange.active == seg
            assert step.overlap == (mode == 'overlapped')
            if mode == 'overlapped':          # generator + postnet are neighbours in the flat buffer: one collective
                sl = step.g_slices('gen', 'post')
                assert len(sl) == 1 and sl[0].numel() == step.gflat['gen'].numel() + step.gflat['post'].numel()
                assert sum(t.numel() for t in sl + step.g_slices('v_front')) == step.g_opt.all_grads.numel()
            # the first eager step's gradients (same weights, same batch, same noise in every mode)
            step(b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'])
            grads[mode] = (step.g_opt.all_grads.clone(), step.d_opt.all_grads.clone())
            gs = train_step.GraphedStep(step, b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'], warmup=1,
                                        segmented=seg)
            assert len(gs.graphs) == {'single': 1, 'segmented': 3, 'overlapped': 5}[mode]
            for _ in range(2):
                res = gs(b['mel'], b['spec'], b['vid'], None, b['noise'])
                float(res['gen_loss'])
            torch.cuda.synchronize()
            out[mode] = {k: float(res[k]) for k in LOSSES}
        for mode in ('segmented', 'overlapped'):
            for k in LOSSES:
                assert abs(out['single'][k] - out[mode][k]) <= 3e-2 * max(1.0, abs(out['single'][k])), (mode, k, out)
        # Cutting the backward pass at the phoneme features only changes the order of one fp32 sum there.  Two runs of
        # the very same schedule differ by about 1e-2 (generator side) and 2e-4 ... 3e-3 (discriminator side) already: the
        # order of the weight-gradient atomics, amplified by the network's kinks.  A lost contribution (the D phase's
        # sync gradient, the sentence encoder's path) would change the front-end's gradient by tens of percent.
        sizes = [step.gflat[n].numel() for n in train_step.G_NAMES]
        rel, same = {}, {}
        for name, a, c, o in zip(('g', 'd'), grads['segmented'], grads['overlapped'], grads['single']):
            parts = zip(train_step.G_NAMES, a.split(sizes), c.split(sizes), o.split(sizes)) if name == 'g' else \
                [('d', a, c, o)]
            for n, pa, pc, po in parts:
                rel[n] = float((pa - pc).norm() / pa.norm())
                same[n] = float((pa - po).norm() / pa.norm())
        print('schedules: overlapped vs plain', rel, 'plain vs plain', same)
        assert all(rel[n] < 5e-2 for n in rel), (rel, same)
    finally:
        dist.destroy_process_group()
