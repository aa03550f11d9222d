"""Benchmark of the VCA-GAN hot path: one full G+D training step (forward, R1 double backward, both backward
passes, both Adam steps) on synthetic lip-video clips.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Workload = BASELINE.json configs[1]: batch 32 per GPU, 75 frames of 96x96
(mel 80x300, spectrogram 321x300), random-init weights, synthetic inputs already resident in HBM.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, T, H = 32, 75, 96
FLOP_PER_FRAME = 9.84e9          # full step at 96x96, counted on the reference with torch's FlopCounterMode
PEAK_TFLOPS = {'fp32': 157.3, 'bf16': 2500.0, 'bf16x3': 2500.0, 'mixed': 2500.0,
               'fp16': 2500.0, 'fp16x3': 2500.0, 'fp16-mixed': 2500.0}
MFMA_PER_PRODUCT = {'fp32': 1, 'bf16': 1, 'bf16x3': 3, 'mixed': 3, 'fp16': 1, 'fp16x3': 3, 'fp16-mixed': 3}
DTYPE_NOTE = {
    'mixed': 'bf16 matrix cores, fp32 accumulate: split-bf16 (3 products, ~fp32 accuracy) in every forward product and '
             'in the gradient chain whose result is the R1 penalty (a loss value), single bf16 products in the '
             'input/weight gradients incl. the second-order terms of that penalty; everything else fp32',
    'bf16x3': 'bf16 matrix cores, fp32 accumulate: split-bf16 (3 products, ~fp32 accuracy) in every product; '
              'everything else fp32',
    'fp16': 'IEEE-half matrix cores, fp32 accumulate, one product everywhere, fp32 master weights, dynamic loss scale '
            '(device-side, skip-on-overflow); everything else fp32',
    'fp16-mixed': 'IEEE-half matrix cores, fp32 accumulate: split-half (3 products) in every forward product and the R1 '
                  'gradient chain, single half products in the input/weight gradients under a dynamic loss scale; '
                  'everything else fp32',
    'fp16x3': 'IEEE-half matrix cores, fp32 accumulate: split-half (3 products) in every product, dynamic loss scale; '
              'everything else fp32',
}


def synth(b, t, h, device, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    vid = torch.randn(b, 1, t, h, h, generator=g)
    mel = torch.rand(b, 1, 80, 4 * t, generator=g) * 2 - 1
    spec = torch.rand(b, 1, 321, 4 * t, generator=g)
    noise = torch.randn(b, 128, 20, t, generator=g)
    return dict(vid=vid.to(device), mel=mel.to(device), spec=spec.to(device), noise=noise.to(device),
                vid_len=[t] * b)


def _time(fn, reps=5, warm=2):
    """Average duration (seconds) of fn() from HIP events on the current stream (the one the kernels launch on)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def _time_graph(fn, reps=10, replays=3):
    """Device time of fn() (seconds) with `reps` calls captured into one HIP graph and replayed: what the call costs
    inside a replayed training step.  Issued from Python a chain of short launches measures the interpreter (about
    13 us per launch here), not the GPU."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * replays) * 1e-3


def roofline_probe(device, dtype):
    """Time the dominant kernel (the implicit-GEMM convolution, forward = split product in both 'bf16x3' and 'mixed')
    on the largest layer the step runs, decode.0.conv2 (5x5, 512->512 on the 20x75 map of 32 clips; decode.0.conv1
    no longer sees the tiled 640-channel input), with HIP events on the stream it is launched on."""
    from vcagan_amd import ops
    x = torch.randn(B, 20, T, 512, device=device)
    w = torch.nn.Parameter(torch.randn(512, 512, 5, 5, device=device) * 0.01)
    bias = torch.zeros(512, device=device)
    flops = 2.0 * B * 20 * T * 512 * 25 * 512
    with torch.no_grad():
        dt = _time(lambda: ops.conv2d(x, w, bias, 1, 2))
    peak = PEAK_TFLOPS[dtype]
    ach = flops / dt / 1e12                         # algorithmic (fp32-equivalent) rate
    issued = ach * MFMA_PER_PRODUCT[dtype]          # matrix-core work actually issued (3 bf16 MFMAs per product)
    kern = 'conv_f32_fwd_kernel<128,2,2>' if dtype == 'fp32' else \
        'conv_bf16_halo_kernel<128,2,2,%d>' % MFMA_PER_PRODUCT[dtype]
    # `traffic` is NOT measured by this run: it is the fabric-side byte count of one launch of this kernel on this
    # layer from separate rocprofv3 --pmc passes, 2 x FETCH_SIZE (gfx950 counts 128-byte reads at 64) + WRITE_SIZE;
    # `traffic_source` names the committed profile.  Algorithmic bytes (input + packed weights + output, each once)
    # are 98 + 26 + 98 = 222 MB; the excess is the input halo re-read once per 128-channel output block and the
    # weight stream re-read per pixel block, absorbed by the 256 MiB Infinity Cache.
    traffic = (2 * 533887 + 98268) * 1024 if dtype in ('bf16x3', 'mixed') else None
    return dict(bound='mfma', kernel=kern + ' on decode.0.conv2', achieved=round(ach, 2), peak=peak, unit='TFLOP/s',
                frac=round(ach / peak, 4), issued_mfma_tflops=round(issued, 2),
                issued_frac=round(issued / peak, 4), traffic=traffic,
                traffic_source='profiles/r3_decode0conv2_pmc.txt (separate --pmc passes, not this run)'
                if traffic else None, launch_ms=round(dt * 1e3, 3), flops_per_launch=flops)


def north_star_probes(device, models):
    """The two stages BASELINE.json's north_star sets targets for, timed live with HIP events:
    the 3-D convolution front-end against the HBM roofline, and the Visual Context Attention module (its four
    projections + QK^T + mask + softmax + AV, generator.py:142-171) against the matrix-core peak."""
    from vcagan_amd import ops
    out = {}
    vf = models['v_front']
    vid = torch.randn(B, 1, T, H, H, device=device)
    with torch.no_grad():
        xs = vid.permute(0, 2, 3, 4, 1).contiguous()                # (B, T, H, W, 1) channels-last
        dt = _time_graph(lambda: vf.front(xs), reps=5)
    oh = H // 4
    # algorithmic bytes of a1 + a2 as one stage: the clip once, the pooled map it hands to the ResNet trunk and one
    # argmax byte per pooled element for the backward pass (the full-resolution stem output is an implementation
    # detail: whatever of it an implementation writes and re-reads shows up as time, not as algorithmic bytes)
    alg = B * T * H * H * 4 + B * T * oh * oh * 64 * 5
    # what this implementation moves in train mode: the clip, the full-resolution pre-BN stem output written once
    # (batch statistics need the whole batch before anything can be normalised, and the backward pass needs the map)
    # and read once by the fused BN + PReLU + max-pool pass, the pooled map and its argmax bytes
    moved = B * T * H * H * 4 + 2 * B * T * (H // 2) * (H // 2) * 64 * 4 + B * T * oh * oh * 64 * 5
    out['frontend'] = dict(stage='Conv3d stem + BN3d + PReLU + MaxPool3d (visual_front.py:11-14), train mode',
                           ms=round(dt * 1e3, 3), algorithmic_bytes=alg, achieved_GBps=round(alg / dt / 1e9, 1),
                           peak_GBps=8000.0, frac=round(alg / dt / 8e12, 4), moved_bytes=moved,
                           moved_GBps=round(moved / dt / 1e9, 1), moved_frac=round(moved / dt / 8e12, 4),
                           issued_mfma_tflops=round(3 * 2.0 * B * T * (H // 2) * (H // 2) * 64 * 288 / dt / 1e12, 1),
                           note='three bf16 MFMAs per product over 5x7x7 taps (padded to 288): the stem kernel is '
                                'bound by the matrix cores and LDS, not by HBM')
    att = models['gen'].att1
    ctx = torch.randn(B, T, 512, device=device)
    g1 = torch.randn(B, 20, T, 128, device=device)
    lens = torch.full((B,), T, device=device, dtype=torch.int32)
    with torch.no_grad():
        dt = _time_graph(lambda: att(ctx, g1, lens))
        dt_eager = _time(lambda: att(ctx, g1, lens))
    S, Tq, D = T, T, 256
    fl = 2.0 * B * (2 * S * 512 * D + Tq * 2560 * D + 2 * Tq * S * D + Tq * D * 1280)
    peak = PEAK_TFLOPS[ops.get_precision()]
    out['vca_module'] = dict(stage='att1: k/v/q/out projections + QK^T + mask + softmax + AV', ms=round(dt * 1e3, 3),
                             gflop=round(fl / 1e9, 2), achieved_TFLOPs=round(fl / dt / 1e12, 1), peak_TFLOPs=peak,
                             frac=round(fl / dt / 1e12 / peak, 4),
                             qk_av_share_of_flops=round(2.0 * B * 2 * Tq * S * D / fl, 3),
                             timing='device time under graph replay (as inside a replayed step); issued launch by '
                                    'launch from Python the same call takes %.3f ms, host-bound' % (dt_eager * 1e3))
    return out


def cpu_baseline():
    """The CPU oracle (restatement of the reference's PyTorch CPU path) on the host cores: B2 x T75 x 112x112,
    one warm-up step and three timed ones.  Threads: every core this job may use (a one-GPU lease of the box owns 16
    of the machine's cores, published through OMP_NUM_THREADS; os.cpu_count() there counts the whole machine)."""
    from oracle import cpu_ref, fixtures
    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get('OMP_NUM_THREADS', '0') or 0) or (os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    m = cpu_ref.build_models()
    for k in m:
        m[k].train()
    opts = cpu_ref.build_optimizers(m)
    b = fixtures.make_batch(2, 75, 112, 1)
    cpu_ref.gan_step(m, opts, b, noise=b['noise'])
    n = 3
    t0 = time.time()
    for _ in range(n):
        cpu_ref.gan_step(m, opts, b, noise=b['noise'])
    dt = (time.time() - t0) / n
    return dict(value=round(2 * 75 / dt, 2), unit='frames/s', cores=cores, kind='port',
                sample='%d full G+D steps at B2xT75x112x112 fp32 (oracle/cpu_ref.py), %.2f s/step' % (n, dt))


def bench_buckets(args, step, device, train_step, result_fd):
    """train_LRS.py steps over a synthetic dataset of ragged clip lengths, batches drawn per length bucket
    (train_step.bucket_batches) and replayed from one captured step per bucket (train_step.BucketedStep).  Reports
    frames/s over VALID frames beside the padded figure, and what padding every clip to 150 frames would give."""
    g = torch.Generator().manual_seed(11)
    nb = max(args.steps, 10)
    lengths = [int(v) for v in torch.randint(40, 151, (args.batch * nb * 2,), generator=g)]
    batches = train_step.bucket_batches(lengths, args.batch, seed=5)[:nb]
    data = {}
    for Tb in sorted({Tb for Tb, _ in batches} | {150}):
        data[Tb] = synth(args.batch, Tb, H, device, 1234 + Tb)
    run = train_step.BucketedStep(step)

    def one(Tb, idx):
        d = data[Tb]
        return run(d['mel'], d['spec'], d['vid'], [lengths[i] for i in idx])
    for Tb in sorted(data):                       # first batch of a bucket: eager step + capture
        one(Tb, list(range(args.batch)))
    for Tb, idx in batches[:args.warmup]:
        one(Tb, idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        Tb, idx = batches[k % len(batches)]
        out = one(Tb, idx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    used = [batches[k % len(batches)] for k in range(args.steps)]
    valid = sum(lengths[i] for _, idx in used for i in idx)
    padded = sum(Tb * len(idx) for Tb, idx in used)
    per_bucket = {}
    for Tb in sorted(data):
        per_bucket[str(Tb)] = round(_time(lambda: one(Tb, list(range(args.batch))), reps=3, warm=1) * 1e3, 2)
    ms150 = per_bucket['150']
    assert torch.isfinite(out['gen_loss']).item() and torch.isfinite(out['dis_loss']).item()
    res = {'metric': 'lip-video VALID frames/sec (train_LRS.py G+D step, length buckets) @ B%d x 96x96 per GPU' % args.batch,
           'value': round(valid / dt, 1), 'unit': 'frames/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
           'config': {'workload': 'train_LRS.py full GAN step on clips of 40-150 valid frames, batches of %d drawn per '
                                  'length bucket %r and padded to the bucket, one captured step per bucket in a shared '
                                  'memory pool' % (args.batch, list(train_step.LRS_BUCKETS))},
           'padded_frames_per_s': round(padded / dt, 1),
           'valid_fraction_of_padded': round(valid / padded, 3),
           'ms_per_step_by_bucket': per_bucket,
           'pad_every_clip_to_150': {'ms_per_step': ms150, 'valid_frames_per_s': round(valid / args.steps / (ms150 * 1e-3), 1)},
           'batches_per_bucket': {str(Tb): sum(1 for t, _ in used if t == Tb) for Tb in sorted(data)}}
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(res) + '\n').encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=B)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the extra mixed-precision measurement')
    ap.add_argument('--no-graph', action='store_true', help='issue every kernel from Python instead of replaying a HIP graph')
    ap.add_argument('--precision', default='mixed', choices=['fp32', 'bf16x3', 'bf16', 'mixed', 'fp16', 'fp16-mixed', 'fp16x3'])
    ap.add_argument('--host-inputs', nargs='?', const='inline', default=None, choices=['inline', 'prefetch'],
                    help='hand every step its inputs as pinned host tensors: the PCIe-inclusive rate, an extra '
                         'measurement.  inline: copied on the compute stream in front of the replay, nothing overlapped; '
                         'prefetch: copied one batch ahead on a side stream (video.InputPrefetcher)')
    ap.add_argument('--workload', default='grid', choices=['grid', 'lrs', 'lrs_ctc', 'lrs_buckets'],
                    help="grid: BASELINE configs[1] (the headline).  lrs: the train_LRS.py step at configs[3]'s shape "
                         '(batch 16/GPU, clips padded to T=150, ragged valid lengths).  lrs_ctc: the same with the CTC '
                         'auxiliary loss of the frozen ASR scorer (configs[4]).  lrs_buckets: the lrs step on clips of '
                         'ragged lengths grouped into length buckets (50/75/100/125/150 frames), one captured step per '
                         'bucket.  Extra measurements, never the headline.')
    args = ap.parse_args()
    global T
    lrs = args.workload != 'grid'
    if lrs:
        T = 150
        if args.batch == B:
            args.batch = 16

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes a version banner of several
    # lines to stdout when its communicator comes up), so everything this process and its libraries write to file
    # descriptor 1 goes to stderr, and only the result line is written to the real stdout at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get('VCA_BENCH_HANG_TRACE'):      # debugging aid: dump every thread's stack if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['VCA_BENCH_HANG_TRACE']), exit=True)
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE is %d (launch N ranks with torch.distributed.run for --gpus N)'
                 % (args.gpus, world))
    # One process per GPU, each on a device of its own.  Two ranks on one device never get out of their first RCCL
    # collective (both sit in the parameter broadcast until a watchdog kills them), so this is checked before the
    # process group exists, where a wrong launch can still fail by itself with a message.
    ndev = torch.cuda.device_count()
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    if ndev < local_world or local >= ndev:
        sys.exit('bench.py: rank %d (local rank %d of %d) has no GPU of its own: %d device(s) visible; '
                 'run with --gpus <= %d' % (rank, local, local_world, ndev, ndev))
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    use_dist = world > 1 or os.environ.get('VCA_FORCE_EXCHANGE') == '1'   # the latter: one-rank RCCL rehearsal
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=device)
        # The exchange must be RCCL on device buffers: any other backend stages 400 MB per step through the host
        # (grad_exchange.GradExchange) and the run would measure PCIe.
        if dist.get_backend() != 'nccl':
            sys.exit('bench.py: the gradient exchange needs the nccl (RCCL) backend, got %r' % dist.get_backend())

    import vcagan_amd  # noqa: F401
    from vcagan_amd import ops, train_step
    ops.set_precision(args.precision)
    torch.manual_seed(1)
    models = train_step.build_models(device=device)
    for k in models:
        models[k].train()
    if use_dist:                       # identical replicas: broadcast rank 0's initialisation
        for k in sorted(models):
            for t in list(models[k].parameters()) + list(models[k].buffers()):
                dist.broadcast(t.data, 0)
    opts = train_step.build_optimizers(models, amsgrad=not lrs)      # train.py:82-83 / train_LRS.py:97-98
    ctc = None
    if args.workload == 'lrs_ctc':
        from vcagan_amd.models import asr
        ctc = dict(front=asr.Audio_front().to(device), back=asr.Backend().to(device), weight=1.0)
    step = train_step.GanStep(models, opts, 'lrs' if lrs else 'grid', ctc=ctc)
    if args.workload == 'lrs_buckets':
        return bench_buckets(args, step, device, train_step, result_fd)
    data = synth(args.batch, T, H, device, 1234 + rank)
    text = text_len = None
    if lrs:
        g = torch.Generator().manual_seed(7 + rank)
        data['vid_len'] = [T] + [int(v) for v in torch.randint(T // 3, T + 1, (args.batch - 1,), generator=g)]
        if ctc is not None:
            text = torch.randint(1, 28, (args.batch, 40), generator=g).to(device)
            text_len = torch.randint(10, 41, (args.batch,), generator=g).to(device)

    graphed = None
    if not args.no_graph:
        try:
            graphed = train_step.GraphedStep(step, data['mel'], data['spec'], data['vid'], data['vid_len'],
                                             'device',
                                             segmented=True if os.environ.get('VCA_BENCH_SEGMENTED') == '1' else None,
                                             text=text, text_len=text_len)
        except Exception as e:
            # An eager step is a different benchmark (several thousand launches issued from Python): never time it
            # under the headline's name.  --no-graph asks for it explicitly and is labelled in `launch`.
            print('bench.py: graph capture failed: %r' % (e,), file=sys.stderr)
            sys.exit(3)

    feed = data
    prefetch = None
    if args.host_inputs and graphed is not None:
        feed = {k: (v.cpu().pin_memory() if torch.is_tensor(v) else v) for k, v in data.items()}
        if args.host_inputs == 'prefetch':
            from vcagan_amd import video
            import itertools
            host = {k: feed[k] for k in ('mel', 'spec', 'vid')}
            prefetch = video.InputPrefetcher(itertools.repeat(host), device)

    trace_loss = os.environ.get('VCA_BENCH_TRACE_LOSS')    # debugging aid: '1' prints the losses, 'sync' only syncs

    def run(n):
        out = None
        for _ in range(n):
            if prefetch is not None:
                dev_in = next(prefetch)          # already on the device: copied under the previous step
                out = graphed(dev_in['mel'], dev_in['spec'], dev_in['vid'])
            elif graphed is not None:
                out = graphed(feed['mel'], feed['spec'], feed['vid'])
            else:
                out = step(data['mel'], data['spec'], data['vid'], data['vid_len'], noise='device', text=text,
                           text_len=text_len)
            if trace_loss == 'sync':
                torch.cuda.synchronize()
            elif trace_loss:
                print('loss trace: ' + ' '.join('%s %.6f' % (k, float(v)) for k, v in sorted(out.items())
                                                if torch.is_tensor(v) and v.numel() == 1), file=sys.stderr)
        return out

    run(args.warmup)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    if use_dist:                       # HIP events around the two all-reduces of every timed step
        step.d_exchange.timed, step.g_exchange.timed = [], []
    if os.environ.get('VCA_BENCH_CHECK') == '1':
        for n, m in models.items():
            print('check', n, 'params finite', all(bool(torch.isfinite(p).all()) for p in m.parameters()),
                  'grads finite', all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None),
                  file=sys.stderr)
        for k, v in graphed.static.items():
            print('check static', k, bool(torch.isfinite(v.float()).all()), file=sys.stderr)
    t0 = time.perf_counter()
    out = run(args.steps)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    print('final: gen %.6f dis %.6f' % (float(out['gen_loss']), float(out['dis_loss'])), file=sys.stderr)
    assert torch.isfinite(out['gen_loss']).item() and torch.isfinite(out['dis_loss']).item()

    if rank == 0:
        dtype = args.precision
        ms = dt / args.steps * 1e3
        frames = args.batch * T * world
        res = {
            'metric': 'lip-video frames/sec (G+D fwd+bwd step) @ B%dxT%dx96x96 per GPU' % (args.batch, T),
            'value': round(frames / (ms * 1e-3), 1), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 2), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'launch': ('eager' if graphed is None else
                       '3 hip-graphs + 2 all-reduces' if graphed.segmented else 'hip-graph') +
                      (', the discriminators and the sentence encoder on side streams' if train_step.D_STREAMS else ''),
            'config': {'workload': ('GRID full GAN step, batch %d/GPU x 75 frames x 96x96, mel 80x300, random-init '
                                    'weights' % args.batch) if not lrs else
                                   ('train_LRS.py full GAN step%s, batch %d/GPU, clips padded to 150 frames x 96x96 with '
                                    'ragged valid lengths (padded frames are counted), mel 80x600, random-init weights'
                                    % (' + CTC loss of the frozen ASR scorer on the generated mel' if ctc else '',
                                       args.batch)), 'global_batch': args.batch * world,
                       'parallelism': 'dp%d' % world,
                       'noise': 'decoder noise drawn afresh inside every timed step, on the device (counter-based '
                                'generator) instead of torch.randn on the CPU + host-to-device copy'},
            'inputs': ('pinned host tensors, copied one batch ahead on a side stream' if prefetch is not None else
                       'pinned host tensors, copied in every step') if feed is not data else 'resident in HBM',
            'ref_algorithmic_tflops': round(FLOP_PER_FRAME * frames / (ms * 1e-3) / 1e12, 2),
            'final_losses': {'gen': round(float(out['gen_loss']), 4), 'dis': round(float(out['dis_loss']), 4)},
        }
        if use_dist:
            ex = [sum(a.elapsed_time(b) for a, b in e.timed) / max(1, args.steps)
                  for e in (step.d_exchange, step.g_exchange)]
            res['exchange_ms'] = {'d_side': round(ex[0], 3), 'g_side': round(ex[1], 3),
                                  'bytes': [int(sum(g.numel() for g in e.gflats) * 4)
                                            for e in (step.d_exchange, step.g_exchange)],
                                  'schedule': 'overlapped' if step.overlap else 'blocking',
                                  'note': 'per step on rank 0, HIP events on the compute stream around its wait for '
                                          'each side\'s mean all-reduce (issued between the per-phase graphs; the '
                                          'overlapped schedule, VCA_EXCHANGE_OVERLAP=1, starts them slice by slice '
                                          'beside the phases that do not need them)'}
        res['roofline'] = roofline_probe(device, dtype)
        res['dtype_note'] = DTYPE_NOTE.get(dtype, dtype)
        if world == 1 and not lrs:
            res['north_star'] = north_star_probes(device, models)
        if (world == 1 and graphed is not None and not args.no_alt and args.precision in ('mixed', 'bf16x3')
                and not lrs):
            # the same step at the other precision, reported beside `value`, never as `value`
            alt = 'bf16x3' if args.precision == 'mixed' else 'mixed'
            ops.set_precision(alt)
            g2 = train_step.GraphedStep(step, data['mel'], data['spec'], data['vid'], data['vid_len'], 'device')
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                g2(data['mel'], data['spec'], data['vid'])
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t1) / args.steps * 1e3
            res['alt_precision'] = {'dtype': alt, 'dtype_note': DTYPE_NOTE[alt], 'ms_per_step': round(ms2, 2),
                                    'value': round(frames / (ms2 * 1e-3), 1)}
            # and on IEEE-half operands under a dynamic loss scale (the library's second build)
            ops.set_precision('fp16-mixed')
            step.scaler = ops.LossScaler(device)
            g3 = train_step.GraphedStep(step, data['mel'], data['spec'], data['vid'], data['vid_len'], 'device')
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                g3(data['mel'], data['spec'], data['vid'])
            torch.cuda.synchronize()
            ms3 = (time.perf_counter() - t1) / args.steps * 1e3
            res['alt_precision_half'] = {'dtype': 'fp16-mixed', 'dtype_note': DTYPE_NOTE['fp16-mixed'],
                                         'ms_per_step': round(ms3, 2), 'value': round(frames / (ms3 * 1e-3), 1),
                                         'loss_scale': step.scaler.value, 'skipped_steps': step.scaler.skipped}
            step.scaler = None
            ops.set_precision(args.precision)
        if world == 1 and not args.no_cpu_baseline and not lrs:
            res['cpu_baseline'] = cpu_baseline()
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
