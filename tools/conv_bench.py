"""Per-layer timing of the convolution kernels on the shapes of one B32xT75x96x96 GAN step (GPU box only).
Records every distinct (geometry) that ops.conv2d/conv1d/linear sees during one step, then times forward,
input-gradient and weight-gradient of each with HIP events.   python tools/conv_bench.py [precision] [batch]"""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa
from vcagan_amd import ops, train_step
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ops.set_precision(prec)
dev = torch.device('cuda', 0)
seen = collections.OrderedDict()
orig = {k: getattr(ops, k) for k in ('_conv_fwd_raw', '_conv_dgrad_raw', '_conv_wgrad_raw')}


def rec(kind):
    def f(a, b, *rest):
        cfg = [r for r in rest if isinstance(r, ops.ConvCfg)][0]
        key = (kind, cfg.xshape, cfg.wshape, cfg.stride, cfg.pad_h, cfg.pad_w, cfg.ups)
        seen[key] = seen.get(key, 0) + 1
        return orig['_conv_%s_raw' % kind](a, b, *rest)
    return f


for k in ('fwd', 'dgrad', 'wgrad'):
    setattr(ops, '_conv_%s_raw' % k, rec(k))
torch.manual_seed(1)
models = train_step.build_models(device=dev)
for m in models.values():
    m.train()
opts = train_step.build_optimizers(models)
step = train_step.GanStep(models, opts, 'grid')
data = bench.synth(batch, bench.T, bench.H, dev, 1)
step(data['mel'], data['spec'], data['vid'], data['vid_len'], noise=data['noise'])
torch.cuda.synchronize()
for k in ('fwd', 'dgrad', 'wgrad'):
    setattr(ops, '_conv_%s_raw' % k, orig['_conv_%s_raw' % k])

rows = []
for (kind, xs, ws, stride, ph, pw, ups), cnt in seen.items():
    cfg = ops.ConvCfg(ws, xs, stride, (ph, pw), ups, pad_d=(2 if len(xs) == 5 else 0))
    x = torch.randn(xs, device=dev)
    w = torch.randn(ws, device=dev) * 0.02
    gy = torch.randn(cfg.yshape, device=dev)
    fn = {'fwd': lambda: ops._conv_fwd_raw(x, w, None, cfg), 'dgrad': lambda: ops._conv_dgrad_raw(gy, w, cfg),
          'wgrad': lambda: ops._conv_wgrad_raw(x, gy, cfg)}[kind]
    fn(); fn()
    torch.cuda.synchronize()
    # ten launches captured into a graph and replayed: device time per launch without the host's launch overhead
    # (issued from Python, anything shorter than ~25 us measures the interpreter, not the kernel)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    del gr
    taps = cfg.kd * cfg.kh * cfg.kw
    P = 1
    for s in cfg.yshape[:-1]:
        P *= s
    gf = 2.0 * P * cfg.cin * taps * cfg.cout / 1e9
    rows.append((ms * cnt, kind, xs, ws, stride, ups, cnt, ms, gf, gf / ms))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total conv ms/step %.1f' % tot)
for r in rows[:int(os.environ.get("CONV_BENCH_ROWS", "60"))]:
    print('%7.2fms %-5s x=%-22s w=%-20s s%d u%d n=%2d %7.3fms %8.1fGF %7.1f TF/s' % r)
