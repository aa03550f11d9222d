"""Debugging aid: capture a forward-only piece of the step (idempotent), replay, read a value on the host and
synchronise, replay again and compare the two results bit for bit.
usage: replay_probe2.py <piece>   piece in vfront|gen|dis|post|sdis"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import ops, train_step  # noqa: E402

piece = sys.argv[1]
grad = len(sys.argv) > 2 and sys.argv[2] == 'grad'
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
ops.set_precision('bf16x3')
torch.manual_seed(1)
models = train_step.build_models(device=dev)
for k in models:
    models[k].train()
data = bench.synth(32, 75, 96, dev, 1234)
phon0 = torch.randn(32, 75, 512, device=dev)
sent0 = torch.randn(32, 512, 75, device=dev)
lens = torch.full((32,), 75, device=dev, dtype=torch.int32)


def fn():
    if piece == 'vfront':
        out = list(models['v_front'](data['vid']))
    elif piece == 'gen':
        out = list(models['gen'](sent0, phon0, lens, noise=data['noise']))
    elif piece == 'dis':
        out = list(models['dis3'](data['mel'], sent0, 75))
    elif piece == 'post':
        out = [models['post'](data['mel'])]
    elif piece == 'sdis':
        out = [models['s_dis'](phon0, data['mel'])]
    if grad:
        params = [p for m in models.values() for p in m.parameters()]
        for p in params:
            p.grad = None
        with ops.direct_param_grad():
            sum(o.sum() for o in out).backward()
        out = out + [p.grad for p in params if p.grad is not None]
    return [o.detach() for o in out]


inter = []


def hook(name):
    def h(mod, inp, out):
        if capturing:
            for i, o in enumerate(out if isinstance(out, (tuple, list)) else [out]):
                if torch.is_tensor(o):
                    inter.append((name + ':%d' % i, o.detach()))
    return h


capturing = False
for mn, mod in models.items():
    for n, sub in mod.named_modules():
        sub.register_forward_hook(hook(mn + '.' + n))

ctxm = torch.enable_grad() if grad else torch.no_grad()
with ctxm:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    capturing = True
    with torch.cuda.graph(g):
        outs = fn()
    capturing = False
names = ['out%d' % i for i in range(len(outs))] + [n for n, _ in inter]
outs = outs + [t for _, t in inter]
g.replay()
torch.cuda.synchronize()
ref = [o.clone() for o in outs]
g.replay()
torch.cuda.synchronize()
def close(a, b):
    return bool(torch.isfinite(b).all()) and float((a - b).abs().max()) <= 1e-3 * (1e-6 + float(a.abs().max()))


same = all(close(a, b) for a, b in zip(ref, outs))
g.replay()
outs[0].flatten()[0].item()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
bad = [names[i] for i, (a, b) in enumerate(zip(ref, outs)) if not close(a, b)]
nan = [i for i, b in enumerate(outs) if not torch.isfinite(b).all()]
print(piece, 'grad' if grad else 'fwd', 'plain replay close:', same, '| after item+sync: differing outputs', bad[:12],
      'of', len(outs), 'nonfinite', nan[:10], flush=True)
