"""Debugging aid: does a host read between two replays of the captured step disturb the second replay?
usage: replay_probe.py <variant>   variant in none|sync|item|cpu|clone|other_item|other_alloc"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import ops, train_step  # noqa: E402

variant = sys.argv[1]
seg = len(sys.argv) > 2 and sys.argv[2] == 'seg'
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
ops.set_precision('bf16x3')
torch.manual_seed(1)
models = train_step.build_models(device=dev)
for k in models:
    models[k].train()
opts = train_step.build_optimizers(models)
step = train_step.GanStep(models, opts, 'grid')
data = bench.synth(32, 75, 96, dev, 1234)
if variant == 'dump':
    _orig = torch.cuda.CUDAGraph

    class _DebugGraph(_orig):
        def __new__(cls):
            o = super().__new__(cls)
            o.enable_debug_mode()
            return o
    torch.cuda.CUDAGraph = _DebugGraph
g = train_step.GraphedStep(step, data['mel'], data['spec'], data['vid'], data['vid_len'], data['noise'],
                           segmented=seg)


def report(tag):
    torch.cuda.synchronize()
    bad = []
    for k, v in g.static.items():
        if v.is_floating_point() and not torch.isfinite(v).all():
            bad.append('static.' + k)
    for k, v in g.out.items():
        if torch.is_tensor(v) and v.is_floating_point() and not torch.isfinite(v).all():
            bad.append('out.' + k)
    for n in step.gflat:
        if not torch.isfinite(step.gflat[n]).all():
            bad.append('grad.' + n)
    for n, m in models.items():
        if any(not torch.isfinite(p).all() for p in m.parameters()):
            bad.append('param.' + n)
    print(variant, tag, 'gen %.5f dis %.5f' % (float(g.out['gen_loss']), float(g.out['dis_loss'])), 'nonfinite:', bad,
          flush=True)


if variant == 'dump':
    os.makedirs('gpurun_out', exist_ok=True)
    g.graphs[0].debug_dump('gpurun_out/step_graph.dot')
    print('dumped', flush=True)
    sys.exit(0)


def trace():
    print('loss trace: ' + ' '.join('%s %.6f' % (k, float(v)) for k, v in sorted(g.out.items())
                                    if torch.is_tensor(v) and v.numel() == 1)[:70], flush=True)


if variant == 'bench':
    def run(n):
        out = None
        for _ in range(n):
            out = g(data['mel'], data['spec'], data['vid'], None, data['noise'])
            trace()
        return out
    run(1)
    torch.cuda.synchronize()
    run(3)
    report('bench-like')
    sys.exit(0)
if variant in ('trace_sync', 'item_sync', 'sync_item'):
    for i in range(3):
        g(data['mel'], data['spec'], data['vid'], None, data['noise'])
        if variant == 'trace_sync':
            trace()
            torch.cuda.synchronize()
        elif variant == 'item_sync':
            g.out['gen_loss'].item()
            torch.cuda.synchronize()
        else:
            torch.cuda.synchronize()
            g.out['gen_loss'].item()
    report(variant)
    sys.exit(0)

for i in range(3):
    g(data['mel'], data['spec'], data['vid'], None, data['noise'])
    if variant == 'sync':
        torch.cuda.synchronize()
    elif variant == 'item':
        g.out['gen_loss'].item()
    elif variant == 'item2':
        float(g.out['gen_loss']), float(g.out['dis_loss'])
    elif variant == 'trace':
        print('loss trace: ' + ' '.join('%s %.6f' % (k, float(v)) for k, v in sorted(g.out.items())
                                        if torch.is_tensor(v) and v.numel() == 1), file=sys.stderr)
    elif variant == 'gc':
        import gc
        print('collected', gc.collect(), flush=True)
    elif variant == 'cpu':
        g.out['gen_loss'].cpu()
    elif variant == 'clone':
        g.out['gen_loss'].clone()
    elif variant == 'other_item':
        data['mel'][0, 0, 0, 0].item()
    elif variant == 'other_alloc':
        torch.zeros(16, device=dev)
report('after 3 replays')
