"""Where do the ATen kernels of a training step come from?  Runs one eager step at the bench shape under
torch.profiler and groups every aten:: operator that launches a kernel by (operator, input shapes), with launch counts
and device time; a second eager step under a dispatch mode records, per (operator, input shapes), the nearest Python
frames of this package that issued it (operators issued by the autograd engine's own nodes have none).

    python tools/aten_sources.py [--batch 32] [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--top', type=int, default=60)
    args = ap.parse_args()
    import vcagan_amd  # noqa: F401
    from vcagan_amd import ops, train_step
    ops.set_precision('mixed')
    dev = torch.device('cuda', 0)
    torch.manual_seed(1)
    m = train_step.build_models(device=dev)
    for k in m:
        m[k].train()
    step = train_step.GanStep(m, train_step.build_optimizers(m), 'grid')
    d = bench.synth(args.batch, bench.T, bench.H, dev, 1)
    for _ in range(2):
        step(d['mel'], d['spec'], d['vid'], d['vid_len'], noise='device')
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step(d['mel'], d['spec'], d['vid'], d['vid_len'], noise='device')
        torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sources = collections.defaultdict(collections.Counter)

    class Where(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = 'aten::' + func.__name__.split('.')[0]
            shapes = str([tuple(a.shape) for a in args if isinstance(a, torch.Tensor) and a.dim() > 0])
            where = [('%s:%d' % (os.path.basename(f.filename), f.lineno)) for f in traceback.extract_stack()
                     if 'visual-context-attentional-gan_amd' in f.filename]
            sources[(name, shapes)][' < '.join(reversed(where[-2:])) or '(autograd engine)'] += 1
            return func(*args, **(kwargs or {}))

    with Where():
        step(d['mel'], d['spec'], d['vid'], d['vid_len'], noise='device')
        torch.cuda.synchronize()
    groups = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    for ev in prof.events():
        if not ev.name.startswith('aten::'):
            continue
        t = getattr(ev, 'self_device_time_total', 0) or 0
        if t <= 0:
            continue
        shapes = str([tuple(x) for x in (ev.input_shapes or []) if x])
        key = (ev.name, shapes)
        groups[key][0] += 1
        groups[key][1] += t
        total += t
    print('ATen device time per step: %.3f ms in %d launches' % (total / 1e3, sum(v[0] for v in groups.values())))
    for (name, shapes), (n, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:args.top]:
        frame = '; '.join('%s x%d' % kv for kv in sources.get((name, shapes), collections.Counter()).most_common(3))
        print('%8.1f us  n=%3d  %-14s %-50s %s' % (t, n, name, shapes[:50], frame[:150]))
    print('\ncopies as issued (clone / copy_ / contiguous), by source:')
    for (name, shapes), c in sorted(sources.items()):
        if name.split('::')[1] in ('clone', 'copy_', 'contiguous', '_to_copy'):
            print('   %-14s %-50s %s' % (name, shapes[:50], '; '.join('%s x%d' % kv for kv in c.most_common(4))[:200]))


if __name__ == '__main__':
    main()
