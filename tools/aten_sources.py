"""Where do the ATen kernels of a training step come from?  Runs one eager step at the bench shape under
torch.profiler and groups every aten:: operator that launches a kernel by (operator, input shapes, nearest frame of
this package), with launch counts and device time.

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
    groups = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    for ev in prof.events():
        if not ev.name.startswith('aten::'):
            continue
        t = getattr(ev, 'self_device_time_total', 0) or 0
        if t <= 0:
            continue
        frame = '(autograd engine)'
        for s in ev.stack or []:
            if 'visual-context-attentional-gan_amd' in s or 'vcagan_amd' in s:
                frame = s.split('visual-context-attentional-gan_amd/')[-1]
                break
        shapes = str([tuple(x) for x in (ev.input_shapes or []) if x])
        key = (ev.name, shapes, frame)
        groups[key][0] += 1
        groups[key][1] += t
        total += t
    print('ATen device time per step: %.3f ms in %d launches' % (total / 1e3, sum(v[0] for v in groups.values())))
    for (name, shapes, frame), (n, t) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:args.top]:
        print('%8.1f us  n=%3d  %-14s %-60s %s' % (t, n, name, shapes[:60], frame[:90]))


if __name__ == '__main__':
    main()
