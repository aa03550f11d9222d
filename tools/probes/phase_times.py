"""Time of the step's three phases (three per-phase graphs replayed one after the other, HIP events around each).
python tools/probes/phase_times.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import ops, train_step  # noqa: E402

ops.set_precision('mixed')
dev = torch.device('cuda', 0)
torch.manual_seed(1)
m = train_step.build_models(device=dev)
for k in m:
    m[k].train()
step = train_step.GanStep(m, train_step.build_optimizers(m), 'grid')
d = bench.synth(32, bench.T, bench.H, dev, 1)
gs = train_step.GraphedStep(step, d['mel'], d['spec'], d['vid'], d['vid_len'], 'device', segmented=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tot = [0.0, 0.0, 0.0]
N = 10
for it in range(N + 2):
    ev[0].record()
    for i, g in enumerate(gs.graphs):
        g.replay()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if it >= 2:
        for i in range(3):
            tot[i] += ev[i].elapsed_time(ev[i + 1])
print('streams', train_step.D_STREAMS, 'D phase %.2f ms  G phase %.2f ms  finish %.2f ms  total %.2f' % (
    tot[0] / N, tot[1] / N, tot[2] / N, sum(tot) / N))
