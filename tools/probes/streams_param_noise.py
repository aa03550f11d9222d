"""Per-parameter spread of the first step's gradients between runs, with and without the side streams: the largest
per-parameter relative L2 deviation (a lost or doubled contribution of one launch would stand out here while hiding
in the norm over all parameters).   python tools/probes/streams_param_noise.py [precision]"""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')
import vcagan_amd  # noqa
from oracle import fixtures
from vcagan_amd import ops, train_step

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'


def run(streams):
    ops.set_precision(prec)
    train_step.D_STREAMS = streams
    m = fixtures.fill_models(train_step.build_models(), 300)
    for k in m:
        m[k].cuda().train()
    step = train_step.GanStep(m, train_step.build_optimizers(m), 'grid')
    b = fixtures.make_batch(2, 20, 96, 5, [20, 13])
    d = {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}
    step(d['mel'], d['spec'], d['vid'], d['vid_len'], noise=d['noise'])
    torch.cuda.synchronize()
    return {k + '.' + n: p.grad.detach().clone() for k in m for n, p in m[k].named_parameters() if p.grad is not None}


base = run(False)
for name, s in (('one stream', False), ('one stream', False), ('streams', True), ('streams', True), ('streams', True),
                ('streams', True)):
    g = run(s)
    worst = []
    top = {}
    for k in base:
        top[k.split('.')[0]] = max(top.get(k.split('.')[0], 0.0), float(base[k].double().norm()))
    for k in base:
        den = float(base[k].double().norm())
        if den < 1e-3 * top[k.split('.')[0]]:      # gradients that are zero up to rounding (a bias in front of a
            continue                               # batch-norm) or negligible beside the network's others
        worst.append((float((g[k].double() - base[k].double()).norm()) / den, k))
    worst.sort(reverse=True)
    side = {'D': [w for w in worst if w[1].split('.')[0] in ('dis1', 'dis2', 'dis3', 's_dis')],
            'G': [w for w in worst if w[1].split('.')[0] in ('v_front', 'gen', 'post')]}
    print('%s %-10s  D worst %.2e (%s)  G worst %.2e (%s)' % (prec, name, side['D'][0][0], side['D'][0][1],
                                                              side['G'][0][0], side['G'][0][1]), flush=True)
