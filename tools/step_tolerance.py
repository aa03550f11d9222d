"""Measured deviations of the HIP step from the golden step fixtures, per precision and step (GPU box only): the
numbers behind the bounds of tests/test_models_gpu.py::test_step_matches_reference.
    python tools/step_tolerance.py [precision ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import vcagan_amd  # noqa: E402,F401
from oracle import fixtures  # noqa: E402
from parity_util import load, measure_step  # noqa: E402
from vcagan_amd import ops, train_step  # noqa: E402

D = ('dis1', 'dis2', 'dis3', 's_dis')
for prec in (sys.argv[1:] or ['fp32', 'bf16x3', 'mixed']):
    for variant in ('grid', 'lrs'):
        ops.set_precision(prec)
        g = load('step_' + variant)
        B, T, H = int(g['B']), int(g['T']), int(g['H'])
        m = fixtures.fill_models(train_step.build_models(), int(g['wseed']))
        for k in m:
            m[k].cuda().train()
        step = train_step.GanStep(m, train_step.build_optimizers(m, amsgrad=(variant == 'grid')), variant)
        for s in range(int(g['nsteps'])):
            b = fixtures.make_batch(B, T, H, int(g['seed']) + s, list(g['lens']))
            b = {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}
            res = step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
            res = {k: (v.item() if v.numel() == 1 else v.cpu()) for k, v in res.items()}
            mt = measure_step(m, res, g, s, D)
            worst = lambda pre: max(v for k, v in mt.items() if k.startswith(pre))
            print('%-7s %-4s step %d: loss %.2e  out_max %.2e  mel_l1 %.2e  grad_rel %.3f  gradnorm %.3f  '
                  'param_max %.2e p90 %.2e' % (prec, variant, s, worst('loss:'), worst('out_max:'), worst('out_l1:'),
                                               worst('grad_rel:'), worst('gradnorm_rel:'), mt['param_max'],
                                               mt['param_p90']), flush=True)
