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
from parity_util import load, measure_d_side, measure_step  # noqa: E402
from vcagan_amd import ops, train_step  # noqa: E402

D = ('dis1', 'dis2', 'dis3', 's_dis')
for prec in (sys.argv[1:] or ['fp32', 'bf16x3', 'mixed', 'mixed+r1x3']):
    for variant in ('grid', 'lrs'):
        # 'mixed+r1x3': mixed with the second-order convolutions of the R1 penalty on three products
        ops.R1_SECOND_ORDER_SPLIT[0] = prec.endswith('+r1x3')
        ops.set_precision(prec.split('+')[0])
        g = load('step_' + variant)
        B, T, H = int(g['B']), int(g['T']), int(g['H'])
        m = fixtures.fill_models(train_step.build_models(), int(g['wseed']))
        for k in m:
            m[k].cuda().train()
        step = train_step.GanStep(m, train_step.build_optimizers(m, amsgrad=(variant == 'grid')), variant)
        for s in range(int(g['nsteps'])):
            b = fixtures.make_batch(B, T, H, int(g['seed']) + s, list(g['lens']))
            b = {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}
            # the step phase by phase, so that the gradients d_optimizer.step() consumes can be read in between
            sc = step.scaler.value if step.scaler is not None else 1.0   # half modes: gradients carry the loss scale
            opts = (step.g_opt, step.d_opt)
            step.d_phase(b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'])
            torch.cuda.synchronize()
            for o in opts:
                o.all_grads.mul_(1.0 / sc)
            dm = measure_d_side(m, step._st['phon_d'].grad / sc, g, s)
            for o in opts:
                o.all_grads.mul_(sc)
            step.d_exchange.reduce()
            step.g_phase()
            step.g_exchange.reduce()
            res = step.finish()
            for o in opts:
                o.all_grads.mul_(1.0 / sc)
            res = {k: (v.item() if v.numel() == 1 else v.cpu()) for k, v in res.items()}
            # D: after the step the reference's discriminators hold the gradient of gen_loss, which nothing consumes
            mt = measure_step(m, res, g, s, D)
            worst = lambda pre: max(v for k, v in mt.items() if k.startswith(pre))
            print('%-7s %-4s step %d: loss %.2e  out_max %.2e  mel_l1 %.2e  grad_rel %.3f  gradnorm %.3f  '
                  'param_max %.2e p90 %.2e' % (prec, variant, s, worst('loss:'), worst('out_max:'), worst('out_l1:'),
                                               worst('grad_rel:'), worst('gradnorm_rel:'), mt['param_max'],
                                               mt['param_p90']), flush=True)
            print('%-7s %-4s step %d: D side  ' % (prec, variant, s) +
                  '  '.join('%s %.4f' % (k.replace('dgrad_rel:', '').replace('dgradnorm_rel:', 'norm:'), v)
                            for k, v in sorted(dm.items())), flush=True)
