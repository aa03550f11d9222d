import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import vcagan_amd
from vcagan_amd import ops, train_step
from oracle import fixtures
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16x3')
b = fixtures.make_batch(2, 20, 112, 5, [20, 13])
b = {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}
def models():
    m = fixtures.fill_models(train_step.build_models(), 300)
    for k in m: m[k].cuda().train()
    return m
for mode in ('eager', 'eager', 'graph', 'graph1'):
    m = models(); opts = train_step.build_optimizers(m); step = train_step.GanStep(m, opts, 'grid')
    hist = []
    if mode == 'eager':
        for _ in range(5):
            res = step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise']); hist.append(float(res['gen_loss']))
    else:
        wu = 2 if mode == 'graph' else 1
        gs = train_step.GraphedStep(step, b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'], warmup=wu)
        hist = ['-'] * wu
        for _ in range(5 - wu):
            res = gs(b['mel'], b['spec'], b['vid'], None, b['noise']); hist.append(float(res['gen_loss']))
    print(mode, hist)
