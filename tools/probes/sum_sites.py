"""Debugging aid: which Python call sites issue torch reductions (sum/mean) during one eager training step."""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import ops, train_step  # noqa: E402

dev = torch.device('cuda', 0)
ops.set_precision('bf16x3')
torch.manual_seed(1)
models = train_step.build_models(device=dev)
for k in models:
    models[k].train()
opts = train_step.build_optimizers(models)
step = train_step.GanStep(models, opts, 'grid')
data = bench.synth(4, 75, 96, dev, 1234)
step(data['mel'], data['spec'], data['vid'], data['vid_len'], noise=data['noise'])
sites = collections.Counter()


class Mode(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if 'sum' in name or 'mean' in name:
            fr = [f for f in traceback.extract_stack() if 'visual-context' in f.filename or 'vcagan' in f.filename]
            key = name + ' @ ' + (' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-2:]) or '(autograd)')
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)]
            sites[key + ' ' + str(shapes)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    step(data['mel'], data['spec'], data['vid'], data['vid_len'], noise=data['noise'])
torch.cuda.synchronize()
for k, v in sites.most_common():
    print(v, k)
