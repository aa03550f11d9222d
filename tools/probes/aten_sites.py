"""Debugging aid: which Python call sites issue ATen kernels (anything that is not a view) during one eager training
step, with call counts and element totals.   python tools/probes/aten_sites.py [batch]"""
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
ops.set_precision('mixed')
torch.manual_seed(1)
models = train_step.build_models(device=dev)
for k in models:
    models[k].train()
opts = train_step.build_optimizers(models)
step = train_step.GanStep(models, opts, 'grid')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
data = bench.synth(B, 75, 96, dev, 1234)
step(data['mel'], data['spec'], data['vid'], data['vid_len'], noise='device')
sites = collections.Counter()
elems = collections.Counter()
VIEWS = ('view', 'reshape', 'permute', 'transpose', 'select', 'slice', 'expand', 'detach', 'alias', 'as_strided',
         'unsqueeze', 'squeeze', 't.default', 'unbind', 'split', 'narrow', '_unsafe_view', 'empty', 'is_', 'size',
         'stride', 'numel', 'dim', 'lift_fresh', '_local_scalar', 'item', 'set_', 'resize', 'record_stream', 'unfold',
         'chunk', 'movedim', 'flatten', 'view_as', 'result_type', 'sym_', 'prim.', 'new_empty', 'storage_offset',
         '_has_compatible')


class Mode(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            fr = [f for f in traceback.extract_stack() if ('visual-context' in f.filename or 'vcagan' in f.filename)
                  and 'aten_sites' not in f.filename]
            key = name + ' @ ' + (' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-3:]) or
                                  '(autograd engine) ' + str([tuple(a.shape) for a in args if torch.is_tensor(a)][:1]))
            n = max([a.numel() for a in args if torch.is_tensor(a)] + [0])
            sites[key] += 1
            elems[key] += n
        return func(*args, **(kwargs or {}))


with Mode():
    step(data['mel'], data['spec'], data['vid'], data['vid_len'], noise='device')
torch.cuda.synchronize()
print('total aten kernels per step:', sum(sites.values()))
for k, v in sorted(sites.items(), key=lambda kv: -elems[kv[0]]):
    print('%5d calls %12d elems  %s' % (v, elems[k], k))
