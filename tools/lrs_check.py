"""One LRS-variant step at the BASELINE configs[3] shape (B16, T up to 150, 96x96, variable lengths): finite losses,
timing.   python tools/lrs_check.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa
from vcagan_amd import ops, train_step
import bench

ops.set_precision('bf16x3')
dev = torch.device('cuda', 0)
B, T, H = 16, 150, 96
torch.manual_seed(1)
models = train_step.build_models(device=dev)
for m in models.values():
    m.train()
opts = train_step.build_optimizers(models, amsgrad=False)
step = train_step.GanStep(models, opts, 'lrs')
d = bench.synth(B, T, H, dev, 3)
g = torch.Generator().manual_seed(0)
lens = [int(x) for x in torch.randint(T // 3, T + 1, (B,), generator=g)]
lens[0] = T
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = step(d['mel'], d['spec'], d['vid'], lens, noise=d['noise'])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vals = {k: float(v) for k, v in out.items() if v.numel() == 1}
    assert all(v == v and abs(v) < 1e6 for v in vals.values()), vals
    print('step %d: %.1f ms (eager)  gen %.4f dis %.4f recon %.4f' % (i, dt * 1e3, vals['gen_loss'], vals['dis_loss'], vals['recon_loss']))
print('frames/s (eager, B16xT150): %.0f' % (B * T / dt))
