import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import vcagan_amd
from vcagan_amd import ops
B, T, H = 32, 75, 96
dev = torch.device('cuda', 0)
for name, x, w in (('random', torch.randn(B, T, H, H, 1, device=dev), torch.randn(64, 1, 5, 7, 7, device=dev) * 0.05),
                   ('zeros', torch.zeros(B, T, H, H, 1, device=dev), torch.zeros(64, 1, 5, 7, 7, device=dev))):
    w = torch.nn.Parameter(w)
    for prec in ('bf16x3', 'bf16'):
        ops.set_precision(prec)
        with torch.no_grad():
            for _ in range(3):
                ops.conv3d_stem(x, w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv3d_stem(x, w)
            e1.record()
            torch.cuda.synchronize()
        print(name, prec, '%.3f ms' % (e0.elapsed_time(e1) / 20), flush=True)
