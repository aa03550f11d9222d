"""Run one convolution shape a few times (for rocprofv3 counter collection).
usage: conv_probe.py kind N H W Cin Cout K [stride] [prec]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import vcagan_amd
from vcagan_amd import ops, _lib
if os.environ.get('VCA_LIB'):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ['VCA_LIB'])
kind = sys.argv[1]
N, H, W, Cin, Cout, K = map(int, sys.argv[2:8])
stride = int(sys.argv[8]) if len(sys.argv) > 8 else 1
ops.set_precision(sys.argv[9] if len(sys.argv) > 9 else 'bf16x3')
dev = torch.device('cuda', 0)
x = torch.randn(N, H, W, Cin, device=dev)
w = torch.randn(Cout, Cin, K, K, device=dev) * 0.02
pad = int(os.environ.get('PAD', K // 2))
cfg = ops.ConvCfg(w.shape, x.shape, stride, (pad, pad), 0)
gy = torch.randn(cfg.yshape, device=dev)
stats = torch.zeros(32, 2, Cout, device=dev) if os.environ.get('STATS') else None   # VCA_CONV_STAT_SLOTS buffers


def run():
    if kind == 'fwd':
        ops._conv_fwd_raw(x, w, None, cfg, stats)
    elif kind == 'dgrad':
        ops._conv_dgrad_raw(gy, w, cfg)
    else:
        ops._conv_wgrad_raw(x, gy, cfg)


reps = int(os.environ.get('REPS', 3))
for _ in range(2):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * N * cfg.oh * cfg.ow * Cin * Cout * K * K
print('%s %s: %.3f ms  %.1f TF algorithmic' % (os.environ.get('VCA_LIB', 'default'), ' '.join(sys.argv[1:]), ms, fl / ms / 1e9))
