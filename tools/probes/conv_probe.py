"""Run one convolution shape a few times (for rocprofv3 counter collection).
usage: conv_probe.py kind N H W Cin Cout K [stride] [prec]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import vcagan_amd
from vcagan_amd import ops
kind = sys.argv[1]
N, H, W, Cin, Cout, K = map(int, sys.argv[2:8])
stride = int(sys.argv[8]) if len(sys.argv) > 8 else 1
ops.set_precision(sys.argv[9] if len(sys.argv) > 9 else 'bf16x3')
dev = torch.device('cuda', 0)
x = torch.randn(N, H, W, Cin, device=dev)
w = torch.randn(Cout, Cin, K, K, device=dev) * 0.02
cfg = ops.ConvCfg(w.shape, x.shape, stride, (K // 2, K // 2), 0)
gy = torch.randn(cfg.yshape, device=dev)
for _ in range(3):
    if kind == 'fwd':
        ops._conv_fwd_raw(x, w, None, cfg)
    elif kind == 'dgrad':
        ops._conv_dgrad_raw(gy, w, cfg)
    else:
        ops._conv_wgrad_raw(x, gy, cfg)
torch.cuda.synchronize()
