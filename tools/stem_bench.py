"""Time the dedicated Conv3d-stem kernel at the benchmark shape (B32 x T75 x 96x96) against the HBM roofline.
Algorithmic bytes = input (fp32) + packed weights + output (fp32)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa
from vcagan_amd import ops

B, T, H = 32, 75, 96
dev = torch.device('cuda', 0)
x = torch.randn(B, T, H, H, 1, device=dev)
w = torch.nn.Parameter(torch.randn(64, 1, 5, 7, 7, device=dev) * 0.05)
nbytes = x.numel() * 4 + 73728 + B * T * (H // 2) ** 2 * 64 * 4
flops = 2.0 * B * T * (H // 2) ** 2 * 64 * 245
for prec in ('bf16x3', 'bf16', 'fp32'):
    ops.set_precision(prec)
    with torch.no_grad():
        for _ in range(2):
            ops.conv3d_stem(x, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv3d_stem(x, w)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('%-7s %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)  %.1f TFLOP/s algorithmic' %
          (prec, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, flops / ms / 1e9))
