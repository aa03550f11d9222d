"""Probe: the dominant convolution (decode.0.conv2, 5x5 512->512 on 32x20x75) on random and on all-zero operands,
forward (three products) and input gradient (one product).  Same instruction stream, same cycles: a gap between the two
is the clock the chip holds under the data's switching activity, not something the kernel's schedule can recover."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
B, T = 32, 75
flops = 2.0 * B * 20 * T * 512 * 25 * 512


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, mk in (('random', torch.randn), ('zeros', torch.zeros)):
    x = mk(B, 20, T, 512, device=dev)
    w = torch.nn.Parameter(mk(512, 512, 5, 5, device=dev) * 0.01)
    bias = torch.zeros(512, device=dev)
    for prec in ('bf16x3', 'bf16'):
        ops.set_precision(prec)
        with torch.no_grad():
            ms = t(lambda: ops.conv2d(x, w, bias, 1, 2))
        print('%-6s %-6s fwd  %.3f ms  %.0f TFLOP/s algorithmic, %.0f issued' %
              (name, prec, ms, flops / ms / 1e9, flops / ms / 1e9 * (3 if prec == 'bf16x3' else 1)), flush=True)
