"""Achieved HBM bandwidth of the batch-norm / activation passes on two map sizes of the training step.
python tools/probes/pointwise_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import _lib  # noqa: E402
from vcagan_amd._lib import ptr, stream, check  # noqa: E402

dev = torch.device('cuda', 0)
L = _lib.lib()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for P, C in ((2400 * 24 * 24, 64), (32 * 20 * 75, 512), (2400 * 12 * 12, 128), (32 * 80 * 300, 32)):
    x, gy = torch.randn(P, C, device=dev), torch.randn(P, C, device=dev)
    y, gx = torch.empty_like(x), torch.empty_like(x)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    mean, inv = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
    sums = torch.zeros(16 * 3 * C, device=dev)
    nb = P * C * 4
    t = timeit(lambda: check(L.vca_affine_act_f32(P, C, ptr(x), ptr(sc), ptr(sh), None, None, 1, 0.2, 1.0, ptr(y), stream()), 'a'))
    print('P=%d C=%d  affine_act      %.3f ms  %.2f TB/s (8 B/el)' % (P, C, t, 2 * nb / t / 1e9))
    t2 = timeit(lambda: (sums.zero_(), check(L.vca_affine_act_bwd_f32(P, C, ptr(gy), ptr(x), ptr(sc), ptr(sh), None, None, ptr(mean), ptr(inv), 1, 0.2, 1.0, 1, ptr(sums), None, None, stream()), 'r')))
    print('             bwd reduce      %.3f ms  %.2f TB/s (8 B/el)' % (t2, 2 * nb / t2 / 1e9))
    t3 = timeit(lambda: (sums.zero_(), check(L.vca_affine_act_bwd_f32(P, C, ptr(gy), ptr(x), ptr(sc), ptr(sh), None, None, ptr(mean), ptr(inv), 1, 0.2, 1.0, 1, ptr(sums), ptr(gx), None, stream()), 'b')))
    print('             bwd apply       %.3f ms  %.2f TB/s (12 B/el)' % (t3 - t2, 3 * nb / (t3 - t2) / 1e9))
    t4 = timeit(lambda: y.copy_(x))
    print('             torch copy      %.3f ms  %.2f TB/s (8 B/el)' % (t4, 2 * nb / t4 / 1e9), flush=True)
