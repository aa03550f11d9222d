"""Weight-gradient kernel time on layer shapes of the training step (single-product bf16).
python tools/probes/wgrad_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import _lib, ops  # noqa: E402
from vcagan_amd._lib import ptr, stream, check  # noqa: E402

if os.environ.get('VCA_LIB'):   # A/B runs against another build of the library
    _lib.LIB_PATH = os.environ['VCA_LIB']
dev = torch.device('cuda', 0)
L = _lib.lib()


def timeit(fn, n=20):
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


# (N, H, W, Cin, Cout, K): decoder 5x5 blocks, lip-reading ResNet 3x3 blocks
SHAPES = ((32, 20, 75, 512, 512, 5), (32, 40, 150, 256, 256, 5), (32, 80, 300, 128, 128, 5),
          (2400, 24, 24, 64, 64, 3), (2400, 12, 12, 128, 128, 3), (2400, 6, 6, 256, 256, 3))
for n, h, w, cin, cout, k in SHAPES:
    cfg = ops.ConvCfg((cout, cin, k, k), (n, h, w, cin), 1, (k // 2, k // 2))
    x = torch.randn(n, h, w, cin, device=dev)
    gy = torch.randn(n, h, w, cout, device=dev)
    dwp = torch.zeros(k * k, cin, cout, device=dev)
    for ns in (1, 3):
        t = timeit(lambda: check(L.vca_conv_wgrad_bf16(cfg.desc(), ns, ptr(x), ptr(gy), None, None, None, ptr(dwp),
                                                       None, stream()), 'wgrad'))
        fl = 2.0 * n * h * w * cin * cout * k * k
        print('%dx%dx%d %d->%d %dx%d  nsplit %d  %.3f ms  %.1f TFLOP/s (useful)  reads %.2f TB/s (one pass)' % (
            n, h, w, cin, cout, k, k, ns, t, fl / t / 1e9, (x.numel() + gy.numel()) * 4 / t / 1e9), flush=True)
