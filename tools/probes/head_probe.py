"""GPU time (graph replay, no launch overhead) of the discriminator head convolutions on their 5 x 18 maps.
python tools/probes/head_probe.py [precision]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import _lib, ops  # noqa: E402

if os.environ.get('VCA_LIB'):   # A/B runs against another build of the library
    _lib.LIB_PATH = os.environ['VCA_LIB']
dev = torch.device('cuda', 0)
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'mixed')
REP = 20


def graph_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / REP * 1e3


for dim in (128, 256, 512):
    for pad in (0, 2):
        cfg = ops.ConvCfg((dim, dim, 5, 5), (32, 5, 18, dim), 1, (pad, pad))
        x = torch.randn(cfg.xshape, device=dev)
        w = torch.randn(cfg.wshape, device=dev) * 0.02
        gy = torch.randn(cfg.yshape, device=dev)
        gf = 2.0 * gy.numel() * dim * 25 / 1e9
        for kind, fn in (('fwd', lambda: ops._conv_fwd_raw(x, w, None, cfg)), ('dgrad', lambda: ops._conv_dgrad_raw(gy, w, cfg)),
                         ('wgrad', lambda: ops._conv_wgrad_raw(x, gy, cfg))):
            t = graph_time(fn)
            print('dim %3d pad %d %-5s %7.1f us  %6.1f TF/s' % (dim, pad, kind, t, gf / t * 1e3), flush=True)

print('-- cost per filter tap (unpadded, 5 columns, 1 / 3 / 5 rows)')
for dim in (128, 512):
    for kh in (1, 3, 5):
        cfg = ops.ConvCfg((dim, dim, kh, 5), (32, 5, 18, dim), 1, (0, 0))
        x = torch.randn(cfg.xshape, device=dev)
        w = torch.randn(cfg.wshape, device=dev) * 0.02
        gy = torch.randn(cfg.yshape, device=dev)
        for kind, fn in (('fwd', lambda: ops._conv_fwd_raw(x, w, None, cfg)), ('dgrad', lambda: ops._conv_dgrad_raw(gy, w, cfg))):
            print('dim %3d %dx5 %-5s %7.1f us' % (dim, kh, kind, graph_time(fn)), flush=True)
