"""Probe: the attention query projection (generator.py:159-160) as an (F x 1) convolution over the channels-last decoder
map (no permute copy) against permute + linear, forward / input gradient / weight gradient, device time under replay."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa: E402,F401
import bench  # noqa: E402
from vcagan_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'mixed')
for (B, Fq, T, C) in ((32, 20, 75, 128), (32, 40, 150, 64)):
    g = torch.randn(B, Fq, T, C, device=dev)
    w = torch.nn.Parameter(torch.randn(256, C * Fq, device=dev) * 0.02)
    bias = torch.zeros(256, device=dev)
    gq = torch.randn(B, T, 256, device=dev)

    def lin():
        qin = g.permute(0, 2, 3, 1).reshape(B, T, C * Fq)
        return ops.linear(qin, w, bias)

    def conv():
        return ops.conv2d(g, w.view(256, C, Fq, 1), bias, 1, 0)
    with torch.no_grad():
        a, b = lin(), conv().view(B, T, 256)
        print((B, Fq, T, C), 'max diff', float((a - b).abs().max()), 'of', float(a.abs().max()))
        print('  fwd  permute+linear %.1f us   conv form %.1f us' % (bench._time_graph(lin) * 1e6, bench._time_graph(conv) * 1e6))
        cfg = ops.ConvCfg((256, C, Fq, 1), g.shape, 1, (0, 0), 0)
        gy = gq.view(B, 1, T, 256).contiguous()
        qin = g.permute(0, 2, 3, 1).reshape(B * T, 1, 1, C * Fq).contiguous()
        lcfg = ops.ConvCfg((256, C * Fq, 1, 1), qin.shape, 1, (0, 0), 0)
        gyl = gq.view(B * T, 1, 1, 256).contiguous()
        wv = w.view(256, C, Fq, 1)
        wl = w.view(256, C * Fq, 1, 1)
        for name, fc, fl in (('dgrad', lambda: ops._conv_dgrad_raw(gy, wv, cfg), lambda: ops._conv_dgrad_raw(gyl, wl, lcfg)),
                             ('wgrad', lambda: ops._conv_wgrad_raw(g, gy, cfg), lambda: ops._conv_wgrad_raw(qin, gyl, lcfg))):
            try:
                tc = bench._time_graph(fc) * 1e6
            except Exception as e:
                tc = float('nan')
                print('  conv-form', name, 'failed:', repr(e)[:120])
            print('  %s linear form %.1f us   conv form %.1f us' % (name, bench._time_graph(fl) * 1e6, tc))
