"""Probe: device time (graph replay) of the larger linear-layer shapes of a step on the tiled kernel, the skinny kernel
and the convolution kernel, three products and one."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa: E402,F401
import bench  # noqa: E402
from vcagan_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
shapes = [(2400, 512, 5120), (2400, 1536, 1024), (2400, 1536, 512), (2400, 256, 2560), (4800, 256, 2560),
          (2400, 1280, 256), (4800, 1280, 256), (2400, 256, 512), (48000, 256, 512), (2400, 512, 1024)]
for prec in ('bf16x3', 'bf16'):
    ops.set_precision(prec)
    ns = 3 if prec == 'bf16x3' else 1
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.02)
        bias = torch.zeros(N, device=dev)
        out = torch.empty(M, N, device=dev)
        wp = ops._pack(w, 10)
        cfg = ops.ConvCfg((N, K, 1, 1), (M, 1, 1, K), 1, (0, 0), 0)
        x4 = x.view(M, 1, 1, K)
        with torch.no_grad():
            t_t = bench._time_graph(lambda: ops._linear_raw(x, wp, bias, None, out, M, N, K, ns, tiled=True)) * 1e6
            t_s = bench._time_graph(lambda: ops._linear_raw(x, wp, bias, None, out, M, N, K, ns, tiled=False)) * 1e6
            ops.LINEAR_KERNEL = False
            t_c = bench._time_graph(lambda: ops._conv_fwd_raw(x4, w, bias, cfg)) * 1e6
            ops.LINEAR_KERNEL = True
        gf = 2.0 * M * N * K / 1e9
        print('%-6s M=%5d N=%4d K=%4d %5.1f GF  tiled %6.1f us (%4.0f TF/s issued)  skinny %6.1f  conv %6.1f' %
              (prec, M, N, K, gf, t_t, gf * ns / t_t * 1e3, t_s, t_c), flush=True)
