"""Run one pointwise product on the tiled kernel a few times (for rocprofv3 counter collection).
usage: gemm_one.py M N K [prec]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vcagan_amd  # noqa: E402,F401
from vcagan_amd import ops  # noqa: E402

M, N, K = map(int, sys.argv[1:4])
ops.set_precision(sys.argv[4] if len(sys.argv) > 4 else 'bf16x3')
ns = 3 if ops.get_precision().endswith('x3') else 1
dev = torch.device('cuda', 0)
x = torch.randn(M, K, device=dev)
w = torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.02)
bias = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev)
wp = ops._pack(w, 10)
for _ in range(5):
    ops._linear_raw(x, wp, bias, None, out, M, N, K, ns, tiled=True)
torch.cuda.synchronize()
print('done', M, N, K, ops.get_precision())
