"""Run the front-end stem stage (Conv3d stem, fused BN+PReLU+max-pool) and one Visual Context Attention module a few
times at the B32 x T75 x 96x96 shape, for rocprofv3 counter collection."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa: E402,F401
from vcagan_amd import ops  # noqa: E402
from vcagan_amd.models import generator as G, visual_front as VF  # noqa: E402

ops.set_precision('bf16x3')
dev = torch.device('cuda', 0)
torch.manual_seed(0)
vf = VF.Visual_front().to(dev).train()
att = G.AVAttention(256).to(dev).train()
x = torch.randn(32, 75, 96, 96, 1, device=dev)
ph = torch.randn(32, 75, 512, device=dev)
g = torch.randn(32, 20, 75, 128, device=dev)
lens = torch.full((32,), 75, device=dev, dtype=torch.int32)
with torch.no_grad():
    for _ in range(4):
        f, st = ops.conv3d_stem(x, vf.frontend[0].weight)
        f = ops.bn_act_maxpool3(f.view(32 * 75, 48, 48, 64), vf.frontend[1], vf.frontend[2], stats=st)
        o = att(ph, g, lens)
torch.cuda.synchronize()
print('ok', tuple(f.shape), tuple(o.shape))
