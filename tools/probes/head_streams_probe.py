"""Probe (GPU box): captures with the discriminators' conditional heads on helper streams, under faulthandler.
    VCA_D_HEAD_STREAMS=1 [VCA_D_STREAMS=0] [PROBE_MODE=step|dis_fwd|dis_fwdbwd] python tools/probes/head_streams_probe.py"""
import faulthandler
import os
import sys

faulthandler.enable(all_threads=True)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vcagan_amd  # noqa: E402,F401
from oracle import fixtures  # noqa: E402
from vcagan_amd import ops, train_step  # noqa: E402
from vcagan_amd.models import generator as G  # noqa: E402

mode = os.environ.get('PROBE_MODE', 'step')
print('mode', mode, 'HEAD_STREAMS', G.HEAD_STREAMS, 'D_STREAMS', train_step.D_STREAMS, flush=True)
ops.set_precision('mixed')
B, T, H = int(os.environ.get('PROBE_B', 2)), int(os.environ.get('PROBE_T', 20)), 112
b = fixtures.make_batch(B, T, H, 5, [T] * B)
b = {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}
m = fixtures.fill_models(train_step.build_models(), 300)
for k in m:
    m[k].cuda().train()
if mode == 'step':
    step = train_step.GanStep(m, train_step.build_optimizers(m), 'grid')
    for i in range(2):
        res = step(b['mel'], b['spec'], b['vid'], b['vid_len'], noise=b['noise'])
        torch.cuda.synchronize()
        print('eager step', i, float(res['dis_loss']), float(res['gen_loss']), 'streams', len(ops._side_streams),
              flush=True)
    gs = train_step.GraphedStep(step, b['mel'], b['spec'], b['vid'], b['vid_len'], b['noise'], warmup=1)
    print('captured', flush=True)
    for i in range(3):
        res = gs(b['mel'], b['spec'], b['vid'], None, b['noise'])
        torch.cuda.synchronize()
        print('replay', i, float(res['dis_loss']), float(res['gen_loss']), flush=True)
else:
    dis = m['dis3']
    mel = b['mel'].clone().requires_grad_(True)
    sent = torch.randn(B, 512, T, device='cuda')

    def run():
        u, c = dis(mel, sent, T)
        if mode == 'dis_fwdbwd':
            (u.sum() + c.sum()).backward()
            ops.join_side_streams()
        return u, c
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        e = run()
        torch.cuda.synchronize()
        print('eager', float(e[0].sum()), float(e[1].sum()), flush=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print('eager ok, streams', len(ops._side_streams), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    print('captured', flush=True)
    g.replay()
    torch.cuda.synchronize()
    print('replay', float(out[0].sum()), float(out[1].sum()), flush=True)
print('done', flush=True)
