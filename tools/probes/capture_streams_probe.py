"""Probe (GPU box): how many streams a HIP graph capture can fork and join on this runtime.  A tree of streams (the
capture stream forks K branch streams, each of which forks one helper), a few kernels on each, joined in reverse; R
rounds per capture.  Prints the largest configuration that captures and replays correctly.
    python tools/probes/capture_streams_probe.py K HELPERS ROUNDS"""
import faulthandler
import sys

import torch

faulthandler.enable(all_threads=True)
K, HELP, R = (int(v) for v in (sys.argv[1:4] + ['2', '1', '3'])[:3])
OPTS = sys.argv[4:]
PRE = 'pre' in OPTS     # helpers are first forked from the capture's origin stream
x = torch.ones(1 << 16, device='cuda')
branches = [torch.cuda.Stream() for _ in range(K)]
helpers = [torch.cuda.Stream() for _ in range(K)] if HELP else []


def work():
    main = torch.cuda.current_stream()
    outs = []
    for r in range(R):
        for i, s in enumerate(branches):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                y = x * (i + 1)
                if HELP:
                    if PRE:
                        helpers[i].wait_stream(main)
                    if 'nofork_s' not in OPTS:
                        helpers[i].wait_stream(s)
                    with torch.cuda.stream(helpers[i]):
                        z = (y if 'nofork_s' not in OPTS else x) + r
                    y = y * 2
                    if 'nojoin_s' not in OPTS:
                        s.wait_stream(helpers[i])
                        y = y + z
                if 'cross' in OPTS and i > 0:      # a forked stream waits for its forked sibling
                    s.wait_stream(branches[i - 1])
                    y = y + outs[-1]
                outs.append(y)
        for s in branches:
            main.wait_stream(s)
        if 'joinmain' in OPTS:            # helpers also joined to the origin stream directly
            for h in helpers:
                main.wait_stream(h)
    return torch.stack(outs).sum(0)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    want = work()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    got = work()
print('captured K=%d helpers=%d rounds=%d pre=%s: %d streams' % (K, HELP, R, PRE, 1 + K + len(helpers)), flush=True)
g.replay()
torch.cuda.synchronize()
print('replay ok', bool(torch.equal(got, want)), flush=True)
