"""Debugging aid: are memset nodes of a replayed HIP graph still ordered correctly after a device-to-host read plus
a device synchronisation between two replays?  torch plus hipMemsetAsync through ctypes, none of this package's
kernels.  (torch.zeros itself is a fill kernel on the GPU, not a memset.)
usage: memset_graph_probe.py <nodes> <elements> [hip]"""
import ctypes
import sys
import torch

use_hip = len(sys.argv) > 3 and sys.argv[3] == 'hip'
use_copy = len(sys.argv) > 3 and sys.argv[3] == 'copy'     # memcpy nodes (clone of a contiguous tensor) instead
hip = ctypes.CDLL('libamdhip64.so') if use_hip else None

dev = torch.device('cuda', 0)
elems = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
src = torch.ones(max(1 << 20, elems * 7), device=dev)
n_nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def fn():
    outs = []
    for i in range(n_nodes):
        if use_copy:
            z = src[:elems * (1 + i % 7)].clone()           # then += 1 below: 2, or more if the copy did not land
        elif use_hip:
            z = torch.empty(elems * (1 + i % 7), device=dev)
            rc = hip.hipMemsetAsync(ctypes.c_void_p(z.data_ptr()), 0, ctypes.c_size_t(z.numel() * 4),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
        else:
            z = torch.zeros(elems * (1 + i % 7), device=dev)
        z += src[:z.numel()]                                  # kernel node that depends on it
        outs.append(z)
    return outs


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    outs = fn()


def check(tag):
    torch.cuda.synchronize()
    bad = sum(int((o != (2 if use_copy else 1)).any()) for o in outs)
    print(tag, 'tensors with a wrong value:', bad, 'of', len(outs), flush=True)


g.replay()
check('replay 1')
g.replay()
check('replay 2 (after the check itself: reads + sync)')
g.replay()
g.replay()
check('replay 3+4 back to back')
g.replay()
outs[0][0].item()
torch.cuda.synchronize()
g.replay()
check('replay after item + synchronize')
