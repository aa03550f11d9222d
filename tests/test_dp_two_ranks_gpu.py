"""The data-parallel training step with two live ranks (the reference's counterpart is nn.DataParallel,
train.py:112-119).  Two processes, one shard of the batch each, run the captured per-phase graphs with the gradient
exchanges between them, in both schedules (three graphs and one blocking collective per side; five graphs with the
collectives started slice by slice).  Only one GPU is available to the tests, so both ranks share it and the
exchange runs over gloo; the orchestration (seeding, broadcast, segmented replay, exchange points) is the one an
RCCL job with a GPU per rank runs."""
import os
import socket
import subprocess
import sys

import pytest
import torch

gpu = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@gpu
@pytest.mark.parametrize('overlap', ['0', '1'])
def test_two_ranks_stay_in_step(tmp_path, overlap):
    port = str(_free_port())
    outs = [str(tmp_path / ('rank%d.pt' % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py'), str(r), '2', port, outs[r]],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(os.environ, VCA_EXCHANGE_OVERLAP=overlap)) for r in range(2)]
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=240)
            logs.append(o.decode(errors='replace')[-3000:])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), '\n'.join(logs)
    r0, r1 = (torch.load(o, weights_only=False) for o in outs)
    # replicas stay bit-identical: the same averaged gradients went into the same Adam step on both ranks
    for k in r0['sample']:
        assert r0['sample'][k][2] and r1['sample'][k][2], k
        assert torch.equal(r0['sample'][k][0], r1['sample'][k][0]), k
        assert r0['sample'][k][1] == r1['sample'][k][1], k
    for side in ('g', 'd'):
        assert torch.equal(r0['grads'][side], r1['grads'][side]), side
        assert float(r0['grads'][side].abs().sum()) > 0
    # the shards differ, so the ranks' local losses do
    assert r0['losses'][-1]['recon_loss'] != r1['losses'][-1]['recon_loss']
    for r in (r0, r1):
        for l in r['losses']:
            assert all(v == v and abs(v) < 1e6 for v in l.values()), l
