"""Data-parallel gradient exchange (world_size 2, gloo, CPU): the mean all-reduce of flat gradient buffers that
train_step.GanStep issues before each optimizer step, and the equivalence it relies on: the mean over ranks of
shard-mean gradients equals the full-batch gradient for equal shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import importlib.util
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location(
            'ts_exchange', os.path.join(root, 'visual-context-attentional-gan_amd', 'grad_exchange.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        torch.manual_seed(0)
        lin = torch.nn.Linear(16, 4)
        x = torch.randn(8, 16)
        y = torch.randn(8, 4)
        # full-batch gradient
        lin.zero_grad()
        torch.nn.functional.mse_loss(lin(x), y).backward()
        full = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]).clone()
        # shard gradient into a flat buffer, exchange
        lin.zero_grad()
        sl = slice(rank * 4, rank * 4 + 4)
        torch.nn.functional.mse_loss(lin(x[sl]), y[sl]).backward()
        flat = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]).clone()
        other = torch.full((5,), float(rank + 1))
        ex = mod.GradExchange([flat, other])
        assert ex.world == world
        ex.reduce()
        q.put((rank, float((flat - full).abs().max()), other.tolist()))
    finally:
        dist.destroy_process_group()


def test_grad_exchange_two_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, other in res:
        assert err < 1e-6, (rank, err)
        assert other == [1.5] * 5
