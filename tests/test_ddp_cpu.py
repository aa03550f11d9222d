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


def _worker_models(rank, world, port, q):
    """The exchange on the real layout: the seven networks' parameters re-homed into one flat buffer per optimizer
    side (ops.flatten_modules, as train_step.build_optimizers does), one all-reduce per side (train.py:112-119:
    nn.DataParallel averages the replicas' gradients)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import vcagan_amd  # noqa: F401
        from vcagan_amd import ops, train_step
        from vcagan_amd.grad_exchange import GradExchange
        torch.manual_seed(1)
        models = train_step.build_models()
        out = {}
        for side, names in (('g', train_step.G_NAMES), ('d', train_step.D_NAMES)):
            flats, gflats, allg = ops.flatten_modules([models[n] for n in names])
            params = [p for n in names for p in models[n].parameters()]
            # every parameter and gradient is a view of the side's buffers
            lo, hi = allg.data_ptr(), allg.data_ptr() + allg.numel() * 4
            assert all(lo <= p.grad.data_ptr() < hi for p in params)
            assert sum((p.numel() + 3) // 4 * 4 for p in params) == allg.numel()
            for i, p in enumerate(params):
                p.grad.fill_(float((rank + 1) * (i % 7 + 1)))
            ex = GradExchange([allg])
            assert ex.active and ex.world == world
            ex.reduce()
            want = [1.5 * (i % 7 + 1) for i in range(len(params))]
            err = max(float((p.grad - w).abs().max()) for p, w in zip(params, want))
            # the same exchange slice by slice (one network at a time, as the overlapped schedule starts them)
            for i, p in enumerate(params):
                p.grad.fill_(float((rank + 1) * (i % 7 + 1)))
            nets = [gf for _, gf in flats]
            assert sum(g.numel() for g in nets) == allg.numel()
            assert all(a.data_ptr() + a.numel() * 4 == b.data_ptr() for a, b in zip(nets, nets[1:]))
            for g in nets[1:]:
                ex.start([g])
            ex.start(nets[:1])
            ex.wait()
            err = max(err, max(float((p.grad - w).abs().max()) for p, w in zip(params, want)))
            out[side] = (len(params), allg.numel(), err)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_grad_exchange_on_the_real_flat_buffers():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_models, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1]['g'][:2] == res[1][1]['g'][:2] and res[0][1]['d'][:2] == res[1][1]['d'][:2]
    for rank, out in res:
        for side in ('g', 'd'):
            assert out[side][2] == 0.0, (rank, side, out[side])


def test_bench_refuses_ranks_without_their_own_device():
    """Two ranks on fewer than two devices never leave their first RCCL collective; bench.py has to say so and exit
    before init_process_group instead of hanging (the failure recorded for a 2-rank launch on a one-GPU box)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a host with fewer than 2 GPUs')
    env = dict(os.environ, WORLD_SIZE='2', RANK='1', LOCAL_RANK='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert 'no GPU of its own' in (r.stderr + r.stdout), r.stderr[-500:]
    env['WORLD_SIZE'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)
