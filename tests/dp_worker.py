"""One rank of the two-process data-parallel rehearsal (tests/test_dp_two_ranks_gpu.py): the training step's
multi-rank code path (per-rank seeds, initial broadcast, the per-phase graphs with the gradient all-reduces
between them) with a live process group.  Both ranks use the same GPU, so the collective backend is gloo (two RCCL
ranks on one device cannot make progress); everything else is what a one-GPU-per-rank RCCL job runs.
usage: dp_worker.py RANK WORLD PORT OUT.pt"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _log(*a):
    print('[dp_worker %s]' % sys.argv[1], *a, flush=True)


def _spread(n, device, k=997):
    """k indices spread over [0, n) in exact integer arithmetic (a float32 linspace rounds its end point past n - 1
    for n above 2**24)."""
    return (torch.arange(k, dtype=torch.int64) * (n - 1) // (k - 1)).to(device)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get('DP_WORKER_TRACE_S', '150')), exit=True)   # a stall names its line
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _log('process group up')
    try:
        import vcagan_amd  # noqa: F401
        from oracle import fixtures
        from vcagan_amd import ops, train_step
        torch.cuda.set_device(0)
        ops.set_precision('mixed')
        m = fixtures.fill_models(train_step.build_models(), 300 + rank)      # ranks start apart on purpose
        for k in m:
            m[k].cuda().train()
        for k in sorted(m):                                                   # rank 0's initialisation everywhere
            for t in list(m[k].parameters()) + list(m[k].buffers()):
                h = t.data.cpu()                                              # gloo: through the host
                dist.broadcast(h, 0)
                t.data.copy_(h)
        _log('weights broadcast')
        step = train_step.GanStep(m, train_step.build_optimizers(m), 'grid')
        assert step.d_exchange.active and step.g_exchange.active and step.d_exchange.world == world
        b = fixtures.make_batch(2, 20, 96, 100 + rank, [20, 14 + rank])      # each rank its own shard
        d = {k: (v.cuda() if k != 'vid_len' else v) for k, v in b.items()}
        gs = train_step.GraphedStep(step, d['mel'], d['spec'], d['vid'], d['vid_len'], 'device', warmup=1)
        assert step.overlap == (os.environ.get('VCA_EXCHANGE_OVERLAP') == '1')
        assert gs.segmented and len(gs.graphs) == (5 if step.overlap else 3)
        _log('captured')
        losses = []
        for _ in range(2):
            res = gs(d['mel'], d['spec'], d['vid'])
            losses.append({k: float(res[k]) for k in ('dis_loss', 'gen_loss', 'recon_loss')})
        torch.cuda.synchronize()
        _log('replayed', losses[-1])
        sample = {}
        for k in sorted(m):
            flat = torch.cat([p.detach().reshape(-1) for p in m[k].parameters()])
            idx = _spread(flat.numel(), flat.device)
            sample[k] = (flat[idx].cpu(), float(flat.double().sum()), bool(torch.isfinite(flat).all()))
        grads = {n: o.all_grads[_spread(o.all_grads.numel(), o.all_grads.device)].cpu()
                 for n, o in (('g', step.g_opt), ('d', step.d_opt))}
        torch.save(dict(rank=rank, losses=losses, sample=sample, grads=grads), out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
