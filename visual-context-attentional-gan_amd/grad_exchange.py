"""Data-parallel gradient exchange: mean all-reduce of flat gradient buffers over torch.distributed (RCCL on the
GPUs, gloo in the CPU tests).  Kept free of any GPU-library import so it can be exercised without a GPU."""
import torch.distributed as dist


class GradExchange:
    """Mean all-reduce of flat gradient buffers (one RCCL call per module, issued back to back and waited together)."""

    def __init__(self, gflats, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.gflats = list(gflats)

    def reduce(self):
        if self.world == 1:
            return
        works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for g in self.gflats]
        for w, g in zip(works, self.gflats):
            w.wait()
            g.mul_(1.0 / self.world)
