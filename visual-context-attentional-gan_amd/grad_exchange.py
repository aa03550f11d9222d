"""Data-parallel gradient exchange: mean all-reduce of flat gradient buffers over torch.distributed (RCCL on the
GPUs, gloo in the CPU tests).  Kept free of any GPU-library import so it can be exercised without a GPU."""
import os

import torch.distributed as dist


class GradExchange:
    """Mean all-reduce of flat gradient buffers (one RCCL call per module, issued back to back and waited together)."""

    def __init__(self, gflats, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.gflats = list(gflats)
        # VCA_FORCE_EXCHANGE=1 keeps the collectives (and the segmented graph replay built around them) in a
        # single-rank job: a one-GPU rehearsal of everything the multi-GPU run does except the xGMI traffic.
        forced = os.environ.get('VCA_FORCE_EXCHANGE') == '1' and dist.is_available() and dist.is_initialized()
        self.active = self.world > 1 or forced

    def reduce(self):
        if not self.active:
            return
        works = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for g in self.gflats]
        for w, g in zip(works, self.gflats):
            w.wait()
            g.mul_(1.0 / self.world)
