"""Data-parallel gradient exchange: mean all-reduce of flat gradient buffers over torch.distributed (RCCL on the
GPUs, gloo in the CPU tests).  Kept free of any GPU-library import so it can be exercised without a GPU."""
import os

import torch.distributed as dist


class GradExchange:
    """Mean all-reduce of flat gradient buffers.  A side of the step (the discriminators, or v_front + generator +
    postnet) keeps its gradients in one contiguous buffer, each network a contiguous slice of it, so its exchange is
    one collective (reduce()) or one per slice, started as the slices become final (start() / wait()): RCCL averages
    in the reduction itself (ReduceOp.AVG); backends without it (gloo) sum and scale.  gloo gets device buffers through
    a host copy: its collectives on device tensors do not complete on this ROCm stack (both ranks sit in their first
    broadcast), and gloo is only the fallback for rehearsals of the multi-rank path without one GPU per rank."""

    def __init__(self, gflats, group=None, force=False):
        self.group = group
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        self.gflats = list(gflats)
        # force (or VCA_FORCE_EXCHANGE=1) keeps the collectives, and the segmented graph replay built around them,
        # in a single-rank job: a one-GPU rehearsal of everything the multi-GPU run does except the xGMI traffic.
        forced = (force or os.environ.get('VCA_FORCE_EXCHANGE') == '1') and ready
        self.active = self.world > 1 or forced
        self.avg = ready and dist.get_backend(group) == 'nccl'
        self.timed = None     # a list: wait() appends a (start, end) pair of device events per call (bench.py)
        self._works = []

    def start(self, tensors=None):
        """Issue the mean all-reduce of `tensors` (contiguous slices of the flat buffers; default: the whole buffers)
        and return at once.  With RCCL the collective runs on the process group's own stream, behind everything the
        calling stream has been given so far, so kernels launched on the calling stream after start() run beside it:
        a side's exchange is started slice by slice as soon as the backward pass has finished with a slice, and
        wait() is called where the averaged gradients are needed.  Backends without device collectives (gloo)
        reduce here, blocking, through the host."""
        if not self.active:
            return
        tensors = self.gflats if tensors is None else list(tensors)
        if self.avg:
            self._works += [dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.group, async_op=True) for g in tensors]
            return
        for g in tensors:
            h = g.cpu() if g.is_cuda else g
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            h.mul_(1.0 / self.world)
            if g.is_cuda:
                g.copy_(h)

    def wait(self):
        """The calling stream waits for every collective started so far (the host does not).  The event pair kept
        in `timed` brackets exactly that wait: the part of the exchange the step is exposed to."""
        if not self.active or not self.avg:
            return
        ev = None
        if self.timed is not None:
            import torch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()
        self._works = []
        if ev is not None:
            ev[1].record()
            self.timed.append(ev)

    def reduce(self):
        """start() of the whole buffers and wait(): nothing overlaps."""
        self.start()
        self.wait()
