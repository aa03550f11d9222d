"""One GAN training step (host-side mirror of the loop body of the reference's train.py:166-237 and
train_LRS.py:179-243) plus the data-parallel gradient exchange.

The reference parallelises with nn.DataParallel (train.py:112-119): one process, weights re-broadcast on every
module call, gradients reduced onto GPU 0.  Here each GPU is its own process holding a full replica; a step
does its local forward/backward on the local shard of the batch and exchanges gradients twice with RCCL
all-reduce (mean): the discriminator-side gradients before d_optimizer.step(), and the generator-side
gradients (which for v_front include the sync-loss contribution of the D phase) before g_optimizer.step().
Batch-norm statistics stay per replica, as they do under DataParallel.
"""
import math

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops
from .grad_exchange import GradExchange
from .models import generator as G
from .models.visual_front import Visual_front

G_NAMES = ('v_front', 'gen', 'post')
D_NAMES = ('dis1', 'dis2', 'dis3', 's_dis')
_LOG_EPS = math.log(1e-5)


def build_models(temp=1.0, device=None):
    """The seven networks of train.py:70-76."""
    m = dict(v_front=Visual_front(in_channels=1), gen=G.Decoder(), post=G.Postnet(),
             dis1=G.Discriminator(phase='1'), dis2=G.Discriminator(phase='2'), dis3=G.Discriminator(phase='3'),
             s_dis=G.sync_Discriminator(temp=temp))
    if device is not None:
        for k in m:
            m[k].to(device)
    return m


def build_optimizers(models, lr=1e-4, weight_decay=1e-5, amsgrad=True):
    """train.py:78-83 (GRID, amsgrad) / train_LRS.py:94-98 (no amsgrad): one Adam over (v_front, gen, post) and one
    over (dis1, dis2, dis3, s_dis).  Each module's parameters are re-homed into one flat buffer with a persistent
    flat gradient, so an optimizer step is one kernel per module and a gradient exchange is one all-reduce per
    module with no packing copies.  Call after the modules are on their device."""
    flats = {n: ops.flatten_module(models[n]) for n in G_NAMES + D_NAMES}
    g = ops.FusedAdam([flats[n] for n in G_NAMES], lr=lr, weight_decay=weight_decay, amsgrad=amsgrad)
    d = ops.FusedAdam([flats[n] for n in D_NAMES], lr=lr, weight_decay=weight_decay, amsgrad=amsgrad)
    g.flats = {n: flats[n] for n in G_NAMES}
    d.flats = {n: flats[n] for n in D_NAMES}
    return g, d


def denormalize(m):
    """Log-mel de-normalisation of src/data/vid_aud_grid.py:238-240."""
    return (m + 1.0) * (-_LOG_EPS / 2.0) + _LOG_EPS


class GanStep:
    """Callable running one full G+D step; variant 'grid' = train.py, 'lrs' = train_LRS.py."""

    def __init__(self, models, optimizers, variant='grid', group=None):
        self.m = models
        self.g_opt, self.d_opt = optimizers
        self.variant = variant
        self.gflat = {n: gf for opt in optimizers for n, (_, gf) in opt.flats.items()}
        self.d_exchange = GradExchange([self.gflat[n] for n in D_NAMES], group=group)
        self.g_exchange = GradExchange([self.gflat[n] for n in G_NAMES], group=group)

    def _zero(self, names):
        for n in names:
            self.gflat[n].zero_()

    def __call__(self, mel, spec, vid, vid_len, noise=None):
        """mel (B,1,80,4T) in [-1,1], spec (B,1,321,4T), vid (B,1,T,H,W), all on the device; vid_len: lengths."""
        self.d_phase(mel, spec, vid, vid_len, noise)
        self.d_exchange.reduce()
        self.g_phase()
        self.g_exchange.reduce()
        return self.finish()

    # The step is three device-only phases separated by the two gradient exchanges, so that a caller can replay
    # each phase as one HIP graph and keep the collectives outside the graphs.
    def d_phase(self, mel, spec, vid, vid_len, noise=None):
        """Forward of the generator side, discriminator losses and their backward (train.py:170-210)."""
        m = self.m
        ops.step_begin(mel.device)
        self._zero(G_NAMES)
        ops.rng_advance(mel.device)
        mel1 = ops.mel_down(mel, 4)          # F.interpolate(mel, scale_factor=0.25, mode='bilinear'), train.py:170
        mel2 = ops.mel_down(mel, 2)
        phon, sent = m['v_front'](vid)
        g1, g2, g3 = m['gen'](sent, phon, vid_len, noise=noise)
        reals = [t.detach().clone().requires_grad_(True) for t in (mel1, mel2, mel)]
        T = phon.size(1)
        sd = sent.detach()
        dis = (m['dis1'], m['dis2'], m['dis3'])

        real_lo = [d(r, sd, T) for d, r in zip(dis, reals)]
        # The sync loss is the only D-phase term that reaches v_front (train.py:186,210 "accumulate v_front grad").
        # Back-propagation is linear in the incoming gradient, so instead of running v_front's backward twice (once
        # per phase) the D phase stops at the phoneme features and their gradient is added to the G phase's before
        # v_front is traversed once: v_front.grad = J^T (g_D + g_G) = J^T g_D + J^T g_G.
        phon_d = phon.detach().requires_grad_(True)
        sync_loss = m['s_dis'](phon_d, reals[2]).mean()
        grad_p = []
        for (u, _), r in zip(real_lo, reals):
            with ops.input_grads_only():
                gr, = torch.autograd.grad(u.sum(), r, create_graph=True)
            grad_p.append(gr.reshape(gr.size(0), -1).pow(2).sum(1).mean())
        fake_lo = [d(g.detach(), sd, T) for d, g in zip(dis, (g1, g2, g3))]
        real_loss = (sum(G.gan_loss(u, True) for u, _ in real_lo)
                     + sum(G.gan_loss(c, True) for _, c in real_lo)) / 3 + sum(grad_p) / 3
        fake_loss = (sum(G.gan_loss(u, False) for u, _ in fake_lo)
                     + sum(G.gan_loss(c, False) for _, c in fake_lo)) / 3
        dis_loss = real_loss + fake_loss + (1.0 if self.variant == 'grid' else 0.5) * sync_loss
        self._zero(D_NAMES)
        with ops.direct_param_grad():
            dis_loss.backward()
        self._st = dict(spec=spec, phon=phon, phon_d=phon_d, sd=sd, T=T, g=(g1, g2, g3),
                        tgt=[r.detach() for r in reals], dis_loss=dis_loss.detach(), sync_loss=sync_loss.detach(),
                        grad_p=torch.stack([x.detach() for x in grad_p]))

    def g_phase(self):
        """d_optimizer.step(), then the generator losses and their backward (train.py:211-235)."""
        m, st = self.m, self._st
        self.d_opt.step()
        g1, g2, g3 = st['g']
        phon, sd, T = st['phon'], st['sd'], st['T']
        dis = (m['dis1'], m['dis2'], m['dis3'])
        # The reference also back-propagates gen_loss into the discriminators' weights and throws those gradients
        # away (d_optimizer.zero_grad() precedes their next use, train.py:209); here they are simply not computed.
        d_params = [p for n in D_NAMES for p in m[n].parameters()]
        for p in d_params:
            p.requires_grad_(False)
        gs = m['post'](g3)
        gen_lo = [d(g, sd, T) for d, g in zip(dis, (g1, g2, g3))]
        g_sync = m['s_dis'](phon.detach(), g3, gen=True).mean()
        g_adv = (sum(G.gan_loss(u, True) for u, _ in gen_lo) + sum(G.gan_loss(c, True) for _, c in gen_lo)) / 3
        tgt = st['tgt']
        if self.variant == 'grid':
            g_loss = g_adv + g_sync
            k = -_LOG_EPS / 2.0           # slope of the de-normalisation applied to both L1 arguments
            recon = sum(ops.l1_mean(g, t, k) for g, t in zip((g1, g2, g3), tgt)) / 3 + ops.l1_mean(gs, st['spec'])
            gen_loss = g_loss + 50.0 * recon
        else:
            g_loss = g_adv
            recon = sum(ops.l1_mean(g, t) for g, t in zip((g1, g2, g3), tgt)) / 3 + ops.l1_mean(gs, st['spec'])
            gen_loss = g_loss + 50.0 * recon + g_sync
        self._zero(D_NAMES + ('gen', 'post'))
        with ops.direct_param_grad():
            torch.autograd.backward([gen_loss, phon], [None, st['phon_d'].grad])
        for p in d_params:
            p.requires_grad_(True)
        self._st = dict(dis_loss=st['dis_loss'], gen_loss=gen_loss.detach(), sync_loss=st['sync_loss'],
                        g_sync_loss=g_sync.detach(), recon_loss=recon.detach(), g_loss=g_loss.detach(),
                        grad_p=st['grad_p'], g1=g1.detach(), g2=g2.detach(), g3=g3.detach(), gs=gs.detach())

    def finish(self):
        """g_optimizer.step() (train.py:236); returns the step's losses and outputs."""
        self.g_opt.step()
        ops.step_end(self._st['g3'].device)
        out, self._st = self._st, None
        return out


class GraphedStep:
    """A GanStep captured once into HIP graphs and replayed instead of several thousand kernel launches issued from
    Python per training step.  A single process captures the whole step as one graph; with data parallelism (or
    segmented=True) the three phases are three graphs sharing one memory pool, replayed in order with the two
    gradient all-reduces issued between them on the same stream, so no collective is ever captured.  Inputs are
    copied into static device buffers before each replay; the returned tensors are the graphs' static outputs
    (overwritten by the next call).  Shapes are fixed at capture."""

    def __init__(self, step, mel, spec, vid, vid_len, noise, warmup=2, segmented=None):
        self.step = step
        dev = mel.device
        self.static = dict(mel=mel.clone(), spec=spec.clone(), vid=vid.clone(), noise=noise.clone(),
                           vid_len=torch.as_tensor([int(l) for l in vid_len], device=dev))
        self.segmented = step.d_exchange.active if segmented is None else bool(segmented)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator, lazy module state and kernel attributes settle here
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        s = self.static
        if not self.segmented:
            self.graphs = [torch.cuda.CUDAGraph()]
            with torch.cuda.graph(self.graphs[0]):
                self.out = self._run()
        else:
            # Other threads (the process group's watchdog) may touch the device while a phase is being captured.
            mode = 'thread_local' if step.d_exchange.active else 'global'
            pool = torch.cuda.graph_pool_handle()
            self.graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
            with torch.cuda.graph(self.graphs[0], pool=pool, capture_error_mode=mode):
                step.d_phase(s['mel'], s['spec'], s['vid'], s['vid_len'], s['noise'])
            step.d_exchange.reduce()         # keeps the ranks' collective sequences aligned; the data is not used
            with torch.cuda.graph(self.graphs[1], pool=pool, capture_error_mode=mode):
                step.g_phase()
            step.g_exchange.reduce()
            with torch.cuda.graph(self.graphs[2], pool=pool, capture_error_mode=mode):
                self.out = step.finish()
        self.warmup_steps = warmup

    def _run(self):
        s = self.static
        return self.step(s['mel'], s['spec'], s['vid'], s['vid_len'], noise=s['noise'])

    def __call__(self, mel, spec, vid, vid_len=None, noise=None):
        s = self.static
        for k, v in (('mel', mel), ('spec', spec), ('vid', vid), ('noise', noise)):
            if v is not None and v.data_ptr() != s[k].data_ptr():
                s[k].copy_(v, non_blocking=True)
        if vid_len is not None and not (torch.is_tensor(vid_len) and vid_len.data_ptr() == s['vid_len'].data_ptr()):
            s['vid_len'].copy_(torch.as_tensor([int(l) for l in vid_len]), non_blocking=True)
        if not self.segmented:
            self.graphs[0].replay()
        else:
            self.graphs[0].replay()
            self.step.d_exchange.reduce()
            self.graphs[1].replay()
            self.step.g_exchange.reduce()
            self.graphs[2].replay()
        return self.out
