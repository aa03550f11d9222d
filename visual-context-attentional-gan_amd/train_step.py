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
import os

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
# The three multi-scale discriminators are independent networks, and most of their kernels work on small maps
# (5x18 ... 20x75) that leave the chip half empty.  With D_STREAMS each one runs on a stream of its own (forward and,
# because autograd replays an operator's backward on the stream of its forward, backward too), forked from and joined
# to the step's stream, also inside a captured graph: 80.5 -> 74.0 ms per step at B32 x T75 x 96x96.  VCA_D_STREAMS=0
# (or train_step.D_STREAMS = False) keeps everything on one stream.
D_STREAMS = __import__('os').environ.get('VCA_D_STREAMS', '1') != '0'
# Branch -> stream: dis3 (80x300 mels, four down-sampling blocks: more work than dis1 and dis2 together) has a stream to
# itself and is issued first, dis1 and dis2 share one, a fourth branch (the ASR scorer of the multi-task step) gets its
# own.  Measured against one stream per discriminator: 73.1 vs 73.5 ms per step, with one stream fewer (a process has
# four hardware queues on this part, and the collectives of a data-parallel run bring their own stream).
_BRANCH_STREAM = {0: 0, 1: 0, 2: 1, 3: 2}
_BRANCH_ORDER = (2, 3, 0, 1)


class _quiet_leaf_streams:
    """Around the step's two backward calls, and only with D_STREAMS: the full-resolution real mel is a leaf that both
    dis3 (on its stream) and the sync discriminator (on the step's stream) differentiate; autograd synchronises the two
    producers itself and warns that it had to.  The process-wide switch is restored on exit, so the diagnostic stays
    on for everything else."""

    def __enter__(self):
        self.prev = None
        get = getattr(torch._C, '_warn_on_accumulate_grad_stream_mismatch', None)
        if D_STREAMS and get is not None and hasattr(torch.autograd.graph,
                                                     'set_warn_on_accumulate_grad_stream_mismatch'):
            self.prev = bool(get())
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(self.prev)


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
    opts = []
    for names in (G_NAMES, D_NAMES):
        # one contiguous parameter buffer and one gradient buffer per optimizer, a slice per network
        flats, _, allg = ops.flatten_modules([models[n] for n in names])
        o = ops.FusedAdam(flats, lr=lr, weight_decay=weight_decay, amsgrad=amsgrad)
        o.flats = dict(zip(names, flats))
        o.all_grads = allg
        opts.append(o)
    return tuple(opts)


def denormalize(m):
    """Log-mel de-normalisation of src/data/vid_aud_grid.py:238-240."""
    return (m + 1.0) * (-_LOG_EPS / 2.0) + _LOG_EPS


class _Heads:
    """Collects the loss terms of a phase for ops.loss_heads (one forward and one backward launch for all of them)."""

    def __init__(self):
        self.heads, self.tensors = [], []

    def add(self, kind, x, slots, y=None, sign=1.0, scale=None):
        xi = len(self.tensors)
        self.tensors.append(x)
        yi = None
        if y is not None:
            yi = len(self.tensors)
            self.tensors.append(y)
        self.heads.append((kind, xi, yi, float(sign), float(1.0 / x.numel() if scale is None else scale),
                           tuple((int(s), float(w)) for s, w in slots)))

    def run(self, nslots):
        return ops.loss_heads(self.heads, nslots, self.tensors)


class GanStep:
    """Callable running one full G+D step; variant 'grid' = train.py, 'lrs' = train_LRS.py."""

    def __init__(self, models, optimizers, variant='grid', group=None, force_exchange=False, ctc=None, overlap=None,
                 scaler=None):
        """scaler: an ops.LossScaler; required (and created by default) when the precision mode computes gradients in
        IEEE half ('fp16', 'fp16-mixed', 'fp16x3'; BASELINE configs[4] "fp16 with loss-scaling"): both backward passes
        are seeded with the scale, the optimizers divide it out and skip their step on overflow, the scale adapts at
        the end of every step.  The reported losses are unscaled.
        overlap: run the gradient exchanges beside the parts of the step that do not need their results (only
        with an active exchange; default: off, VCA_EXCHANGE_OVERLAP=1 turns it on); see schedule().
        ctc: None, or dict(front=Audio_front, back=Backend, weight=w): the multi-task configuration (BASELINE
        configs[4]) adds w * CTC(ASR scorer(g3), text) to gen_loss.  The reference ships that scorer for inference only
        (ASR_model/GRID/test.py) and has no such training term, so this one is defined here: the scorer is frozen and
        in eval mode, it reads the generated mel g3 as it is, its frame rate is the video's (mel frames / 4), the loss
        is torch's ctc_loss 'mean' reduction (per-sample loss / target length, batch mean, blank 0, zero_infinity)."""
        self.m = models
        self.ctc = ctc
        if ctc is not None:
            for k in ('front', 'back'):
                ctc[k].eval()
                for p in ctc[k].parameters():
                    p.requires_grad_(False)
        self.g_opt, self.d_opt = optimizers
        self.variant = variant
        self.gflat = {n: gf for opt in optimizers for n, (_, gf) in opt.flats.items()}
        self.d_exchange = GradExchange([self.d_opt.all_grads], group=group, force=force_exchange)
        self.g_exchange = GradExchange([self.g_opt.all_grads], group=group, force=force_exchange)
        if overlap is None:
            overlap = os.environ.get('VCA_EXCHANGE_OVERLAP', '0') == '1'
        self.overlap = bool(overlap) and self.g_exchange.active
        self._seeds = {}
        if scaler is None and ops.half_gradients():
            scaler = ops.LossScaler(next(models['gen'].parameters()).device)
        self.scaler = scaler
        if dist.is_available() and dist.is_initialized():
            ops.set_seed(torch.initial_seed(), dist.get_rank(group))   # each replica draws its own dropout masks

    def _each_dis(self, fn, meanwhile=None, n=3):
        """([fn(0), ..., fn(n-1)], meanwhile()): one call per discriminator (and per further independent branch), each
        on its own stream when D_STREAMS is set, and `meanwhile` (work that needs none of their results) on the step's
        stream before it joins them."""
        if not D_STREAMS or not torch.cuda.is_available():
            return [fn(i) for i in range(n)], (meanwhile() if meanwhile is not None else None)
        main = torch.cuda.current_stream()
        if getattr(self, '_side', None) is None:
            self._side = []
        used = sorted({_BRANCH_STREAM[i] for i in range(n)})
        while len(self._side) <= used[-1]:
            self._side.append(torch.cuda.Stream())
        for k in used:
            self._side[k].wait_stream(main)
        res = {}
        for i in (j for j in _BRANCH_ORDER if j < n):
            with torch.cuda.stream(self._side[_BRANCH_STREAM[i]]):
                res[i] = fn(i)
        other = meanwhile() if meanwhile is not None else None
        for k in used:
            main.wait_stream(self._side[k])
        return [res[i] for i in range(n)], other

    def _join_dis(self):
        """The step's stream waits for the discriminators' streams.  After a backward pass this is not implied:
        autograd only makes the caller's stream wait for streams on which a leaf's gradient was accumulated, and the
        last kernels of a discriminator's backward (weight gradients added straight into the flat buffer) are
        followed by none."""
        if getattr(self, '_side', None) is not None:
            main = torch.cuda.current_stream()
            for st in self._side:
                main.wait_stream(st)
        ops.join_side_streams()

    def _zero(self, names):
        """Clear the flat gradients of the named networks; whole sides (contiguous buffers) with one fill."""
        names = tuple(names)
        if names[:len(D_NAMES)] == D_NAMES:
            self.d_opt.all_grads.zero_()
            names = names[len(D_NAMES):]
        if names == G_NAMES:
            self.g_opt.all_grads.zero_()
            names = ()
        for n in names:
            self.gflat[n].zero_()

    def __call__(self, mel, spec, vid, vid_len, noise=None, text=None, text_len=None):
        """mel (B,1,80,4T) in [-1,1], spec (B,1,321,4T), vid (B,1,T,H,W), all on the device; vid_len: lengths.
        text (B, Lmax) class ids and text_len (B,): the transcripts, with ctc= only."""
        sched = self.schedule()
        self.d_phase(mel, spec, vid, vid_len, noise, text, text_len)
        sched[0][1]()
        for phase, exchange in sched[1:]:
            phase()
            exchange()
        return self._out

    def g_slices(self, *names):
        """The named networks' gradients as contiguous runs of the generator side's flat buffer (neighbours merged)."""
        flat = self.g_opt.all_grads
        runs = []
        for n in names:
            g = self.gflat[n]
            off = (g.data_ptr() - flat.data_ptr()) // flat.element_size()
            if runs and runs[-1][1] == off:
                runs[-1][1] = off + g.numel()
            else:
                runs.append([off, off + g.numel()])
        return [flat[a:b] for a, b in runs]

    def schedule(self):
        """The step as (phase, exchange) pairs: device-only phases, each followed by the collective calls issued
        behind it (a caller may replay every phase as one HIP graph and keep the collectives outside the graphs; the
        first phase is d_phase, which takes the batch).  Without overlap: the reference's order (train.py:209-236 under
        nn.DataParallel, train.py:112-119), each side's gradients exchanged in one blocking collective.  With overlap
        the discriminator side's all-reduce runs beside the postnet's forward pass (the one part of the G phase that
        does not read the updated discriminators), and the generator side's is split where its backward pass is:
        generator + postnet gradients travel while the visual front-end is still being back-propagated, the front-end's
        slice follows, and the step waits for both before g_optimizer.step().  The price is two more cuts in a
        captured step (five graphs instead of three), and a cut costs 0.5-0.9 ms of drained pipeline and lost
        cross-phase overlap: a single-rank rehearsal takes 72.1 ms per step this way against 70.2 (69.1 as one graph),
        which is about what an 8-GPU xGMI node can hide (two 200 MB all-reduces, roughly 1.4 ms each, of which the
        postnet and the front-end's backward pass cover some 1.8 ms).  Hence off by default; it is meant for
        exchanges slower than one node's."""
        def fin():
            self._out = self.finish()
        if not self.overlap:
            return [(self.d_phase, self.d_exchange.reduce), (self.g_phase, self.g_exchange.reduce), (fin, lambda: None)]

        def last():
            self.g_exchange.start(self.g_slices('v_front'))
            self.g_exchange.wait()
        return [(self.d_phase, self.d_exchange.start),
                (self.g_pre, self.d_exchange.wait),
                (self.g_main, lambda: self.g_exchange.start(self.g_slices('gen', 'post'))),
                (self.g_trunk, last),
                (fin, lambda: None)]

    def d_phase(self, mel, spec, vid, vid_len, noise=None, text=None, text_len=None):
        """Forward of the generator side, discriminator losses and their backward (train.py:170-210)."""
        m = self.m
        assert (self.ctc is None) == (text is None), 'transcripts go with the ctc= configuration, and only with it'
        ops.step_begin(mel.device)
        self._zero(G_NAMES)
        ops.rng_advance(mel.device)
        mel1 = ops.mel_down(mel, 4)          # F.interpolate(mel, scale_factor=0.25, mode='bilinear'), train.py:170
        mel2 = ops.mel_down(mel, 2)
        # The sentence encoder (a bi-GRU: 2 x T launches of a few microseconds) on a side stream beside the decoder's
        # first blocks, which do not read its output: requested for this call only, because the output then has to be
        # waited for (ops.ensure_ready) by whoever reads it; the decoder does so before its first attention module.
        m['v_front'].defer_sentence = D_STREAMS
        try:
            if self.overlap:
                # The G phase's backward pass is cut at the phoneme features (g_main / g_trunk): everything downstream
                # of them, the sentence encoder included, hangs on a leaf.
                trunk = m['v_front'].trunk(vid)
                phon = trunk.detach().requires_grad_(True)
                sent = m['v_front'].sentence(phon)
            else:
                trunk = None
                phon, sent = m['v_front'](vid)
        finally:
            m['v_front'].defer_sentence = False
        g1, g2, g3 = m['gen'](sent, phon, vid_len, noise=noise)
        ops.ensure_ready(sent)
        reals = [t.detach().clone().requires_grad_(True) for t in (mel1, mel2, mel)]
        T = phon.size(1)
        sd = sent.detach()
        dis = (m['dis1'], m['dis2'], m['dis3'])

        # The sync loss is the only D-phase term that reaches v_front (train.py:186,210 "accumulate v_front grad").
        # Back-propagation is linear in the incoming gradient, so instead of running v_front's backward twice (once
        # per phase) the D phase stops at the phoneme features and their gradient is added to the G phase's before
        # v_front is traversed once: v_front.grad = J^T (g_D + g_G) = J^T g_D + J^T g_G.
        phon_d = phon.detach().requires_grad_(True)
        real_lo, sync = self._each_dis(lambda i: dis[i](reals[i], sd, T), lambda: m['s_dis'](phon_d, reals[2]))
        # All loss terms of the phase go through one reduction launch (and one backward launch):
        # slot 0 dis_loss = real + fake + w * sync (train.py:200-205 / train_LRS.py:211-218), 1 sync_loss, 2-4 the R1
        # terms mean_b ||d u_i / d real_i||^2 (train.py:188-194, weight 1/3 each in real_loss).
        hd = _Heads()
        B = mel.size(0)
        def r1(i):
            with ops.input_grads_only():
                return torch.autograd.grad(real_lo[i][0].sum(), reals[i], create_graph=True)[0]
        for i, gr in enumerate(self._each_dis(r1)[0]):
            hd.add(ops.HEAD_SQSUM, gr, ((2 + i, 1.0), (0, 1.0 / 3)), scale=1.0 / B)
        fakes = [g.detach() for g in (g1, g2, g3)]
        fake_lo = self._each_dis(lambda i: dis[i](fakes[i], sd, T))[0]
        for u, c in real_lo:                  # gan_loss(x, True) = mean softplus(-x), generator.py:363-366
            hd.add(ops.HEAD_SOFTPLUS, u, ((0, 1.0 / 3),), sign=-1.0)
            hd.add(ops.HEAD_SOFTPLUS, c, ((0, 1.0 / 3),), sign=-1.0)
        for u, c in fake_lo:
            hd.add(ops.HEAD_SOFTPLUS, u, ((0, 1.0 / 3),), sign=1.0)
            hd.add(ops.HEAD_SOFTPLUS, c, ((0, 1.0 / 3),), sign=1.0)
        hd.add(ops.HEAD_SUM, sync, ((1, 1.0), (0, 1.0 if self.variant == 'grid' else 0.5)))
        out = hd.run(5)
        self._zero(D_NAMES)
        with ops.direct_param_grad(), _quiet_leaf_streams():
            out.backward(self._onehot(5, 0, out.device))
            self._join_dis()
        self._st = dict(spec=spec, phon=phon, phon_d=phon_d, trunk=trunk, sd=sd, T=T, g=(g1, g2, g3),
                        tgt=[r.detach() for r in reals], dis_loss=out[0].detach(), sync_loss=out[1].detach(),
                        grad_p=out[2:5].detach(), text=text, text_len=text_len, vid_len=vid_len)

    def _onehot(self, n, i, device):
        """Constant seed gradient selecting slot i of a loss vector (built once, so a captured step holds no fill)."""
        key = (n, i, device)
        t = self._seeds.get(key)
        if t is None:
            t = torch.zeros(n, device=device, dtype=torch.float32)
            t[i] = 1.0
            self._seeds[key] = t
        return t if self.scaler is None else self.scaler.seed(t)

    def g_phase(self):
        """d_optimizer.step(), then the generator losses and their backward (train.py:211-235)."""
        self.g_pre()
        self.g_main()
        self.g_trunk()

    def g_pre(self):
        """With overlap: the postnet's forward pass on g3 (train.py:214), ahead of d_optimizer.step()."""
        if self.overlap:
            self._st['gs'] = self.m['post'](self._st['g'][2])

    def g_main(self):
        """d_optimizer.step(), the generator losses and their backward pass; with overlap the pass ends at the phoneme
        features and g_trunk() takes it through the visual front-end."""
        m, st = self.m, self._st
        self.d_opt.step(self.scaler, 0)
        g1, g2, g3 = st['g']
        phon, sd, T = st['phon'], st['sd'], st['T']
        dis = (m['dis1'], m['dis2'], m['dis3'])
        # The reference also back-propagates gen_loss into the discriminators' weights and throws those gradients
        # away (d_optimizer.zero_grad() precedes their next use, train.py:209); here they are simply not computed.
        d_params = [p for n in D_NAMES for p in m[n].parameters()]
        for p in d_params:
            p.requires_grad_(False)
        gens = (g1, g2, g3)

        def branch(i):      # 0-2: the discriminators on the generated mels; 3: the ASR scorer's CTC loss (ctc= only)
            if i < 3:
                return dis[i](gens[i], sd, T)
            logits = self.ctc['back'](self.ctc['front'](g3))                 # (B, T, 28), one frame per video frame
            frames = ops.as_lens(st['vid_len'], logits.device).clamp(max=logits.shape[1])
            tl = ops.as_lens(st['text_len'], logits.device)
            return ops.ctc_nll(logits, st['text'], frames, tl) / tl.clamp(min=1).to(logits.dtype)

        post = (lambda: st['gs']) if 'gs' in st else (lambda: m['post'](g3))
        lo, (gs, g_sync) = self._each_dis(branch, lambda: (post(), m['s_dis'](phon.detach(), g3, gen=True)),
                                          n=4 if self.ctc is not None else 3)
        gen_lo = lo[:3]
        # slot 0 gen_loss, 1 g_loss, 2 recon_loss, 3 g_sync_loss.  GRID (train.py:222-231): g_loss = adv/3 + g_sync,
        # recon = L1(denorm g_i, denorm mel_i)/3 + L1(gs, spec), gen = g_loss + 50 recon; LRS (train_LRS.py:226-237):
        # g_loss = adv/3, recon on the normalised mels, gen = g_loss + 50 recon + g_sync.
        hd = _Heads()
        for u, c in gen_lo:
            hd.add(ops.HEAD_SOFTPLUS, u, ((1, 1.0 / 3), (0, 1.0 / 3)), sign=-1.0)
            hd.add(ops.HEAD_SOFTPLUS, c, ((1, 1.0 / 3), (0, 1.0 / 3)), sign=-1.0)
        grid = self.variant == 'grid'
        hd.add(ops.HEAD_SUM, g_sync, ((3, 1.0), (1, 1.0), (0, 1.0)) if grid else ((3, 1.0), (0, 1.0)))
        k = (-_LOG_EPS / 2.0) if grid else 1.0      # slope of the de-normalisation applied to both L1 arguments
        for g, t in zip((g1, g2, g3), st['tgt']):
            hd.add(ops.HEAD_L1, g, ((2, 1.0 / 3), (0, 50.0 / 3)), y=t, scale=k / g.numel())
        hd.add(ops.HEAD_L1, gs, ((2, 1.0), (0, 50.0)), y=st['spec'])
        if self.ctc is not None:             # slot 4 ctc_loss: batch mean of the per-sample loss / target length
            hd.add(ops.HEAD_SUM, lo[3], ((4, 1.0), (0, float(self.ctc['weight']))))
        out = hd.run(5 if self.ctc is not None else 4)
        # No gradient buffer needs clearing here: the generator side was cleared at the start of the step and the D
        # phase's backward stops at the detached generator outputs and at phon_d, so it has added nothing to gen /
        # post; the discriminators' parameters take no gradient in this phase.
        with ops.direct_param_grad(), _quiet_leaf_streams():
            if st['trunk'] is None:
                torch.autograd.backward([out, phon], [self._onehot(out.numel(), 0, out.device), st['phon_d'].grad])
            else:
                out.backward(self._onehot(out.numel(), 0, out.device))
            self._join_dis()
        for p in d_params:
            p.requires_grad_(True)
        out = out.detach()
        cut = (st['trunk'], phon, st['phon_d']) if st['trunk'] is not None else None
        self._st = dict(dis_loss=st['dis_loss'], gen_loss=out[0], sync_loss=st['sync_loss'],
                        g_sync_loss=out[3], recon_loss=out[2], g_loss=out[1],
                        grad_p=st['grad_p'], g1=g1.detach(), g2=g2.detach(), g3=g3.detach(), gs=gs.detach())
        if self.ctc is not None:
            self._st['ctc_loss'] = out[4]
        self._cut = cut

    def g_trunk(self):
        """With overlap: the visual front-end's backward pass, from the sum of both phases' gradients at the phoneme
        features (see d_phase)."""
        cut, self._cut = getattr(self, '_cut', None), None
        if cut is None:
            return
        trunk, phon, phon_d = cut
        with ops.direct_param_grad(), _quiet_leaf_streams():
            trunk.backward(phon.grad + phon_d.grad)
            self._join_dis()

    def finish(self):
        """g_optimizer.step() (train.py:236); returns the step's losses and outputs."""
        self.g_opt.step(self.scaler, 1)
        if self.scaler is not None:
            self.scaler.update()
        ops.step_end(self._st['g3'].device)
        out, self._st = self._st, None
        return out


class GraphedStep:
    """A GanStep captured once into HIP graphs and replayed instead of several thousand kernel launches issued from
    Python per training step.  A single process captures the whole step as one graph; with data parallelism (or
    segmented=True) the phases of GanStep.schedule() (three, or five with overlapped exchanges) are graphs sharing
    one memory pool, replayed in order with the gradient all-reduces issued between them from the same stream, so
    no collective is ever captured.  Inputs are
    copied into static device buffers before each replay; the returned tensors are the graphs' static outputs
    (overwritten by the next call).  Shapes are fixed at capture.

    Decoder noise (generator.py:248 draws it afresh every step): with noise='device' (the default) the captured step
    draws it on the GPU from the counter-based generator, so every replay sees new noise; with a tensor the caller
    supplies the noise of every call (passing None on a later call is an error, never a silent reuse)."""

    def __init__(self, step, mel, spec, vid, vid_len, noise='device', warmup=2, segmented=None, text=None,
                 text_len=None, pool=None):
        """pool: a torch.cuda.graph_pool_handle() shared with other GraphedSteps whose replays never overlap (the
        length buckets of BucketedStep): their captures then reuse one another's activation memory."""
        assert warmup >= 1, 'one uncaptured step must run first: lazy state and the scratch arena settle there'
        if D_STREAMS and G.HEAD_STREAMS:
            # fork + join on a forked stream: hipStreamEndCapture of this runtime crashes the process (see
            # models/generator.py HEAD_STREAMS); every fork of a captured step must start from and return to the
            # capture's origin stream
            raise RuntimeError('a captured step cannot run the discriminators on streams of their own (D_STREAMS) and '
                               'fork helper streams from those streams (generator.HEAD_STREAMS): the HIP runtime '
                               'crashes when a capture joins a helper back into a forked stream')
        self.step = step
        dev = mel.device
        self.device_noise = isinstance(noise, str)
        self.static = dict(mel=mel.clone(), spec=spec.clone(), vid=vid.clone(),
                           vid_len=torch.as_tensor([int(l) for l in vid_len], device=dev))
        if not self.device_noise:
            self.static['noise'] = noise.clone()
        self.static['text'] = text.to(dev).clone() if text is not None else None
        self.static['text_len'] = torch.as_tensor(text_len, device=dev).clone() if text is not None else None
        self.segmented = step.d_exchange.active if segmented is None else bool(segmented)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator, lazy module state and kernel attributes settle here
                self.warm_out = self._run()  # (real training steps on the given batch; the last one's results)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        s = self.static
        nz = 'device' if self.device_noise else s['noise']
        if not self.segmented:
            self.graphs = [torch.cuda.CUDAGraph()]
            with torch.cuda.graph(self.graphs[0], pool=pool):
                self.out = self._run()
        else:
            # Other threads (the process group's watchdog) may touch the device while a phase is being captured.
            mode = 'thread_local' if step.d_exchange.active else 'global'
            pool = torch.cuda.graph_pool_handle() if pool is None else pool
            sched = step.schedule()
            first = lambda: step.d_phase(s['mel'], s['spec'], s['vid'], s['vid_len'], nz, s['text'], s['text_len'])
            self.exchanges = [ex for _, ex in sched]
            self.graphs = [torch.cuda.CUDAGraph() for _ in sched]
            for g, (phase, exchange) in zip(self.graphs, [(first, sched[0][1])] + sched[1:]):
                with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                    phase()
                exchange()                   # keeps the ranks' collective sequences aligned; the data is not used
            self.out = step._out
        self.warmup_steps = warmup

    def _run(self):
        s = self.static
        return self.step(s['mel'], s['spec'], s['vid'], s['vid_len'],
                         noise='device' if self.device_noise else s['noise'], text=s['text'], text_len=s['text_len'])

    def __call__(self, mel, spec, vid, vid_len=None, noise=None, text=None, text_len=None):
        s = self.static
        for k, v in (('text', text), ('text_len', text_len)):
            if v is not None:
                assert s[k] is not None, 'this step was captured without transcripts'
                s[k].copy_(torch.as_tensor(v), non_blocking=True)
        if self.device_noise:
            assert noise is None or isinstance(noise, str), 'this step was captured with on-device noise'
        else:
            assert torch.is_tensor(noise), 'this step was captured with caller-supplied noise: pass the tensor'
        for k, v in (('mel', mel), ('spec', spec), ('vid', vid), ('noise', noise)):
            if torch.is_tensor(v) and v.data_ptr() != s[k].data_ptr():
                s[k].copy_(v, non_blocking=True)
        if vid_len is not None and not (torch.is_tensor(vid_len) and vid_len.data_ptr() == s['vid_len'].data_ptr()):
            s['vid_len'].copy_(torch.as_tensor([int(l) for l in vid_len]), non_blocking=True)
        self.step.g_opt.sync_lr()            # the captured Adam launches read the learning rate from device words
        self.step.d_opt.sync_lr()
        if not self.segmented:
            self.graphs[0].replay()
        else:
            for g, exchange in zip(self.graphs, self.exchanges):
                g.replay()
                exchange()
        return self.out


LRS_BUCKETS = (50, 75, 100, 125, 150)


def bucket_of(n_frames, buckets=LRS_BUCKETS):
    """The padded clip length a clip of n_frames is trained at: the smallest bucket that holds it."""
    for b in buckets:
        if n_frames <= b:
            return b
    raise ValueError('a clip of %d frames exceeds the largest bucket (%d)' % (n_frames, buckets[-1]))


def bucket_batches(lengths, batch_size, buckets=LRS_BUCKETS, seed=None, drop_last=True):
    """Batch sampler for variable-length clips (the train_LRS.py loop pads every clip to one window,
    vid_aud_lrs2.py:106,171-188, so a 60-frame clip costs as much as a 150-frame one): sample indices are grouped by
    bucket_of(length) and each batch is drawn from one bucket, so it is padded to that bucket's length only.
    Returns [(T_bucket, [indices]), ...] in a (seeded) shuffled order, like DataLoader(shuffle=True) over batches."""
    import random
    rng = random.Random(seed)
    by = {}
    for i, n in enumerate(lengths):
        by.setdefault(bucket_of(int(n), buckets), []).append(i)
    out = []
    for T, idx in sorted(by.items()):
        if seed is not None:
            rng.shuffle(idx)
        for k in range(0, len(idx), batch_size):
            chunk = idx[k:k + batch_size]
            if len(chunk) == batch_size or not drop_last:
                out.append((T, chunk))
    if seed is not None:
        rng.shuffle(out)
    return out


class BucketedStep:
    """A GanStep for clips of several padded lengths: one captured step (GraphedStep) per bucket length, captured the
    first time a batch of that length arrives, all captures in one memory pool (replays are sequential, so the
    buckets reuse each other's activation memory).  The batch size is fixed; a batch's length is vid.shape[2] and must
    be one of `buckets` (bucket_batches groups a dataset that way; the valid lengths inside the batch still go through
    vid_len and reach the attention masks, generator.py:162-163).

    The first batch of a bucket is trained eagerly (it is the warm-up step every capture needs) and the capture
    follows; from its second batch on a bucket costs one graph replay."""

    def __init__(self, step, buckets=LRS_BUCKETS, noise='device'):
        assert isinstance(noise, str), 'bucketed steps draw the decoder noise on the device'
        self.step, self.buckets = step, tuple(buckets)
        self.pool = None
        self.graphed = {}

    def __call__(self, mel, spec, vid, vid_len, text=None, text_len=None):
        T = int(vid.shape[2])
        assert T in self.buckets, 'clip length %d is not one of the buckets %r' % (T, self.buckets)
        assert mel.shape[-1] == 4 * T and spec.shape[-1] == 4 * T
        gs = self.graphed.get(T)
        if gs is None:
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            gs = self.graphed[T] = GraphedStep(self.step, mel, spec, vid, vid_len, 'device', warmup=1, text=text,
                                               text_len=text_len, pool=self.pool)
            return gs.warm_out
        return gs(mel, spec, vid, vid_len, text=text, text_len=text_len)
