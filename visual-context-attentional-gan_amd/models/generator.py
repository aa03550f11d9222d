"""Generator, Postnet and discriminators (mirror of the reference's src/models/generator.py).

Public tensors keep the reference's shapes (mel-spectrograms (B, 1, F, T), sentence features (B, 512, T),
phonemes (B, T, 512)); inside a module feature maps are channels-last (B, F, T, C), which makes the
single-channel boundary tensors free views.  nn.Conv*/nn.Linear/nn.BatchNorm* objects only hold parameters
(same state_dict keys as the reference); the arithmetic runs through vcagan_amd.ops.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .resnet import BasicBlock

_RSQRT2 = 1.0 / math.sqrt(2.0)
SUBPIXEL_UPSAMPLE = True   # False: 5x5 convolution reading through the nearest x2 upsample (the reference's formulation)
TILED_INPUT = True     # False: build the (B, 20, T, 640) decoder input explicitly (the reference's formulation)
# True: a discriminator's conditional head runs on a helper stream beside its unconditional head (one helper per calling
# stream).  Off by default (no gain measured).  It must not be combined with a graph capture in which the discriminator
# itself runs on a forked stream: a forked stream that forks a helper and waits for it again (fork and join on a stream
# other than the capture's origin) makes hipStreamEndCapture of the ROCm 7.0 runtime crash the process (host-side
# segmentation fault; tools/probes/capture_streams_probe.py reproduces it with three streams and four torch kernels:
# `1 1 1`).  Forks from the origin stream, nested forks joined only to the origin, and waits between forked siblings all
# capture fine.  train_step.GraphedStep refuses the combination before it starts capturing.
HEAD_STREAMS = __import__('os').environ.get('VCA_D_HEAD_STREAMS') == '1'


def _conv(x, m, ups=0, **kw):
    return ops.conv2d(x, m.weight, m.bias, m.stride[0], m.padding[0], ups, **kw)


class ResBlk1D(nn.Module):
    """generator.py:8-49 (Postnet body; built without normalisation or down-sampling, :181-183)."""

    def __init__(self, dim_in, dim_out, actv=None, normalize=False, downsample=False):
        super().__init__()
        assert not normalize and not downsample
        self.learned_sc = dim_in != dim_out
        self.conv1 = nn.Conv1d(dim_in, dim_in, 5, 1, 2)
        self.conv2 = nn.Conv1d(dim_in, dim_out, 5, 1, 2)
        if self.learned_sc:
            self.conv1x1 = nn.Conv1d(dim_in, dim_out, 1, 1, 0, bias=False)

    def forward(self, x):                                     # (B, L, C)
        r = ops.conv1d(ops.leaky(x), self.conv1.weight, self.conv1.bias, 2)
        r = ops.conv1d(ops.leaky(r), self.conv2.weight, self.conv2.bias, 2)
        s = ops.conv1d(x, self.conv1x1.weight, None, 0) if self.learned_sc else x
        return ops.add_scale(s, r, _RSQRT2)


class ResBlk(nn.Module):
    """generator.py:51-92 (discriminator body; built with downsample=True and no normalisation, :283)."""

    def __init__(self, dim_in, dim_out, actv=None, normalize=False, downsample=False):
        super().__init__()
        assert not normalize
        self.downsample = downsample
        self.learned_sc = dim_in != dim_out
        self.conv1 = nn.Conv2d(dim_in, dim_in, 5, 1, 2)
        self.conv2 = nn.Conv2d(dim_in, dim_out, 5, 1, 2)
        if self.learned_sc:
            self.conv1x1 = nn.Conv2d(dim_in, dim_out, 1, 1, 0, bias=False)

    def forward(self, x):                                     # (B, H, W, C)
        # shortcut (generator.py:71-77: 1x1 convolution, then 2x2 average): both are linear and the convolution is
        # pointwise, so they commute; pooling first runs the convolution on a quarter of the positions
        a, s = ops.leaky_avgpool2(x) if self.downsample else (ops.leaky(x), x)
        s = _conv(s, self.conv1x1) if self.learned_sc else s
        r = _conv(a, self.conv1)
        if self.downsample:
            r = ops.avgpool2(r)
        r = _conv(ops.leaky(r), self.conv2)
        return ops.add_scale(s, r, _RSQRT2)


class GenResBlk(nn.Module):
    """generator.py:94-131."""

    def __init__(self, dim_in, dim_out, actv=None, upsample=False):
        super().__init__()
        self.upsample = upsample
        self.learned_sc = dim_in != dim_out
        self.conv1 = nn.Conv2d(dim_in, dim_out, 5, 1, 2)
        self.conv2 = nn.Conv2d(dim_out, dim_out, 5, 1, 2)
        self.norm1 = nn.BatchNorm2d(dim_in)
        self.norm2 = nn.BatchNorm2d(dim_out)
        if self.learned_sc:
            self.conv1x1 = nn.Conv2d(dim_in, dim_out, 1, 1, 0, bias=False)
        if upsample:
            # tap-merging matrices of the sub-pixel form: for output parity p, A[p] (3x5) sums the 5 kernel taps that
            # read the same source pixel of the nearest x2 upsampled map
            self.register_buffer('_tapmerge', torch.tensor(
                [[[1, 1, 0, 0, 0], [0, 0, 1, 1, 0], [0, 0, 0, 0, 1]],
                 [[1, 0, 0, 0, 0], [0, 1, 1, 0, 0], [0, 0, 0, 1, 1]]], dtype=torch.float32), persistent=False)

    def _forward_subpixel(self, x):
        """The up-sampling block (generator.py:112-126: nearest x2, then conv1 5x5 pad 2 / conv1x1) without reading
        every source pixel four times: output pixels of parity (py, px) see only 3x3 distinct source pixels, so the
        5x5 convolution of the upsampled map is four 3x3 pad-1 convolutions of the source map with tap-merged weights
        (computed here as one convolution with 4*Cout channels, then interleaved), 9 taps per output instead of 25;
        the 1x1 shortcut commutes with the upsample."""
        assert self.learned_sc and tuple(self.conv1.kernel_size) == (5, 5) and tuple(self.conv1.padding) == (2, 2)
        B, H, W, ci = x.shape
        co = self.conv1.out_channels
        A = self._tapmerge
        w = self.conv1.weight
        t = (w.view(co, ci, 1, 1, 5, 5) * A.view(1, 1, 2, 3, 5, 1)).sum(4)                 # (co, ci, py, a, kw)
        t = (t.view(co, ci, 2, 3, 1, 1, 5) * A.view(1, 1, 1, 1, 2, 3, 5)).sum(6)           # (co, ci, py, a, px, b)
        w3 = t.permute(2, 4, 0, 1, 3, 5).reshape(4 * co, ci, 3, 3)
        s, _, x = ops.conv2d_skip(x, self.conv1x1.weight, 1, 0)      # shortcut first: see forward()
        a = ops.bn_act(x, self.norm1, ops.LEAKY)
        y4 = ops.conv2d(a, w3, self.conv1.bias.repeat(4), 1, 1)                             # (B, H, W, (py, px, co))
        r = y4.view(B, H, W, 2, 2, co).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, co)
        r = _conv(ops.bn_act(r, self.norm2, ops.LEAKY), self.conv2)
        s = s.view(B, H, 1, W, 1, co).expand(B, H, 2, W, 2, co).reshape(B, 2 * H, 2 * W, co)
        return ops.add_scale(r, s, _RSQRT2)

    def forward(self, x):
        if self.upsample and SUBPIXEL_UPSAMPLE:
            return self._forward_subpixel(x)
        ups = 1 if self.upsample else 0
        s = None
        if self.learned_sc and not ups:
            # the block input feeds the shortcut convolution and the batch-norm branch: taking the latter from the
            # convolution's alias output lets its input-gradient kernel add the branch's gradient in its epilogue
            s, _, x = ops.conv2d_skip(x, self.conv1x1.weight, 1, 0)
        if not self.learned_sc:
            # identity shortcut: taken from the batch-norm node's alias output, whose backward kernel then adds the
            # shortcut's gradient while it writes the branch's
            a, x = ops.bn_act(x, self.norm1, ops.LEAKY, skip=True)
        else:
            a = ops.bn_act(x, self.norm1, ops.LEAKY)
        r, st = _conv(a, self.conv1, ups, want_stats=True)   # nearest x2 read inside the conv
        r = _conv(ops.bn_act(r, self.norm2, ops.LEAKY, stats=st), self.conv2)
        if s is None:
            if self.learned_sc:
                s = _conv(x, self.conv1x1, ups)
            else:
                assert not ups
                s = x
        return ops.add_scale(r, s, _RSQRT2)

    def forward_tiled(self, x, nz):
        """The block applied to cat([x tiled over the rows of nz, nz], channels) (generator.py:249-251) without
        forming the tiled map.  x: (B, W, C1), one row per clip; nz: (B, H, W, C2).  For the tiled channels every
        per-channel step (batch-norm, activation) acts on the single row, the 5x5 convolution is five 1x5
        convolutions of that row (kernel rows stacked in the output channels) summed per output row over the kernel
        rows that stay inside the map, and the 1x1 shortcut is a linear layer repeated over the rows."""
        assert self.learned_sc and not self.upsample
        B, H, W, _ = nz.shape
        C1 = x.shape[-1]
        co = self.conv1.out_channels
        ax, an = ops.bn_act_parts([(x, H), (nz, 1)], self.norm1, ops.LEAKY)
        w1 = self.conv1.weight
        kh, kw = w1.shape[2], w1.shape[3]
        w_rows = ops.param_view(w1, lambda t: t[:, :C1].permute(2, 0, 1, 3), (kh * co, C1, 1, kw))
        p = ops.conv2d(ax.view(B, 1, W, C1), w_rows, None, 1, (0, kw // 2))           # (B, 1, W, kh*co)
        r = ops.conv2d(an, ops.param_view(w1, lambda t: t[:, C1:]), self.conv1.bias, 1, (kh // 2, kw // 2))
        r = ops.hband_expand(p.view(B, W, kh * co), H, kh, kh // 2, base=r)
        r = _conv(ops.bn_act(r, self.norm2, ops.LEAKY), self.conv2)
        wsc = self.conv1x1.weight
        s = ops.conv2d(nz, ops.param_view(wsc, lambda t: t[:, C1:]), None, 1, 0)
        s = ops.hband_expand(ops.linear(x, ops.param_view(wsc, lambda t: t[:, :C1, 0, 0])), H, 1, 0, base=s)
        return ops.add_scale(r, s, _RSQRT2)


class Flatten(nn.Module):
    def forward(self, input):
        return input.view(input.size(0), -1)


class Avgpool(nn.Module):
    """generator.py:137-140; here over the two spatial axes of a channels-last map."""

    def forward(self, input):
        return input.mean([1, 2])


class AVAttention(nn.Module):
    """Visual Context Attention, generator.py:142-171."""

    def __init__(self, out_dim):
        super().__init__()
        self.k = nn.Linear(512, out_dim)
        self.v = nn.Linear(512, out_dim)
        self.q = nn.Linear(2560, out_dim)
        self.out_dim = out_dim
        self.mel = nn.Linear(out_dim, 20 * 64)

    def forward(self, ph, g, len):
        """ph: (B, S, 512) context; g: (B, F, T, C) channels-last decoder map; len: valid context lengths.
        Returns (B, F, T, 1280 // F) channels-last."""
        B, Fq, T, C = g.shape
        # query features are ordered c*F + f (generator.py:159); context features f*C' + c' (generator.py:169)
        qin = g.permute(0, 2, 3, 1).reshape(B, T, C * Fq)
        q = ops.linear(qin, self.q.weight, self.q.bias)
        k = ops.linear(ph, self.k.weight, self.k.bias)
        v = ops.linear(ph, self.v.weight, self.v.bias)
        ctx = ops.masked_attention(q, k, v, ops.as_lens(len, q.device), 1.0 / math.sqrt(self.out_dim))
        out = ops.linear(ctx, self.mel.weight, self.mel.bias)
        return out.view(B, T, Fq, -1).transpose(1, 2)


class Postnet(nn.Module):
    """generator.py:173-192."""

    def __init__(self):
        super().__init__()
        self.postnet = nn.Sequential(
            nn.Conv1d(80, 128, 7, 1, 3), nn.BatchNorm1d(128), nn.LeakyReLU(0.2),
            ResBlk1D(128, 256), ResBlk1D(256, 256), ResBlk1D(256, 256),
            nn.Conv1d(256, 321, 1, 1, 0, bias=False))

    def forward(self, x):                                      # (B, 1, 80, T)
        p = self.postnet
        h = x.squeeze(1).transpose(1, 2).contiguous()          # (B, T, 80)
        h, st = ops.conv1d(h, p[0].weight, p[0].bias, 3, want_stats=True)
        h = ops.bn_act(h, p[1], ops.LEAKY, stats=st)
        h = p[5](p[4](p[3](h)))
        h = ops.conv1d(h, p[6].weight, None, 0)                # (B, T, 321)
        return h.transpose(1, 2).unsqueeze(1)                  # (B, 1, 321, T)


def _to_mel(c):
    return nn.Sequential(nn.BatchNorm2d(c), nn.LeakyReLU(0.2), nn.Conv2d(c, 1, 1, 1, 0), nn.Tanh())


class Decoder(nn.Module):
    """generator.py:194-265."""

    def __init__(self):
        super().__init__()
        self.decode = nn.ModuleList()
        self.g1 = nn.ModuleList()
        self.g2 = nn.ModuleList()
        self.g3 = nn.ModuleList()
        self.att1 = AVAttention(256)
        self.attconv1 = nn.Conv2d(128 + 64, 128, 5, 1, 2)
        self.att2 = AVAttention(256)
        self.attconv2 = nn.Conv2d(64 + 32, 64, 5, 1, 2)
        self.to_mel1, self.to_mel2, self.to_mel3 = _to_mel(128), _to_mel(64), _to_mel(32)
        self.decode.extend([GenResBlk(512 + 128, 512), GenResBlk(512, 256), GenResBlk(256, 256)])
        self.g1.extend([GenResBlk(256, 128), GenResBlk(128, 128), GenResBlk(128, 128)])
        self.g2.extend([GenResBlk(128, 64, upsample=True), GenResBlk(64, 64), GenResBlk(64, 64)])
        self.g3.extend([GenResBlk(64, 32, upsample=True), GenResBlk(32, 32), GenResBlk(32, 32)])

    @staticmethod
    def _mel(head, h, skip=False):
        """to_mel head; skip=True also returns an alias of h for its other consumers (their gradients then arrive at
        the batch-norm node and are added by its backward kernel)."""
        a = ops.bn_act(h, head[0], ops.LEAKY, skip=skip)
        if skip:
            a, h = a
        y = _conv(a, head[2])                                           # (B, F, T, 1)
        y = ops.tanh(y).view(y.shape[0], 1, y.shape[1], y.shape[2])
        return (y, h) if skip else y

    def forward(self, s, x, len, noise=None):
        """s: (B, 512, T) sentence features; x: (B, T, 512) phonemes; len: valid frames per clip.
        noise: optional (B, 128, 20, T) tensor; by default drawn from the CPU generator like generator.py:248;
        the string 'device' draws it on the GPU instead (ops.randn: no host draw and no host-to-device copy, fresh
        values on every replay of a captured step; not torch's random stream)."""
        B, T = x.size(0), x.size(1)
        len = ops.as_lens(len, x.device)
        if isinstance(noise, str):
            assert noise == 'device', noise
            noise = ops.randn((B, 20, T, 128), x.device)                 # channels-last already
        else:
            if noise is None:
                noise = torch.randn([B, 128, 20, T])
            noise = noise.to(x.device).permute(0, 2, 3, 1)               # (B, 20, T, 128) channels-last view
        if TILED_INPUT:
            h = self.decode[0].forward_tiled(x, noise.contiguous())      # phonemes tiled | noise, never materialised
        else:
            h = self.decode[0](ops.concat2(x, noise, B, 20, T))          # (B, 20, T, 640)
        for block in list(self.decode)[1:]:
            h = block(h)
        for block in self.g1:
            h = block(h)
        m1, h = self._mel(self.to_mel1, h, skip=True)
        ctx = ops.ensure_ready(s).transpose(1, 2).contiguous()           # (B, T, 512); first use of s
        h = _conv(ops.concat2(h, self.att1(ctx, h, len), B, h.shape[1], h.shape[2]), self.attconv1)
        for block in self.g2:
            h = block(h)
        m2, h = self._mel(self.to_mel2, h, skip=True)
        h = _conv(ops.concat2(h, self.att2(ctx, h, len), B, h.shape[1], h.shape[2]), self.attconv2)
        for block in self.g3:
            h = block(h)
        return m1, m2, self._mel(self.to_mel3, h)


class Discriminator(nn.Module):
    """generator.py:267-317."""

    def __init__(self, num_class=1, max_conv_dim=512, phase='1'):
        super().__init__()
        dim_in = 32
        blocks = [nn.Conv2d(1, dim_in, 5, 1, 2)]
        repeat_num = {'1': 2, '2': 3}.get(phase, 4)
        for _ in range(repeat_num):                       # 80,4T -> 40,2T -> 20,T -> 10,T/2 -> 5,T/4
            dim_out = min(dim_in * 2, max_conv_dim)
            blocks.append(ResBlk(dim_in, dim_out, downsample=True))
            dim_in = dim_out
        self.main = nn.Sequential(*blocks)
        self.uncond = nn.Sequential(nn.LeakyReLU(0.2), nn.Conv2d(dim_out, dim_out, 5, 1, 0), nn.LeakyReLU(0.2),
                                    Avgpool(), nn.Linear(dim_out, num_class))
        self.cond = nn.Sequential(nn.LeakyReLU(0.2), nn.Conv2d(dim_out + 512, dim_out, 5, 1, 2), nn.LeakyReLU(0.2),
                                  nn.Conv2d(dim_out, dim_out, 5, 1, 0), nn.LeakyReLU(0.2),
                                  Avgpool(), nn.Linear(dim_out, num_class))

    def _cond_conv(self, ah, av):
        """cond[1] applied to cat([ah, av tiled over the map], channels) (generator.py:312-315) without forming the
        tiled channels: a vector repeated over the whole map contributes, per kernel tap, one matrix-vector product
        (all taps as one linear layer), and an output position sums the taps that stay inside the map, first over
        the kernel columns for every output column, then over the kernel rows for every output row."""
        conv = self.cond[1]
        B, H, W, C = ah.shape
        co, kh, kw = conv.out_channels, conv.kernel_size[0], conv.kernel_size[1]
        wc = conv.weight
        wq = ops.param_view(wc, lambda t: t[:, C:].permute(2, 3, 0, 1), (kh * kw * co, av.shape[-1]))
        q = ops.linear(av, wq)                                                                 # (B, kh*kw*co)
        qw = ops.hband_expand(q.view(B * kh, 1, kw * co), W, kw, conv.padding[1])              # (B*kh, W, 1, co)
        qw = qw.view(B, kh, W, co).permute(0, 2, 1, 3).reshape(B, W, kh * co)
        o = ops.conv2d(ah, ops.param_view(wc, lambda t: t[:, :C]), conv.bias, 1, tuple(conv.padding))
        return ops.hband_expand(qw, H, kh, conv.padding[0], base=o)

    def forward(self, x, c, vid_max_length):
        """x: (B, 1, F, M) mel; c: (B, 512, T) sentence features.  Returns (B, 1) unconditional and conditional logits."""
        B = x.size(0)
        f_len = final_length(vid_max_length)
        h = _conv(x.reshape(B, x.size(2), x.size(3), 1), self.main[0])
        for blk in list(self.main)[1:]:
            h = blk(h)
        ah = ops.leaky(h)

        def uncond():
            u = _conv(ah, self.uncond[1])
            return ops.linear(ops.spatial_mean(ops.leaky(u)), self.uncond[4].weight, self.uncond[4].bias)

        def cond():
            cvec = ops.WindowMeanFn.apply(c.reshape(B * 512, 1, c.size(2), 1), 1, c.size(2)).view(B, 512)  # c.mean(2)
            if TILED_INPUT:
                o = self._cond_conv(ah, ops.leaky(cvec))
            else:
                o = _conv(ops.leaky(ops.concat2(h, cvec, B, 5, f_len)), self.cond[1])
            o = _conv(ops.leaky(o), self.cond[3])
            return ops.linear(ops.spatial_mean(ops.leaky(o)), self.cond[6].weight, self.cond[6].bias)

        if HEAD_STREAMS and x.is_cuda:
            cur = torch.cuda.current_stream()
            helper = ops.side_stream(('dis_head', cur.cuda_stream))
            helper.wait_stream(cur)
            with torch.cuda.stream(helper):
                o = cond()
            u = uncond()
            cur.wait_stream(helper)
        else:
            u, o = uncond(), cond()
        return u.view(B, -1), o.view(B, -1)


class sync_Discriminator(nn.Module):
    """generator.py:319-361."""

    def __init__(self, temp=1.0):
        super().__init__()
        self.frontend = nn.Sequential(
            nn.Conv2d(1, 128, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1)), nn.BatchNorm2d(128), nn.PReLU(128),
            nn.Conv2d(128, 256, kernel_size=(3, 3), stride=(2, 2), padding=(1, 1)), nn.BatchNorm2d(256), nn.PReLU(256))
        self.Res_block = nn.Sequential(BasicBlock(256, 256))
        self.Linear = nn.Linear(256 * 20, 512)
        self.temp = temp

    def forward(self, v_feat, aud, gen=False):
        """v_feat: (B, S, 512); aud: (B, 1, 80, 4S).  Returns the per-sample loss (B,)."""
        B = aud.size(0)
        fe = self.frontend
        a = _conv(aud.reshape(B, aud.size(2), aud.size(3), 1), fe[0])
        a = ops.bn_act(a, fe[1], fe[2])
        a, st = _conv(a, fe[3], want_stats=True)
        a = ops.bn_act(a, fe[4], fe[5], stats=st)
        a = self.Res_block(a)                                             # (B, 20, S, 256)
        b, f, t, ch = a.shape
        a = a.permute(0, 2, 3, 1).reshape(b, t, ch * f)                   # feature index c*F + f (generator.py:344)
        a = ops.linear(a, self.Linear.weight, self.Linear.bias)          # (B, S, 512)
        return ops.sync_loss(v_feat, a, self.temp, gen)


def gan_loss(inputs, label=None):
    """generator.py:363-366."""
    return ops.softplus_mean(inputs, -1.0 if label else 1.0)


def final_length(vid_length):
    """generator.py:368-371."""
    return (vid_length // 2) // 2
