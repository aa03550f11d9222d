"""torch.autograd wrappers over the C ABI of libvca_hip.so.

Activations are fp32 channels-last tensors: (N, H, W, C) feature maps, (N, L, C) sequences and
(P, C) matrices.  Weights stay in the reference's parameter layout ((Cout, Cin, KH, KW) /
(Cout, Cin, K) / (Cout, Cin)) so state_dicts are interchangeable; they are re-laid-out on the
device into the kernels' [tap][Cin][Cout] form once per parameter version.

Every Function's backward is itself built from differentiable Functions (forward conv, input
gradient and weight gradient form a closed family), which is what the R1 penalty's
autograd.grad(create_graph=True) (train.py:188-194) needs.
"""
import weakref

import torch

from . import _lib
from ._lib import ConvDesc, check, ptr, stream

ACT_NONE, ACT_LEAKY, ACT_PRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3, 4


# --------------------------------------------------------------------------- #
# weight packing
# --------------------------------------------------------------------------- #
_pack_cache = {}     # id(parameter) -> (weakref, {mode: (version, packed tensor)})


def _wdims(w):
    cout, cin = w.shape[0], w.shape[1]
    taps = 1
    for s in w.shape[2:]:
        taps *= s
    return cout, cin, taps


def _pack(w, mode):
    """mode 0: forward layout [taps][Cin][Cout4]; mode 1: input-gradient layout [taps][Cout][Cin4]."""
    cout, cin, taps = _wdims(w)
    cacheable = isinstance(w, torch.nn.Parameter)
    if cacheable:
        ent = _pack_cache.get(id(w))
        if ent is not None and ent[0]() is w and ent[1].get(mode, (None,))[0] == w._version:
            return ent[1][mode][1]
    wd = w.detach().contiguous()
    row = ((cout + 3) // 4 * 4) if mode == 0 else ((cin + 3) // 4 * 4)
    out = torch.empty((taps, cin if mode == 0 else cout, row), device=w.device, dtype=torch.float32)
    check(_lib.lib().vca_pack_weight_f32(mode, cout, cin, taps, ptr(wd), ptr(out), 0.0, stream()), 'pack_weight')
    if cacheable:
        key = id(w)
        ent = _pack_cache.get(key)
        if ent is None or ent[0]() is not w:
            ent = (weakref.ref(w, lambda _r, key=key: _pack_cache.pop(key, None)), {})
            _pack_cache[key] = ent
        ent[1][mode] = (w._version, out)
    return out


class ConvCfg:
    """Static geometry of one convolution call (2-D maps; KD > 1 only for the single-channel stem)."""

    def __init__(self, wshape, xshape, stride=1, pad=(0, 0), ups=0, pad_d=0, pshape=None):
        self.wshape = tuple(wshape)
        self.pshape = tuple(pshape) if pshape is not None else tuple(wshape)   # the parameter's own shape
        self.cout, self.cin = wshape[0], wshape[1]
        k = tuple(wshape[2:])
        self.kd, self.kh, self.kw = ((1, 1, 1) + k)[-3:]
        self.stride, self.ups, self.pad_d = stride, ups, pad_d
        self.pad_h, self.pad_w = pad
        # x: (N, [D,] H, W, C)
        if len(xshape) == 5:
            self.n, self.d, self.hin, self.win = xshape[:4]
        else:
            self.n, self.hin, self.win = xshape[:3]
            self.d = 1
        assert xshape[-1] == self.cin, (xshape, wshape)
        he, we = self.hin << ups, self.win << ups
        self.od = self.d + 2 * pad_d - self.kd + 1
        self.oh = (he + 2 * self.pad_h - self.kh) // stride + 1
        self.ow = (we + 2 * self.pad_w - self.kw) // stride + 1
        self.xshape = tuple(xshape)
        self.yshape = (self.n,) + ((self.od,) if len(xshape) == 5 else ()) + (self.oh, self.ow, self.cout)

    def desc(self, out_scale=1.0):
        return ConvDesc(self.n, self.d, self.hin, self.win, self.cin, self.od, self.oh, self.ow, self.cout,
                        self.kd, self.kh, self.kw, self.stride, self.pad_h, self.pad_w, self.pad_d, 1, self.ups,
                        0, 0.0, out_scale)

    def dgrad_desc(self):
        """The input gradient as a stride-1 convolution over the (zero-dilated) output gradient."""
        assert self.kd == 1
        he, we = self.hin << self.ups, self.win << self.ups
        return ConvDesc(self.n, 1, self.oh, self.ow, self.cout, 1, he, we, self.cin,
                        1, self.kh, self.kw, 1, self.kh - 1 - self.pad_h, self.kw - 1 - self.pad_w, 0,
                        self.stride, 0, 0, 0.0, 1.0)


def _conv_fwd_raw(x, w, bias, cfg):
    y = torch.empty(cfg.yshape, device=x.device, dtype=torch.float32)
    d = cfg.desc()
    check(_lib.lib().vca_conv_fwd_f32(d, ptr(x), ptr(_pack(w, 0)), ptr(bias), None, None, None, None, ptr(y), None,
                                      stream()), 'conv_fwd')
    return y


def _conv_dgrad_raw(gy, w, cfg):
    he, we = cfg.hin << cfg.ups, cfg.win << cfg.ups
    gx = torch.empty((cfg.n, he, we, cfg.cin), device=gy.device, dtype=torch.float32)
    d = cfg.dgrad_desc()
    check(_lib.lib().vca_conv_fwd_f32(d, ptr(gy), ptr(_pack(w, 1)), None, None, None, None, None, ptr(gx), None,
                                      stream()), 'conv_dgrad')
    if cfg.ups:
        gx = gx.view(cfg.n, cfg.hin, 2, cfg.win, 2, cfg.cin).sum((2, 4))
    return gx.view(cfg.xshape)


def _conv_wgrad_raw(x, gy, cfg):
    cout, cin, taps = cfg.cout, cfg.cin, cfg.kd * cfg.kh * cfg.kw
    dwp = torch.zeros((taps, cin, (cout + 3) // 4 * 4), device=x.device, dtype=torch.float32)
    check(_lib.lib().vca_conv_wgrad_f32(cfg.desc(), ptr(x), ptr(gy), None, None, None, ptr(dwp), stream()),
          'conv_wgrad')
    gw = torch.empty(cfg.pshape, device=x.device, dtype=torch.float32)
    check(_lib.lib().vca_pack_weight_f32(2, cout, cin, taps, ptr(dwp), ptr(gw), 0.0, stream()), 'unpack_wgrad')
    return gw


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, cfg):
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w)
        return _conv_fwd_raw(_c(x), w, bias, cfg)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        cfg = ctx.cfg
        gy = _c(gy)
        gx = ConvDgradFn.apply(gy, w, cfg) if ctx.needs_input_grad[0] else None
        gw = ConvWgradFn.apply(x, gy, cfg) if ctx.needs_input_grad[1] else None
        gb = gy.reshape(-1, cfg.cout).sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None


class ConvDgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, w, cfg):
        ctx.cfg = cfg
        ctx.save_for_backward(gy, w)
        return _conv_dgrad_raw(_c(gy), w, cfg)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        cfg = ctx.cfg
        ggx = _c(ggx)
        ggy = ConvFn.apply(ggx, w, None, cfg) if ctx.needs_input_grad[0] else None
        gw = ConvWgradFn.apply(ggx, gy, cfg) if ctx.needs_input_grad[1] else None
        return ggy, gw, None


class ConvWgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gy, cfg):
        ctx.cfg = cfg
        ctx.save_for_backward(x, gy)
        return _conv_wgrad_raw(_c(x), _c(gy), cfg)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        cfg = ctx.cfg
        ggw = _c(ggw)
        gx = ConvDgradFn.apply(gy, ggw, cfg) if ctx.needs_input_grad[0] else None
        ggy = ConvFn.apply(x, ggw, None, cfg) if ctx.needs_input_grad[1] else None
        return gx, ggy, None


def conv2d(x, w, bias=None, stride=1, pad=0, ups=0):
    """x: (N,H,W,Cin) channels-last; w: (Cout,Cin,KH,KW).  Returns (N,OH,OW,Cout)."""
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    cfg = ConvCfg(w.shape, x.shape, stride, pad, ups)
    return ConvFn.apply(x, w, bias, cfg)


def conv1d(x, w, bias=None, pad=0):
    """x: (N,L,Cin); w: (Cout,Cin,K).  Returns (N,L',Cout)."""
    n, l, c = x.shape
    cfg = ConvCfg((w.shape[0], w.shape[1], 1, w.shape[2]), (n, 1, l, c), 1, (0, pad), 0, pshape=w.shape)
    y = ConvFn.apply(x.reshape(n, 1, l, c), w, bias, cfg)
    return y.view(n, cfg.ow, w.shape[0])


def linear(x, w, bias=None):
    """x: (..., Cin); w: (Cout, Cin)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, 1, 1, x.shape[-1])
    cfg = ConvCfg((w.shape[0], w.shape[1], 1, 1), x2.shape, 1, (0, 0), 0, pshape=w.shape)
    return ConvFn.apply(x2, w, bias, cfg).view(*lead, w.shape[0])


def conv3d_stem(x, w, stride=2, pad=(2, 3, 3)):
    """x: (N,D,H,W,1); w: (Cout,1,KD,KH,KW), spatial stride `stride`, temporal stride 1 (visual_front.py:11)."""
    cfg = ConvCfg(w.shape, x.shape, stride, (pad[1], pad[2]), 0, pad_d=pad[0])
    return ConvFn.apply(x, w, None, cfg)


# --------------------------------------------------------------------------- #
# per-channel / pointwise / pooling / recurrent / loss operators
# --------------------------------------------------------------------------- #
LEAKY = 'leaky0.2'      # the generator/discriminator activation (generator.py:9,52,95: nn.LeakyReLU(0.2))


def _act(y, act):
    F = torch.nn.functional
    if act is None:
        return y
    if act == LEAKY:
        return F.leaky_relu(y, 0.2)
    if isinstance(act, torch.nn.PReLU):
        return torch.where(y > 0, y, y * act.weight)
    if isinstance(act, torch.nn.ReLU):
        return F.relu(y)
    raise ValueError(act)


def bn_act(x, bn, act=None, residual=None):
    """act(batch_norm(x) + residual) over the last (channel) axis of a channels-last tensor."""
    C = x.shape[-1]
    y = torch.nn.functional.batch_norm(x.reshape(-1, C), bn.running_mean, bn.running_var, bn.weight, bn.bias,
                                       bn.training, bn.momentum, bn.eps).view(x.shape)
    if bn.training:
        bn.num_batches_tracked += 1
    if residual is not None:
        y = y + residual
    return _act(y, act)


def leaky(x):
    return torch.nn.functional.leaky_relu(x, 0.2)


def tanh(x):
    return torch.tanh(x)


def add_scale(a, b, scale):
    return (a + b) * scale


def avgpool2(x):
    """F.avg_pool2d(x, 2) on a channels-last map (floor on odd sizes)."""
    n, h, w, c = x.shape
    h2, w2 = h // 2, w // 2
    return x[:, :h2 * 2, :w2 * 2].reshape(n, h2, 2, w2, 2, c).mean((2, 4))


def maxpool3x3s2(x):
    """MaxPool (3x3, stride 2, pad 1) on a channels-last map (visual_front.py:14)."""
    y = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1)
    return y.permute(0, 2, 3, 1).contiguous()


def head_pool4(x):
    """nn.AvgPool2d(4) of resnet.py:82 on a channels-last map whose output is 1x1: the mean of the top-left 4x4
    window when the map is at least 4 wide, the global mean otherwise (96x96 / 48x48 clips)."""
    n, h, w, c = x.shape
    if h >= 4 and w >= 4:
        assert h < 8 and w < 8, 'AvgPool2d(4) output must be 1x1 (visual_front.py:30)'
        return x[:, :4, :4].mean((1, 2))
    return x.mean((1, 2))


def gru(x, mod, training):
    """x: (B, T, 512) -> (B, T, 1024); 2-layer bidirectional GRU (visual_front.py:20,33-34)."""
    out, _ = mod(x.transpose(0, 1).contiguous())
    return out.transpose(0, 1)


def masked_attention(q, k, v, lens, scale):
    """softmax(q k^T * scale, keys >= lens[b] masked) v   (generator.py:161-167)."""
    att = torch.bmm(q, k.transpose(1, 2)) * scale
    S = att.shape[2]
    lens_t = torch.as_tensor([int(l) for l in lens], device=att.device)
    pad = torch.arange(S, device=att.device)[None, None, :] >= lens_t[:, None, None]
    att = att.masked_fill(pad, float('-inf')).softmax(2)
    return torch.bmm(att, v)


def sync_loss(v_feat, a_feat, temp, gen):
    """generator.py:347-361."""
    F = torch.nn.functional
    if gen:
        return 5.0 - F.cosine_similarity(v_feat, a_feat, 2).abs().mean(1)
    sim = torch.bmm(F.normalize(v_feat, dim=2), F.normalize(a_feat, dim=2).transpose(1, 2)) / temp
    d_va = torch.diagonal(F.log_softmax(sim, 2), dim1=-2, dim2=-1).mean(1)
    d_av = torch.diagonal(F.log_softmax(sim, 1), dim1=-2, dim2=-1).mean(1)
    return -0.5 * (d_va + d_av)


def softplus_mean(x, sign):
    return torch.nn.functional.softplus(sign * x).mean()
