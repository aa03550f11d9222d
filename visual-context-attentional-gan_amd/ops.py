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
_pack_cache = weakref.WeakKeyDictionary()


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
        ent = _pack_cache.get(w)
        if ent is not None and ent.get(mode, (None,))[0] == w._version:
            return ent[mode][1]
    wd = w.detach().contiguous()
    row = ((cout + 3) // 4 * 4) if mode == 0 else ((cin + 3) // 4 * 4)
    out = torch.empty((taps, cin if mode == 0 else cout, row), device=w.device, dtype=torch.float32)
    check(_lib.lib().vca_pack_weight_f32(mode, cout, cin, taps, ptr(wd), ptr(out), 0.0, stream()), 'pack_weight')
    if cacheable:
        _pack_cache.setdefault(w, {})[mode] = (w._version, out)
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
