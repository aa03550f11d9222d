"""torch.autograd wrappers over the C ABI of libvca_hip.so.

Activations are fp32 channels-last tensors: (N, H, W, C) feature maps, (N, L, C) sequences and
(P, C) matrices.  Weights stay in the reference's parameter layout ((Cout, Cin, KH, KW) /
(Cout, Cin, K) / (Cout, Cin)) so state_dicts are interchangeable; they are re-laid-out on the
device into the kernels' [tap][Cin][Cout] form once per parameter version.

Every Function's backward is itself built from differentiable Functions (forward conv, input
gradient and weight gradient form a closed family), which is what the R1 penalty's
autograd.grad(create_graph=True) (train.py:188-194) needs.
"""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import ConvDesc, check, ptr, stream

ACT_NONE, ACT_LEAKY, ACT_PRELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3, 4
once = torch.autograd.function.once_differentiable


# --------------------------------------------------------------------------- #
# weight packing
# --------------------------------------------------------------------------- #
_param_epoch = [0]   # bumped whenever parameters are rewritten behind autograd's back (re-homing, checkpoint load)
_pack_cache = {}     # id(parameter) -> (weakref, {mode: (version, packed tensor)})


# Small zero-initialised scratch (batch-norm partial sums, convolution statistics): inside a training step they are
# carved from one arena that a single fill clears at the start of the step, instead of one fill launch each.
_arena = {}          # device -> [buffer, offset, active, used by the previous steps, length cleared for this step]
_ARENA_FLOATS = 16 << 20
_nbt_pending = {}    # device -> {id(counter): [counter tensor, increments]}: batch-norm step counters bumped at step_end


def step_begin(device):
    """Start of a training step: clear the part of the scratch arena the previous steps used and hand it out again.
    Inside the step only the cleared part is handed out (a request beyond it gets its own zero fill), so a captured
    step, whose fill length is frozen into the graph, stays correct whatever it carves out; after one eager step of
    the same shapes (GraphedStep insists on one) the cleared part covers everything."""
    a = _arena.get(device)
    if a is None:
        a = _arena[device] = [torch.zeros(_ARENA_FLOATS, device=device, dtype=torch.float32), 0, False, 0, 0]
    _flush_nbt(device)
    a[3] = max(a[3], a[1])
    n = min(_ARENA_FLOATS, (a[3] + (1 << 18)) if a[3] else _ARENA_FLOATS)   # slack: a step may carve a little more
    a[0][:n].zero_()
    a[1], a[2], a[4] = 0, True, n


def step_end(device):
    a = _arena.get(device)
    if a is not None:
        a[3] = max(a[3], a[1])
        a[2] = False
    _flush_nbt(device)


def _zeros(shape, device):
    """Zero-filled fp32 scratch that is consumed within the current forward/backward call."""
    a = _arena.get(device)
    n = 1
    for d in shape:
        n *= d
    if a is not None and a[2]:
        off = a[1]
        a[1] = off + (n + 63) // 64 * 64      # counted even when it does not fit, so the next step clears enough
        if off + n <= a[4]:
            return a[0][off:off + n].view(shape)
    return torch.zeros(shape, device=device, dtype=torch.float32)


def _bump_nbt(bn):
    """num_batches_tracked += 1 (nn.BatchNorm's train-mode forward).  Inside a training step the increments of all
    batch-norm layers are collected and applied by one multi-tensor launch when the step ends."""
    t = bn.num_batches_tracked
    a = _arena.get(t.device)
    if a is not None and a[2]:
        ent = _nbt_pending.setdefault(t.device, {})
        e = ent.get(id(t))
        if e is None:
            ent[id(t)] = [t, 1]
        else:
            e[1] += 1
    else:
        t += 1


def _flush_nbt(device):
    ent = _nbt_pending.pop(device, None)
    if not ent:
        return
    by_count = {}
    for t, k in ent.values():
        by_count.setdefault(k, []).append(t)
    for k, ts in by_count.items():
        torch._foreach_add_(ts, k)


def _wkey(w):
    """Validity key of a packed copy of w: autograd's version counter, the global epoch, and the epoch of the flat
    buffer the parameter lives in (bumped by the fused Adam step that rewrites exactly that buffer, so one
    network's step does not invalidate the packed weights of the others)."""
    ep = getattr(w, '_vca_epoch', None)
    return (w._version, _param_epoch[0], ep[0] if ep is not None else 0)


def _wdims(w):
    cout, cin = w.shape[0], w.shape[1]
    taps = 1
    for s in w.shape[2:]:
        taps *= s
    return cout, cin, taps


_precision = ['fp32']


_FP16_MODES = {'fp16x3': 'bf16x3', 'fp16': 'bf16', 'fp16-mixed': 'mixed'}


def set_precision(mode):
    """Arithmetic of the matrix-core convolutions: 'fp32' (exact fp32 MFMA), 'bf16x3' (split-bf16, three bf16 MFMAs
    per product, fp32 accumulate: ~fp32 accuracy), 'bf16' (single bf16 product) or 'mixed' (bf16x3 in every forward
    product and in the gradient chain that yields the R1 penalty, so outputs and loss values keep fp32-level parity,
    and single bf16 products in the input/weight gradients, as bf16 mixed-precision training does).
    'fp16x3' / 'fp16' / 'fp16-mixed' are the same three schemes on IEEE half operands (the library's second build):
    11 significant bits per product instead of 8, but a normal range that ends at 6e-5, so a training step in these
    modes seeds its backward passes with a loss scale (LossScaler)."""
    assert mode in ('fp32', 'bf16x3', 'bf16', 'mixed') or mode in _FP16_MODES, mode
    elem = 'fp16' if mode in _FP16_MODES else 'bf16'
    if mode != 'fp32' and elem != _lib.ELEM[0]:
        _lib.ELEM[0] = elem
        _param_epoch[0] += 1          # packed weights are element-type specific: every cached copy is stale
    _precision[0] = _FP16_MODES.get(mode, mode)


def get_precision():
    """The mode set_precision() was given."""
    if _lib.ELEM[0] == 'fp16' and _precision[0] != 'fp32':
        return {v: k for k, v in _FP16_MODES.items()}[_precision[0]]
    return _precision[0]


def half_gradients():
    """True when gradients go through IEEE-half products (and therefore need a loss scale)."""
    return _lib.ELEM[0] == 'fp16' and _precision[0] != 'fp32'


def _pack(w, mode):
    """mode 0: forward layout [taps][Cin][Cout4]; mode 1: input-gradient layout [taps][Cout][Cin4];
    modes 10 / 11: the same two roles in the split-bf16 layout [tap][chunk][n][hi 32 | lo 32]; mode 12: the
    sub-pixel input-gradient layout of a 3x3 stride-2 convolution (4 taps, 4*Cin columns)."""
    cout, cin, taps = _wdims(w)
    cacheable = isinstance(w, torch.nn.Parameter)
    if cacheable:
        ent = _pack_cache.get(id(w))
        if ent is not None and ent[0]() is w and ent[1].get(mode, (None,))[0] == _wkey(w):
            return ent[1][mode][1]
    wd = w.detach().contiguous()
    # A stale packed copy of the same parameter is overwritten in place: its address stays fixed, which captured
    # graphs rely on (a convolution replayed early in a step reads what a pack replayed late in the previous
    # step wrote).
    old = None
    if cacheable:
        ent = _pack_cache.get(id(w))
        if ent is not None and ent[0]() is w and mode in ent[1]:
            old = ent[1][mode][1]
    if mode >= 10:
        nbytes = _lib.lib().vca_pack_weight_bf16_bytes(mode - 10, cout, cin, taps)
        out = old if (old is not None and old.numel() == nbytes and old.device == w.device) else \
            torch.empty(nbytes, device=w.device, dtype=torch.uint8)
        check(_lib.lib().vca_pack_weight_bf16(mode - 10, cout, cin, taps, ptr(wd), out.data_ptr(), stream()),
              'pack_weight_bf16')
    else:
        row = ((cout + 3) // 4 * 4) if mode == 0 else ((cin + 3) // 4 * 4)
        shape = (taps, cin if mode == 0 else cout, row)
        out = old if (old is not None and tuple(old.shape) == shape and old.device == w.device) else \
            torch.empty(shape, device=w.device, dtype=torch.float32)
        check(_lib.lib().vca_pack_weight_f32(mode, cout, cin, taps, ptr(wd), ptr(out), 0.0, stream()), 'pack_weight')
    if cacheable:
        key = id(w)
        ent = _pack_cache.get(key)
        if ent is None or ent[0]() is not w:
            ent = (weakref.ref(w, lambda _r, key=key: _pack_cache.pop(key, None)), {})
            _pack_cache[key] = ent
        ent[1][mode] = (_wkey(w), out)
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
        self.key = (self.wshape, self.xshape, stride, self.pad_h, self.pad_w, ups)

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

    def subpixel_dgrad(self):
        """True for the 3x3 / stride 2 / padding 1 convolutions (resnet.py:5-9 stage entries), whose input gradient
        runs as a 2x2 convolution of the undilated output gradient with a depth-to-space epilogue."""
        return (SUBPIXEL_DGRAD and self.kd == 1 and self.stride == 2 and self.kh == self.kw == 3 and
                self.pad_h == self.pad_w == 1 and not self.ups and self.cout > 1)

    def subpixel_dgrad_desc(self):
        return ConvDesc(self.n, 1, self.oh, self.ow, self.cout, 1, self.oh, self.ow, 4 * self.cin,
                        1, 2, 2, 1, 0, 0, 0, 1, 0, 0, 0.0, 1.0, self.hin, self.win)


SUBPIXEL_DGRAD = True   # False: stride-2 input gradients over the zero-dilated output gradient (the general form)
_exact_bwd = [False]   # True while a backward runs under create_graph (its results feed a loss value: R1 penalty)
_grad_conv = [False]   # True while a forward-type convolution is issued as part of a gradient (the second-order terms
                       # of the R1 penalty in the discriminators' backward pass): 'mixed' runs it like every other
                       # gradient, with a single product


# 'mixed' runs the convolutions that differentiate an input-gradient node (the second-order terms of the R1 penalty in
# the discriminators' backward pass) with one bf16 product, like every other gradient.  R1_SECOND_ORDER_SPLIT[0] = True
# (VCA_R1_SPLIT=1) runs them with three: the A/B switch behind the measured D-side gradient figures.
R1_SECOND_ORDER_SPLIT = [__import__('os').environ.get('VCA_R1_SPLIT') == '1']


class use_precision:
    """Context manager: run the enclosed matrix-core convolutions at the given precision."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_precision()
        set_precision(self.mode)

    def __exit__(self, *exc):
        set_precision(self.prev)


def _nsplit(backward=False):
    if _precision[0] == 'mixed':
        return 1 if (backward and not _exact_bwd[0]) else 3
    return {'fp32': 0, 'bf16x3': 3, 'bf16': 1}[_precision[0]]


_ws_floats = {}   # (role, geometry) -> workspace floats of that launch
CONV_STAT_SLOTS = 32   # VCA_CONV_STAT_SLOTS: partial buffers of the statistics a convolution epilogue produces


def _split_ws(cfg, role, d, device):
    """(tensor, floats): a workspace for the partial sums of the matrix-core launch of descriptor d, (None, 0) for a
    launch that does not split its reduction over workgroups.  The size is asked from the library once per geometry.
    The caller holds the tensor until the launch has been issued (the allocator may hand the memory to the next
    allocation on the stream as soon as it is released)."""
    n = _ws_floats.get((role, cfg.key))
    if n is None:
        n = _ws_floats[(role, cfg.key)] = int(_lib.lib().vca_conv_fwd_bf16_ws_floats(d))
    if not n:
        return None, 0
    return torch.empty(n, device=device, dtype=torch.float32), n


LINEAR_KERNEL = __import__('os').environ.get('VCA_LINEAR', '1') != '0'    # False: pointwise convolutions / linear layers on the general convolution kernel (A/B switch)


# The linear kernel streams both operands from L2 without an LDS stage: no fixed cost, but three times the L2 bytes
# per MFMA of the convolution kernel's 128 x 128 tiles.  Device time per launch (graph replay, tools/conv_bench.py):
# 20-40 % shorter up to ~1 GFLOP (discriminator heads and shortcuts, 512 -> 256 projections: 10-19 us against 17-27),
# level at 2-4 GFLOP, up to 2x longer beyond (2400 x 5120 -> 512: 177 against 98 us).
LINEAR_MAX_FLOP = 1.5e9


def _pointwise(cfg, k, rows=None, n=None):
    """True for a 1x1 / stride-1 / unpadded convolution (a linear layer over the rows of a channels-last matrix) that
    goes to the linear kernel: reduction length k a multiple of 8 and a product small enough for it to win."""
    if not (LINEAR_KERNEL and cfg.kd == cfg.kh == cfg.kw == 1 and cfg.stride == 1 and not cfg.ups and
            cfg.pad_h == 0 and cfg.pad_w == 0 and k % 8 == 0):
        return False
    return rows is None or 2.0 * rows * n * k <= LINEAR_MAX_FLOP or _tiled_wins(rows, n, k)


GEMM_KERNEL = __import__('os').environ.get('VCA_GEMM', '1') != '0'   # False: large pointwise products on the conv kernel
_gemm_ws = {}    # (M, N, K, views) -> workspace floats of the tiled kernel's reduction split


def _tiled_wins(rows, n, k):
    """Large pointwise products where the 128 x 128 LDS-ring kernel measured shorter than the convolution kernel: a
    reduction of at least 16 chunks of 32 and at most two rounds of tiles over the 256 CUs (beyond that the convolution
    kernel's two workgroups per CU hide the short reduction's prologue better)."""
    return GEMM_KERNEL and k >= 512 and ((rows + 127) // 128) * ((n + 127) // 128) <= 512


def _linear_raw(a, wpack, bias, res, out, M, N, K, ns, a_view=None, c_view=None, tiled=None):
    """out[m][n] = sum_k a[m][k] * W[n][k] (+ bias) (+ res); a_view / c_view: optional (tm, bs, ts, ck, ks) strided
    views (see vca_lin_desc); default: plain row-major matrices.  tiled: True = the 128 x 128 LDS-ring kernel, False =
    the LDS-free skinny kernel, None = by size."""
    atm, abs_, ats, ack, aks = a_view if a_view is not None else (M, 0, K, K, 0)
    ctm, cbs, cts, ccn, cns = c_view if c_view is not None else (M, 0, N, N, 0)
    d = _lib.LinDesc(M, N, K, atm, ack, abs_, ats, aks, ctm, ccn, cbs, cts, cns)
    if tiled is None:
        tiled = 2.0 * M * N * K > LINEAR_MAX_FLOP and _tiled_wins(M, N, K)
    if tiled:
        key = (M, N, K, a_view, c_view)
        n = _gemm_ws.get(key)
        if n is None:
            n = _gemm_ws[key] = int(_lib.lib().vca_gemm_bf16_ws_floats(ctypes.byref(d)))
        ws = torch.empty(n, device=out.device, dtype=torch.float32) if n else None
        check(_lib.lib().vca_gemm_bf16(ctypes.byref(d), ns, ptr(a), wpack.data_ptr(), ptr(bias), ptr(res), ptr(out),
                                       ptr(ws), n, stream()), 'gemm_bf16')
        return out
    check(_lib.lib().vca_linear_bf16(ctypes.byref(d), ns, ptr(a), wpack.data_ptr(), ptr(bias), ptr(res), ptr(out),
                                     stream()), 'linear_bf16')
    return out


def _conv_fwd_raw(x, w, bias, cfg, stats=None):
    """stats: optional zeroed (2, Cout) tensor receiving per-channel sum / sum of squares of (y - bias)."""
    y = torch.empty(cfg.yshape, device=x.device, dtype=torch.float32)
    d = cfg.desc()
    ns = _nsplit(_grad_conv[0])
    if ns and stats is None and _pointwise(cfg, cfg.cin, x.numel() // cfg.cin, cfg.cout):
        return _linear_raw(x, _pack(w, 10), bias, None, y, x.numel() // cfg.cin, cfg.cout, cfg.cin, ns)
    if ns and cfg.cin > 1 and cfg.kd == 1:
        ws, nws = (None, 0) if stats is not None else _split_ws(cfg, 'fwd', d, x.device)
        check(_lib.lib().vca_conv_fwd_bf16_ws(d, ns, ptr(x), _pack(w, 10).data_ptr(), ptr(bias), None, None, None, None,
                                              ptr(y), ptr(stats), ptr(ws), nws, stream()), 'conv_fwd_bf16')
    else:
        check(_lib.lib().vca_conv_fwd_f32(d, ptr(x), ptr(_pack(w, 0)), ptr(bias), None, None, None, None, ptr(y),
                                          ptr(stats), stream()), 'conv_fwd')
    return y


def _conv_dgrad_raw(gy, w, cfg, res=None):
    """Input gradient; `res` (a tensor of the input's shape) is added in the kernel epilogue."""
    he, we = cfg.hin << cfg.ups, cfg.win << cfg.ups
    gx = torch.empty((cfg.n, he, we, cfg.cin), device=gy.device, dtype=torch.float32)
    d = cfg.dgrad_desc()
    ns = _nsplit(True)
    fuse = res is not None and not cfg.ups
    r = _c(res) if fuse else None
    if ns and cfg.cout > 1 and _pointwise(cfg, cfg.cout, gy.numel() // cfg.cout, cfg.cin):
        _linear_raw(gy, _pack(w, 11), None, r, gx, gy.numel() // cfg.cout, cfg.cin, cfg.cout, ns)
        return gx.view(cfg.xshape)
    if ns and cfg.subpixel_dgrad():
        d = cfg.subpixel_dgrad_desc()
        ws, nws = (None, 0) if fuse else _split_ws(cfg, 'dgrad2', d, gy.device)
        check(_lib.lib().vca_conv_fwd_bf16_ws(d, ns, ptr(gy), _pack(w, 12).data_ptr(), None, None, None, None, ptr(r),
                                              ptr(gx), None, ptr(ws), nws, stream()), 'conv_dgrad_bf16_subpixel')
    elif ns and cfg.cout > 1:
        ws, nws = (None, 0) if fuse else _split_ws(cfg, 'dgrad', d, gy.device)
        check(_lib.lib().vca_conv_fwd_bf16_ws(d, ns, ptr(gy), _pack(w, 11).data_ptr(), None, None, None, None, ptr(r),
                                              ptr(gx), None, ptr(ws), nws, stream()), 'conv_dgrad_bf16')
    else:
        check(_lib.lib().vca_conv_fwd_f32(d, ptr(gy), ptr(_pack(w, 1)), None, None, None, None, ptr(r), ptr(gx), None,
                                          stream()), 'conv_dgrad')
    if cfg.ups:
        gx = gx.view(cfg.n, cfg.hin, 2, cfg.win, 2, cfg.cin).sum((2, 4))
    gx = gx.view(cfg.xshape)
    if res is not None and not fuse:
        gx = gx + res
    return gx


_wgrad_ws = {}          # (device, stream, shape) -> [taps][Cin][Cout4] workspace, zero whenever no kernel is using it;
                        # shared by every layer of that shape issued on that stream (stream order serialises its users)
_direct_grad = [False]  # set by direct_param_grad()


_wgrad_own = {}         # (grad data_ptr, shape) -> that parameter's own zero-between-uses workspace
_wgrad_pending = {}     # (grad data_ptr, shape) -> (workspace, grad tensor, cout, cin, taps), filled during a backward
_unpack_tables = {}     # signature of a pending set -> (device table, ndesc, tiles)


class direct_param_grad:
    """While active, the weight gradient of a leaf parameter whose .grad is already allocated (the flat gradient
    buffers of flatten_module) is not returned to autograd (which would materialise it and add it with one more
    kernel per parameter): the weight-gradient kernels leave it in a workspace owned by that parameter, and on exit
    all workspaces are transposed and added into the gradients by one launch."""

    def __enter__(self):
        self.prev = _direct_grad[0]
        _direct_grad[0] = True

    def __exit__(self, *exc):
        _direct_grad[0] = self.prev
        if not self.prev:
            _flush_wgrads()


def _flush_wgrads():
    if not _wgrad_pending:
        return
    keys = sorted(_wgrad_pending)
    ents = [_wgrad_pending[k] for k in keys]
    _wgrad_pending.clear()
    L = _lib.lib()
    sig = tuple(keys)
    tab = _unpack_tables.get(sig)
    if tab is None and not torch.cuda.is_current_stream_capturing():
        import numpy as np
        rec = np.zeros((len(ents), 4), dtype=np.int64)     # 32-byte records: src, dst, (Cout,Cin), (taps,tile_begin)
        tiles = 0
        for i, (ws, g, cout, cin, taps) in enumerate(ents):
            rec[i, 0], rec[i, 1] = ws.data_ptr(), g.data_ptr()
            rec[i, 2] = cout | (cin << 32)
            rec[i, 3] = taps | (tiles << 32)
            tiles += ((cin * taps + 63) // 64) * ((((cout + 3) // 4 * 4) + 31) // 32)
        tab = _unpack_tables[sig] = (torch.from_numpy(rec).to(ents[0][0].device), len(ents), tiles)
    if tab is not None:
        check(L.vca_unpack_wgrad_batch_f32(tab[0].data_ptr(), tab[1], tab[2], stream()), 'unpack_batch')
        return
    for ws, g, cout, cin, taps in ents:      # a new set of gradients met while capturing: one launch each
        check(L.vca_unpack_wgrad_f32(cout, cin, taps, ptr(ws), ptr(g), 1.0, 1, stream()), 'unpack_wgrad')


_inputs_only = [False]  # set by input_grads_only()

# A module may compute one of its outputs on a side stream so that a latency-bound chain (the sentence encoder's GRU)
# runs beside its caller's next kernels.  Such an output carries the stream it is produced on until a consumer calls
# ensure_ready(), which makes the current stream wait for it.  Only callers that do so ask for deferred outputs.
_side_streams = {}


def side_stream(key):
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream()
    return s


def ensure_ready(t):
    """The current stream waits for the side stream tensor t was produced on, if any."""
    s = getattr(t, '_vca_stream', None)
    if s is not None:
        torch.cuda.current_stream().wait_stream(s)
    return t


def join_side_streams():
    """The current stream waits for every side stream (after a backward pass, whose last kernels on a side stream,
    gradients added straight into the flat buffers, nothing else waits for)."""
    cur = torch.cuda.current_stream()
    for s in _side_streams.values():
        cur.wait_stream(s)


class input_grads_only:
    """While active, convolution / linear backward passes skip the weight and bias gradients.  For
    torch.autograd.grad(outputs, inputs=<activations>) calls such as the R1 penalty (train.py:189-194): a Python
    autograd Function only knows which of its inputs require grad, not which gradients the caller asked for, so
    it would compute every parameter gradient and autograd would drop them."""

    def __enter__(self):
        self.prev = _inputs_only[0]
        _inputs_only[0] = True

    def __exit__(self, *exc):
        _inputs_only[0] = self.prev


def _grad_sink(w):
    if (_direct_grad[0] and not torch.is_grad_enabled() and w.is_leaf and w.requires_grad and w.grad is not None
            and w.grad.is_contiguous()):
        return w.grad
    return None


def _bias_grad(gy, bias_param, cout):
    """Gradient of a bias from the output gradient gy (..., cout): added into the parameter's flat gradient when the
    step asked for in-place accumulation (returns None), otherwise returned as a fresh tensor."""
    sink = _grad_sink(bias_param) if bias_param is not None else None
    gb = sink if sink is not None else torch.zeros((cout,), device=gy.device, dtype=torch.float32)
    _colsum_add(gy, cout, gb)
    return None if sink is not None else gb


def _colsum_add(x, c, dst):
    """dst[c] += column sums of the contiguous x viewed as (-1, c).  Used instead of torch's sum(0): its multi-block
    reduction zeroes a semaphore buffer with a memset, and memset nodes are not reliable in replayed HIP graphs."""
    x = x.contiguous()
    P = x.numel() // c
    part = torch.empty((_lib.lib().vca_colsum_groups(P, c), c), device=x.device, dtype=torch.float32)
    check(_lib.lib().vca_colsum_add_f32(P, c, ptr(x), ptr(part), ptr(dst), stream()), 'colsum_add')
    return dst


def colsum(x, c):
    """Column sums of x viewed as (-1, c), as a fresh (c,) tensor (no autograd)."""
    return _colsum_add(x, c, torch.zeros((c,), device=x.device, dtype=torch.float32))


_bias_done = [False]   # set by _conv_wgrad_raw: whether the launch also accumulated the bias gradient


def _conv_wgrad_raw(x, gy, cfg, sink=None, bias_sink=None):
    """Weight gradient in the parameter's layout; with `sink` (a contiguous tensor of that layout) it is added to
    sink and None is returned.  bias_sink: optional (Cout,) tensor that receives += the bias gradient (column sums
    of gy); returns False in its place when the kernel in use cannot produce it (the caller then sums itself)."""
    cout, cin, taps = cfg.cout, cfg.cin, cfg.kd * cfg.kh * cfg.kw
    shape = (taps, cin, (cout + 3) // 4 * 4)
    deferred = sink is not None and _direct_grad[0]
    if deferred:
        # the parameter's own workspace: contributions of several backward nodes accumulate in it and it is
        # unpacked once, together with all the others, when the backward pass is over
        key = (sink.data_ptr(), shape)
        dwp = _wgrad_own.get(key)
        if dwp is None:
            dwp = _wgrad_own[key] = torch.zeros(shape, device=x.device, dtype=torch.float32)
        _wgrad_pending[key] = (dwp, sink, cout, cin, taps)
    else:
        key = (x.device, stream(), shape)
        dwp = _wgrad_ws.get(key)
        if dwp is None:
            dwp = _wgrad_ws[key] = torch.zeros(shape, device=x.device, dtype=torch.float32)
    ns = _nsplit(True)
    if ns and cin > 1 and cfg.kd == 1 and taps <= 32:
        check(_lib.lib().vca_conv_wgrad_bf16(cfg.desc(), ns, ptr(x), ptr(gy), None, None, None, ptr(dwp),
                                             ptr(bias_sink), stream()), 'conv_wgrad_bf16')
        _bias_done[0] = bias_sink is not None
    else:
        _bias_done[0] = False
        nws = _lib.lib().vca_conv_wgrad_cin1_ws_floats(cfg.desc()) if cin == 1 else 0
        if nws:
            ws = torch.empty((nws,), device=x.device, dtype=torch.float32)
            check(_lib.lib().vca_conv_wgrad_cin1_f32(cfg.desc(), ptr(x), ptr(gy), ptr(ws), nws, ptr(dwp),
                                                     ptr(bias_sink), stream()), 'conv_wgrad_cin1')
            _bias_done[0] = bias_sink is not None
        else:
            check(_lib.lib().vca_conv_wgrad_f32(cfg.desc(), ptr(x), ptr(gy), None, None, None, ptr(dwp), stream()),
                  'conv_wgrad')
    if deferred:
        return None
    gw = sink if sink is not None else torch.empty(cfg.pshape, device=x.device, dtype=torch.float32)
    # the unpack kernel zeroes the workspace behind its reads
    check(_lib.lib().vca_unpack_wgrad_f32(cout, cin, taps, ptr(dwp), ptr(gw), 1.0 if sink is not None else 0.0, 1,
                                          stream()), 'unpack_wgrad')
    return None if sink is not None else gw


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class ConvStatsFn(torch.autograd.Function):
    """Convolution that also returns the per-channel (sum, sum of squares) of its bias-free output, produced in the
    kernel epilogue, so the train-mode batch-norm that follows needs no statistics pass over the tensor."""

    @staticmethod
    def forward(ctx, x, w, bias, cfg):
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.save_for_backward(x, w)
        stats = _zeros((CONV_STAT_SLOTS, 2, cfg.cout), x.device)
        y = _conv_fwd_raw(_c(x), w, bias, cfg, stats)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)    # no zero-filled gradient tensor for the statistics output
        return y, stats

    @staticmethod
    @once
    def backward(ctx, gy, _gs):
        x, w = ctx.saved_tensors
        cfg = ctx.cfg
        gy = _c(gy)
        gx = _conv_dgrad_raw(gy, w, cfg) if ctx.needs_input_grad[0] else None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        bsink = _grad_sink(ctx.bias_ref) if want_b else None
        _bias_done[0] = False
        gw = _conv_wgrad_raw(_c(x), gy, cfg, _grad_sink(w), bsink) if ctx.needs_input_grad[1] else None
        gb = None
        if want_b and not (bsink is not None and _bias_done[0]):
            gb = _bias_grad(gy, ctx.bias_ref, cfg.cout)
        return gx, gw, gb, None


class ConvStatsSkipFn(torch.autograd.Function):
    """ConvStatsFn (no bias) that also hands its input back as a third output.  A block whose input feeds both this
    convolution and a shortcut takes the shortcut from that output: autograd then delivers the shortcut's gradient
    to this node together with the convolution's, and the input-gradient kernel adds it in its epilogue instead of
    autograd adding two full-size tensors afterwards (resnet.py:53-66: `out += residual`)."""

    @staticmethod
    def forward(ctx, x, w, cfg):
        ctx.cfg = cfg
        ctx.save_for_backward(x, w)
        stats = _zeros((CONV_STAT_SLOTS, 2, cfg.cout), x.device)
        y = _conv_fwd_raw(_c(x), w, None, cfg, stats)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats, x.view_as(x)

    @staticmethod
    @once
    def backward(ctx, gy, _gs, gskip):
        x, w = ctx.saved_tensors
        cfg = ctx.cfg
        gy = _c(gy)
        gx = _conv_dgrad_raw(gy, w, cfg, gskip) if ctx.needs_input_grad[0] else None
        gw = _conv_wgrad_raw(_c(x), gy, cfg, _grad_sink(w)) if ctx.needs_input_grad[1] else None
        return gx, gw, None


def conv2d_skip(x, w, stride=1, pad=0):
    """Bias-free conv2d with statistics that also returns an alias of x for the block's shortcut branch:
    (y, BnStats, x_alias)."""
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    cfg = ConvCfg(w.shape, x.shape, stride, pad, 0)
    if cfg.cin > 1:
        y, sums, xa = ConvStatsSkipFn.apply(x, w, cfg)
        return y, BnStats(sums, None), xa
    y, st = conv2d(x, w, None, stride, pad, want_stats=True)
    return y, st, x


class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, cfg):
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.save_for_backward(x, w)
        return _conv_fwd_raw(_c(x), w, bias, cfg)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        cfg = ctx.cfg
        gy = _c(gy)
        _exact_bwd[0] = torch.is_grad_enabled()
        gx = ConvDgradFn.apply(gy, w, cfg) if ctx.needs_input_grad[0] else None
        gw = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not _inputs_only[0]
        bsink = _grad_sink(ctx.bias_ref) if want_b else None
        _bias_done[0] = False
        if ctx.needs_input_grad[1] and not _inputs_only[0]:
            sink = _grad_sink(w)
            gw = ConvWgradFn.apply(x, gy, cfg) if sink is None else _conv_wgrad_raw(_c(x), gy, cfg, sink, bsink)
        _exact_bwd[0] = False
        gb = None
        if want_b and not (bsink is not None and _bias_done[0]):
            gb = gy.reshape(-1, cfg.cout).sum(0) if torch.is_grad_enabled() else _bias_grad(gy, ctx.bias_ref, cfg.cout)
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
        _exact_bwd[0] = torch.is_grad_enabled() or R1_SECOND_ORDER_SPLIT[0]
        _grad_conv[0] = not _exact_bwd[0]
        ggy = ConvFn.apply(ggx, w, None, cfg) if ctx.needs_input_grad[0] else None
        _grad_conv[0] = False
        gw = None
        if ctx.needs_input_grad[1]:
            sink = _grad_sink(w)
            gw = ConvWgradFn.apply(ggx, gy, cfg) if sink is None else _conv_wgrad_raw(ggx, _c(gy), cfg, sink)
        _exact_bwd[0] = False
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
        _exact_bwd[0] = torch.is_grad_enabled() or R1_SECOND_ORDER_SPLIT[0]
        gx = ConvDgradFn.apply(gy, ggw, cfg) if ctx.needs_input_grad[0] else None
        _grad_conv[0] = not _exact_bwd[0]
        ggy = ConvFn.apply(x, ggw, None, cfg) if ctx.needs_input_grad[1] else None
        _grad_conv[0] = False
        _exact_bwd[0] = False
        return gx, ggy, None


class ParamViewFn(torch.autograd.Function):
    """A re-laid-out part of a parameter as a fresh contiguous tensor: y = view_fn(w).reshape(shape), view_fn being
    pure view operations (slicing, permute).  Used where a layer works on channel groups of a weight separately.
    Plain autograd would rebuild a full-size gradient per use (a zero fill, a strided copy and an add, each the
    size of the whole weight); here, while direct_param_grad() is active, the incoming gradient is added straight
    into the matching view of the parameter's flat gradient."""

    @staticmethod
    def forward(ctx, w, view_fn, shape):
        ctx.view_fn, ctx.w = view_fn, w
        v = view_fn(w.detach())
        ctx.vshape = v.shape
        return v.reshape(shape).contiguous()

    @staticmethod
    def backward(ctx, g):
        w = ctx.w
        sink = _grad_sink(w)
        if sink is not None:
            ctx.view_fn(sink).add_(g.view(ctx.vshape))
            return None, None, None
        full = torch.zeros_like(w)
        ctx.view_fn(full).copy_(g.reshape(ctx.vshape))
        return full, None, None


def param_view(w, view_fn, shape=None):
    """view_fn(w) as a contiguous tensor of `shape` (default: the view's own shape); see ParamViewFn."""
    if shape is None:
        shape = tuple(view_fn(w.detach()).shape)
    return ParamViewFn.apply(w, view_fn, tuple(shape))


class BnStats:
    """Statistics of a convolution output handed to the batch-norm that follows: sums of (y - bias)."""

    def __init__(self, sums, bias):
        self.sums, self.bias = sums, bias


def conv2d(x, w, bias=None, stride=1, pad=0, ups=0, want_stats=False):
    """x: (N,H,W,Cin) channels-last; w: (Cout,Cin,KH,KW).  Returns (N,OH,OW,Cout); with want_stats also a BnStats
    for ops.bn_act(..., stats=)."""
    pad = (pad, pad) if isinstance(pad, int) else tuple(pad)
    cfg = ConvCfg(w.shape, x.shape, stride, pad, ups)
    if want_stats and cfg.cin > 1:
        y, sums = ConvStatsFn.apply(x, w, bias, cfg)
        return y, BnStats(sums, bias)
    y = ConvFn.apply(x, w, bias, cfg)
    return (y, None) if want_stats else y


def conv1d(x, w, bias=None, pad=0, want_stats=False):
    """x: (N,L,Cin); w: (Cout,Cin,K).  Returns (N,L',Cout)."""
    n, l, c = x.shape
    cfg = ConvCfg((w.shape[0], w.shape[1], 1, w.shape[2]), (n, 1, l, c), 1, (0, pad), 0, pshape=w.shape)
    if want_stats:
        y, sums = ConvStatsFn.apply(x.reshape(n, 1, l, c), w, bias, cfg)
        return y.view(n, cfg.ow, w.shape[0]), BnStats(sums, bias)
    y = ConvFn.apply(x.reshape(n, 1, l, c), w, bias, cfg)
    return y.view(n, cfg.ow, w.shape[0])


def linear(x, w, bias=None):
    """x: (..., Cin); w: (Cout, Cin)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, 1, 1, x.shape[-1])
    cfg = ConvCfg((w.shape[0], w.shape[1], 1, 1), x2.shape, 1, (0, 0), 0, pshape=w.shape)
    return ConvFn.apply(x2, w, bias, cfg).view(*lead, w.shape[0])


class StemConvFn(torch.autograd.Function):
    """The 5x7x7 stride-(1,2,2) stem on the dedicated split-bf16 kernel; also returns the per-channel
    (sum, sum of squares) of its output for the batch-norm that follows."""

    @staticmethod
    def forward(ctx, x, w, cfg):
        x = _c(x)
        B, T, H, W, _ = x.shape
        L = _lib.lib()
        key = 20
        wp = None
        cacheable = isinstance(w, torch.nn.Parameter)
        if cacheable:
            ent = _pack_cache.get(id(w))
            if ent is not None and ent[0]() is w and ent[1].get(key, (None,))[0] == _wkey(w):
                wp = ent[1][key][1]
        if wp is None:
            old = ent[1][key][1] if (cacheable and ent is not None and ent[0]() is w and key in ent[1]) else None
            wp = old if old is not None else torch.empty(L.vca_stem_pack_bytes(), device=x.device, dtype=torch.uint8)
            check(L.vca_stem_pack_weight(ptr(_c(w.detach())), wp.data_ptr(), stream()), 'stem_pack')
            if cacheable:
                k2 = id(w)
                ent = _pack_cache.get(k2)
                if ent is None or ent[0]() is not w:
                    ent = (weakref.ref(w, lambda _r, k2=k2: _pack_cache.pop(k2, None)), {})
                    _pack_cache[k2] = ent
                ent[1][key] = (_wkey(w), wp)
        y = torch.empty(cfg.yshape, device=x.device, dtype=torch.float32)
        stats = _zeros((CONV_STAT_SLOTS, 2, 64), x.device)     # partial buffers, totalled by the batch-norm finalize
        check(L.vca_stem_fwd_bf16(_nsplit(), B, T, H, W, ptr(x), wp.data_ptr(), ptr(y), ptr(stats), CONV_STAT_SLOTS,
                                  stream()), 'stem_fwd')
        ctx.save_for_backward(x, w)
        ctx.cfg = cfg
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    @once
    def backward(ctx, gy, _gstats):
        x, w = ctx.saved_tensors
        if not ctx.needs_input_grad[1]:
            return None, None, None
        B, T, H, W, _ = x.shape
        # weight gradient on the matrix cores: per output position the 8x8 patch of the clip as 64 channels, then a
        # 5-tap convolution along t (one 64x64 tile of 5 taps instead of 245 single-channel taps)
        oh, ow = ctx.cfg.oh, ctx.cfg.ow
        xp = torch.empty((B, T, oh * ow, 64), device=x.device, dtype=torch.float32)
        check(_lib.lib().vca_stem_patches_f32(B, T, H, W, ptr(x), ptr(xp), stream()), 'stem_patches')
        cfg = ConvCfg((64, 64, 5, 1), xp.shape, 1, (2, 0), 0)
        g5 = _conv_wgrad_raw(xp, _c(gy).view(B, T, oh * ow, 64), cfg)                  # (co, kh*8+kw, kt, 1)
        gw = g5.view(64, 8, 8, 5).permute(0, 3, 1, 2)[:, :, :7, :7].reshape(w.shape)
        sink = _grad_sink(w)
        if sink is not None:
            sink.add_(gw)
            return None, None, None
        return None, gw.contiguous(), None


def conv3d_stem(x, w, stride=2, pad=(2, 3, 3)):
    """x: (N,D,H,W,1); w: (64,1,5,7,7), spatial stride 2, temporal stride 1 (visual_front.py:11).
    Returns (y, stats): stats is None or the per-channel (sum, sum of squares) of y."""
    cfg = ConvCfg(w.shape, x.shape, stride, (pad[1], pad[2]), 0, pad_d=pad[0])
    if _nsplit() and tuple(w.shape) == (64, 1, 5, 7, 7) and stride == 2 and tuple(pad) == (2, 3, 3):
        return StemConvFn.apply(x, w, cfg)
    return ConvFn.apply(x, w, None, cfg), None


# --------------------------------------------------------------------------- #
# per-channel kernels: batch-norm + activation, activations, residual scale-add
# --------------------------------------------------------------------------- #
LEAKY = 'leaky0.2'      # the generator/discriminator activation (generator.py:9,52,95: nn.LeakyReLU(0.2))


def _act_code(act):
    """-> (code, slope, alpha tensor or None)"""
    if act is None:
        return ACT_NONE, 0.0, None
    if act == LEAKY:
        return ACT_LEAKY, 0.2, None
    if act == 'tanh':
        return ACT_TANH, 0.0, None
    if isinstance(act, torch.nn.PReLU):
        return ACT_PRELU, 0.0, act.weight
    if isinstance(act, torch.nn.ReLU):
        return ACT_RELU, 0.0, None
    if isinstance(act, torch.nn.LeakyReLU):
        return ACT_LEAKY, float(act.negative_slope), None
    raise ValueError(act)


def _pc(x):
    c = x.shape[-1]
    return x.numel() // c, c


def _bn_param_sinks(ctx_params, coef, act, want_alpha):
    """Gradient storage of (beta, gamma, alpha) when the step accumulates in place: the backward kernels then add the
    per-channel totals there themselves and autograd gets None (no accumulate launch per parameter).  All or nothing."""
    gamma, beta, alpha = ctx_params
    need = []
    if coef is not None:
        need += [beta, gamma]
    if act == ACT_PRELU and want_alpha:
        need.append(alpha)
    if not need or any(p is None for p in need):
        return None
    sinks = [_grad_sink(p) for p in need]
    if any(t is None for t in sinks):
        return None
    sb, sg = (sinks[0], sinks[1]) if coef is not None else (None, None)
    sa = sinks[-1] if (act == ACT_PRELU and want_alpha) else None
    return sb, sg, sa


class BnActFn(torch.autograd.Function):
    """y = act(x*scale + shift + res) * oscale with (scale, shift) from batch-norm coefficients `coef`
    (rows: mean, invstd, scale, shift) or absent; gradients for x, gamma, beta, alpha, res.
    With skip=True the input is handed back as a second (alias) output: a block whose input also feeds an identity
    shortcut (generator.py:128-130) takes the shortcut from it, so autograd delivers the shortcut's gradient to this
    node and the backward kernel adds it while writing gx, instead of a full-size add afterwards."""

    @staticmethod
    def forward(ctx, x, gamma, beta, alpha, res, coef, act, slope, oscale, train, skip=False):
        x = _c(x)
        res = _c(res) if res is not None else None
        P, C = _pc(x)
        y = torch.empty_like(x)
        sc, sh = (coef[2], coef[3]) if coef is not None else (None, None)
        check(_lib.lib().vca_affine_act_f32(P, C, ptr(x), ptr(sc), ptr(sh), ptr(alpha), ptr(res), act, slope, oscale,
                                            ptr(y), stream()), 'affine_act')
        ctx.save_for_backward(x, alpha, res, coef)
        ctx.cfg = (act, slope, oscale, train, gamma is not None, beta is not None)
        ctx.params = (gamma, beta, alpha)
        ctx.set_materialize_grads(False)
        if skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    @once
    def backward(ctx, gy, gskip=None):
        x, alpha, res, coef = ctx.saved_tensors
        act, slope, oscale, train, has_g, has_b = ctx.cfg
        if gy is None:                       # only the alias output was used
            return gskip, None, None, None, None, None, None, None, None, None, None
        P, C = _pc(x)
        gy = _c(gy)
        need_sums = coef is not None or act == ACT_PRELU
        sums = _zeros((16, 3, C), x.device) if need_sums else None   # VCA_BN_SLOTS
        want_gx = ctx.needs_input_grad[0]
        gx = torch.empty_like(x) if want_gx else None
        gres = torch.empty_like(x) if (res is not None and ctx.needs_input_grad[4]) else None
        mean, invstd, sc, sh = (coef[0], coef[1], coef[2], coef[3]) if coef is not None else (None,) * 4
        sinks = _bn_param_sinks(ctx.params, coef, act, ctx.needs_input_grad[3]) if (need_sums and has_g == has_b) else None
        sb, sg, sa = sinks if sinks is not None else (None, None, None)
        gadd = _c(gskip) if (gskip is not None and want_gx) else None
        check(_lib.lib().vca_affine_act_bwd2_f32(P, C, ptr(gy), ptr(x), ptr(sc), ptr(sh), ptr(alpha), ptr(res),
                                                 ptr(mean), ptr(invstd), act, slope, oscale, 1 if train else 0,
                                                 ptr(sums), ptr(gx), ptr(gres), ptr(gadd), ptr(sb), ptr(sg), ptr(sa),
                                                 stream()), 'affine_act_bwd')
        if sinks is not None:
            ggamma = gbeta = galpha = None
        else:
            ggamma = sums[0, 1] if (coef is not None and has_g) else None
            gbeta = sums[0, 0] if (coef is not None and has_b) else None
            galpha = sums[0, 2] if (act == ACT_PRELU and ctx.needs_input_grad[3]) else None
        return gx, ggamma, gbeta, galpha, gres, None, None, None, None, None, None


def _bn_coef(x, bn, stats=None):
    """Batch-norm coefficients (mean, invstd, scale, shift) for a channels-last tensor; updates running stats.
    `stats`: optional precomputed per-channel (sum, sum of squares) of x (from a convolution epilogue)."""
    P, C = _pc(x)
    coef = torch.empty((4, C), device=x.device, dtype=torch.float32)
    L = _lib.lib()
    if bn.training or bn.running_mean is None:
        pivot = None
        if stats is None:
            sums = _zeros((16, 2, C), x.device)   # VCA_BN_SLOTS partial buffers
            pivot = _c(x.detach())
            check(L.vca_colstats_f32(P, C, ptr(pivot), ptr(sums), stream()), 'colstats')
        elif isinstance(stats, BnStats):
            sums = stats.sums
            pivot = stats.bias.detach() if stats.bias is not None else None
        else:
            sums = stats
        mom = bn.momentum if bn.momentum is not None else 0.1
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        nslots = sums.shape[0] if (stats is not None and sums.dim() == 3) else 1   # convolution epilogue: partial buffers
        check(L.vca_bn_finalize_f32(C, float(P), ptr(sums), nslots, ptr(pivot), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                    bn.eps, mom, ptr(bn.running_mean) if track else None,
                                    ptr(bn.running_var) if track else None,
                                    ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), stream()), 'bn_finalize')
        if track:
            _bump_nbt(bn)
        return coef, True
    check(L.vca_bn_eval_coeff_f32(C, ptr(bn.running_mean), ptr(bn.running_var), ptr(bn.weight.detach()),
                                  ptr(bn.bias.detach()), bn.eps, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]),
                                  ptr(coef[3]), stream()), 'bn_eval_coeff')
    return coef, False


def bn_act_parts(parts, bn, act=None):
    """Batch-norm + activation of a tensor that is only given by its channel groups: parts = [(x_i, rep_i), ...],
    x_i channels-last holding the channels [c_i, c_i + C_i) of the normalised tensor, each row of x_i standing for
    rep_i identical rows of it (rows * rep is the same for all parts).  Returns the activated parts in their own
    shapes.  Statistics, running statistics and gradients are those of batch-norm over the assembled tensor."""
    code, slope, alpha = _act_code(act)
    assert alpha is None
    dev = parts[0][0].device
    C = sum(x.shape[-1] for x, _ in parts)
    n = _pc(parts[0][0])[0] * parts[0][1]
    L = _lib.lib()
    coef = torch.empty((4, C), device=dev, dtype=torch.float32)
    g, b = bn.weight, bn.bias
    if bn.training or bn.running_mean is None:
        sums, pivots = [], []
        for x, rep in parts:
            P, Ci = _pc(x)
            assert P * rep == n, (P, rep, n)
            xs = _c(x.detach())
            sm = _zeros((16, 2, Ci), dev)   # VCA_BN_SLOTS partial buffers
            check(L.vca_colstats_f32(P, Ci, ptr(xs), ptr(sm), stream()), 'colstats')
            sums.append(sm[0] * float(rep) if rep != 1 else sm[0])
            pivots.append(xs.reshape(P, Ci)[0])
        sums, pivot = torch.cat(sums, 1).contiguous(), torch.cat(pivots).contiguous()
        mom = bn.momentum if bn.momentum is not None else 0.1
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        check(L.vca_bn_finalize_f32(C, float(n), ptr(sums), 1, ptr(pivot), ptr(g.detach()), ptr(b.detach()), bn.eps, mom,
                                    ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                    ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), stream()), 'bn_finalize')
        if track:
            _bump_nbt(bn)
        train = True
    else:
        check(L.vca_bn_eval_coeff_f32(C, ptr(bn.running_mean), ptr(bn.running_var), ptr(g.detach()), ptr(b.detach()),
                                      bn.eps, ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]), stream()),
              'bn_eval_coeff')
        train = False
    outs, c0 = [], 0
    for x, rep in parts:
        Ci = x.shape[-1]
        # A repeated row receives the sum of its copies' gradients, and the batch-norm correction terms summed
        # over the rep copies are rep/n = 1/rows of x: the ordinary backward over the rows of x is exact.
        outs.append(BnActFn.apply(x, g[c0:c0 + Ci], b[c0:c0 + Ci], None, None, coef[:, c0:c0 + Ci], code, slope, 1.0,
                                  train))
        c0 += Ci
    return outs


class HBandExpandFn(torch.autograd.Function):
    """y[b,h,w,c] = base[b,h,w,c] + sum over kernel rows kh with 0 <= h+kh-pad < H of p[b,w,kh*C+c]: assembles the
    KxK convolution of a map that repeats one row over H from the K row-wise 1xK convolutions of that row."""

    @staticmethod
    def forward(ctx, p, base, H, K, pad):
        p = _c(p)
        B, W, KC = p.shape
        C = KC // K
        base = _c(base) if base is not None else None
        y = torch.empty((B, H, W, C), device=p.device, dtype=torch.float32)
        check(_lib.lib().vca_hband_expand_f32(B, H, W, C, K, pad, ptr(p), ptr(base), ptr(y), stream()), 'hband_expand')
        ctx.cfg = (B, H, W, C, K, pad)
        return y

    @staticmethod
    @once
    def backward(ctx, gy):
        B, H, W, C, K, pad = ctx.cfg
        gy = _c(gy)
        gp = None
        if ctx.needs_input_grad[0]:
            gp = torch.empty((B, W, K * C), device=gy.device, dtype=torch.float32)
            check(_lib.lib().vca_hband_reduce_f32(B, H, W, C, K, pad, ptr(gy), ptr(gp), stream()), 'hband_reduce')
        return gp, (gy if ctx.needs_input_grad[1] else None), None, None, None


def hband_expand(p, H, K=1, pad=0, base=None):
    return HBandExpandFn.apply(p, base, H, K, pad)


def bn_act(x, bn, act=None, residual=None, stats=None, skip=False):
    """act(batch_norm(x) + residual) over the last (channel) axis of a channels-last tensor.  skip=True: returns
    (y, alias of x) for a block whose input also feeds an identity shortcut (see BnActFn)."""
    code, slope, alpha = _act_code(act)
    coef, train = _bn_coef(x, bn, stats)
    return BnActFn.apply(x, bn.weight, bn.bias, alpha, residual, coef, code, slope, 1.0, train, skip)


class ActFn(torch.autograd.Function):
    """Parameter-free activation with a differentiable backward (the discriminators' R1 path needs it)."""

    @staticmethod
    def forward(ctx, x, act, slope):
        x = _c(x)
        P, C = _pc(x)
        y = torch.empty_like(x)
        check(_lib.lib().vca_affine_act_f32(P, C, ptr(x), None, None, None, None, act, slope, 1.0, ptr(y), stream()),
              'act')
        ctx.save_for_backward(x)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        return ActBwdFn.apply(gy, x, *ctx.cfg), None, None


class ActBwdFn(torch.autograd.Function):
    """gx = gy * act'(x): linear in gy, so its own backward is the same operator."""

    @staticmethod
    def forward(ctx, gy, x, act, slope):
        gy = _c(gy)
        P, C = _pc(x)
        gx = torch.empty_like(x)
        check(_lib.lib().vca_affine_act_bwd_f32(P, C, ptr(gy), ptr(x), None, None, None, None, None, None, act, slope,
                                                1.0, 0, None, ptr(gx), None, stream()), 'act_bwd')
        ctx.save_for_backward(x)
        ctx.cfg = (act, slope)
        return gx

    @staticmethod
    def backward(ctx, ggx):
        x, = ctx.saved_tensors
        return ActBwdFn.apply(ggx, x, *ctx.cfg), None, None, None


def _pad4(x):
    """Views whose channel count is not a multiple of 4 (single-channel mel maps) are processed as flat rows."""
    c = x.shape[-1]
    if c % 4 == 0:
        return x, None
    n = x.numel()
    assert n % 4 == 0, x.shape
    return x.reshape(-1, 4), x.shape


def leaky(x):
    return ActFn.apply(x, ACT_LEAKY, 0.2)


def tanh(x):
    x2, shp = _pad4(x)
    y = BnActFn.apply(x2, None, None, None, None, None, ACT_TANH, 0.0, 1.0, False)
    return y if shp is None else y.view(shp)


class ScaleFn(torch.autograd.Function):
    """y = x * scale (linear, so its backward is itself)."""

    @staticmethod
    def forward(ctx, x, scale):
        x2, shp = _pad4(_c(x))
        P, C = _pc(x2)
        y = torch.empty_like(x2)
        check(_lib.lib().vca_affine_act_f32(P, C, ptr(x2), None, None, None, None, ACT_NONE, 0.0, scale, ptr(y),
                                            stream()), 'scale')
        ctx.scale = scale
        return y if shp is None else y.view(shp)

    @staticmethod
    def backward(ctx, g):
        return ScaleFn.apply(g, ctx.scale), None


class AddScaleFn(torch.autograd.Function):
    """y = (a + b) * scale with a differentiable backward (the discriminator blocks sit on the R1 path)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = _c(a), _c(b)
        P, C = _pc(a)
        y = torch.empty_like(a)
        check(_lib.lib().vca_affine_act_f32(P, C, ptr(a), None, None, None, ptr(b), ACT_NONE, 0.0, scale, ptr(y),
                                            stream()), 'add_scale')
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, g):
        gs = ScaleFn.apply(g, ctx.scale)
        return gs, gs, None


def add_scale(a, b, scale):
    """(a + b) * scale  (the residual merge of generator.py:49,92,130)."""
    return AddScaleFn.apply(a, b, scale)


# --------------------------------------------------------------------------- #
# pooling
# --------------------------------------------------------------------------- #
def _pool2_raw(mode, x, small_hw, large_hw):
    n, c = x.shape[0], x.shape[-1]
    (hs, ws), (hl, wl) = small_hw, large_hw
    out_hw = small_hw if mode in (0, 2) else large_hw
    y = torch.empty((n,) + tuple(out_hw) + (c,), device=x.device, dtype=torch.float32)
    check(_lib.lib().vca_pool2_f32(mode, n, hs, ws, hl, wl, c, ptr(_c(x)), ptr(y), stream()), 'pool2')
    return y


class Pool2Fn(torch.autograd.Function):
    """modes 0/1 (average and its adjoint) and 2/3 (block sum and nearest upsample) are adjoint pairs."""

    @staticmethod
    def forward(ctx, x, mode, small_hw, large_hw):
        ctx.cfg = (mode, small_hw, large_hw)
        return _pool2_raw(mode, x, small_hw, large_hw)

    @staticmethod
    def backward(ctx, gy):
        mode, small_hw, large_hw = ctx.cfg
        return Pool2Fn.apply(gy, mode ^ 1, small_hw, large_hw), None, None, None


def avgpool2(x):
    """F.avg_pool2d(x, 2) on a channels-last map (floor on odd sizes)."""
    n, h, w, c = x.shape
    return Pool2Fn.apply(x, 0, (h // 2, w // 2), (h, w))


class LeakyPool2Fn(torch.autograd.Function):
    """(LeakyReLU(x), avg_pool2d(x, 2)) of one map in one pass: the entry of a down-sampling discriminator block
    (generator.py:71-83).  Its backward is one kernel too (no separate activation backward, pooling adjoint and the
    add that joins the two branches), and differentiable (the blocks sit on the R1 path)."""

    @staticmethod
    def forward(ctx, x, slope):
        x = _c(x)
        n, h, w, c = x.shape
        ya = torch.empty_like(x)
        yp = torch.empty((n, h // 2, w // 2, c), device=x.device, dtype=torch.float32)
        check(_lib.lib().vca_leaky_pool2_fwd_f32(n, h, w, c, slope, ptr(x), ptr(ya), ptr(yp), stream()), 'leaky_pool2')
        ctx.save_for_backward(x)
        ctx.slope = slope
        ctx.set_materialize_grads(False)
        return ya, yp

    @staticmethod
    def backward(ctx, ga, gp):
        x, = ctx.saved_tensors
        if ga is None and gp is None:
            return None, None
        return LeakyPool2BwdFn.apply(ga, gp, x, ctx.slope), None


class LeakyPool2BwdFn(torch.autograd.Function):
    """gx = ga * LeakyReLU'(x) + avg_pool2d-adjoint(gp): linear in (ga, gp), so its own backward is the forward pair
    applied to the incoming gradient (with the activation's mask taken from x)."""

    @staticmethod
    def forward(ctx, ga, gp, x, slope):
        n, h, w, c = x.shape
        ga = None if ga is None else _c(ga)
        gp = None if gp is None else _c(gp)
        gx = torch.empty_like(x)
        check(_lib.lib().vca_leaky_pool2_bwd_f32(n, h, w, c, slope, ptr(ga), ptr(gp), ptr(x), ptr(gx), stream()),
              'leaky_pool2_bwd')
        ctx.save_for_backward(x)
        ctx.cfg = (slope, ga is not None, gp is not None)
        return gx

    @staticmethod
    def backward(ctx, ggx):
        x, = ctx.saved_tensors
        slope, has_a, has_p = ctx.cfg
        return (ActBwdFn.apply(ggx, x, ACT_LEAKY, slope) if has_a else None,
                avgpool2(ggx) if has_p else None, None, None)


def leaky_avgpool2(x):
    """(leaky(x), avgpool2(x)) in one pass."""
    if x.shape[-1] % 4 or x.shape[1] < 2 or x.shape[2] < 2:
        return leaky(x), avgpool2(x)
    return LeakyPool2Fn.apply(x, 0.2)


class MaxPool3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        n, h, w, c = x.shape
        oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
        idx = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.uint8)
        check(_lib.lib().vca_maxpool3s2_fwd_f32(n, h, w, c, ptr(x), ptr(y), idx.data_ptr(), stream()), 'maxpool_fwd')
        ctx.save_for_backward(idx)
        ctx.shape = (n, h, w, c)
        return y

    @staticmethod
    @once
    def backward(ctx, gy):
        idx, = ctx.saved_tensors
        n, h, w, c = ctx.shape
        gx = torch.empty(ctx.shape, device=gy.device, dtype=torch.float32)
        check(_lib.lib().vca_maxpool3s2_bwd_f32(n, h, w, c, ptr(_c(gy)), idx.data_ptr(), ptr(gx), stream()),
              'maxpool_bwd')
        return gx


class BnActPool3Fn(torch.autograd.Function):
    """maxpool3x3s2(act(batch_norm(x))) in one forward pass (visual_front.py:12-14); the activated full-resolution
    map is neither written nor kept.  Backward: the pooled gradient is scattered through the saved argmax taps, then
    the ordinary batch-norm backward runs on x.  (Gathering the pooled gradient inside the two batch-norm backward
    passes instead was measured slower: the gather costs more than the pass it saves.)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, alpha, coef, act, slope, train):
        x = _c(x)
        n, h, w, c = x.shape
        oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
        idx = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.uint8)
        check(_lib.lib().vca_affine_act_pool3_fwd_f32(n, h, w, c, ptr(x), ptr(coef[2]), ptr(coef[3]), ptr(alpha), act,
                                                      slope, ptr(y), idx.data_ptr(), stream()), 'affine_act_pool3')
        ctx.save_for_backward(x, alpha, coef, idx)
        ctx.cfg = (act, slope, train)
        ctx.params = (gamma, beta, alpha)
        return y

    @staticmethod
    @once
    def backward(ctx, gy):
        x, alpha, coef, idx = ctx.saved_tensors
        act, slope, train = ctx.cfg
        n, h, w, c = x.shape
        L = _lib.lib()
        g = torch.empty_like(x)
        check(L.vca_maxpool3s2_bwd_f32(n, h, w, c, ptr(_c(gy)), idx.data_ptr(), ptr(g), stream()), 'maxpool_bwd')
        sums = _zeros((16, 3, c), x.device)   # VCA_BN_SLOTS
        gx = torch.empty_like(x)
        sinks = _bn_param_sinks(ctx.params, coef, act, ctx.needs_input_grad[3])
        sb, sg, sa = sinks if sinks is not None else (None, None, None)
        check(L.vca_affine_act_bwd2_f32(n * h * w, c, ptr(g), ptr(x), ptr(coef[2]), ptr(coef[3]), ptr(alpha), None,
                                        ptr(coef[0]), ptr(coef[1]), act, slope, 1.0, 1 if train else 0, ptr(sums),
                                        ptr(gx), None, None, ptr(sb), ptr(sg), ptr(sa), stream()), 'affine_act_bwd')
        gx = gx if ctx.needs_input_grad[0] else None
        if sinks is not None:
            return gx, None, None, None, None, None, None, None
        galpha = sums[0, 2] if (act == ACT_PRELU and ctx.needs_input_grad[3]) else None
        return gx, sums[0, 1], sums[0, 0], galpha, None, None, None, None


def bn_act_maxpool3(x, bn, act=None, stats=None):
    """maxpool3x3s2(act(batch_norm(x))) on a channels-last map (N,H,W,C) as one kernel."""
    code, slope, alpha = _act_code(act)
    coef, train = _bn_coef(x, bn, stats)
    return BnActPool3Fn.apply(x, bn.weight, bn.bias, alpha, coef, code, slope, train)


def maxpool3x3s2(x):
    """MaxPool (3x3, stride 2, pad 1) on a channels-last map (visual_front.py:14)."""
    return MaxPool3Fn.apply(x)


class WindowMeanFn(torch.autograd.Function):
    """(N,H,W,C) -> (N,C): mean over the window [0:wh, 0:ww]."""

    @staticmethod
    def forward(ctx, x, wh, ww):
        x = _c(x)
        n, h, w, c = x.shape
        ctx.geom = (n, h, w, c, wh, ww)
        y = torch.empty((n, c), device=x.device, dtype=torch.float32)
        check(_lib.lib().vca_window_mean_f32(n, h, w, c, wh, ww, ptr(x), ptr(y), stream()), 'window_mean')
        return y

    @staticmethod
    def backward(ctx, gy):
        return WindowMeanBwdFn.apply(gy, ctx.geom), None, None


class WindowMeanBwdFn(torch.autograd.Function):
    """The adjoint of WindowMeanFn (linear, so its own backward is WindowMeanFn: R1 differentiates through the
    spatial mean of the unconditional head twice)."""

    @staticmethod
    def forward(ctx, gy, geom):
        n, h, w, c, wh, ww = geom
        ctx.geom = geom
        gx = torch.empty((n, h, w, c), device=gy.device, dtype=torch.float32)
        check(_lib.lib().vca_window_mean_bwd_f32(n, h, w, c, wh, ww, ptr(_c(gy)), ptr(gx), stream()), 'window_mean_bwd')
        return gx

    @staticmethod
    def backward(ctx, ggx):
        return WindowMeanFn.apply(ggx, ctx.geom[4], ctx.geom[5]), None


def spatial_mean(x):
    """(N,H,W,C) -> (N,C), the .mean([2,3]) of generator.py:310,316."""
    return WindowMeanFn.apply(x, x.shape[1], x.shape[2])


def head_pool4(x):
    """nn.AvgPool2d(4) of resnet.py:82 on a channels-last map whose output is 1x1: the mean of the top-left 4x4
    window when the map is at least 4 wide, the global mean otherwise (96x96 / 48x48 clips)."""
    n, h, w, c = x.shape
    if h >= 4 and w >= 4:
        assert h < 8 and w < 8, 'AvgPool2d(4) output must be 1x1 (visual_front.py:30)'
        return WindowMeanFn.apply(x, 4, 4)
    return WindowMeanFn.apply(x, h, w)


def _row_strides(t, n, h, w):
    """Row strides (in rows of C floats) of a source given as (N,H,W,C), (N,W,C) [broadcast over H] or (N,C)
    [broadcast over H and W]."""
    if t.dim() == 4:
        assert t.shape[:3] == (n, h, w), (t.shape, n, h, w)
        return h * w, w, 1
    if t.dim() == 3:
        assert t.shape[:2] == (n, w), (t.shape, n, w)
        return w, 0, 1
    assert t.dim() == 2 and t.shape[0] == n
    return 1, 0, 0


class Concat2Fn(torch.autograd.Function):
    """Channel concatenation of two channels-last sources, either of which may be broadcast over H (given as
    (N,W,C)) or over H and W (given as (N,C))."""

    @staticmethod
    def forward(ctx, x1, x2, n, h, w):
        x1, x2 = _c(x1), _c(x2)
        c1, c2 = x1.shape[-1], x2.shape[-1]
        s1, s2 = _row_strides(x1, n, h, w), _row_strides(x2, n, h, w)
        y = torch.empty((n, h, w, c1 + c2), device=x1.device, dtype=torch.float32)
        check(_lib.lib().vca_concat2_f32(n, h, w, c1, c2, ptr(x1), *s1, ptr(x2), *s2, ptr(y), stream()), 'concat2')
        ctx.meta = (n, h, w, c1, c2, s1, s2, x1.shape, x2.shape)
        return y

    @staticmethod
    @once
    def backward(ctx, gy):
        n, h, w, c1, c2, s1, s2, sh1, sh2 = ctx.meta
        gy = _c(gy)
        outs = []
        for k, (c0, cs, st, shp) in enumerate(((0, c1, s1, sh1), (c1, c2, s2, sh2))):
            if not ctx.needs_input_grad[k]:
                outs.append(None)
                continue
            bcast = len(shp) < 4
            gx = (torch.zeros if bcast else torch.empty)(shp, device=gy.device, dtype=torch.float32)
            check(_lib.lib().vca_concat2_bwd_f32(n, h, w, c1 + c2, c0, cs, ptr(gy), *st, 1 if bcast else 0, ptr(gx),
                                                 stream()), 'concat2_bwd')
            outs.append(gx)
        return outs[0], outs[1], None, None, None


def concat2(x1, x2, n, h, w):
    return Concat2Fn.apply(x1, x2, n, h, w)


_rng_step = {}      # device -> int32 word mixed into every dropout stream; advanced once per training step
_rng_calls = [0]    # dropout call sites seen since the last advance (distinct seeds within a step)
_rng_seed = [0x5EED]


def set_seed(seed, rank=0):
    """Seed of every dropout stream of this process.  Data-parallel ranks must pass their rank (build_models /
    GanStep do): with equal seeds every replica would draw the same masks on its own shard."""
    _rng_seed[0] = ((int(seed) & 0xFFFFFFFF) * 0x9E3779B1 + (int(rank) + 1) * 0xC2B2AE3D) & 0xFFFFFFFFFFFFFFFF


def rng_advance(device):
    """Move every dropout stream on: call once per training step (a device-side add, so it is captured with the
    step and replays draw fresh masks)."""
    t = _rng_step.get(device)
    if t is None:
        t = _rng_step[device] = torch.zeros(1, device=device, dtype=torch.int32)
    t.add_(1)
    _rng_calls[0] = 0


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        x = _c(x)
        y = torch.empty_like(x)
        mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        step = _rng_step.get(x.device)
        if step is None:
            step = _rng_step[x.device] = torch.zeros(1, device=x.device, dtype=torch.int32)
        _rng_calls[0] += 1
        seed = (_rng_seed[0] * 0x9E3779B1 + _rng_calls[0] * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF   # set_seed()
        check(_lib.lib().vca_dropout_f32(x.numel(), ptr(x), float(p), seed, step.data_ptr(), ptr(y), mask.data_ptr(),
                                         stream()), 'dropout')
        ctx.save_for_backward(mask)
        ctx.p = float(p)
        return y

    @staticmethod
    @once
    def backward(ctx, gy):
        mask, = ctx.saved_tensors
        gx = torch.empty(gy.shape, device=gy.device, dtype=torch.float32)
        check(_lib.lib().vca_dropout_bwd_f32(gy.numel(), ptr(_c(gy)), mask.data_ptr(), ctx.p, ptr(gx), stream()),
              'dropout_bwd')
        return gx, None


def randn(shape, device):
    """Standard-normal tensor drawn on the device by the counter-based generator (its own stream per call site and
    training step, like the dropout masks; not torch's generator)."""
    y = torch.empty(shape, device=device, dtype=torch.float32)
    step = _rng_step.get(y.device)
    if step is None:
        step = _rng_step[y.device] = torch.zeros(1, device=y.device, dtype=torch.int32)
    _rng_calls[0] += 1
    seed = (_rng_seed[0] * 0x9E3779B1 + _rng_calls[0] * 0x85EBCA77 + 0x6E6F697365) & 0xFFFFFFFFFFFFFFFF
    check(_lib.lib().vca_randn_f32(y.numel(), seed, step.data_ptr(), ptr(y), stream()), 'randn')
    return y


def dropout(x, p, training):
    """nn.Dropout / F.dropout: identity in eval mode or with p = 0."""
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p)


def mel_down(mel, factor):
    """F.interpolate(mel, scale_factor=1/factor, mode='bilinear') of a (B,1,F,M) mel for factor 2 or 4."""
    b, one, f, m = mel.shape
    assert one == 1
    y = torch.empty((b, 1, f // factor, m // factor), device=mel.device, dtype=torch.float32)
    check(_lib.lib().vca_mel_down_f32(b, f, m, factor, ptr(_c(mel)), ptr(y), stream()), 'mel_down')
    return y


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
class SoftplusMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sign):
        x = _c(x)
        out = torch.zeros((), device=x.device, dtype=torch.float32)
        n = x.numel()
        check(_lib.lib().vca_softplus_sum_f32(n, ptr(x), sign, 1.0 / n, ptr(out), stream()), 'softplus_sum')
        ctx.save_for_backward(x)
        ctx.sign = sign
        return out

    @staticmethod
    @once
    def backward(ctx, g):
        x, = ctx.saved_tensors
        gx = torch.empty_like(x)
        n = x.numel()
        check(_lib.lib().vca_softplus_bwd_f32(n, ptr(x), ctx.sign, 1.0 / n, ptr(_c(g)), ptr(gx), stream()),
              'softplus_bwd')
        return gx, None


def softplus_mean(x, sign):
    """mean(softplus(sign * x))  (gan_loss, generator.py:363-366)."""
    return SoftplusMeanFn.apply(x, float(sign))


class L1MeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mul):
        a, b = _c(a), _c(b)
        out = torch.zeros((), device=a.device, dtype=torch.float32)
        n = a.numel()
        ctx.scale = abs(mul) / n
        check(_lib.lib().vca_l1_sum_f32(n, ptr(a), ptr(b), ctx.scale, ptr(out), stream()), 'l1_sum')
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    @once
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        check(_lib.lib().vca_l1_bwd_f32(a.numel(), ptr(a), ptr(b), ctx.scale, ptr(_c(g)), ptr(ga), stream()), 'l1_bwd')
        return ga, None, None


def l1_mean(a, b, mul=1.0):
    """mean |mul*a - mul*b| with the gradient flowing to `a` only (nn.L1Loss of an affine map of both arguments:
    the mel de-normalisation of vid_aud_grid.py:238-240 has slope `mul`)."""
    return L1MeanFn.apply(a, b.detach(), float(mul))


HEAD_SOFTPLUS, HEAD_L1, HEAD_SQSUM, HEAD_SUM = 0, 1, 2, 3


class LossHeadsFn(torch.autograd.Function):
    """Every loss term of one phase of the step in one launch, forward and backward.  `heads` is a tuple of
    (kind, x index, y index or None, sign, scale, ((slot, weight), ...)) over the tensor arguments: head e reduces
    v = scale * sum f(x[, y]) and adds weight * v into each of its slots of the returned (nslots,) vector, so a slot
    holds a reported term or a weighted total of several.  Gradients flow to the x tensors only."""

    @staticmethod
    def _table(heads, tensors, gxs):
        L = _lib.lib()
        chunk = L.vca_loss_head_chunk()
        t = _lib.LossHeads()
        assert len(heads) <= _lib.MAX_LOSS_HEADS
        t.nhead = len(heads)
        blocks = 0
        for e, (kind, xi, yi, sign, scale, slots) in enumerate(heads):
            h = t.h[e]
            x = tensors[xi]
            h.x, h.y = x.data_ptr(), (tensors[yi].data_ptr() if yi is not None else None)
            h.gx = gxs[e].data_ptr() if (gxs is not None and gxs[e] is not None) else None
            h.n, h.kind, h.block_begin, h.sign, h.scale = x.numel(), kind, blocks, sign, scale
            for k in range(3):
                h.slot[k], h.weight[k] = (slots[k][0], slots[k][1]) if k < len(slots) else (-1, 0.0)
            blocks += (x.numel() + chunk - 1) // chunk
        t.total_blocks = blocks
        return t

    @staticmethod
    def forward(ctx, heads, nslots, *tensors):
        assert nslots <= _lib.MAX_LOSS_SLOTS
        tensors = tuple(_c(t.detach()) for t in tensors)
        for t in tensors:
            assert t.is_cuda and t.dtype == torch.float32
        out = _zeros((nslots,), tensors[0].device)
        tab = LossHeadsFn._table(heads, tensors, None)
        check(_lib.lib().vca_loss_heads_fwd_f32(ctypes_ref(tab), ptr(out), stream()), 'loss_heads_fwd')
        ctx.heads = heads
        ctx.save_for_backward(*tensors)
        return out.clone() if _arena_active(out.device) else out   # the arena slice is recycled by the next step

    @staticmethod
    @once
    def backward(ctx, gout):
        tensors = ctx.saved_tensors
        heads = ctx.heads
        gxs, grads = [], [None] * len(tensors)
        for kind, xi, yi, sign, scale, slots in heads:
            if ctx.needs_input_grad[2 + xi]:
                assert grads[xi] is None, 'a tensor may feed one head only'
                grads[xi] = torch.empty_like(tensors[xi])
                gxs.append(grads[xi])
            else:
                gxs.append(None)
        if any(g is not None for g in gxs):
            tab = LossHeadsFn._table(heads, tensors, gxs)
            check(_lib.lib().vca_loss_heads_bwd_f32(ctypes_ref(tab), ptr(_c(gout)), stream()), 'loss_heads_bwd')
        return (None, None) + tuple(grads)


def ctypes_ref(obj):
    return ctypes.byref(obj)


def _arena_active(device):
    a = _arena.get(device)
    return a is not None and a[2]


def loss_heads(heads, nslots, tensors):
    """See LossHeadsFn.  Returns the (nslots,) vector of weighted sums."""
    return LossHeadsFn.apply(tuple(heads), nslots, *tensors)


# --------------------------------------------------------------------------- #
# recurrent / attention / sync loss (torch operators; HIP kernels replace them one by one)
# --------------------------------------------------------------------------- #
def _lin_cfg(w, rows):
    return ConvCfg((w.shape[0], w.shape[1], 1, 1), (rows, 1, 1, w.shape[1]), 1, (0, 0), 0, pshape=w.shape)


def _gemm_nt(x2d, w, bias):
    """x (P,I) @ w(O,I)^T + bias -> (P,O)"""
    return _conv_fwd_raw(x2d, w, bias, _lin_cfg(w, x2d.shape[0])).view(x2d.shape[0], w.shape[0])


def _gemm_nn(gy2d, w):
    """gy (P,O) @ w(O,I) -> (P,I)"""
    return _conv_dgrad_raw(gy2d, w, _lin_cfg(w, gy2d.shape[0])).view(gy2d.shape[0], w.shape[1])


def _gemm_tn(x2d, gy2d, w):
    """gy (P,O)^T @ x (P,I) -> (O,I), the gradient of w"""
    return _conv_wgrad_raw(x2d, gy2d, _lin_cfg(w, x2d.shape[0]))


def _lin_wgrad(x2d, gy2d, w, b):
    """Gradients (gw, gb) of y = x @ w^T + b; None where a gradient was added straight into the parameter's gradient
    storage (direct_param_grad)."""
    cfg = _lin_cfg(w, x2d.shape[0])
    bsink = _grad_sink(b)
    _bias_done[0] = False
    gw = _conv_wgrad_raw(x2d, gy2d, cfg, _grad_sink(w), bsink)
    gb = None
    if not (bsink is not None and _bias_done[0]):
        gb = _bias_grad(gy2d, b, cfg.cout)
    return gw, gb


def _rec_gemm(xs, ws, biases, mode):
    """The recurrent products of the two directions of a layer in one launch: y_i = x_i @ W_i^T + b_i (mode 0) or
    y_i = x_i @ W_i (mode 1, the input gradient).  Exact-fp32 precision falls back to the general GEMM path."""
    ns = _nsplit()
    if not ns:
        if mode == 0:
            return [_gemm_nt(x, w, b) for x, w, b in zip(xs, ws, biases)]
        return [_gemm_nn(x, w) for x, w in zip(xs, ws)]
    M, K = xs[0].shape
    N = ws[0].shape[0] if mode == 0 else ws[0].shape[1]
    ys = [torch.empty((M, N), device=xs[0].device, dtype=torch.float32) for _ in xs]
    packs = [_pack(w, 10 + mode) for w in ws]
    check(_lib.lib().vca_skinny_gemm_bf16(ns, 2, M, N, K, ptr(xs[0]), ptr(xs[1]), packs[0].data_ptr(),
                                          packs[1].data_ptr(), ptr(biases[0]), ptr(biases[1]), ptr(ys[0]),
                                          ptr(ys[1]), stream()), 'skinny_gemm')
    return ys


class GruLayerFn(torch.autograd.Function):
    """One bidirectional GRU layer over a time-major sequence: x (T,B,I) -> (T,B,2H).  The input-gate GEMMs of all
    time steps run on the matrix-core convolution kernels, the per-step recurrent products of both directions share
    one latency-oriented launch, the gate arithmetic is vca_gru_cell_*; backward is explicit BPTT."""

    @staticmethod
    def forward(ctx, x, *wts):
        x = _c(x)
        T, B, I = x.shape
        H = wts[1].shape[1]
        L = _lib.lib()
        dev = x.device
        out = torch.empty((T, B, 2 * H), device=dev, dtype=torch.float32)
        x2 = x.view(T * B, I)
        gi = [_gemm_nt(x2, wts[4 * d], wts[4 * d + 2]).view(T, B, 3 * H) for d in range(2)]
        hbuf = [torch.empty((T, B, H), device=dev, dtype=torch.float32) for _ in range(2)]
        save = [torch.empty((T, B, 4 * H), device=dev, dtype=torch.float32) for _ in range(2)]
        h0 = torch.zeros((B, H), device=dev, dtype=torch.float32)
        w_hh, b_hh = [wts[1], wts[5]], [wts[3], wts[7]]
        ns = _nsplit()
        for s in range(T):
            ts = (s, T - 1 - s)
            t0, t1 = ts
            if ns and H % 8 == 0:
                # recurrent product and gate arithmetic of the step in one launch for both directions
                packs = [_pack(w, 10) for w in w_hh]
                hp = [None, None] if s == 0 else [hbuf[0][s - 1], hbuf[1][T - s]]
                check(L.vca_gru_fwd_step_bf16(ns, B, H, ptr(hp[0]), ptr(hp[1]), packs[0].data_ptr(),
                                              packs[1].data_ptr(), ptr(b_hh[0].detach()), ptr(b_hh[1].detach()),
                                              ptr(gi[0][t0]), ptr(gi[1][t1]), ptr(hbuf[0][t0]), ptr(hbuf[1][t1]),
                                              out[t0].data_ptr(), out[t1].data_ptr() + 4 * H, 2 * H,
                                              ptr(save[0][t0]), ptr(save[1][t1]), stream()), 'gru_fwd_step')
                continue
            hp = [h0 if s == 0 else hbuf[0][s - 1], h0 if s == 0 else hbuf[1][T - s]]
            gh = _rec_gemm(hp, w_hh, b_hh, 0)
            check(L.vca_gru_cell_fwd2_f32(B, H, 2 * H, ptr(gi[0][t0]), ptr(gi[1][t1]), ptr(gh[0]), ptr(gh[1]),
                                          ptr(hp[0]), ptr(hp[1]), ptr(hbuf[0][t0]), ptr(hbuf[1][t1]),
                                          out[t0].data_ptr(), out[t1].data_ptr() + 4 * H, ptr(save[0][t0]),
                                          ptr(save[1][t1]), stream()), 'gru_cell_fwd2')
        ctx.save_for_backward(x, *wts, hbuf[0], save[0], hbuf[1], save[1])
        return out

    @staticmethod
    @once
    def backward(ctx, dout):
        x = ctx.saved_tensors[0]
        wts = ctx.saved_tensors[1:9]
        hbuf = [ctx.saved_tensors[9], ctx.saved_tensors[11]]
        save = [ctx.saved_tensors[10], ctx.saved_tensors[12]]
        dout = _c(dout)
        T, B, I = x.shape
        H = wts[1].shape[1]
        L = _lib.lib()
        dev = x.device
        x2 = x.view(T * B, I)
        zeros = torch.zeros((1, B, H), device=dev, dtype=torch.float32)
        # h_prev(t): the state the step at time t started from
        hprev = [torch.cat([zeros, hbuf[0][:-1]], 0), torch.cat([hbuf[1][1:], zeros], 0)]
        d_gi = [torch.empty((T, B, 3 * H), device=dev, dtype=torch.float32) for _ in range(2)]
        d_gh = [torch.empty((T, B, 3 * H), device=dev, dtype=torch.float32) for _ in range(2)]
        w_hh = [wts[1], wts[5]]
        dh_gemm, dh_dir = [None, None], [None, None]
        ns = _nsplit(True)      # 'mixed': single bf16 products in the backward recurrence, like every other gradient
        for s in range(T):
            t0, t1 = T - 1 - s, s
            nd = [torch.empty((B, H), device=dev, dtype=torch.float32) for _ in range(2)]
            if s > 0 and ns:
                # recurrent product of the previous step's gate gradients with the gate arithmetic of this step in
                # its epilogue: one launch per time step for both directions
                p0, p1 = T - s, s - 1
                packs = [_pack(w, 11) for w in w_hh]
                check(L.vca_gru_bwd_step_bf16(ns, B, H, ptr(d_gh[0][p0]), ptr(d_gh[1][p1]), packs[0].data_ptr(),
                                              packs[1].data_ptr(), 2 * H, dout[t0].data_ptr(),
                                              dout[t1].data_ptr() + 4 * H, ptr(dh_dir[0]), ptr(dh_dir[1]),
                                              ptr(save[0][t0]), ptr(save[1][t1]), ptr(hprev[0][t0]),
                                              ptr(hprev[1][t1]), ptr(d_gi[0][t0]), ptr(d_gi[1][t1]),
                                              ptr(d_gh[0][t0]), ptr(d_gh[1][t1]), ptr(nd[0]), ptr(nd[1]), stream()),
                      'gru_bwd_step')
            else:
                if s > 0:
                    dh_gemm = _rec_gemm([d_gh[0][T - s], d_gh[1][s - 1]], w_hh, [None, None], 1)
                check(L.vca_gru_cell_bwd2_f32(B, H, 2 * H, dout[t0].data_ptr(), dout[t1].data_ptr() + 4 * H,
                                              ptr(dh_gemm[0]), ptr(dh_gemm[1]), ptr(dh_dir[0]), ptr(dh_dir[1]),
                                              ptr(save[0][t0]), ptr(save[1][t1]), ptr(hprev[0][t0]),
                                              ptr(hprev[1][t1]), ptr(d_gi[0][t0]), ptr(d_gi[1][t1]),
                                              ptr(d_gh[0][t0]), ptr(d_gh[1][t1]), ptr(nd[0]), ptr(nd[1]), stream()),
                      'gru_cell_bwd2')
            dh_dir = nd
        dx = None
        grads = []
        for d in range(2):
            w_ih = wts[4 * d]
            gi2, gh2 = d_gi[d].view(T * B, 3 * H), d_gh[d].view(T * B, 3 * H)
            if any(ctx.needs_input_grad[1 + 4 * d:5 + 4 * d]):
                gw_ih, gb_ih = _lin_wgrad(x2, gi2, w_ih, wts[4 * d + 2])
                gw_hh, gb_hh = _lin_wgrad(hprev[d].view(T * B, H), gh2, w_hh[d], wts[4 * d + 3])
                grads += [gw_ih, gw_hh, gb_ih, gb_hh]
            else:                              # frozen weights (the ASR scorer under the CTC loss)
                grads += [None] * 4
            if ctx.needs_input_grad[0]:
                part = _gemm_nn(gi2, w_ih)
                dx = part if dx is None else dx + part
        return (dx.view(T, B, I) if dx is not None else None,) + tuple(grads)


def gru(x, mod, training):
    """x: (B, T, 512) -> (B, T, 1024); 2-layer bidirectional GRU (visual_front.py:20,33-34); dropout between the
    layers in training mode as nn.GRU(dropout=0.3) does."""
    h = x.transpose(0, 1)
    for layer in range(mod.num_layers):
        wts = []
        for sfx in ('', '_reverse'):
            wts += [getattr(mod, 'weight_ih_l%d%s' % (layer, sfx)), getattr(mod, 'weight_hh_l%d%s' % (layer, sfx)),
                    getattr(mod, 'bias_ih_l%d%s' % (layer, sfx)), getattr(mod, 'bias_hh_l%d%s' % (layer, sfx))]
        h = GruLayerFn.apply(h, *wts)
        if training and mod.dropout > 0 and layer + 1 < mod.num_layers:
            h = dropout(h, mod.dropout, True)
    return h.transpose(0, 1)


def as_lens(lens, device):
    """Valid-length vector as an int32 device tensor (no host sync when it already lives on the device)."""
    if torch.is_tensor(lens) and lens.device == device:
        return lens.to(torch.int32)
    return torch.tensor([int(l) for l in lens], dtype=torch.int32).to(device)


class AttnCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, lens, scale):
        q, k, v = _c(q), _c(k), _c(v)
        B, Tq, D = q.shape
        S = k.shape[1]
        out = torch.empty_like(q)
        probs = torch.empty((B, Tq, S), device=q.device, dtype=torch.float32)
        check(_lib.lib().vca_attention_fwd_f32(B, Tq, S, D, ptr(q), ptr(k), ptr(v), lens.data_ptr(), scale, ptr(out),
                                               ptr(probs), stream()), 'attention_fwd')
        ctx.save_for_backward(q, k, v, probs)
        ctx.scale = scale
        return out

    @staticmethod
    @once
    def backward(ctx, g):
        q, k, v, probs = ctx.saved_tensors
        B, Tq, D = q.shape
        S = k.shape[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        check(_lib.lib().vca_attention_bwd_f32(B, Tq, S, D, ptr(_c(g)), ptr(q), ptr(k), ptr(v), ptr(probs), ctx.scale,
                                               ptr(dq), ptr(dk), ptr(dv), stream()), 'attention_bwd')
        return dq, dk, dv, None, None


def masked_attention(q, k, v, lens, scale):
    """softmax(q k^T * scale, keys >= lens[b] masked) v   (generator.py:161-167); lens: int32 device tensor."""
    assert lens.dtype == torch.int32 and lens.is_cuda
    return AttnCoreFn.apply(q, k, v, lens, float(scale))


class SyncNceFn(torch.autograd.Function):
    """Symmetric InfoNCE over the time steps of each sample (generator.py:352-361)."""

    @staticmethod
    def forward(ctx, v, a, temp):
        v, a = _c(v), _c(a)
        B, T, C = v.shape
        S = torch.empty((B, T, T), device=v.device, dtype=torch.float32)
        stats = torch.empty((B, 4, T), device=v.device, dtype=torch.float32)
        loss = torch.empty((B,), device=v.device, dtype=torch.float32)
        check(_lib.lib().vca_sync_nce_fwd_f32(B, T, C, ptr(v), ptr(a), float(temp), ptr(S), ptr(stats), ptr(loss),
                                              stream()), 'sync_nce_fwd')
        ctx.save_for_backward(v, a, S, stats)
        ctx.temp = float(temp)
        return loss

    @staticmethod
    @once
    def backward(ctx, g):
        v, a, S, stats = ctx.saved_tensors
        B, T, C = v.shape
        dv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        da = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        if dv is not None or da is not None:
            check(_lib.lib().vca_sync_nce_bwd_f32(B, T, C, ptr(v), ptr(a), ptr(S), ptr(stats), ptr(_c(g)), ctx.temp,
                                                  ptr(dv), ptr(da), stream()), 'sync_nce_bwd')
        return dv, da, None


class SyncCosFn(torch.autograd.Function):
    """5 - mean_t |cos(v_t, a_t)| per sample (generator.py:348-350)."""

    @staticmethod
    def forward(ctx, v, a):
        v, a = _c(v), _c(a)
        B, T, C = v.shape
        cosv = torch.empty((B, T), device=v.device, dtype=torch.float32)
        loss = torch.empty((B,), device=v.device, dtype=torch.float32)
        check(_lib.lib().vca_sync_cos_fwd_f32(B, T, C, ptr(v), ptr(a), ptr(cosv), ptr(loss), stream()), 'sync_cos_fwd')
        ctx.save_for_backward(v, a, cosv)
        return loss

    @staticmethod
    @once
    def backward(ctx, g):
        v, a, cosv = ctx.saved_tensors
        B, T, C = v.shape
        dv = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        da = torch.empty_like(a) if ctx.needs_input_grad[1] else None
        if dv is not None or da is not None:
            check(_lib.lib().vca_sync_cos_bwd_f32(B, T, C, ptr(v), ptr(a), ptr(cosv), ptr(_c(g)), ptr(dv), ptr(da),
                                                  stream()), 'sync_cos_bwd')
        return dv, da


def sync_loss(v_feat, a_feat, temp, gen):
    """generator.py:347-361: per-sample loss (B,) from the (B, T, C) visual and audio embeddings."""
    if gen:
        return SyncCosFn.apply(v_feat, a_feat)
    return SyncNceFn.apply(v_feat, a_feat, temp)


class CtcNllFn(torch.autograd.Function):
    """Per-sample CTC negative log-likelihood of frame logits (B, T, V), log-softmax over V included; the semantics of
    torch.nn.functional.ctc_loss(log_softmax(logits), ..., reduction='none').  Gradient for the logits only."""

    @staticmethod
    def forward(ctx, logits, targets, in_len, tgt_len, blank, zero_infinity):
        logits = _c(logits)
        B, T, V = logits.shape
        dev = logits.device
        targets, in_len, tgt_len = (_c(t.to(device=dev, dtype=torch.int32)) for t in (targets, in_len, tgt_len))
        assert targets.dim() == 2 and targets.shape[0] == B and in_len.shape == (B,) and tgt_len.shape == (B,)
        Lmax = targets.shape[1]
        alpha = torch.empty((B, T, 2 * Lmax + 1), device=dev, dtype=torch.float64)
        nll = torch.empty((B,), device=dev, dtype=torch.float32)
        check(_lib.lib().vca_ctc_loss_fwd_f32(B, T, V, Lmax, ptr(logits), targets.data_ptr(), in_len.data_ptr(),
                                              tgt_len.data_ptr(), blank, 1 if zero_infinity else 0,
                                              alpha.data_ptr(), ptr(nll), stream()), 'ctc_loss_fwd')
        ctx.save_for_backward(logits, targets, in_len, tgt_len, alpha, nll)
        ctx.cfg = (blank, zero_infinity)
        return nll

    @staticmethod
    @once
    def backward(ctx, gnll):
        logits, targets, in_len, tgt_len, alpha, nll = ctx.saved_tensors
        blank, zero_infinity = ctx.cfg
        B, T, V = logits.shape
        gl = torch.empty_like(logits)
        check(_lib.lib().vca_ctc_loss_bwd_f32(B, T, V, targets.shape[1], ptr(logits), targets.data_ptr(),
                                              in_len.data_ptr(), tgt_len.data_ptr(), blank, 1 if zero_infinity else 0,
                                              alpha.data_ptr(), ptr(nll), ptr(_c(gnll)), ptr(gl), stream()),
              'ctc_loss_bwd')
        return gl, None, None, None, None, None


def ctc_nll(logits, targets, in_len, tgt_len, blank=0, zero_infinity=True):
    """CTC loss per sample, (B,): logits (B, T, V) raw scores, targets (B, Lmax) class ids (padding ignored beyond
    tgt_len), in_len / tgt_len (B,) valid frames / labels.  blank 0 is the ASR scorer's '_'
    (ASR_model/GRID/src/data/vid_aud_GRID_test.py:20)."""
    return CtcNllFn.apply(logits, targets, in_len, tgt_len, int(blank), bool(zero_infinity))


# --------------------------------------------------------------------------- #
# flat parameter storage + fused Adam
# --------------------------------------------------------------------------- #
_flat_epoch = {}     # data_ptr of a flat parameter buffer -> its epoch cell (shared with its parameters)
_flat_params = {}    # data_ptr of a flat parameter buffer -> its parameters


class _RepackPlan:
    """All split-bf16 packed copies of one flat buffer's weights, refreshed by a single launch after the optimizer
    step that rewrote the buffer (instead of one small launch per weight and role at first use).  The descriptor
    table is rebuilt when the set of packed copies changes; never while a graph is being captured (the per-use path
    covers that step)."""

    def __init__(self, params):
        self.params = params
        self.sig = None
        self.table = None
        self.meta = (0, 0, 0)

    def _entries(self):
        out = []
        for p in self.params:
            ent = _pack_cache.get(id(p))
            if ent is None or ent[0]() is not p:
                continue
            cout, cin, taps = _wdims(p) if p.dim() >= 2 else (0, 0, 0)
            if taps < 1 or taps > 32 or p.dim() == 5:
                continue
            for mode in (10, 11, 12):
                if mode in ent[1]:
                    out.append((p, mode - 10, ent[1][mode][1], cout, cin, taps, ent))
        return out

    def run(self):
        ents = self._entries()
        if not ents:
            return
        sig = tuple((id(p), m, t.data_ptr()) for p, m, t, *_ in ents)
        if sig != self.sig:
            if torch.cuda.is_current_stream_capturing():
                return
            import numpy as np
            rec = np.zeros((len(ents), 5), dtype=np.int64)      # 40-byte records: src, dst, (mode,Cout), (Cin,taps), (tile_begin,0)
            tiles, mt = 0, 0
            for i, (p, m, t, cout, cin, taps, _) in enumerate(ents):
                n, k = (cout, cin) if m == 0 else ((cin, cout) if m == 1 else (4 * cin, cout))
                rec[i, 0], rec[i, 1] = p.data_ptr(), t.data_ptr()
                rec[i, 2] = m | (cout << 32)
                rec[i, 3] = cin | (taps << 32)
                rec[i, 4] = tiles
                tiles += ((k + 31) // 32) * (((n + 31) // 32 * 32) // 8)
                mt = max(mt, taps)
            self.table = torch.from_numpy(rec).to(ents[0][0].device)
            self.meta = (len(ents), tiles, mt)
            self.sig = sig
        nd, tiles, mt = self.meta
        check(_lib.lib().vca_pack_weight_bf16_batch(self.table.data_ptr(), nd, tiles, mt, stream()), 'pack_batch')
        for p, m, t, cout, cin, taps, ent in ents:
            ent[1][m + 10] = (_wkey(p), t)


def flatten_module(module, storage=None):
    """Re-home every parameter of `module` (already on its device) into one flat fp32 buffer and give it a flat,
    persistent gradient buffer: p.data and p.grad become views.  Returns (flat_params, flat_grads).  `storage`:
    optional (params, grads) pair of zeroed buffers of module_flat_size(module) floats to use instead of new ones."""
    ps = [p for p in module.parameters()]
    sizes = [(p.numel() + 3) // 4 * 4 for p in ps]
    total = sum(sizes)
    dev = ps[0].device
    if storage is None:
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        gflat = torch.zeros(total, device=dev, dtype=torch.float32)
    else:
        flat, gflat = storage
        assert flat.numel() == total and gflat.numel() == total and flat.device == dev
    off = 0
    epoch = [0]
    for p, sz in zip(ps, sizes):
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        p.grad = gflat[off:off + n].view(p.shape)
        p._vca_epoch = epoch
        off += sz
    _flat_epoch[flat.data_ptr()] = epoch
    _flat_params[flat.data_ptr()] = ps
    _param_epoch[0] += 1
    return flat, gflat


def module_flat_size(module):
    return sum((p.numel() + 3) // 4 * 4 for p in module.parameters())


def flatten_modules(modules):
    """flatten_module for several networks that share an optimizer: their flat buffers are consecutive slices of one
    parameter buffer and one gradient buffer, so the data-parallel exchange of the whole side is a single all-reduce.
    Returns ([(flat_params, flat_grads) per module], all_params, all_grads)."""
    sizes = [module_flat_size(m) for m in modules]
    dev = next(modules[0].parameters()).device
    allp = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
    allg = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
    out, off = [], 0
    for m, sz in zip(modules, sizes):
        out.append(flatten_module(m, (allp[off:off + sz], allg[off:off + sz])))
        off += sz
    return out, allp, allg


class LossScaler:
    """Dynamic loss scale for half-precision gradients, all of it in device words (a captured step keeps adapting):
    seed the backward passes with seed(), hand the scaler to FusedAdam.step(), call update() once per training step
    after both optimizers.  Semantics of torch.cuda.amp.GradScaler: an overflow (Inf / NaN in a side's gradients)
    skips that side's optimizer step and multiplies the scale by `backoff`; `interval` clean steps in a row multiply it
    by `growth`."""

    def __init__(self, device, init_scale=65536.0, growth=2.0, backoff=0.5, interval=2000):
        self.scale = torch.tensor([init_scale, 1.0 / init_scale], device=device, dtype=torch.float32)
        self.state = torch.zeros(4, device=device, dtype=torch.int32)   # clean steps, flag D, flag G, skipped steps
        self.growth, self.backoff, self.interval = float(growth), float(backoff), int(interval)

    def seed(self, onehot):
        """The backward seed: `onehot` times the current scale (a device-side product: follows the scale in replay)."""
        return onehot * self.scale[0]

    def update(self):
        check(_lib.lib().vca_loss_scale_update(self.scale.data_ptr(), self.state.data_ptr(), self.growth, self.backoff,
                                               self.interval, stream()), 'loss_scale_update')

    def set_scale(self, value):
        self.scale.copy_(torch.tensor([value, 1.0 / value], dtype=torch.float32))

    @property
    def value(self):
        return float(self.scale[0])

    @property
    def skipped(self):
        return int(self.state[3])


class FusedAdam:
    """torch.optim.Adam semantics (coupled L2 weight decay, optional amsgrad; train.py:82-83) as one kernel launch
    per flat buffer.  `groups` is a list of (flat_params, flat_grads) from flatten_module.

    The step count and the learning rate live in device words read by the kernel, so a step captured into a HIP
    graph keeps counting and keeps following `param_groups[0]['lr']` (the MultiStepLR schedules of train.py:85-86):
    step() refreshes the device word when the host value changed, and a caller that replays a captured step calls
    sync_lr() before each replay (GraphedStep does)."""

    def __init__(self, groups, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        self.groups = groups
        self.lr, self.betas, self.eps, self.weight_decay, self.amsgrad = lr, betas, eps, weight_decay, amsgrad
        self.param_groups = [dict(lr=lr)]       # the reference reads param_groups[0]['lr'] for logging
        self.t = 0
        dev = groups[0][0].device
        self.t_dev = torch.zeros(1, device=dev, dtype=torch.int32)   # graph-replayable step count
        self.lr_dev = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
        self._lr_written = float(lr)
        self.state = [dict(m=torch.zeros_like(f), v=torch.zeros_like(f),
                           vmax=torch.zeros_like(f) if amsgrad else None) for f, _ in groups]
        self.repack = [_RepackPlan(_flat_params[f.data_ptr()]) if f.data_ptr() in _flat_params else None
                       for f, _ in groups]

    def zero_grad(self, set_to_none=False):
        for _, g in self.groups:
            g.zero_()

    def sync_lr(self):
        """Copy param_groups[0]['lr'] to the device word the kernel reads, if it changed.  Never call while the
        stream is capturing (the copy would be baked into the graph with today's value)."""
        lr = float(self.param_groups[0]['lr'])
        if lr != self._lr_written:
            self.lr_dev.copy_(torch.full((1,), lr, dtype=torch.float32), non_blocking=False)
            self._lr_written = lr

    def step(self, scaler=None, side=0):
        """scaler: a LossScaler whose scale the gradients carry; side: 0 / 1, which of its overflow flags this
        optimizer raises (discriminator / generator side).  With a scaler the step is skipped on the device when any
        gradient is Inf / NaN (parameters, moments and the step count stay)."""
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        L = _lib.lib()
        if scaler is not None:
            flag = scaler.state.data_ptr() + 4 * (1 + side)
            for _, g in self.groups:
                check(L.vca_nonfinite_f32(g.numel(), ptr(g), flag, stream()), 'nonfinite')
            check(L.vca_loss_scale_count(self.t_dev.data_ptr(), flag, stream()), 'loss_scale_count')
            self.t += 1                # host copy: an upper bound once a step has been skipped
        else:
            self.t += 1
            self.t_dev += 1
        lr = self.param_groups[0]['lr']
        for (f, g), st in zip(self.groups, self.state):
            if scaler is not None:
                check(L.vca_adam_scaled_f32(f.numel(), ptr(f), ptr(g), ptr(st['m']), ptr(st['v']), ptr(st['vmax']),
                                            self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                            self.t_dev.data_ptr(), self.lr_dev.data_ptr(), scaler.scale.data_ptr(),
                                            flag, stream()), 'adam_scaled')
            else:
                check(L.vca_adam_f32(f.numel(), ptr(f), ptr(g), ptr(st['m']), ptr(st['v']), ptr(st['vmax']), lr,
                                     self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t,
                                     self.t_dev.data_ptr(), self.lr_dev.data_ptr(), stream()), 'adam')
            ep = _flat_epoch.get(f.data_ptr())
            if ep is not None:
                ep[0] += 1
            else:
                _param_epoch[0] += 1
        if _precision[0] != 'fp32':
            for plan in self.repack:
                if plan is not None:
                    plan.run()
