"""CPU-side checks: the shared library builds for gfx950, loads, and exports every symbol include/vca_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_declared_symbols():
    import __graft_entry__ as ge
    ge.build()
    from vcagan_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'vca_hip.h')).read()
    declared = set(re.findall(r'\b(?:int|long long)\s+(vca_\w+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.exported_symbols())
    # both builds of the sources (bf16 and IEEE-half matrix operands) export the same entry points
    for path, elem in ((_lib.LIB_PATH, 0), (_lib.LIB_PATH_F16, 1)):
        L = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(L, name), (path, name)
        assert L.vca_abi_version() >= 1 and L.vca_elem_type() == elem


def _desc(_lib, **kw):
    base = dict(N=32, D=1, Hin=80, Win=300, Cin=1, OD=1, OH=80, OW=300, Cout=32, KD=1, KH=5, KW=5, stride=1, pad_h=2,
                pad_w=2, pad_d=0, dil=1, ups=0, act=0, slope=0.0, out_scale=1.0, d2s_h=0, d2s_w=0)
    base.update(kw)
    return _lib.ConvDesc(**base)


def test_single_channel_weight_gradient_workspace_query_and_argument_checks():
    """Host-only parts of the single-input-channel weight-gradient entry points (generator.py:291,346 layers): the
    workspace query covers exactly the 5x5 / 3x3, stride <= 2, one-channel descriptors, and the launch entry point
    rejects (-22, before any GPU call) a descriptor the query does not cover, null pointers and a short workspace."""
    import __graft_entry__ as ge
    ge.build()
    from vcagan_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    q = L.vca_conv_wgrad_cin1_ws_floats
    q.restype, q.argtypes = ctypes.c_longlong, [ctypes.POINTER(_lib.ConvDesc)]
    f = L.vca_conv_wgrad_cin1_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.POINTER(_lib.ConvDesc)] + [ctypes.c_void_p] * 3 + [ctypes.c_longlong] + [ctypes.c_void_p] * 3
    d = _desc(_lib)
    n = q(ctypes.byref(d))
    # 32 x 80 output rows in units of two rows = 1280 slabs of (25 taps + the bias row) x 32 channels
    assert n == 1280 * 26 * 32
    s2 = _desc(_lib, Cout=128, KH=3, KW=3, stride=2, pad_h=1, pad_w=1, OH=40, OW=150)
    assert q(ctypes.byref(s2)) == 4 * 1280 * 10 * 32
    for bad in (dict(Cin=2), dict(KH=7, KW=7), dict(stride=3), dict(ups=1), dict(act=1), dict(dil=2),
                dict(Win=700, OW=700), dict(KD=5, D=5, OD=5, pad_d=2)):
        b = _desc(_lib, **bad)
        assert q(ctypes.byref(b)) == 0, bad
        assert f(ctypes.byref(b), 8, 8, 8, 1 << 40, 8, None, None) == -22, bad
    assert f(ctypes.byref(d), None, 8, 8, n, 8, None, None) == -22          # null input
    assert f(ctypes.byref(d), 8, 8, 8, n - 1, 8, None, None) == -22         # workspace one float short


def test_gru_forward_step_rejects_hidden_sizes_it_cannot_slice():
    """vca_gru_fwd_step_bf16 (nn.GRU step, visual_front.py:20,33-34) reads h_prev in 16-wide reduction slices: a hidden
    size that is a multiple of 8 but not of 16 would read past a row, so it is refused (-22) before any GPU call."""
    import __graft_entry__ as ge
    ge.build()
    from vcagan_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    f = L.vca_gru_fwd_step_bf16
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 12 + [ctypes.c_int] + [ctypes.c_void_p] * 3
    ptrs = [8] * 12
    for H in (8, 24, 520):
        assert f(3, 32, H, *ptrs, H, 8, 8, None) == -22, H
    assert f(2, 32, 512, *ptrs, 512, 8, 8, None) == -22          # nsplit must be 1 or 3
    assert f(3, 32, 512, *([8, 8] + [None] + [8] * 9), 512, 8, 8, None) == -22   # null weight
