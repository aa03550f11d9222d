"""ctypes binding of libvca_hip.so (the C ABI declared in include/vca_hip.h).

There is no CPU fallback: importing this module without the built library raises, and every call
checks the kernel's return code.  Build with `python __graft_entry__.py` (hipcc, gfx950).

The sources are compiled twice: libvca_hip.so computes its 16-bit matrix products in bf16, libvca_hip_f16.so (same
entry points, -DVCA_E16_FP16) in IEEE half.  lib() returns the one ops.set_precision() selected (ELEM); every
entry point without matrix products is the same code in both.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_HERE, 'libvca_hip.so')
LIB_PATH_F16 = os.path.join(_HERE, 'libvca_hip_f16.so')
ELEM = ['bf16']          # element type of the matrix products: 'bf16' or 'fp16' (set by ops.set_precision)
SOURCES = ('conv_f32.hip', 'conv_bf16.hip', 'pointwise.hip', 'misc.hip', 'rnn.hip', 'attention.hip', 'stem.hip',
           'sync_loss.hip', 'layout.hip', 'audio.hip', 'video.hip', 'ctc.hip', 'wgrad_cin1.hip', 'linear.hip')


def build(force=False, verbose=False):
    """Compile every HIP source into the two shared libraries for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, 'common.h'), os.path.join(_HERE, '..', 'include', 'vca_hip.h')]
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    procs = []
    for path, flags in ((LIB_PATH, []), (LIB_PATH_F16, ['-DVCA_E16_FP16'])):
        if not force and os.path.exists(path) and all(os.path.getmtime(path) >= os.path.getmtime(d) for d in deps):
            continue
        cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-shared', '-fPIC'] + flags + ['-o', path] + srcs
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))          # the two builds side by side
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    return LIB_PATH


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ('N', 'D', 'Hin', 'Win', 'Cin', 'OD', 'OH', 'OW', 'Cout', 'KD', 'KH', 'KW', 'stride', 'pad_h',
                 'pad_w', 'pad_d', 'dil', 'ups', 'act')] + [('slope', ctypes.c_float), ('out_scale', ctypes.c_float),
                                                               ('d2s_h', ctypes.c_int), ('d2s_w', ctypes.c_int)]


class LinDesc(ctypes.Structure):
    _fields_ = [('M', ctypes.c_int), ('N', ctypes.c_int), ('K', ctypes.c_int), ('a_tm', ctypes.c_int),
                ('a_ck', ctypes.c_int), ('a_bs', ctypes.c_longlong), ('a_ts', ctypes.c_longlong),
                ('a_ks', ctypes.c_longlong), ('c_tm', ctypes.c_int), ('c_cn', ctypes.c_int),
                ('c_bs', ctypes.c_longlong), ('c_ts', ctypes.c_longlong), ('c_ns', ctypes.c_longlong)]


class LossHead(ctypes.Structure):
    _fields_ = [('x', ctypes.c_void_p), ('y', ctypes.c_void_p), ('gx', ctypes.c_void_p), ('n', ctypes.c_longlong),
                ('kind', ctypes.c_int), ('block_begin', ctypes.c_int), ('sign', ctypes.c_float),
                ('scale', ctypes.c_float), ('slot', ctypes.c_int * 3), ('weight', ctypes.c_float * 3)]


MAX_LOSS_HEADS, MAX_LOSS_SLOTS = 24, 16


class LossHeads(ctypes.Structure):
    _fields_ = [('nhead', ctypes.c_int), ('total_blocks', ctypes.c_int), ('h', LossHead * MAX_LOSS_HEADS)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_LL = ctypes.c_longlong
_SIGS = {
    'vca_abi_version': ([], _I),
    'vca_elem_type': ([], _I),
    'vca_conv_fwd_f32': ([ctypes.POINTER(ConvDesc)] + [_P] * 10, _I),
    'vca_conv_wgrad_f32': ([ctypes.POINTER(ConvDesc)] + [_P] * 7, _I),
    'vca_pack_weight_f32': ([_I, _I, _I, _I, _P, _P, _F, _P], _I),
    'vca_unpack_wgrad_f32': ([_I, _I, _I, _P, _P, _F, _I, _P], _I),
    'vca_unpack_wgrad_batch_f32': ([_P, _I, _I, _P], _I),
    'vca_conv_fwd_bf16': ([ctypes.POINTER(ConvDesc), _I] + [_P] * 10, _I),
    'vca_conv_fwd_bf16_ws': ([ctypes.POINTER(ConvDesc), _I] + [_P] * 10 + [_LL, _P], _I),
    'vca_conv_fwd_bf16_ws_floats': ([ctypes.POINTER(ConvDesc)], _LL),
    'vca_conv_wgrad_cin1_ws_floats': ([ctypes.POINTER(ConvDesc)], _LL),
    'vca_conv_wgrad_cin1_f32': ([ctypes.POINTER(ConvDesc), _P, _P, _P, _LL, _P, _P, _P], _I),
    'vca_conv_wgrad_bf16': ([ctypes.POINTER(ConvDesc), _I] + [_P] * 8, _I),
    'vca_pack_weight_bf16_bytes': ([_I, _I, _I, _I], _LL),
    'vca_pack_weight_bf16': ([_I, _I, _I, _I, _P, _P, _P], _I),
    'vca_pack_weight_bf16_batch': ([_P, _I, _I, _I, _P], _I),
    'vca_linear_bf16': ([ctypes.POINTER(LinDesc), _I, _P, _P, _P, _P, _P, _P], _I),
    'vca_gemm_bf16_ws_floats': ([ctypes.POINTER(LinDesc)], _LL),
    'vca_gemm_bf16': ([ctypes.POINTER(LinDesc), _I, _P, _P, _P, _P, _P, _P, _LL, _P], _I),
    'vca_stem_pack_weight': ([_P, _P, _P], _I),
    'vca_stem_pack_bytes': ([], _LL),
    'vca_stem_fwd_bf16': ([_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P], _I),
    'vca_stem_patches_f32': ([_I, _I, _I, _I, _P, _P, _P], _I),
    'vca_colstats_f32': ([_I, _I, _P, _P, _P], _I),
    'vca_bn_finalize_f32': ([_I, _F, _P, _I, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P], _I),
    'vca_bn_eval_coeff_f32': ([_I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P], _I),
    'vca_affine_act_f32': ([_I, _I, _P, _P, _P, _P, _P, _I, _F, _F, _P, _P], _I),
    'vca_affine_act_bwd_f32': ([_I, _I] + [_P] * 8 + [_I, _F, _F, _I, _P, _P, _P, _P], _I),
    'vca_ctc_loss_fwd_f32': ([_I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P], _I),
    'vca_ctc_loss_bwd_f32': ([_I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P], _I),
    'vca_affine_act_bwd2_f32': ([_I, _I] + [_P] * 8 + [_I, _F, _F, _I] + [_P] * 8, _I),
    'vca_affine_act_pool3_fwd_f32': ([_I, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _P, _P], _I),
    'vca_pool2_f32': ([_I] * 7 + [_P, _P, _P], _I),
    'vca_leaky_pool2_fwd_f32': ([_I, _I, _I, _I, _F, _P, _P, _P, _P], _I),
    'vca_leaky_pool2_bwd_f32': ([_I, _I, _I, _I, _F, _P, _P, _P, _P, _P], _I),
    'vca_maxpool3s2_fwd_f32': ([_I, _I, _I, _I, _P, _P, _P, _P], _I),
    'vca_maxpool3s2_bwd_f32': ([_I, _I, _I, _I, _P, _P, _P, _P], _I),
    'vca_attention_fwd_f32': ([_I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _P], _I),
    'vca_attention_bwd_f32': ([_I, _I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P], _I),
    'vca_gru_cell_fwd_f32': ([_I, _I, _P, _P, _P, _P, _P, _I, _P, _P], _I),
    'vca_gru_cell_bwd_f32': ([_I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P], _I),
    'vca_gru_cell_fwd2_f32': ([_I, _I, _I] + [_P] * 13, _I),
    'vca_gru_cell_bwd2_f32': ([_I, _I, _I] + [_P] * 17, _I),
    'vca_gru_fwd_step_bf16': ([_I, _I, _I] + [_P] * 12 + [_I, _P, _P, _P], _I),
    'vca_gru_bwd_step_bf16': ([_I, _I, _I, _P, _P, _P, _P, _I] + [_P] * 15, _I),
    'vca_skinny_gemm_bf16': ([_I, _I, _I, _I, _I] + [_P] * 9, _I),
    'vca_softplus_sum_f32': ([_I, _P, _F, _F, _P, _P], _I),
    'vca_softplus_bwd_f32': ([_I, _P, _F, _F, _P, _P, _P], _I),
    'vca_l1_sum_f32': ([_LL, _P, _P, _F, _P, _P], _I),
    'vca_l1_bwd_f32': ([_LL, _P, _P, _F, _P, _P, _P], _I),
    'vca_sync_nce_fwd_f32': ([_I, _I, _I, _P, _P, _F, _P, _P, _P, _P], _I),
    'vca_sync_nce_bwd_f32': ([_I, _I, _I, _P, _P, _P, _P, _P, _F, _P, _P, _P], _I),
    'vca_sync_cos_fwd_f32': ([_I, _I, _I, _P, _P, _P, _P, _P], _I),
    'vca_sync_cos_bwd_f32': ([_I, _I, _I, _P, _P, _P, _P, _P, _P, _P], _I),
    'vca_concat2_f32': ([_I, _I, _I, _I, _I, _P, _LL, _LL, _LL, _P, _LL, _LL, _LL, _P, _P], _I),
    'vca_concat2_bwd_f32': ([_I, _I, _I, _I, _I, _I, _P, _LL, _LL, _LL, _I, _P, _P], _I),
    'vca_colsum_groups': ([_I, _I], _I),
    'vca_colsum_add_f32': ([_I, _I, _P, _P, _P, _P], _I),
    'vca_noalloc_init': ([], _I),
    'vca_colsum_noalloc_max_floats': ([], _LL),
    'vca_colsum_add_noalloc_f32': ([_I, _I, _P, _P, _P], _I),
    'vca_hband_expand_f32': ([_I, _I, _I, _I, _I, _I, _P, _P, _P, _P], _I),
    'vca_hband_reduce_f32': ([_I, _I, _I, _I, _I, _I, _P, _P, _P], _I),
    'vca_window_mean_f32': ([_I, _I, _I, _I, _I, _I, _P, _P, _P], _I),
    'vca_window_mean_bwd_f32': ([_I, _I, _I, _I, _I, _I, _P, _P, _P], _I),
    'vca_dropout_f32': ([_LL, _P, _F, ctypes.c_ulonglong, _P, _P, _P, _P], _I),
    'vca_randn_f32': ([_LL, ctypes.c_ulonglong, _P, _P, _P], _I),
    'vca_dropout_bwd_f32': ([_LL, _P, _P, _F, _P, _P], _I),
    'vca_mel_down_f32': ([_I, _I, _I, _I, _P, _P, _P], _I),
    'vca_reflect_pad_f32': ([_I, _I, _I, _I, _P, _P, _P], _I),
    'vca_mag_phase_f32': ([_I, _I, _I, _P, _P, _P, _P], _I),
    'vca_polar_to_rect_f32': ([_I, _I, _I, _P, _P, _P, _P], _I),
    'vca_gl_project_f32': ([_I, _I, _I, _P, _P, _P, _P], _I),
    'vca_istft_finish_f32': ([_I, _I, _I, _I, _F, _F, _P, _P, _P, _P], _I),
    'vca_deemphasis_clip_f32': ([_I, _I, _F, _P, _P, _P], _I),
    'vca_build_video_f32': ([_I] * 10 + [_F, _F] + [_P] * 11 + [_I, _P, _P], _I),
    'vca_loss_head_chunk': ([], _I),
    'vca_loss_heads_fwd_f32': ([_P, _P, _P], _I),
    'vca_loss_heads_bwd_f32': ([_P, _P, _P], _I),
    'vca_adam_f32': ([_LL, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _P, _P, _P], _I),
    'vca_nonfinite_f32': ([_LL, _P, _P, _P], _I),
    'vca_loss_scale_count': ([_P, _P, _P], _I),
    'vca_adam_scaled_f32': ([_LL, _P, _P, _P, _P, _P, _F, _F, _F, _F, _P, _P, _P, _P, _P], _I),
    'vca_loss_scale_update': ([_P, _P, _F, _F, _I, _P], _I),
}

_libs = {}


def lib():
    """The library whose matrix products use the element type in ELEM."""
    L = _libs.get(ELEM[0])
    if L is None:
        path = {'bf16': LIB_PATH, 'fp16': LIB_PATH_F16}[ELEM[0]]
        if not os.path.exists(path):
            raise RuntimeError('%s is not built (%s); run `python __graft_entry__.py`' % (os.path.basename(path), path))
        L = ctypes.CDLL(path)
        for name, (args, res) in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        assert L.vca_elem_type() == {'bf16': 0, 'fp16': 1}[ELEM[0]], path
        _libs[ELEM[0]] = L
    return L


def exported_symbols():
    return sorted(_SIGS)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with code %d' % (what, rc))
