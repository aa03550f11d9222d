"""Deterministic, platform-independent weights and inputs for parity tests.

TEST INFRASTRUCTURE ONLY (see oracle/cpu_ref.py).  Everything here is drawn from
numpy.random.RandomState, so this container and the GPU box regenerate exactly
the same tensors and no large weight file has to be committed.
"""
import zlib

import numpy as np
import torch
import torch.nn as nn


def _rs(seed, name):
    return np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))


def det_fill(module, seed):
    """Overwrite every parameter/buffer of `module` with a seeded draw keyed by its state_dict name."""
    kinds = {}
    for mname, m in module.named_modules():
        pre = mname + '.' if mname else ''
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            kinds[pre + 'weight'] = 'bn_w'
            kinds[pre + 'bias'] = 'bn_b'
            kinds[pre + 'running_mean'] = 'bn_m'
            kinds[pre + 'running_var'] = 'bn_v'
        elif isinstance(m, nn.PReLU):
            kinds[pre + 'weight'] = 'prelu'
        elif isinstance(m, nn.GRU):
            for n, _ in m.named_parameters(recurse=False):
                kinds[pre + n] = 'gru'
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if name.endswith('num_batches_tracked'):
                t.zero_()
                continue
            r = _rs(seed, name)
            kind = kinds.get(name)
            shape = tuple(t.shape)
            if kind == 'bn_w':
                v = r.uniform(0.9, 1.1, shape)
            elif kind == 'bn_b':
                v = r.uniform(-0.1, 0.1, shape)
            elif kind == 'bn_m':
                v = r.normal(0.0, 0.1, shape)
            elif kind == 'bn_v':
                v = r.uniform(0.5, 1.5, shape)
            elif kind == 'prelu':
                v = r.uniform(0.2, 0.3, shape)
            elif kind == 'gru':
                v = r.uniform(-1.0, 1.0, shape) / np.sqrt(512.0)
            elif t.dim() >= 2:
                fan_in = int(np.prod(shape[1:]))
                v = r.uniform(-1.0, 1.0, shape) * np.sqrt(3.0 / fan_in)
            else:
                v = r.uniform(-0.05, 0.05, shape)
            t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module


def make_batch(B, T, H, seed, lens=None):
    """Synthetic clip batch of the shapes train.py:166 receives (normalised video, mel in [-1,1], spec in [0,1])."""
    r = np.random.RandomState(seed)
    vid = r.normal(0, 1, (B, 1, T, H, H)).astype(np.float32)
    mel = r.uniform(-1, 1, (B, 1, 80, 4 * T)).astype(np.float32)
    spec = r.uniform(0, 1, (B, 1, 321, 4 * T)).astype(np.float32)
    noise = r.normal(0, 1, (B, 128, 20, T)).astype(np.float32)
    if lens is None:
        lens = [T] * B
    return dict(vid=torch.from_numpy(vid), mel=torch.from_numpy(mel), spec=torch.from_numpy(spec),
                noise=torch.from_numpy(noise), vid_len=torch.tensor(lens, dtype=torch.int64))


def set_dropout(v_front, p):
    """Dropout draws depend on the RNG stream; parity cases run with p=0 (visual_front.py:18,20)."""
    v_front.dropout.p = p
    v_front.sentence_encoder.dropout = p
    return v_front


def sample_idx(numel, k, name):
    r = _rs(7, name)
    return np.sort(r.randint(0, numel, size=min(k, numel)))


def fill_models(models, seed):
    """Seed every module of a build_models()-style dict (sorted key order) and switch dropout off."""
    for i, k in enumerate(sorted(models)):
        det_fill(models[k], seed + i)
    set_dropout(models['v_front'], 0.0)
    return models
