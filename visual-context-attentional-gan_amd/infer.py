"""Inference path of the reference's test.py:127-141 / validate() (train.py:358-391): lip video -> mel-spectrogram
with horizontal-flip test-time augmentation -> linear spectrogram (Postnet) -> waveform (Griffin-Lim).

    g3 = (gen(v_front(vid)) + gen(v_front(vid.flip(4)))) / 2;  gs = post(g3);  wav = inverse_spec(gs, stft)

In eval mode every layer is per-sample (batch-norm uses its running statistics, dropout is off), so the clip and
its mirror image are run as ONE batch of 2B: same numbers as the reference's two passes, half the launches and
twice the rows per GEMM.  Checkpoints saved by the reference's train.py:303-309 load with load_checkpoint().
"""
import torch

from . import audio, ops


def load_checkpoint(models, path, map_location='cpu'):
    """train.py:95-101 / test.py:65-84: the reference stores each network's state_dict under these keys."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    for k in ('v_front', 'gen', 'post', 'dis1', 'dis2', 'dis3', 's_dis'):
        ck_key = k + '_state_dict'
        if k in models and ck_key in ck:
            sd = {n.replace('module.', '', 1) if n.startswith('module.') else n: v for n, v in ck[ck_key].items()}
            models[k].load_state_dict(sd, strict=True)
    ops._param_epoch[0] += 1          # parameters were rewritten behind autograd's back: drop packed copies
    return ck


@torch.no_grad()
def synthesize_mel(models, vid, vid_len, tta=True, noise=None):
    """vid (B,1,T,H,W) on the device -> (g3 (B,1,80,4T), gs (B,1,321,4T)).  noise: None (drawn like generator.py:248),
    a (B,128,20,T) tensor used for both passes, or a pair (plain, flipped)."""
    v_front, gen, post = models['v_front'], models['gen'], models['post']
    assert not (v_front.training or gen.training or post.training), 'call .eval() first (test.py:61-63)'
    B = vid.size(0)
    lens = ops.as_lens(vid_len, vid.device)
    if tta:
        x = torch.cat([vid, vid.flip(4)], 0)
        lens2 = torch.cat([lens, lens], 0)
        if noise is None:
            nz = None
        elif isinstance(noise, (tuple, list)):
            nz = torch.cat([noise[0], noise[1]], 0)
        else:
            nz = torch.cat([noise, noise], 0)
        phon, sent = v_front(x)
        _, _, g3 = gen(sent, phon, lens2, noise=nz)
        g3 = ops.add_scale(g3[:B], g3[B:], 0.5)             # (g3 + g3_flip) / 2, test.py:139
    else:
        phon, sent = v_front(vid)
        _, _, g3 = gen(sent, phon, lens, noise=noise if not isinstance(noise, (tuple, list)) else noise[0])
    gs = post(g3)
    return g3, gs


@torch.no_grad()
def synthesize(models, vid, vid_len, stft, mel_len=None, tta=True, noise=None, n_iters=60, angles=None):
    """test.py:127-143: returns (g3, gs, wav (B, L) device tensor in [-1, 1])."""
    g3, gs = synthesize_mel(models, vid, vid_len, tta, noise)
    spec = gs if mel_len is None else gs[:, :, :, :int(mel_len)]
    wav = audio.inverse_spec(spec.contiguous(), stft, n_iters, angles)
    return g3, gs, wav
