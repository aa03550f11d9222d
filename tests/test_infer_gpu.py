"""Inference path with flip test-time augmentation (vcagan_amd.infer, mirror of test.py:127-143) against the CPU
oracle modules run the reference's way (two passes, averaged), and checkpoint interop in the reference's format."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import cpu_ref, fixtures

gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize('prec,atol', [('fp32', 2e-4), ('bf16x3', 3e-3)])
def test_tta_synthesis_matches_oracle_and_checkpoint_roundtrip(prec, atol):
    from vcagan_amd import audio, infer, ops, train_step
    ops.set_precision(prec)
    try:
        B, T, H = 2, 20, 112
        ref = fixtures.fill_models(cpu_ref.build_models(), 11)
        for m in ref.values():
            m.eval()
        b = fixtures.make_batch(B, T, H, 21, [20, 13])
        noise2 = torch.randn(B, 128, 20, T, generator=torch.Generator().manual_seed(5))
        with torch.no_grad():
            phon, sent = ref['v_front'](b['vid'])
            _, _, g3a = ref['gen'](sent, phon, b['vid_len'], noise=b['noise'])
            phon, sent = ref['v_front'](b['vid'].flip(4))
            _, _, g3b = ref['gen'](sent, phon, b['vid_len'], noise=noise2)
            g3r = (g3a + g3b) / 2.0
            gsr = ref['post'](g3r)
        # the reference's checkpoint layout (train.py:303-309), written from the oracle's weights
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, 'ck.ckpt')
            torch.save({k + '_state_dict': ref[k].state_dict() for k in ref}, path)
            models = train_step.build_models(device=torch.device('cuda', 0))
            infer.load_checkpoint(models, path)
        for m in models.values():
            m.eval()
        g3, gs = infer.synthesize_mel(models, b['vid'].cuda(), b['vid_len'], True, (b['noise'].cuda(), noise2.cuda()))
        assert (g3.cpu() - g3r).abs().max().item() < atol, (g3.cpu() - g3r).abs().max().item()
        assert (g3.cpu() - g3r).abs().mean().item() < 1e-3          # the mel-L1 gate
        assert (gs.cpu() - gsr).abs().max().item() < 10 * atol
        # without TTA = the plain eval forward
        g3p, _ = infer.synthesize_mel(models, b['vid'].cuda(), b['vid_len'], False, b['noise'].cuda())
        assert (g3p.cpu() - g3a).abs().max().item() < atol
        # waveform end of the path: shapes, range, finiteness (Griffin-Lim itself is covered by test_audio_gpu)
        stft = audio.TacotronSTFT(640, 160, 640, 80, 16000, 55.0, 7500.0).cuda()
        ang = torch.rand(B, 321, 4 * T, generator=torch.Generator().manual_seed(6)) * 6.28 - 3.14
        _, _, wav = infer.synthesize(models, b['vid'].cuda(), b['vid_len'], stft, None, True,
                                     (b['noise'].cuda(), noise2.cuda()), 4, ang)
        assert tuple(wav.shape) == (B, 160 * (4 * T - 1)) and torch.isfinite(wav).all() and wav.abs().max() <= 1.0
    finally:
        ops.set_precision('fp32')
