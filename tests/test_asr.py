"""ASR scorer (ASR_model/GRID/src/models): oracle restatement vs the golden written from the reference classes
(CPU), and the HIP mirror vs the same golden (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import asr_ref, fixtures

gpu = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'asr_fwd.npz')


def _mel(g):
    return torch.rand(int(g['B']), 1, 80, int(g['T']), generator=torch.Generator().manual_seed(int(g['mel_seed']))) * 2 - 1


def test_oracle_matches_reference_golden():
    g = np.load(GOLD)
    front, back = asr_ref.Audio_front().eval(), asr_ref.Backend().eval()
    fixtures.det_fill(front, int(g['front_seed']))
    fixtures.det_fill(back, int(g['back_seed']))
    with torch.no_grad():
        feat = front(_mel(g))
        pred = back(feat)
    assert np.abs(feat.numpy() - g['feat']).max() < 1e-5
    assert np.abs(pred.numpy() - g['pred']).max() < 1e-5


@gpu
@pytest.mark.parametrize('prec,atol', [('fp32', 1e-4), ('bf16x3', 1e-3)])
def test_hip_scorer_matches_reference_golden(prec, atol):
    from vcagan_amd import ops
    from vcagan_amd.models import asr
    g = np.load(GOLD)
    ops.set_precision(prec)
    try:
        front, back = asr.Audio_front(), asr.Backend()
        ref_f, ref_b = asr_ref.Audio_front(), asr_ref.Backend()
        assert list(front.state_dict().keys()) == list(ref_f.state_dict().keys())
        assert list(back.state_dict().keys()) == list(ref_b.state_dict().keys())
        fixtures.det_fill(front, int(g['front_seed']))
        fixtures.det_fill(back, int(g['back_seed']))
        front.cuda().eval(); back.cuda().eval()
        with torch.no_grad():
            feat = front(_mel(g).cuda())
            pred = back(feat)
        assert np.abs(feat.cpu().numpy() - g['feat']).max() < atol
        assert np.abs(pred.cpu().numpy() - g['pred']).max() < atol
        tok = asr.ctc_greedy_tokens(pred).cpu().numpy()
        assert np.array_equal(tok, g['pred'].argmax(2)) or (tok != g['pred'].argmax(2)).mean() < 0.05
    finally:
        ops.set_precision('fp32')
