"""Generate tests/golden/asr_fwd.npz by running the REFERENCE's ASR scorer classes (ASR_model/GRID/src/models) on CPU
with the deterministic fixture weights.  Development-container script (needs /root/reference); only seeds and the
reference's outputs are written.   Usage: PYTHONDONTWRITEBYTECODE=1 python oracle/make_asr_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
REF = os.environ.get('VCA_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
from oracle import fixtures  # noqa: E402

sys.path.insert(0, os.path.join(REF, 'ASR_model', 'GRID'))
from src.models.audio_front import Audio_front  # noqa: E402  (the ASR package's own `src`)
from src.models.classifier import Backend  # noqa: E402

torch.set_num_threads(8)
front, back = Audio_front().eval(), Backend().eval()
fixtures.det_fill(front, 41)
fixtures.det_fill(back, 42)
B, T = 2, 120
mel = torch.rand(B, 1, 80, T, generator=torch.Generator().manual_seed(7)) * 2 - 1
with torch.no_grad():
    feat = front(mel)
    pred = back(feat)
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'asr_fwd.npz'), B=B, T=T, mel_seed=7, front_seed=41,
                    back_seed=42, feat=feat.numpy(), pred=pred.numpy())
print('asr_fwd.npz', feat.shape, pred.shape, float(pred.abs().max()))
