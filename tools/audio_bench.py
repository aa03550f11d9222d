"""Timing of the validation waveform path on the GPU: inverse_mel (60 Griffin-Lim iterations) for a batch of
3-second clips, beside the numpy restatement on the host.   python tools/audio_bench.py [batch]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vcagan_amd  # noqa
from vcagan_amd import audio
from oracle import audio_ref as A

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ts = audio.TacotronSTFT(640, 160, 640, 80, 16000, 55.0, 7500.0).cuda()
mel = torch.rand(B, 1, 80, 300, device='cuda') * 2 - 1
ang = torch.rand(B, 321, 300) * 6.28 - 3.14
for _ in range(2):
    w = audio.inverse_mel(mel, ts, 60, ang)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    w = audio.inverse_mel(mel, ts, 60, ang)
torch.cuda.synchronize()
gpu_ms = (time.perf_counter() - t0) / n * 1e3
spec = np.random.rand(2, 321, 300).astype(np.float32)
t0 = time.perf_counter()
A.griffin_lim(spec, ang[:2].numpy(), 5)
cpu_s = (time.perf_counter() - t0) / 5 * 60 / 2
print('inverse_mel B=%d x 300 frames, 60 iterations: %.1f ms on the GPU (%.2f ms per clip); numpy oracle %.1f s per clip'
      % (B, gpu_ms, gpu_ms / B, cpu_s))
