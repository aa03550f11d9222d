"""CPU restatement (plain PyTorch) of the reference's ASR scorer used to compute WER/CER on generated mels:
ASR_model/GRID/src/models/audio_front.py:5-37 (Audio_front), classifier.py:3-17 (Backend), with the BasicBlock of
ASR_model/GRID/src/models/resnet.py (same block as the main model's: oracle/cpu_ref.BasicBlock).

TEST INFRASTRUCTURE ONLY.  Pinned against the reference classes by tests/golden/asr_fwd.npz (oracle/make_asr_golden.py).
"""
import torch
import torch.nn as nn

from .cpu_ref import BasicBlock


class Audio_front(nn.Module):
    def __init__(self, in_channels=1):
        super().__init__()
        self.in_channels = in_channels
        self.frontend = nn.Sequential(
            nn.Conv2d(in_channels, 32, kernel_size=(5, 5), stride=(2, 2), padding=(2, 2)), nn.BatchNorm2d(32), nn.PReLU(32),
            nn.Conv2d(32, 64, kernel_size=(5, 5), stride=(2, 2), padding=(2, 2)), nn.BatchNorm2d(64), nn.PReLU(64))
        self.Res_block = nn.Sequential(BasicBlock(64, 64, relu_type='prelu'))
        self.Linear = nn.Linear(64 * 20, 256)
        self.dropout = nn.Dropout(0.3)

    def forward(self, x):                      # (B, 1, 80, T)
        x = self.Res_block(self.frontend(x))   # (B, 64, 20, T/4)
        b, c, f, t = x.size()
        x = x.view(b, c * f, t).transpose(1, 2).contiguous()
        return self.Linear(self.dropout(x))    # (B, T/4, 256)


class Backend(nn.Module):
    def __init__(self):
        super().__init__()
        self.gru = nn.GRU(256, 256, 2, bidirectional=True, dropout=0.3)
        self.fc = nn.Linear(512, 27 + 1)

    def forward(self, x):
        x, _ = self.gru(x.permute(1, 0, 2).contiguous())
        return self.fc(x.permute(1, 0, 2).contiguous())     # (B, S, 28)
