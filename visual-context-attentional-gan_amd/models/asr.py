"""ASR scorer that computes WER/CER on generated mels (mirror of the reference's ASR_model/GRID/src/models:
audio_front.py:5-37 Audio_front, classifier.py:3-17 Backend; same class names, constructor arguments, forward
signatures and state_dict keys, so the reference's GRID_*_wer_*.ckpt files load with strict=True).
nn modules are parameter holders; the arithmetic runs through vcagan_amd.ops on channels-last maps."""
import torch
import torch.nn as nn

from .. import ops
from .resnet import BasicBlock


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

    def forward(self, x):
        """x: (B, 1, 80, T) mel -> (B, T/4, 256)."""
        B, _, F, T = x.shape
        f = self.frontend
        h = x.reshape(B, F, T, 1)                                                        # channels-last view
        h, st = ops.conv2d(h, f[0].weight, f[0].bias, 2, 2, want_stats=True)
        h = ops.bn_act(h, f[1], f[2], stats=st)
        h, st = ops.conv2d(h, f[3].weight, f[3].bias, 2, 2, want_stats=True)
        h = ops.bn_act(h, f[4], f[5], stats=st)
        h = self.Res_block[0](h)                                                         # (B, F/4, T/4, 64)
        b, fq, tq, c = h.shape
        # the reference flattens (c, f) of an NCHW map (audio_front.py:33-34): feature index = c*F/4 + f
        h = h.permute(0, 2, 3, 1).reshape(b, tq, c * fq)
        h = ops.dropout(h, self.dropout.p, self.training)
        return ops.linear(h, self.Linear.weight, self.Linear.bias)


class Backend(nn.Module):
    def __init__(self):
        super().__init__()
        self.gru = nn.GRU(256, 256, 2, bidirectional=True, dropout=0.3)
        self.fc = nn.Linear(512, 27 + 1)

    def forward(self, x):
        """x: (B, S, 256) -> logits (B, S, 28)."""
        h = ops.gru(x, self.gru, self.training)
        return ops.linear(h, self.fc.weight, self.fc.bias)


def ctc_greedy_tokens(logits):
    """Best path of ASR_model/GRID/test.py:181-183 (topk(1) of the softmax = argmax of the logits): (B, S) token ids."""
    return torch.argmax(logits, dim=2)
