"""Visual front-end (mirror of the reference's src/models/visual_front.py:4-37)."""
import torch
import torch.nn as nn

from .. import ops
from .resnet import BasicBlock, ResNet


class Visual_front(nn.Module):
    def __init__(self, in_channels=1):
        super().__init__()
        assert in_channels == 1, 'the lip crops are single-channel (visual_front.py:11, vid_aud_grid.py:104-111)'
        self.in_channels = in_channels
        self.frontend = nn.Sequential(
            nn.Conv3d(in_channels, 64, kernel_size=(5, 7, 7), stride=(1, 2, 2), padding=(2, 3, 3), bias=False),
            nn.BatchNorm3d(64), nn.PReLU(64),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)))
        self.resnet = ResNet(BasicBlock, [2, 2, 2, 2], relu_type='prelu')
        self.dropout = nn.Dropout(0.3)
        self.sentence_encoder = nn.GRU(512, 512, 2, bidirectional=True, dropout=0.3)
        self.fc = nn.Linear(1024, 512)
        self.defer_sentence = False     # see sentence(); set by a caller that waits for the sentence features itself

    def front(self, x):
        """The 3-D front-end (visual_front.py:11-14,25-27): x (B, T, H, W, 1) channels-last clip -> (B*T, H/4, W/4, 64)
        pooled feature maps, frames flattened for free."""
        B, T = x.shape[0], x.shape[1]
        f, stats = ops.conv3d_stem(x, self.frontend[0].weight)                       # (B,T,H/2,W/2,64)
        # BatchNorm3d -> PReLU -> MaxPool3d in one pass
        return ops.bn_act_maxpool3(f.view(B * T, f.shape[2], f.shape[3], 64), self.frontend[1], self.frontend[2],
                                   stats=stats)

    def trunk(self, x):
        """x: (B, 1, T, H, W) video -> phoneme features (B, T, 512): front-end, ResNet-18, dropout
        (visual_front.py:25-31)."""
        B, _, T, H, W = x.shape
        f = self.front(x.reshape(B, T, H, W, 1))
        f = self.resnet(f)                                                           # (B*T, 512)
        return ops.dropout(f, self.dropout.p, self.training).view(B, T, 512)

    def sentence(self, f):
        """Sentence features (B, 512, T) of the phoneme features (visual_front.py:32-36)."""
        if getattr(self, 'defer_sentence', False) and f.is_cuda:
            # The recurrence is 2 x T launches of a few microseconds each, and the decoder does not read the sentence
            # features before its first attention module: on a side stream it runs beside the decoder's first blocks
            # (and its backward pass beside the second half of the decoder's).  The caller has asked for this and
            # calls ops.ensure_ready(sent) before it reads the tensor.
            cur, side = torch.cuda.current_stream(), ops.side_stream('sentence')
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                sent = self._sentence(f)
            sent._vca_stream = side
            return sent
        return self._sentence(f)

    def forward(self, x):
        """x: (B, 1, T, H, W) video.  Returns phonemes (B, T, 512) and sentence features (B, 512, T)."""
        f = self.trunk(x)
        return f, self.sentence(f)

    def _sentence(self, f):
        sent = ops.gru(f, self.sentence_encoder, self.training)                      # (B, T, 1024)
        sent = ops.linear(sent, self.fc.weight, self.fc.bias)                        # (B, T, 512)
        return sent.transpose(1, 2).contiguous()
