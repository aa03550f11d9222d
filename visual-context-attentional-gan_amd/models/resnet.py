"""ResNet-18 trunk of the visual front-end (mirror of the reference's src/models/resnet.py:25-123).

Feature maps are channels-last (N, H, W, C).  nn.Conv2d / nn.BatchNorm2d / nn.PReLU objects are used as
parameter holders only (identical state_dict keys, shapes and initialisation); the arithmetic runs through
vcagan_amd.ops.
"""
import math

import torch.nn as nn

from .. import ops


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def downsample_basic_block(inplanes, outplanes, stride):
    return nn.Sequential(nn.Conv2d(inplanes, outplanes, kernel_size=1, stride=stride, bias=False),
                         nn.BatchNorm2d(outplanes))


class BasicBlock(nn.Module):
    """resnet.py:25-66."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type='relu'):
        super().__init__()
        assert relu_type in ('relu', 'prelu')
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        if relu_type == 'relu':
            self.relu1, self.relu2 = nn.ReLU(inplace=True), nn.ReLU(inplace=True)
        else:
            self.relu1, self.relu2 = nn.PReLU(num_parameters=planes), nn.PReLU(num_parameters=planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # x also feeds the shortcut: it is taken from the convolution's alias output so that the shortcut's gradient
        # is added inside conv1's input-gradient kernel
        y, st, x = ops.conv2d_skip(x, self.conv1.weight, self.stride, 1)
        y = ops.bn_act(y, self.bn1, self.relu1, stats=st)
        y, st = ops.conv2d(y, self.conv2.weight, None, 1, 1, want_stats=True)
        if self.downsample is not None:
            sc, st2 = ops.conv2d(x, self.downsample[0].weight, None, self.downsample[0].stride[0], 0, want_stats=True)
            sc = ops.bn_act(sc, self.downsample[1], None, stats=st2)
        else:
            sc = x
        return ops.bn_act(y, self.bn2, self.relu2, residual=sc, stats=st)


class ResNet(nn.Module):
    """resnet.py:69-123 (BasicBlock, [2,2,2,2]); forward takes and returns channels-last tensors."""

    def __init__(self, block, layers, num_classes=1000, relu_type='relu', gamma_zero=False,
                 avg_pool_downsample=False):
        super().__init__()
        assert not avg_pool_downsample, 'the hot path only builds the 1x1-conv shortcut (resnet.py:10-14)'
        self.inplanes = 64
        self.relu_type = relu_type
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if gamma_zero:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    m.bn2.weight.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = downsample_basic_block(self.inplanes, planes * block.expansion, stride)
        layers = [block(self.inplanes, planes, stride, downsample, relu_type=self.relu_type)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, relu_type=self.relu_type) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):                      # (N, H, W, 64)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return ops.head_pool4(x)               # (N, 512)
