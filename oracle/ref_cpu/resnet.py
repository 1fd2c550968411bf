"""Oracle (test infrastructure): ResNet-50/101 trunk, stock torch ops on CPU.

Restates /root/reference/network/Resnet.py:
  Bottleneck        :137-216  (1x1 -> BN -> ReLU -> 3x3(stride) -> BN -> ReLU -> 1x1 -> BN -> +res -> ReLU)
  ResNet.__init__   :400-448  (7x7 s2 stem, maxpool 3x3 s2, kaiming_normal fan_out, BN 1/0)
  _make_layer       :450-465
  resnet50/101      :527-559
Only the iw=0 (no whitening) path used by the pinmem scripts is restated.
Module/attribute names are kept so state_dict keys match the reference.
"""
import torch
import torch.nn as nn


def norm2d(c, bn=nn.BatchNorm2d):
    # network/mynn.py:8-14 -- cfg.MODEL.BNFUNC(c); the oracle always uses local BatchNorm2d
    return bn(c)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm2d(planes * 4)
        self.downsample = downsample
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        r = x if self.downsample is None else self.downsample(x)
        return self.relu(y + r)


class ResNet(nn.Module):
    """Constructed in the reference's order (incl. the unused fc) so that the same
    torch seed yields the same initial weights (Resnet.py:400-448)."""

    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.fc = nn.Linear(2048, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 norm2d(planes * 4))
        seq = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        seq += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)


def resnet50():
    return ResNet([3, 4, 6, 3])


def resnet101():
    return ResNet([3, 4, 23, 3])
