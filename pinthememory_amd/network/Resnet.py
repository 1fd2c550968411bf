"""ResNet-50/101 trunk whose forward runs on the HIP kernels. Mirrors the class surface and state_dict layout of
/root/reference/network/Resnet.py (Bottleneck :137-216 with its [x, w_arr] list protocol, ResNet :395-495,
resnet50/resnet101 :527-559); only the whitening-free (iw = 0) path exists."""
import torch
import torch.nn as nn

from . import mynn
from ..hip import ops

__all__ = ['ResNet', 'Bottleneck', 'resnet50', 'resnet101']


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, iw=0):
        super().__init__()
        assert iw == 0, 'instance whitening is out of scope of the pinmem hot path'
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = mynn.Norm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = mynn.Norm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = mynn.Norm2d(planes * self.expansion)
        self.downsample = downsample
        self.stride = stride
        self.iw = iw
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x_tuple):
        if len(x_tuple) != 2:
            print("error!!!")
            return
        x, w_arr = x_tuple
        if self.training and torch.is_grad_enabled() and all(c.bias is None for c in (self.conv1, self.conv2, self.conv3)):
            return [ops.bottleneck(x, self), w_arr]
        out = ops.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        out = ops.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        residual = x if self.downsample is None else ops.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        out = ops.conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=residual)   # out += residual; relu
        return [out, w_arr]


class ResNet(nn.Module):
    def __init__(self, block, layers, wt_layer=None, num_classes=1000):
        self.inplanes = 64
        super().__init__()
        wt_layer = [0] * 7 if wt_layer is None else wt_layer
        assert all(v == 0 for v in wt_layer), 'whitening layers are out of scope'
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = mynn.Norm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        self.wt_layer = wt_layer
        for m in self.modules():                       # Resnet.py:441-448
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                       mynn.Norm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, iw=0)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, iw=0))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = stem(self.conv1, self.bn1, x)
        x_tuple = self.layer4(self.layer3(self.layer2(self.layer1([x, []]))))
        return x_tuple[0]


def stem(conv1, bn1, x):
    """layer0: 7x7 s2 conv (input padded 3 -> 4 channels so every gathered row is one 16 B vector) + BN + ReLU + maxpool
    (Resnet.py:404-405,432,471-478)."""
    import torch.nn.functional as F
    from ..hip import kernels as K
    x4 = ops.nchw(K.nchw_to_nhwc(x.float(), c_pad=4)) if x.shape[1] == 3 else x
    w4 = F.pad(conv1.weight, (0, 0, 0, 0, 0, 4 - conv1.weight.shape[1])) if conv1.weight.shape[1] == 3 else conv1.weight
    y = ops._ConvBnAct.apply(x4, w4, None, bn1.weight, bn1.bias, None, ops._geom(conv1), ops.BNState(bn1), True, None)
    return ops.maxpool3x3s2(y)


def _pretrained(model, name):
    # The reference downloads ImageNet weights here (Resnet.py:535-539). There is no network on the build/bench
    # boxes; callers load checkpoints through forgiving_state_restore instead.
    return model


def resnet50(pretrained=True, wt_layer=None, **kwargs):
    return _pretrained(ResNet(Bottleneck, [3, 4, 6, 3], wt_layer=wt_layer, **kwargs), 'resnet50')


def resnet101(pretrained=True, wt_layer=None, **kwargs):
    return _pretrained(ResNet(Bottleneck, [3, 4, 23, 3], wt_layer=wt_layer, **kwargs), 'resnet101')
