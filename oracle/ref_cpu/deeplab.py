"""Oracle (test infrastructure): DeepLabV3+ (ResNet-50, OS16/OS8) and DeepLabV2 (ResNet-50/101, OS8)
with the memory hook, stock torch ops on CPU.

Restates /root/reference/network/deepv3plus.py
  _AtrousSpatialPyramidPoolingModule :40-101   DeepV3Plus.__init__ :112-472 (resnet-50 branch)
  DeepV3Plus.forward :485-630                  factories :647-661
and /root/reference/network/deepv2.py
  _ASPPofDeeplabv2 :40-58   DeepV2.__init__ :61-196   DeepV2.forward :210-334   factories :343-357
and network/mynn.py: Upsample :57-62, initialize_weights :27-44.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import resnet as R
from .memory import Memory_sup


def upsample(x, size):                                # mynn.py:57-62
    return F.interpolate(x, size=size, mode='bilinear', align_corners=True)


def init_weights(*models):                            # mynn.py:27-44
    for model in models:
        for m in model.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, nonlinearity='relu')
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


def _cbr(cin, cout, k, **kw):
    return nn.Sequential(nn.Conv2d(cin, cout, k, bias=False, **kw), R.norm2d(cout), nn.ReLU(inplace=True))


class ASPP(nn.Module):                                # deepv3plus.py:40-101
    def __init__(self, in_dim, reduction_dim=256, output_stride=16, rates=(6, 12, 18)):
        super().__init__()
        if output_stride == 8:
            rates = [2 * r for r in rates]
        elif output_stride != 16:
            raise ValueError('output stride of {} not supported'.format(output_stride))
        feats = [_cbr(in_dim, reduction_dim, 1)]
        feats += [_cbr(in_dim, reduction_dim, 3, dilation=r, padding=r) for r in rates]
        self.features = nn.ModuleList(feats)
        self.img_pooling = nn.AdaptiveAvgPool2d(1)
        self.img_conv = _cbr(in_dim, 256, 1)

    def forward(self, x):
        img = upsample(self.img_conv(self.img_pooling(x)), x.shape[2:])
        return torch.cat([img] + [f(x) for f in self.features], 1)


class ASPPv2(nn.Module):                              # deepv2.py:40-58
    def __init__(self, inplanes, series=(6, 12, 18, 24), outdim=256):
        super().__init__()
        self.conv2d_list = nn.ModuleList([_cbr(inplanes, outdim, 3, stride=1, padding=d, dilation=d) for d in series])

    def forward(self, x):
        out = self.conv2d_list[0](x)
        for f in list(self.conv2d_list)[1:]:
            out = out + f(x)
        return out


def _dilate(layer, d):
    for n, m in layer.named_modules():
        if 'conv2' in n:
            m.dilation, m.padding, m.stride = (d, d), (d, d), (1, 1)
        elif 'downsample.0' in n:
            m.stride = (1, 1)


def _dsn(cin, num_classes):                           # deepv3plus.py:419-425 / deepv2.py:145-151
    return nn.Sequential(nn.Conv2d(cin, 512, 3, stride=1, padding=1), R.norm2d(512), nn.ReLU(inplace=True),
                         nn.Dropout2d(0.1), nn.Conv2d(512, num_classes, 1, bias=True))


class _Base(nn.Module):
    def _adopt_trunk(self, trunk):
        net = {'resnet-50': R.resnet50, 'resnet-101': R.resnet101}[trunk]()
        self.layer0 = nn.Sequential(net.conv1, net.bn1, net.relu, net.maxpool)
        self.layer1, self.layer2, self.layer3, self.layer4 = net.layer1, net.layer2, net.layer3, net.layer4

    def _trunk(self, x):
        x = self.layer0(x)
        low = self.layer1(x)
        aux = self.layer3(self.layer2(low))
        return low, aux, self.layer4(aux)

    def _make_memory(self, args):
        if args.memory:
            assert args.mem_slot == 19                # memory.py:336 hard-codes 19
            self.memory = Memory_sup(args.mem_slot, args.mem_dim, args.mem_dim, args.mem_momentum,
                                     args.mem_temp, gumbel_read=(not args.gumbel_off))

    def _tail(self, main_out, aux_out, gts, aux_gts, mem_output, writeloss, readloss, inter):
        # deepv3plus.py:577-630
        if self.training:
            loss1 = self.criterion(main_out, gts)
            aux_out = self.dsn(aux_out)
            if aux_gts.dim() == 1:
                aux_gts = gts
            a = F.interpolate(aux_gts.unsqueeze(1).float(), size=aux_out.shape[2:], mode='nearest').squeeze(1).long()
            loss2 = self.criterion_aux(aux_out, a)
            out = [loss1, loss2]
            if self.args.memory:
                out += [mem_output, writeloss, readloss]
            return out + [inter]
        out = [main_out]
        if self.args.memory:
            out.append(mem_output)
        return out + [inter]


class DeepV3Plus(_Base):
    def __init__(self, num_classes, trunk='resnet-50', criterion=None, criterion_aux=None, variant='D16', args=None):
        super().__init__()
        self.criterion, self.criterion_aux, self.variant, self.args, self.trunk = criterion, criterion_aux, variant, args, trunk
        assert trunk == 'resnet-50' and all(v == 0 for v in args.wt_layer)
        self._adopt_trunk(trunk)
        if variant == 'D':                            # deepv3plus.py:347-357
            _dilate(self.layer3, 2)
            _dilate(self.layer4, 4)
            os = 8
        elif variant == 'D16':                        # deepv3plus.py:374-379
            _dilate(self.layer4, 2)
            os = 16
        else:
            raise ValueError(variant)
        self.output_stride = os
        self.aspp = ASPP(2048, 256, output_stride=os)
        self.bot_fine = _cbr(256, 48, 1)
        self.bot_aspp = _cbr(1280, 256, 1)
        self.final1 = nn.Sequential(*(list(_cbr(304, 256, 3, padding=1)) + list(_cbr(256, 256, 3, padding=1))))
        self.final2 = nn.Sequential(nn.Conv2d(256, num_classes, 1, bias=True))
        self.dsn = _dsn(1024, num_classes)
        init_weights(self.dsn)
        init_weights(self.aspp, self.bot_aspp, self.bot_fine, self.final1, self.final2)
        self._make_memory(args)

    def forward(self, x, gts=None, aux_gts=None, img_gt=None, visualize=False, cal_covstat=False,
                apply_wtloss=True, memory_writing=False, writing_detach=True):
        size = x.shape[2:]
        low, aux_out, x = self._trunk(x)
        dec0_up = self.bot_aspp(self.aspp(x))
        inter = dec0_up.clone()
        mem_output = writeloss = readloss = None
        if self.args.memory:
            dec0_up, sq, sm, readloss, writeloss = self.memory(dec0_up, gts, memory_writing, writing_detach)
            mem_output = [sq, sm, dec0_up.clone().detach()]
        dec0 = torch.cat([self.bot_fine(low), upsample(dec0_up, low.shape[2:])], 1)
        main_out = upsample(self.final2(self.final1(dec0)), size)
        return self._tail(main_out, aux_out, gts, aux_gts, mem_output, writeloss, readloss, inter)


class DeepV2(_Base):
    def __init__(self, num_classes, trunk='resnet-101', criterion=None, criterion_aux=None, variant='D', args=None):
        super().__init__()
        self.criterion, self.criterion_aux, self.variant, self.args, self.trunk = criterion, criterion_aux, variant, args, trunk
        assert variant == 'D' and all(v == 0 for v in args.wt_layer)
        self._adopt_trunk(trunk)
        self.layer2[0].conv1.stride = (2, 2)          # deepv2.py:122-123 (caffe-style stride placement)
        self.layer2[0].conv2.stride = (1, 1)
        _dilate(self.layer3, 2)
        _dilate(self.layer4, 4)
        self.output_stride = 8
        self.aspp = ASPPv2(2048)
        self.final1 = _cbr(256, 256, 3, padding=1)
        self.final2 = nn.Sequential(nn.Conv2d(256, num_classes, 1, bias=True))
        self.dsn = _dsn(1024, num_classes)
        init_weights(self.dsn)
        init_weights(self.aspp, self.final1, self.final2)
        self._make_memory(args)

    def forward(self, x, gts=None, aux_gts=None, img_gt=None, visualize=False, cal_covstat=False,
                apply_wtloss=True, memory_writing=False, writing_detach=True):
        size = x.shape[2:]
        low, aux_out, x = self._trunk(x)
        dec0_up = self.aspp(x)
        inter = dec0_up.clone()
        mem_output = writeloss = readloss = None
        if self.args.memory:
            dec0_up, sq, sm, readloss, writeloss = self.memory(dec0_up, gts, memory_writing, writing_detach)
            mem_output = [sq, sm, dec0_up.clone().detach()]
        main_out = upsample(self.final2(self.final1(dec0_up)), size)
        return self._tail(main_out, aux_out, gts, aux_gts, mem_output, writeloss, readloss, inter)


def DeepR50V3PlusD(args, num_classes, criterion, criterion_aux):
    return DeepV3Plus(num_classes, 'resnet-50', criterion, criterion_aux, 'D16', args)


def DeepR50V3PlusD_OS8(args, num_classes, criterion, criterion_aux):
    return DeepV3Plus(num_classes, 'resnet-50', criterion, criterion_aux, 'D', args)


def DeepR50V2D(args, num_classes, criterion, criterion_aux):
    return DeepV2(num_classes, 'resnet-50', criterion, criterion_aux, 'D', args)


def DeepR101V2D(args, num_classes, criterion, criterion_aux):
    return DeepV2(num_classes, 'resnet-101', criterion, criterion_aux, 'D', args)
