"""DeepLabV2 (ResNet-50/101, output stride 8, summed 4-branch dilated ASPP 6/12/18/24, no skip decoder) with the memory
hook, on the HIP kernels. Drop-in for /root/reference/network/deepv2.py (_ASPPofDeeplabv2 :40-58, DeepV2 :61-334,
factories :343-357)."""
import torch.nn as nn

from .deepv3plus import _Base, _cbr, run_cbr
from .mynn import Norm2d, channels_last_weights, initialize_weights
from ..hip import ops


class _ASPPofDeeplabv2(nn.Module):
    def __init__(self, inplanes, dilation_series=(6, 12, 18, 24), padding_series=(6, 12, 18, 24), outdim=256):
        super().__init__()
        self.conv2d_list = nn.ModuleList([_cbr(inplanes, outdim, 3, stride=1, padding=p, dilation=d)
                                          for d, p in zip(dilation_series, padding_series)])

    def forward(self, x):
        # out0 + out1 + out2 + out3 (deepv2.py:53-58): the running sum rides the residual input applied AFTER the ReLU
        out = run_cbr(self.conv2d_list[0], x)
        for f in list(self.conv2d_list)[1:]:
            out = ops.add(out, run_cbr(f, x))
        return out


class DeepV2(_Base):
    def __init__(self, num_classes, trunk='resnet-101', criterion=None, criterion_aux=None, variant='D', skip='m1', skip_num=48, args=None):
        super().__init__()
        self.criterion, self.criterion_aux, self.variant, self.args, self.trunk = criterion, criterion_aux, variant, args, trunk
        assert list(args.wt_layer) == [0] * 7, "deeplabv2 did not fit with robustnet"      # deepv2.py:183
        self._adopt_trunk(trunk)
        if variant != 'D':
            raise ValueError('unknown deepv2 variant: {}'.format(variant))
        self.layer2[0].conv1.stride = (2, 2)           # deepv2.py:122-123
        self.layer2[0].conv2.stride = (1, 1)
        for layer, d in ((self.layer3, 2), (self.layer4, 4)):
            for n, m in layer.named_modules():
                if 'conv2' in n:
                    m.dilation, m.padding, m.stride = (d, d), (d, d), (1, 1)
                elif 'downsample.0' in n:
                    m.stride = (1, 1)
        self.output_stride = 8
        self.aspp = _ASPPofDeeplabv2(2048)
        self.final1 = _cbr(256, 256, 3, padding=1)
        self.final2 = nn.Sequential(nn.Conv2d(256, num_classes, kernel_size=1, bias=True))
        self.dsn = nn.Sequential(nn.Conv2d(1024, 512, kernel_size=3, stride=1, padding=1), Norm2d(512), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
                                 nn.Conv2d(512, num_classes, kernel_size=1, stride=1, padding=0, bias=True))
        initialize_weights(self.dsn)
        initialize_weights(self.aspp)
        initialize_weights(self.final1)
        initialize_weights(self.final2)
        self.eps = 1e-5
        self.whitening = False
        self.three_input_layer = False
        self.cov_matrix_layer, self.cov_type = [], []
        self._make_memory()
        channels_last_weights(self)

    def forward(self, x, gts=None, aux_gts=None, img_gt=None, visualize=False, cal_covstat=False, apply_wtloss=True,
                memory_writing=False, writing_detach=True):
        with ops.prefold(self):      # no-grad eval forward: all BatchNorm folds in one launch (no-op otherwise)
            return self._forward(x, gts, aux_gts, img_gt, visualize, cal_covstat, apply_wtloss, memory_writing, writing_detach)

    def _forward(self, x, gts, aux_gts, img_gt, visualize, cal_covstat, apply_wtloss, memory_writing, writing_detach):
        assert not cal_covstat and not visualize, 'whitening statistics are out of scope'
        x_size = x.size()
        _, aux_out, x = self._trunk(x)
        dec0_up = self.aspp(x)
        inter_feature = dec0_up
        mem_output = writeloss = readloss = None
        if self.args.memory:
            dec0_up, mem_output, readloss, writeloss = self._run_memory(dec0_up, gts, memory_writing, writing_detach)
        dec1 = run_cbr(self.final1, dec0_up)
        dec2 = ops.conv(dec1, self.final2[0])
        return self._finish(dec2, x_size, aux_out, gts, aux_gts, mem_output, writeloss, readloss, inter_feature)


def DeepR50V2D(args, num_classes, criterion, criterion_aux):
    print("Model : DeepLabv2, Backbone : ResNet-50")
    return DeepV2(num_classes, trunk='resnet-50', criterion=criterion, criterion_aux=criterion_aux, variant='D', skip='m1', args=args)


def DeepR101V2D(args, num_classes, criterion, criterion_aux):
    print("Model : DeepLabv2, Backbone : ResNet-101")
    return DeepV2(num_classes, trunk='resnet-101', criterion=criterion, criterion_aux=criterion_aux, variant='D', skip='m1', args=args)
