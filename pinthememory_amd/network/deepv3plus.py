"""DeepLabV3+ (ResNet-50, output stride 16 / 8) with the categorical memory, forward+backward on the HIP kernels.
Drop-in for /root/reference/network/deepv3plus.py on the pinmem path: same class / attribute / parameter names
(`layer0..4`, `aspp.features`, `aspp.img_pooling`, `aspp.img_conv`, `bot_fine`, `bot_aspp`, `final1`, `final2`, `dsn`,
`memory`), same forward signature and return lists (:485-630), same factories (:647-661)."""
import torch
import torch.nn as nn

from . import Resnet, memory
from .mynn import Norm2d, Upsample, channels_last_weights, initialize_weights
from ..hip import kernels as K
from ..hip import ops


def _cbr(cin, cout, k, **kw):
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=k, bias=False, **kw), Norm2d(cout), nn.ReLU(inplace=True))


def run_cbr(seq, x, out=None, residual=None):
    """Sequential(Conv2d, Norm2d, ReLU) as ONE fused launch sequence."""
    return ops.conv_bn_act(x, seq[0], seq[1], relu=True, out=out, residual=residual)


def fused_ce_ok(criterion):
    return (isinstance(criterion, nn.CrossEntropyLoss) and criterion.ignore_index == 255 and criterion.reduction == 'mean'
            and criterion.weight is None and getattr(criterion, 'label_smoothing', 0.0) == 0.0)


def segmentation_loss(criterion, logits, labels, size=None):
    """criterion(Upsample(logits, size), labels) (deepv3plus.py:575-578). For the reference's own criterion
    (loss.py:38-39: CrossEntropyLoss(mean, ignore_index=255)) the up-sampled logits are never materialised."""
    if fused_ce_ok(criterion):
        return ops.upsample_ce(logits, labels)
    full = logits if size is None else Upsample(logits, size)
    return criterion(full, labels)


class _AtrousSpatialPyramidPoolingModule(nn.Module):
    """deepv3plus.py:40-101; the five branches write straight into their channel slice of the 1280-wide buffer."""

    def __init__(self, in_dim, reduction_dim=256, output_stride=16, rates=(6, 12, 18)):
        super().__init__()
        if output_stride == 8:
            rates = [2 * r for r in rates]
        elif output_stride == 4:
            rates = [4 * r for r in rates]
        elif output_stride == 32:
            rates = [r // 2 for r in rates]
        elif output_stride != 16:
            raise ValueError('output stride of {} not supported'.format(output_stride))
        feats = [_cbr(in_dim, reduction_dim, 1)]
        feats += [_cbr(in_dim, reduction_dim, 3, dilation=r, padding=r) for r in rates]
        self.features = nn.ModuleList(feats)
        self.img_pooling = nn.AdaptiveAvgPool2d(1)
        self.img_conv = _cbr(in_dim, 256, 1)

    def forward(self, x):
        widths = [self.img_conv[0].out_channels] + [f[0].out_channels for f in self.features]
        buf = ops.concat_buffer(x, widths, x.shape[2:])
        xs = ops.fanout(x, 1 + len(self.features))     # one alias per branch: their gradients are summed in a single pass
        if isinstance(self.img_pooling, nn.AdaptiveAvgPool2d) and self.img_pooling.output_size in (1, (1, 1)):
            pooled = ops.global_avgpool(xs[0])
        else:                                          # callers may swap the pooling module (eval.py:744-745)
            pooled = self.img_pooling(xs[0])
        outs, off = [None], widths[0]
        for wd in widths[1:]:
            outs.append(buf[:, off:off + wd])
            off += wd
        # the five conv -> BatchNorm -> ReLU branches are independent: under SyncBatchNorm their statistics (and gradient sums) travel as ONE exchange per direction
        # (ops.conv_bn_act_n; the per-branch path of rounds 1-5 with one rank)
        res = ops.conv_bn_act_n([pooled] + list(xs[1:]), [self.img_conv] + list(self.features), outs)
        parts = [ops.resize(res[0], x.shape[2:], out=buf[:, :widths[0]])] + res[1:]
        return ops.concat(buf, parts)


class _Base(nn.Module):
    def _adopt_trunk(self, trunk):
        # the BN layers' num_batches_tracked counters are bumped in one multi-tensor launch when forward() returns
        self.register_forward_hook(lambda mod, inp, out: ops.flush_bn_counters())
        if trunk == 'resnet-50':
            resnet = Resnet.resnet50(wt_layer=self.args.wt_layer)
        elif trunk == 'resnet-101':
            resnet = Resnet.resnet101(pretrained=True, wt_layer=self.args.wt_layer)
        else:
            raise ValueError("Not a valid network arch")
        resnet.layer0 = nn.Sequential(resnet.conv1, resnet.bn1, resnet.relu, resnet.maxpool)
        self.layer0 = resnet.layer0
        self.layer1, self.layer2, self.layer3, self.layer4 = resnet.layer1, resnet.layer2, resnet.layer3, resnet.layer4

    def _trunk(self, x):
        # per stage: weight proxies first (ops.defer_weights), so each stage's weight gradients are handed to autograd
        # once that stage's backward has been launched -- the wgrad kernels run on a side stream meanwhile
        train = self.training
        ops.begin_forward()
        x = Resnet.stem(self.layer0[0], self.layer0[1], x)      # layer0[0..3]: conv, bn, relu, maxpool
        x_tuple = [x, []]
        outs = []
        layers = [self.layer1, self.layer2, self.layer3, self.layer4]
        heads = [getattr(self, n) for n in ('aspp', 'bot_aspp', 'bot_fine', 'final1', 'memory') if hasattr(self, n)]
        # proxies of stage i+1 are created before stage i runs: their gradients are handed over one stage AFTER they were
        # computed, so the main stream never idles on a just-launched wgrad
        ahead = [[layers[0], layers[1]], [layers[2]], [layers[3]], heads]
        for layer, nxt in zip(layers, ahead):
            if train:
                for m in nxt:
                    ops.defer_weights(m)
            x_tuple = layer(x_tuple)
            outs.append(x_tuple[0])
        return outs[0], outs[2], outs[3]

    def _make_memory(self):
        if self.args.memory:
            assert self.args.mem_slot == 19, 'memory.py:336 hard-codes 19 slots'
            self.memory = memory.Memory_sup(memory_size=self.args.mem_slot, input_feature_dim=self.args.mem_dim, feature_dim=self.args.mem_dim,
                                            momentum=self.args.mem_momentum, temperature=self.args.mem_temp, gumbel_read=(not self.args.gumbel_off))

    def _aux_loss(self, aux_out, gts, aux_gts):       # deepv3plus.py:589-595
        a = ops.conv_bn_act(aux_out, self.dsn[0], self.dsn[1], relu=True)
        if self.dsn[3].p > 0:
            a = self.dsn[3](a)
        a = ops.conv(a, self.dsn[4])
        if aux_gts.dim() == 1:
            aux_gts = gts
        small = K.label_nearest(aux_gts, a.shape[2:])
        return segmentation_loss(self.criterion_aux, a, small)

    def _finish(self, dec2, x_size, aux_out, gts, aux_gts, mem_output, writeloss, readloss, inter_feature):
        if self.training:                              # deepv3plus.py:577-615
            loss1 = segmentation_loss(self.criterion, dec2, gts, x_size[2:])
            return_loss = [loss1, self._aux_loss(aux_out, gts, aux_gts)]
            if self.args.memory:
                return_loss += [mem_output, writeloss, readloss]
            return_loss.append(inter_feature)
            return return_loss
        outputs = [Upsample(dec2, x_size[2:])]         # deepv3plus.py:616-630
        if self.args.memory:
            outputs.append(mem_output)
        outputs.append(inter_feature)
        return outputs

    def _run_memory(self, dec0_up, gts, memory_writing, writing_detach):
        dec0_up, sq, sm, readloss, writeloss = self.memory(dec0_up, gts, memory_writing, writing_detach)
        return dec0_up, [sq, sm, dec0_up.detach()], readloss, writeloss


class DeepV3Plus(_Base):
    def __init__(self, num_classes, trunk='resnet-101', criterion=None, criterion_aux=None, variant='D', skip='m1', skip_num=48, args=None):
        super().__init__()
        self.criterion, self.criterion_aux, self.variant, self.args, self.trunk = criterion, criterion_aux, variant, args, trunk
        assert all(v == 0 for v in args.wt_layer) and not args.use_wtloss, 'whitening is out of scope of the pinmem hot path'
        if trunk != 'resnet-50':
            raise ValueError("Not a valid network arch")   # other trunks (and the reference's broken R101-V3+) are out of scope
        self._adopt_trunk(trunk)

        def dilate(layer, d):
            for n, m in layer.named_modules():
                if 'conv2' in n:
                    m.dilation, m.padding, m.stride = (d, d), (d, d), (1, 1)
                elif 'downsample.0' in n:
                    m.stride = (1, 1)
        if variant == 'D':
            dilate(self.layer3, 2), dilate(self.layer4, 4)
            os = 8
        elif variant == 'D16':
            dilate(self.layer4, 2)
            os = 16
        else:
            raise ValueError('unknown deepv3 variant: {}'.format(variant))
        self.output_stride = os
        self.aspp = _AtrousSpatialPyramidPoolingModule(2048, 256, output_stride=os)
        self.bot_fine = _cbr(256, 48, 1)
        self.bot_aspp = _cbr(1280, 256, 1)
        self.final1 = nn.Sequential(*(list(_cbr(304, 256, 3, padding=1)) + list(_cbr(256, 256, 3, padding=1))))
        self.final2 = nn.Sequential(nn.Conv2d(256, num_classes, kernel_size=1, bias=True))
        self.dsn = nn.Sequential(nn.Conv2d(1024, 512, kernel_size=3, stride=1, padding=1), Norm2d(512), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
                                 nn.Conv2d(512, num_classes, kernel_size=1, stride=1, padding=0, bias=True))
        initialize_weights(self.dsn)
        initialize_weights(self.aspp)
        initialize_weights(self.bot_aspp)
        initialize_weights(self.bot_fine)
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
        low_level, aux_out, x = self._trunk(x)
        x = self.aspp(x)
        dec0_up = run_cbr(self.bot_aspp, x)
        inter_feature = dec0_up
        mem_output = writeloss = readloss = None
        if self.args.memory:
            dec0_up, mem_output, readloss, writeloss = self._run_memory(dec0_up, gts, memory_writing, writing_detach)
        buf = ops.concat_buffer(low_level, [48, 256], low_level.shape[2:])
        dec0_fine = run_cbr(self.bot_fine, low_level, out=buf[:, :48])
        dec0_up = ops.resize(dec0_up, low_level.size()[2:], out=buf[:, 48:])
        dec0 = ops.concat(buf, [dec0_fine, dec0_up])
        dec1 = ops.conv_bn_act(dec0, self.final1[0], self.final1[1], relu=True)
        dec1 = ops.conv_bn_act(dec1, self.final1[3], self.final1[4], relu=True)
        dec2 = ops.conv(dec1, self.final2[0])
        return self._finish(dec2, x_size, aux_out, gts, aux_gts, mem_output, writeloss, readloss, inter_feature)


def DeepR50V3PlusD_OS8(args, num_classes, criterion, criterion_aux):
    print("Model : DeepLabv3+, Backbone : ResNet-50")
    return DeepV3Plus(num_classes, trunk='resnet-50', criterion=criterion, criterion_aux=criterion_aux, variant='D', skip='m1', args=args)


def DeepR50V3PlusD(args, num_classes, criterion, criterion_aux):
    print("Model : DeepLabv3+, Backbone : ResNet-50")
    return DeepV3Plus(num_classes, trunk='resnet-50', criterion=criterion, criterion_aux=criterion_aux, variant='D16', skip='m1', args=args)
