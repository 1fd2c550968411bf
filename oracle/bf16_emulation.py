"""Oracle tooling (test infrastructure; only tests/ may import it): the CPU oracle with the STORAGE ROUNDING of the bf16 tier emulated.

The bf16 tier (BASELINE configs[2], pinthememory_amd/csrc/act16.hip + conv16.hip) keeps every activation and activation gradient between layers as bf16 (8 mantissa
bits, round to nearest even), rounds the convolution operands (activations and weights) to bf16 and does all arithmetic in fp32. What that storage costs in accuracy is
a property of the NETWORK (ReLU units within 2^-9 of zero take the other branch, BatchNorm divides by a standard deviation that can be far smaller than the rounded
mean), not of the kernels -- so the tier's assembled gradients are judged against THIS emulation: the same stock-torch oracle in float64 arithmetic with a rounding to
bf16 inserted at every point where the tier stores bf16 (value in the forward pass, gradient in the backward pass):

  * nn.Conv2d: input and weight rounded as operands (value only: the weight gradient itself stays fp32 on the tier), output rounded (value + gradient) -- except the class
    heads `final2.0` and `dsn.4`, whose logits stay fp32;
  * nn.ReLU output (the tier's fused BatchNorm + residual + ReLU epilogue writes bf16 once, after the ReLU), the BatchNorm of a down-sample branch (no ReLU behind it),
    nn.AdaptiveAvgPool2d, and the bilinear up-sampling of feature maps (not the final logits, which the tier's fused loss never materialises).

It follows /root/reference/network/deepv3plus.py:485-630 / Resnet.py:181-216 / memory.py exactly as oracle/ref_cpu does -- it IS that model, hooked."""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F


class _RoundValue(torch.autograd.Function):
    """bf16 rounding of the value, identity for the gradient (operand rounding: the tier rounds the copy it multiplies with, not the gradient that flows back)."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _round_both(t):
    """bf16 rounding of the value AND of the gradient that flows back through this point (a tensor the tier stores as bf16 in both passes)."""
    return t.to(torch.bfloat16).to(t.dtype)


@contextlib.contextmanager
def bf16_tier(net, deeplab_module):
    """Within the block `net` (an oracle.ref_cpu model) computes like the bf16 tier stores. deeplab_module: oracle.ref_cpu.deeplab (its `upsample` is wrapped)."""
    handles, restored = [], []
    heads = {'final2.0', 'dsn.4'}
    for name, m in net.named_modules():
        if isinstance(m, nn.Conv2d):
            def fwd(x, m=m, keep=name in heads):
                y = F.conv2d(_RoundValue.apply(x), _RoundValue.apply(m.weight), m.bias, m.stride, m.padding, m.dilation, m.groups)
                return y if keep else _round_both(y)
            restored.append((m, m.__dict__.get('forward')))
            m.forward = fwd
        elif isinstance(m, (nn.ReLU, nn.AdaptiveAvgPool2d)) or (isinstance(m, nn.BatchNorm2d) and name.endswith('downsample.1')):
            handles.append(m.register_forward_hook(lambda mod, inp, out: _round_both(out)))
    orig_up = deeplab_module.upsample

    def up(x, size):
        y = orig_up(x, size)
        return y if x.shape[1] == 19 else _round_both(y)      # the class logits are up-sampled inside the fused fp32 loss
    deeplab_module.upsample = up
    try:
        yield net
    finally:
        deeplab_module.upsample = orig_up
        for h in handles:
            h.remove()
        for m, f in restored:
            if f is None:
                del m.forward
            else:
                m.forward = f
