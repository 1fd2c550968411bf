"""GPU debug: per-tensor gradient of the bf16 tier against the fp64 oracle and the fp32 HIP path: relative error, projection coefficient <g16, g64> / <g64, g64> and cosine.
A coefficient near 1 with a cosine near 1 = noise only; a coefficient of 0.5 / 2 = a scale error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_cpu import deeplab as o_deeplab, harness as o_h
from pinthememory_amd import harness as h, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
bs, size = int(os.environ.get('BS', 2)), int(os.environ.get('SIZE', 128))
x, y = synth.make_batch(bs, size)


def oracle(dtype):
    net = synth.load_det_weights(o_deeplab.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)).to(dtype)
    net.memory.m_items = net.memory.m_items.to(dtype)
    net.dsn[3].p = 0.0
    net.train()
    out = net(x.to(dtype), gts=y, aux_gts=y, memory_writing=True, writing_detach=False)
    o_h.total_loss(out).backward()
    return {k: v.grad.detach().double() for k, v in net.named_parameters()}


def hip(prec):
    K.set_conv_precision(prec)
    try:
        net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)).cuda()
        net.dsn[3].p = 0.0
        net.train()
        out = net(x.cuda(), gts=y.cuda(), aux_gts=y.cuda(), memory_writing=True, writing_detach=False)
        h.total_loss(out).backward()
        torch.cuda.synchronize()
        return {k: v.grad.detach().double().cpu() for k, v in net.named_parameters()}
    finally:
        K.set_conv_precision('f32')


g64, g32, g16 = oracle(torch.float64), hip('f32'), hip('bf16')
variants = {'bf16': g16}
for name in os.environ.get('EXTRA', 'bf16_operands').split(','):
    if name:
        variants[name] = hip(name)
print('%-40s %10s | %s' % ('tensor', '|g64|', ' | '.join('%-26s' % (n + ': relerr coef cos') for n in ['f32'] + list(variants))))
for k, t in g64.items():
    if t.norm().item() < 1e-7:
        continue
    row = []
    for g in [g32] + list(variants.values()):
        a = g[k]
        rel = (a - t).norm().item() / t.norm().item()
        coef = (a * t).sum().item() / (t * t).sum().item()
        cos = (a * t).sum().item() / (a.norm().item() * t.norm().item() + 1e-300)
        row.append('%8.2e %6.3f %7.4f   ' % (rel, coef, cos))
    print('%-40s %10.3e | %s' % (k, t.norm().item(), ' | '.join(row)))
