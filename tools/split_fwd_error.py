"""GPU: forward error of the HIP train-mode network vs the fp64 CPU oracle per stage, split path on / off (round 6: why the assembled gradients at bs=2 128^2 move with the
forward form of the split path although every kernel is fp32-accurate): rms and MEAN SIGNED error of the stage outputs relative to their rms, and the number of ReLU
sign disagreements. usage: python tools/split_fwd_error.py [size] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_cpu import deeplab as o_deeplab
from pinthememory_amd import synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
x, y = synth.make_batch(batch, size)
names = ['layer1', 'layer2', 'layer3', 'layer4']


def run(net, xx, yy):
    feats = {}
    hs = []
    for n in names:
        mod = getattr(net, n)
        hs.append(mod.register_forward_hook(lambda m, i, o, n=n: feats.__setitem__(n, (o[0] if isinstance(o, (list, tuple)) else o).detach().double().cpu())))
    net.train()
    net.dsn[3].p = 0.0
    with torch.no_grad():
        net(xx, gts=yy, aux_gts=yy, memory_writing=True, writing_detach=True)
    for h in hs:
        h.remove()
    return feats


ref = synth.load_det_weights(o_deeplab.DeepR50V3PlusD(synth.model_args(), 19, crit, crit)).double()
ref.memory.m_items = ref.memory.m_items.double()
truth = run(ref, x.double(), y)
ref32 = synth.load_det_weights(o_deeplab.DeepR50V3PlusD(synth.model_args(), 19, crit, crit))
o32 = run(ref32, x, y)
res = {'fp32 oracle': o32}
for split in (False, True):
    K.set_split(split)
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, crit, crit)).cuda()
    res['hip split=%d' % split] = run(net, x.cuda(), y.cuda())
K.set_split(True)
for n in names:
    t = truth[n]
    rms = t.pow(2).mean().sqrt().item()
    line = '%-9s rms %.3e |' % (n, rms)
    for k, f in res.items():
        e = f[n] - t
        line += ' %s: rms %.2e mean %+.2e flips %d |' % (k, e.pow(2).mean().sqrt().item() / rms, e.mean().item() / rms, int(((f[n] > 0) != (t > 0)).sum()))
    print(line)
