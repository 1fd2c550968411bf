"""GPU: per-tensor gradient error of the HIP train forward + backward vs the fp64 CPU oracle with the split path on / off (round 6), per loss term. For each tensor:
relative error, and the projection coefficient <g, t> / <t, t> (a common scale error shows as coefficient != 1 with small residual). usage: split_grad_probe.py [size] [which]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_cpu import deeplab as o_deeplab, harness as o_h
from pinthememory_amd import harness as h, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
args = synth.model_args()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
which = sys.argv[2] if len(sys.argv) > 2 else 'all'
x, y = synth.make_batch(2, size)


def pick(o):
    return dict(all=None, loss1=o[0], loss2=o[1], read=o[-2], div=o[-3][0], cls=o[-3][1])[which]


ref = synth.load_det_weights(o_deeplab.DeepR50V3PlusD(args, 19, CRIT, CRIT)).double()
ref.memory.m_items = ref.memory.m_items.double()
ref.dsn[3].p = 0.0
ref.train()
out_r = ref(x.double(), gts=y, aux_gts=y, memory_writing=True, writing_detach=False)
(o_h.total_loss(out_r) if which == 'all' else pick(out_r)).backward()
truth = {k: p.grad.double() for k, p in ref.named_parameters() if p.grad is not None}
res = {}
for split in (False, True):
    K.set_split(split)
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(args, 19, CRIT, CRIT)).cuda()
    net.dsn[3].p = 0.0
    net.train()
    out_g = net(x.cuda(), gts=y.cuda(), aux_gts=y.cuda(), memory_writing=True, writing_detach=False)
    (h.total_loss(out_g) if which == 'all' else pick(out_g)).backward()
    res[split] = {k: p.grad.double().cpu() for k, p in net.named_parameters() if p.grad is not None}
K.set_split(True)
rows = []
for k, t in truth.items():
    if t.norm().item() < 1e-7 or k not in res[True]:
        continue
    r = [k]
    for split in (False, True):
        g = res[split][k]
        r += [(g - t).norm().item() / t.norm().item(), (g.flatten() @ t.flatten()).item() / (t.flatten() @ t.flatten()).item() - 1.0]
    rows.append(r)
import statistics
print('median rel err: split off %.2e, on %.2e' % (statistics.median(r[1] for r in rows), statistics.median(r[3] for r in rows)))
for r in rows:
    print('%-36s off: rel %.2e coef-1 %+.2e | on: rel %.2e coef-1 %+.2e' % tuple(r))
