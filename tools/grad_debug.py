import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_cpu import deeplab as o_deeplab, harness as o_h
from pinthememory_amd import harness as h, synth
from pinthememory_amd.network import deepv3plus
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
args = synth.model_args()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
which = sys.argv[2] if len(sys.argv) > 2 else 'all'
x, y = synth.make_batch(2, size)
ref = synth.load_det_weights(o_deeplab.DeepR50V3PlusD(args, 19, CRIT, CRIT))
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(args, 19, CRIT, CRIT)).cuda()
ref.dsn[3].p = net.dsn[3].p = 0.0
ref.train(); net.train()
out_r = ref(x, gts=y, aux_gts=y, memory_writing=True, writing_detach=False)
out_g = net(x.cuda(), gts=y.cuda(), aux_gts=y.cuda(), memory_writing=True, writing_detach=False)
def pick(o):
    return dict(all=None, loss1=o[0], loss2=o[1], read=o[-2], div=o[-3][0], cls=o[-3][1])[which]
lr = o_h.total_loss(out_r) if which == 'all' else pick(out_r)
lg = h.total_loss(out_g) if which == 'all' else pick(out_g)
lr.backward(); lg.backward()
gr, gg = dict(ref.named_parameters()), dict(net.named_parameters())
for k in gr:
    if gr[k].grad is None or gg[k].grad is None:
        print('%-40s none %s %s' % (k, gr[k].grad is None, gg[k].grad is None)); continue
    a, b = gg[k].grad.cpu().double(), gr[k].grad.double()
    print('%-40s rel %.3e  |ref| %.3e  cos %.8f' % (k, (a - b).norm().item() / (b.norm().item() + 1e-30), b.norm().item(),
          (a.flatten() @ b.flatten()).item() / (a.norm().item() * b.norm().item() + 1e-30)))
