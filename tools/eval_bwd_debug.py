"""GPU debug: eval-mode (frozen BN) backward, per-parameter gradient error vs the fp64 CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_cpu import deeplab
from pinthememory_amd import synth
from pinthememory_amd.network import deepv3plus
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
from pinthememory_amd.hip import kernels as K
if os.environ.get('WINO') is not None:
    K.set_winograd(int(os.environ['WINO']))
args = synth.model_args()
x, y = synth.make_batch(2, 128)
ref = synth.load_det_weights(deeplab.DeepR50V3PlusD(args, 19, CRIT, CRIT)).double().eval()
ref.memory.m_items = ref.memory.m_items.double()
CRIT(ref(x.double())[0], y).backward()
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(args, 19, CRIT, CRIT)).cuda().eval()
CRIT(net(x.cuda())[0], y.cuda()).backward()
rp, hp = dict(ref.named_parameters()), dict(net.named_parameters())
for n in rp:
    if rp[n].grad is None:
        continue
    t = rp[n].grad
    e = (hp[n].grad.double().cpu() - t).abs().max().item() / (t.abs().max().item() + 1e-30) if hp[n].grad is not None else float('nan')
    if e > 1e-5 or n.endswith('conv1.weight'):
        print('%-40s %.3e' % (n, e))

# ---- gradients at the block outputs of layer3 ------------------------------------------------------------------------
def run(net, xin, yin):
    keep = {}
    hs = []
    for i, blk in enumerate(net.layer3):
        def hook(mod, inp, out, i=i):
            t = out[0] if isinstance(out, (list, tuple)) else out
            t.retain_grad()
            keep[i] = t
        hs.append(blk.register_forward_hook(hook))
    for p in net.parameters():
        p.grad = None
    CRIT(net(xin)[0], yin).backward()
    for h in hs:
        h.remove()
    return {i: t.grad.detach().double().cpu() for i, t in keep.items()}
gr, gh = run(ref, x.double(), y), run(net, x.cuda(), y.cuda())
for i in sorted(gr):
    d = (gh[i] - gr[i]).abs()
    print('layer3.%d output grad: rel err %.3e   #elements off by > 1e-4 of max: %d of %d' % (i, d.max().item() / gr[i].abs().max().item(),
          int((d > 1e-4 * gr[i].abs().max()).sum()), d.numel()))
