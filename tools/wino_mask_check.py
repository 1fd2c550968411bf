"""GPU debug: ReLU sign disagreements between the F(4x4) and the direct route in an eval-mode forward (per Bottleneck output)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import synth
from pinthememory_amd.network import deepv3plus
from pinthememory_amd.hip import kernels as K
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)).cuda().eval()
x, y = synth.make_batch(2, 128)
outs = {}
def run(mode):
    K.set_winograd(mode)
    keep, hs = {}, []
    for ln in ('layer1', 'layer2', 'layer3', 'layer4'):
        for i, blk in enumerate(getattr(net, ln)):
            hs.append(blk.register_forward_hook(lambda m, inp, out, k='%s.%d' % (ln, i): keep.__setitem__(k, out[0].detach().clone())))
    with torch.no_grad():
        net(x.cuda())
    for h in hs:
        h.remove()
    return keep
o0, o4 = run(0), run(4)
K.set_winograd(4)
for k in o0:
    a, b = o0[k], o4[k]
    flips = int(((a > 0) != (b > 0)).sum())
    tiny = int(((a > 0) & (a < 1e-5 * a.max())).sum())
    print('%-10s max %.3e  rel diff %.2e  sign flips %6d  positive-but-<1e-5*max %6d  zeros %.1f%%' % (k, a.max().item(), (a - b).abs().max().item() / a.max().item(), flips, tiny,
          100.0 * float((a == 0).float().mean())))
