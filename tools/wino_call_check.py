"""GPU debug: every conv_fwd / conv_bwd_data / conv_bwd_weight call of one eval-mode forward + backward is run on the F(4x4) route and on
the direct route with the same operands; calls whose results differ by more than 1e-4 of the result's scale are listed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import synth
from pinthememory_amd.network import deepv3plus
from pinthememory_amd.hip import kernels as K
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
def wrap(name, pick):
    orig = getattr(K, name)
    def f(*a, **kw):
        r4 = orig(*a, **kw)
        K.set_winograd(0)
        r0 = orig(*a, **kw)
        K.set_winograd(4)
        t4, t0 = pick(r4), pick(r0)
        d = (t4 - t0).abs().max().item() / (t0.abs().max().item() + 1e-30)
        if d > 1e-4 or not torch.isfinite(t4).all():
            shp = [tuple(x.shape) for x in a if torch.is_tensor(x)]
            print('%-16s rel diff %.3e  operands %s  args %s  |in|max %.3e |out|max %.3e' % (name, d, shp, [x for x in a if isinstance(x, (int, tuple))],
                  a[0].abs().max().item(), t0.abs().max().item()), flush=True)
        return r4
    setattr(K, name, f)
wrap('conv_fwd', lambda r: r)
wrap('conv_bwd_data', lambda r: r)
wrap('conv_bwd_weight', lambda r: r[0])
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)).cuda().eval()
x, y = synth.make_batch(2, 128)
CRIT(net(x.cuda())[0], y.cuda()).backward()
print('done')
