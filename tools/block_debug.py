import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_cpu import resnet as o_res
from pinthememory_amd.network import Resnet as p_res
from pinthememory_amd.network.mynn import channels_last_weights
from pinthememory_amd import synth
from pinthememory_amd.hip import ops

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()

torch.manual_seed(0)
for name, mk in (('block_noDS', lambda R: R.Bottleneck(256, 64)),
                 ('block_DS', lambda R: R.Bottleneck(64, 64, 1, torch.nn.Sequential(torch.nn.Conv2d(64, 256, 1, bias=False), torch.nn.BatchNorm2d(256)))),
                 ('block_s2', lambda R: R.Bottleneck(256, 128, 2, torch.nn.Sequential(torch.nn.Conv2d(256, 512, 1, stride=2, bias=False), torch.nn.BatchNorm2d(512))))):
    ref = mk(o_res)
    net = mk(p_res)
    sd = synth.det_state_dict(ref)
    ref.load_state_dict(sd); net.load_state_dict(sd)
    net = channels_last_weights(net).cuda()
    ref.train(); net.train()
    cin = ref.conv1.in_channels
    for hw in (32, 8):
        x = torch.relu(torch.randn(2, cin, hw, hw))
        xr = x.clone().requires_grad_(True)
        xg = ops.nchw(x.permute(0, 2, 3, 1).contiguous().cuda()).requires_grad_(True)
        yr = ref(xr)
        yg = net([xg, []])[0]
        dy = torch.randn_like(yr)
        yr.backward(dy)
        yg.backward(ops.nchw(dy.permute(0, 2, 3, 1).contiguous().cuda()))
        print('%s hw=%d  out %.2e  dx %.2e' % (name, hw, rel(yg, yr.detach()), rel(xg.grad, xr.grad)), end='  ')
        gr, gg = dict(ref.named_parameters()), dict(net.named_parameters())
        print(' '.join('%s %.1e' % (k.replace('.weight', '.w').replace('.bias', '.b'), rel(gg[k].grad, gr[k].grad)) for k in gr))
        ref.zero_grad(); net.zero_grad()
