"""GPU: the assembled-gradient error (vs the fp64 CPU oracle, bs=2 128^2) over several seeded batches: HIP with the split path on / off and the fp32 CPU oracle itself.
Round 6: is the 4 x difference on the suite's single batch a property of the split path or of the ReLU-flip lottery at the decoder? usage: split_grad_seeds.py [nseeds] [size]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_cpu import deeplab as o_deeplab, harness as o_h
from pinthememory_amd import harness as h, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
args = synth.model_args()
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.set_num_threads(min(24, os.cpu_count() or 1))


def oracle(dtype, x, y):
    ref = synth.load_det_weights(o_deeplab.DeepR50V3PlusD(args, 19, CRIT, CRIT)).to(dtype)
    ref.memory.m_items = ref.memory.m_items.to(dtype)
    ref.dsn[3].p = 0.0
    ref.train()
    o_h.total_loss(ref(x.to(dtype), gts=y, aux_gts=y, memory_writing=True, writing_detach=False)).backward()
    return {k: p.grad.double() for k, p in ref.named_parameters() if p.grad is not None}


def hip(split, x, y):
    K.set_split(split)
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(args, 19, CRIT, CRIT)).cuda()
    net.dsn[3].p = 0.0
    net.train()
    h.total_loss(net(x.cuda(), gts=y.cuda(), aux_gts=y.cuda(), memory_writing=True, writing_detach=False)).backward()
    return {k: p.grad.double().cpu() for k, p in net.named_parameters() if p.grad is not None}


def stats(g, t):
    e = sorted(((g[k] - t[k]).norm().item() / t[k].norm().item(), k) for k in t if t[k].norm().item() > 1e-7)
    return e[len(e) // 2][0], e[-1]


for seed in range(nseeds):
    x, y = synth.make_batch(2, size, seed=seed * 7)
    t = oracle(torch.float64, x, y)
    rows = [('fp32 oracle', stats(oracle(torch.float32, x, y), t)), ('hip fp32-MFMA', stats(hip(False, x, y), t)), ('hip split', stats(hip(True, x, y), t))]
    print('seed %2d: ' % (seed * 7) + ' | '.join('%s median %.2e worst %.2e (%s)' % (n, m, w[0], w[1]) for n, (m, w) in rows), flush=True)
K.set_split(True)
