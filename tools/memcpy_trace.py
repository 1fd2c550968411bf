"""GPU debug: where do the small device copies of one agg train step come from? (torch.profiler, python stacks)"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(2, 256)
x, y = x.cuda(), y.cuda()
for _ in range(2):
    harness.agg_train_step(net, opt, x, y, sched=sched)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    harness.agg_train_step(net, opt, x, y, sched=sched)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if 'copy' in n.lower() or 'Memcpy' in n or 'memcpy' in n:
        st = [s for s in (e.stack or []) if 'pinthememory_amd' in s or 'optim' in s]
        cnt[(n[:40], st[0][-90:] if st else '?')] += 1
for k, v in cnt.most_common(25):
    print(v, k)
