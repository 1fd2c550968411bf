"""GPU: where do the D2D copies / fills of one agg step come from? torch.profiler with stacks, grouped by the innermost repo frame."""
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
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    harness.agg_train_step(net, opt, x, y, sched=sched)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::clone', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::contiguous', 'aten::zeros', 'aten::zeros_like'):
        chain, q = [], ev.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name)
            q = q.cpu_parent
        cnt[(ev.name, ' < '.join(chain) + '  shape ' + str(ev.input_shapes[:1]))] += 1
for (n, f), c in cnt.most_common(40):
    print('%4d %-18s %s' % (c, n, f[-110:]))
