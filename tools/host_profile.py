"""GPU debug: cProfile of the host side of three agg train steps (where do the ~42 ms of Python per step go?)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
K.set_conv_precision(os.environ.get('DTYPE', 'f32'))
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768)
x, y = x.cuda(), y.cuda()
for _ in range(3):
    harness.agg_train_step(net, opt, x, y, sched=sched)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    harness.agg_train_step(net, opt, x, y, sched=sched)
    torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
