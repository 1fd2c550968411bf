"""GPU box: cProfile of the host side of the agg step (5 steps after warm-up), top functions by own and cumulative time."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
from pinthememory_amd.hip import kernels as _K
_K.set_conv_precision(os.environ.get('DTYPE', 'f32'))
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768)
x, y = x.cuda(), y.cuda()
for _ in range(3):
    harness.agg_train_step(net, opt, x, y, sched=sched)
torch.cuda.synchronize()
import contextlib
ctx = torch.autograd.set_multithreading_enabled(False) if '--inline-backward' in sys.argv else contextlib.nullcontext()
ctx.__enter__()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    harness.agg_train_step(net, opt, x, y, sched=sched)
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print('\n'.join(l[:150] for l in s.getvalue().splitlines()[4:56]))
