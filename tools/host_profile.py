"""GPU box: where the HOST time of one step goes (cProfile over a few steps, top functions by own time). usage: python tools/host_profile.py [agg|mldg] [f32|bf16] [steps]"""
import sys, os, copy, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus
what = sys.argv[1] if len(sys.argv) > 1 else 'agg'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
K.set_conv_precision(dtype)
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=(what == 'agg')), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768, seed=304)
x, y = x.cuda(), y.cuda()
if what == 'mldg':
    u1, u2 = copy.deepcopy(net), copy.deepcopy(net)
    inner = [harness.INNER_LR]

    def step():
        out = harness.mldg_train_step(net, u1, u2, opt, x[:4], y[:4], x[4:], y[4:], inner_lr=inner[0], sched=sched, inner_lr_anneal=True)
        inner[0] = out.pop('next_inner_lr')
else:
    gts, aux = y, None

    def step():
        harness.agg_train_step(net, opt, x, y, sched=sched)
for _ in range(3):
    step()
torch.cuda.synchronize()
# enqueue time of one step from an idle GPU
t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('%s %s: enqueue %.1f ms, step incl. GPU %.1f ms' % (what, dtype, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(28)
    print('== by', key, '(%d steps)' % steps)
    print('\n'.join(l[:150] for l in s.getvalue().split('\n')[4:40]))
