"""GPU debug: which host-side operations launch the small (< 12 us) device kernels of one agg train step -- copies, fills, stock elementwise kernels -- and how much
stream time they take (torch.profiler with python stacks; kernels are attributed to the aten op / autograd node that launched them).
usage: DTYPE=bf16 python tools/small_kernel_trace.py [bs] [size]"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pinthememory_amd import harness, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus
K.set_conv_precision(os.environ.get('DTYPE', 'bf16'))
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 768
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(bs, size)
x, y = x.cuda(), y.cuda()
for _ in range(3):
    harness.agg_train_step(net, opt, x, y, sched=sched)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    harness.agg_train_step(net, opt, x, y, sched=sched)
    torch.cuda.synchronize()
# CPU-side ops that are not library launches: aten ops with their python origin
cpu = collections.Counter()
dur = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::') and e.cpu_parent is not None and not e.cpu_parent.name.startswith('aten::'):
        st = [s for s in (e.stack or []) if 'pinthememory_amd' in s or '/tests/' in s or 'bench.py' in s]
        key = (e.name, e.cpu_parent.name[:50], st[0][-80:] if st else '?')
        cpu[key] += 1
        dur[key] += sum(k.duration for k in e.kernels) if hasattr(e, 'kernels') else 0
print('top-level aten ops of one step (count, device us, op, parent, first repo frame):')
for k, v in cpu.most_common(45):
    print('%4d %8.1f  %s' % (v, dur[k], k))
kern = collections.Counter()
kdur = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:70]] += 1
        kdur[e.name[:70]] += e.device_time if hasattr(e, 'device_time') else 0
print('\ndevice kernels under 12 us average (count, total us, name):')
for k, v in sorted(kern.items(), key=lambda kv: -kdur[kv[0]]):
    if kdur[k] / v < 12:
        print('%4d %8.1f  %s' % (v, kdur[k], k))
