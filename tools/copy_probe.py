"""GPU box: which host-side ops put the ~85 copies (__amd_rocclr_copyBuffer) and the ~65 fill kernels into one agg step? torch.profiler over 2 steps with
device activity: every memcpy / memset / fill device event, grouped by the CPU op that launched it and its first project frame."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv3plus
from pinthememory_amd.hip import kernels as _K
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
_K.set_conv_precision(os.environ.get('DTYPE', 'f32'))
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768)
x, y = x.cuda(), y.cuda()
for _ in range(3):
    harness.agg_train_step(net, opt, x, y, sched=sched)
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(STEPS):
        harness.agg_train_step(net, opt, x, y, sched=sched)
    torch.cuda.synchronize()
evs = prof.events()
agg = collections.Counter()
dur = collections.Counter()
for ev in evs:
    dk = [k for k in (ev.kernels or [])]
    for k in dk:
        nm = k.name
        if 'copyBuffer' in nm or 'Memcpy' in nm or 'Memset' in nm or 'fillBuffer' in nm or 'FillFunctor' in nm:
            st = [s for s in (ev.stack or []) if 'pinthememory_amd' in s or 'bench' in s]
            key = (nm[:40], ev.name, str(ev.input_shapes)[:50], st[0][-100:] if st else '-')
            agg[key] += 1
            dur[key] += k.duration
for key, n in sorted(agg.items(), key=lambda kv: -kv[1])[:50]:
    print('%6.1f/step %7.1f us/step  %-40s %-22s %-50s %s' % ((n / STEPS, dur[key] / STEPS) + key))
print('total device copy/fill events per step', sum(agg.values()) / STEPS)
