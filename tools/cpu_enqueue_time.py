"""GPU debug: host time to enqueue one agg train step vs its GPU time (is the step launch-bound anywhere?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
from pinthememory_amd.hip import kernels as _K
_K.set_conv_precision(os.environ.get('DTYPE', 'f32'))      # DTYPE=bf16: is the configs[2] tier host-bound?
buckets = None
force = os.environ.get('PM_DIST_FORCE', '0') == '1'      # one-rank RCCL rehearsal: what do the collectives of the N > 1 path cost?
if force:
    import torch.distributed as dist
    from pinthememory_amd.network import mynn
    for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29533'), ('RANK', '0'), ('WORLD_SIZE', '1')):
        os.environ.setdefault(k, v)
    torch.cuda.set_device(0)
    dist.init_process_group(backend='nccl', device_id=torch.device('cuda', 0))
    mynn.set_bnfunc(torch.nn.SyncBatchNorm)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
if force:
    from pinthememory_amd import dist as D
    buckets = D.GradBuckets(net.parameters())
x, y = synth.make_batch(8, 768)
x, y = x.cuda(), y.cuda()
for _ in range(3):
    harness.agg_train_step(net, opt, x, y, sched=sched, buckets=buckets)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(6):
    a = time.perf_counter()
    harness.agg_train_step(net, opt, x, y, sched=sched, buckets=buckets)
    enq.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('host enqueue per step: %s ms' % ['%.1f' % (e * 1e3) for e in enq])
print('enqueue total %.1f ms, wall incl. final sync %.1f ms -> %.1f ms/step; host is ahead of the GPU by %.1f ms at the end' % (t_enq * 1e3, t_all * 1e3, t_all / 6 * 1e3, (t_all - t_enq) * 1e3))
# host-only cost: the same step with the GPU drained before and after (enqueue time when the queue never back-pressures)
for _ in range(2):
    torch.cuda.synchronize()
    a = time.perf_counter()
    harness.agg_train_step(net, opt, x, y, sched=sched, buckets=buckets)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    print('from idle: host returns after %.1f ms, GPU done after %.1f ms' % ((b - a) * 1e3, (c - a) * 1e3))
if force:
    from pinthememory_amd import rccl
    rccl.shutdown()
    dist.destroy_process_group()
