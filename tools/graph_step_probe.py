"""GPU: harness.GraphedAggStep (the agg step captured in a hipGraph) against eager steps -- same bits after N steps (deterministic configuration), ms/step of both.
usage: graph_step_probe.py [f32|bf16] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bs, size = int(os.environ.get('BS', 8)), int(os.environ.get('SIZE', 768))
K.set_conv_precision(dtype)
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)


def make():
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=True), 19, crit, crit)).cuda()
    net.dsn[3].p = 0.0
    opt, sched = harness.make_optimizer(net)
    return net, opt, sched


x, y = synth.make_batch(bs, size)
x, y = x.cuda(), y.cuda()
# eager: 3 + steps serial-commit steps (the graph's order), then the timed overlapped steps
harness.COMMIT_OVERLAP = False
net_e, opt_e, sched_e = make()
for _ in range(3 + steps):
    le = harness.agg_train_step(net_e, opt_e, x, y, sched=sched_e)
torch.cuda.synchronize()
net_g, opt_g, sched_g = make()
g = harness.GraphedAggStep(net_g, opt_g, x, y, sched=sched_g, warmup=3)
for _ in range(steps):
    lg = g.step(x, y)
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(net_e.state_dict().values(), net_g.state_dict().values())) and torch.equal(net_e.memory.m_items, net_g.memory.m_items)
print('after %d steps: graph == eager bit for bit: %s; losses eager %.6f graph %.6f' % (3 + steps, same, le['total'].item(), lg['total'].item()), flush=True)
harness.COMMIT_OVERLAP = True
for name, fn in (('eager (overlapped commit)', lambda: harness.agg_train_step(net_e, opt_e, x, y, sched=sched_e)), ('graph replay', lambda: g.step(x, y))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print('%-28s %.2f ms/step' % (name, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
