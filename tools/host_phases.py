"""GPU debug: HOST time of the phases of one agg train step (enqueue only, no synchronisation inside), to see where the host falls behind the two
streams that now consume its launches (training forward on the main stream, commit forward on its own)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768)
x, y = x.cuda(), y.cuda()
for _ in range(3):
    harness.agg_train_step(net, opt, x, y, sched=sched)
torch.cuda.synchronize()
T = {}
def tick(name, t0):
    T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
for it in range(6):
    t = time.perf_counter(); net.train(); tick('net.train()', t)
    t = time.perf_counter(); net.eval(); tick('net.eval()', t)
    net.train()
    t = time.perf_counter(); opt.zero_grad(); tick('zero_grad', t)
    t = time.perf_counter(); out = net(x, gts=y, aux_gts=y, memory_writing=True, writing_detach=False); tick('train forward enqueue', t)
    t = time.perf_counter(); loss = harness.total_loss(out); tick('total_loss', t)
    t = time.perf_counter(); loss.backward(); tick('backward enqueue', t)
    t = time.perf_counter(); opt.step(); tick('opt.step', t)
    t = time.perf_counter()
    with torch.no_grad():
        net.eval(); net(x, gts=y, aux_gts=y, memory_writing=True); net.train()
    tick('commit forward enqueue (incl. eval/train toggles)', t)
    t = time.perf_counter(); sched.step(); tick('sched.step', t)
    torch.cuda.synchronize()
for k, v in T.items():
    print('%-52s %s' % (k, ' '.join('%6.2f' % a for a in v)))
