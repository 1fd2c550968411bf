"""GPU: the commit-forward overlap at the flagship size -- N agg steps at bs=8, 768 x 768 (default configuration: gumbel read, dropout) with the overlap on
and off from the same initial state and the same random draws: every loss, the committed memory and the whole state_dict must carry the same bits. The
flagship-size kernels are long enough for the two streams to really run side by side, which the small test case cannot guarantee."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
x, y = synth.make_batch(8, 768)
x, y = x.cuda(), y.cuda()


def run(overlap):
    harness.COMMIT_OVERLAP = overlap
    torch.manual_seed(7)
    torch.cuda.manual_seed(7)
    net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
    opt, sched = harness.make_optimizer(net)
    losses = [harness.agg_train_step(net, opt, x, y, sched=sched) for _ in range(steps)]
    mem = net.memory.m_items.clone()
    torch.cuda.synchronize()
    return net, losses, mem


n1, l1, m1 = run(True)
n0, l0, m0 = run(False)
bad = [(i, k) for i, (a, b) in enumerate(zip(l1, l0)) for k in a if not torch.equal(a[k], b[k])]
sd1, sd0 = n1.state_dict(), n0.state_dict()
bad_state = [k for k in sd1 if not torch.equal(sd1[k], sd0[k])]
print('steps %d: losses differing %s, memory equal %s, state entries differing %d / %d' % (steps, bad[:5], torch.equal(m1, m0), len(bad_state), len(sd1)))
print('final loss', float(l1[-1]['total']), float(l0[-1]['total']))
assert not bad and torch.equal(m1, m0) and not bad_state, bad_state[:5]
print('OK: bit-identical')
