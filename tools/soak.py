import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768); x, y = x.cuda(), y.cuda()
for blk in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        out = harness.agg_train_step(net, opt, x, y, sched=sched)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print('steps %3d-%3d: %.2f ms/step  loss %.4f  alloc %.2f GB  reserved %.2f GB  peak %.2f GB' % (blk * 50, blk * 50 + 49, dt * 1e3, out['total'].item(),
          torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9, torch.cuda.max_memory_allocated() / 1e9), flush=True)
assert torch.isfinite(out['total']).item()
