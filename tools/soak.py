"""GPU: N blocks of 50 agg steps on one batch (default 4; `soak.py 20` = 1000 steps), with a fresh uint8 batch per step through the side-stream
prefetcher when `--edge` is given: step time, loss, allocator state per block -- drift and growth would show here."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
if '--bf16' in sys.argv:
    from pinthememory_amd.hip import kernels as K
    K.set_conv_precision('bf16')
det = '--deterministic' in sys.argv      # no gumbel noise, no dropout: two runs (fp32 / bf16) see the same optimisation problem
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=det), 19, crit, crit)).cuda()
if det:
    net.dsn[3].p = 0.0
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768); x, y = x.cuda(), y.cuda()
blocks = int(next((a for a in sys.argv[1:] if a.isdigit()), 4))
pf = None
if '--edge' in sys.argv:
    from pinthememory_amd import input_edge
    pf = input_edge.DevicePrefetcher(input_edge.SyntheticDomainSource(4, 2, 768, n_buffers=3, seed=1, static=True), depth=1)
for blk in range(blocks):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        if pf is not None:
            x, y = pf.next()
        out = harness.agg_train_step(net, opt, x, y, sched=sched)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print('steps %3d-%3d: %.2f ms/step  loss %.4f (loss1 %.4f loss2 %.4f read %.4f)  alloc %.2f GB  reserved %.2f GB  peak %.2f GB' % (blk * 50, blk * 50 + 49, dt * 1e3, out['total'].item(), out['loss1'].item(), out['loss2'].item(), out['readloss'].item(),
          torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9, torch.cuda.max_memory_allocated() / 1e9), flush=True)
assert torch.isfinite(out['total']).item()
