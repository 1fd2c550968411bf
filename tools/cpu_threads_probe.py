"""Host: oracle agg step (bs=2, 768^2) at several torch thread counts -- which one is the fairest CPU baseline on this box?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.ref_cpu import deeplab, harness
from pinthememory_amd import synth
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
x, y = synth.make_batch(2, 768)
for th in [int(v) for v in sys.argv[1:]] or [16, 32, 64, 128]:
    torch.set_num_threads(th)
    net = synth.load_det_weights(deeplab.DeepR50V3PlusD(synth.model_args(), 19, crit, crit))
    opt, _ = harness.make_optimizer(net)
    harness.agg_train_step(net, opt, x, y)
    t0 = time.time()
    harness.agg_train_step(net, opt, x, y)
    dt = time.time() - t0
    print('threads %3d: %.2f s/step = %.3f img/s' % (th, dt, 2 / dt), flush=True)
