"""GPU: what does running the memory-commit (eval-mode) forward on a second stream under the next step's training forward buy?
Two model instances with the same weights, both under no_grad: sequential vs concurrent wall time of (train-mode forward, eval-mode forward)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net_t = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda().train()
net_e = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda().eval()
x, y = synth.make_batch(8, 768)
x, y = x.cuda(), y.cuda()
s2 = torch.cuda.Stream()

def f_t():
    with torch.no_grad():
        net_t(x, gts=y, aux_gts=y, memory_writing=True, writing_detach=True)

def f_e():
    with torch.no_grad():
        net_e(x, gts=y, aux_gts=y, memory_writing=True)

def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def seq():
    f_t(); f_e()

def conc():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        f_e()
    f_t()
    torch.cuda.current_stream().wait_stream(s2)

print('train-mode forward alone %.2f ms, eval-mode forward alone %.2f ms' % (timed(f_t), timed(f_e)))
print('sequential %.2f ms, concurrent on two streams %.2f ms' % (timed(seq), timed(conc)))
print('sequential %.2f ms, concurrent on two streams %.2f ms' % (timed(seq), timed(conc)))
