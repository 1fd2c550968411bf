"""GPU: which part of the agg step refuses hipGraph capture? Captures pieces separately and reports."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.hip import kernels as K, ops
from pinthememory_amd.network import deepv3plus
K.set_conv_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=True), 19, crit, crit)).cuda()
net.dsn[3].p = 0.0
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(2, 256)
x, y = x.cuda(), y.cuda()
x = ops.nchw(K.nchw_to_nhwc(x.float(), c_pad=4))
harness.COMMIT_OVERLAP = False
for _ in range(3):
    harness.agg_train_step(net, opt, x, y)
torch.cuda.synchronize()


def attempt(name, fn):
    g = torch.cuda.CUDAGraph()
    try:
        with torch.autograd.set_multithreading_enabled(os.environ.get('MT', '0') == '1'), torch.cuda.graph(g):
            fn()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        print('CAPTURE OK   ', name, flush=True)
    except Exception as e:      # noqa: BLE001
        print('CAPTURE FAIL ', name, '->', str(e).splitlines()[0][:120], flush=True)
        if os.environ.get('TB'):
            print(''.join(traceback.format_exc().splitlines(True)[-14:]), flush=True)
        try:
            torch.cuda.synchronize()
        except Exception:      # noqa: BLE001
            pass


def eval_fwd():
    net.eval()
    with torch.no_grad():
        net(x, gts=y, aux_gts=y, memory_writing=True)
    net.train()


def train_fwd():
    net.train()
    return net(x, gts=y, aux_gts=y, memory_writing=True, writing_detach=False)


def trunk_only():
    net.train()
    return net._trunk(x)


def fwd_bwd(overlap):
    ops.OVERLAP_WGRAD = overlap
    opt.zero_grad()
    out = train_fwd()
    harness.total_loss(out).backward()
    if overlap:
        torch.cuda.current_stream().wait_stream(ops._side_stream())


def mem_losses():
    m = net.memory
    mem = m.m_items
    return m.diversityloss(mem) + m.classification_loss(mem)


attempt('memory losses (matmul + Linear + CE: rocBLAS / hipBLASLt)', mem_losses)
attempt('eval-mode commit forward', eval_fwd)
attempt('train forward, trunk only', trunk_only)
attempt('train forward, whole', train_fwd)
attempt('train forward + backward, weight gradients inline', lambda: fwd_bwd(False))
attempt('train forward + backward, weight gradients on the side stream', lambda: fwd_bwd(True))
attempt('SGD step', lambda: opt.step())
