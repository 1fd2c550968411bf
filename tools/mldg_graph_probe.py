"""GPU: capture the train_memory_mldg iteration piece by piece in a hipGraph (round 6 debugging: hipStreamEndCapture crashed on the whole step).
usage: mldg_graph_probe.py <stage 1..7> [size] [tier]"""
import os, sys, copy, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness as h, synth
from pinthememory_amd.hip import kernels as K, ops
from pinthememory_amd.network import deepv3plus
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 7
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
tier = sys.argv[3] if len(sys.argv) > 3 else 'f32'
variant = sys.argv[4] if len(sys.argv) > 4 else ''
CRIT = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
K.set_conv_precision(tier)
if 'nowgrad' in variant:
    ops.OVERLAP_WGRAD = False
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, CRIT, CRIT)).cuda()
net.dsn[3].p = 0.0
u1, u2 = copy.deepcopy(net), copy.deepcopy(net)
opt, sched = h.make_optimizer(net)
x, y = synth.make_batch(4, size, seed=60)
x, y = x.cuda(), y.cuda()
x_tr, y_tr, x_te, y_te = x[:2], y[:2], x[2:], y[2:]
lr = torch.zeros(1, device='cuda')
inner_lr = torch.zeros((), device='cuda')


def step(stage):
    h.set_mode(net, True)
    h.finish_commit(net)
    mem_t = net.memory.m_items.clone().detach()
    opt.zero_grad()
    out_in = net(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True, writing_detach=False)
    inner = h.total_loss(out_in)
    if stage == 0:
        return inner
    inner.backward(retain_graph='noretain' not in variant)
    if 'optstep' in variant:
        opt.step()
    if stage == 1:
        return inner
    theta = h.functional_theta(net, inner_lr)
    if stage == 2:
        return theta
    un = h.set_mode(h.get_updated_network(net, u1, inner_lr, theta), True)
    theta2 = {k: (v if k.split('.')[0] == 'memory' else v.detach()) for k, v in theta.items()}
    un2 = h.set_mode(h.get_updated_network(net, u2, inner_lr, theta2), True)
    if stage == 3 and 'w3a' in variant:
        return theta
    un2.memory.m_items = mem_t
    if stage == 3 and 'w3b' in variant:
        with torch.no_grad():
            un2(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True, writing_detach=False)
        return theta
    if stage == 3 and 'w3c' in variant:
        un2(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=False)
        return theta
    if stage == 3 and 'w3d' in variant:
        h.set_mode(un2, False)
        with torch.no_grad():
            un2(x_tr)
        h.set_mode(un2, True)
        return theta
    un2(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True, writing_detach=False)
    if stage == 3:
        return theta
    un.memory.m_items = un2.memory.m_items.clone()
    out_te = un(x_te, gts=y_te, aux_gts=y_te, memory_writing=False)
    outer = out_te[0] + h.LOSS_W['aux'] * out_te[1] + h.LOSS_W['read'] * out_te[-2]
    if stage == 4:
        return outer
    outer.backward()
    if stage == 5:
        return outer
    opt.step()
    if stage == 6:
        return outer
    with torch.no_grad():
        h.set_mode(net, False)
        net.memory.m_items = mem_t
        net(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True)
        h.set_mode(net, True)
    return outer


opt.lr_device = lr
cap_stream = torch.cuda.Stream() if 'sidestream' in variant else None
if cap_stream is not None:
    cap_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(cap_stream)
for _ in range(2):
    lr.fill_(0.01), inner_lr.fill_(1e-3)
    step(stage if 'warmsame' in variant else (int(variant.split('warm=')[1][0]) if 'warm=' in variant else 7))
torch.cuda.synchronize()
mem = net.memory.m_items.detach().clone()
net.memory.m_items = mem
if 'drop' in variant:
    import gc
    for u in (u1, u2):
        h.put_theta(u, {k: v.detach().clone() for k, v in net.named_parameters()})
    gc.collect()
    torch.cuda.synchronize()
if 'clean' in variant:
    net.memory.pending = None
    ops.commit_done.clear()
    ops.last_prefold_event = None
    K.forget_filter_events()
print('stage', stage, 'warm, overlap_wgrad', ops.overlap_wgrad(), flush=True)
g = torch.cuda.CUDAGraph()
mode = 'relaxed' if 'relaxed' in variant else ('thread_local' if 'tlocal' in variant else 'global')
if 'gc' in variant:
    import gc
    gc.collect()
    torch.cuda.synchronize()
if 'empty' in variant:
    torch.cuda.empty_cache()
with torch.cuda.graph(g, capture_error_mode=mode, stream=cap_stream):
    keep = step(stage)
    if ops.overlap_wgrad():
        torch.cuda.current_stream().wait_stream(ops._side_stream())
print('captured', flush=True)
g.replay()
torch.cuda.synchronize()
print('ok', flush=True)
