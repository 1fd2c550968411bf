"""GPU, by hand: N real ranks (gloo, all on the box's one GPU -- RCCL refuses several ranks on one device) against ONE process running the whole batch, through the harness'
agg train step (SyncBatchNorm exchanges, bucketed gradient all-reduce, memory-slot all-reduce, commit forward): the committed memory, the losses and a few post-step
parameters of the N-rank run equal the single-process big-batch run (with the memory-slot sum on, every rank holds the single-process memory).
usage: python tools/gloo_ranks_probe.py <ranks> <images per rank> <size>       (e.g. 8 8 256: BASELINE configs[3]'s bs=64 partitioning at a reduced crop)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[5])
out_path, world, per, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rank = int(os.environ.get('RANK', '0'))
torch.cuda.set_device(0)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
from pinthememory_amd import dist as D, harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, crit, crit)).cuda()
net.dsn[3].p = 0.0
buckets = None
if world > 1:
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
opt, sched = harness.make_optimizer(net)
if world > 1:
    buckets = D.GradBuckets(net.parameters())
total = per * (world if world > 1 else int(os.environ['PROBE_TOTAL_RANKS']))
x, y = synth.make_batch(total, size, seed=5)
if world > 1:
    x, y = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
x, y = x.cuda(), y.cuda()
n0 = D.COLLECTIVES[0]
losses = harness.agg_train_step(net, opt, x, y, sched=sched, buckets=buckets)
harness.finish_commit(net)
ncoll = D.COLLECTIVES[0] - n0
torch.cuda.synchronize()
res = dict(m_items=net.memory.m_items.detach().cpu(), collectives=ncoll,
           params={n: p.detach().cpu() for n, p in net.named_parameters() if n in ('layer0.0.weight', 'layer2.1.conv2.weight', 'layer4.2.bn2.bias', 'aspp.features.2.0.weight', 'final2.0.weight')},
           running={n: b.cpu() for n, b in net.named_buffers() if n in ('layer0.1.running_mean', 'layer3.2.bn2.running_var', 'aspp.img_conv.1.running_var')})
if rank == 0:
    torch.save(res, out_path)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print('PROBE_DONE', rank, flush=True)
'''


def main():
    import socket
    import torch
    ranks, per, size = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    tmp = tempfile.mkdtemp()
    one, many = os.path.join(tmp, 'one.pt'), os.path.join(tmp, 'many.pt')
    t0 = time.time()
    r = subprocess.run([sys.executable, '-c', WORKER, one, '1', str(per), str(size), ROOT], env=dict(os.environ, PROBE_TOTAL_RANKS=str(ranks)), capture_output=True, text=True, timeout=1800)
    assert 'PROBE_DONE 0' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    t1 = time.time()
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = str(s.getsockname()[1])
    procs = []
    for rank in range(ranks):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(ranks))
        procs.append(subprocess.Popen([sys.executable, '-c', WORKER, many, str(ranks), str(per), str(size), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=3000) for p in procs]
    assert all('PROBE_DONE' in o[0] for o in outs), ''.join(o[0][-1500:] + o[1][-3000:] for o in outs)
    t2 = time.time()
    a, b = torch.load(one), torch.load(many)

    def rel(u, v):
        return ((u.double() - v.double()).norm() / (v.double().norm() + 1e-30)).item()
    print('%d ranks x %d images %dx%d (gloo, one GPU) vs one process with %d images: single %.0f s, ranks %.0f s' % (ranks, per, size, size, ranks * per, t1 - t0, t2 - t1))
    print('collectives issued per rank in one agg step (SyncBN forward + backward exchanges, gradient buckets, memory slots): %d' % b['collectives'])
    print('committed memory: max |ranks - single| = %.3e (rel %.2e)' % ((b['m_items'] - a['m_items']).abs().max().item(), rel(b['m_items'], a['m_items'])))
    for n in a['running']:
        print('running moment %-32s rel %.2e' % (n, rel(b['running'][n], a['running'][n])))
    for n in a['params']:
        print('post-step parameter %-28s rel %.2e' % (n, rel(b['params'][n], a['params'][n])))
    ok = rel(b['m_items'], a['m_items']) < 1e-4 and all(rel(b['running'][n], a['running'][n]) < 1e-4 for n in a['running']) and all(rel(b['params'][n], a['params'][n]) < 1e-4 for n in a['params'])
    print('EQUAL TO THE SINGLE-PROCESS BIG BATCH: %s' % ok)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
