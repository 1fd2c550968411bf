"""CPU, world_size 2 over gloo: the three data-parallel exchanges of the hot path (pinthememory_amd/dist.py) reproduce the
single-process big-batch result. The math on each rank is the oracle's (this is host logic: no GPU, no HIP library)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, fn, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        out[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def spawn(fn, world=2):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, out), nprocs=world, join=True)
    return [out[r] for r in range(world)]


# ---- C3: memory-slot all-reduce ------------------------------------------------------------------------------------
def _mem_inputs():
    from pinthememory_amd import synth
    z = torch.relu(synth.det_tensor((4, 256, 6, 6), 5))
    _, mask = synth.make_batch(4, 48, seed=9, block=8)
    return z, mask


def _c3(rank, world):
    from oracle.ref_cpu.memory import Memory_sup
    from pinthememory_amd import dist as D, synth
    z, mask = _mem_inputs()
    M = Memory_sup(19, 256, 256, 0.8, 1, gumbel_read=False)
    M.m_items = synth.det_memory()
    lo, hi = rank * 2, rank * 2 + 2                                  # rank r owns images [2r, 2r+2)
    zl = z[lo:hi].clone().requires_grad_(True)
    nom, den = M.accumulate(F.normalize(zl, dim=1), mask[lo:hi])
    flat = D.all_reduce_sum_autograd(torch.cat([nom.reshape(-1), den.reshape(-1)]))
    nom, den = flat[:20 * 256].view(20, 256), flat[20 * 256:]
    upd = M.update(nom, den)
    (upd * synth.det_tensor((19, 256), 3)).sum().backward()
    return upd.detach(), zl.grad


def test_memory_slot_allreduce_equals_big_batch():
    from oracle.ref_cpu.memory import Memory_sup
    from pinthememory_amd import synth
    res = spawn(_c3)
    z, mask = _mem_inputs()
    M = Memory_sup(19, 256, 256, 0.8, 1, gumbel_read=False)
    M.m_items = synth.det_memory()
    zr = z.clone().requires_grad_(True)
    nom, den = M.accumulate(F.normalize(zr, dim=1), mask)
    upd = M.update(nom, den)
    (upd * synth.det_tensor((19, 256), 3)).sum().backward()
    for r in range(2):
        assert torch.allclose(res[r][0], upd.detach(), atol=1e-6)            # every rank holds the big-batch memory
        # each rank's loss sees every rank's features: grads arrive summed over ranks (= world x the single-process grad)
        assert torch.allclose(res[r][1], 2 * zr.grad[r * 2:r * 2 + 2], atol=1e-6, rtol=1e-4)


# ---- C2: BatchNorm statistics merge ----------------------------------------------------------------------------------
def _c2(rank, world):
    from pinthememory_amd import dist as D
    g = torch.Generator().manual_seed(7)
    x = torch.randn(6, 16, 5, 5, generator=g) * 3 + 1
    xl = x[rank * 3:rank * 3 + 3]
    n = xl.numel() // 16
    mean = xl.mean((0, 2, 3))
    m2 = ((xl - mean[None, :, None, None]) ** 2).sum((0, 2, 3))
    mom = torch.cat([mean, m2, torch.full((16,), float(n))])
    return D.merge_moments(mom, 16)


def test_syncbn_moment_merge_equals_global_stats():
    res = spawn(_c2)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(6, 16, 5, 5, generator=g) * 3 + 1
    mean = x.mean((0, 2, 3))
    m2 = ((x - mean[None, :, None, None]) ** 2).sum((0, 2, 3))
    for r in range(2):
        assert torch.allclose(res[r][:16], mean, atol=1e-6)
        assert torch.allclose(res[r][16:32], m2, rtol=1e-5)
        assert torch.all(res[r][32:] == 150)


# ---- C1: bucketed gradient all-reduce ---------------------------------------------------------------------------------
def _c1(rank, world):
    from pinthememory_amd import dist as D
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1))
    net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
    buckets = D.GradBuckets(net.parameters(), bucket_bytes=64)           # tiny buckets -> several async all-reduces
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 3, 6, 6, generator=g)
    out = []
    for it in range(2):                                                  # two steps: zero() must reset the arena and the hooks
        buckets.zero()
        net(x[rank * 2:rank * 2 + 2] * (it + 1)).square().mean().backward()
        buckets.finish()
        out.append([p.grad.clone() for p in net.parameters()])
    assert all(p.grad.data_ptr() >= buckets.flat.data_ptr() for p in net.parameters())
    return out, len(buckets.buckets)


def test_grad_buckets_average_equals_full_batch_grad():
    res = spawn(_c1)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 4, 1))
    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 3, 6, 6, generator=g)
    assert res[0][1] > 1
    for it in range(2):
        net.zero_grad()
        net(x * (it + 1)).square().mean().backward()                    # mean over 4 images == average of the two rank means
        for r in range(2):
            for p, got in zip(net.parameters(), res[r][0][it]):
                assert torch.allclose(got, p.grad, atol=1e-6, rtol=1e-4)


def _counted(rank, world):
    from pinthememory_amd import dist as D
    t = torch.full((8,), float(rank + 1))
    mom = torch.cat([torch.full((4,), float(rank)), torch.ones(4), torch.full((4,), 2.0)])

    def step():
        D.all_reduce_sum(t.clone())                                   # C3-style in-place sum
        D.all_reduce_sum_copy(t)                                      # SyncBN backward sums (out of place)
        D.gather_moments(mom)                                         # SyncBN forward moments
        D.merge_moments(mom, 4)
    n = D.count_collectives(step)
    return n, D.direct_fallback_reason('gloo')


def test_collectives_are_counted_identically_on_every_rank():
    """bench.py's `config.collectives_per_step` / `rccl_direct_reason` (round 5): every exchange helper of dist.py counts itself, the count is the same on every rank,
    and a non-RCCL backend names itself as the reason torch.distributed carries the exchanges."""
    res = spawn(_counted)
    assert [r[0] for r in res] == [4, 4]
    assert all('gloo' in r[1] and 'torch.distributed' in r[1] for r in res)
    from pinthememory_amd import dist as D
    assert D.count_collectives(lambda: D.all_reduce_sum(torch.ones(3))) == 0      # single process: no-ops are not counted


def test_single_process_is_a_noop():
    from pinthememory_amd import dist as D
    t = torch.arange(6.0)
    assert D.all_reduce_sum(t.clone()).equal(t) and D.merge_moments(t, 2).equal(t) and D.group_size() == 1
    assert D.all_reduce_sum_autograd(t) is t and D.bn_group(torch.nn.SyncBatchNorm(4)) is None


# ---- GPU: the same-stream RCCL communicator (pinthememory_amd/rccl.py), one-rank group on the 1-GPU box -------------------
_RCCL_PROBE = r'''
import os, sys, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], RANK='0', WORLD_SIZE='1', PM_DIST_FORCE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from pinthememory_amd import dist as D, rccl
comm = rccl.get(None)
assert comm is not None, 'direct RCCL communicator was not created'
x = torch.arange(768, dtype=torch.float32, device='cuda') * 0.5
y = x.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):                      # the collective runs on whatever stream is current
    D.all_reduce_sum(y)
    mom = D.merge_moments(torch.cat([x[:256], x[256:512].abs() + 1, torch.full((256,), 64.0, device='cuda')]), 256)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
assert torch.equal(y, x)
assert torch.allclose(mom[:256], x[:256]) and torch.allclose(mom[512:], torch.full((256,), 64.0, device='cuda'))
rccl.shutdown()
dist.destroy_process_group()
print('RCCL_DIRECT_OK')
'''


@pytest.mark.gpu
def test_direct_rccl_communicator_one_rank():
    """ncclCommInitRank / ncclAllReduce / ncclAllGather through ctypes on torch's own librccl, issued on a non-default stream."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', _RCCL_PROBE, str(_free_port())], cwd=root, capture_output=True, text=True, timeout=300)
    assert 'RCCL_DIRECT_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---- GPU: the whole N > 1 code path in a one-rank RCCL group must reproduce the plain single-process step bit for bit ---------------
_STEP_PROBE = r'''
import hashlib, os, sys, torch
mode = sys.argv[2]                 # '0' plain single process | '1' N > 1 path with the build's GradBuckets | '2' N > 1 path under the reference's DDP wrapper
                                   # '3' as '1', the second step replayed from a hipGraph (harness.GraphedAggStep(buckets=...): every RCCL call a node of the graph)
force = mode != '0'
if force:
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], RANK='0', WORLD_SIZE='1', PM_DIST_FORCE='1')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from pinthememory_amd import dist as D, harness, synth
from pinthememory_amd.network import deepv3plus, mynn
if force:
    mynn.set_bnfunc(torch.nn.SyncBatchNorm)
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, crit, crit)).cuda()
net.dsn[3].p = 0.0
if force:                          # train.py:95: every BatchNorm incl. Memory_sup's two becomes a SyncBatchNorm
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
    assert not any(type(m) is torch.nn.BatchNorm2d for m in net.modules())
    assert sum(isinstance(m, torch.nn.SyncBatchNorm) for m in net.modules()) == 65
opt, sched = harness.make_optimizer(net)
buckets = D.GradBuckets(net.parameters()) if mode in '13' else None
if mode == '2':                    # network/__init__.py:25-33: DistributedDataParallel(net, device_ids=[gpuid], find_unused_parameters=True)
    from pinthememory_amd.network import warp_network_in_dataparallel
    from pinthememory_amd.hip import ops
    assert ops.OVERLAP_WGRAD       # the weight gradients still run on the side stream; DDP's reducer sees them after the _Defer join
    net = warp_network_in_dataparallel(net, 0)
x, y = synth.make_batch(2, 128)
x, y = x.cuda(), y.cuda()
if mode == '3':
    n0 = D.COLLECTIVES[0]
    g = harness.GraphedAggStep(net, opt, x, y, sched=sched, warmup=1, buckets=buckets)      # one eager step, then the capture (records, does not execute)
    per_step = (D.COLLECTIVES[0] - n0) // 2
    assert per_step > 100, per_step                                                          # the capture went through every exchange of the step
    out = g.step(x, y)
    g.close()
else:
    for _ in range(2):
        out = harness.agg_train_step(net, opt, x, y, sched=sched, buckets=buckets)
harness.finish_commit(net)          # every rank: the memory commit of the last step is finished explicitly (an attribute read never hides a collective)
torch.cuda.synchronize()
h = hashlib.sha256()
core = net.module if hasattr(net, 'module') else net
for p in list(core.parameters()) + [core.memory.m_items] + [b for b in core.buffers() if b.dtype == torch.float32]:
    h.update(p.detach().float().contiguous().cpu().numpy().tobytes())
if force:
    from pinthememory_amd import rccl
    assert rccl.get(None) is not None
    rccl.shutdown()
    dist.destroy_process_group()
print('STEP_DIGEST', h.hexdigest(), '%.6f' % out['total'].item())
'''


@pytest.mark.gpu
def test_one_rank_rccl_step_equals_plain_step():
    """SyncBN moments through all-gather + merge (incl. the memory's own two BatchNorms, converted as train.py:95 does), bucketed gradient
    all-reduce on the step's single communicator, memory-slot all-reduce: with one rank every exchange is the identity, so parameters,
    buffers and memory after two agg steps carry the same bits as the plain path -- with the build's GradBuckets (mode 1) and with the
    network wrapped by the reference's own `warp_network_in_dataparallel` (DDP reducer over the side-stream weight gradients, mode 2). Mode 3
    (VERDICT r5 item 6b): the same N > 1 path with the second step replayed from a hipGraph -- the RCCL calls of the direct communicator are captured
    on the step's stream like any other kernel; the total loss is the replayed step's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for force in ('0', '1', '2', '3'):
        r = subprocess.run([sys.executable, '-c', _STEP_PROBE, str(_free_port()), force], cwd=root, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith('STEP_DIGEST')]
        assert lines, r.stdout[-2000:] + r.stderr[-4000:]
        digests.append(lines[0].split()[1:])
    assert digests[0] == digests[1] == digests[2] == digests[3], digests


# ---- GPU: two REAL ranks (gloo, sharing the one GPU of the box) against the single-process big batch ------------------------------------
_TWO_RANK_PROBE = r'''
import os, sys, torch
out_path, world = sys.argv[1], int(sys.argv[2])
split = [int(v) for v in sys.argv[3].split(',')]                     # images per rank (uneven splits: the element count travels with the BN sums)
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
if world > 1:
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)          # train.py:95
x, y = synth.make_batch(4, 128, seed=5)
lo = sum(split[:rank]) if world > 1 else 0
hi = lo + split[rank] if world > 1 else 4
x, y = x[lo:hi].cuda(), y[lo:hi].cuda()
net.train()
outs = net(x, gts=y, aux_gts=y, memory_writing=True, writing_detach=True)
# a loss that is a plain SUM over images: sum over ranks == the big-batch value, so all-reduced (SUM) gradients must equal the big-batch gradients
loss = outs[-1].square().sum() * 1e-3
loss.backward()
names = ['layer0.0.weight', 'layer1.0.bn3.weight', 'layer2.1.conv2.weight', 'layer3.0.downsample.0.weight', 'layer4.2.bn2.bias', 'aspp.features.2.0.weight',
         'aspp.img_conv.1.weight', 'bot_aspp.0.weight']
params = dict(net.named_parameters())
grads = {}
for n in names:
    g = params[n].grad.detach().clone().contiguous()
    if world > 1:
        dist.all_reduce(g)                                            # C1: SUM (the probe's loss is a sum over images)
    grads[n] = g.cpu()
bufs = dict(net.named_buffers())
res = dict(m_items=net.memory.m_items.detach().cpu(), loss=loss.detach().cpu(), grads=grads,
           running={n: bufs[n].cpu() for n in ('layer0.1.running_mean', 'layer3.2.bn2.running_var', 'aspp.img_conv.1.running_var', 'memory.writenet.writefeat.1.running_mean')})
if world > 1:
    t = loss.detach().clone()
    dist.all_reduce(t)
    res['loss'] = t.cpu()
if rank == 0:
    torch.save(res, out_path)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print('PROBE_DONE', rank)
'''


@pytest.fixture(scope='module')
def single_process_reference(tmp_path_factory):
    """ONE process running all four images of _TWO_RANK_PROBE: what every split of the two-rank test is compared with."""
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = str(tmp_path_factory.mktemp('two_rank') / 'one.pt')
    r = subprocess.run([sys.executable, '-c', _TWO_RANK_PROBE, one, '1', '4'], cwd=root, capture_output=True, text=True, timeout=600)
    assert 'PROBE_DONE 0' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return one


@pytest.mark.gpu
@pytest.mark.parametrize('split', ['2,2', '3,1'])         # '1,1,1,1' (four ranks) also passes, run by hand; 8 ranks x 8 images: tools/gloo_ranks_probe.py, profiles/r05_ranks_probe_8x8_256.log
def test_two_ranks_on_gpu_equal_single_process_big_batch(tmp_path, split, single_process_reference):
    """World size 2 for real (two processes, gloo, both on the box's one GPU; RCCL refuses two ranks on one device): rank r runs images
    [2r, 2r + 2) through the HIP path with every BatchNorm converted to SyncBatchNorm (train.py:95) and the memory-slot all-reduce on. Against
    ONE process running all four images: the committed memory (C3), the BatchNorm running moments incl. Memory_sup's own and the 4-sample
    image-pooling BN (C2 forward), the summed loss to fp32 round-off, and the all-reduced gradients of eight parameters from stem to ASPP (C2 backward with the
    all-reduced element count, C1) to the gradient gates of the parity tests. The 3 + 1 split is the uneven last batch: rank 1's image-pooling BatchNorm sees ONE
    value per channel locally, the merged statistics and the all-reduced count are those of the four images."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one, two = single_process_reference, str(tmp_path / 'two.pt')      # the single-process run is shared by the two splits (round 5: the suite's time budget)
    port = str(_free_port())
    procs = []
    world = len(split.split(','))                                     # '1,1,1,1': four ranks, one image each
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, '-c', _TWO_RANK_PROBE, two, str(world), split], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all('PROBE_DONE' in o[0] for o in outs), ''.join(o[0][-1500:] + o[1][-3000:] for o in outs)
    a, b = torch.load(one), torch.load(two)

    def rel(u, v):
        return ((u.double() - v.double()).norm() / (v.double().norm() + 1e-30)).item()
    assert rel(b['m_items'], a['m_items']) < 1e-5
    assert abs(b['loss'].item() - a['loss'].item()) < 1e-4 * abs(a['loss'].item())
    for n in a['running']:
        assert rel(b['running'][n], a['running'][n]) < 1e-5, n
    errs = sorted((rel(b['grads'][n], a['grads'][n]), n) for n in a['grads'])
    print('two-rank vs big-batch gradient errors:', [(round(e, 5), n) for e, n in errs])
    # two fp32 evaluations of a ReLU network differ by the units that flip within round-off: the bounds of tests/test_model_parity.py
    assert errs[-1][0] < 1e-2 and errs[0][0] < 1e-4, errs       # heads behind the last ReLU layers: round-off only; trunk: the ReLU-flip floor (~3e-3)


# ---- GPU: two real ranks run the harness' agg step with the commit forward overlapped; the memory commit is never hidden behind an attribute read -----
_SAVE_PROBE = r'''
import os, sys, torch
out_dir = sys.argv[1]
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
import torch.distributed as dist
dist.init_process_group('gloo', rank=rank, world_size=2)
from pinthememory_amd import checkpoint, dist as D, harness, synth
from pinthememory_amd.network import deepv3plus
crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, crit, crit)).cuda()
net.dsn[3].p = 0.0
net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)
opt, sched = harness.make_optimizer(net)
buckets = D.GradBuckets(net.parameters())
x, y = synth.make_batch(4, 128, seed=5)
x, y = x[2 * rank:2 * rank + 2].cuda(), y[2 * rank:2 * rank + 2].cuda()
assert harness.COMMIT_OVERLAP
for _ in range(2):
    harness.agg_train_step(net, opt, x, y, sched=sched, buckets=buckets)
raised = False
if rank == 0:                      # the reference's save path (train.py:188-191): rank 0 alone reads m_items -> must fail loudly, not issue a collective
    try:
        checkpoint.snapshot_dict(net, opt, sched)
    except RuntimeError as e:
        raised = 'finish_commit' in str(e)
    assert raised
harness.save_checkpoint(os.path.join(out_dir, 'ck.pt'), net, opt, sched, epoch=1)      # every rank: finishes the commit collectively, rank 0 writes
mem = net.memory.m_items.detach().clone()
other = mem.clone()
dist.all_reduce(other)
assert torch.allclose(other, 2 * mem, rtol=0, atol=1e-6), 'the committed memory differs between the ranks'
torch.cuda.synchronize()
if rank == 0:
    ck = torch.load(os.path.join(out_dir, 'ck.pt'), map_location='cpu')
    assert torch.equal(ck['memory'], mem.cpu()) and ck['epoch'] == 1
harness.agg_train_step(net, opt, x, y, sched=sched, buckets=buckets)      # and the loop goes on
harness.finish_commit(net)
assert torch.isfinite(net.memory.m_items).all()
dist.barrier()
dist.destroy_process_group()
print('SAVE_PROBE_DONE', rank)
'''


@pytest.mark.gpu
def test_two_ranks_rank0_only_save_does_not_hide_a_collective(tmp_path):
    """two real ranks (gloo, one GPU), harness.agg_train_step with the commit forward on its own stream and its memory-slot sum
    deferred. Rank 0 alone reading `m_items` (checkpoint.snapshot_dict, the reference's rank-0-only save) raises instead of enqueuing an all-reduce
    nobody answers; harness.save_checkpoint -- called by every rank -- finishes the commit, rank 0 writes the file, both ranks hold the same memory,
    and training continues."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE='2')
        procs.append(subprocess.Popen([sys.executable, '-c', _SAVE_PROBE, str(tmp_path)], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all('SAVE_PROBE_DONE' in o[0] for o in outs), ''.join(o[0][-1500:] + o[1][-3000:] for o in outs)
