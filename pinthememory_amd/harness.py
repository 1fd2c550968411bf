"""Callers either side of the hot path, on the GPU: the aggregation train step, memory initialisation, sliding-window
evaluation and mIoU -- the build's counterparts of /root/reference/train.py:284-374 (+ calculate_loss :213-244),
train.py:1000-1042, eval.py:148-274,340-405 and utils/misc.py:65-73."""
import contextlib
import math

import torch
import torch.nn.functional as F

from . import dist as D
from .hip import kernels as K
from .hip import ops

LOSS_W = dict(aux=0.4, read=0.02, div=0.4, cls=0.2)    # train.py:1213-1215 defaults


def make_optimizer(net, lr=0.01, momentum=0.9, poly_exp=9):
    """optimizer.py:11-32: SGD over all named parameters, weight decay hard-coded 5e-4, lr * exp(-poly_exp*it/120000).
    The optimizer is torch.optim.SGD (same state_dict) whose step() is one multi-tensor launch of the HIP library (optim.py);
    it registers the weights it owns with the transformed-filter cache (hip/kernels.py) and bumps their versions when it updates them."""
    from .optim import SGD
    opt = SGD([p for _, p in net.named_parameters()], lr=lr, weight_decay=5e-4, momentum=momentum, nesterov=False)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: math.exp(-1 * poly_exp * it / 120000))
    return opt, sched


def total_loss(outputs, w=LOSS_W):
    main, aux, readloss, writeloss = outputs[0], outputs[1], outputs[-2], outputs[-3]
    return main + w['aux'] * aux + w['read'] * readloss + w['div'] * writeloss[0] + w['cls'] * writeloss[1]


def memory_only_forward(net, x, gts):
    """What the reference's second forward (train.py:330-335) is for: eval-mode features -> memory.write(); the decoder
    and losses it also computes are discarded there, so skipping them leaves every result identical."""
    m = net.module if hasattr(net, 'module') else net
    _, _, feat = m._trunk(x)
    feat = m.aspp(feat)
    if hasattr(m, 'bot_aspp'):
        from .network.deepv3plus import run_cbr
        feat = run_cbr(m.bot_aspp, feat)
    m.memory._mem(feat)
    m.memory.m_items = m.memory.m_items.detach()
    m.memory.write(feat, gts, True)
    from .hip import ops
    ops.flush_bn_counters()


# The memory-commit forward of step t (eval mode, no graph, post-SGD weights) and the training forward of step t + 1 are independent up to the
# memory read of the latter: the commit forward runs on its own stream, and the next step's HBM-bound BatchNorm / transform passes run under its
# GEMMs (and vice versa). Ordering: (1) it starts after the SGD launches; (2) the main stream continues once the commit forward's ONE fold launch has
# read every BatchNorm running moment (ops.last_prefold_event) -- the next training forward updates them; (3) Memory_sup waits for the committed memory
# where it is first read (Memory_sup.pending); (4) transformed filters the commit forward wrote are event-ordered per cache entry (hip/kernels.py);
# (5) the next SGD is behind (3) on the main stream, so the weights are not written under it. Results are bit-identical to the serial order.
# With more than one rank every collective of a step stays on the MAIN stream (dist.py: one communicator, program order): the commit forward's only
# exchange -- the memory-slot sum of its write -- is deferred (Memory_sup.defer_sync) to the next point EVERY rank passes: the next training forward's
# memory read, or an explicit harness.finish_commit(net) -- never to an attribute read (a rank-0-only save would pair it with another rank's
# collective; `m_items` raises while the sum is owed); eval-mode BatchNorm has no exchange. Under the reference's DDP wrapper the commit forward calls
# the bare module (see agg_train_step). PM_COMMIT_OVERLAP=0 disables the overlap.
COMMIT_OVERLAP = __import__('os').environ.get('PM_COMMIT_OVERLAP', '1') == '1'
_commit_streams = {}


COMMIT_CUS = int(__import__('os').environ.get('PM_COMMIT_CUS', '0'))      # A/B knob: confine the commit forward's stream to n CUs (ops.masked_stream)


def _commit_stream(device):
    if device.index not in _commit_streams:
        _commit_streams[device.index] = ops.masked_stream(COMMIT_CUS, device.index, first=1) if COMMIT_CUS > 0 else torch.cuda.Stream(device=device)
    return _commit_streams[device.index]


def finish_commit(net):
    """Finish the memory write of the last commit forward where it was left to the next reader (more than one rank, commit forward on its own
    stream: Memory_sup.defer_sync). COLLECTIVE in that case -- call it on EVERY rank, at the same point of the program, before anything reads
    `m_items` between steps on a subset of ranks (validation, the rank-0-only save of train.py:188-191). The next training forward does it by
    itself (Memory_sup.read). A no-op with one rank or when nothing is pending."""
    m = net.module if hasattr(net, 'module') else net
    if getattr(m, 'memory', None) is not None:
        m.memory.finish_commit()


def sync_commit(net=None):
    """Order the CURRENT stream behind the commit forward of the last agg_train_step (it reads every weight on its own stream). Needed before
    anything WRITES weights outside agg_train_step / mldg_train_step / optim.SGD.step / checkpoint.restore_snapshot, which wait by themselves:
    a foreign optimizer, an EMA update, in-place edits. No collective, no host synchronisation."""
    ops.wait_commit()


_WITHHELD = {}      # id(Memory_sup) -> weakref to the open GraphedAggStep(pipelined=True) whose last commit is withheld (m_items is one commit behind between replays)


def _check_not_withheld(net, what):
    """with GraphedAggStep(pipelined=True) open, `m_items` is one commit behind between replays -- a checkpoint, a validation pass or an mldg step that read it
    would silently see stale memory. Raise instead (as restore_snapshot does for an owed cross-rank commit): read g.committed_memory(), or g.close() first."""
    m = net.module if hasattr(net, 'module') else net
    ref = _WITHHELD.get(id(getattr(m, 'memory', None)))
    g = ref() if ref is not None else None
    if g is not None and not g.closed:
        raise RuntimeError('%s: a GraphedAggStep(pipelined=True) is open on this model -- its last memory commit is withheld until the next replay, so net.memory.m_items is '
                           'one commit behind. Use g.committed_memory() for the up-to-date memory, or g.close() before handing the model to eager code.' % what)


def save_checkpoint(path, net, optimizer=None, scheduler=None, epoch=0, mean_iu=0.0):
    """utils/misc.py:195-216 for the harness: EVERY rank calls it (it finishes a pending memory commit collectively), rank 0 writes the file."""
    from . import checkpoint
    _check_not_withheld(net, 'save_checkpoint')
    finish_commit(net)
    if not D.is_dist() or torch.distributed.get_rank() == 0:
        checkpoint.save_snapshot(path, net, optimizer, scheduler, epoch, mean_iu)


_TRAIN_OVERRIDES = {}      # frozenset of module classes -> does any of them override nn.Module.train()?


def set_mode(net, training):
    """net.train(training) without nn.Module.__setattr__ on each of the ~600 modules (0.5 ms per step over the three toggles of an agg step): the flag lives in
    each module's __dict__. The module tree is walked every time (modules may have been swapped: convert_sync_batchnorm). A tree in which some class overrides
    train() (frozen-BatchNorm variants, user wrappers that hook the mode switch) gets the real net.train(training): checked once per set of classes."""
    mods = list(net.modules())
    classes = frozenset(type(m) for m in mods)
    over = _TRAIN_OVERRIDES.get(classes)
    if over is None:
        over = _TRAIN_OVERRIDES[classes] = any(c.train is not torch.nn.Module.train for c in classes)
    if over:
        return net.train(training)
    for m in mods:
        m.__dict__['training'] = training
    return net


def _all_syncbn(m):
    """Every BatchNorm of the tree is a SyncBatchNorm (train.py:95)? Walked on every call -- only multi-rank steps under the DDP wrapper ask, and a conversion or a
    module swap after the first step must not be answered from a stale cache."""
    bns = [b for b in m.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm)]
    return all(isinstance(b, torch.nn.SyncBatchNorm) for b in bns)


def _commit_forward(net, m, x, gts, aux_gts, mem_t, overlap, dist_on, truncate_second_forward=False, main=None):
    """The memory-commit forward of train.py:330-335: eval mode, no graph, post-SGD weights, starting from the memory `mem_t` the training forward read;
    on its own stream when `overlap` (the caller's stream otherwise). Returns the event recorded at its end (overlap) or None."""
    wrapped = hasattr(net, 'module')
    fwd_net = m if (overlap and wrapped) else net
    main = (torch.cuda.current_stream() if x.is_cuda else None) if main is None else main
    side = _commit_stream(x.device) if overlap else None
    done = None
    if overlap:
        side.wait_stream(main)
        for t in (x, gts, aux_gts, mem_t):
            t.record_stream(side)
    with torch.no_grad(), (torch.cuda.stream(side) if overlap else contextlib.nullcontext()):
        set_mode(net, False)
        m.memory.m_items = mem_t
        m.memory.defer_sync = overlap and dist_on
        ops.last_prefold_event, ops.fold_misses = None, 0
        try:
            if truncate_second_forward:
                memory_only_forward(net, x, gts)
            else:
                fwd_net(x, gts=gts, aux_gts=aux_gts, memory_writing=True)
        finally:
            m.memory.defer_sync = False
        set_mode(net, True)
        if overlap:
            done = side.record_event()
            m.memory.pending = done
            ops.commit_done[x.device.index] = done      # whoever writes weights next waits for it (optim.SGD.step, checkpoint restore, sync_commit)
    if overlap:
        # the next training forward rewrites the BatchNorm running moments: wait for the fold launch that read them (or, without one, for everything)
        main.wait_event(ops.last_prefold_event if (ops.last_prefold_event is not None and ops.fold_misses == 0) else done)
    return done


def agg_train_step(net, opt, x, gts, aux_gts=None, sched=None, buckets=None, truncate_second_forward=False, commit=True):
    """One iteration of train_memory_agg. `buckets` (dist.GradBuckets) replaces DDP's reducer for N > 1.
    commit=False withholds the memory-commit forward (GraphedAggStep(pipelined=True) runs it at the head of the next captured step, beside that step's
    training forward): the returned dict then carries `mem_t`, the memory the commit forward has to start from."""
    aux_gts = gts if aux_gts is None else aux_gts
    m = net.module if hasattr(net, 'module') else net
    set_mode(net, True)
    if x.is_cuda and x.shape[1] == 3:
        # both forward passes of the step read the same batch: lay it out once as the stem's NHWC / 4-channel input
        x = ops.nchw(K.nchw_to_nhwc(x.float(), c_pad=4))
    if buckets is not None:
        buckets.zero()
    else:
        opt.zero_grad()
    outputs = net(x, gts=gts, aux_gts=aux_gts, memory_writing=True, writing_detach=False)
    # train.py:312 clones the memory before the forward; it is re-assigned, never written in place (memory.py:253,256; Memory_sup.write here), so the
    # tensor the read just used IS that clone -- taken after the forward, because with an overlapped commit forward it only becomes final at that read
    mem_t = m.memory.last_read
    loss = total_loss(outputs)
    loss.backward()
    if buckets is not None:
        buckets.finish()
    opt.step()
    out = dict(loss1=outputs[0].detach(), loss2=outputs[1].detach(), readloss=outputs[-2].detach(), div=outputs[-3][0].detach(),
               cls=outputs[-3][1].detach(), total=loss.detach())
    if not commit:
        out['mem_t'] = mem_t
        if sched is not None:
            sched.step()
        return out
    dist_on = D.is_dist()
    # Under the reference's DDP wrapper (network/__init__.py:25-33) the commit forward goes through the bare module: DDP's forward would broadcast the
    # buffers on its own communicator, from a second stream. With every BatchNorm converted to SyncBatchNorm (train.py:95) the running moments are
    # already identical on all ranks, so that broadcast changes nothing; with local BatchNorms it does (rank 0's moments win) and the overlap stays off.
    wrapped = hasattr(net, 'module')
    overlap = COMMIT_OVERLAP and x.is_cuda and (not dist_on or (D.SYNC_MEMORY and (not wrapped or _all_syncbn(m))))
    _commit_forward(net, m, x, gts, aux_gts, mem_t, overlap, dist_on, truncate_second_forward)
    if sched is not None:
        sched.step()
    return out


class GraphedAggStep:
    """One agg train step (train forward + backward + SGD + memory-commit forward) captured in a hipGraph and replayed: ~1 300 kernel launches, their stream forks /
    joins and allocations become ONE graph launch per step. For the regime where the step is launch-bound -- the bf16 tier: 20 ms of Python + HIP enqueue per step
    against 26 ms of GPU time (DESIGN section 7 "Round 4"), i.e. host-bound as soon as the kernels get 20 % faster.

    What makes the step replayable: static input buffers (`x`, `gts` are copied in before each replay), the committed memory lives in one static buffer
    (`Memory_sup.m_items` points at it; the captured step reads it first and writes it last), weights / BatchNorm buffers / momentum buffers are updated in place, the
    learning rate is read from device memory (`optim.SGD.lr_device`, pm_sgd_momentum_multi_dev), random draws (gumbel noise, Dropout2d) go through torch's graph-safe
    generator, and the kept filter transforms are recomputed INSIDE the captured step exactly where an eager step recomputes them (after the SGD), so the
    cache stays consistent with the weights across replays. Single process only.

    pipelined=False: the commit forward runs serially at the END of the captured step (no cross-step overlap: a graph boundary is a join).
    pipelined=True (round 5): the captured unit is [commit forward of step t - 1 on its own stream  ||  training forward of step t] -> backward -> SGD, i.e. the eager
    path's cross-step overlap (worth 1.0-1.8 ms on the bf16 tier) INSIDE one graph: the commit forward of the previous batch (static copies `x_prev`, `gts_prev`, kept by
    the replay itself) forks off at the head of the graph, the training forward joins it where it reads the memory -- the same dependencies the eager overlapped step has
    (BatchNorm running moments: the training forward waits for the commit forward's one fold launch). Every step performs exactly the reference's operations in the
    reference's dependency order; what shifts is only WHEN the commit of step t is executed: at the head of replay t + 1. Hence `Memory_sup.m_items` is one commit
    behind between replays; `committed_memory()` returns the up-to-date memory (it runs the withheld commit forward eagerly into a scratch tensor, leaving the pipeline
    untouched) and `close()` folds it in and hands the model back to eager use.

        g = GraphedAggStep(net, opt, x, gts, sched=sched, pipelined=True)      # warms up eagerly, then captures
        losses = g.step(x, gts)                                  # every further step; tensors of the returned dict are overwritten by the next replay
        g.close()                                                # model, optimizer and memory are consistent for eager code again
    """

    def __init__(self, net, opt, x, gts, sched=None, warmup=3, pipelined=False, buckets=None):
        global COMMIT_OVERLAP
        assert x.is_cuda, 'GraphedAggStep: GPU training only'
        if D.is_dist():
            # N > 1 (round 6): the serial form only, with the build's GradBuckets, and only while EVERY exchange of the step (SyncBN moments / backward sums, the
            # memory-slot sum, the gradient buckets) is an RCCL call enqueued on the capturing stream through the direct communicator (rccl.py): those are
            # ordinary kernel nodes of the graph, replayed in the captured order on every rank. torch.distributed's own process group hops to an internal
            # stream and keeps host-side work objects -- not something to bake into a graph -- so dist.py raises if a capture would reach it.
            assert not pipelined, 'GraphedAggStep: pipelined=True is single-process (the withheld commit would defer a collective across a graph boundary)'
            assert buckets is not None and not hasattr(net, 'module'), 'GraphedAggStep under torch.distributed: pass the bare module and a dist.GradBuckets'
            assert D.direct_ready(x.device), 'GraphedAggStep under torch.distributed needs the direct RCCL communicator: ' + D.direct_fallback_reason()
        self.buckets = buckets
        assert len(opt.param_groups) == 1, 'GraphedAggStep: one parameter group (optimizer.py:21-25)'
        self.net, self.opt, self.sched, self.pipelined, self.closed = net, opt, sched, pipelined, False
        m = self.m = net.module if hasattr(net, 'module') else net
        self.x = ops.nchw(K.nchw_to_nhwc(x.float(), c_pad=4)) if x.shape[1] == 3 else x.clone()      # the stem's NHWC4 layout, converted once per step outside the graph
        self.gts = gts.clone()
        self.lr = torch.zeros(1, dtype=torch.float32, device=x.device)
        prev_overlap, COMMIT_OVERLAP = COMMIT_OVERLAP, False
        ok = False
        try:
            opt.lr_device = self.lr
            for _ in range(warmup):      # allocates workspaces, momentum buffers, filter caches, sets every kernel's LDS attribute
                self.lr.fill_(float(opt.param_groups[0]['lr']))
                agg_train_step(net, opt, self.x, self.gts, sched=sched, buckets=buckets)
            finish_commit(net)
            self.mem = m.memory.m_items.detach().clone()
            m.memory.m_items = self.mem
            if pipelined:
                import weakref
                _WITHHELD[id(m.memory)] = weakref.ref(self)
                # one more eager step whose commit is withheld: the state every replay starts from (weights after SGD t, memory committed through t - 1, batch t in x_prev)
                self.x_prev, self.gts_prev = self.x.clone(), self.gts.clone()
                self.lr.fill_(float(opt.param_groups[0]['lr']))
                agg_train_step(net, opt, self.x, self.gts, sched=sched, commit=False)
                m.memory.m_items = self.mem
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            self.lr.fill_(float(opt.param_groups[0]['lr']))
            with torch.cuda.graph(self.graph):
                if pipelined:
                    main = torch.cuda.current_stream()
                    side = _commit_stream(x.device)
                    self._commit(main, overlap=True)
                    self.out = agg_train_step(net, opt, self.x, self.gts, sched=None, commit=False)
                    self.out.pop('mem_t')
                    main.wait_stream(side)
                    self.x_prev.copy_(self.x)
                    self.gts_prev.copy_(self.gts)
                else:
                    self.out = agg_train_step(net, opt, self.x, self.gts, sched=None, buckets=buckets)
                    self.mem.copy_(m.memory.m_items)
                if ops.overlap_wgrad():      # every stream forked into the capture rejoins it (the weight-gradient stream's last event record trails its last join)
                    torch.cuda.current_stream().wait_stream(ops._side_stream())
            torch.cuda.synchronize()
            # events recorded inside the capture mean nothing to eager code (waiting for one from a non-capturing stream is an error): everything they ordered is complete
            m.memory.pending = None
            m.memory.m_items = self.mem
            ops.commit_done.pop(x.device.index, None)
            ops.last_prefold_event = None
            K.forget_filter_events()
            ok = True
        finally:
            COMMIT_OVERLAP = prev_overlap
            if not ok:      # a failed warm-up / capture must not leave the optimizer reading a dead device scalar or the memory pointing into the capture
                opt.lr_device = None
                m.memory.m_items = (self.mem if getattr(self, 'mem', None) is not None else m.memory._m_items).detach().clone()
                m.memory.pending = None
                ops.commit_done.pop(x.device.index, None)

    def _commit(self, main, overlap, into=None):
        """The withheld commit forward of the previous batch: from the static memory buffer, result copied back into it (or into `into`)."""
        m = self.m
        _commit_forward(self.net, m, self.x_prev, self.gts_prev, self.gts_prev, self.mem, overlap, False, main=main)
        m.memory.pending = None            # dropped, not waited for: the event below (after the copy-back) replaces it
        side = _commit_stream(self.x.device) if overlap else None
        done = None
        with torch.no_grad(), (torch.cuda.stream(side) if overlap else contextlib.nullcontext()):
            (self.mem if into is None else into).copy_(m.memory._m_items)
            if overlap:
                done = side.record_event()
        m.memory.m_items = self.mem
        if overlap:
            m.memory.pending = done       # the training forward's memory read (and the SGD: ops.commit_done) wait for the copy-back
            ops.commit_done[self.x.device.index] = done

    def step(self, x=None, gts=None):
        assert not self.closed, 'GraphedAggStep.step() after close()'
        if x is not None and x.data_ptr() != self.x.data_ptr():
            if x.shape[1] == 3:
                self.x.copy_(ops.nchw(K.nchw_to_nhwc(x.float(), c_pad=4)))
            else:
                self.x.copy_(x)
        if gts is not None and gts.data_ptr() != self.gts.data_ptr():
            self.gts.copy_(gts)
        self.lr.fill_(float(self.opt.param_groups[0]['lr']))
        self.graph.replay()
        if self.sched is not None:
            self.sched.step()
        return self.out

    def committed_memory(self):
        """The memory as of the last step INCLUDING its commit. pipelined: the withheld commit forward is run eagerly into a scratch tensor (the pipeline state --
        static memory buffer, x_prev -- is left as it is, so replays may continue); otherwise the static buffer itself."""
        if not self.pipelined or self.closed:
            return self.m.memory.m_items
        out = torch.empty_like(self.mem)
        self._commit(torch.cuda.current_stream(), overlap=False, into=out)      # reads the static buffer, writes `out`; m_items points at the static buffer again
        return out

    def close(self):
        """Hand model and optimizer back to eager code: the withheld commit (pipelined) is folded into `m_items`, the optimizer forgets the device-side learning
        rate of the capture (an eager opt.step() after a graphed phase silently used the LR of the last replay), graph-internal events are dropped."""
        if self.closed:
            return
        torch.cuda.synchronize()
        if self.pipelined:
            self._commit(torch.cuda.current_stream(), overlap=False)
            _WITHHELD.pop(id(self.m.memory), None)
        self.m.memory.m_items = self.mem.clone()
        self.m.memory.pending = None
        ops.commit_done.pop(self.x.device.index, None)
        if getattr(self.opt, 'lr_device', None) is self.lr:
            self.opt.lr_device = None
        self.closed = True
        torch.cuda.synchronize()


def memory_initialize(net, batches, epochs=2):
    """Class-prototype initialisation: sum of normalised bot_aspp features per soft class / count (no writenet).
    The reference builds a one-hot + F.interpolate per batch (train.py:1020-1030); here it is the write kernel's
    4-tap gather with normalize=1, summed across batches (and ranks) before the division."""
    m = net.module if hasattr(net, 'module') else net
    mem = m.memory
    net.eval()
    acc = None
    with torch.no_grad():
        for _ in range(epochs):
            for x, gt in batches:
                feat = net(x, gts=gt, aux_gts=gt)[-1]
                nomden = K.mem_write_accum(K.cast(ops.nhwc(feat), torch.float32), gt.contiguous(), mem.memory_size, normalize=True)
                acc = nomden if acc is None else acc + nomden
        acc = D.all_reduce_sum(acc)
        s, d = mem.memory_size, mem.feature_dim
        basket = acc[:(s + 1) * d].view(s + 1, d)[:s]
        count = acc[(s + 1) * d:][:s].clone().unsqueeze(1)
        count[count == 0] = 1
        mem.m_items = F.normalize(basket / count, dim=1)
    net.train()
    return mem.m_items


def sliding_tiles(h, w, crop, overlap=1.0 / 3, scale=1.0):
    """eval.py:158-182: [(x1,y1,x2,y2)] in the reference's order."""
    tile = int(crop * max(scale, 1.0))
    stride = math.ceil(tile * (1 - overlap))
    rows = int(math.ceil((w - tile) / stride) + 1)
    cols = int(math.ceil((h - tile) / stride) + 1)
    out = []
    for r in range(rows):
        for c in range(cols):
            x2, y2 = min(int(r * stride) + tile, w), min(int(c * stride) + tile, h)
            out.append((max(int(x2 - tile), 0), max(int(y2 - tile), 0), x2, y2))
    return out


def sliding_logits(net, img, crop, overlap=1.0 / 3, flips=(False, True), batch_tiles=True):
    """Single-scale sliding-window logits for one CHW image, stitched on the GPU in float64 (the reference stitches
    per class in numpy threads, eval.py:210-274). Logits (not probabilities) are averaged (eval.py:380-392); the count
    divides by the true per-pixel tile count (the reference's count array is mis-indexed but class-uniform)."""
    c, h, w = img.shape
    tiles = sliding_tiles(h, w, crop, overlap)
    net.eval()
    acc = None
    with torch.no_grad():
        for flip in flips:
            src = torch.flip(img, dims=[2]) if flip else img
            crops = torch.stack([src[:, y1:y2, x1:x2] for (x1, y1, x2, y2) in tiles])
            outs = net(crops)[0] if batch_tiles else torch.cat([net(cr[None])[0] for cr in crops])   # eval.py:379-390 (--faster batches)
            if len(tiles) <= STITCH_MAX_TILES and outs.is_cuda:
                # one kernel per flip: float64 sum of the covering tiles (tile order) / count, written un-flipped into the accumulator
                acc = K.sliding_stitch(ops.nhwc(outs), tiles, h, w, flip, acc)
            else:
                acc = _stitch_torch(outs, tiles, h, w, flip, acc)
    return acc / len(flips)


STITCH_MAX_TILES = 64      # pm_sliding_stitch takes the tile table by value in its kernel arguments


def _stitch_torch(outs, tiles, h, w, flip, acc):
    """The same stitching in stock float64 torch ops for what the kernel does not take: more than 64 tiles per image (a 1024 x 2048 image at
    crop 256 has 72) -- same sums in the same tile order, the same division by the true count, the same un-flip."""
    full = torch.zeros(outs.shape[1], h, w, dtype=torch.float64, device=outs.device)
    cnt = torch.zeros(1, h, w, dtype=torch.float64, device=outs.device)
    for lg, (x1, y1, x2, y2) in zip(outs, tiles):
        full[:, y1:y2, x1:x2] += lg.to(torch.float64)
        cnt[:, y1:y2, x1:x2] += 1
    full = full / cnt
    if flip:
        full = torch.flip(full, dims=[2])
    return full if acc is None else acc + full


def fast_hist(pred, gt, n=19):
    """utils/misc.py:65-70 as one device bincount."""
    pred, gt = pred.reshape(-1), gt.reshape(-1)
    k = (gt >= 0) & (gt < n)
    return torch.bincount(n * gt[k].long() + pred[k].long(), minlength=n * n).view(n, n)


def miou(hist):
    hist = hist.double()
    iu = torch.diag(hist) / (hist.sum(1) + hist.sum(0) - torch.diag(hist))
    return float(torch.nanmean(iu)), iu


# ---- the meta-learning regime every pinmem script runs: train_memory_mldg (train.py:493-632) ----------------------------
def put_theta(model, theta):
    """train.py:262-277: rewire every leaf module's _parameters with (non-leaf) tensors; the HIP ops take weights as
    autograd inputs at call time, so functional parameters work without touching the modules."""
    def walk(mod, name=None):
        if len(mod._modules) != 0:
            for k, v in mod._modules.items():
                walk(v, str(k) if name is None else str(name + '.' + k))
        else:
            for k, v in mod._parameters.items():
                if isinstance(v, torch.Tensor):
                    mod._parameters[k] = theta[str(name + '.' + k)]
    walk(model)
    return model


def functional_theta(old, lr):
    """train.py:246-260: theta' = theta - lr * grad (first order) for every parameter that has a gradient, the detached value otherwise (the reference takes the
    state_dict entry there). The ~160 multiply / subtract pairs are two multi-tensor launches (same arithmetic, same bits as `p - lr * p.grad` one by one): 640 tiny
    launches and ~6 ms of host time per mldg step less. Only parameters: put_theta rewires nothing else."""
    names, ps = [], []
    for k, p in old.named_parameters():
        names.append(k)
        ps.append(p)
    idx = [i for i, p in enumerate(ps) if p.grad is not None]
    # lr: a python float, or a 0-dim fp32 device tensor (GraphedMldgStep: the annealed inner rate changes between replays) -- the same fp32 multiply either way
    upd = torch._foreach_sub([ps[i] for i in idx], torch._foreach_mul([ps[i].grad for i in idx], lr)) if idx else []
    theta = {k: p.detach() for k, p in zip(names, ps)}
    for i, u in zip(idx, upd):
        theta[names[i]] = u
    return theta


def get_updated_network(old, new, lr, theta=None):
    """train.py:246-260: `new` rewired with theta' = theta - lr * grad of `old` (first order)."""
    return put_theta(new, functional_theta(old, lr) if theta is None else theta)


INNER_LR = 0.001      # train.py:1208 `--inner_lr` default


def annealed_inner_lr(opt):
    """train.py:625-626 (`--inner_lr_anneal`, passed by every pinmem script: train_GS_pinmem_DR50V3P.sh:18): after each scheduler step the inner
    learning rate of the NEXT iteration is a quarter of the outer one."""
    return opt.param_groups[-1]['lr'] / 4


def mldg_train_step(net, updated_net, updated_net2, opt, x_tr, y_tr, x_te, y_te, inner_lr=INNER_LR, sched=None, inner_lr_anneal=False):
    """One iteration of train_memory_mldg (memory configuration, whitening off): inner step on the meta-train domains,
    frozen-encoder memory write with the stepped weights, read-only meta-test forward whose loss back-propagates through the
    WRITTEN memory into the write graph (memory.py:323-324 only detaches when writing), outer step, memory commit.
    inner_lr: the reference's default 1e-3 (train.py:1208); with inner_lr_anneal the returned dict carries `next_inner_lr` = lr / 4 of the
    outer schedule after this step (train.py:625-626) for the caller to pass into the next iteration."""
    _check_not_withheld(net, 'mldg_train_step')
    set_mode(net, True)
    finish_commit(net)          # every rank is here: a memory commit deferred by a preceding agg step is finished before m_items is read
    mem_t = net.memory.m_items.clone().detach()
    opt.zero_grad()
    out_in = net(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True, writing_detach=False)
    inner = total_loss(out_in)
    inner.backward(retain_graph=True)
    # both functional networks carry the same theta' (train.py:543,546 build it twice from the same gradients); the second one's encoder is frozen (train.py:549-552):
    # its non-memory entries are the detached values
    theta = functional_theta(net, inner_lr)
    updated_net = set_mode(get_updated_network(net, updated_net, inner_lr, theta), True)
    theta2 = {k: (v if k.split('.')[0] == 'memory' else v.detach()) for k, v in theta.items()}
    updated_net2 = set_mode(get_updated_network(net, updated_net2, inner_lr, theta2), True)
    updated_net2.memory.m_items = mem_t
    updated_net2(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True, writing_detach=False)
    updated_net.memory.m_items = updated_net2.memory.m_items.clone()
    out_te = updated_net(x_te, gts=y_te, aux_gts=y_te, memory_writing=False)
    outer = out_te[0] + LOSS_W['aux'] * out_te[1] + LOSS_W['read'] * out_te[-2]
    outer.backward()
    opt.step()
    with torch.no_grad():
        set_mode(net, False)
        net.memory.m_items = mem_t
        net(x_tr, gts=y_tr, aux_gts=y_tr, memory_writing=True)
        set_mode(net, True)
    if sched is not None:
        sched.step()
    out = dict(inner=inner.detach(), outer=outer.detach(), inner_loss1=out_in[0].detach(), outer_loss1=out_te[0].detach(),
               outer_read=out_te[-2].detach())
    if inner_lr_anneal:
        out['next_inner_lr'] = annealed_inner_lr(opt)
    return out


def release_functional_state(net, *functional_nets):
    """Drop every reference the harness objects hold into the autograd graph of an earlier iteration: the functional networks' theta' (rewired to detached views of
    `net`'s parameters), the memories written without detaching, the deferred weight proxies of the last forward. See GraphedMldgStep.__init__ for why a capture needs it."""
    import gc
    plain = {k: p.detach() for k, p in net.named_parameters()}
    for u in functional_nets:
        put_theta(u, plain)
    for m in (net,) + tuple(functional_nets):
        mem = getattr(m, 'memory', None)
        if mem is not None:
            mem._m_items = mem._m_items.detach()
            mem.last_read = None
    ops._proxy.clear()
    gc.collect()


class GraphedMldgStep:
    """One train_memory_mldg iteration (train.py:493-632: inner forward + backward with retain_graph, the two functional weight sets theta' = theta - inner_lr g,
    frozen-encoder memory write, meta-test forward + backward through the written memory, SGD, eval-mode memory-commit forward) captured in ONE hipGraph and
    replayed. The regime every pinmem script runs (train_GS_pinmem_DR50V3P.sh:9-10,18) enqueues ~2 400 launches per step: on the bf16 tier 33-35 ms of host time
    against 30 ms of GPU time -- host-bound as eager launches. What makes it replayable on top of GraphedAggStep's list: the inner learning rate is a device
    scalar (`--inner_lr_anneal` changes it every iteration: train.py:625-626), the functional networks are rewired once at capture (their theta' tensors live in
    the graph's memory pool and are recomputed by every replay), the committed memory lives in one static buffer that the captured step reads first and writes last.
    Bit-identical to eager mldg_train_step calls (tests/test_model_parity.py::test_graphed_mldg_step_is_bit_identical_to_eager). Single process only.

        g = GraphedMldgStep(net, u1, u2, opt, x_tr, y_tr, x_te, y_te, sched=sched, inner_lr_anneal=True)      # warms up eagerly, then captures
        out = g.step(x_tr, y_tr, x_te, y_te)      # tensors of the returned dict are overwritten by the next replay; out['next_inner_lr'] as mldg_train_step
        g.close()
    """

    def __init__(self, net, updated_net, updated_net2, opt, x_tr, y_tr, x_te, y_te, inner_lr=INNER_LR, sched=None, warmup=2, inner_lr_anneal=False):
        assert x_tr.is_cuda and not D.is_dist(), 'GraphedMldgStep: single-process GPU training only'
        assert len(opt.param_groups) == 1, 'GraphedMldgStep: one parameter group (optimizer.py:21-25)'
        self.net, self.u1, self.u2, self.opt, self.sched, self.anneal, self.closed = net, updated_net, updated_net2, opt, sched, inner_lr_anneal, False
        self.x_tr, self.y_tr, self.x_te, self.y_te = x_tr.clone(), y_tr.clone(), x_te.clone(), y_te.clone()
        dev = x_tr.device
        self.lr = torch.zeros(1, dtype=torch.float32, device=dev)
        self.inner = torch.zeros((), dtype=torch.float32, device=dev)      # 0-dim: torch._foreach_mul's tensor-scalar overload
        self.inner_lr = float(inner_lr)
        # Warm-up and capture run on ONE side stream, after every reference into an earlier autograd graph has been dropped. An autograd AccumulateGrad node remembers the
        # stream it was created on and lives as long as any graph refers to it; the functional networks keep theta' -- hence the graph of the inner step, hence every
        # parameter's node -- alive BETWEEN iterations. A node born in eager code on the default stream would be reused inside the capture and enqueue its accumulation on
        # that (non-capturing) stream: hipStreamEndCapture crashes (tools/mldg_graph_probe.py: the bisection). GraphedAggStep never meets this: its graphs die with each step.
        self.stream = torch.cuda.Stream(device=dev)
        ok = False
        try:
            opt.lr_device = self.lr
            finish_commit(net)
            release_functional_state(net, updated_net, updated_net2)
            torch.cuda.synchronize()
            with torch.cuda.stream(self.stream):
                for _ in range(max(1, warmup)):      # allocates workspaces, momentum buffers, filter caches, sets every kernel's LDS attribute; >= 1: the nodes are reborn on this stream
                    self._eager()
                self.mem = net.memory.m_items.detach().clone()
                net.memory.m_items = self.mem
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            self._fill()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.out = mldg_train_step(net, updated_net, updated_net2, opt, self.x_tr, self.y_tr, self.x_te, self.y_te, inner_lr=self.inner, sched=None)
                self.mem.copy_(net.memory.m_items)
                if ops.overlap_wgrad():      # every stream forked into the capture rejoins it
                    torch.cuda.current_stream().wait_stream(ops._side_stream())
            torch.cuda.synchronize()
            net.memory.pending = None
            net.memory.m_items = self.mem
            ops.commit_done.pop(dev.index, None)
            ops.last_prefold_event = None
            K.forget_filter_events()
            ok = True      # (a capture records, it does not execute: the model is where the warm-up left it)
        finally:
            if not ok:      # a failed warm-up / capture must not leave the optimizer reading a dead device scalar
                opt.lr_device = None
                if getattr(self, 'mem', None) is not None:
                    net.memory.m_items = self.mem.clone()
                net.memory.pending = None

    def _fill(self):
        self.lr.fill_(float(self.opt.param_groups[0]['lr']))
        self.inner.fill_(self.inner_lr)

    def _after(self):
        if self.sched is not None:
            self.sched.step()
        if self.anneal:
            self.inner_lr = annealed_inner_lr(self.opt)

    def _eager(self):
        self._fill()
        mldg_train_step(self.net, self.u1, self.u2, self.opt, self.x_tr, self.y_tr, self.x_te, self.y_te, inner_lr=self.inner, sched=None)
        self._after()

    def step(self, x_tr=None, y_tr=None, x_te=None, y_te=None):
        assert not self.closed, 'GraphedMldgStep.step() after close()'
        for dst, src in ((self.x_tr, x_tr), (self.y_tr, y_tr), (self.x_te, x_te), (self.y_te, y_te)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        self._fill()
        self.graph.replay()
        self._after()
        out = dict(self.out)
        if self.anneal:
            out['next_inner_lr'] = self.inner_lr
        return out

    def close(self):
        """Hand model and optimizer back to eager code (the memory becomes a private tensor again, the optimizer forgets the capture's device-side learning rate)."""
        if self.closed:
            return
        torch.cuda.synchronize()
        self.net.memory.m_items = self.mem.clone()
        self.net.memory.pending = None
        ops.commit_done.pop(self.x_tr.device.index, None)
        if getattr(self.opt, 'lr_device', None) is self.lr:
            self.opt.lr_device = None
        release_functional_state(self.net, self.u1, self.u2)      # eager code gets fresh autograd nodes on ITS stream (the captured ones were born on the capture stream)
        self.closed = True
        torch.cuda.synchronize()


# ---- pooled multi-scale / flip evaluation: inference_pool + MeanFusion (eval.py:133-145,277-337) -------------------------
def inference_pool(net, imgs, orisize, no_flip=False):
    """imgs[flip][scale]: pre-scaled (and pre-flipped for flip = 1) NCHW image batches as the reference's loader provides them.
    Per entry: logits -> half-pixel bilinear to `orisize` (un-flipping flipped inputs) -> softmax -> float64 running mean;
    returns (probs, preds) = max over classes of the fused buffer. All of it stays on the GPU (the reference moves every
    map to the CPU, eval.py:328-330)."""
    net.eval()
    n = imgs[0][0].shape[0]
    buf, cnt = None, 0
    with torch.no_grad():
        for flip in range(1 if no_flip else 2):
            for img in imgs[flip]:
                lg = ops.nhwc(net(img)[0])
                up = K.resize_hp_fwd(lg, orisize, flip_w=(flip == 1))
                if buf is None:
                    buf = torch.zeros((n, orisize[0], orisize[1], up.shape[3]), dtype=torch.float64, device=up.device)
                cnt += 1
                K.softmax_mean_update(up, buf, cnt)
    return K.argmax_f64(buf)


# ---- input edge -----------------------------------------------------------------------------------
def prepare_batch(inputs, gts, aux_gts=None):
    """train.py:297-307: DomainUniformConcatDataset batches arrive as [B, D, 3, H, W] / [B, D, H, W]; merge the domain axis."""
    c, h, w = inputs.shape[-3:]
    x = inputs.reshape(-1, c, h, w).cuda(non_blocking=True)
    gt = gts.reshape(-1, h, w).cuda(non_blocking=True)
    aux = gt if aux_gts is None else aux_gts.reshape(-1, h, w).cuda(non_blocking=True)
    return x, gt, aux


def prepare_batch_u8(images_u8, labels_u8):
    """GPU-side ToTensor + Normalize(ImageNet) + MaskToTensor for uint8 [.., H, W, 3] images and uint8 [.., H, W] label maps:
    4x / 8x less host->device traffic than the fp32 / int64 tensors the reference's loader ships. Returns the image as a
    logical-NCHW view of NHWC4 memory (what the stem consumes without another conversion) and int64 labels."""
    h, w = images_u8.shape[-3:-1]
    img = K.image_u8_to_nhwc4(images_u8.reshape(-1, h, w, 3).cuda(non_blocking=True).contiguous())
    lab = K.labels_u8_to_i64(labels_u8.reshape(-1, h, w).cuda(non_blocking=True).contiguous())
    return ops.nchw(img), lab
