"""Categorical memory (read / write / losses) on the HIP kernels. Same class surface, parameter names, init and
m_items handling as /root/reference/network/memory.py (Memory_sup :94-361, Writingnet :67-87)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import dist as D
from ..hip import kernels as K
from ..hip import ops


def initialize_weights(*models):                      # memory.py:9-19
    for model in models:
        for m in model.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight.data, nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1.)
                m.bias.data.fill_(1e-4)
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, 0.0001)
                m.bias.data.zero_()


class Writingnet(nn.Module):
    def __init__(self, input_feature_dim, feature_dim):
        super().__init__()
        assert input_feature_dim == feature_dim, "Should match when residual mode is on ({} != {})".format(input_feature_dim, feature_dim)
        self.writefeat = nn.Sequential(nn.Conv2d(input_feature_dim, feature_dim, kernel_size=1, stride=1, bias=False),
                                       nn.BatchNorm2d(feature_dim))
        self.relu = nn.ReLU(inplace=True)
        initialize_weights(self)

    def forward(self, x):                             # relu(x + BN(conv1x1(x))), memory.py:83-87
        return ops.conv_bn_act(x, self.writefeat[0], self.writefeat[1], relu=True, residual=x)


# Memory_sup -> event after which its m_items is complete, when the tensor was produced on another stream (harness.agg_train_step runs the commit
# forward on its own). Kept OUTSIDE the module: the reference's callers deep-copy networks (train.py:246-277 get_updated_network) and pickle them, and a
# stream event is neither.
_PENDING = __import__('weakref').WeakKeyDictionary()
# Memory_sup -> (memory the write started from, local nominator | denominator) of a write whose cross-rank sum was deferred (Memory_sup.defer_sync): the
# commit forward of a multi-rank step runs on its own stream and must not issue a collective there; whoever reads m_items next finishes the write on
# ITS stream -- all-reduce + momentum update, the same two launches write() would have issued -- so every collective of a step stays on one stream.
_DEFERRED = __import__('weakref').WeakKeyDictionary()


class Memory_sup(nn.Module):
    def __init__(self, memory_size, input_feature_dim, feature_dim, momentum, temperature, gumbel_read):
        super().__init__()
        self.memory_size = memory_size
        self.feature_dim = feature_dim
        self.momentum = momentum
        self.initial_momentum = momentum
        self.temperature = temperature
        self.output = nn.Sequential(nn.Conv2d(feature_dim * 2, input_feature_dim, kernel_size=1, stride=1, bias=False),
                                    nn.BatchNorm2d(input_feature_dim), nn.ReLU(inplace=True))
        self.writenet = Writingnet(input_feature_dim, feature_dim)
        self.mem_cls = torch.arange(memory_size)
        self.clsfier = nn.Linear(in_features=feature_dim, out_features=memory_size, bias=True)
        self.celoss = nn.CrossEntropyLoss(ignore_index=255)
        self.gumbel_read = gumbel_read
        self.writeTF = lambda x: x.clone()
        self.defer_sync = False   # harness: leave the cross-rank sum of a no-grad write to the next reader of m_items (see _DEFERRED)
        self.last_read = None     # the memory tensor the latest read() used (harness: what the commit forward starts from)
        self.m_items = F.normalize(torch.rand((memory_size, feature_dim), dtype=torch.float), dim=1)
        initialize_weights(self)
        self.noise_fn = None      # parity hook: callable(rows, slots, device) -> (noise_dim0, noise_dim1)

    # m_items stays the reference's plain get / set attribute (train.py:312,332,547,558,580,1040; optimizer.py:65). Behind it: when the tensor was
    # produced on another stream (harness.agg_train_step runs the commit forward on its own), the first READ from anywhere -- the next memory read,
    # a checkpoint, a test -- orders the reader's stream behind the producer; harness code that only passes the tensor on uses _m_items.
    @property
    def pending(self):
        return _PENDING.get(self)

    @pending.setter
    def pending(self, event):
        if event is None:
            _PENDING.pop(self, None)
        else:
            _PENDING[self] = event

    @property
    def m_items(self):
        if self in _DEFERRED:
            # A cross-rank sum is still owed to this memory (harness.agg_train_step with more than one rank). An attribute read must never hide a
            # collective: a rank-0-only reader (the reference saves on rank 0, train.py:188-191 -> utils/misc.py:214) would pair its all-reduce with
            # whatever the other ranks issue next. The write is finished where EVERY rank passes: the next forward's read(), or harness.finish_commit(net).
            raise RuntimeError('Memory_sup.m_items: the memory-slot all-reduce of the last commit forward is still pending. Call '
                               'pinthememory_amd.harness.finish_commit(net) on EVERY rank (harness.save_checkpoint does) before reading m_items between steps.')
        ev = _PENDING.pop(self, None)
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            if self._m_items.is_cuda:
                self._m_items.record_stream(cur)
        return self._m_items

    @m_items.setter
    def m_items(self, value):
        ev = _PENDING.pop(self, None)
        if ev is not None:      # the tensor being replaced may still be written on another stream: order this stream behind it instead of forgetting the event
            torch.cuda.current_stream().wait_event(ev)
        _DEFERRED.pop(self, None)
        self._m_items = value

    @property
    def commit_owed(self):
        """True while the cross-rank sum of a deferred commit write is pending (more than one rank): `m_items` raises, assignment would drop it on this rank only."""
        return self in _DEFERRED

    def finish_commit(self):
        """COLLECTIVE when a deferred write is pending (Memory_sup.defer_sync): all-reduce of the local nominator | denominator and the momentum update,
        on the caller's stream -- the two launches write() would have issued. Every rank must call it at the same point of its program: read() does
        (the next forward, which every rank runs), harness.finish_commit(net) does it explicitly before validation / saving. No-op otherwise."""
        d = _DEFERRED.pop(self, None)
        if d is None:
            return
        ev = _PENDING.pop(self, None)
        cur = torch.cuda.current_stream() if d[1].is_cuda else None
        if ev is not None:
            cur.wait_event(ev)
        mem, nomden = d
        if cur is not None:
            nomden.record_stream(cur), mem.record_stream(cur)
        with torch.no_grad():
            self._m_items = K.mem_write_update(mem, D.all_reduce_sum(nomden), self.momentum)[0]

    def _apply(self, fn, *args, **kwargs):
        # m_items / mem_cls are plain attributes in the reference (hard .cuda() at memory.py:111,120); follow the module instead
        super()._apply(fn, *args, **kwargs)
        self.m_items = fn(self.m_items)
        self.mem_cls = fn(self.mem_cls)
        return self

    def _mem(self, like):
        if self.m_items.device != like.device:
            self.m_items = self.m_items.to(like.device)
            self.mem_cls = self.mem_cls.to(like.device)
        return self.m_items

    def _gumbel(self, rows, device):
        if not self.gumbel_read:
            return None, None
        if self.noise_fn is not None:
            return self.noise_fn(rows, self.memory_size, device)
        # F.gumbel_softmax (memory.py:183-184): g = -log(Exp(1)), tau = 1; two independent draws
        g0 = -torch.empty(rows, self.memory_size, device=device).exponential_().log()
        g1 = -torch.empty(rows, self.memory_size, device=device).exponential_().log()
        return g0, g1

    def get_score(self, query, mask, mem):            # memory.py:167-189; query NHWC, normalised by the caller
        bs, h, w, d = query.size()
        g0, g1 = self._gumbel(bs * h * w, query.device)
        _, score, pmem, pq = ops.mem_read_pq(ops.cast(query.permute(0, 3, 1, 2), torch.float32), mem, g1, g0)
        readloss = ops.upsample_ce(score.permute(0, 3, 1, 2), mask, 1.0 / self.temperature) if mask is not None else 0
        return pq.reshape(bs * h * w, -1), pmem.reshape(bs * h * w, -1), readloss

    def read(self, query, mask, memory_writing):      # memory.py:317-336
        b, d, h, w = query.size()
        self.finish_commit()      # every rank reads here: the one place a deferred cross-rank sum is finished implicitly
        mem = self._mem(query)
        if memory_writing:
            self.m_items = mem = mem.detach()
        self.last_read = mem.detach()
        g0, g1 = self._gumbel(b * h * w, query.device)
        qr, score, pmem, pq = ops.mem_read_pq(ops.cast(query, torch.float32), mem, g1, g0)      # the memory works in fp32 (a no-op off the bf16 tier)
        readloss = ops.upsample_ce(score.permute(0, 3, 1, 2), mask, 1.0 / self.temperature) if mask is not None else 0
        updated_query = ops.conv_bn_act(qr, self.output[0], self.output[1], relu=True)
        return updated_query, pq, pmem, readloss

    def write(self, input, mask, writing_detach=True):  # memory.py:206-257
        self.finish_commit()
        mem = self._mem(input)
        z = self.writenet(input)
        nomden = ops.mem_write_accum(ops.cast(z, torch.float32), mask, self.memory_size)
        if self.defer_sync and D.SYNC_MEMORY and D.is_dist() and not torch.is_grad_enabled():
            _DEFERRED[self] = (mem.detach(), nomden)      # finished by the next reader of m_items, on its stream
            return [0, 0]
        if D.SYNC_MEMORY:
            nomden = D.all_reduce_sum_autograd(nomden)
        updated_memory = ops.mem_write_update(mem.detach(), nomden, self.momentum)
        writing_loss = [self.diversityloss(updated_memory), self.classification_loss(updated_memory)]
        self.m_items = updated_memory.detach() if writing_detach else updated_memory
        return writing_loss

    def classification_loss(self, mem):               # memory.py:259-262
        return self.celoss(self.clsfier(mem), self.mem_cls)

    def diversityloss(self, mem):                     # memory.py:264-272
        cos_sim_pos = torch.matmul(mem, torch.t(mem)).clamp_min(0)    # == cos[cos < 0] = 0 without the host sync of a mask write
        # the trace as the sum of the diagonal view: torch.trace's BACKWARD (index_fill_ with a tensor value -> .item()) synchronises the host with the GPU at the
        # start of every backward pass -- the host loses its lead over the GPU each step (round 4: the bf16 tier is host-bound) and the step cannot be captured
        return (torch.sum(cos_sim_pos) - cos_sim_pos.diagonal().sum()) / (self.memory_size * (self.memory_size - 1))

    def forward(self, query, mask=None, memory_writing=True, writing_detach=True):   # memory.py:191-204
        updated_query, softmax_score_query, softmax_score_memory, readloss = self.read(query, mask, memory_writing)
        writeloss = self.write(query, mask, writing_detach) if memory_writing else [0, 0]
        return updated_query, softmax_score_query, softmax_score_memory, readloss, writeloss
