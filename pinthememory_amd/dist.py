"""Data-parallel glue over torch.distributed (backend 'nccl' == RCCL over xGMI on ROCm; 'gloo' in CPU tests).

One process per GPU. Three exchanges exist on the hot path:
  C1 gradients      -- flat-bucket all-reduce (mean), the DDP of /root/reference/network/__init__.py:31
  C2 BN statistics  -- per-layer (mean, M2, count) merge == nn.SyncBatchNorm (/root/reference/train.py:95); on GPUs these 130
                       tiny exchanges per step go through a direct RCCL communicator on the compute stream (rccl.py)
  C3 memory slots   -- one all-reduce(SUM) of nominator[20,256] | denominator[20] inside write(); the reference never
                       syncs m_items across ranks -- with it every rank holds the single-process
                       big-batch memory (memory.py:229-230 already sums over the batch).
All functions are no-ops for world_size 1 and work on CPU tensors (gloo) as well as GPU tensors (RCCL).
"""
import os

import torch
import torch.distributed as dist

SYNC_MEMORY = True          # C3 on by default (north star); set False to reproduce the reference's per-rank memory


# PM_DIST_FORCE=1: issue every collective even in a one-rank group -- rehearses the RCCL call sequence (all-gather of BN moments,
# bucketed async all-reduce on the side stream, autograd-aware memory-slot all-reduce) on a 1-GPU box; results are unchanged.
FORCE = os.environ.get('PM_DIST_FORCE', '0') == '1'


# collectives issued through this module since the last reset (bench.py: `config.collectives_per_step`): every all-reduce / all-gather of the
# SyncBN, memory-slot and gradient-bucket paths passes one of the functions below
COLLECTIVES = [0]


def count_collectives(fn):
    """Run fn() (one training step) and return how many collectives this module issued meanwhile (the same number on every rank)."""
    n0 = COLLECTIVES[0]
    fn()
    return COLLECTIVES[0] - n0


def direct_fallback_reason(backend='nccl'):
    """Why the exchanges of this process go through torch.distributed instead of the direct same-stream RCCL communicator (rccl.py)."""
    from . import rccl
    if backend != 'nccl':
        return "backend '%s' is not RCCL (CPU / one-GPU rehearsal): torch.distributed carries every exchange" % backend
    if not rccl.ENABLED:
        return 'PM_DIRECT_RCCL=0'
    return rccl.FALLBACK.get(None) or 'no direct communicator was created (no GPU tensor exchanged yet)'


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def group_size(group=None):
    return dist.get_world_size(group) if is_dist() else 1


def bn_group(bn):
    """Process group a BN layer synchronises over, or None for local statistics."""
    if isinstance(bn, torch.nn.SyncBatchNorm) and bn.training and is_dist():
        return bn.process_group if bn.process_group is not None else dist.group.WORLD
    return None


def _direct(t, group):
    """The same-stream RCCL communicator (rccl.py) for small fp32 GPU tensors; None -> torch.distributed. While the current stream is being captured into a hipGraph
    (harness.GraphedAggStep under N > 1) None is an error: only the direct communicator's calls are plain kernel nodes on the capturing stream."""
    comm = None
    if t.is_cuda and t.is_contiguous() and t.dtype == torch.float32:
        from . import rccl
        comm = rccl.get(group)
    if comm is None and t.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError('a collective on a %s %s tensor would go through torch.distributed inside a hipGraph capture (%s)'
                           % (t.dtype, 'contiguous' if t.is_contiguous() else 'strided', direct_fallback_reason()))
    return comm


def direct_ready(device, group=None):
    """Is the direct same-stream communicator up for `group` on this device (creates it on first use, collectively -- every rank asks at the same point)?"""
    return _direct(torch.zeros(1, dtype=torch.float32, device=device), group) is not None


def all_reduce_sum(t, group=None):
    if is_dist():
        COLLECTIVES[0] += 1
        comm = _direct(t, group)
        if comm is not None:
            comm.all_reduce_sum_(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_reduce_sum_copy(t, group=None):
    """-> (sum over ranks, t untouched). Out of place on the direct communicator (no copy); a clone + in-place all-reduce otherwise."""
    if not is_dist():
        return t
    COLLECTIVES[0] += 1
    comm = _direct(t, group)
    if comm is not None:
        return comm.all_reduce_sum_into(torch.empty_like(t), t)
    out = t.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


def merge_moments_list(parts, c):
    """Chan et al. parallel merge of per-rank (mean[c] | M2[c] | count[c]) -> same layout, exact in real arithmetic."""
    stack = torch.stack(parts)                          # [W, 3c]
    mean, m2, cnt = stack[:, :c].double(), stack[:, c:2 * c].double(), stack[:, 2 * c:].double()
    n = cnt.sum(0)
    gmean = (mean * cnt).sum(0) / n
    gm2 = (m2 + cnt * (mean - gmean) ** 2).sum(0)
    return torch.cat([gmean, gm2, n]).to(parts[0].dtype)


def gather_moments(mom, group=None):
    """All-gather of the per-rank [3c] moments -> (flat float[world * 3c], world)."""
    world = dist.get_world_size(group)
    COLLECTIVES[0] += 1
    flat = torch.empty(world * mom.numel(), dtype=mom.dtype, device=mom.device)
    mom = mom.contiguous()
    comm = _direct(mom, group)
    if comm is not None:
        comm.all_gather_into(flat, mom)
    else:
        dist.all_gather_into_tensor(flat, mom, group=group)
    return flat, world


def merge_moments(mom, c, group=None):
    """One all-gather of [3c] floats per BN layer, then one merge kernel (GPU) / a few torch ops (CPU tensors in the gloo tests)."""
    if not is_dist():
        return mom
    COLLECTIVES[0] += 1
    world = dist.get_world_size(group)
    flat = torch.empty(world * mom.numel(), dtype=mom.dtype, device=mom.device)
    mom = mom.contiguous()
    comm = _direct(mom, group)
    if comm is not None:
        comm.all_gather_into(flat, mom)
    else:
        dist.all_gather_into_tensor(flat, mom, group=group)
    if mom.is_cuda:
        from .hip import kernels as K
        return K.bn_merge(flat, world, c)
    return merge_moments_list(list(flat.view(world, -1)), c)


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x; dx = sum over ranks of dy (each rank's loss depends on every rank's x)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return all_reduce_sum(x.clone().contiguous(), group)

    @staticmethod
    def backward(ctx, g):
        return all_reduce_sum(g.clone().contiguous(), ctx.group), None


def all_reduce_sum_autograd(x, group=None):
    return _AllReduceSum.apply(x, group) if is_dist() else x


class GradBuckets:
    """C1: flat gradient arena + bucketed asynchronous all-reduce(mean).

    Parameters' .grad become views into one flat fp32 buffer (reverse registration order ~ backward order); a
    post-accumulate hook fires a bucket's all-reduce as soon as its last gradient is written. xGMI is point-to-point: few large
    buckets (default 32 MiB) keep every link busy without latency-bound small messages.

    ONE communicator per step. RCCL only guarantees progress across communicators when their kernels start in the same order on every
    rank, which two streams do not promise -- so the buckets never run on a second communicator beside the SyncBN / memory-slot
    exchanges. With the direct same-stream communicator (rccl.py) in use the bucket all-reduce is enqueued on the compute stream
    through it, in program order with every other collective of the step (181 MB over 7 xGMI links is 0.3-2 ms of a ~70 ms step,
    and the weight-gradient side stream keeps the matrix cores busy meanwhile). Without it (gloo tests, PM_DIRECT_RCCL=0, agreed
    fall-back) every collective of the step -- these buckets and the BN / memory exchanges alike -- goes through torch's process
    group, again one communicator; there the bucket runs asynchronously on a side stream.
    """

    def __init__(self, params, bucket_bytes=32 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = group_size(group)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets, self.owner = [], {}
        off, start, pending = 0, 0, []
        for p in reversed(self.params):
            n = p.numel()
            view = self.flat[off:off + n]
            # keep the parameter's memory layout (channels_last conv weights) so grads written by the kernels alias it
            p.grad = view.as_strided(p.shape, p.stride()) if p.is_contiguous() or p.dim() != 4 else \
                view.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
            pending.append(p)
            off += n
            if (off - start) * 4 >= bucket_bytes:
                self._close(start, off, pending)
                start, pending = off, []
        if pending:
            self._close(start, off, pending)
        self.ready = [0] * len(self.buckets)
        self.works = []
        self.stream = torch.cuda.Stream() if dev.type == 'cuda' else None
        if is_dist():
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def _close(self, start, end, plist):
        idx = len(self.buckets)
        self.buckets.append((start, end, len(plist)))
        for p in plist:
            self.owner[p] = idx

    def _hook(self, p):
        b = self.owner[p]
        self.ready[b] += 1
        if self.ready[b] == self.buckets[b][2]:
            self._launch(b)

    def _launch(self, b):
        start, end, _ = self.buckets[b]
        chunk = self.flat[start:end]
        COLLECTIVES[0] += 1
        comm = _direct(chunk, self.group)
        if comm is not None:            # same communicator and stream as the BN / memory exchanges: ordered by the stream itself
            comm.all_reduce_sum_(chunk)
        elif self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def zero(self):
        self.flat.zero_()
        self.ready = [0] * len(self.buckets)

    def finish(self):
        """Call after backward: flush buckets whose hooks never fired (unused params), wait, average."""
        if not is_dist():
            return
        for b, r in enumerate(self.ready):
            if r != self.buckets[b][2]:
                self._launch(b)
        for w in self.works:
            w.wait()
        self.works = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.flat.div_(self.world)
