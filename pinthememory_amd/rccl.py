"""Direct RCCL communicator for the small, latency-bound collectives of the data-parallel path (the
per-layer SyncBatchNorm moments and backward sums, 65 + 65 exchanges of a few KB per training step).

torch.distributed's ProcessGroupNCCL runs every collective on its own internal stream: compute stream -> event -> RCCL stream ->
event -> compute stream. On MI355X each such hop costs ~20-25 us of idle GPU, i.e. ~7 ms of a 71 ms step for the 137 collectives
(measured with a one-rank group, `PM_DIST_FORCE=1 python tools/cpu_enqueue_time.py`: 68.7 -> 76.0 ms/step). Here the RCCL call is
enqueued **on the stream the producing / consuming kernels run on** (`ncclAllGather(..., stream)`), so the exchange is ordered by
the stream itself and costs only its own kernel.

The library is the librccl.so torch already loaded (one RCCL instance per process); the communicator is bootstrapped through the
existing torch.distributed group (rank 0's ncclUniqueId is broadcast over it). Any failure falls back to torch.distributed on every
rank (the ranks agree on that with one all-reduce), so the exchange itself never depends on this module.
`PM_DIRECT_RCCL=0` disables it.
"""
import ctypes
import os
import time
import warnings

import torch
import torch.distributed as dist

ENABLED = os.environ.get('PM_DIRECT_RCCL', '1') == '1'
NCCL_FLOAT32, NCCL_FLOAT64, NCCL_SUM = 7, 8, 0          # rccl.h: ncclDataType_t / ncclRedOp_t


class _UniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]        # NCCL_UNIQUE_ID_BYTES


_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), mode=ctypes.RTLD_GLOBAL)
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        lib.ncclCommAbort.argtypes = [ctypes.c_void_p]
        lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed: %s' % (what, _load().ncclGetErrorString(rc).decode()))


def _dtype(t):
    if t.dtype == torch.float32:
        return NCCL_FLOAT32
    if t.dtype == torch.float64:
        return NCCL_FLOAT64
    raise TypeError('direct RCCL path carries fp32 / fp64 only, got %s' % t.dtype)


class DirectComm:
    """One RCCL communicator over the ranks of a torch.distributed group, used from whatever stream is current."""

    def __init__(self, group=None):
        lib = _load()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        uid = _UniqueId()
        box = [None]                  # None = "rank 0 could not create the id": EVERY rank then raises and reaches the agreed fall-back of get()
        if self.rank == 0:
            rc = lib.ncclGetUniqueId(ctypes.byref(uid))
            box = [ctypes.string_at(ctypes.addressof(uid), 128) if rc == 0 else None]     # all 128 bytes (a c_char field read stops at the first NUL)
        if self.world > 1:            # always broadcast, also after a rank-0 failure: no rank may be left waiting in it
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0] is None:
            raise RuntimeError('ncclGetUniqueId failed on rank 0')
        assert len(box[0]) == 128
        ctypes.memmove(ctypes.addressof(uid), box[0], 128)
        self.comm = ctypes.c_void_p()
        _check(lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), 'ncclCommInitRank')

    def all_reduce_sum_(self, t):
        assert t.is_cuda and t.is_contiguous()
        _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _dtype(t), NCCL_SUM, self.comm,
                                     torch.cuda.current_stream(t.device).cuda_stream), 'ncclAllReduce')
        return t

    def all_reduce_sum_into(self, out, t):
        """out = sum over ranks of t; t is left untouched (the SyncBN backward keeps its rank-local sums without a copy)."""
        assert t.is_cuda and t.is_contiguous() and out.is_contiguous() and out.numel() == t.numel() and out.dtype == t.dtype
        _check(_load().ncclAllReduce(t.data_ptr(), out.data_ptr(), t.numel(), _dtype(t), NCCL_SUM, self.comm,
                                     torch.cuda.current_stream(t.device).cuda_stream), 'ncclAllReduce')
        return out

    def all_gather_into(self, out, t):
        assert t.is_cuda and t.is_contiguous() and out.is_contiguous() and out.numel() == self.world * t.numel() and out.dtype == t.dtype
        _check(_load().ncclAllGather(t.data_ptr(), out.data_ptr(), t.numel(), _dtype(t), self.comm,
                                     torch.cuda.current_stream(t.device).cuda_stream), 'ncclAllGather')
        return out

    def self_test(self, timeout_s=30.0):
        """One small all-reduce, waited for with a bounded host-side poll: a communicator whose kernels never finish is reported
        (and must then be aborted) instead of hanging the first BatchNorm layer."""
        t = torch.ones(1024, dtype=torch.float32, device='cuda')
        self.all_reduce_sum_(t)
        ev = torch.cuda.Event()
        ev.record()
        deadline = time.time() + timeout_s
        while not ev.query() and time.time() < deadline:
            time.sleep(0.005)
        return bool(ev.query()) and float(t[0].item()) == float(self.world)

    def count(self):
        """Rank count the communicator itself reports (ncclCommCount) -- the bench line's evidence that RCCL carried `world` ranks."""
        n = ctypes.c_int(0)
        _check(_load().ncclCommCount(self.comm, ctypes.byref(n)), 'ncclCommCount')
        return n.value

    def destroy(self, abort=False):
        if self.comm:
            (_load().ncclCommAbort if abort else _load().ncclCommDestroy)(self.comm)
            self.comm = ctypes.c_void_p()


_comms = {}      # group (None = WORLD) -> DirectComm, or False after an agreed fall-back
FALLBACK = {}    # group -> why the agreed fall-back to torch.distributed was taken (bench.py quotes it as `config.rccl_direct_reason`)


def get(group=None):
    """The direct communicator of `group` (created collectively on first use by every rank of the group), or None when disabled,
    when the backend is not RCCL, or when any rank failed to create it (then every rank falls back to torch.distributed)."""
    if not ENABLED or dist.get_backend(group) != 'nccl':
        return None
    key = group
    c = _comms.get(key)
    if c is None:
        err, alive = None, False
        try:
            c = DirectComm(group)
            alive = c.self_test()
            if not alive:
                err = 'self-test all-reduce did not complete'
        except Exception as e:      # noqa: BLE001 -- every failure mode ends in the same agreed fall-back
            err = e
        if c is not None and not alive:
            c.destroy(abort=True)       # before anything else is queued behind a kernel that may never finish
            c = None
        ok = torch.tensor([1.0 if alive else 0.0], device='cuda')
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if ok.item() < 1.0:
            if c is not None:
                c.destroy()
            warnings.warn('direct RCCL communicator unavailable (%r); using torch.distributed for the BN exchanges' % (err,))
            FALLBACK[key] = 'agreed fall-back to torch.distributed: %r on this rank (some rank failed to create or self-test the communicator)' % (err,)
            c = False
        _comms[key] = c
    return c or None


def shutdown():
    for c in _comms.values():
        if c:
            c.destroy()
    _comms.clear()
