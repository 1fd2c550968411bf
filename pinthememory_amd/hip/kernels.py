"""Raw (non-autograd) launches of the HIP kernels on torch-owned NHWC fp32 tensors.
Every function enqueues on torch's current stream and returns torch tensors; no CPU fallback exists."""
import ctypes
from ctypes import byref

import os

import torch

from . import lib as L
from .lib import PmConvEpilogue, PmConvParams, check, ptr, stream, tdesc, workspace


def _lib():
    return L.load()


# 0: fp32 MFMA (parity path, BASELINE configs[1]). 2: configs[2], the bf16 tier -- each convolution's operands are converted to bf16 in HBM
# (one streaming pass, RNE), bf16 tiles of 64 k-values per row in LDS, v_mfma_f32_32x32x16_bf16 with fp32 accumulation; activations between
# layers, BatchNorm, losses and the memory stay fp32. 1: the first form of that tier (fp32 tiles staged in LDS, rounded per fragment), which
# prec 2 falls back to for the call sites it does not cover (3-channel stem, 19-class heads, stride-2 data gradients).
CONV_PREC = 0
# Element type of the activations BETWEEN layers. 'bf16' (round 4) = the whole tier: convolutions write bf16, every elementwise / reduction kernel between them
# (BatchNorm, pooling, resize, adds: csrc/act16.hip) reads and writes bf16 with fp32 arithmetic; the image, the class logits, the losses, the memory module,
# every statistic and every parameter / parameter gradient stay fp32. 'bf16_operands' = round 2-3's form (fp32 activations, operands cast per convolution).
ACT_DTYPE = torch.float32


def set_conv_precision(name):
    global CONV_PREC, ACT_DTYPE
    CONV_PREC = {'f32': 0, 'fp32': 0, 'bf16': 2, 'bf16_operands': 2, 'bf16_staged': 1}[name]
    ACT_DTYPE = torch.bfloat16 if name == 'bf16' else torch.float32


def _act_dtype(like):
    return like.dtype if like.dtype in (torch.float32, torch.bfloat16) else torch.float32


def new(shape, like, pitch_pad=False, zero_pad=True, dtype=None):
    """Fresh NHWC tensor of `dtype` (default: like.dtype). fp32: channel counts that are not a multiple of 4 (the 19 logits) get a padded pitch so rows
    stay 16B aligned; the pad lanes are zero unless the producer writes them itself (zero_pad=False). bf16: a channel count that is not a multiple of 64
    (the 48-channel skip branch, the 304-channel decoder gradient) is allocated with zero pad channels up to the next multiple and registered, so that a
    convolution gathers it in place (lib.register_zero_pad)."""
    n, h, w, c = shape
    dtype = _act_dtype(like) if dtype is None else dtype
    if dtype == torch.bfloat16:
        if c % 64:
            base = torch.empty((n, h, w, (c + 63) // 64 * 64), dtype=dtype, device=like.device)
            base[..., c:].zero_()              # only the pad channels: every producer writes all c valid ones (a full clear of the 304 -> 320 concat buffer was 31 us)
            L.register_zero_pad(base, c, base.shape[3])
            return base[..., :c]
        return torch.empty((n, h, w, c), dtype=dtype, device=like.device)
    if c % 4 and pitch_pad:
        cp = (c + 3) // 4 * 4
        alloc = torch.zeros if zero_pad else torch.empty
        return alloc((n, h, w, cp), dtype=torch.float32, device=like.device)[..., :c]
    return torch.empty((n, h, w, c), dtype=torch.float32, device=like.device)


def cast(x, dtype):
    """NHWC tensor -> the same values as `dtype` (fp32 <-> bf16, round to nearest even): the edges of the bf16 tier."""
    if x.dtype == dtype:
        return x
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(_lib().pm_cast(byref(tdesc(x)), byref(tdesc(y)), stream()), 'pm_cast')
    return y


def conv_out_hw(h, w, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1


def krsc(w):
    """[Cout,Cin,kh,kw] parameter -> contiguous KRSC memory (a free view when the parameter is channels_last)."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


KEEP_WINOGRAD_V = os.environ.get('PM_KEEP_V', '1') == '1'   # forward keeps the Winograd-transformed input for the weight gradient


# BatchNorm statistics handed out by the producing convolution's epilogue: the unbatched direct GEMMs (every 1x1 and direct
# 3x3 convolution) emit (mean, M2) per 32-row slab and channel from their staged epilogue, merged in double by pm_bn_partials_finalize, so that no
# separate pass re-reads the convolution output: -1.0 ms/step of statistics passes for +0.35 ms of epilogue and the slab merge. Same-box A/B over
# five alternated pairs (tools/gpu_env_abn.sh): 66.41 vs 66.60 ms/step, every pair in favour (-0.08 ... -0.34). It stays opt-in (PM_BN_EPILOGUE=1):
# the statistics are as accurate as the separate pass (kernel test: 1e-6 of torch's double) but rounded differently, other ReLU units sit within
# round-off of zero, and with the default flipped one bias gradient of the mldg test (layer4.2.bn3.bias: 5.9e-3 of its norm, inside the fixed 1e-2
# bound) lands outside that test's RELATIVE bar (3 x the reference's own fp32 error + 1e-4). A 0.3 % gain is not worth a looser gate.
BN_EPILOGUE = os.environ.get('PM_BN_EPILOGUE', '0') == '1'
# bf16 tier (round 4): the same epilogue exists in both bf16 kernels (pm_slab_stats16) and measures SLOWER than the separate pass -- same box, two alternated
# pairs (tools/gpu_r4_envab.sh PM_BN_EPILOGUE16): 26.52 / 26.61 ms/step without, 26.94 / 26.94 with. The statistics pass reads a tensor the convolution has just
# left in the 256 MB infinity cache (10 us per layer), while the epilogue form adds two LDS sweeps and 48 lane exchanges per slab to an MFMA-bound kernel and
# hands the finalize 9 216 slab partials instead of <= 288 block partials. Opt-in (PM_BN_EPILOGUE16=1), covered by test_conv_epilogue_bn_statistics_bf16.
BN_EPILOGUE16 = os.environ.get('PM_BN_EPILOGUE16', '0') == '1'
# Round 5: per shape. Where the PRODUCING convolution is HBM-bound and its output does not stay in the 256 MB infinity cache -- the 1x1 convolutions onto the 192 x 192
# and 96 x 96 maps, 75-151 MB of bf16 output -- the separate statistics pass is a full HBM read (28-54 us per layer) while the epilogue's extra LDS sweeps hide under
# the convolution's own memory time. PM_BN_EP16_MIN_MB: outputs of at least this many MB take the epilogue route (0 = never).
BN_EP16_MIN_BYTES = int(float(os.environ.get('PM_BN_EP16_MIN_MB', '0')) * (1 << 20))
BN_EP_MIN_BYTES = int(float(os.environ.get('PM_BN_EP_MIN_MB', '0')) * (1 << 20))      # the same per-shape rule on the fp32 tier (outputs of at least this many MB)


# Transformed filters -- the Winograd U = G w G^T, or the bf16 copy of the weights in the bf16 tier -- kept between calls: the eval-mode forward of
# step t and the training forward of step t + 1 read the same weights. The cache is SCOPED TO AN OWNER that vouches for its updates: optim.SGD
# registers the parameters it owns (register_filter_owners) and bumps their tensor VERSION for its fused raw-pointer update; only registered leaf
# parameters are cached, keyed by storage address + transformed size, validated by (the owner is alive, still at that address, same version).
# Every in-place update through torch's ops bumps the version too. What does NOT bump it is a write through `w.data` or raw pointers by code that
# is not the owner (legacy optimizers, manual EMA, an in-place collective on parameters): weights nobody registered are never cached, so such code
# cannot meet a stale filter. Entries hold a WEAK reference to the owner -- functional weights (train_memory_mldg's theta), `.contiguous()` copies and
# models that are gone are neither cached nor retained. PM_KEEP_U=0 switches the cache off, PM_KEEP_U=1 caches every weight (A/B runs and tests).
KEEP_WINOGRAD_U = {'0': False, '1': True}.get(os.environ.get('PM_KEEP_U', ''))      # None = owner-scoped (default)

_U_OWNERS = {}      # storage address -> weakref to the registered leaf parameter
_U_CACHE = {}       # key -> [weakref(owner) or the weight itself when forced, version, U]
_U_CACHE_MAX = 256
_U_CACHE_BYTES = int(os.environ.get('PM_KEEP_U_MB', '2048')) << 20     # one ResNet-50 DeepLabV3+ keeps 0.43 GB, a ResNet-101 DeepLabV2 0.65 GB


def register_filter_owners(params):
    """Called by an optimizer for the weights it owns: it promises that every update of them moves the tensor version (optim.SGD does)."""
    import weakref
    for p in params:
        if isinstance(p, torch.nn.Parameter) and p.is_leaf and p.is_cuda and p.dim() == 4 and p.dtype == torch.float32:
            _U_OWNERS[p.data_ptr()] = weakref.ref(p)


def unregister_filter_owners(params=None):
    """Forget the owner's weights (all of them when params is None) and drop their cached filters."""
    ptrs = None if params is None else {p.data_ptr() for p in params}
    for k in [k for k in _U_OWNERS if ptrs is None or k in ptrs]:
        del _U_OWNERS[k]
    for k in [k for k in _U_CACHE if ptrs is None or k[0] in ptrs]:
        del _U_CACHE[k]


def _filter_owner(w_krsc):
    ref = _U_OWNERS.get(w_krsc.data_ptr())
    p = ref() if ref is not None else None
    if p is None:
        if ref is not None:
            del _U_OWNERS[w_krsc.data_ptr()]
        return None
    return p if (p.data_ptr() == w_krsc.data_ptr() and p.numel() == w_krsc.numel() and p.is_leaf) else None


_U_STREAMS = {}     # every stream that ever read or wrote a kept transform (main stream, the harness' commit stream)


def _evict(ent):
    """The buffer goes back to the pool of the stream that allocated it, while a launch on ANOTHER stream (the commit forward) may still be reading it:
    tell the allocator about every stream that uses the cache, so that the block is not handed out again before their queued work is done."""
    if ent[2].is_cuda:
        for s in _U_STREAMS.values():
            ent[2].record_stream(s)


def _wino_u(p, w_krsc, nbu, dgrad=False):
    """nbu: bytes of the transformed filter of this call (pm_conv_wxf_bytes / _dgrad; 0: none). dgrad: the kept transform is the rotated / transposed bf16 filter
    of a data gradient on the bf16 tier."""
    if KEEP_WINOGRAD_U is False or not nbu:
        return None
    owner = _filter_owner(w_krsc)
    if owner is None and KEEP_WINOGRAD_U is not True:
        return None
    import weakref
    key = (w_krsc.data_ptr(), tuple(w_krsc.shape), nbu, w_krsc.device.index, p.prec, dgrad)
    ent = _U_CACHE.pop(key, None)
    if ent is not None and (ent[0]() if isinstance(ent[0], weakref.ref) else ent[0]) is not (owner if owner is not None else w_krsc):
        ent = None                 # another tensor now lives at this address: the kept transform is somebody else's
    if ent is None:
        # oldest entries go first (dicts keep insertion order; hits are re-inserted): models that are gone release their buffers here
        for k in [k for k, e in _U_CACHE.items() if isinstance(e[0], weakref.ref) and e[0]() is None]:
            _evict(_U_CACHE.pop(k))
        while _U_CACHE and (len(_U_CACHE) >= _U_CACHE_MAX or sum(e[2].numel() * 4 for e in _U_CACHE.values()) + nbu > _U_CACHE_BYTES):
            _evict(_U_CACHE.pop(next(iter(_U_CACHE))))
        ent = [weakref.ref(owner) if owner is not None else w_krsc, -1, torch.empty(nbu // 4, dtype=torch.float32, device=w_krsc.device), None, None, True]
    ent[5] = True                       # used since the last refresh_bf16_filters()
    version = (owner if owner is not None else w_krsc)._version
    valid = ent[1] == version
    if not valid:
        ent[1] = -1                     # invalid until the launch that fills it has been enqueued (conv_fwd commits the version after check())
    _U_CACHE[key] = ent
    p.wxf, p.wxf_bytes, p.wxf_valid = ent[2].data_ptr(), nbu, 1 if valid else 0
    raw = stream()
    if raw not in _U_STREAMS:
        _U_STREAMS[raw] = L.stream_obj()
    if valid and ent[3] is not None and ent[4] != raw:
        _U_STREAMS[raw].wait_event(ent[3])          # the transform was written by a launch on another stream (harness: the commit forward runs on its own)
    return None if valid else (ent, version)


def forget_filter_events():
    """Drop the 'written by a launch on another stream' events of every kept transform. Call after torch.cuda.synchronize(): harness.GraphedAggStep does once its capture has
    ended -- events recorded inside a stream capture must not be waited for from eager code, and a replay (a single launch in stream order) needs none."""
    for ent in _U_CACHE.values():
        ent[3] = ent[4] = None


WXF_REFRESH = os.environ.get('PM_WXF_REFRESH', '1') != '0'      # A/B knob


def refresh_bf16_filters():
    """bf16 tier, called by the optimizer right after it moved the weights: rewrite every kept bf16 filter (forward copy, rotated copy of the data gradient) that
    was used since the last call and is now out of date, in ONE or two launches (pm_conv_wxf_refresh_bf16) instead of one small cast in front of each of the
    ~142 convolution calls of the next step. Entries nobody used in the last step are left to go stale (their convolution re-derives them if it returns).
    The caller has already made sure no other stream still reads the buffers (optim.SGD.step: ops.wait_commit()). Returns the number of filters rewritten."""
    if CONV_PREC != 2 or KEEP_WINOGRAD_U is False or not _U_CACHE or not WXF_REFRESH:
        return 0
    import weakref
    dev = torch.cuda.current_device()
    todo = []
    for key, ent in _U_CACHE.items():
        if key[4] != 2 or key[3] != dev or not ent[5]:
            continue
        ent[5] = False
        owner = ent[0]() if isinstance(ent[0], weakref.ref) else ent[0]
        if owner is None or owner.data_ptr() != key[0] or ent[1] == owner._version or not ent[2].is_cuda:
            continue
        todo.append((key, ent, owner))
    if not todo:
        return 0
    jobs = (L.PmWxfJob * len(todo))()
    for j, (key, ent, owner) in zip(jobs, todo):
        cout, kh, kw, cin = key[1]
        j.w, j.wxf, j.wxf_bytes, j.cout, j.kh, j.kw, j.cin, j.dgrad = key[0], ent[2].data_ptr(), key[2], cout, kh, kw, cin, 1 if key[5] else 0
        ent[1] = -1
    check(_lib().pm_conv_wxf_refresh_bf16(jobs, len(todo), stream()), 'pm_conv_wxf_refresh_bf16')
    cur = L.stream_obj()
    raw = cur.cuda_stream
    if raw not in _U_STREAMS:
        _U_STREAMS[raw] = cur
    ev = cur.record_event()
    for key, ent, owner in todo:
        ent[1], ent[3], ent[4] = owner._version, ev, raw
    return len(todo)


# Size queries of the convolution entry points (workspace, kept Winograd V, transformed filter, statistics partials) depend on shapes and on the library's
# process-wide switches only: asked once per distinct call signature instead of three ctypes round trips per launch (the bf16 tier is host-bound).
_SIZES = {}
_GEN = [0]      # bumped by every switch that changes the library's routing (set_winograd, set_conv16, ...)


def _sizes(key, fn):
    v = _SIZES.get(key)
    if v is None:
        if len(_SIZES) > 8192:
            _SIZES.clear()
        v = _SIZES[key] = fn()
    return v


_XFORM_COUNT = [0, False]      # [convolution calls that had to derive a transformed filter (Winograd U / bf16 copy) instead of finding a kept one, counting on]


def filter_transform_count(enable=None):
    """Diagnostic (bench.py --workload mldg): number of forward convolutions that transformed their filter while counting was on."""
    if enable is not None:
        _XFORM_COUNT[1] = bool(enable)
    return _XFORM_COUNT[0]


def conv_fwd(x, w_krsc, stride, pad, dil, bias=None, scale=None, shift=None, residual=None, relu=False, out=None, keep_v=None, bn_partials=None, out_dtype=None):
    """keep_v: a list; if this convolution and its weight gradient both take the Winograd route, the transformed input V is written
    to a fresh tensor that is appended to the list (else None is appended) -- pass it to conv_bwd_weight(wino_v=...).
    bn_partials: a list; if this call can hand the train-mode BatchNorm statistics of its output out of its own epilogue ((mean, M2) per
    32-row slab and channel), the partials tensor is appended (else None) -- merge it with bn_partials_finalize(...) instead of bn_stats(y)."""
    cout, kh, kw, cin = w_krsc.shape
    n, h, w_, c = x.shape
    assert c == cin, 'conv: Cin mismatch %d vs %d' % (c, cin)
    ho, wo = conv_out_hw(h, w_, kh, stride, pad, dil)
    y = out if out is not None else new((n, ho, wo, cout), x, pitch_pad=True, dtype=out_dtype if out_dtype is not None else ACT_DTYPE)
    assert residual is None or residual.dtype == y.dtype, 'conv_fwd: the fused residual has the dtype of the output'
    xd, yd = tdesc(x), tdesc(y)
    p = L.conv_params(kh, kw, stride, pad, dil, CONV_PREC)
    lib = _lib()
    nbv, nbu, nb, npb = _sizes(('f', x.shape, x.stride(), x.dtype, xd.flags, w_krsc.shape, y.stride(), y.dtype, stride, pad, dil, CONV_PREC, _GEN[0]),
                               lambda: (lib.pm_conv_winograd_v_bytes(byref(xd), byref(yd), byref(p)), lib.pm_conv_wxf_bytes(byref(xd), byref(yd), byref(p)),
                                        lib.pm_conv_workspace(byref(xd), byref(yd), byref(p), 0), lib.pm_conv_bn_partials_bytes(byref(xd), byref(yd), byref(p))))
    if keep_v is not None:
        nbv = nbv if KEEP_WINOGRAD_V else 0
        v = torch.empty(nbv // 4, dtype=torch.float32, device=x.device) if nbv else None
        keep_v.append(v)
        if v is not None:
            p.wino_v, p.wino_v_bytes = v.data_ptr(), nbv
    u_ent = None
    if (kh == 3 and CONV_PREC == 0) or CONV_PREC == 2:
        u_ent = _wino_u(p, w_krsc, nbu)        # (cache entry this call is about to (re)write, version to commit once the launch is enqueued) or None
    if _XFORM_COUNT[1] and p.wxf_valid == 0 and nbu:
        _XFORM_COUNT[0] += 1      # this call transforms its filter
    ws = workspace(nb, x.device) if nb else None
    part = None
    if bn_partials is not None:
        ep16 = BN_EPILOGUE16 or (BN_EP16_MIN_BYTES > 0 and y.numel() * 2 >= BN_EP16_MIN_BYTES)
        ep32 = BN_EPILOGUE or (BN_EP_MIN_BYTES > 0 and y.numel() * 4 >= BN_EP_MIN_BYTES)
        npb = npb if ((ep16 if y.dtype == torch.bfloat16 else ep32) and residual is None and not relu and scale is None) else 0
        part = torch.empty(npb // 4, dtype=torch.float32, device=x.device) if npb else None
        bn_partials.append(part)
    ep = None
    if bias is not None or scale is not None or residual is not None or relu or part is not None:
        rd = tdesc(residual) if residual is not None else None
        ep = L.conv_epilogue(ptr(bias), ptr(scale), ptr(shift), rd.ptr if rd else None, rd.pitch if rd else 0, 1 if relu else 0, ptr(part),
                            part.numel() * 4 if part is not None else 0)
    check(lib.pm_conv_fwd(byref(xd), w_krsc.data_ptr(), byref(yd), byref(p), byref(ep) if ep else None, ptr(ws), nb, stream()), 'pm_conv_fwd')
    if u_ent is not None:               # the launch is enqueued: only now is the kept transform valid; hits from another stream wait for this event
        ent, version = u_ent
        cur = L.stream_obj()
        ent[3], ent[4] = cur.record_event(), cur.cuda_stream
        ent[1] = version
    return y


def conv_bwd_data(dy, w_krsc, x_shape, stride, pad, dil, add=None, dtype=None):
    """dtype: element type of dx = the type of the tensor it is the gradient of (default: dy's)."""
    cout, kh, kw, cin = w_krsc.shape
    dx = new(x_shape, dy, dtype=dtype)
    dyd, dxd = tdesc(dy), tdesc(dx)
    p = L.conv_params(kh, kw, stride, pad, dil, CONV_PREC)
    lib = _lib()
    nb, nbu = _sizes(('d', dy.shape, dy.stride(), dy.dtype, dyd.flags, w_krsc.shape, dx.shape, dx.stride(), dx.dtype, stride, pad, dil, CONV_PREC, _GEN[0]),
                     lambda: (lib.pm_conv_workspace(byref(dxd), byref(dyd), byref(p), 1), lib.pm_conv_wxf_bytes_dgrad(byref(dyd), byref(dxd), byref(p))))
    ws = workspace(nb, dy.device) if nb else None
    ad = tdesc(add) if add is not None else None
    u_ent = _wino_u(p, w_krsc, nbu, dgrad=True) if CONV_PREC == 2 else None      # bf16 tier: the rotated bf16 filter is kept per weight version
    check(lib.pm_conv_bwd_data(byref(dyd), w_krsc.data_ptr(), byref(dxd), byref(p), byref(ad) if ad else None, ptr(ws), nb, stream()), 'pm_conv_bwd_data')
    if u_ent is not None:
        ent, version = u_ent
        cur = L.stream_obj()
        ent[3], ent[4] = cur.record_event(), cur.cuda_stream
        ent[1] = version
    return dx


def conv_bwd_weight(x, dy, w_shape_krsc, stride, pad, dil, want_bias=False, wino_v=None, on_stream=None):
    """on_stream: launch on that torch stream instead of the current one (the caller orders it and tells the allocator: ops._wgrad)."""
    cout, kh, kw, cin = w_shape_krsc
    dw = torch.empty(w_shape_krsc, dtype=torch.float32, device=x.device)
    db = torch.empty(cout, dtype=torch.float32, device=x.device) if want_bias else None
    xd, dyd = tdesc(x), tdesc(dy)
    p = L.conv_params(kh, kw, stride, pad, dil, CONV_PREC)
    if wino_v is not None:       # Winograd-transformed x kept by conv_fwd(keep_v=...)
        p.wino_v, p.wino_v_bytes = wino_v.data_ptr(), wino_v.numel() * 4
    lib = _lib()
    nb = _sizes(('w', x.shape, x.stride(), x.dtype, dy.shape, dy.stride(), dy.dtype, w_shape_krsc, stride, pad, dil, CONV_PREC, wino_v is not None, _GEN[0]),
                lambda: lib.pm_conv_workspace(byref(xd), byref(dyd), byref(p), 2))
    raw = stream() if on_stream is None else on_stream.cuda_stream
    ws = workspace(nb, x.device, on_stream)
    check(lib.pm_conv_bwd_weight(byref(xd), byref(dyd), dw.data_ptr(), ptr(db), byref(p), ptr(ws), nb, raw), 'pm_conv_bwd_weight')
    return dw, db


# ---- batch norm --------------------------------------------------------------------------------------------------
def bn_stats(x):
    """-> moments float[3C] = mean | M2 | count (local to this rank)."""
    c = x.shape[3]
    mom = torch.empty(3 * c, dtype=torch.float32, device=x.device)
    xd = tdesc(x)
    lib = _lib()
    nb = _sizes(('bn', x.shape, x.dtype), lambda: lib.pm_bn_workspace(byref(xd)))
    ws = workspace(nb, x.device)
    check(lib.pm_bn_stats(byref(xd), mom.data_ptr(), ptr(ws), nb, stream()), 'pm_bn_stats')
    return mom


def bn_stats_finalize(x, eps, running_mean=None, running_var=None, momentum=0.1):
    """bn_stats + bn_finalize for local statistics in one library call -> (mean, invstd); running moments updated in place."""
    c = x.shape[3]
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    xd = tdesc(x)
    lib = _lib()
    nb = _sizes(('bn', x.shape, x.dtype), lambda: lib.pm_bn_workspace(byref(xd)))
    ws = workspace(nb, x.device)
    check(lib.pm_bn_stats_finalize(byref(xd), eps, mean.data_ptr(), invstd.data_ptr(), ptr(running_mean), ptr(running_var), momentum, ptr(ws), nb, stream()),
          'pm_bn_stats_finalize')
    return mean, invstd


def bn_partials_finalize(part, pixels, c, eps, running_mean=None, running_var=None, momentum=0.1):
    """(mean, M2) slab partials emitted by a convolution epilogue -> (mean, invstd); running moments updated in place."""
    mean = torch.empty(c, dtype=torch.float32, device=part.device)
    invstd = torch.empty_like(mean)
    check(_lib().pm_bn_partials_finalize(part.data_ptr(), pixels, c, eps, mean.data_ptr(), invstd.data_ptr(), ptr(running_mean), ptr(running_var), momentum, None,
                                         stream()), 'pm_bn_partials_finalize')
    return mean, invstd


def bn_partials_moments(part, pixels, c):
    """the same partials -> moments float[3C] = mean | M2 | count (the SyncBatchNorm exchange format of bn_stats)."""
    mom = torch.empty(3 * c, dtype=torch.float32, device=part.device)
    check(_lib().pm_bn_partials_finalize(part.data_ptr(), pixels, c, 0.0, None, None, None, None, 0.0, mom.data_ptr(), stream()), 'pm_bn_partials_finalize')
    return mom


def bn_merge(parts, world, c):
    """parts: float[world, 3c] gathered per-rank moments -> merged float[3c]."""
    out = torch.empty(3 * c, dtype=torch.float32, device=parts.device)
    check(_lib().pm_bn_merge(parts.data_ptr(), world, c, out.data_ptr(), stream()), 'pm_bn_merge')
    return out


def bn_merge_finalize(parts, world, c, eps, running_mean=None, running_var=None, momentum=0.1):
    """bn_merge + bn_finalize in one launch: gathered per-rank moments float[world, 3c] -> (mean, invstd); running moments in place."""
    mean = torch.empty(c, dtype=torch.float32, device=parts.device)
    invstd = torch.empty_like(mean)
    check(_lib().pm_bn_merge_finalize(parts.data_ptr(), world, c, eps, mean.data_ptr(), invstd.data_ptr(), ptr(running_mean), ptr(running_var), momentum,
                                      stream()), 'pm_bn_merge_finalize')
    return mean, invstd


def bn_finalize(moments, c, eps, running_mean=None, running_var=None, momentum=0.1):
    mean = torch.empty(c, dtype=torch.float32, device=moments.device)
    invstd = torch.empty_like(mean)
    check(_lib().pm_bn_finalize(moments.data_ptr(), c, eps, mean.data_ptr(), invstd.data_ptr(), ptr(running_mean), ptr(running_var), momentum, stream()),
          'pm_bn_finalize')
    return mean, invstd


def bn_fold_multi(table, cs, offs, n, max_c, total, eps):
    """n BatchNorm layers folded in one launch -> arena float[2 * total] (scales | shifts); table / cs / offs are device tensors."""
    arena = torch.empty(2 * total, dtype=torch.float32, device=table.device)
    check(_lib().pm_bn_fold_multi(table.data_ptr(), cs.data_ptr(), offs.data_ptr(), n, max_c, total, eps, arena.data_ptr(), stream()), 'pm_bn_fold_multi')
    return arena


def bn_fold(gamma, beta, running_mean, running_var, eps, conv_bias=None):
    c = gamma.numel()
    scale = torch.empty(c, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    check(_lib().pm_bn_fold(gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), ptr(conv_bias), c, eps, scale.data_ptr(),
                            shift.data_ptr(), stream()), 'pm_bn_fold')
    return scale, shift


def bn_apply(x, mean, invstd, gamma, beta, residual=None, relu=False, out=None, want_mask=False):
    """want_mask: also -> uint8 [pixels, C / 4], bit e of a byte = output e of that float4 group is positive (the ReLU mask for bn_bwd_reduce_mask)."""
    y = out if out is not None else torch.empty_like(x, memory_format=torch.contiguous_format)
    rd = tdesc(residual) if residual is not None else None
    grp = 8 if x.dtype == torch.bfloat16 else 4      # channels per mask byte = channels per 16-byte lane access
    mask = torch.empty((x.shape[0] * x.shape[1] * x.shape[2], x.shape[3] // grp), dtype=torch.uint8, device=x.device) if want_mask else None
    check(_lib().pm_bn_apply_mask(byref(tdesc(x)), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), byref(rd) if rd else None,
                                  1 if relu else 0, byref(tdesc(y)), ptr(mask), stream()), 'pm_bn_apply_mask')
    return (y, mask) if want_mask else y


def bn_bwd_reduce_mask(dy, mask, x, mean, invstd, want_gmask=True, with_count=False):
    """bn_bwd_reduce(relu=1) with the ReLU mask read from bn_apply(want_mask=True)'s bytes instead of the forward output."""
    c = x.shape[3]
    sums = torch.empty(2 * c + (1 if with_count else 0), dtype=torch.float32, device=x.device)
    if with_count:
        sums[2 * c:].fill_(float(x.shape[0] * x.shape[1] * x.shape[2]))
    xd = tdesc(x)
    lib = _lib()
    nb = _sizes(('bn', x.shape, x.dtype), lambda: lib.pm_bn_workspace(byref(xd)))
    ws = workspace(nb, x.device)
    gm = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_gmask else None
    gd = tdesc(gm) if want_gmask else None
    check(lib.pm_bn_bwd_reduce_mask(byref(tdesc(dy)), mask.data_ptr(), byref(xd), mean.data_ptr(), invstd.data_ptr(), byref(gd) if gd else None,
                                    sums.data_ptr(), ptr(ws), nb, stream()), 'pm_bn_bwd_reduce_mask')
    return sums, gm


def scale_shift_act(x, scale, shift, residual=None, relu=False):
    """y = relu?(x * scale[c] + shift[c] + residual?) -- per-channel affine on an NHWC tensor."""
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    rd = tdesc(residual) if residual is not None else None
    check(_lib().pm_scale_shift_act(byref(tdesc(x)), scale.data_ptr(), shift.data_ptr(), byref(rd) if rd else None, 1 if relu else 0, byref(tdesc(y)), stream()),
          'pm_scale_shift_act')
    return y


def bn_bwd_reduce(dy, y, x, mean, invstd, relu, gamma=None, beta=None, want_gmask=False, with_count=False):
    """relu: 0 / False none, 1 / True mask from the forward output y, 2 mask rebuilt from x (needs gamma, beta). -> (sums, dy * mask or None).
    with_count: sums gets a third section [2c] = this rank's element count (SyncBatchNorm all-reduces it together with the sums)."""
    relu = int(relu)
    c = x.shape[3]
    sums = torch.empty(2 * c + (1 if with_count else 0), dtype=torch.float32, device=x.device)
    if with_count:
        sums[2 * c:].fill_(float(x.shape[0] * x.shape[1] * x.shape[2]))
    xd = tdesc(x)
    lib = _lib()
    nb = _sizes(('bn', x.shape, x.dtype), lambda: lib.pm_bn_workspace(byref(xd)))
    ws = workspace(nb, x.device)
    yd = tdesc(y) if relu == 1 else None
    gm = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_gmask else None
    gd = tdesc(gm) if want_gmask else None
    check(lib.pm_bn_bwd_reduce(byref(tdesc(dy)), byref(yd) if yd else None, byref(xd), mean.data_ptr(), invstd.data_ptr(), ptr(gamma), ptr(beta), relu,
                               byref(gd) if gd else None, sums.data_ptr(), ptr(ws), nb, stream()), 'pm_bn_bwd_reduce')
    return sums, gm


def bn_bwd_apply(dy, y, x, mean, invstd, gamma, sums, count, relu, want_dres, beta=None):
    relu = int(relu)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    dres = torch.empty(x.shape, dtype=x.dtype, device=x.device) if want_dres else None
    yd = tdesc(y) if relu == 1 else None
    dr = tdesc(dres) if want_dres else None
    check(_lib().pm_bn_bwd_apply(byref(tdesc(dy)), byref(yd) if yd else None, byref(tdesc(x)), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                 ptr(beta), sums.data_ptr(), float(count), relu, byref(tdesc(dx)), byref(dr) if dr else None, stream()), 'pm_bn_bwd_apply')
    return dx, dres


def relu_bwd(dy, y):
    dx = torch.empty(y.shape, dtype=torch.float32, device=y.device)
    check(_lib().pm_relu_bwd(byref(tdesc(dy)), byref(tdesc(y)), byref(tdesc(dx)), stream()), 'pm_relu_bwd')
    return dx


def add_n(xs):
    """Sum of 2..8 same-shape NHWC tensors in one pass."""
    from .lib import PmTensor
    from ctypes import POINTER, pointer
    descs = [tdesc(x) for x in xs]
    arr = (POINTER(PmTensor) * len(xs))(*[pointer(d) for d in descs])
    y = torch.empty(xs[0].shape, dtype=xs[0].dtype, device=xs[0].device)
    check(_lib().pm_add_n(arr, len(xs), byref(tdesc(y)), stream()), 'pm_add_n')
    return y


def add(a, b, out=None):
    y = out if out is not None else torch.empty(a.shape, dtype=a.dtype, device=a.device)
    check(_lib().pm_add(byref(tdesc(a)), byref(tdesc(b)), byref(tdesc(y)), stream()), 'pm_add')
    return y


def copy(src, dst):
    check(_lib().pm_copy(byref(tdesc(src)), byref(tdesc(dst)), stream()), 'pm_copy')
    return dst


# ---- pooling / resize / layout -------------------------------------------------------------------------------------
def maxpool_fwd(x):
    n, h, w, c = x.shape
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=x.dtype, device=x.device)
    arg = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
    check(_lib().pm_maxpool3x3s2_fwd(byref(tdesc(x)), byref(tdesc(y)), arg.data_ptr(), stream()), 'pm_maxpool3x3s2_fwd')
    return y, arg


def maxpool_bwd(dy, arg, x_shape):
    dx = torch.empty(x_shape, dtype=dy.dtype, device=dy.device)
    check(_lib().pm_maxpool3x3s2_bwd(byref(tdesc(dy)), arg.data_ptr(), byref(tdesc(dx)), stream()), 'pm_maxpool3x3s2_bwd')
    return dx


def global_avgpool_fwd(x):
    n, h, w, c = x.shape
    y = torch.empty((n, 1, 1, c), dtype=x.dtype, device=x.device)
    check(_lib().pm_global_avgpool_fwd(byref(tdesc(x)), byref(tdesc(y)), stream()), 'pm_global_avgpool_fwd')
    return y


def global_avgpool_bwd(dy, x_shape):
    dx = torch.empty(x_shape, dtype=dy.dtype, device=dy.device)
    check(_lib().pm_global_avgpool_bwd(byref(tdesc(dy)), byref(tdesc(dx)), 0, stream()), 'pm_global_avgpool_bwd')
    return dx


def resize_fwd(x, size, out=None):
    n, h, w, c = x.shape
    # pitch-padded input (zero pad lanes) -> the float4 path writes the output's pad lanes too: no zero fill of a 358 MB tensor
    own_pad = c % 4 != 0 and x.stride(2) == (c + 3) // 4 * 4
    y = out if out is not None else new((n, size[0], size[1], c), x, pitch_pad=True, zero_pad=not own_pad)
    check(_lib().pm_resize_bilinear_fwd(byref(tdesc(x)), byref(tdesc(y)), stream()), 'pm_resize_bilinear_fwd')
    return y


RESIZE_SEPARABLE = os.environ.get('PM_RESIZE_SEP', '1') == '1'      # A/B knob: 0 = gather formulation everywhere


def resize_bwd(dy, x_shape, separable=None):
    separable = RESIZE_SEPARABLE if separable is None else separable
    dx = new(x_shape, dy, pitch_pad=True)
    lib, dyd, dxd = _lib(), tdesc(dy), tdesc(dx)
    nb = lib.pm_resize_bilinear_bwd_workspace(byref(dyd), byref(dxd)) if separable else 0
    if nb:      # up-sampling by >= 2: column pass + row pass, the large gradient is read once
        ws = workspace(nb, dy.device)
        check(lib.pm_resize_bilinear_bwd_separable(byref(dyd), byref(dxd), 0, ptr(ws), nb, stream()), 'pm_resize_bilinear_bwd_separable')
    else:
        check(lib.pm_resize_bilinear_bwd(byref(dyd), byref(dxd), 0, stream()), 'pm_resize_bilinear_bwd')
    return dx


def resize_hp_fwd(x, size, flip_w=False):
    """F.interpolate(mode='bilinear', align_corners=False) of an NHWC tensor, optionally un-flipping the width on the way out."""
    n, h, w, c = x.shape
    y = torch.empty((n, size[0], size[1], c), dtype=torch.float32, device=x.device)
    check(_lib().pm_resize_bilinear_hp_fwd(byref(tdesc(x)), byref(tdesc(y)), 1 if flip_w else 0, stream()), 'pm_resize_bilinear_hp_fwd')
    return y


def softmax_mean_update(logits, buffer, counter):
    check(_lib().pm_softmax_mean_update(byref(tdesc(logits)), buffer.data_ptr(), counter, stream()), 'pm_softmax_mean_update')


def sliding_stitch(logits, tiles, H, W, flip_w, acc=None):
    """logits NHWC [T, th, tw, C] of the tiles [(x1, y1, x2, y2)] -> float64 [C, H, W]: sum over covering tiles / count, un-flipped; added to `acc` if given."""
    import ctypes
    T, th, tw, c = logits.shape
    arr = (ctypes.c_int32 * (4 * T))(*[int(v) for t in tiles for v in t])
    out = acc if acc is not None else torch.empty((c, H, W), dtype=torch.float64, device=logits.device)
    check(L.load().pm_sliding_stitch(byref(tdesc(logits)), arr, T, H, W, 1 if flip_w else 0, out.data_ptr(), 1 if acc is not None else 0, stream()), 'pm_sliding_stitch')
    return out


def argmax_f64(buffer):
    n, h, w, c = buffer.shape
    cls = torch.empty((n, h, w), dtype=torch.int64, device=buffer.device)
    prob = torch.empty((n, h, w), dtype=torch.float64, device=buffer.device)
    check(_lib().pm_argmax_f64(buffer.data_ptr(), n, h, w, c, cls.data_ptr(), prob.data_ptr(), stream()), 'pm_argmax_f64')
    return prob, cls


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # datasets/gtav.py:284, eval.py:128


def image_u8_to_nhwc4(img_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 [N,H,W,3] -> normalised fp32 NHWC4 [N,H,W,4] (ToTensor + Normalize, zero 4th channel)."""
    import ctypes
    n, h, w, c = img_u8.shape
    assert c == 3 and img_u8.dtype == torch.uint8 and img_u8.is_contiguous()
    out = torch.empty((n, h, w, 4), dtype=torch.float32, device=img_u8.device)
    m, s = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(_lib().pm_image_u8_to_nhwc4(img_u8.data_ptr(), n * h * w, m, s, out.data_ptr(), stream()), 'pm_image_u8_to_nhwc4')
    return out


def labels_u8_to_i64(lab_u8):
    assert lab_u8.dtype == torch.uint8 and lab_u8.is_contiguous()
    out = torch.empty(lab_u8.shape, dtype=torch.int64, device=lab_u8.device)
    check(_lib().pm_labels_u8_to_i64(lab_u8.data_ptr(), lab_u8.numel(), out.data_ptr(), stream()), 'pm_labels_u8_to_i64')
    return out


def nchw_to_nhwc(x, c_pad=None):
    n, c, h, w = x.shape
    x = x.contiguous()
    y = torch.empty((n, h, w, c_pad or c), dtype=torch.float32, device=x.device)
    check(_lib().pm_nchw_to_nhwc(x.data_ptr(), c, byref(tdesc(y)), stream()), 'pm_nchw_to_nhwc')
    return y


def nhwc_to_nchw(x):
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    check(_lib().pm_nhwc_to_nchw(byref(tdesc(x)), y.data_ptr(), stream()), 'pm_nhwc_to_nchw')
    return y


def label_nearest(lab, size):
    n, H, W = lab.shape
    lab = lab.contiguous()
    out = torch.empty((n, size[0], size[1]), dtype=torch.int64, device=lab.device)
    check(_lib().pm_label_nearest(lab.data_ptr(), n, H, W, out.data_ptr(), size[0], size[1], stream()), 'pm_label_nearest')
    return out


# ---- fused upsample + cross entropy -----------------------------------------------------------------------------
def upsample_ce_fwd(logits, labels, inv_temp=1.0):
    n, H, W = labels.shape
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    lib = _lib()
    nb = lib.pm_upsample_ce_workspace(n, H, W)
    ws = workspace(nb, logits.device)
    check(lib.pm_upsample_ce_fwd(byref(tdesc(logits)), inv_temp, labels.data_ptr(), H, W, out.data_ptr(), ptr(ws), nb, stream()), 'pm_upsample_ce_fwd')
    return out


def upsample_ce_bwd(logits, labels, loss_out, gscale, inv_temp=1.0):
    n, H, W = labels.shape
    dl = new(tuple(logits.shape), logits, pitch_pad=(logits.stride(2) != logits.shape[3]))   # same pitch as the logits
    lib, ld = _lib(), tdesc(logits)
    nb = lib.pm_upsample_ce_bwd_workspace(byref(ld), H, W)
    ws = workspace(nb, logits.device)
    check(lib.pm_upsample_ce_bwd(byref(ld), inv_temp, labels.data_ptr(), H, W, loss_out.data_ptr(), ptr(gscale), byref(tdesc(dl)), ptr(ws), nb, stream()),
          'pm_upsample_ce_bwd')
    return dl


def upsample_ce_fused_ok(logits, label_hw):
    """Can the fused up-sample + CE kernels take this shape (two low-res logit rows + one label row in LDS: up to ~1000 low-res columns at 19 classes)?"""
    return _sizes(('ce', logits.shape, logits.stride(), tuple(label_hw)), lambda: _lib().pm_upsample_ce_field_bytes(byref(tdesc(logits)), label_hw[0], label_hw[1])) != 0


def upsample_ce_fwd_field(logits, labels, inv_temp=1.0):
    """Training forward: -> (loss_out[2], field). The field is what upsample_ce_bwd_field needs instead of a second sweep over labels and logits."""
    n, H, W = labels.shape
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    lib, ld = _lib(), tdesc(logits)
    field = torch.empty(lib.pm_upsample_ce_field_bytes(byref(ld), H, W) // 4, dtype=torch.float32, device=logits.device)
    nb = lib.pm_upsample_ce_workspace(n, H, W)
    ws = workspace(nb, logits.device)
    check(lib.pm_upsample_ce_fwd_field(byref(ld), inv_temp, labels.data_ptr(), H, W, out.data_ptr(), field.data_ptr(), ptr(ws), nb, stream()),
          'pm_upsample_ce_fwd_field')
    return out, field


def upsample_ce_bwd_field(logits, label_hw, loss_out, field, gscale, inv_temp=1.0):
    H, W = label_hw
    dl = new(tuple(logits.shape), logits, pitch_pad=(logits.stride(2) != logits.shape[3]))   # same pitch as the logits
    check(_lib().pm_upsample_ce_bwd_field(byref(tdesc(logits)), inv_temp, H, W, loss_out.data_ptr(), ptr(gscale), field.data_ptr(), byref(tdesc(dl)), stream()),
          'pm_upsample_ce_bwd_field')
    return dl


def refresh_f32_filters():
    """fp32 tier twin of refresh_bf16_filters(): every kept Winograd FORWARD transform (U = G g Gt of the wide stride-1 3x3 layers) that was used since the last call and
    is now out of date, rewritten in one launch (pm_conv_wxf_refresh_f32) behind the optimizer step instead of ~20 latency-bound per-layer launches in front of the
    next forward pass. Same kernel body, same bits. Returns the number of filters rewritten."""
    if CONV_PREC != 0 or KEEP_WINOGRAD_U is False or not _U_CACHE or not WXF_REFRESH:
        return 0
    import weakref
    dev = torch.cuda.current_device()
    todo = []
    for key, ent in _U_CACHE.items():
        if key[4] != 0 or key[5] or key[3] != dev or not ent[5] or key[1][1] != 3 or key[1][2] != 3:
            continue
        ent[5] = False
        owner = ent[0]() if isinstance(ent[0], weakref.ref) else ent[0]
        if owner is None or owner.data_ptr() != key[0] or ent[1] == owner._version or not ent[2].is_cuda:
            continue
        todo.append((key, ent, owner))
    if not todo:
        return 0
    jobs = (L.PmWxfJob * len(todo))()
    for j, (key, ent, owner) in zip(jobs, todo):
        cout, kh, kw, cin = key[1]
        j.w, j.wxf, j.wxf_bytes, j.cout, j.kh, j.kw, j.cin, j.dgrad = key[0], ent[2].data_ptr(), key[2], cout, kh, kw, cin, 0
        ent[1] = -1
    check(_lib().pm_conv_wxf_refresh_f32(jobs, len(todo), stream()), 'pm_conv_wxf_refresh_f32')
    cur = L.stream_obj()
    raw = cur.cuda_stream
    if raw not in _U_STREAMS:
        _U_STREAMS[raw] = cur
    ev = cur.record_event()
    for key, ent, owner in todo:
        ent[1], ent[3], ent[4] = owner._version, ev, raw
    return len(todo)


def refresh_filters():
    """Called by the optimizer right after it moved the weights: the kept filter transforms of the active tier, rewritten in one launch."""
    return refresh_bf16_filters() if CONV_PREC == 2 else refresh_f32_filters()


# ---- memory ---------------------------------------------------------------------------------------------------------
def mem_read_fwd(x, mem, noise=None):
    n, h, w, d = x.shape
    m = mem.shape[0]
    qr = torch.empty((n, h, w, 2 * d), dtype=torch.float32, device=x.device)
    score = torch.empty((n * h * w, m), dtype=torch.float32, device=x.device)
    pmem = torch.empty_like(score)
    check(_lib().pm_mem_read_fwd(byref(tdesc(x)), mem.data_ptr(), m, ptr(noise), byref(tdesc(qr)), score.data_ptr(), pmem.data_ptr(), stream()), 'pm_mem_read_fwd')
    return qr, score, pmem


def mem_read_fwd_pq(x, mem, noise=None, noise_q=None):
    """mem_read_fwd + the softmax over all queries (p_query) from the read kernel's column partials: two launches instead of three."""
    n, h, w, d = x.shape
    m = mem.shape[0]
    qr = torch.empty((n, h, w, 2 * d), dtype=torch.float32, device=x.device)
    score = torch.empty((n * h * w, m), dtype=torch.float32, device=x.device)
    pmem, pq = torch.empty_like(score), torch.empty_like(score)
    lib = _lib()
    nb = lib.pm_mem_read_fwd_pq_workspace(n * h * w, m)
    ws = workspace(nb, x.device)
    check(lib.pm_mem_read_fwd_pq(byref(tdesc(x)), mem.data_ptr(), m, ptr(noise), ptr(noise_q), byref(tdesc(qr)), score.data_ptr(), pmem.data_ptr(), pq.data_ptr(),
                                 ptr(ws), nb, stream()), 'pm_mem_read_fwd_pq')
    return qr, score, pmem, pq


def mem_colsoftmax(score, noise=None):
    rows, m = score.shape
    out = torch.empty_like(score)
    lib = _lib()
    nb = lib.pm_mem_colsoftmax_workspace(rows, m)
    ws = workspace(nb, score.device)
    check(lib.pm_mem_colsoftmax(score.data_ptr(), ptr(noise), rows, m, out.data_ptr(), ptr(ws), nb, stream()), 'pm_mem_colsoftmax')
    return out


def mem_read_bwd(x, mem, pmem, dqr, dscore_extra=None, want_dmem=False):
    n, h, w, d = x.shape
    m = mem.shape[0]
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dmem = torch.empty_like(mem) if want_dmem else None
    lib = _lib()
    nb = lib.pm_mem_read_bwd_workspace(n * h * w, m, d) if want_dmem else 0
    ws = workspace(nb, x.device) if nb else None
    check(lib.pm_mem_read_bwd(byref(tdesc(x)), mem.data_ptr(), m, pmem.data_ptr(), byref(tdesc(dqr)), ptr(dscore_extra), byref(tdesc(dx)), ptr(dmem),
                              ptr(ws), nb, stream()), 'pm_mem_read_bwd')
    return dx, dmem


def mem_write_accum(z, labels, m, normalize=True):
    n, H, W = labels.shape
    d = z.shape[3]
    nomden = torch.empty((m + 1) * (d + 1), dtype=torch.float32, device=z.device)
    zd = tdesc(z)
    lib = _lib()
    nb = lib.pm_mem_write_accum_workspace(byref(zd), m)
    ws = workspace(nb, z.device)
    check(lib.pm_mem_write_accum(byref(zd), labels.data_ptr(), H, W, m, 1 if normalize else 0, nomden.data_ptr(), ptr(ws), nb, stream()), 'pm_mem_write_accum')
    return nomden


def mem_write_accum_bwd(z, labels, m, dnom, normalize=True):
    n, H, W = labels.shape
    dz = torch.empty(z.shape, dtype=torch.float32, device=z.device)
    check(_lib().pm_mem_write_accum_bwd(byref(tdesc(z)), labels.data_ptr(), H, W, m, 1 if normalize else 0, dnom.data_ptr(), byref(tdesc(dz)), stream()),
          'pm_mem_write_accum_bwd')
    return dz


def mem_write_update(mem, nomden, momentum, want_u=False):
    m, d = mem.shape
    out = torch.empty_like(mem)
    u = torch.empty_like(mem) if want_u else None
    check(_lib().pm_mem_write_update(mem.data_ptr(), nomden.data_ptr(), m, d, momentum, out.data_ptr(), ptr(u), stream()), 'pm_mem_write_update')
    return out, u


def mem_write_update_bwd(u, nomden, momentum, dout):
    m, d = u.shape
    dnom = torch.empty((m + 1, d), dtype=torch.float32, device=u.device)
    check(_lib().pm_mem_write_update_bwd(u.data_ptr(), nomden.data_ptr(), m, d, momentum, dout.data_ptr(), dnom.data_ptr(), None, stream()),
          'pm_mem_write_update_bwd')
    return dnom


def sgd_momentum(param, grad, buf, lr, momentum, wd, first):
    check(_lib().pm_sgd_momentum(param.data_ptr(), grad.data_ptr(), buf.data_ptr(), param.numel(), lr, momentum, wd, 1 if first else 0, stream()), 'pm_sgd_momentum')


def sgd_momentum_multi(triples, lr, momentum, wd, lr_device=None):
    """triples: [(param, grad, momentum_buffer)] of same-layout dense fp32 CUDA tensors; updated in place by a few launches.
    lr_device: a one-element fp32 CUDA tensor the kernel reads the learning rate from (captured steps); `lr` is then ignored."""
    arr = (L.PmSgdEntry * len(triples))()
    for i, (p, g, m) in enumerate(triples):
        arr[i].param, arr[i].grad, arr[i].momentum_buffer, arr[i].numel = p.data_ptr(), g.data_ptr(), m.data_ptr(), p.numel()
    check(_lib().pm_sgd_momentum_multi_dev(arr, len(triples), lr, ptr(lr_device), momentum, wd, stream()), 'pm_sgd_momentum_multi')


def set_winograd(mode):
    """Winograd route of the wide stride-1 3x3 convs: 4 / True = prefer F(4x4,3x3) (default), 2 = F(2x2,3x3) only, 0 / False = direct."""
    mode = 4 if mode is True else (0 if mode is False else int(mode))
    check(_lib().pm_set_winograd(mode), 'pm_set_winograd')
    _GEN[0] += 1


def set_winograd_fused(on):
    """F(4x4) layers: GEMMs + output transform in one kernel (opt-in; slower than the two-pass form on the flagship layers)."""
    check(_lib().pm_set_winograd_fused(1 if on else 0), 'pm_set_winograd_fused')
    _GEN[0] += 1


def set_conv16(on):
    """bf16 tier, forward / stride-1 data gradient: 1 = per shape (default: the LDS-DMA kernels where they win), 2 = LDS-DMA everywhere on the narrow tiles,
    0 = register-staged everywhere; 3 / 7 / 8 = the wide kernels wherever the shape allows (persistent ring / 256 x 256 / ring with one block per tile), 4-6 A/B
    routes (include/pinmem_hip.h pm_set_conv16)."""
    check(_lib().pm_set_conv16(int(on)), 'pm_set_conv16')
    _GEN[0] += 1


def set_wgrad16(on):
    """bf16 tier, weight gradient of bf16 x / dy: True (default) = the LDS-DMA persistent ring (csrc/wgrad16.hip) where the shape allows, False = register-staged everywhere."""
    check(_lib().pm_set_wgrad16(1 if on else 0), 'pm_set_wgrad16')
    _GEN[0] += 1


def routing(**fields):
    """The library's routing state (include/pinmem_hip.h pm_routing) as a dict; keyword arguments replace fields -- ONE pm_routing_set call, the form a production caller
    uses (once, before the first convolution). Returns the state before the change."""
    from .lib import PmRouting
    r = PmRouting()
    r.struct_size = ctypes.sizeof(PmRouting)
    check(_lib().pm_routing_get(byref(r)), 'pm_routing_get')
    before = {k: getattr(r, k) for k, _ in PmRouting._fields_ if k != 'struct_size'}
    if fields:
        for k, v in fields.items():
            if k not in before:
                raise KeyError('pm_routing has no field %r' % k)
            setattr(r, k, int(v))
        check(_lib().pm_routing_set(byref(r)), 'pm_routing_set')
        _GEN[0] += 1
    return before


def set_split(on):
    """fp32 tier: True (default) = eligible fp32 convolutions / Winograd GEMMs on the bf16 matrix pipe (three-way exact bf16 split, six products, fp32 accumulation:
    csrc/conv_split.hip), False = the fp32-MFMA kernel everywhere."""
    check(_lib().pm_set_split(1 if on else 0), 'pm_set_split')
    _GEN[0] += 1


def set_bf16_wgrad(on):
    check(_lib().pm_set_bf16_wgrad(1 if on else 0), 'pm_set_bf16_wgrad')
    _GEN[0] += 1


def profile_enable(on):
    check(_lib().pm_profile_enable(1 if on else 0), 'pm_profile_enable')


def profile_dump(path):
    check(_lib().pm_profile_dump(str(path).encode()), 'pm_profile_dump')


def profile_read_bytes(mode=-1, bm=-1, bn=-1, km=-1, nst=-1, prec=-1):
    """-> algorithmic HBM bytes (operands once + output) summed over the recorded launches of one instantiation: what bench.py's roofline.traffic is held against."""
    b = ctypes.c_double(0)
    check(_lib().pm_profile_read_bytes(mode, bm, bn, km, nst, prec, byref(b)), 'pm_profile_read_bytes')
    return b.value


def profile_read(mode=-1, bm=-1, bn=-1, km=-1, nst=-1, clear=False, prec=-1):
    """-> (total_ms, total_flops, launches) of the conv_igemm_kernel<mode, bm, bn, .., km, prec, nst> launches recorded since the last clear."""
    import ctypes
    ms, fl, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
    check(_lib().pm_profile_read_prec(mode, bm, bn, km, nst, prec, byref(ms), byref(fl), byref(n), 1 if clear else 0), 'pm_profile_read_prec')
    return ms.value, fl.value, n.value
