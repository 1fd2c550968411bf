"""Autograd-wrapped HIP ops. Tensors crossing these functions are LOGICAL NCHW (the reference's shapes) whose
memory is NHWC (`t.permute(0,2,3,1)` is a dense pixel-major view): every call site keeps the reference's shape
arithmetic (`x.size()[2:]`, `dim=1` channel slices) while every kernel sees coalesced channel vectors.
Weights enter as Function inputs (never cached pointers) so functional re-wiring of module._parameters works."""
import os as _os

import torch

from . import kernels as K
from . import lib as L
from .. import dist as D


def nhwc(t):
    """Logical-NCHW tensor -> NHWC view usable by the kernels (zero-copy when the memory is already channels-last,
    one transpose kernel when a caller hands over a true NCHW-contiguous tensor)."""
    v = t.permute(0, 2, 3, 1)
    n, h, w, c = v.shape
    sn, sh, sw, sc = v.stride()
    pitch = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
    ok = (sc == 1 or c == 1) and pitch >= c and (h == 1 or w == 1 or sh == w * pitch) and (n == 1 or h * w == 1 or sn == h * w * pitch)
    if ok and v.dtype == torch.float32:
        return v
    if v.dtype == torch.bfloat16:      # the bf16 tier's activations (K.ACT_DTYPE): 16-byte rows of eight channels
        if ok and pitch % 8 == 0 and c % 8 == 0 and v.data_ptr() % 16 == 0:
            return v
        return v.contiguous()
    if h * w == 1:
        return v.contiguous().float()
    return K.nchw_to_nhwc(t.float())


def nchw(v):
    return v.permute(0, 3, 1, 2)


def _grad_view(g):
    """Incoming gradient (logical NCHW, arbitrary strides) -> NHWC view with 16B-aligned rows."""
    v = nhwc(g)
    if v.stride(2) % (8 if v.dtype == torch.bfloat16 else 4) or v.data_ptr() % 16:
        v = v.contiguous()
    return v


class BNState:
    """What the BN part of a fused op needs besides gamma/beta: the module's buffers and mode."""
    __slots__ = ('running_mean', 'running_var', 'eps', 'momentum', 'training', 'group', 'grad_enabled')

    def __init__(self, bn):
        self.running_mean, self.running_var = bn.running_mean, bn.running_var
        self.eps = bn.eps
        self.momentum = 0.1 if bn.momentum is None else bn.momentum
        self.training = bn.training or bn.running_mean is None
        # nn.SyncBatchNorm (train.py:95 convert_sync_batchnorm) -> exchange statistics over RCCL
        self.group = D.bn_group(bn)
        self.grad_enabled = torch.is_grad_enabled()      # the caller's mode: inside autograd.Function.forward it always reads False
        if bn.training and bn.num_batches_tracked is not None:
            _nbt_pending.append(bn.num_batches_tracked)


# num_batches_tracked += 1 of every BN layer that ran in training mode: one multi-tensor launch per forward pass (flush_bn_counters,
# called where the networks' forward ends) instead of 65 single-element kernels.
_nbt_pending = []


def flush_bn_counters():
    if _nbt_pending:
        torch._foreach_add_(_nbt_pending, 1)
        _nbt_pending.clear()


# ---- weight-gradient overlap ------------------------------------------------------------------------------------------
# wgrad(L) only needs x(L) and dy(L); nothing on the backward critical path (bn_bwd(L-1) -> dgrad(L-1) -> ...) needs its result.
# It is launched on a side stream so that the HBM-bound BatchNorm-backward kernels and the tile-quantisation tails of the dgrad
# kernels run under its MFMA work.
# Safety: the weight enters the conv node through a `_Defer` identity node created at the START of its stage's forward, i.e. with
# a lower autograd sequence number than every node of that stage. The engine therefore runs `_Defer.backward` -- which makes the
# main stream wait for the side stream and only then hands the gradient on -- after the whole stage's backward has been
# launched: AccumulateGrad / DDP hooks / functional-weight graphs never see a gradient that is still being written.
# Round 5, second session: on the bf16 tier the weight gradients run INLINE by default. Their kernel there (csrc/wgrad16.hip) is persistent -- one block per CU with
# 144 KB of LDS -- so a second stream cannot share the chip with it any more and the fork / join only adds contention (same box, captured step: 24.72 side stream,
# 24.46 inline; eager: level). PM_OVERLAP_WGRAD=0 / 1 forces one form on both tiers; the attribute stays the master switch (bench.py's serialised leg, tests).
_OVERLAP_ENV = _os.environ.get('PM_OVERLAP_WGRAD')
OVERLAP_WGRAD = True if _OVERLAP_ENV is None else _OVERLAP_ENV == '1'


def overlap_wgrad():
    """Do the weight gradients of the pass being recorded go to the side stream?"""
    if not OVERLAP_WGRAD:
        return False
    return not (_OVERLAP_ENV is None and K.CONV_PREC == 2 and K.ACT_DTYPE == torch.bfloat16)


_side = {}
_proxy = {}            # id(weight tensor) -> deferred proxy, valid for the current forward only


def masked_stream(n_cus, device=None, first=0):
    """A HIP stream whose kernels may only run on `n_cus` of the 256 CUs (hipExtStreamCreateWithCUMask through ctypes on the process's libamdhip64, wrapped as a
    torch ExternalStream) -- VERDICT r3 item 2a: a side stream that owns a fixed CU set instead of alternating whole-GPU kernels with the main stream. The mask
    takes every (256 / n_cus)-th CU starting at `first`, so the set is spread evenly over the 8 XCDs / 32 shader engines."""
    import ctypes
    dev = torch.cuda.current_device() if device is None else device
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    n_cus = max(1, min(int(n_cus), total))
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    picked = {(first + (i * total) // n_cus) % total for i in range(n_cus)}
    for cu in picked:
        mask[cu // 32] |= 1 << (cu % 32)
    hip = ctypes.CDLL('libamdhip64.so')
    st = ctypes.c_void_p()
    with torch.cuda.device(dev):
        err = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if err != 0 or not st.value:
        raise RuntimeError('hipExtStreamCreateWithCUMask(%d CUs) failed: %d' % (n_cus, err))
    return torch.cuda.ExternalStream(st.value, device=dev)


# PM_WGRAD_CUS=n: the weight-gradient side stream is confined to n CUs (A/B knob; default: an ordinary stream sharing all 256 with the main stream)
WGRAD_CUS = int(_os.environ.get('PM_WGRAD_CUS', '0'))


def _side_stream():
    dev = torch.cuda.current_device()
    if dev not in _side:
        _side[dev] = masked_stream(WGRAD_CUS, dev) if WGRAD_CUS > 0 else torch.cuda.Stream(device=dev)
    return _side[dev]


class _Defer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w):
        return w.view_as(w)

    @staticmethod
    def backward(ctx, g):
        L.stream_obj().wait_stream(_side_stream())
        return g


def begin_forward():
    flush_bn_counters()      # backstop for callers that drive sub-modules directly
    _proxy.clear()


def defer_weights(module):
    """Call right before running `module` (a stage of the network) in training mode with gradients enabled."""
    if not (overlap_wgrad() and torch.is_grad_enabled()):
        return
    for m in module.modules():
        if isinstance(m, torch.nn.Conv2d) and m.bias is None and m.weight.requires_grad and id(m.weight) not in _proxy:
            _proxy[id(m.weight)] = _Defer.apply(m.weight)


def _w(weight):
    """(tensor the conv node should take as its weight input, whether its wgrad may run asynchronously)"""
    p = _proxy.get(id(weight))
    return (p, True) if p is not None else (weight, False)


def _wgrad(x, dy, wshape, geom, want_bias=False, deferred=False, wino_v=None):
    if not deferred:
        return K.conv_bwd_weight(x, dy, wshape, *geom, want_bias=want_bias, wino_v=wino_v)
    # on the side stream, behind everything the main stream has enqueued so far; the launch takes the stream explicitly (no torch.cuda.stream() context: its
    # enter / exit cost 10 us per weight gradient on a host-bound step). Outputs are allocated by the main stream's pool and first written on the side stream.
    side, main = _side_stream(), L.stream_obj()
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)
    out = K.conv_bwd_weight(x, dy, wshape, *geom, want_bias=want_bias, wino_v=wino_v, on_stream=side)
    if wino_v is not None:
        wino_v.record_stream(side)
    x.record_stream(side)
    dy.record_stream(side)
    for t in out:
        if t is not None:
            t.record_stream(side)
    return out


# harness.agg_train_step runs the memory-commit forward on its own stream; it READS every weight there. device index -> event recorded at its end.
# Whoever writes weights on another stream waits for it first (optim.SGD.step, checkpoint.restore_snapshot, harness.sync_commit).
commit_done = {}


def wait_commit():
    if not commit_done or not torch.cuda.is_available():
        return
    ev = commit_done.pop(torch.cuda.current_device(), None)
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


BN_FUSED_FINALIZE = _os.environ.get('PM_BN_FUSED', '1') == '1'      # A/B knob: 0 = separate bn_stats / bn_finalize launches


RELU_MASK_BYTES = _os.environ.get('PM_BN_MASK', '1') == '1'      # A/B knob: 0 = the residual BatchNorms' backward re-reads the forward output for its ReLU mask


MERGE_SYNCBN = _os.environ.get('PM_MERGE_SYNCBN', '1') == '1'      # A/B knob: 0 = one exchange per BatchNorm (rounds 1-5)


def _bn_train_stats_multi(ys, bns, partials):
    """Batch statistics of SEVERAL independent SyncBatchNorms of equal width in ONE exchange (ASPP's five branches, a stage's first block's bn3 +
    downsample.1): local moments of each, one all-gather of their concatenation, one merge + finalise per layer. -> [(mean, invstd)]; running moments updated in place.
    Same arithmetic per layer as _bn_train_fwd's SyncBN branch (the merge kernel sees the same per-rank moments in the same rank order)."""
    c = ys[0].shape[3]
    moms = []
    for y, part in zip(ys, partials):
        pixels = y.shape[0] * y.shape[1] * y.shape[2]
        moms.append(K.bn_partials_moments(part, pixels, c) if part is not None else K.bn_stats(y))
    n = len(ys)
    flat, world = D.gather_moments(torch.cat(moms), bns[0].group)
    parts = flat.view(world, n, 3 * c)
    return [K.bn_merge_finalize(parts[:, i].contiguous().view(-1), world, c, bn.eps, bn.running_mean, bn.running_var, bn.momentum) for i, bn in enumerate(bns)]


def _mergeable(bns, ys):
    """One exchange for these layers? Every one a SyncBatchNorm of the same process group and width, more than one rank, the fused finalise on."""
    return (MERGE_SYNCBN and BN_FUSED_FINALIZE and D.is_dist() and len(bns) > 1 and all(b.group is not None and b.group is bns[0].group for b in bns)
            and len({y.shape[3] for y in ys}) == 1)


def _bn_train_fwd(y, gamma, beta, bn, residual, relu, out=None, partials=None, want_mask=False):
    """Batch statistics (merged across ranks for SyncBN) -> normalise + residual + ReLU. Returns (o, mean, invstd).
    partials: the (mean, M2) slab partials the producing convolution's epilogue emitted for y (K.conv_fwd(bn_partials=...)), or None:
    with them no kernel re-reads y for its statistics."""
    c = y.shape[3]
    pixels = y.shape[0] * y.shape[1] * y.shape[2]
    if bn.group is None and BN_FUSED_FINALIZE:
        if partials is not None:
            mean, invstd = K.bn_partials_finalize(partials, pixels, c, bn.eps, bn.running_mean, bn.running_var, bn.momentum)
        else:
            mean, invstd = K.bn_stats_finalize(y, bn.eps, bn.running_mean, bn.running_var, bn.momentum)
    elif bn.group is not None and D.is_dist() and BN_FUSED_FINALIZE:
        mom = K.bn_partials_moments(partials, pixels, c) if partials is not None else K.bn_stats(y)
        flat, world = D.gather_moments(mom, bn.group)                  # SyncBN: stats, all-gather, merge + finalise, apply
        mean, invstd = K.bn_merge_finalize(flat, world, c, bn.eps, bn.running_mean, bn.running_var, bn.momentum)
    else:
        mom = K.bn_partials_moments(partials, pixels, c) if partials is not None else K.bn_stats(y)
        if bn.group is not None:
            mom = D.merge_moments(mom, c, bn.group)
        mean, invstd = K.bn_finalize(mom, c, bn.eps, bn.running_mean, bn.running_var, bn.momentum)
    if want_mask:      # BN + residual + ReLU: one byte per float4 group tells the backward which gradients the ReLU passes (1 / 16 of re-reading the output)
        o, mask = K.bn_apply(y, mean, invstd, gamma, beta, residual=residual, relu=relu, out=out, want_mask=True)
        return o, mean, invstd, mask
    return K.bn_apply(y, mean, invstd, gamma, beta, residual=residual, relu=relu, out=out), mean, invstd


def _bn_train_bwd(dv, o, y, mean, invstd, gamma, beta, relu, group, want_dres, has_res, mask=None):
    """-> (dy, dres, dgamma, dbeta): gradient of the conv output, of the residual input, and of the affine parameters.
    ReLU mask: rebuilt from the conv output when no residual entered the activation (the forward output is not read at all);
    otherwise taken from the forward output once, in the reduce pass, which then hands the masked gradient (= dres) to the apply pass."""
    c = y.shape[3]
    mode = 0 if not relu else (1 if has_res else 2)
    hand_over = mode == 1 and want_dres
    if hand_over and mask is not None:
        sums, gm = K.bn_bwd_reduce_mask(dv, mask, y, mean, invstd, want_gmask=True, with_count=group is not None)
    else:
        sums, gm = K.bn_bwd_reduce(dv, o, y, mean, invstd, mode, gamma, beta, want_gmask=hand_over, with_count=group is not None)
    local = sums
    if group is not None:
        # dgamma / dbeta are THIS rank's sums (torch.nn.SyncBatchNorm returns the local grad_weight / grad_bias and lets DDP average them);
        # only the input gradient needs the sums over all ranks
        sums = D.all_reduce_sum_copy(local, group)  # [sum dy | sum dy * xhat | count]: the global count travels with the sums (uneven batches); out of place
    count = float(y.shape[0] * y.shape[1] * y.shape[2]) if group is None else -1.0
    if hand_over:
        dy, _ = K.bn_bwd_apply(gm, None, y, mean, invstd, gamma, sums, count, 0, False)
        dres = gm
    else:
        dy, dres = K.bn_bwd_apply(dv, o, y, mean, invstd, gamma, sums, count, mode, want_dres, beta)
    return dy, dres, local[c:2 * c], local[:c]


class _Bottleneck(torch.autograd.Function):
    """One ResNet Bottleneck (Resnet.py:181-216) as a single autograd node in train mode: the gradient of the block input is
    produced by conv1's dgrad with the skip-path gradient fused in (`add`), so autograd never launches a separate add, and the
    16 blocks cost 16 nodes instead of ~60."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd, geoms, bns, deferred):
        xv = nhwc(x)
        k1, k2, k3 = K.krsc(w1), K.krsc(w2), K.krsc(w3)
        ps = []                                            # BatchNorm statistics handed out by the convolution epilogues (None: separate pass)
        y1 = K.conv_fwd(xv, k1, *geoms[0], bn_partials=ps)
        o1, m1, i1 = _bn_train_fwd(y1, g1, b1, bns[0], None, True, partials=ps[0])
        kv = []
        y2 = K.conv_fwd(o1, k2, *geoms[1], keep_v=kv, bn_partials=ps)      # Winograd layers keep the transformed input for the weight gradient
        o2, m2, i2 = _bn_train_fwd(y2, g2, b2, bns[1], None, True, partials=ps[1])
        y3 = K.conv_fwd(o2, k3, *geoms[2], bn_partials=ps)
        merged = False
        if wd is not None:
            kd = K.krsc(wd)
            yd = K.conv_fwd(xv, kd, *geoms[3], bn_partials=ps)
            merged = _mergeable([bns[3], bns[2]], [yd, y3])
            if merged:      # SyncBN: the two independent statistics exchanges of the block's tail travel as one
                (md, idd), (m3, i3) = _bn_train_stats_multi([yd, y3], [bns[3], bns[2]], [ps[3], ps[2]])
                res = K.bn_apply(yd, md, idd, gd, bd, residual=None, relu=False)
            else:
                res, md, idd = _bn_train_fwd(yd, gd, bd, bns[3], None, False, partials=ps[3])
        else:
            kd = yd = md = idd = None
            res = xv
        if merged:
            out, mask3 = K.bn_apply(y3, m3, i3, g3, b3, residual=res, relu=True, want_mask=True) if RELU_MASK_BYTES else (K.bn_apply(y3, m3, i3, g3, b3, residual=res, relu=True), None)
        elif RELU_MASK_BYTES:
            out, m3, i3, mask3 = _bn_train_fwd(y3, g3, b3, bns[2], res, True, partials=ps[2], want_mask=True)
        else:
            (out, m3, i3), mask3 = _bn_train_fwd(y3, g3, b3, bns[2], res, True, partials=ps[2]), None
        ctx.geoms, ctx.groups, ctx.has_ds, ctx.deferred = geoms, [b.group for b in bns], wd is not None, deferred
        ctx.save_for_backward(xv, k1, k2, k3, kd, y1, o1, y2, o2, y3, yd, out, m1, i1, m2, i2, m3, i3, md, idd, g1, g2, g3, gd, b1, b2, b3, kv[0], mask3)
        return nchw(out)

    @staticmethod
    def backward(ctx, dout):
        xv, k1, k2, k3, kd, y1, o1, y2, o2, y3, yd, out, m1, i1, m2, i2, m3, i3, md, idd, g1, g2, g3, gd, b1, b2, b3, v2, mask3 = ctx.saved_tensors
        ge, gr, df = ctx.geoms, ctx.groups, ctx.deferred
        dv = _grad_view(dout)
        pair = None
        if ctx.has_ds and MERGE_SYNCBN and D.is_dist() and gr[2] is not None and gr[3] is gr[2] and y3.shape[3] == yd.shape[3]:
            # SyncBN, first block of a stage: bn3's masked gradient (= the skip gradient) is known after bn3's REDUCE pass, before its exchange -- so the downsample
            # BatchNorm's reduce pass runs on it right away and the two sum exchanges travel as ONE all-reduce; then both apply passes.
            c3 = y3.shape[3]
            if mask3 is not None:
                s3, gm = K.bn_bwd_reduce_mask(dv, mask3, y3, m3, i3, want_gmask=True, with_count=True)
            else:
                s3, gm = K.bn_bwd_reduce(dv, out, y3, m3, i3, 1, g3, b3, want_gmask=True, with_count=True)
            sd, _ = K.bn_bwd_reduce(gm, None, yd, md, idd, 0, gd, None, want_gmask=False, with_count=True)
            both = D.all_reduce_sum_copy(torch.cat([s3, sd]), gr[2])
            g3sum, gdsum = both[:s3.numel()], both[s3.numel():]
            dy3, _ = K.bn_bwd_apply(gm, None, y3, m3, i3, g3, g3sum, -1.0, 0, False)
            dres, dg3, db3 = gm, s3[c3:2 * c3], s3[:c3]
            dyd, _ = K.bn_bwd_apply(gm, None, yd, md, idd, gd, gdsum, -1.0, 0, False, None)
            pair = (dyd, sd[c3:2 * c3], sd[:c3])
        else:
            dy3, dres, dg3, db3 = _bn_train_bwd(dv, out, y3, m3, i3, g3, b3, True, gr[2], True, True, mask=mask3)
        dw3, _ = _wgrad(o2, dy3, tuple(k3.shape), ge[2], deferred=df[2])
        do2 = K.conv_bwd_data(dy3, k3, tuple(o2.shape), *ge[2])
        dy2, _, dg2, db2 = _bn_train_bwd(do2, o2, y2, m2, i2, g2, b2, True, gr[1], False, False)
        dw2, _ = _wgrad(o1, dy2, tuple(k2.shape), ge[1], deferred=df[1], wino_v=v2)
        do1 = K.conv_bwd_data(dy2, k2, tuple(o1.shape), *ge[1])
        dy1, _, dg1, db1 = _bn_train_bwd(do1, o1, y1, m1, i1, g1, b1, True, gr[0], False, False)
        dw1, _ = _wgrad(xv, dy1, tuple(k1.shape), ge[0], deferred=df[0])
        dwd = dgd = dbd = None
        skip = dres
        if ctx.has_ds:
            if pair is not None:
                dyd, dgd, dbd = pair
            else:
                dyd, _, dgd, dbd = _bn_train_bwd(dres, None, yd, md, idd, gd, None, False, gr[3], False, False)
            dwdk, _ = _wgrad(xv, dyd, tuple(kd.shape), ge[3], deferred=df[3])
            dwd = dwdk.permute(0, 3, 1, 2)
            skip = K.conv_bwd_data(dyd, kd, tuple(xv.shape), *ge[3]) if ctx.needs_input_grad[0] else None
        dx = nchw(K.conv_bwd_data(dy1, k1, tuple(xv.shape), *ge[0], add=skip)) if ctx.needs_input_grad[0] else None
        p = lambda d: d.permute(0, 3, 1, 2)
        return dx, p(dw1), dg1, db1, p(dw2), dg2, db2, p(dw3), dg3, db3, dwd, dgd, dbd, None, None, None


# ---- eval-mode forward: every BatchNorm folded in ONE launch ---------------------------------------------------------------------
# The no-grad eval forward folds each BN into its conv's epilogue; doing that per layer puts 64 four-microsecond kernels (plus their
# launch boundaries) on the critical chain of the memory-commit forward. `prefold(model)` folds all of them at once when the forward
# starts; _ConvBnAct picks its (scale, shift) views from the cache (keyed by the running_mean storage) and falls back to the per-layer
# fold for anything not covered (a conv with a bias, a module swapped in later). PM_FOLD_BATCH=0 disables it.
FOLD_BATCH = _os.environ.get('PM_FOLD_BATCH', '1') == '1'
_fold_cache = None
fold_misses = 0
last_prefold_event = None      # recorded right after the one fold launch: from there on the forward no longer reads any BatchNorm running moment


class prefold:
    def __init__(self, model):
        self.model, self.active = model, False

    def __enter__(self):
        global _fold_cache, last_prefold_event
        last_prefold_event = None
        m = self.model
        if not FOLD_BATCH or m.training or torch.is_grad_enabled() or _fold_cache is not None:
            return self
        bns = getattr(m, '_pm_bn_list', None)
        if bns is None:
            bns = [b for b in m.modules() if isinstance(b, torch.nn.modules.batchnorm._BatchNorm)]
            m.__dict__['_pm_bn_list'] = bns
        bns = [b for b in bns if b.running_mean is not None and b.weight is not None and b.running_mean.is_cuda and not b.training]
        if not bns or len({b.eps for b in bns}) != 1:
            return self
        ptrs = tuple(t.data_ptr() for b in bns for t in (b.weight, b.bias, b.running_mean, b.running_var))
        plan = m.__dict__.get('_pm_fold_plan')
        if plan is None or plan[0] != ptrs:            # pointers only change when parameters / buffers are re-created (load, .to, put_theta)
            dev = bns[0].running_mean.device
            cs = [b.num_features for b in bns]
            offs, tot = [], 0
            for c in cs:
                offs.append(tot)
                tot += (c + 3) // 4 * 4                # 16-byte aligned views
            as_i64 = [p - (1 << 64) if p >= (1 << 63) else p for p in ptrs]
            plan = (ptrs, torch.tensor(as_i64, dtype=torch.int64, device=dev), torch.tensor(cs, dtype=torch.int32, device=dev),
                    torch.tensor(offs, dtype=torch.int32, device=dev), cs, offs, tot)
            m.__dict__['_pm_fold_plan'] = plan
        _, table, dcs, doffs, cs, offs, tot = plan
        arena = K.bn_fold_multi(table, dcs, doffs, len(cs), max(cs), tot, bns[0].eps)
        _fold_cache = {b.running_mean.data_ptr(): (arena[o:o + c], arena[tot + o:tot + o + c]) for b, c, o in zip(bns, cs, offs)}
        self.active = True
        last_prefold_event = torch.cuda.current_stream().record_event() if len(bns) == len(m.__dict__['_pm_bn_list']) else None
        return self

    def __exit__(self, *exc):
        global _fold_cache
        if self.active:
            _fold_cache = None
        return False


class _ConvBnAct(torch.autograd.Function):
    """conv -> BatchNorm(train: batch statistics | eval: folded into the conv epilogue) -> (+residual) -> ReLU.
    Replaces e.g. Resnet.py:195-216 conv/bn/relu triples and every Sequential(Conv2d, Norm2d, ReLU) of deepv3plus.py."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, residual, geom, bn, relu, out, deferred=False):
        stride, pad, dil = geom
        xv, wk = nhwc(x), K.krsc(w)
        rv = nhwc(residual) if residual is not None else None
        ov = nhwc(out) if out is not None else None
        ctx.geom, ctx.relu, ctx.train, ctx.deferred = geom, relu, bn.training, deferred and bias is None
        needs_grad = bn.grad_enabled and any(ctx.needs_input_grad[:6])   # needs_input_grad alone ignores torch.no_grad()
        if not bn.training and not needs_grad:     # inference: BN folded into the conv epilogue, nothing saved
            hit = _fold_cache.get(bn.running_mean.data_ptr()) if (_fold_cache is not None and bias is None) else None
            if hit is None:
                global fold_misses
                fold_misses += 1                   # this forward reads running moments after its fold launch (harness: no early hand-back of the main stream)
            scale, shift = hit if hit is not None else K.bn_fold(gamma, beta, bn.running_mean, bn.running_var, bn.eps, bias)
            o = K.conv_fwd(xv, wk, stride, pad, dil, scale=scale, shift=shift, residual=rv, relu=relu, out=ov)
            return nchw(o)
        kv, ps = [], []
        y = K.conv_fwd(xv, wk, stride, pad, dil, bias=bias, keep_v=kv if ctx.needs_input_grad[1] else None, bn_partials=ps if bn.training else None)
        if bn.training:
            o, mean, invstd = _bn_train_fwd(y, gamma, beta, bn, rv, relu, ov, partials=ps[0])
        else:                                      # frozen statistics with a graph: normalise by the running moments, keep y for backward
            mean, invstd = bn.running_mean, torch.rsqrt(bn.running_var + bn.eps)
            o = K.bn_apply(y, mean, invstd, gamma, beta, residual=rv, relu=relu, out=ov)
        ctx.group, ctx.has_bias, ctx.has_res = bn.group, bias is not None, residual is not None
        ctx.save_for_backward(xv, wk, y, o, mean, invstd, gamma, beta, kv[0] if kv else None)
        return nchw(o)

    @staticmethod
    def backward(ctx, dout):
        xv, wk, y, o, mean, invstd, gamma, beta, vk = ctx.saved_tensors
        stride, pad, dil = ctx.geom
        dv = _grad_view(dout)
        want_dres = ctx.has_res and ctx.needs_input_grad[5]
        if ctx.train:
            dy, dres, dgamma, dbeta = _bn_train_bwd(dv, o, y, mean, invstd, gamma, beta, ctx.relu, ctx.group, want_dres, ctx.has_res)
        else:
            # frozen statistics (eval mode): xhat does not depend on the batch, so dy = g * gamma * invstd with g the ReLU-masked
            # gradient; dgamma = sum(g * xhat), dbeta = sum(g) come from the same reduce pass as in training
            c = y.shape[3]
            mode = 0 if not ctx.relu else (1 if ctx.has_res else 2)
            sums, g = K.bn_bwd_reduce(dv, o, y, mean, invstd, mode, gamma, beta, want_gmask=mode != 0)
            g = dv if g is None else g
            dy = K.scale_shift_act(g, gamma * invstd, torch.zeros_like(gamma))
            dres, dgamma, dbeta = (g if want_dres else None), sums[c:], sums[:c]
        dx = nchw(K.conv_bwd_data(dy, wk, tuple(xv.shape), stride, pad, dil, dtype=xv.dtype)) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dwk, db = _wgrad(xv, dy, tuple(wk.shape), (stride, pad, dil), want_bias=ctx.has_bias, deferred=ctx.deferred, wino_v=vk)
            dw = dwk.permute(0, 3, 1, 2)
        return dx, dw, db, dgamma, dbeta, (nchw(dres) if dres is not None else None), None, None, None, None, None


class _ConvBnActN(torch.autograd.Function):
    """N independent conv -> SyncBatchNorm(train) -> ReLU branches of equal width as ONE autograd node whose BatchNorm exchanges travel together: the
    five branches of the ASPP head (deepv3plus.py:72-101: four convolutions of the trunk output + the image-pooling branch). Forward: every convolution, every local
    statistic, ONE all-gather, every finalise + apply; backward: every reduce pass, ONE all-reduce, every apply pass, then the data / weight gradients. Per branch the
    same kernels on the same values as _ConvBnAct -- only the number of collectives changes (10 -> 2 for the ASPP head). Used only with more than one rank."""

    @staticmethod
    def forward(ctx, n, geoms, bns, outs, deferreds, *ts):
        xs, ws, gammas, betas = ts[0:n], ts[n:2 * n], ts[2 * n:3 * n], ts[3 * n:4 * n]
        xvs, wks = [nhwc(x) for x in xs], [K.krsc(w) for w in ws]
        ys, kvs, parts = [], [], []
        for i in range(n):
            kv, ps = [], []
            ys.append(K.conv_fwd(xvs[i], wks[i], *geoms[i], keep_v=kv if ctx.needs_input_grad[5 + n + i] else None, bn_partials=ps))
            kvs.append(kv[0] if kv else None)
            parts.append(ps[0])
        stats = _bn_train_stats_multi(ys, bns, parts)
        os_ = [K.bn_apply(ys[i], stats[i][0], stats[i][1], gammas[i], betas[i], residual=None, relu=True, out=nhwc(outs[i]) if outs[i] is not None else None) for i in range(n)]
        ctx.n, ctx.geoms, ctx.group, ctx.deferreds = n, geoms, bns[0].group, deferreds
        ctx.save_for_backward(*xvs, *wks, *ys, *os_, *[st[0] for st in stats], *[st[1] for st in stats], *gammas, *betas, *kvs)
        return tuple(nchw(o) for o in os_)

    @staticmethod
    def backward(ctx, *douts):
        n = ctx.n
        sv = ctx.saved_tensors
        xvs, wks, ys, os_, means, invs, gammas, betas, kvs = (sv[i * n:(i + 1) * n] for i in range(9))
        c = ys[0].shape[3]
        red = [K.bn_bwd_reduce(_grad_view(douts[i]), os_[i], ys[i], means[i], invs[i], 2, gammas[i], betas[i], want_gmask=False, with_count=True) for i in range(n)]
        tot = D.all_reduce_sum_copy(torch.cat([r[0] for r in red]), ctx.group)
        ln = red[0][0].numel()
        dxs, dws, dgs, dbs = [], [], [], []
        for i in range(n):
            dy, _ = K.bn_bwd_apply(_grad_view(douts[i]), os_[i], ys[i], means[i], invs[i], gammas[i], tot[i * ln:(i + 1) * ln], -1.0, 2, False, betas[i])
            dxs.append(nchw(K.conv_bwd_data(dy, wks[i], tuple(xvs[i].shape), *ctx.geoms[i], dtype=xvs[i].dtype)) if ctx.needs_input_grad[5 + i] else None)
            dwk, _ = _wgrad(xvs[i], dy, tuple(wks[i].shape), ctx.geoms[i], deferred=ctx.deferreds[i], wino_v=kvs[i])
            dws.append(dwk.permute(0, 3, 1, 2))
            dgs.append(red[i][0][c:2 * c]), dbs.append(red[i][0][:c])
        return (None, None, None, None, None, *dxs, *dws, *dgs, *dbs)


def conv_bn_act_n(xs, seqs, outs):
    """[conv_bn_act(x, seq[0], seq[1], relu=True, out=o)] for independent (x, Sequential(conv, bn, relu)) branches; ONE SyncBatchNorm exchange per direction when the
    branches qualify (training, gradients on, more than one rank, equal width, no conv bias), the per-branch path otherwise."""
    convs, bns_m = [q[0] for q in seqs], [q[1] for q in seqs]
    ok = MERGE_SYNCBN and BN_FUSED_FINALIZE and D.is_dist() and torch.is_grad_enabled() and all(b.training for b in bns_m) and all(cv.bias is None for cv in convs)
    if ok:
        groups = [D.bn_group(b) for b in bns_m]
        ok = all(g is not None and g is groups[0] for g in groups) and len({cv.out_channels for cv in convs}) == 1
    if not ok:
        return [conv_bn_act(x, cv, bn, relu=True, out=o) for x, cv, bn, o in zip(xs, convs, bns_m, outs)]
    states = [BNState(b) for b in bns_m]      # (registers each layer's num_batches_tracked increment: once, here)
    wd = [_w(cv.weight) for cv in convs]
    n = len(xs)
    return list(_ConvBnActN.apply(n, [_geom(cv) for cv in convs], states, list(outs), [d for _, d in wd], *xs, *[w for w, _ in wd],
                                  *[b.weight for b in bns_m], *[b.bias for b in bns_m]))


class _Conv(torch.autograd.Function):
    """Plain convolution (+bias): final2 / dsn.4 (deepv3plus.py:416-417,424)."""

    @staticmethod
    def forward(ctx, x, w, bias, geom):
        stride, pad, dil = geom
        xv, wk = nhwc(x), K.krsc(w)
        y = K.conv_fwd(xv, wk, stride, pad, dil, bias=bias, out_dtype=torch.float32)      # the class logits (and every other un-normalised head) stay fp32 on the bf16 tier
        ctx.geom, ctx.has_bias = geom, bias is not None
        ctx.save_for_backward(xv, wk)
        return nchw(y)

    @staticmethod
    def backward(ctx, dout):
        xv, wk = ctx.saved_tensors
        stride, pad, dil = ctx.geom
        dv = _grad_view(dout)
        dx = nchw(K.conv_bwd_data(dv, wk, tuple(xv.shape), stride, pad, dil, dtype=xv.dtype)) if ctx.needs_input_grad[0] else None
        dwk, db = _wgrad(xv, dv, tuple(wk.shape), (stride, pad, dil), want_bias=ctx.has_bias)
        return dx, dwk.permute(0, 3, 1, 2), db, None


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xv = nhwc(x)
        y, arg = K.maxpool_fwd(xv)
        ctx.shape = tuple(xv.shape)
        ctx.save_for_backward(arg)
        return nchw(y)

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        return nchw(K.maxpool_bwd(_grad_view(dout), arg, ctx.shape))


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xv = nhwc(x)
        ctx.shape = tuple(xv.shape)
        return nchw(K.global_avgpool_fwd(xv))

    @staticmethod
    def backward(ctx, dout):
        return nchw(K.global_avgpool_bwd(_grad_view(dout), ctx.shape))


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, out):
        xv = nhwc(x)
        ctx.shape = tuple(xv.shape)
        return nchw(K.resize_fwd(xv, (int(size[0]), int(size[1])), out=nhwc(out) if out is not None else None))

    @staticmethod
    def backward(ctx, dout):
        return nchw(K.resize_bwd(_grad_view(dout), ctx.shape)), None, None


class _Concat(torch.autograd.Function):
    """The branches already wrote their channel slices of `buf`; this node only ties the graph together and
    hands each branch its slice of the incoming gradient (no copy either way)."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.widths = [p.shape[1] for p in parts]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, dout):
        outs, off = [], 0
        for wd in ctx.widths:
            outs.append(dout[:, off:off + wd])
            off += wd
        return (None,) + tuple(outs)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return nchw(K.add(nhwc(a), nhwc(b)))

    @staticmethod
    def backward(ctx, g):
        return g, g


class _Fanout(torch.autograd.Function):
    """n aliases of a tensor that feeds n branches (the ASPP input, deepv3plus.py:72-95): their gradients are summed left to right
    in ONE pass (pm_add_n) instead of autograd's chain of n - 1 two-operand adds over the 151 MB feature map."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [_grad_view(g) for g in gs if g is not None]
        return (nchw(K.add_n(gs)) if len(gs) > 1 else nchw(gs[0])), None


def fanout(x, n):
    if n < 2 or n > 8 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return _Fanout.apply(x, n)


CE_FUSED = _os.environ.get('PM_CE_FUSED', '1') == '1'      # A/B knob: 0 = forward and backward each sweep labels and logits


class _UpsampleCE(torch.autograd.Function):
    """mean CE(ignore 255) of bilinearly up-sampled logits vs full-resolution labels, logits never materialised. With a graph attached the
    forward also leaves the column-reduced gradient field (K.upsample_ce_fwd_field), so the backward is one short row pass."""

    @staticmethod
    def forward(ctx, logits, labels, inv_temp, want_grad):
        lv = nhwc(logits)
        labels = labels.contiguous()
        ctx.inv_temp, ctx.hw = inv_temp, tuple(labels.shape[1:])
        if want_grad and CE_FUSED:
            out, field = K.upsample_ce_fwd_field(lv, labels, inv_temp)
            ctx.fused = True
            ctx.save_for_backward(lv, out, field)
        else:
            out = K.upsample_ce_fwd(lv, labels, inv_temp)
            ctx.fused = False
            ctx.save_for_backward(lv, labels, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        gs = g.reshape(1).float().contiguous()
        if ctx.fused:
            lv, out, field = ctx.saved_tensors
            return nchw(K.upsample_ce_bwd_field(lv, ctx.hw, out, field, gs, ctx.inv_temp)), None, None, None
        lv, labels, out = ctx.saved_tensors
        return nchw(K.upsample_ce_bwd(lv, labels, out, gs, ctx.inv_temp)), None, None, None


class _MemRead(torch.autograd.Function):
    """memory.py:317-336. Returns (qr [B,2d,h,w], score [B,h,w,m], p_mem [B,h,w,m])."""

    @staticmethod
    def forward(ctx, x, mem, noise, noise_q=None, with_pq=False):
        xv = nhwc(x)
        mem = mem.contiguous()
        if with_pq:
            qr, score, pmem, pq = K.mem_read_fwd_pq(xv, mem, noise, noise_q)
        else:
            qr, score, pmem = K.mem_read_fwd(xv, mem, noise)
        n, h, w, _ = xv.shape
        m = mem.shape[0]
        ctx.save_for_backward(xv, mem, pmem)
        score, pmem = score.view(n, h, w, m), pmem.view(n, h, w, m)
        ctx.with_pq = with_pq
        if with_pq:
            pq = pq.view(n, h, w, m)
            ctx.mark_non_differentiable(pmem, pq)
            return nchw(qr), score, pmem, pq
        ctx.mark_non_differentiable(pmem)
        return nchw(qr), score, pmem

    @staticmethod
    def backward(ctx, dqr, dscore, _dp, _dq=None):
        xv, mem, pmem = ctx.saved_tensors
        n, h, w, d = xv.shape
        dq = _grad_view(dqr) if dqr is not None else torch.zeros((n, h, w, 2 * d), dtype=torch.float32, device=xv.device)
        ds = dscore.contiguous() if dscore is not None else None
        dx, dmem = K.mem_read_bwd(xv, mem, pmem, dq, ds, want_dmem=ctx.needs_input_grad[1])
        return nchw(dx), dmem, None, None, None


class _MemWriteAccum(torch.autograd.Function):
    """memory.py:219-231 without the one-hot: flat [nominator (m+1,d) | denominator (m+1)]."""

    @staticmethod
    def forward(ctx, z, labels, m):
        zv = nhwc(z)
        labels = labels.contiguous()
        ctx.m = m
        ctx.save_for_backward(zv, labels)
        return K.mem_write_accum(zv, labels, m, normalize=True)

    @staticmethod
    def backward(ctx, dnomden):
        zv, labels = ctx.saved_tensors
        m, d = ctx.m, zv.shape[3]
        dnom = dnomden[:(m + 1) * d].contiguous()
        return nchw(K.mem_write_accum_bwd(zv, labels, m, dnom, normalize=True)), None, None


class _MemWriteUpdate(torch.autograd.Function):
    """memory.py:233-239: momentum update of the slots whose class occurs, then row-normalise."""

    @staticmethod
    def forward(ctx, mem, nomden, momentum):
        mem = mem.contiguous()
        out, u = K.mem_write_update(mem, nomden, momentum, want_u=True)
        ctx.momentum = momentum
        ctx.save_for_backward(u, nomden)
        return out

    @staticmethod
    def backward(ctx, dout):
        u, nomden = ctx.saved_tensors
        m, d = u.shape
        dnom = K.mem_write_update_bwd(u, nomden, ctx.momentum, dout.contiguous())
        g = torch.zeros_like(nomden)
        g[:(m + 1) * d] = dnom.reshape(-1)
        return None, g, None


# ---- functional front-ends (module holders in, logical-NCHW tensors out) ---------------------------------------------
def _geom(conv):
    assert conv.kernel_size[0] == conv.kernel_size[1] and conv.stride[0] == conv.stride[1] and conv.groups == 1
    assert conv.padding[0] == conv.padding[1] and conv.dilation[0] == conv.dilation[1]
    return conv.stride[0], conv.padding[0], conv.dilation[0]


def conv_bn_act(x, conv, bn, relu=True, residual=None, out=None):
    w, deferred = _w(conv.weight)
    return _ConvBnAct.apply(x, w, conv.bias, bn.weight, bn.bias, residual, _geom(conv), BNState(bn), relu, out, deferred)


def bottleneck(x, blk):
    """Whole Bottleneck as one autograd node (train mode with gradients); eval / no-grad keeps the fused-epilogue per-conv path."""
    ds = blk.downsample
    mods = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)] + ([(ds[0], ds[1])] if ds is not None else [])
    geoms = [_geom(c) for c, _ in mods]
    bns = [BNState(b) for _, b in mods]
    ws = [_w(c.weight) for c, _ in mods]
    wd, gd, bd = (ws[3][0], ds[1].weight, ds[1].bias) if ds is not None else (None, None, None)
    return _Bottleneck.apply(x, ws[0][0], blk.bn1.weight, blk.bn1.bias, ws[1][0], blk.bn2.weight, blk.bn2.bias,
                             ws[2][0], blk.bn3.weight, blk.bn3.bias, wd, gd, bd, geoms, bns, [d for _, d in ws] + [False])


def conv(x, conv_mod):
    return _Conv.apply(x, conv_mod.weight, conv_mod.bias, _geom(conv_mod))


def conv_raw(x, weight, bias, geom):
    return _Conv.apply(x, weight, bias, geom)


def maxpool3x3s2(x):
    return _MaxPool.apply(x)


def global_avgpool(x):
    return _GlobalAvgPool.apply(x)


def resize(x, size, out=None):
    return _Resize.apply(x, size, out)


def concat_buffer(like, channels, hw):
    """Logical-NCHW [B, sum(channels), h, w] buffer (NHWC memory) for branches to write into, of the dtype of `like` (bf16 on the bf16 tier: a channel
    total that is not a multiple of 64 -- the decoder's 48 + 256 -- comes with zero pad channels, K.new)."""
    return nchw(K.new((like.shape[0], hw[0], hw[1], sum(channels)), like, dtype=like.dtype if like.dtype == torch.bfloat16 else torch.float32))


def concat(buf, parts):
    return _Concat.apply(buf, *parts)


def add(a, b):
    return _Add.apply(a, b)


class _Cast(torch.autograd.Function):
    """fp32 <-> bf16 at the edges of the bf16 tier (the memory module and the losses stay fp32): one conversion kernel each way."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return nchw(K.cast(nhwc(x), dtype))

    @staticmethod
    def backward(ctx, g):
        return nchw(K.cast(_grad_view(g), ctx.src)), None


def cast(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


def upsample_ce(logits, labels, inv_temp=1.0):
    if not K.upsample_ce_fused_ok(nhwc(logits), tuple(labels.shape[1:])):
        # rows too wide for the fused kernels' LDS (> ~1000 low-res columns: no training crop of the reference reaches it): the same loss composed from the
        # bilinear kernel and torch's cross entropy on the materialised logits -- correct for any size, not tuned
        full = resize(logits, tuple(labels.shape[1:]))
        return torch.nn.functional.cross_entropy(full * float(inv_temp), labels, ignore_index=255, reduction='mean')
    return _UpsampleCE.apply(logits, labels, float(inv_temp), bool(torch.is_grad_enabled() and logits.requires_grad))


def mem_read(x, mem, noise=None):
    return _MemRead.apply(x, mem, noise, None, False)


def mem_read_pq(x, mem, noise=None, noise_q=None):
    """(qr, score, p_mem, p_query): the read with the softmax over all queries (memory.py:183-186) finished from the read kernel's column partials."""
    return _MemRead.apply(x, mem, noise, noise_q, True)


def mem_write_accum(z, labels, m):
    return _MemWriteAccum.apply(z, labels, m)


def mem_write_update(mem, nomden, momentum):
    return _MemWriteUpdate.apply(mem, nomden, float(momentum))
