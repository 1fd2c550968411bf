"""ctypes binding of libpinmem_hip.so (include/pinmem_hip.h). The product path has NO CPU fallback:
if the library cannot be loaded, or a kernel returns an error, this raises."""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get('PM_LIB') or os.path.join(_HERE, 'libpinmem_hip.so')   # PM_LIB: A/B runs against another build of the library


class PmTensor(ctypes.Structure):
    _fields_ = [('ptr', c_void_p), ('n', c_int32), ('h', c_int32), ('w', c_int32), ('c', c_int32), ('pitch', c_int64), ('dtype', c_int32), ('flags', c_int32)]


PM_F32, PM_BF16 = 0, 1          # include/pinmem_hip.h pm_dtype
PM_TF_ZERO_PAD64 = 1            # pm_tensor.flags


class PmSgdEntry(ctypes.Structure):
    _fields_ = [('param', c_void_p), ('grad', c_void_p), ('momentum_buffer', c_void_p), ('numel', c_int64)]


class PmWxfJob(ctypes.Structure):
    _fields_ = [('w', c_void_p), ('wxf', c_void_p), ('wxf_bytes', c_int64), ('cout', c_int32), ('kh', c_int32), ('kw', c_int32), ('cin', c_int32),
                ('dgrad', c_int32), ('reserved', c_int32)]


class PmConvParams(ctypes.Structure):
    _fields_ = [('struct_size', c_int32), ('kh', c_int32), ('kw', c_int32), ('stride', c_int32), ('pad', c_int32), ('dil', c_int32), ('prec', c_int32),
                ('wino_v', c_void_p), ('wino_v_bytes', c_int64),
                ('wxf', c_void_p), ('wxf_bytes', c_int64), ('wxf_valid', c_int32)]


class PmRouting(ctypes.Structure):
    """include/pinmem_hip.h pm_routing: the library's one piece of mutable state -- which kernel takes a call."""
    _fields_ = [('struct_size', c_int32), ('winograd', c_int32), ('winograd_fused', c_int32), ('conv16', c_int32), ('conv16_wide', c_int32),
                ('conv16_persistent', c_int32), ('wgrad16', c_int32), ('bf16_wgrad', c_int32), ('split', c_int32)]


class PmConvEpilogue(ctypes.Structure):
    _fields_ = [('struct_size', c_int64), ('bias', c_void_p), ('scale', c_void_p), ('shift', c_void_p), ('residual', c_void_p),
                ('residual_pitch', c_int64), ('relu', c_int32), ('bn_partials', c_void_p), ('bn_partials_bytes', c_int64)]


def conv_params(kh, kw, stride, pad, dil, prec=0):
    return PmConvParams(ctypes.sizeof(PmConvParams), kh, kw, stride, pad, dil, prec)


def conv_epilogue(*fields):
    return PmConvEpilogue(ctypes.sizeof(PmConvEpilogue), *fields)


ABI_VERSION = 400          # include/pinmem_hip.h PM_ABI_VERSION this binding was written against


class PinmemError(RuntimeError):
    pass


_T, _P, _E = POINTER(PmTensor), POINTER(PmConvParams), POINTER(PmConvEpilogue)
_vp, _sz, _i, _f, _i64 = c_void_p, c_size_t, c_int, c_float, c_int64

# name -> (restype, argtypes); every symbol declared in include/pinmem_hip.h
SIGNATURES = {
    'pm_last_error': (c_char_p, []),
    'pm_version': (c_int, []),
    'pm_conv_winograd_v_bytes': (_sz, [_T, _T, _P]),
    'pm_conv_wxf_bytes': (_sz, [_T, _T, _P]),
    'pm_conv_wxf_bytes_dgrad': (_sz, [_T, _T, _P]),
    'pm_conv_wxf_refresh_bf16': (_i, [_vp, _i, _vp]),
    'pm_conv_wxf_refresh_f32': (_i, [_vp, _i, _vp]),
    'pm_conv_workspace': (_sz, [_T, _T, _P, _i]),
    'pm_conv_fwd': (_i, [_T, _vp, _T, _P, _E, _vp, _sz, _vp]),
    'pm_conv_bwd_data': (_i, [_T, _vp, _T, _P, _T, _vp, _sz, _vp]),
    'pm_conv_bwd_weight': (_i, [_T, _T, _vp, _vp, _P, _vp, _sz, _vp]),
    'pm_set_winograd': (_i, [_i]),
    'pm_set_winograd_fused': (_i, [_i]),
    'pm_set_conv16': (_i, [_i]),
    'pm_profile_enable': (_i, [_i]),
    'pm_profile_dump': (_i, [ctypes.c_char_p]),
    'pm_profile_read': (_i, [_i, _i, _i, _i, _i, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64), _i]),
    'pm_profile_read_prec': (_i, [_i, _i, _i, _i, _i, _i, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64), _i]),
    'pm_profile_read_bytes': (_i, [_i, _i, _i, _i, _i, _i, POINTER(ctypes.c_double)]),
    'pm_bn_workspace': (_sz, [_T]),
    'pm_bn_stats': (_i, [_T, _vp, _vp, _sz, _vp]),
    'pm_bn_merge': (_i, [_vp, _i, _i, _vp, _vp]),
    'pm_bn_merge_finalize': (_i, [_vp, _i, _i, _f, _vp, _vp, _vp, _vp, _f, _vp]),
    'pm_bn_finalize': (_i, [_vp, _i, _f, _vp, _vp, _vp, _vp, _f, _vp]),
    'pm_bn_stats_finalize': (_i, [_T, _f, _vp, _vp, _vp, _vp, _f, _vp, _sz, _vp]),
    'pm_bn_fold': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp]),
    'pm_bn_fold_multi': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    'pm_bn_apply': (_i, [_T, _vp, _vp, _vp, _vp, _T, _i, _T, _vp]),
    'pm_bn_apply_mask': (_i, [_T, _vp, _vp, _vp, _vp, _T, _i, _T, _vp, _vp]),
    'pm_bn_bwd_reduce_mask': (_i, [_T, _vp, _T, _vp, _vp, _T, _vp, _vp, _sz, _vp]),
    'pm_bn_bwd_reduce': (_i, [_T, _T, _T, _vp, _vp, _vp, _vp, _i, _T, _vp, _vp, _sz, _vp]),
    'pm_bn_bwd_apply': (_i, [_T, _T, _T, _vp, _vp, _vp, _vp, _vp, _f, _i, _T, _T, _vp]),
    'pm_relu_bwd': (_i, [_T, _T, _T, _vp]),
    'pm_add': (_i, [_T, _T, _T, _vp]),
    'pm_add_n': (_i, [POINTER(_T), _i, _T, _vp]),
    'pm_copy': (_i, [_T, _T, _vp]),
    'pm_scale_shift_act': (_i, [_T, _vp, _vp, _T, _i, _T, _vp]),
    'pm_maxpool3x3s2_fwd': (_i, [_T, _T, _vp, _vp]),
    'pm_maxpool3x3s2_bwd': (_i, [_T, _vp, _T, _vp]),
    'pm_global_avgpool_fwd': (_i, [_T, _T, _vp]),
    'pm_global_avgpool_bwd': (_i, [_T, _T, _i, _vp]),
    'pm_resize_bilinear_fwd': (_i, [_T, _T, _vp]),
    'pm_resize_bilinear_bwd': (_i, [_T, _T, _i, _vp]),
    'pm_resize_bilinear_bwd_workspace': (_sz, [_T, _T]),
    'pm_resize_bilinear_bwd_separable': (_i, [_T, _T, _i, _vp, _sz, _vp]),
    'pm_resize_bilinear_hp_fwd': (_i, [_T, _T, _i, _vp]),
    'pm_softmax_mean_update': (_i, [_T, _vp, _i, _vp]),
    'pm_sliding_stitch': (_i, [_T, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    'pm_argmax_f64': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'pm_image_u8_to_nhwc4': (_i, [_vp, _i64, POINTER(c_float), POINTER(c_float), _vp, _vp]),
    'pm_labels_u8_to_i64': (_i, [_vp, _i64, _vp, _vp]),
    'pm_nchw_to_nhwc': (_i, [_vp, _i, _T, _vp]),
    'pm_nhwc_to_nchw': (_i, [_T, _vp, _vp]),
    'pm_label_nearest': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp]),
    'pm_upsample_ce_workspace': (_sz, [_i, _i, _i]),
    'pm_upsample_ce_fwd': (_i, [_T, _f, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    'pm_upsample_ce_bwd_workspace': (_sz, [_T, _i, _i]),
    'pm_upsample_ce_bwd': (_i, [_T, _f, _vp, _i, _i, _vp, _vp, _T, _vp, _sz, _vp]),
    'pm_upsample_ce_field_bytes': (_sz, [_T, _i, _i]),
    'pm_upsample_ce_fwd_field': (_i, [_T, _f, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'pm_upsample_ce_bwd_field': (_i, [_T, _f, _i, _i, _vp, _vp, _vp, _T, _vp]),
    'pm_mem_read_fwd': (_i, [_T, _vp, _i, _vp, _T, _vp, _vp, _vp]),
    'pm_mem_read_fwd_pq_workspace': (_sz, [_i64, _i]),
    'pm_mem_read_fwd_pq': (_i, [_T, _vp, _i, _vp, _vp, _T, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pm_mem_colsoftmax_workspace': (_sz, [_i64, _i]),
    'pm_mem_colsoftmax': (_i, [_vp, _vp, _i64, _i, _vp, _vp, _sz, _vp]),
    'pm_mem_read_bwd_workspace': (_sz, [_i64, _i, _i]),
    'pm_mem_read_bwd': (_i, [_T, _vp, _i, _vp, _T, _vp, _T, _vp, _vp, _sz, _vp]),
    'pm_mem_write_accum_workspace': (_sz, [_T, _i]),
    'pm_mem_write_accum': (_i, [_T, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'pm_mem_write_accum_bwd': (_i, [_T, _vp, _i, _i, _i, _i, _vp, _T, _vp]),
    'pm_mem_write_update': (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    'pm_mem_write_update_bwd': (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    'pm_sgd_momentum': (_i, [_vp, _vp, _vp, _i64, _f, _f, _f, _i, _vp]),
    'pm_set_bf16_wgrad': (_i, [_i]),
    'pm_set_wgrad16': (_i, [_i]),
    'pm_set_split': (_i, [_i]),
    'pm_routing_get': (_i, [POINTER(PmRouting)]),
    'pm_routing_set': (_i, [POINTER(PmRouting)]),
    'pm_conv_bn_partials_bytes': (_sz, [_T, _T, POINTER(PmConvParams)]),
    'pm_bn_partials_finalize': (_i, [_vp, _i64, _i, _f, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    'pm_sgd_momentum_multi': (_i, [_vp, _i, _f, _f, _f, _vp]),
    'pm_sgd_momentum_multi_dev': (_i, [_vp, _i, _f, _vp, _f, _f, _vp]),
    'pm_cast': (_i, [_T, _T, _vp]),
}

_lib = None


def load():
    """dlopen the in-tree library (building it with hipcc when it is absent or stale) and bind every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get('PM_LIB'):
        from .. import build as _build
        _build.build()          # content-fingerprint gated: a no-op unless a source changed since the .so was built (never a stale binary)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the header and the library drift apart
        fn.restype, fn.argtypes = res, args
    if lib.pm_version() != ABI_VERSION:
        raise PinmemError('%s reports ABI %d, this binding is written against %d (include/pinmem_hip.h)' % (LIB_PATH, lib.pm_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code, what=''):
    if code != 0:
        raise PinmemError('%s failed (%d): %s' % (what, code, load().pm_last_error().decode()))


# torch.cuda.current_stream() walks device-index / availability helpers on every call (~4 us; 1 500 calls per step made the bf16 tier host-bound, cProfile in
# round 4). The raw handle of torch's current stream comes from two C calls instead.
_raw_stream = torch._C._cuda_getCurrentRawStream
_cur_device = torch._C._cuda_getDevice


def stream():
    return _raw_stream(_cur_device())


# bf16 tensors allocated with zero-filled pad channels up to a multiple of 64 (kernels.new / ops.concat_buffer): base pointer -> (weakref(base), c, pitch).
# A convolution then gathers such an input in place (PM_TF_ZERO_PAD64) instead of copying it to a padded buffer. A view keeps its base alive (`_base`),
# so a dead weak reference means the memory may belong to somebody else: the promise is withdrawn.
_ZERO_PAD = {}


def register_zero_pad(base, c, pitch):
    import weakref
    if len(_ZERO_PAD) > 64:
        for k in [k for k, v in _ZERO_PAD.items() if v[0]() is None]:
            del _ZERO_PAD[k]
    _ZERO_PAD[base.data_ptr()] = (weakref.ref(base), c, pitch)


_TD = {}      # (shape, stride, dtype) -> (n, h, w, c, pitch, dtype code): the layout checks of a view run once per distinct layout, not once per launch


def tdesc(t):
    """pm_tensor view of a torch tensor shaped [N,H,W,C] (channels innermost; may be a channel slice of a wider buffer); fp32 or bf16."""
    key = (t.shape, t.stride(), t.dtype)
    m = _TD.get(key)
    if m is None:
        assert t.dim() == 4 and t.dtype in (torch.float32, torch.bfloat16) and t.is_cuda, 'expected a CUDA fp32 / bf16 NHWC tensor, got %s %s' % (tuple(t.shape), t.dtype)
        n, h, w, c = t.shape
        sn, sh, sw, sc = t.stride()
        # size-1 dims carry arbitrary strides in torch: take the pixel pitch from the innermost spatial dim that is > 1
        pitch = sw if w > 1 else (sh if h > 1 else (sn if n > 1 else c))
        assert (sc == 1 or c == 1) and pitch >= c, 'channels must be innermost: shape %s stride %s' % (tuple(t.shape), t.stride())
        assert (h == 1 or w == 1 or sh == w * pitch) and (n == 1 or h * w == 1 or sn == h * w * pitch), \
            'not an NHWC view: shape %s stride %s' % (tuple(t.shape), t.stride())
        if len(_TD) > 4096:
            _TD.clear()
        m = _TD[key] = (n, h, w, c, pitch, PM_F32 if t.dtype == torch.float32 else PM_BF16)
    assert t.is_cuda
    if m[5] == PM_F32 or m[3] % 64 == 0:
        return PmTensor(t.data_ptr(), m[0], m[1], m[2], m[3], m[4], m[5], 0)
    flags = 0
    z = _ZERO_PAD.get(t.data_ptr())
    if z is not None and z[0]() is not None and z[1] == m[3] and z[2] == m[4]:
        flags = PM_TF_ZERO_PAD64
    return PmTensor(t.data_ptr(), m[0], m[1], m[2], m[3], m[4], m[5], flags)


_ws = {}


def workspace(nbytes, device, on_stream=None):
    """Grow-only scratch buffer per (device, stream); kernels on one stream are ordered, so reuse is safe. on_stream: the torch stream the launch goes to when
    that is not the current one (the weight-gradient side stream): the buffer comes from the current stream's pool, so the allocator is told who really uses it."""
    key = (device.index, stream() if on_stream is None else on_stream.cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        if on_stream is not None:
            buf.record_stream(on_stream)
        _ws[key] = buf
    return buf


_stream_objs = {}


def stream_obj():
    """torch.cuda.current_stream() without its per-call device bookkeeping: the Stream object is cached per raw handle."""
    raw = stream()
    s = _stream_objs.get(raw)
    if s is None:
        s = _stream_objs[raw] = torch.cuda.current_stream()
    return s


def ptr(t):
    return None if t is None else t.data_ptr()
