"""CPU: the C-ABI shared library loads and exports exactly the entry points include/pinmem_hip.h declares
(no compute calls here -- there is no GPU in the build container)."""
import os
import re

from pinthememory_amd import build
from pinthememory_amd.hip import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, 'include', 'pinmem_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(pm_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build()
    lib = L.load()
    names = declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), 'missing export ' + n
    assert sorted(L.SIGNATURES) == names, 'ctypes table and header drifted apart'
    hdr = int(re.search(r'#define PM_ABI_VERSION (\d+)', open(os.path.join(ROOT, 'include', 'pinmem_hip.h')).read()).group(1))
    assert lib.pm_version() == hdr == L.ABI_VERSION          # library, header and ctypes binding agree on the struct layouts
    assert lib.pm_last_error() is not None


def test_struct_size_guards_reject_a_caller_built_against_another_header():
    """pm_conv_params / pm_conv_epilogue grew in rounds 2 and 3; their first member is their own size, and an entry point that receives another size
    (a consumer compiled against an older header) refuses the call instead of reading past the caller's object."""
    import ctypes
    from ctypes import byref
    lib = L.load()
    assert ctypes.sizeof(L.PmConvParams) == L.conv_params(3, 3, 1, 1, 1).struct_size
    buf = (ctypes.c_float * 64)()
    x = L.PmTensor(ctypes.addressof(buf), 1, 2, 2, 4, 4)
    y = L.PmTensor(ctypes.addressof(buf), 1, 2, 2, 4, 4)
    p = L.conv_params(1, 1, 1, 0, 1)
    p.struct_size -= 8                                     # "older header"
    assert lib.pm_conv_fwd(byref(x), ctypes.addressof(buf), byref(y), byref(p), None, None, 0, None) == -1
    assert b'struct_size' in lib.pm_last_error()
    p = L.conv_params(1, 1, 1, 0, 1)
    ep = L.conv_epilogue(None, None, None, None, 0, 0, None, 0)
    ep.struct_size = 48
    assert lib.pm_conv_fwd(byref(x), ctypes.addressof(buf), byref(y), byref(p), byref(ep), None, 0, None) == -1
    assert b'pm_conv_epilogue.struct_size' in lib.pm_last_error()


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (error convention of the boundary)."""
    lib = L.load()
    assert lib.pm_mem_colsoftmax(None, None, 0, 19, None, None, 0, None) == -1
    assert b'mem_colsoftmax' in lib.pm_last_error()
    assert lib.pm_sgd_momentum(None, None, None, 10, 0.1, 0.9, 0.0, 1, None) == -1


def test_routing_is_one_struct_with_single_field_wrappers():
    """Round 6: the library's only mutable global state is ONE pm_routing value -- struct_size-guarded, read and replaced as a whole; the pm_set_*
    entry points change one field of it. No compute: which kernel WOULD take a call."""
    import ctypes
    from ctypes import byref
    lib = L.load()
    r = L.PmRouting()
    r.struct_size = ctypes.sizeof(L.PmRouting)
    assert lib.pm_routing_get(byref(r)) == 0
    before = {k: getattr(r, k) for k, _ in L.PmRouting._fields_}
    assert before['winograd'] in (0, 2, 4) and before['split'] in (0, 1) and before['conv16'] in (0, 1, 2)
    bad = L.PmRouting()
    bad.struct_size = ctypes.sizeof(L.PmRouting) - 4                 # "another header"
    assert lib.pm_routing_get(byref(bad)) == -1 and b'struct_size' in lib.pm_last_error()
    assert lib.pm_routing_set(byref(bad)) == -1
    try:
        assert lib.pm_set_split(0) == 0 and lib.pm_set_winograd(2) == 0 and lib.pm_set_conv16(8) == 0
        assert lib.pm_routing_get(byref(r)) == 0
        assert (r.split, r.winograd, r.conv16, r.conv16_wide, r.conv16_persistent) == (0, 2, 2, 2, 0)
        assert lib.pm_set_conv16(5) == 0 and lib.pm_routing_get(byref(r)) == 0 and r.conv16 == 1      # the streaming 1x1 kernel left the library: 5 / 6 mean 1
        r.winograd = 3
        assert lib.pm_routing_set(byref(r)) == -1 and b'winograd' in lib.pm_last_error()              # validated as a whole, nothing applied
        assert lib.pm_routing_get(byref(r)) == 0 and r.winograd == 2
    finally:
        for k, v in before.items():
            setattr(r, k, v)
        assert lib.pm_routing_set(byref(r)) == 0
    assert lib.pm_routing_get(byref(r)) == 0 and {k: getattr(r, k) for k, _ in L.PmRouting._fields_} == before
