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
    assert lib.pm_version() >= 100
    assert lib.pm_last_error() is not None


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (error convention of the boundary)."""
    lib = L.load()
    assert lib.pm_mem_colsoftmax(None, None, 0, 19, None, None, 0, None) == -1
    assert b'mem_colsoftmax' in lib.pm_last_error()
    assert lib.pm_sgd_momentum(None, None, None, 10, 0.1, 0.9, 0.0, 1, None) == -1
