"""Builds libpinmem_hip.so (gfx950 only) in-tree with hipcc. No torch dependency in the library."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libpinmem_hip.so')
SOURCES = ['misc.hip', 'bf16.hip', 'conv_igemm.hip', 'winograd.hip', 'bn.hip', 'pool_resize.hip', 'loss.hip', 'memory.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-Wall', '-Wno-unused-function']


STAMP = LIB + '.stamp'


def _fingerprint():
    """sha256 over the flags and every source / header the library is built from. Content, not mtimes: a snapshot copied to another box
    (gpurun) carries the prebuilt .so with arbitrary timestamps and must not be rebuilt there, while an edited source always is."""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for path in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, '..', 'include', 'pinmem_hip.h')]:
        h.update(os.path.basename(path).encode())
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _fingerprint()


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    bdir = os.path.join(HERE, 'build')
    os.makedirs(bdir, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(bdir, s.replace('.hip', '.o'))
        objs.append(o)
        procs.append((s, subprocess.Popen([hipcc] + FLAGS + ['-c', os.path.join(CSRC, s), '-o', o])))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % s)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB])
    with open(STAMP, 'w') as f:
        f.write(_fingerprint())
    if verbose:
        print('built', LIB, file=sys.stderr)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
