"""Builds libpinmem_hip.so (gfx950 only) in-tree with hipcc. No torch dependency in the library."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libpinmem_hip.so')
SOURCES = ['misc.hip', 'act16.hip', 'bf16.hip', 'conv_igemm.hip', 'conv_split.hip', 'pwstream.hip', 'conv16.hip', 'conv16w.hip', 'wgrad16.hip', 'winograd.hip', 'bn.hip', 'pool_resize.hip', 'loss.hip', 'memory.hip']
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
    """Build if the sources changed since the library was built. Safe to call from every rank of a multi-process launch at once: one
    process builds under an exclusive file lock (objects and the .so go to temporary names and are renamed into place, so a concurrent
    dlopen never sees a half-written file), the others block on the lock, re-check the fingerprint and find the work done."""
    if not force and not _stale():
        return LIB
    import fcntl
    with open(LIB + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():      # another process built it while this one waited
                return LIB
            _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


def _build_locked(verbose):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    fp = _fingerprint()
    objs = []
    procs = []
    bdir = os.path.join(HERE, 'build')
    os.makedirs(bdir, exist_ok=True)
    tag = '.%d.tmp' % os.getpid()
    for s in SOURCES:
        o = os.path.join(bdir, s.replace('.hip', '.o'))
        objs.append(o)
        procs.append((s, o, subprocess.Popen([hipcc] + FLAGS + ['-c', os.path.join(CSRC, s), '-o', o + tag])))
    failed = [s for s, o, p in procs if p.wait() != 0]
    if failed:
        for s, o, p in procs:
            if os.path.exists(o + tag):
                os.remove(o + tag)
        raise RuntimeError('hipcc failed on %s' % ', '.join(failed))
    for s, o, p in procs:
        os.replace(o + tag, o)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB + tag])
    os.replace(LIB + tag, LIB)
    with open(STAMP + tag, 'w') as f:
        f.write(fp)
    os.replace(STAMP + tag, STAMP)
    if verbose:
        print('built', LIB, file=sys.stderr)


if __name__ == '__main__':
    build(force='--force' in sys.argv)
