#!/usr/bin/env python3
"""A/B builds of the library: recompile ONE source with extra -D flags, link it with the in-tree objects of the others into ab/lib_<name>.so
(load it with PM_LIB=ab/lib_<name>.so; tools/gpu_lib_abn.sh alternates bench runs of several such builds on one box).
usage: build_variant.py <name> <source.hip> [-DFLAG[=v] ...]"""
import os
import subprocess
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinthememory_amd import build as B

name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()                                                    # in-tree objects up to date
root = os.path.dirname(B.HERE)
os.makedirs(os.path.join(root, 'ab'), exist_ok=True)
obj = os.path.join(root, 'ab', '%s_%s.o' % (name, src.replace('.hip', '')))
subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + extra + ['-c', os.path.join(B.CSRC, src), '-o', obj])
objs = [obj if s == src else os.path.join(B.HERE, 'build', s.replace('.hip', '.o')) for s in B.SOURCES]
out = os.path.join(root, 'ab', 'lib_%s.so' % name)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
print('built', out)
