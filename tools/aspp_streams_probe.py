"""GPU probe: the three dilated ASPP branches (2048 -> 256, d6 / d12 / d18 @48x48, Winograd GEMMs of 1152-2592 tile rows: 1.27 rounds of blocks each) one after
the other on one stream vs side by side on three streams -- does filling each other's tile-quantisation tails pay?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K

x = torch.randn(8, 48, 48, 2048, device='cuda')
ws = [torch.randn(256, 3, 3, 2048, device='cuda') * 0.02 for _ in range(3)]
w1 = torch.randn(256, 1, 1, 2048, device='cuda') * 0.02
buf = torch.empty(8, 48, 48, 1280, device='cuda')
dil = (6, 12, 18)
streams = [torch.cuda.Stream() for _ in range(3)]


def serial():
    K.conv_fwd(x, w1, 1, 0, 1, out=buf[..., 256:512])
    for i, d in enumerate(dil):
        K.conv_fwd(x, ws[i], 1, d, d, out=buf[..., 512 + 256 * i:768 + 256 * i])


def parallel():
    cur = torch.cuda.current_stream()
    ev = cur.record_event()
    K.conv_fwd(x, w1, 1, 0, 1, out=buf[..., 256:512])
    for i, d in enumerate(dil):
        with torch.cuda.stream(streams[i]):
            streams[i].wait_event(ev)
            K.conv_fwd(x, ws[i], 1, d, d, out=buf[..., 512 + 256 * i:768 + 256 * i])
    for s in streams:
        cur.wait_stream(s)


def bench(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for rep in range(3):
    print('serial %.3f ms   three streams %.3f ms' % (bench(serial), bench(parallel)), flush=True)
