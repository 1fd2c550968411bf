"""GPU: inter-kernel gap of back-to-back launches, eager (host queued ahead) vs hipGraph replay, on a ~20 us kernel and a ~5 us kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K
x = torch.relu(torch.randn(8, 48, 48, 256, device='cuda'))
mem = torch.nn.functional.normalize(torch.randn(19, 256, device='cuda'), dim=1)
big_a = torch.randn(8, 192, 192, 256, device='cuda'); big_b = torch.empty_like(big_a)
small_a = torch.randn(1, 16, 16, 256, device='cuda'); small_b = torch.empty_like(small_a)

def timed(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    for _ in range(40):
        K.copy(big_a, big_b)          # ~4.5 ms queued: the host is ahead for the whole timed region
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for name, one, n in (('mem_read_fwd (~19 us kernel)', lambda: K.mem_read_fwd(x, mem), 100), ('copy 256 KB (~3 us kernel)', lambda: K.copy(small_a, small_b), 100)):
    def eager():
        for _ in range(n):
            one()
    eager(); torch.cuda.synchronize()
    te = min(timed(eager, n) for _ in range(3))
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        eager()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        eager()
    g.replay(); torch.cuda.synchronize()
    tg = min(timed(g.replay, n) for _ in range(3))
    print('%-32s eager %.2f us/launch   graph %.2f us/launch' % (name, te, tg), flush=True)
