"""GPU probe: does the Winograd pipeline of the big layers run faster when its intermediates (V = Bt x B, M = V U^T) stay in the 256 MB
Infinity Cache? Same convolution as ONE call over the batch (V + M of final1: 0.85 + 0.68 GB, streamed through HBM three times) vs one
call per chunk of images on the same workspace addresses (per image 106 + 85 MB). No new kernel code: the per-chunk calls go through the C ABI."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K

SHAPES = [  # name, n, cin, h, w, cout, dil
    ('final1.0 304->256 @192', 8, 304, 192, 192, 256, 1),
    ('final1.3 256->256 @192', 8, 256, 192, 192, 256, 1),
    ('layer2.conv2 128->128 @96', 8, 128, 96, 96, 128, 1),
    ('aspp d6 2048->256 @48', 8, 2048, 48, 48, 256, 6),
    ('layer4.conv2 d2 512->512 @48', 8, 512, 48, 48, 512, 2),
]


def bench(fn, iters=6):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for name, n, cin, h, w, cout, d in SHAPES:
    x = torch.randn(n, h, w, cin, device='cuda')
    wt = torch.randn(cout, 3, 3, cin, device='cuda') * 0.05
    dy = torch.randn(n, h, w, cout, device='cuda')
    y = torch.empty(n, h, w, cout, device='cuda')
    dx = torch.empty(n, h, w, cin, device='cuda')
    line = '%-30s' % name
    for chunk in (8, 4, 2, 1):
        def fwd():
            for i in range(0, n, chunk):
                K.conv_fwd(x[i:i + chunk], wt, 1, d, d, out=y[i:i + chunk])

        def dgrad():
            for i in range(0, n, chunk):
                K.conv_bwd_data(dy[i:i + chunk], wt, (chunk, h, w, cin), 1, d, d)
        line += ' | chunk %d: fwd %6.3f dgrad %6.3f' % (chunk, bench(fwd), bench(dgrad))
    print(line, flush=True)
