"""Tiny conv workload for PMC collection: a few launches of 3 representative shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K
SHAPES = [(8, 256, 192, 192, 256, 3, 1, 1, 1), (8, 512, 48, 48, 512, 3, 1, 2, 2), (8, 64, 192, 192, 256, 1, 1, 0, 1)]
for n, cin, h, w, cout, k, s, p, d in SHAPES:
    x = torch.randn(n, h, w, cin, device='cuda'); wt = torch.randn(cout, k, k, cin, device='cuda') * 0.05
    for _ in range(3):
        y = K.conv_fwd(x, wt, s, p, d)
    dy = torch.randn_like(y)
    for _ in range(2):
        K.conv_bwd_data(dy, wt, tuple(x.shape), s, p, d); K.conv_bwd_weight(x, dy, tuple(wt.shape), s, p, d)
torch.cuda.synchronize()
