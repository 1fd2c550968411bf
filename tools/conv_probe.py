"""GPU probe: time the implicit-GEMM conv on the flagship shapes (bs=8, 768^2) and print TFLOP/s."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K

SHAPES = [  # name, n, cin, h, w, cout, k, stride, pad, dil
    ('final1.0 3x3 304->256 @192', 8, 304, 192, 192, 256, 3, 1, 1, 1),
    ('final1.3 3x3 256->256 @192', 8, 256, 192, 192, 256, 3, 1, 1, 1),
    ('layer4.conv2 3x3 d2 512->512 @48', 8, 512, 48, 48, 512, 3, 1, 2, 2),
    ('aspp 3x3 d12 2048->256 @48', 8, 2048, 48, 48, 256, 3, 1, 12, 12),
    ('layer4.conv3 1x1 512->2048 @48', 8, 512, 48, 48, 2048, 1, 1, 0, 1),
    ('layer3.conv2 3x3 256->256 @48', 8, 256, 48, 48, 256, 3, 1, 1, 1),
    ('layer1.conv2 3x3 64->64 @192', 8, 64, 192, 192, 64, 3, 1, 1, 1),
    ('layer1.conv3 1x1 64->256 @192', 8, 64, 192, 192, 256, 1, 1, 0, 1),
    ('layer2.conv2 3x3 128->128 @96', 8, 128, 96, 96, 128, 3, 1, 1, 1),
    ('layer2.conv3 1x1 128->512 @96', 8, 128, 96, 96, 512, 1, 1, 0, 1),
    ('layer1.conv1 1x1 256->64 @192', 8, 256, 192, 192, 64, 1, 1, 0, 1),
    ('layer3.conv1 1x1 1024->256 @48', 8, 1024, 48, 48, 256, 1, 1, 0, 1),
    ('layer3.conv3 1x1 256->1024 @48', 8, 256, 48, 48, 1024, 1, 1, 0, 1),
    ('stem 7x7 s2 4->64 @768', 8, 4, 768, 768, 64, 7, 2, 3, 1),
    ('final2 1x1 256->19 @192', 8, 256, 192, 192, 19, 1, 1, 0, 1),
]

def bench(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

import os
ONLY = os.environ.get('PROBE_ONLY')
rows = []
for name, n, cin, h, w, cout, k, s, p, d in SHAPES:
    if ONLY and ONLY not in name:
        continue
    x = torch.randn(n, h, w, cin, device='cuda')
    wt = torch.randn(cout, k, k, cin, device='cuda') * 0.05
    y = K.conv_fwd(x, wt, s, p, d)
    dy = torch.randn_like(y) if y.is_contiguous() else None
    if dy is None:
        dy = K.new(tuple(y.shape), y, pitch_pad=True); dy.normal_()
    fl = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * cout * cin * k * k
    t_f = bench(lambda: K.conv_fwd(x, wt, s, p, d))
    t_d = bench(lambda: K.conv_bwd_data(dy, wt, tuple(x.shape), s, p, d))
    t_w = bench(lambda: K.conv_bwd_weight(x, dy, tuple(wt.shape), s, p, d))
    rows.append(dict(name=name, gflop=fl / 1e9, fwd_ms=t_f, fwd_tf=fl / t_f / 1e9, dgrad_ms=t_d, dgrad_tf=fl / t_d / 1e9, wgrad_ms=t_w, wgrad_tf=fl / t_w / 1e9))
    print('%-36s %7.1f GF  fwd %6.3f ms %6.1f TF | dgrad %6.3f ms %6.1f TF | wgrad %6.3f ms %6.1f TF' % (
        name, fl / 1e9, t_f, fl / t_f / 1e9, t_d, fl / t_d / 1e9, t_w, fl / t_w / 1e9), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(rows, open('gpurun_out/conv_probe.json', 'w'), indent=1)
