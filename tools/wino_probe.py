"""GPU probe: Winograd F(2x2,3x3) route vs direct implicit GEMM on the wide 3x3 layers of the flagship step (bs=8, 768^2)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K

SHAPES = [  # name, n, cin, h, w, cout, dil
    ('final1.0 304->256 @192', 8, 304, 192, 192, 256, 1),
    ('final1.3 256->256 @192', 8, 256, 192, 192, 256, 1),
    ('layer3.conv2 256->256 @48', 8, 256, 48, 48, 256, 1),
    ('layer4.conv2 d2 512->512 @48', 8, 512, 48, 48, 512, 2),
    ('aspp d6 2048->256 @48', 8, 2048, 48, 48, 256, 6),
    ('aspp d12 2048->256 @48', 8, 2048, 48, 48, 256, 12),
    ('aspp d18 2048->256 @48', 8, 2048, 48, 48, 256, 18),
    ('dsn.0 1024->512 @48', 8, 1024, 48, 48, 512, 1),
    ('layer2.conv2 128->128 @96', 8, 128, 96, 96, 128, 1),
]

def bench(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

ONLY = os.environ.get('PROBE_ONLY')
rows = []
for name, n, cin, h, w, cout, d in SHAPES:
    if ONLY and ONLY not in name:
        continue
    x = torch.randn(n, h, w, cin, device='cuda')
    wt = torch.randn(cout, 3, 3, cin, device='cuda') * 0.05
    dy = torch.randn(n, h, w, cout, device='cuda')
    fl = 2.0 * n * h * w * cout * cin * 9
    r = dict(name=name, gflop=fl / 1e9)
    for wino in (0, 2, 4):
        K.set_winograd(wino)
        r['fwd_ms_%d' % wino] = bench(lambda: K.conv_fwd(x, wt, 1, d, d))
        r['dgrad_ms_%d' % wino] = bench(lambda: K.conv_bwd_data(dy, wt, tuple(x.shape), 1, d, d))
        r['wgrad_ms_%d' % wino] = bench(lambda: K.conv_bwd_weight(x, dy, tuple(wt.shape), 1, d, d))
    K.set_winograd(4)
    rows.append(r)
    print('%-30s %6.1f GF direct/F2/F4 ms: fwd %6.3f %6.3f %6.3f | dgrad %6.3f %6.3f %6.3f | wgrad %6.3f %6.3f %6.3f' % (
        name, fl / 1e9, r['fwd_ms_0'], r['fwd_ms_2'], r['fwd_ms_4'], r['dgrad_ms_0'], r['dgrad_ms_2'], r['dgrad_ms_4'],
        r['wgrad_ms_0'], r['wgrad_ms_2'], r['wgrad_ms_4']), flush=True)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(rows, open('gpurun_out/wino_probe.json', 'w'), indent=1)
