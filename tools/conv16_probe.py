"""GPU probe: the bf16 tier's forward convolution on the shapes the LDS-DMA kernel carries (bs=8 768^2: the 48 x 48 maps and two 192 x 192 controls), TFLOP/s per shape.
Run once per library configuration (PM_C16_NST / PM_C16_BM / PM_C16_HYBRID / PM_CONV16 are read at load time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K
SHAPES = [  # name, n, cin, h, w, cout, k, pad, dil
    ('aspp 3x3 d12 2048->256 @48', 8, 2048, 48, 48, 256, 3, 12, 12),
    ('layer4.conv2 3x3 d2 512->512 @48', 8, 512, 48, 48, 512, 3, 2, 2),
    ('layer3.conv2 3x3 256->256 @48', 8, 256, 48, 48, 256, 3, 1, 1),
    ('layer4.conv1 1x1 2048->512 @48', 8, 2048, 48, 48, 512, 1, 0, 1),
    ('layer3.conv1 1x1 1024->256 @48', 8, 1024, 48, 48, 256, 1, 0, 1),
    ('layer3.conv3 1x1 256->1024 @48', 8, 256, 48, 48, 1024, 1, 0, 1),
    ('layer4.conv3 1x1 512->2048 @48', 8, 512, 48, 48, 2048, 1, 0, 1),
    ('aspp dgrad-like 3x3 256->2048 @48', 8, 256, 48, 48, 2048, 3, 1, 1),
    ('dsn-like 3x3 1024->512 @48', 8, 1024, 48, 48, 512, 3, 1, 1),
    ('dsn dgrad-like 3x3 512->1024 @48', 8, 512, 48, 48, 1024, 3, 1, 1),
    ('layer3 dgrad-like 1x1 1024->1024 @48', 8, 1024, 48, 48, 1024, 1, 0, 1),
    ('bot_aspp 1x1 1280->256 @48', 8, 1280, 48, 48, 256, 1, 0, 1),
    ('layer3.0.conv2-like 3x3 256->256 @96', 8, 256, 96, 96, 256, 3, 1, 1),
    ('layer2.conv3 1x1 128->512 @96', 8, 128, 96, 96, 512, 1, 0, 1),
    ('layer1.conv3 1x1 64->256 @192', 8, 64, 192, 192, 256, 1, 0, 1),
    ('final1.0 3x3 320->256 @192', 8, 320, 192, 192, 256, 3, 1, 1),
    ('layer2.conv2 3x3 128->128 @96', 8, 128, 96, 96, 128, 3, 1, 1),
    ('final1.3 3x3 256->256 @192', 8, 256, 192, 192, 256, 3, 1, 1),
    ('final1.0 dgrad-like 3x3 256->320 @192', 8, 256, 192, 192, 320, 3, 1, 1),
    ('aspp d6 3x3 2048->256 @48', 8, 2048, 48, 48, 256, 3, 6, 6),
    ('aspp d18 3x3 2048->256 @48', 8, 2048, 48, 48, 256, 3, 18, 18),
]
K.set_conv_precision('bf16')
if os.environ.get('PROBE_CONV16'):
    K.set_conv16(int(os.environ['PROBE_CONV16']))
tot = 0.0
for name, n, cin, h, w, cout, k, p, d in SHAPES:
    x = torch.randn(n, h, w, cin, device='cuda').bfloat16()
    wt = torch.randn(cout, k, k, cin, device='cuda') * 0.05
    fn = lambda: K.conv_fwd(x, wt, 1, p, d)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    fl = 2.0 * n * h * w * cout * cin * k * k
    tot += ms
    print('%-36s %7.1f GF %7.3f ms %7.1f TF' % (name, fl / 1e9, ms, fl / ms / 1e9), flush=True)
print('sum %.3f ms' % tot)
