"""GPU: ONE fp32 convolution shape, repeated (kernel durations out of a trace): forward, or with ONE_MODE=wgrad / dgrad in the environment the weight / data gradient.
usage: python tools/one_conv32.py n cin h w cout k pad dil [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K
n, cin, h, w, cout, k, p, d = [int(v) for v in sys.argv[1:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 10
K.set_conv_precision('f32')
x = torch.relu(torch.randn(n, h, w, cin, device='cuda'))
wt = torch.randn(cout, k, k, cin, device='cuda') * 0.05
mode = os.environ.get('ONE_MODE', 'fwd')
dy = torch.randn(n, h + 2 * p - d * (k - 1), w + 2 * p - d * (k - 1), cout, device='cuda')
ring = int(os.environ.get('ONE_RING', '0'))      # > 0: the forward writes into `ring` different outputs in turn (as in a training step: the output never sits in the Infinity Cache already)
outs, turn = [torch.empty_like(dy) for _ in range(ring)], [0]
def fwd_ring():
    turn[0] += 1
    return K.conv_fwd(x, wt, 1, p, d, out=outs[turn[0] % ring])
run = {'fwd': fwd_ring if ring else (lambda: K.conv_fwd(x, wt, 1, p, d)), 'wgrad': lambda: K.conv_bwd_weight(x, dy, tuple(wt.shape), 1, p, d)[0],
       'dgrad': lambda: K.conv_bwd_data(dy, wt, tuple(x.shape), 1, p, d)}[mode]
for _ in range(3):
    y = run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    y = run()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / reps
print('done', tuple(y.shape), '%.4f ms  %.1f TF' % (ms, 2.0 * n * h * w * cout * cin * k * k / ms / 1e9))
