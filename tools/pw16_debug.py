"""GPU debug: error map of the streaming 1x1 kernel (pw16.hip) by (32-pixel block, 8-channel group)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pinthememory_amd.hip import kernels as K
n, cin, h, w, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 64, 16, 16, 256))]
g = torch.Generator().manual_seed(1)
x = torch.randn(n, cin, h, w, generator=g).bfloat16().float()
wt = (torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5)
ref = F.conv2d(x, wt.bfloat16().float())
K.set_conv_precision('bf16')
K.set_conv16(5)
wg = wt.permute(0, 2, 3, 1).contiguous().cuda()
xg = x.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
y = K.conv_fwd(xg, wg, 1, 0, 1)
torch.cuda.synchronize()
got = y.float().cpu().permute(0, 3, 1, 2)
err = (got - ref).abs().reshape(n, cout, h * w)[0]          # [cout, pixels] of image 0
bad = err > 0.05 * ref.abs().max()
print('bad fraction', bad.float().mean().item())
cg = bad.reshape(cout // 8, 8, -1).float().mean(1)          # [channel groups of 8, pixels]
pb = cg.reshape(cout // 8, -1, 32).mean(2)                  # [channel groups, 32-pixel blocks]
print('rows = 8-channel groups, columns = 32-pixel blocks (fraction bad):')
for i, r in enumerate(pb[:40]):
    print('cg %2d (ch %3d..): ' % (i, i * 8) + ' '.join('%.1f' % v for v in r[:16]))
# is the wrong data a permutation of channels? correlate got channel c with ref channel c'
gc, rc = got[0].reshape(cout, -1), ref[0].reshape(cout, -1)
gc = (gc - gc.mean(1, keepdim=True)) / (gc.std(1, keepdim=True) + 1e-9)
rc = (rc - rc.mean(1, keepdim=True)) / (rc.std(1, keepdim=True) + 1e-9)
corr = gc @ rc.t() / gc.shape[1]
best = corr.argmax(1)
print('output channel -> best matching reference channel (first 64):', best[:64].tolist())
print('match quality (first 16):', [round(v, 2) for v in corr.max(1).values[:16].tolist()])
