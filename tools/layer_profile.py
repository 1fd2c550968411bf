"""GPU: time every conv launch of one real agg train step (bs=8, 768^2) by shape class; prints ms and TFLOP/s sorted by cost."""
import sys, os, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus

recs = []
def wrap(name, fn, shape_of):
    def f(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = fn(*a, **k); e.record()
        recs.append((name, shape_of(*a, **k), s, e))
        return r
    return f
def fl(n, ho, wo, cout, kh, kw, cin): return 2.0 * n * ho * wo * cout * kh * kw * cin
def sh_f(x, w, s, p, d, **k):
    n, h, wd, c = x.shape; co, kh, kw, ci = w.shape; ho, wo = K.conv_out_hw(h, wd, kh, s, p, d)
    return ('%dx%d s%d d%d %d->%d @%d' % (kh, kw, s, d, ci, co, ho), fl(n, ho, wo, co, kh, kw, ci))
def sh_d(dy, w, xs, s, p, d, **k):
    n, ho, wo, co = dy.shape; _, kh, kw, ci = w.shape
    return ('%dx%d s%d d%d %d->%d @%d' % (kh, kw, s, d, ci, co, ho), fl(n, ho, wo, co, kh, kw, ci))
def sh_w(x, dy, ws, s, p, d, **k):
    n, ho, wo, co = dy.shape; _, kh, kw, ci = ws
    return ('%dx%d s%d d%d %d->%d @%d' % (kh, kw, s, d, ci, co, ho), fl(n, ho, wo, co, kh, kw, ci))
K.conv_fwd = wrap('fwd', K.conv_fwd, sh_f); K.conv_bwd_data = wrap('dgrad', K.conv_bwd_data, sh_d); K.conv_bwd_weight = wrap('wgrad', K.conv_bwd_weight, sh_w)

crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(), 19, crit, crit)).cuda()
opt, _ = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768); x, y = x.cuda(), y.cuda()
harness.agg_train_step(net, opt, x, y); torch.cuda.synchronize(); recs.clear()
harness.agg_train_step(net, opt, x, y); torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, (shape, flops), s, e in recs:
    k = (name, shape); a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += s.elapsed_time(e); a[2] += flops
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print('total conv ms (incl. split-K reduce) %.2f  TF/s %.1f' % (tot, sum(v[2] for _, v in rows) / tot / 1e9))
for (name, shape), (n, ms, flops) in rows:
    print('%-6s %-28s n=%2d %8.3f ms %5.1f%%  %6.1f TF/s' % (name, shape, n, ms, 100 * ms / tot, flops / ms / 1e9))
by = collections.defaultdict(lambda: [0.0, 0.0])
for (name, shape), (n, ms, flops) in rows: by[name][0] += ms; by[name][1] += flops
for k, (ms, f) in by.items(): print('%-6s %8.2f ms %6.1f TF/s' % (k, ms, f / ms / 1e9))
