"""GPU debug: accuracy of the train-mode parameter gradients (vs the fp64 CPU oracle) for the three conv routes, next to the fp32 CPU
oracle's own error. Prints the distribution of e_hip / e_oracle32 over the parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
import torch
import test_model_parity as T
from pinthememory_amd.hip import kernels as K
from oracle.ref_cpu import deeplab as o_deeplab, harness as o_harness
from pinthememory_amd import harness, synth
from pinthememory_amd.network import deepv2, deepv3plus
env = dict(o_deeplab=o_deeplab, o_harness=o_harness, harness=harness, synth=synth, deepv2=deepv2, deepv3plus=deepv3plus)
size = int(os.environ.get('SIZE', '128'))
x, y = synth.make_batch(2, size)
truth, o32 = T._oracle(env, torch.float64, x, y, False), T._oracle(env, torch.float32, x, y, False)
for mode in (0, 2, 4):
    K.set_winograd(mode)
    hip = T._hip(env, x, y, False)
    ratios, ehs = [], []
    for k, t in truth['grads'].items():
        if t.norm().item() < 1e-7:
            continue
        e_h, e_o = T._relerr(hip['grads'][k], t), T._relerr(o32['grads'][k], t)
        ratios.append(e_h / max(e_o, 1e-9)); ehs.append(e_h)
    r = np.array(ratios); e = np.array(ehs)
    print('route %d: e_hip/e_o32 median %.2f  p90 %.2f  max %.2f | e_hip median %.2e max %.2e | loss1 err %.2e' % (
        mode, np.median(r), np.percentile(r, 90), r.max(), np.median(e), e.max(), abs(hip['losses']['loss1'].item() - truth['losses']['loss1'].item())))
K.set_winograd(4)
