"""GPU: the fp32 HIP path's per-tensor gradient error of the `memory` stage vs the fp32 CPU oracle at bs=2 768^2 (the batch of test_bf16_tier_gradients_at_production_size_vs_fp32_oracle),
under the routing given in the environment (PM_PWSTREAM, PM_SPLIT, ...). Prints the worst tensors of every stage with their norms. The oracle result is cached in /tmp between runs."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import tests.test_model_parity as T
env = T.env.__wrapped__()
x, y = env['synth'].make_batch(2, 768)
cache = '/tmp/o32_768.pt'
if os.path.exists(cache):
    o32 = torch.load(cache)
else:
    torch.set_num_threads(24)
    o32 = T._oracle_uncached(env, torch.float32, x, y, True)
    torch.save({'grads': o32['grads']}, cache)
h32 = T._hip(env, x, y, True)
rows = []
for k, t in o32['grads'].items():
    if t.norm().item() < 1e-7:
        continue
    g = h32['grads'][k]
    rows.append(((g - t).norm().item() / t.norm().item(), k, t.norm().item(), t.numel()))
rows.sort(reverse=True)
print('routing', {k: v for k, v in os.environ.items() if k.startswith('PM_')})
for e, k, nrm, n in rows[:8]:
    print('%.3e  %-45s |g| %.3e  n %d' % (e, k, nrm, n))
