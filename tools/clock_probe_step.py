"""GPU: shader clock / socket power while the agg train step runs (rocm-smi sampled from a thread), per dtype: is a slow reading of the step a clock / power event?
usage: clock_probe_step.py <f32|bf16> [seconds]   -- prints ms/step per block of 25 steps with the sclk / power samples seen during that block."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import harness, synth
from pinthememory_amd.hip import kernels as K
from pinthememory_amd.network import deepv3plus

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
K.set_conv_precision(dtype)
samples, stop = [], False


def sampler():
    while not stop:
        try:
            d = json.loads(subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--json'], capture_output=True, text=True, timeout=10).stdout)
            card = next(iter(d.values()))
            samples.append((time.time(), {k: v for k, v in card.items() if any(s in k.lower() for s in ('sclk', 'power', 'junction'))}))
        except Exception as e:      # noqa: BLE001
            samples.append((time.time(), {'error': repr(e)}))
        time.sleep(0.25)


crit = torch.nn.CrossEntropyLoss(reduction='mean', ignore_index=255)
net = synth.load_det_weights(deepv3plus.DeepR50V3PlusD(synth.model_args(gumbel_off=False), 19, crit, crit)).cuda()
opt, sched = harness.make_optimizer(net)
x, y = synth.make_batch(8, 768)
x, y = x.cuda(), y.cuda()
for _ in range(3):
    harness.agg_train_step(net, opt, x, y, sched=sched)
torch.cuda.synchronize()
threading.Thread(target=sampler, daemon=True).start()
time.sleep(1.0)
print('idle:', [s[1] for s in samples][-2:], flush=True)
t_end = time.time() + seconds
while time.time() < t_end:
    t0 = time.time()
    for _ in range(25):
        harness.agg_train_step(net, opt, x, y, sched=sched)
    torch.cuda.synchronize()
    t1 = time.time()
    seen = [s[1] for s in samples if t0 <= s[0] <= t1]
    sclk = sorted({v for s in seen for k, v in s.items() if 'sclk' in k.lower()})
    power = sorted({v for s in seen for k, v in s.items() if 'power' in k.lower()})
    print('%s: %.2f ms/step | sclk %s | power %s' % (dtype, (t1 - t0) / 25 * 1e3, sclk[:1] + sclk[-1:], power[:1] + power[-1:]), flush=True)
stop = True
