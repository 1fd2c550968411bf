"""GPU debug: is the fp32-MFMA rate clock-limited under load? Samples rocm-smi (sclk / power) from a thread while (a) idle, (b) a back-to-back stream of the
flagship 1x1 GEMM (512 -> 2048 @48^2) runs, (c) the training step runs; prints the observed clock / power ranges next to the achieved TFLOP/s."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd.hip import kernels as K

samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            card = next(iter(d.values()))
            samples.append({k: v for k, v in card.items() if 'sclk' in k.lower() or 'power' in k.lower() or 'mclk' in k.lower() or 'fclk' in k.lower()})
        except Exception as e:      # noqa: BLE001
            samples.append({'error': repr(e)})
        time.sleep(0.2)


def phase(name, fn, seconds):
    global samples
    samples = []
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        fn()
        n += 1
        if n % 20 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print('%-28s %d calls in %.2f s' % (name, n, dt), flush=True)
    for s in samples[1:-1][:12]:
        print('     ', s, flush=True)
    return n / dt


th = threading.Thread(target=sampler, daemon=True)
th.start()
phase('idle', lambda: time.sleep(0.05), 2.0)
x = torch.randn(8, 48, 48, 512, device='cuda')
w = torch.randn(2048, 1, 1, 512, device='cuda') * 0.05
fl = 2.0 * 8 * 48 * 48 * 2048 * 512
r = phase('1x1 512->2048 @48^2 loop', lambda: K.conv_fwd(x, w, 1, 0, 1), 4.0)
print('   -> %.1f TFLOP/s' % (r * fl / 1e12))
x2 = torch.randn(8, 192, 192, 256, device='cuda')
y2 = torch.empty_like(x2)
r = phase('302 MB copy loop', lambda: K.copy(x2, y2), 3.0)
print('   -> %.2f TB/s' % (r * 2 * x2.numel() * 4 / 1e12))
stop = True
