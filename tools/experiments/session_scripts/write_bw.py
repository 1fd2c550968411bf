"""GPU: what a write-dominated stream reaches (ring of 4 outputs so that nothing sits in the Infinity Cache): fill, copy, and 1-read-4-writes (the byte pattern of the 64 -> 256 1x1 convolution at 192^2)."""
import torch
n = 294912 * 256
outs = [torch.empty(n, device='cuda') for _ in range(4)]
src = torch.randn(n, device='cuda')
small = torch.randn(294912, 64, device='cuda')
def timeit(f, reps=20):
    for i in range(4): f(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps): f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
t = timeit(lambda i: outs[i % 4].zero_())
print('fill   302 MB: %.1f us  %.2f TB/s' % (t * 1e3, n * 4 / t / 1e9))
t = timeit(lambda i: outs[i % 4].copy_(src))
print('copy   302+302 MB: %.1f us  %.2f TB/s' % (t * 1e3, 2 * n * 4 / t / 1e9))
o4 = [o.view(294912, 4, 64) for o in outs]
t = timeit(lambda i: o4[i % 4].copy_(small.unsqueeze(1).expand(294912, 4, 64)))
print('expand 75+302 MB: %.1f us  %.2f TB/s' % (t * 1e3, (n * 4 + small.numel() * 4) / t / 1e9))
t = timeit(lambda i: torch.add(src, 1.0, out=outs[i % 4]))
print('add    302+302 MB: %.1f us  %.2f TB/s' % (t * 1e3, 2 * n * 4 / t / 1e9))
