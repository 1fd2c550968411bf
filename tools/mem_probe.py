"""GPU: the memory read / write / read-loss kernels at the flagship size (bs=8, 48x48 features, 768x768 masks).
Prints achieved algorithmic GB/s (HIP-event timing); run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for HBM traffic.
A pm_copy of a known size calibrates the counters (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide reads 2x on gfx950)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pinthememory_amd import synth
from pinthememory_amd.hip import kernels as K

def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

B, h, d, m, H = 8, 48, 256, 19, 768
x = torch.relu(torch.randn(B, h, h, d, device='cuda'))
mem = torch.nn.functional.normalize(torch.randn(m, d, device='cuda'), dim=1)
_, lab = synth.make_batch(B, H); lab = lab.cuda()
qr, score, pmem = K.mem_read_fwd(x, mem)
dqr = torch.randn_like(qr); dsx = torch.randn_like(score)
cal_a = torch.randn(B, 192, 192, 256, device='cuda'); cal_b = torch.empty_like(cal_a)      # 302 MB each way
N = B * h * h
out = {}
def rec(name, nbytes, fn, iters=20):
    t = bench(fn, iters); out[name] = dict(ms=t, algorithmic_MB=nbytes / 1e6, GBps=nbytes / t / 1e6)
    print('%-22s %8.3f ms  %8.1f MB algorithmic  %8.1f GB/s' % (name, t, nbytes / 1e6, nbytes / t / 1e6), flush=True)
rec('copy_calibration', 2 * cal_a.numel() * 4, lambda: K.copy(cal_a, cal_b))
cal_c = torch.randn(N * 3 * d // 2, device='cuda'); cal_d = torch.empty_like(cal_c)                      # 28.3 MB each way = the read's 56.6 MB
rec('copy_small_calibration', 2 * cal_c.numel() * 4, lambda: K.copy(cal_c.view(1, 1, -1, d), cal_d.view(1, 1, -1, d)))
rec('fill_calibration', cal_b.numel() * 4, lambda: cal_b.zero_())
rec('mem_read_fwd', N * d * 4 + m * d * 4 + N * 2 * d * 4 + 2 * N * m * 4, lambda: K.mem_read_fwd(x, mem))            # 59.4 MB
rec('mem_read_bwd', N * 2 * d * 4 + N * d * 4 + N * d * 4 + 2 * N * m * 4, lambda: K.mem_read_bwd(x, mem, pmem, dqr, dsx))   # 75.5 MB + scores
rec('mem_read_fwd_pq', N * d * 4 + m * d * 4 + N * 2 * d * 4 + 3 * N * m * 4, lambda: K.mem_read_fwd_pq(x, mem))    # + p_query from the read kernel's column partials
rec('mem_colsoftmax', 2 * N * m * 4, lambda: K.mem_colsoftmax(score))
rec('mem_write_accum', N * d * 4 + B * 4 * h * h * 8 + 20 * 257 * 4, lambda: K.mem_write_accum(x, lab, m))               # 19.5 MB
lg = score.view(B, h, h, m)
rec('readloss_fwd', B * H * H * 8 + N * m * 4, lambda: K.upsample_ce_fwd(lg, lab, 1.0))                                    # 39.2 MB
lo = K.upsample_ce_fwd(lg, lab, 1.0)
rec('readloss_bwd', B * H * H * 8 + 2 * N * m * 4, lambda: K.upsample_ce_bwd(lg, lab, lo, None, 1.0), iters=5)
lo_f, lo_field = K.upsample_ce_fwd_field(lg, lab, 1.0)
rec('readloss_fwd_field', B * H * H * 8 + N * m * 4 + lo_field.numel() * 4, lambda: K.upsample_ce_fwd_field(lg, lab, 1.0))      # training forward: loss + gradient field
rec('readloss_bwd_field', lo_field.numel() * 4 + N * m * 4, lambda: K.upsample_ce_bwd_field(lg, (H, H), lo_f, lo_field, None, 1.0))
main = K.new((B, H // 4, H // 4, m), x, pitch_pad=True); main.copy_(torch.randn(B, H // 4, H // 4, m, device='cuda'))
mo_f, mo_field = K.upsample_ce_fwd_field(main, lab, 1.0)
rec('main_ce_fwd_field', B * H * H * 8 + B * (H // 4) ** 2 * m * 4 + mo_field.numel() * 4, lambda: K.upsample_ce_fwd_field(main, lab, 1.0))
rec('main_ce_bwd_field', mo_field.numel() * 4 + B * (H // 4) ** 2 * m * 4, lambda: K.upsample_ce_bwd_field(main, (H, H), mo_f, mo_field, None, 1.0))
os.makedirs('gpurun_out', exist_ok=True)
json.dump(out, open('gpurun_out/mem_probe.json', 'w'), indent=1)
