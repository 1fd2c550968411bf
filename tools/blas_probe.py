"""GPU: the vendor's fp32 GEMM (torch.matmul -> hipBLASLt / rocBLAS, no TF32) on the flagship GEMM shapes, as a calibration of gemm_lab / the conv kernel."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
import sys
DT = torch.bfloat16 if "bf16" in sys.argv else torch.float32
shapes = [('wino 256->256 @48 (b36)', 36, 18432, 256, 256), ('wino 512->512 d2 (b36)', 36, 1152, 512, 512), ('wino 2048->256 (b36)', 36, 1152, 256, 2048),
          ('1x1 512->2048 @48', 1, 18432, 2048, 512), ('1x1 1024->256 @48', 1, 18432, 256, 1024), ('1x1 256->1024 @48', 1, 18432, 1024, 256),
          ('1x1 64->256 @192', 1, 294912, 256, 64), ('1x1 256->64 @192', 1, 294912, 64, 256), ('1x1 128->512 @96', 1, 73728, 512, 128)]
for name, b, M, N, K in shapes:
    A = torch.randn(b, M, K, device='cuda').to(DT); B = torch.randn(b, N, K, device='cuda').to(DT)
    C = torch.empty(b, M, N, device='cuda', dtype=DT)
    f = (lambda: torch.bmm(A, B.transpose(1, 2), out=C)) if b > 1 else (lambda: torch.matmul(A[0], B[0].t(), out=C[0]))
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print('%-26s %9.1f us  %7.1f TF' % (name, us, 2.0 * b * M * N * K / us * 1e-6), flush=True)
