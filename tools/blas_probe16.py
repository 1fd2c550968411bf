"""GPU, calibration only (never on the product path): the vendor's bf16 GEMM (torch.matmul -> hipBLASLt, bf16 in / bf16 out, fp32 accumulate) on the 20 heaviest GEMM shapes of the
bf16 tier's step, as the per-shape 'reachable ceiling' next to this build's kernels (VERDICT r4 missing 4 / next 1d). A 3x3 convolution appears as its implicit GEMM
(M = output pixels, N = Cout, K = 9 Cin: the vendor multiplies a materialised [M, K] matrix, i.e. it gets the im2col for free); weight gradients as the TN product
dW[N = Cout, 9 Cin] = dy[pixels, Cout]^T x[pixels, 9 Cin] (K = pixels). Columns: vendor us / TF / fraction of 2.5 PF, and the HBM floor of the shape (bytes / 6.3 TB/s).
usage: python tools/blas_probe16.py [out.txt]"""
import sys
import torch

# (name, kind, M, N, K, launches per step) -- from profiles/r04c_conv_shapes_bf16.txt (its K column counts 4-byte units: doubled here)
SHAPES = [
    ('ASPP 3x3 2048->256 @48 fwd', 'nt', 18432, 256, 18432, 6),
    ('decoder 3x3 256->256 @192 fwd/dgrad', 'nt', 294912, 256, 2304, 3),
    ('layer4 3x3 512->512 @48 fwd/dgrad', 'nt', 18432, 512, 4608, 9),
    ('decoder 3x3 304->256 @192 fwd', 'nt', 294912, 256, 2880, 2),
    ('ASPP 3x3 wgrad 256 x 18432, K=18432', 'tn', 256, 18432, 18432, 3),
    ('layer3 3x3 256->256 @48 fwd/dgrad', 'nt', 18432, 256, 2304, 17),
    ('decoder wgrad 256 x 2736, K=294912', 'tn', 256, 2736, 294912, 1),
    ('layer1 1x1 64->256 @192', 'nt', 294912, 256, 64, 11),
    ('layer4 1x1 512->2048 @48', 'nt', 18432, 2048, 512, 8),
    ('layer3 1x1 256->1024 @48', 'nt', 18432, 1024, 256, 17),
    ('ASPP dgrad 3x3 256->2048 @48', 'nt', 18432, 2048, 2304, 3),
    ('decoder dgrad 3x3 256->304 @192', 'nt', 294912, 304, 2304, 1),
    ('layer4 1x1 2048->512 @48', 'nt', 18432, 512, 2048, 7),
    ('decoder wgrad 256 x 2304, K=294912', 'tn', 256, 2304, 294912, 1),
    ('layer2 3x3 128->128 @96', 'nt', 73728, 128, 1152, 11),
    ('layer2 1x1 128->512 @96', 'nt', 73728, 512, 128, 11),
    ('layer3 1x1 1024->256 @48', 'nt', 18432, 256, 1024, 17),
    ('layer1 3x3 64->64 @192', 'nt', 294912, 64, 576, 9),
    ('layer4 wgrad 512 x 4608, K=18432', 'tn', 512, 4608, 18432, 3),
    ('layer2 1x1 512->128 @96', 'nt', 73728, 128, 512, 11),
    ('layer1 1x1 256->64 @192', 'nt', 294912, 64, 256, 8),
    ('layer3 wgrad 256 x 2304, K=18432', 'tn', 256, 2304, 18432, 6),
    ('layer4 wgrad 2048 x 512, K=18432', 'tn', 2048, 512, 18432, 3),
]


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else None

    def emit(line):
        print(line, flush=True)
        if out:
            out.write(line + '\n')
    emit('# python tools/blas_probe16.py: torch.matmul (hipBLASLt) bf16 x bf16 -> bf16, fp32 accumulation, uniform random [-1, 1) operands, 20 back-to-back launches per event pair')
    emit('# torch %s, %s' % (torch.__version__, torch.cuda.get_device_name(0)))
    emit('%-40s %4s %8s %6s %8s | %9s %8s %7s | %9s %s' % ('shape', 'kind', 'M', 'N', 'K', 'vendor us', 'TF', 'of peak', 'HBM floor', 'n/step'))
    g = torch.Generator(device='cuda').manual_seed(5)
    tot_us = 0.0
    for name, kind, M, N, K, n in SHAPES:
        if kind == 'nt':
            A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
            B = (torch.rand(N, K, device='cuda', generator=g) * 2 - 1).bfloat16()
            C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
            f = lambda: torch.matmul(A, B.t(), out=C)
        else:
            A = (torch.rand(K, M, device='cuda', generator=g) * 2 - 1).bfloat16()      # dy [pixels, Cout]
            B = (torch.rand(K, N, device='cuda', generator=g) * 2 - 1).bfloat16()      # x  [pixels, 9 Cin]
            C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
            f = lambda: torch.matmul(A.t(), B, out=C)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        tf = 2.0 * M * N * K / us * 1e-6
        floor_us = (M * K + N * K + M * N) * 2 / 6.3e12 * 1e6
        tot_us += us * n
        emit('%-40s %4s %8d %6d %8d | %9.1f %8.1f %7.3f | %7.1f us %d' % (name, kind, M, N, K, us, tf, tf / 2500.0, floor_us, n))
        del A, B, C
    emit('# sum over the listed launches of one step at the vendor rate: %.2f ms' % (tot_us / 1e3))


if __name__ == '__main__':
    main()
