"""The kernel test of the streaming 1x1 kernel as it stood in tests/test_hip_kernels.py (round 5). Not collected: the kernel is not in the library any more
(hook it back in through hook notes in README.md, then run this file with the helpers of tests/test_hip_kernels.py in scope)."""
from tests.test_hip_kernels import *      # noqa: F401,F403 -- helpers (rnd, nhwc, nchw, close16, b16 ...)


PW16_CASES = [  # n, cin, h, w, cout: pointwise, K = 64 / 128 / 256 after padding
    (2, 64, 37, 41, 256),      # ragged last 64-row tile
    (1, 128, 48, 48, 512),     # two 256-channel chunks
    (2, 256, 24, 24, 1024),    # K = 256: 32-pixel tiles, eight 128-channel chunks
    (1, 64, 40, 40, 128),      # fewer output channels than the chunk: out-of-range weight rows read zeros, stores are masked
    (3, 256, 17, 19, 384),     # three chunks of 128
    (1, 128, 30, 30, 320),     # second chunk: 64 valid channels
    (1, 48, 33, 35, 256),      # 48 input channels inside a zero-padded 64-channel pitch (K.new registers the pad)
    (2, 256, 31, 29, 64),      # the DATA GRADIENT is the streaming case here (dy 64 channels -> dx 256, fused skip gradient = the residual path)
    (1, 512, 20, 20, 128),     # likewise: dgrad K = 128, N = 512
]


@pytest.mark.parametrize('case', PW16_CASES)
def test_pw16_streaming_1x1(K, case):
    """Round 5: the streaming 1x1 convolution of the bf16 tier (csrc/pw16.hip: persistent blocks, weights resident in LDS, four-stage LDS-DMA ring with counted waits,
    16-byte stores straight from the accumulators through a half-wave exchange) forced onto small shapes (pm_set_conv16(5)): forward plain, forward with the whole
    fused epilogue (bias, scale / shift, residual, ReLU -- the residual rows travel as inline-asm loads with their own counted wait), the data gradient of a 1x1
    convolution with its fused skip gradient; each against the fp32 formula on the same bf16 values (one bf16 rounding) and against the implicit-GEMM kernels
    (pm_set_conv16(6)). Repeated three times on fresh outputs: the counted waits are a race screen's business."""
    n, cin, h, w, cout = case
    r16 = lambda t: t.bfloat16().float()
    x, wt = r16(rnd(n, cin, h, w, seed=1)), rnd(cout, cin, 1, 1, seed=2, scale=(2.0 / cin) ** 0.5)
    b = rnd(cout, seed=3)
    sc, sh = rnd(cout, seed=6).abs() + 0.5, rnd(cout, seed=7)
    res = r16(rnd(n, cout, h, w, seed=8))
    y_lin = F.conv2d(x, r16(wt))
    dy, skip = r16(rnd(*y_lin.shape, seed=4)), r16(rnd(n, cin, h, w, seed=5))
    K.set_conv_precision('bf16')
    outs = {}
    try:
        wg = wt.permute(0, 2, 3, 1).contiguous().cuda()
        xg = K.new((n, h, w, cin), wg, dtype=torch.bfloat16)
        xg.copy_(b16(x))
        for route in (5, 6):
            K.set_conv16(route)
            rep = []
            for _ in range(3 if route == 5 else 1):
                y0 = K.conv_fwd(xg, wg, 1, 0, 1)
                y1 = K.conv_fwd(xg, wg, 1, 0, 1, bias=b.cuda(), scale=sc.cuda(), shift=sh.cuda(), residual=b16(res), relu=True)
                dx = K.conv_bwd_data(b16(dy), wg, tuple(xg.shape), 1, 0, 1, add=b16(skip)) if cout in (64, 128, 256) else None
                rep.append((y0, y1, dx))
            for other in rep[1:]:
                assert all(a is None or torch.equal(a, c) for a, c in zip(rep[0], other)), 'the streaming kernel is not run-to-run identical'
            outs[route] = rep[0]
    finally:
        K.set_conv16(1)
        K.set_conv_precision('f32')
    y0, y1, dx = outs[5]
    close16(nchw(y0.float()), y_lin, ulps=1.5)
    close16(nchw(y1.float()), torch.relu((y_lin + b.view(1, -1, 1, 1)) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res), ulps=1.5)
    if dx is not None:
        x2 = x.clone().requires_grad_(True)
        F.conv2d(x2, r16(wt)).backward(dy)
        close16(nchw(dx.float()), x2.grad + skip, ulps=1.5)
    for a, c in zip(outs[5], outs[6]):
        if a is not None:
            close16(a.float(), c.float(), ulps=2.5)


