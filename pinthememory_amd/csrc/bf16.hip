// configs[2] (BASELINE.json: "bf16 with MFMA conv kernels"): operand preparation for the bf16 tier of the convolution kernels.
// The activations / gradients of the network stay fp32 in HBM between layers (BatchNorm statistics, losses and the memory need them);
// each convolution call converts its two GEMM operands to bf16 IN HBM with one streaming pass (round to nearest even), so that the
// implicit-GEMM kernel gathers bf16 rows (half the bytes), stages bf16 tiles in LDS (64 k-values per 128-byte row) and feeds
// v_mfma_f32_32x32x16_bf16 straight from ds_read_b128 fragments; accumulation and the epilogue stay fp32.
//   pm_bf16_cast_rows      x[P][pitch] fp32 -> xb[P][Cp] bf16, channels C..Cp-1 zero (Cp = C rounded up to 64: a K-slab never straddles a tap)
//   pm_bf16_cast_weights   w[Cout][T][Cin] fp32 (KRSC) -> wb[Cout][T][Cp]                       (forward)
//                          or the rotated / transposed filter wr[Cin][T][Coutp], tap t <- T-1-t   (data gradient = forward conv of dy)
//   pm_bf16_transpose_taps x[N][H][W][C] fp32 -> xt[T][C][N*Ho*Wo] bf16: per tap the input pixel each output pixel sees (0 outside the
//                          image), pixel-contiguous; and dy[P][Cout] -> dyt[Cout][P]. The weight gradient dw[co][(t, ci)] = sum_p
//                          dyt[co][p] xt[t][ci][p] is then a plain k-contiguous GEMM over pixels for the same kernel.
// Replaces nothing in the reference by itself: it is the operand edge of nn.Conv2d under the bf16 tier (Resnet.py:145-150, deepv3plus.py:72-81,398-424).
#include "pm_common.h"

namespace {

__device__ __forceinline__ unsigned short f2bf(float f) {      // round to nearest even; NaN stays NaN
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

// thread -> 8 consecutive channels of one pixel: two 16-byte loads, one 16-byte store
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ x, long pitch, int C, int Cp, long P, unsigned short* __restrict__ out) {
  const int g8 = Cp / 8;
  const long total = P * g8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / g8;
    const int c = (int)(i - p * g8) * 8;
    float v[8];
    if (c + 8 <= C) {
      const float4 a = PM_LD4(x + p * pitch + c), b = PM_LD4(x + p * pitch + c + 4);
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = c + e < C ? x[p * pitch + c + e] : 0.f;
    }
    uint4 o;
    o.x = pack2(v[0], v[1]), o.y = pack2(v[2], v[3]), o.z = pack2(v[4], v[5]), o.w = pack2(v[6], v[7]);
    *reinterpret_cast<uint4*>(out + p * Cp + c) = o;
  }
}

// weights are small (<= 19 MB): one element per thread
__global__ __launch_bounds__(256) void cast_weights_kernel(const float* __restrict__ w, int Cout, int T, int Cin, int Cp, int rotate, unsigned short* __restrict__ out) {
  // forward:  out[co][t][c]  (c < Cp)  = w[co][t][c]
  // rotate:   out[ci][t][c]  (c < Cp)  = w[c][T-1-t][ci]      (c runs over Cout)
  const int R = rotate ? Cin : Cout, Cs = rotate ? Cout : Cin;
  const long total = (long)R * T * Cp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const int t = (int)((i / Cp) % T);
    const int r = (int)(i / ((long)Cp * T));
    float v = 0.f;
    if (c < Cs) v = rotate ? w[((long)c * T + (T - 1 - t)) * Cin + r] : w[((long)r * T + t) * Cin + c];
    out[i] = f2bf(v);
  }
}

// rotate == 1 as a tiled transpose: per tap a [Cout][Cin] -> [Cin][Coutp] transpose, 32 x 32 tiles through LDS, coalesced on both sides (the
// element-per-thread form above reads the source at a stride of T * Cin floats: 7 us per layer, 1 ms per step over the ~140 data gradients)
__global__ __launch_bounds__(256) void cast_weights_rot_kernel(const float* __restrict__ w, int Cout, int T, int Cin, int Cp, unsigned short* __restrict__ out) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z, r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;      // r over Cin (output rows), c over Cout (output columns, padded to Cp)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    tile[j][tx] = (c < Cout && r < Cin) ? w[((long)c * T + (T - 1 - t)) * Cin + r] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    if (r < Cin && c < Cp) out[((long)r * T + t) * Cp + c] = f2bf(tile[tx][j]);
  }
}

// Pixel-contiguous bf16 operands of the weight gradient. Block = 64 output pixels x 64 channels of ONE tap: the 64 source rows are read as
// coalesced channel vectors, transposed through LDS, and written as 64 rows (channels) of 64 consecutive pixels (128 bytes each).
// T == 1 with stride 1 / pad 0 is the plain transpose (dy -> dyt, and x of a 1x1 convolution).
struct TapGeom {
  int N, H, W, Ho, Wo, kw, stride, pad, dil;
};
__global__ __launch_bounds__(256) void transpose_taps_kernel(const float* __restrict__ x, long pitch, int C, TapGeom g, long P, unsigned short* __restrict__ out) {
  __shared__ float tile[64][65];
  const int tap = blockIdx.z;
  const long p0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int ky = tap / g.kw, kx = tap - ky * g.kw;
  // load: thread -> (pixel r + 16 i, channel group q): 16 float4 per pixel row
  const int q = threadIdx.x & 15, r = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long p = p0 + r + 16 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < P) {
      const int ox = (int)(p % g.Wo), oy = (int)((p / g.Wo) % g.Ho), n = (int)(p / ((long)g.Wo * g.Ho));
      const int iy = oy * g.stride - g.pad + ky * g.dil, ix = ox * g.stride - g.pad + kx * g.dil;
      const int c = c0 + q * 4;
      if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) {
        const float* src = x + ((long)(n * g.H + iy) * g.W + ix) * pitch + c;
        if (c + 4 <= C) v = PM_LD4(src);
        else {
          if (c < C) v.x = src[0];
          if (c + 1 < C) v.y = src[1];
          if (c + 2 < C) v.z = src[2];
        }
      }
    }
    tile[r + 16 * i][q * 4] = v.x, tile[r + 16 * i][q * 4 + 1] = v.y, tile[r + 16 * i][q * 4 + 2] = v.z, tile[r + 16 * i][q * 4 + 3] = v.w;
  }
  __syncthreads();
  // store: thread -> (channel cc + 32 j, pixel group of 8): eight 16-byte stores cover one channel row of the tile
  const int pg = threadIdx.x & 7, cc = threadIdx.x >> 3;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = c0 + cc + 32 * j;
    if (c >= C) continue;
    const long p = p0 + pg * 8;
    if (p >= P) continue;                                  // P % 8 == 0 is required by the host (whole 16-byte groups)
    uint4 o;
    const float* t = &tile[pg * 8][cc + 32 * j];
    o.x = pack2(t[0], t[65]), o.y = pack2(t[2 * 65], t[3 * 65]), o.z = pack2(t[4 * 65], t[5 * 65]), o.w = pack2(t[6 * 65], t[7 * 65]);
    *reinterpret_cast<uint4*>(out + ((long)tap * C + c) * P + p) = o;
  }
}

// Batched form of the two weight casts above: every block takes one 32 x 32 tile (rows x padded channels of one tap) of one job; the job table travels in the
// kernel arguments (no device-side table to keep alive), blocks find their job by bisection over the tile prefix sums.
constexpr int WX_BATCH = 64;
struct WxJobs {
  const float* w[WX_BATCH];
  unsigned short* out[WX_BATCH];
  int cout[WX_BATCH], T[WX_BATCH], cin[WX_BATCH], cp[WX_BATCH], rot[WX_BATCH];
  int tile_start[WX_BATCH + 1];
  int n;
};
__global__ __launch_bounds__(256) void cast_weights_multi_kernel(const WxJobs tab) {
  __shared__ float tile[32][33];
  int lo = 0, hi = tab.n;                        // largest j with tile_start[j] <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tab.tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
  }
  const int j = lo, Cout = tab.cout[j], T = tab.T[j], Cin = tab.cin[j], Cp = tab.cp[j];
  const float* __restrict__ w = tab.w[j];
  unsigned short* __restrict__ out = tab.out[j];
  const int R = tab.rot[j] ? Cin : Cout;         // rows of the output
  const int ctiles = Cp / 32, rtiles = (R + 31) / 32;
  int id = blockIdx.x - tab.tile_start[j];
  const int ct = id % ctiles; id /= ctiles;
  const int rt = id % rtiles;
  const int t = id / rtiles;
  const int r0 = rt * 32, c0 = ct * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (!tab.rot[j]) {                             // out[r][t][c] = w[r][t][c], c < Cin, else 0
    for (int jj = ty; jj < 32; jj += 8) {
      const int r = r0 + jj, c = c0 + tx;
      if (r < R) out[((long)r * T + t) * Cp + c] = f2bf(c < Cin ? w[((long)r * T + t) * Cin + c] : 0.f);
    }
    return;
  }
  for (int jj = ty; jj < 32; jj += 8) {          // out[r][t][c] = w[c][T-1-t][r]: r over Cin, c over Cout (padded to Cp)
    const int c = c0 + jj, r = r0 + tx;
    tile[jj][tx] = (c < Cout && r < Cin) ? w[((long)c * T + (T - 1 - t)) * Cin + r] : 0.f;
  }
  __syncthreads();
  for (int jj = ty; jj < 32; jj += 8) {
    const int r = r0 + jj, c = c0 + tx;
    if (r < Cin) out[((long)r * T + t) * Cp + c] = f2bf(tile[tx][jj]);
  }
}

// Stride-2 data gradient on the bf16 tier: the sub-filter of one input-pixel parity class, as the [Cin][taps'][Coutp] operand of a stride-1 forward convolution
// of dy (conv16.hip). Taps of the class: ky = ky0 + 2 i, kx = kx0 + 2 j; the forward form visits them in flipped order (dy row = py + i' with i' = nky - 1 - i).
// blockIdx.z = (class, tap'); a 32 x 32 [Cout][Cin] -> [Cin][Coutp] transpose per block.
__global__ __launch_bounds__(256) void cast_weights_s2_kernel(const float* __restrict__ w, int Cout, int kh, int kw, int Cin, int Cp, const PmS2Classes cl) {
  __shared__ float tile[32][33];
  int zz = blockIdx.z, c = 0;
  while (c < 4 && zz >= cl.nky[c] * cl.nkx[c]) zz -= cl.nky[c] * cl.nkx[c], ++c;
  if (c >= 4) return;
  const int nkx = cl.nkx[c], iy = zz / nkx, ix = zz - iy * nkx;
  const int ky = cl.ky0[c] + 2 * (cl.nky[c] - 1 - iy), kx = cl.kx0[c] + 2 * (nkx - 1 - ix);
  const int Tc = cl.nky[c] * nkx;
  unsigned short* __restrict__ out = reinterpret_cast<unsigned short*>(cl.out[c]);
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // r over Cin (rows of the output), c over Cout (padded to Cp)
  for (int j = ty; j < 32; j += 8) {
    const int co = c0 + j, ci = r0 + tx;
    tile[j][tx] = (co < Cout && ci < Cin) ? w[(((long)co * kh + ky) * kw + kx) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int ci = r0 + j, co = c0 + tx;
    if (ci < Cin && co < Cp) out[((long)ci * Tc + zz) * Cp + co] = f2bf(tile[tx][j]);
  }
}

}  // namespace

int pm_bf16_cast_weights_s2(const float* w, int Cout, int kh, int kw, int Cin, int Cp, const PmS2Classes* cl, hipStream_t st) {
  int taps = 0;
  for (int c = 0; c < 4; ++c) taps += cl->nky[c] * cl->nkx[c];
  if (taps == 0) return PM_OK;
  hipLaunchKernelGGL(cast_weights_s2_kernel, dim3((Cin + 31) / 32, (Cp + 31) / 32, taps), dim3(256), 0, st, w, Cout, kh, kw, Cin, Cp, *cl);
  return pm_check_launch("bf16_cast_weights_s2");
}

extern "C" int pm_conv_wxf_refresh_bf16(const pm_wxf_job* jobs, int n, void* stream) {
  PM_REQUIRE(n >= 0 && (jobs || n == 0), PM_EINVAL, "conv_wxf_refresh_bf16: bad job table");
  for (int i = 0; i < n; ++i) {
    const pm_wxf_job& q = jobs[i];
    PM_REQUIRE(q.w && q.wxf && q.cout > 0 && q.cin > 0 && q.kh > 0 && q.kw > 0, PM_EINVAL, "conv_wxf_refresh_bf16: job %d: bad arguments", i);
    const long rows = q.dgrad ? q.cin : q.cout, cp = ((q.dgrad ? q.cout : q.cin) + 63) / 64 * 64;
    PM_REQUIRE(q.wxf_bytes >= rows * q.kh * q.kw * cp * 2, PM_EINVAL, "conv_wxf_refresh_bf16: job %d: buffer of %ld bytes, the filter needs %ld", i, (long)q.wxf_bytes,
               rows * q.kh * q.kw * cp * 2);
    PM_REQUIRE(pm_aligned16(q.wxf), PM_EINVAL, "conv_wxf_refresh_bf16: job %d: buffer must be 16-byte aligned", i);
  }
  for (int i0 = 0; i0 < n; i0 += WX_BATCH) {
    WxJobs tab;
    tab.n = std::min(WX_BATCH, n - i0);
    long tiles = 0;
    for (int i = 0; i < tab.n; ++i) {
      const pm_wxf_job& q = jobs[i0 + i];
      tab.w[i] = q.w, tab.out[i] = (unsigned short*)q.wxf, tab.cout[i] = q.cout, tab.T[i] = q.kh * q.kw, tab.cin[i] = q.cin, tab.rot[i] = q.dgrad ? 1 : 0;
      tab.cp[i] = ((q.dgrad ? q.cout : q.cin) + 63) / 64 * 64;
      tab.tile_start[i] = (int)tiles;
      tiles += (long)(((q.dgrad ? q.cin : q.cout) + 31) / 32) * (tab.cp[i] / 32) * tab.T[i];
      PM_REQUIRE(tiles < (1l << 30), PM_EINVAL, "conv_wxf_refresh_bf16: too many tiles in one batch");
    }
    tab.tile_start[tab.n] = (int)tiles;
    hipLaunchKernelGGL(cast_weights_multi_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, tab);
    if (int e = pm_check_launch("conv_wxf_refresh_bf16")) return e;
  }
  return PM_OK;
}

int pm_bf16_cast_rows(const float* x, long pitch, int C, int Cp, long P, void* out, hipStream_t st) {
  const long total = P * (Cp / 8);
  if (total == 0) return PM_OK;
  hipLaunchKernelGGL(cast_rows_kernel, dim3((int)std::min<long>((total + 255) / 256, 256 * 16)), dim3(256), 0, st, x, pitch, C, Cp, P, (unsigned short*)out);
  return pm_check_launch("bf16_cast_rows");
}

int pm_bf16_cast_weights(const float* w, int Cout, int T, int Cin, int Cp, bool rotate, void* out, hipStream_t st) {
  const long total = (long)(rotate ? Cin : Cout) * T * Cp;
  if (rotate) {
    hipLaunchKernelGGL(cast_weights_rot_kernel, dim3((Cin + 31) / 32, (Cp + 31) / 32, T), dim3(256), 0, st, w, Cout, T, Cin, Cp, (unsigned short*)out);
    return pm_check_launch("bf16_cast_weights(rot)");
  }
  hipLaunchKernelGGL(cast_weights_kernel, dim3((int)std::min<long>((total + 255) / 256, 256 * 16)), dim3(256), 0, st, w, Cout, T, Cin, Cp, rotate ? 1 : 0,
                     (unsigned short*)out);
  return pm_check_launch("bf16_cast_weights");
}

int pm_bf16_transpose_taps(const float* x, long pitch, int C, int N, int H, int W, int Ho, int Wo, int kh, int kw, int stride, int pad, int dil, void* out,
                           hipStream_t st) {
  const long P = (long)N * Ho * Wo;
  const TapGeom g = {N, H, W, Ho, Wo, kw, stride, pad, dil};
  hipLaunchKernelGGL(transpose_taps_kernel, dim3((unsigned)((P + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)(kh * kw)), dim3(256), 0, st, x, pitch, C, g, P,
                     (unsigned short*)out);
  return pm_check_launch("bf16_transpose_taps");
}
