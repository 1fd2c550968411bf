// BASELINE configs[2], the bf16 tier: every HBM-bound kernel between the convolutions on bf16 ACTIVATIONS (NHWC, pm_tensor.dtype == PM_BF16).
// A lane moves 16 bytes = 8 channels per access (a wave: 1 KB of consecutive memory per instruction); values are widened to fp32 in registers,
// every statistic / reduction / accumulator is fp32 (second stages in double), results are rounded to bf16 (nearest even) once, on the way out.
// Half the bytes of the fp32 kernels in bn.hip / pool_resize.hip / misc.hip, whose arithmetic each kernel here repeats -- same formulas, same
// fixed-order (atomic-free, deterministic) reductions. The extern "C" entry points of those files dispatch here on dtype.
// Replaces on the tier: mynn.Norm2d (/root/reference/network/mynn.py:8-14) train forward / backward, nn.MaxPool2d(3,2,1) (Resnet.py:432),
// nn.AdaptiveAvgPool2d(1) (deepv3plus.py:85), mynn.Upsample (mynn.py:57-62), the residual / fan-in adds of autograd.
#include <stdlib.h>

#include "pm_common.h"

namespace {

constexpr int V = 8;                       // channels per lane

__device__ __forceinline__ float bn_affine(float v, float mu, float is, float ga, float be) {      // == bn.hip: identical in forward and mask rebuild
  const float s = is * ga;
  return fmaf(v, s, fmaf(-mu, s, be));
}
__device__ __forceinline__ void ld8f(const float* p, float* v) {
  const float4 a = PM_LD4(p), b = PM_LD4(p + 4);
  v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
}

// thread -> (pixel, 8 channels)
template <typename F>
__global__ __launch_bounds__(256) void ew16_kernel(long pixels, int cg, F f) {
  const long total = pixels * cg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / cg;
    f(p, (int)(i - p * cg) * V);
  }
}
template <typename F>
int ew16_launch(long pixels, int c, hipStream_t st, const char* name, F f) {
  if (pixels * c == 0) return PM_OK;
  const long work = pixels * (c / V);
  hipLaunchKernelGGL((ew16_kernel<F>), dim3((int)std::min<long>((work + 255) / 256, 256 * 16)), dim3(256), 0, st, pixels, c / V, f);
  return pm_check_launch(name);
}

// ---------------- BatchNorm ----------------------------------------------------------------------------------------------------------
// GPR lane groups per pixel row (8 for tensors of <= 64 channels: no idle half-waves on the 64-channel maps of layer1), RL = 256 / GPR row lanes
template <int GPR>
struct Geo {
  static constexpr int RL = 256 / GPR, CB = GPR * V;
};
inline int chunk_rows16(long P, int C, int RL, int CB) {      // pixels per block: >= 2048 blocks over (pixel chunks x channel groups), >= 64 rows each
  const long colblocks = std::max<long>(1, (C + CB - 1) / CB);
  const long want = std::max<long>(512, 2048 / colblocks);
  long r = (P + want - 1) / want;
  r = std::max<long>(r, 64);
  return (int)((r + RL - 1) / RL * RL);
}
struct Plan16 {
  int gpr, rows, nb, colblocks;
};
inline Plan16 plan16(long P, int C) {
  Plan16 p;
  p.gpr = C <= 64 ? 8 : 16;
  const int RL = 256 / p.gpr, CB = p.gpr * V;
  p.rows = chunk_rows16(P, C, RL, CB);
  p.nb = pm_cdiv(P, p.rows);
  p.colblocks = pm_cdiv(C, CB);
  return p;
}

// block-level reduce of per-thread (s1[8], s2[8]) over the row lanes -> part[blk][c][2]
template <int GPR>
__device__ __forceinline__ void block_reduce_store(const float* s1, const float* s2, int g, int r, int C, float* __restrict__ part, float (*sm)[Geo<GPR>::CB][2]) {
  constexpr int RL = Geo<GPR>::RL, CB = Geo<GPR>::CB;
#pragma unroll
  for (int j = 0; j < V; ++j) sm[r][g * V + j][0] = s1[j], sm[r][g * V + j][1] = s2[j];
  __syncthreads();
  if (threadIdx.x < CB * 2) {
    const int cc = threadIdx.x >> 1, w = threadIdx.x & 1;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RL; ++i) s += sm[i][cc][w];
    const int ch = blockIdx.y * CB + cc;
    if (ch < C) part[((long)blockIdx.x * C + ch) * 2 + w] = s;
  }
}

// partial[blk][c][2] : sum(x - K[c]), sum((x - K[c])^2) over the block's pixel chunk, K = the first pixel (shifted sums: no cancellation)
template <int GPR>
__global__ __launch_bounds__(256) void bn16_stats_partial(const pm_bf16* __restrict__ x, long pitch, long P, int C, int rows, float* __restrict__ part) {
  __shared__ float sm[Geo<GPR>::RL][Geo<GPR>::CB][2];
  const int g = threadIdx.x % GPR, r = threadIdx.x / GPR;
  const int c = blockIdx.y * Geo<GPR>::CB + g * V;
  const long p0 = (long)blockIdx.x * rows, p1 = min(P, p0 + rows);
  float s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.f;
  if (c < C) {
    float k[V];
    pm_ld8(x + c, k);
    for (long p = p0 + r; p < p1; p += Geo<GPR>::RL) {
      float v[V];
      pm_ld8(x + p * pitch + c, v);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float d = v[j] - k[j];
        s1[j] += d, s2[j] += d * d;
      }
    }
  }
  block_reduce_store<GPR>(s1, s2, g, r, C, part, sm);
}

// Second stage: 4 channels x 64 lanes per block, double sums in fixed order (as bn.hip)
constexpr int FC = 4, FL = 64;
__device__ __forceinline__ void final_sums(const float* __restrict__ part, int nb, int C, int c, int lane, double& s1, double& s2) {
  __shared__ double red[FL][FC][2];
  double a = 0.0, b = 0.0;
  if (c < C)
#pragma unroll 8
    for (int i = lane; i < nb; i += FL) {
      const float2 v = *reinterpret_cast<const float2*>(part + ((long)i * C + c) * 2);
      a += (double)v.x, b += (double)v.y;
    }
  red[lane][threadIdx.x & (FC - 1)][0] = a, red[lane][threadIdx.x & (FC - 1)][1] = b;
  __syncthreads();
  s1 = s2 = 0.0;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < FL; ++i) s1 += red[i][threadIdx.x & (FC - 1)][0], s2 += red[i][threadIdx.x & (FC - 1)][1];
  }
}
template <bool FIN>
__global__ __launch_bounds__(256) void bn16_stats_final(const float* __restrict__ part, int nb, const pm_bf16* __restrict__ x, long P, int C, float* __restrict__ moments,
                                                        float eps, float* __restrict__ mean, float* __restrict__ invstd, float* running_mean, float* running_var,
                                                        float momentum) {
  const int c = blockIdx.x * FC + (threadIdx.x & (FC - 1)), lane = threadIdx.x / FC;
  double s1, s2;
  final_sums(part, nb, C, c, lane, s1, s2);
  if (lane != 0 || c >= C) return;
  const double n = (double)P;
  const float m = (float)((double)pm_bf16_to_f32(x[c]) + s1 / n), m2 = (float)fmax(s2 - s1 * s1 / n, 0.0), nf = (float)n;
  if constexpr (!FIN) {
    moments[c] = m, moments[C + c] = m2, moments[2 * C + c] = nf;
  } else {
    const float var = m2 / nf;
    mean[c] = m;
    invstd[c] = 1.f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    if (running_var && nf > 1.f) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / (nf - 1.f));
  }
}

// backward reductions: partial[blk][c][2] = sum(dyz), sum(dyz * xhat), dyz = dy masked by the ReLU of the forward pass.
// RELU 0 none; 1 mask = y > 0 from the forward output; 2 mask rebuilt from x (gamma, beta); 3 mask from the byte per 8-channel group bn16_apply left.
// GOUT: also store dyz (bf16) -- the gradient of the residual branch, and the apply pass's only gradient operand.
template <int GPR, int RELU, bool GOUT>
__global__ __launch_bounds__(256) void bn16_bwd_partial(const pm_bf16* __restrict__ dy, long dpitch, const pm_bf16* __restrict__ y, long ypitch,
                                                        const uint8_t* __restrict__ mask, const pm_bf16* __restrict__ x, long xpitch, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        pm_bf16* __restrict__ gout, long gpitch, long P, int C, int rows, float* __restrict__ part) {
  __shared__ float sm[Geo<GPR>::RL][Geo<GPR>::CB][2];
  const int g = threadIdx.x % GPR, r = threadIdx.x / GPR;
  const int c = blockIdx.y * Geo<GPR>::CB + g * V;
  const long p0 = (long)blockIdx.x * rows, p1 = min(P, p0 + rows);
  float s1[V], s2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s1[j] = s2[j] = 0.f;
  if (c < C) {
    float mu[V], is[V], ga[V], be[V];
    ld8f(mean + c, mu), ld8f(invstd + c, is);
    if (RELU == 2) ld8f(gamma + c, ga), ld8f(beta + c, be);
    const int cg = C >> 3;
    for (long p = p0 + r; p < p1; p += Geo<GPR>::RL) {
      float d[V], v[V];
      pm_ld8(dy + p * dpitch + c, d);
      pm_ld8(x + p * xpitch + c, v);
      if (RELU == 1) {
        float o[V];
        pm_ld8(y + p * ypitch + c, o);
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
      } else if (RELU == 3) {
        const unsigned mb = mask[p * cg + (c >> 3)];
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] = ((mb >> j) & 1u) ? d[j] : 0.f;
      } else if (RELU == 2) {
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] = bn_affine(v[j], mu[j], is[j], ga[j], be[j]) > 0.f ? d[j] : 0.f;
      }
      if (GOUT) pm_st8(gout + p * gpitch + c, d);
#pragma unroll
      for (int j = 0; j < V; ++j) s1[j] += d[j], s2[j] += d[j] * ((v[j] - mu[j]) * is[j]);
    }
  }
  block_reduce_store<GPR>(s1, s2, g, r, C, part, sm);
}
__global__ __launch_bounds__(256) void bn16_bwd_final(const float* __restrict__ part, int nb, int C, float* __restrict__ sums) {
  const int c = blockIdx.x * FC + (threadIdx.x & (FC - 1)), lane = threadIdx.x / FC;
  double s1, s2;
  final_sums(part, nb, C, c, lane, s1, s2);
  if (lane != 0 || c >= C) return;
  sums[c] = (float)s1;
  sums[C + c] = (float)s2;
}

// The two elementwise BatchNorm passes with the per-channel constants in REGISTERS: thread = one 8-channel group (fixed) x a strided set of pixels. The generic
// driver above re-loads mean / invstd / gamma / beta (/ sums) -- 128-192 bytes of L1 traffic -- for every 16-byte group of activations it moves, which held these
// passes at 4.0-4.4 TB/s where their fp32 twins reach 5.6 (twice the tensor bytes per parameter load). Taken when the channel groups divide the block (C = 64 ... 2048).
template <bool RES, bool MASK, bool RELU>
__global__ __launch_bounds__(256) void bn16_apply_fixed_kernel(const pm_bf16* __restrict__ x, long xp, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, const pm_bf16* __restrict__ res, long rp,
                                                               pm_bf16* __restrict__ y, long yp, uint8_t* __restrict__ mask, long pixels, int cg) {
  const int grp = threadIdx.x % cg, pl = threadIdx.x / cg, ppb = 256 / cg, ch = grp * V;
  float sc[V], sh[V];
  {
    float mu[V], is[V], ga[V], be[V];
    ld8f(mean + ch, mu), ld8f(invstd + ch, is), ld8f(gamma + ch, ga), ld8f(beta + ch, be);
#pragma unroll
    for (int e = 0; e < V; ++e) sc[e] = is[e] * ga[e], sh[e] = fmaf(-mu[e], sc[e], be[e]);      // bn_affine(v) == fmaf(v, sc, sh): the same two FMAs
  }
  for (long p = (long)blockIdx.x * ppb + pl; p < pixels; p += (long)gridDim.x * ppb) {
    float v[V], o[V];
    pm_ld8(x + p * xp + ch, v);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = fmaf(v[e], sc[e], sh[e]);
    if constexpr (RES) {
      float q[V];
      pm_ld8(res + p * rp + ch, q);
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] += q[e];
    }
    if constexpr (MASK) {
      unsigned m = 0;
#pragma unroll
      for (int e = 0; e < V; ++e) m |= o[e] > 0.f ? (1u << e) : 0u;
      mask[p * cg + grp] = (unsigned char)m;
    }
    if constexpr (RELU) {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    pm_st8(y + p * yp + ch, o);
  }
}

// MODE 0: no ReLU (dy is already the masked gradient); 1: mask = forward output > 0; 2: mask rebuilt from x
template <int MODE, bool DRES>
__global__ __launch_bounds__(256) void bn16_bwd_apply_fixed_kernel(const pm_bf16* __restrict__ dy, long dp, const pm_bf16* __restrict__ yo, long op, const pm_bf16* __restrict__ x,
                                                                   long xp, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float* __restrict__ sums, float host_inv_n, int dev_count, int C,
                                                                   pm_bf16* __restrict__ dx, long dxp, pm_bf16* __restrict__ dres, long drp, long pixels, int cg) {
  const int grp = threadIdx.x % cg, pl = threadIdx.x / cg, ppb = 256 / cg, ch = grp * V;
  float mu[V], is[V], k1[V], k2[V], sg[V], sc[V], sh[V];
  {
    const float inv_n = dev_count ? 1.f / sums[2 * C] : host_inv_n;
    float ga[V], s1[V], s2[V];
    ld8f(mean + ch, mu), ld8f(invstd + ch, is), ld8f(gamma + ch, ga), ld8f(sums + ch, s1), ld8f(sums + C + ch, s2);
#pragma unroll
    for (int e = 0; e < V; ++e) k1[e] = s1[e] * inv_n, k2[e] = s2[e] * inv_n, sg[e] = is[e] * ga[e];
    if constexpr (MODE == 2) {
      float be[V];
      ld8f(beta + ch, be);
#pragma unroll
      for (int e = 0; e < V; ++e) sc[e] = sg[e], sh[e] = fmaf(-mu[e], sg[e], be[e]);
    }
  }
  for (long p = (long)blockIdx.x * ppb + pl; p < pixels; p += (long)gridDim.x * ppb) {
    float g[V], v[V], r[V];
    pm_ld8(dy + p * dp + ch, g);
    pm_ld8(x + p * xp + ch, v);
    if constexpr (MODE == 1) {
      float o[V];
      pm_ld8(yo + p * op + ch, o);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] = fmaf(v[e], sc[e], sh[e]) > 0.f ? g[e] : 0.f;
    }
    if constexpr (DRES) pm_st8(dres + p * drp + ch, g);
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = (g[e] - k1[e] - (v[e] - mu[e]) * is[e] * k2[e]) * sg[e];      // the generic kernel's expression, constants hoisted
    pm_st8(dx + p * dxp + ch, r);
  }
}
inline bool fixed_ok(int c) { const int cg = c / V; return c % V == 0 && cg >= 1 && cg <= 256 && 256 % cg == 0; }
inline int fixed_grid(long pixels, int c) { const int ppb = 256 / (c / V); return (int)std::min<long>((pixels + ppb - 1) / ppb, 256 * 16); }

int check16(const pm_tensor* t, const char* who) {
  PM_REQUIRE(t && t->ptr && pm_vec8(t), PM_EINVAL, "%s: bf16 tensors must be 16B aligned with pitch %% 8 == 0 and C %% 8 == 0", who);
  return PM_OK;
}

// ---------------- max pool 3x3 s2 p1 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool16_fwd_kernel(const pm_bf16* __restrict__ x, long xp, int H, int W, pm_bf16* __restrict__ y, long yp, int Ho, int Wo,
                                                            int C, long total, uint8_t* __restrict__ arg) {
  const int cg = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long op = i / cg;
    const int ch = (int)(i - op * cg) * V;
    const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), n = (int)(op / ((long)Wo * Ho));
    float best[V];
    unsigned bi[V];
#pragma unroll
    for (int e = 0; e < V; ++e) best[e] = -INFINITY, bi[e] = 0;
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        float v[V];
        pm_ld8(x + ((long)(n * H + iy) * W + ix) * xp + ch, v);
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (first || v[e] > best[e] || v[e] != v[e]) best[e] = v[e], bi[e] = (unsigned)(ky * 3 + kx);
        first = false;
      }
    }
    pm_st8(y + op * yp + ch, best);      // exact: the maximum is one of the bf16 inputs
    uint2 a;
    a.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24), a.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
    *reinterpret_cast<uint2*>(arg + op * C + ch) = a;
  }
}
__global__ __launch_bounds__(256) void maxpool16_bwd_kernel(const pm_bf16* __restrict__ dy, long dp, int Ho, int Wo, const uint8_t* __restrict__ arg,
                                                            pm_bf16* __restrict__ dx, long xp, int H, int W, int C, long total) {
  const int cg = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long ip = i / cg;
    const int ch = (int)(i - ip * cg) * V;
    const int ix = (int)(ip % W), iy = (int)((ip / W) % H), n = (int)(ip / ((long)W * H));
    float g[V];
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    const int oy_hi = min((iy + 1) >> 1, Ho - 1), ox_hi = min((ix + 1) >> 1, Wo - 1);
    for (int oy = iy >> 1; oy <= oy_hi; ++oy) {
      const int ky = iy + 1 - 2 * oy;
      if (ky < 0 || ky > 2) continue;
      for (int ox = ix >> 1; ox <= ox_hi; ++ox) {
        const int kx = ix + 1 - 2 * ox;
        if (kx < 0 || kx > 2) continue;
        const long op = (long)(n * Ho + oy) * Wo + ox;
        const uint2 a = *reinterpret_cast<const uint2*>(arg + op * C + ch);
        const unsigned want = (unsigned)(ky * 3 + kx);
        float d[V];
        pm_ld8(dy + op * dp + ch, d);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const unsigned b = ((e < 4 ? a.x : a.y) >> (8 * (e & 3))) & 255u;
          g[e] += b == want ? d[e] : 0.f;
        }
      }
    }
    pm_st8(dx + ip * xp + ch, g);
  }
}

// ---------------- global average pool (and the column sum of the 1x1-source resize backward) ------------------------------------------
// block = one image x 128 channels: 16 lane groups x 16 row lanes, four independent partial sums per lane
__global__ __launch_bounds__(256) void gap16_fwd_kernel(const pm_bf16* __restrict__ x, long xp, long HW, int C, pm_bf16* __restrict__ y, long yp, float scale,
                                                        int accumulate) {
  __shared__ float sm[16][128];
  const int g = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = blockIdx.y * 128 + g * V, n = blockIdx.x;
  float s[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = 0.f;
  if (c < C) {
    float a[4][V];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) a[u][j] = 0.f;
    const pm_bf16* base = x + (long)n * HW * xp + c;
    long p = r;
    for (; p + 48 < HW; p += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[V];
        pm_ld8(base + (p + 16 * u) * xp, v);
#pragma unroll
        for (int j = 0; j < V; ++j) a[u][j] += v[j];
      }
    }
    for (; p < HW; p += 16) {
      float v[V];
      pm_ld8(base + p * xp, v);
#pragma unroll
      for (int j = 0; j < V; ++j) a[0][j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = (a[0][j] + a[1][j]) + (a[2][j] + a[3][j]);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) sm[r][g * V + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 128) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sm[i][threadIdx.x];
    const int ch = blockIdx.y * 128 + threadIdx.x;
    if (ch < C) {
      pm_bf16* o = y + (long)n * yp + ch;
      *o = pm_f32_to_bf16((accumulate ? pm_bf16_to_f32(*o) : 0.f) + t * scale);
    }
  }
}

// ---------------- bilinear, align_corners=True ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize16_fwd_kernel(const pm_bf16* __restrict__ x, long xp, int h, int w, pm_bf16* __restrict__ y, long yp, int H, int W,
                                                           int C, long total, float sy, float sx) {
  const int cg = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long op = i / cg;
    const int ch = (int)(i - op * cg) * V;
    const int X = (int)(op % W), Y = (int)((op / W) % H), n = (int)(op / ((long)W * H));
    const pm_lerp ly = pm_ac_lerp(sy, Y, h), lx = pm_ac_lerp(sx, X, w);
    const pm_bf16* r0 = x + ((long)(n * h + ly.i0) * w) * xp + ch;
    const pm_bf16* r1 = x + ((long)(n * h + ly.i1) * w) * xp + ch;
    float a[V], b[V], c[V], d[V], o[V];
    pm_ld8(r0 + lx.i0 * xp, a), pm_ld8(r0 + lx.i1 * xp, b), pm_ld8(r1 + lx.i0 * xp, c), pm_ld8(r1 + lx.i1 * xp, d);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = ly.w0 * (lx.w0 * a[e] + lx.w1 * b[e]) + ly.w1 * (lx.w0 * c[e] + lx.w1 * d[e]);
    pm_st8(y + op * yp + ch, o);
  }
}
__device__ __forceinline__ void support(float scale, int i, int out, int& lo, int& hi) {      // == pool_resize.hip
  if (scale <= 0.f) {
    lo = 0, hi = out - 1;
    return;
  }
  const float inv = 1.f / scale;
  lo = max(0, (int)floorf(((float)i - 1.f) * inv) - 1);
  hi = min(out - 1, (int)ceilf(((float)i + 1.f) * inv) + 1);
}
__device__ __forceinline__ float tap_weight(const pm_lerp& l, int i) { return (l.i0 == i ? l.w0 : 0.f) + (l.i1 == i ? l.w1 : 0.f); }

// gather formulation of the backward (any ratio): each input pixel sums the output pixels whose taps touch it, ascending order
__global__ __launch_bounds__(256) void resize16_bwd_kernel(const pm_bf16* __restrict__ dy, long dp, int H, int W, pm_bf16* __restrict__ dx, long xp, int h, int w,
                                                           int C, long total, float sy, float sx, int accumulate) {
  const int cg = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long ip = i / cg;
    const int ch = (int)(i - ip * cg) * V;
    const int x = (int)(ip % w), y = (int)((ip / w) % h), n = (int)(ip / ((long)w * h));
    int ylo, yhi, xlo, xhi;
    support(sy, y, H, ylo, yhi);
    support(sx, x, W, xlo, xhi);
    float g[V];
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    for (int Y = ylo; Y <= yhi; ++Y) {
      const float wy = tap_weight(pm_ac_lerp(sy, Y, h), y);
      if (wy == 0.f) continue;
      for (int X = xlo; X <= xhi; ++X) {
        const float wx = tap_weight(pm_ac_lerp(sx, X, w), x);
        if (wx == 0.f) continue;
        float q[V];
        pm_ld8(dy + ((long)(n * H + Y) * W + X) * dp + ch, q);
        const float ww = wy * wx;
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] += ww * q[e];
      }
    }
    pm_bf16* o = dx + ip * xp + ch;
    if (accumulate) {
      float q[V];
      pm_ld8(o, q);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] += q[e];
    }
    pm_st8(o, g);
  }
}
// separable backward for up-sampling ratios >= 2 (pool_resize.hip): column pass into an fp32 workspace T[n, Y, x, c], then the row pass
__global__ __launch_bounds__(256) void resize16_bwd_cols_kernel(const pm_bf16* __restrict__ dy, long dp, int H, int W, float* __restrict__ T, int w, int C, long total,
                                                                float sx) {
  const int cg = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long ip = i / cg;
    const int ch = (int)(i - ip * cg) * V;
    const int x = (int)(ip % w);
    const long row = ip / w;
    int xlo, xhi;
    support(sx, x, W, xlo, xhi);
    float g[V];
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    const pm_bf16* base = dy + (row * W) * dp + ch;
    for (int X = xlo; X <= xhi; ++X) {
      const float wx = tap_weight(pm_ac_lerp(sx, X, w), x);
      if (wx == 0.f) continue;
      float q[V];
      pm_ld8(base + (long)X * dp, q);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] += wx * q[e];
    }
    PM_ST4(T + ip * C + ch, make_float4(g[0], g[1], g[2], g[3]));
    PM_ST4(T + ip * C + ch + 4, make_float4(g[4], g[5], g[6], g[7]));
  }
}
__global__ __launch_bounds__(256) void resize16_bwd_rows_kernel(const float* __restrict__ T, int H, pm_bf16* __restrict__ dx, long xp, int h, int w, int C, long total,
                                                                float sy, int accumulate) {
  const int cg = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long ip = i / cg;
    const int ch = (int)(i - ip * cg) * V;
    const int x = (int)(ip % w), y = (int)((ip / w) % h), n = (int)(ip / ((long)w * h));
    int ylo, yhi;
    support(sy, y, H, ylo, yhi);
    float g[V];
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    for (int Y = ylo; Y <= yhi; ++Y) {
      const float wy = tap_weight(pm_ac_lerp(sy, Y, h), y);
      if (wy == 0.f) continue;
      float q[V];
      ld8f(T + (((long)n * H + Y) * w + x) * C + ch, q);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] += wy * q[e];
    }
    pm_bf16* o = dx + ip * xp + ch;
    if (accumulate) {
      float q[V];
      pm_ld8(o, q);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] += q[e];
    }
    pm_st8(o, g);
  }
}

inline int grid_for(long work) { return (int)std::min<long>((work + 255) / 256, 256 * 32); }

struct AddN16 {
  const pm_bf16* p[8];
  long pitch[8];
};

// bf16 rows -> fp32 rows / bf16 rows padded with zero channels: thread = (pixel, 8 channels of the OUTPUT)
__global__ __launch_bounds__(256) void to_f32_kernel(const pm_bf16* __restrict__ x, long pitch, int C, long P, float* __restrict__ out, long op) {
  const int cg = (C + V - 1) / V;
  const long total = P * cg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / cg;
    const int c = (int)(i - p * cg) * V;
    if (c + V <= C && ((pitch | op) & 7) == 0) {
      float v[V];
      pm_ld8(x + p * pitch + c, v);
      PM_ST4(out + p * op + c, make_float4(v[0], v[1], v[2], v[3]));
      PM_ST4(out + p * op + c + 4, make_float4(v[4], v[5], v[6], v[7]));
    } else {
      for (int e = 0; e < V && c + e < C; ++e) out[p * op + c + e] = pm_bf16_to_f32(x[p * pitch + c + e]);
    }
  }
}
__global__ __launch_bounds__(256) void pad_rows_kernel(const pm_bf16* __restrict__ x, long pitch, int C, int Cp, long P, pm_bf16* __restrict__ out) {
  const int cg = Cp / V;
  const long total = P * cg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / cg;
    const int c = (int)(i - p * cg) * V;
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (c + V <= C && (pitch & 7) == 0) q = *reinterpret_cast<const uint4*>(x + p * pitch + c);
    else {
      unsigned short e8[V];
      for (int e = 0; e < V; ++e) e8[e] = c + e < C ? x[p * pitch + c + e] : (unsigned short)0;
      q.x = e8[0] | ((unsigned)e8[1] << 16), q.y = e8[2] | ((unsigned)e8[3] << 16), q.z = e8[4] | ((unsigned)e8[5] << 16), q.w = e8[6] | ((unsigned)e8[7] << 16);
    }
    *reinterpret_cast<uint4*>(out + p * Cp + c) = q;
  }
}

}  // namespace

// ======================================================================================================================================
size_t pm16_bn_workspace(const pm_tensor* x) {
  const Plan16 pl = plan16(pm_pixels(x), x->c);
  return pm_align_up((size_t)pl.nb * x->c * 2 * sizeof(float), 256);
}

int pm16_bn_stats(const pm_tensor* x, float* moments, float eps, float* mean, float* invstd, float* running_mean, float* running_var, float momentum, void* ws,
                  size_t ws_bytes, hipStream_t st) {
  if (int e = check16(x, "bn_stats(bf16)")) return e;
  PM_REQUIRE(ws && ws_bytes >= pm16_bn_workspace(x), PM_EWORKSPACE, "bn_stats(bf16): workspace too small");
  const long P = pm_pixels(x);
  PM_REQUIRE(P > 0, PM_EINVAL, "bn_stats(bf16): empty tensor");
  PM_REQUIRE(moments || P > 1, PM_EINVAL, "bn_stats_finalize: expected more than 1 value per channel when training, got %ld", P);
  const Plan16 pl = plan16(P, x->c);
  const pm_bf16* px = (const pm_bf16*)x->ptr;
  if (pl.gpr == 8)
    hipLaunchKernelGGL(bn16_stats_partial<8>, dim3(pl.nb, pl.colblocks), dim3(256), 0, st, px, (long)x->pitch, P, x->c, pl.rows, (float*)ws);
  else
    hipLaunchKernelGGL(bn16_stats_partial<16>, dim3(pl.nb, pl.colblocks), dim3(256), 0, st, px, (long)x->pitch, P, x->c, pl.rows, (float*)ws);
  if (moments)
    hipLaunchKernelGGL(bn16_stats_final<false>, dim3(pm_cdiv(x->c, FC)), dim3(256), 0, st, (const float*)ws, pl.nb, px, P, x->c, moments, 0.f, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
  else
    hipLaunchKernelGGL(bn16_stats_final<true>, dim3(pm_cdiv(x->c, FC)), dim3(256), 0, st, (const float*)ws, pl.nb, px, P, x->c, (float*)nullptr, eps, mean, invstd,
                       running_mean, running_var, momentum);
  return pm_check_launch("bn_stats(bf16)");
}

int pm16_bn_apply_mask(const pm_tensor* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const pm_tensor* res, int relu,
                       const pm_tensor* y, uint8_t* mask, hipStream_t st) {
  if (int e = check16(x, "bn_apply(bf16)")) return e;
  if (int e = check16(y, "bn_apply(bf16)")) return e;
  PM_REQUIRE(pm_same_shape(x, y) && mean && invstd && gamma && beta, PM_EINVAL, "bn_apply(bf16): bad args");
  if (res) {
    if (int e = check16(res, "bn_apply(bf16)")) return e;
    PM_REQUIRE(pm_same_shape(x, res), PM_EINVAL, "bn_apply(bf16): residual shape mismatch");
  }
  const pm_bf16 *px = (const pm_bf16*)x->ptr, *pr = res ? (const pm_bf16*)res->ptr : nullptr;
  pm_bf16* py = (pm_bf16*)y->ptr;
  const long a = x->pitch, b = res ? res->pitch : 0, c = y->pitch, cq = x->c >> 3;
  static const int fixed_on = getenv("PM_BN16_FIXED") ? atoi(getenv("PM_BN16_FIXED")) : 1;      // A/B knob
  if (fixed_on && fixed_ok(x->c) && pm_pixels(x) > 0) {
    const long P = pm_pixels(x);
    const dim3 grid(fixed_grid(P, x->c));
    const int cg = x->c / V;
#define PM16_APPLY(R, M, L) hipLaunchKernelGGL((bn16_apply_fixed_kernel<R, M, L>), grid, dim3(256), 0, st, px, a, mean, invstd, gamma, beta, pr, b, py, c, mask, P, cg)
    if (pr && mask && relu) PM16_APPLY(true, true, true);
    else if (pr && relu) PM16_APPLY(true, false, true);
    else if (pr && mask) PM16_APPLY(true, true, false);
    else if (pr) PM16_APPLY(true, false, false);
    else if (mask && relu) PM16_APPLY(false, true, true);
    else if (relu) PM16_APPLY(false, false, true);
    else if (mask) PM16_APPLY(false, true, false);
    else PM16_APPLY(false, false, false);
#undef PM16_APPLY
    return pm_check_launch("bn_apply(bf16, fixed groups)");
  }
  return ew16_launch(pm_pixels(x), x->c, st, "bn_apply(bf16)", [=] __device__(long p, int ch) {
    float v[V], mu[V], is[V], ga[V], be[V], o[V];
    pm_ld8(px + p * a + ch, v);
    ld8f(mean + ch, mu), ld8f(invstd + ch, is), ld8f(gamma + ch, ga), ld8f(beta + ch, be);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = bn_affine(v[e], mu[e], is[e], ga[e], be[e]);
    if (pr) {
      float q[V];
      pm_ld8(pr + p * b + ch, q);
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] += q[e];
    }
    if (mask) {
      unsigned m = 0;
#pragma unroll
      for (int e = 0; e < V; ++e) m |= o[e] > 0.f ? (1u << e) : 0u;
      mask[p * cq + (ch >> 3)] = (unsigned char)m;
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    pm_st8(py + p * c + ch, o);
  });
}

int pm16_bn_bwd_reduce(const pm_tensor* dy, const pm_tensor* y, const uint8_t* mask, const pm_tensor* x, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, int relu, const pm_tensor* gmask, float* sums, void* ws, size_t ws_bytes, hipStream_t st) {
  if (int e = check16(dy, "bn_bwd_reduce(bf16)")) return e;
  if (int e = check16(x, "bn_bwd_reduce(bf16)")) return e;
  PM_REQUIRE(pm_same_shape(dy, x) && mean && invstd && sums, PM_EINVAL, "bn_bwd_reduce(bf16): bad args");
  PM_REQUIRE(relu >= 0 && relu <= 3, PM_EINVAL, "bn_bwd_reduce(bf16): relu mode %d", relu);
  PM_REQUIRE(relu != 1 || (y && pm_vec8(y) && pm_same_shape(y, x)), PM_EINVAL, "bn_bwd_reduce(bf16): relu mode 1 needs the forward output");
  PM_REQUIRE(relu != 2 || (gamma && beta), PM_EINVAL, "bn_bwd_reduce(bf16): relu mode 2 needs gamma and beta");
  PM_REQUIRE(relu != 3 || mask, PM_EINVAL, "bn_bwd_reduce(bf16): relu mode 3 needs the mask bytes");
  PM_REQUIRE(!gmask || (relu != 0 && pm_vec8(gmask) && pm_same_shape(gmask, x)), PM_EINVAL, "bn_bwd_reduce(bf16): gmask needs a ReLU mode and the shape of x");
  PM_REQUIRE(ws && ws_bytes >= pm16_bn_workspace(x), PM_EWORKSPACE, "bn_bwd_reduce(bf16): workspace too small");
  const long P = pm_pixels(x);
  const Plan16 pl = plan16(P, x->c);
  dim3 grid(pl.nb, pl.colblocks);
  const pm_bf16 *pdy = (const pm_bf16*)dy->ptr, *py = relu == 1 ? (const pm_bf16*)y->ptr : nullptr, *px = (const pm_bf16*)x->ptr;
  const long yp = relu == 1 ? y->pitch : 0;
  pm_bf16* pg = gmask ? (pm_bf16*)gmask->ptr : nullptr;
  const long gp = gmask ? gmask->pitch : 0;
#define PM16_BWD(G, R, O)                                                                                                                                        \
  hipLaunchKernelGGL((bn16_bwd_partial<G, R, O>), grid, dim3(256), 0, st, pdy, (long)dy->pitch, py, yp, mask, px, (long)x->pitch, mean, invstd, gamma, beta, pg, gp, \
                     P, x->c, pl.rows, (float*)ws)
#define PM16_BWD_G(G)                                   \
  do {                                                  \
    if (relu == 0) PM16_BWD(G, 0, false);               \
    else if (relu == 1 && gmask) PM16_BWD(G, 1, true);  \
    else if (relu == 1) PM16_BWD(G, 1, false);          \
    else if (relu == 2 && gmask) PM16_BWD(G, 2, true);  \
    else if (relu == 2) PM16_BWD(G, 2, false);          \
    else if (gmask) PM16_BWD(G, 3, true);               \
    else PM16_BWD(G, 3, false);                         \
  } while (0)
  if (pl.gpr == 8) PM16_BWD_G(8);
  else PM16_BWD_G(16);
#undef PM16_BWD_G
#undef PM16_BWD
  hipLaunchKernelGGL(bn16_bwd_final, dim3(pm_cdiv(x->c, FC)), dim3(256), 0, st, (const float*)ws, pl.nb, x->c, sums);
  return pm_check_launch("bn_bwd_reduce(bf16)");
}

int pm16_bn_bwd_apply(const pm_tensor* dy, const pm_tensor* y, const pm_tensor* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* sums, float count, int relu, const pm_tensor* dx, const pm_tensor* dres, hipStream_t st) {
  if (int e = check16(dy, "bn_bwd_apply(bf16)")) return e;
  if (int e = check16(x, "bn_bwd_apply(bf16)")) return e;
  if (int e = check16(dx, "bn_bwd_apply(bf16)")) return e;
  PM_REQUIRE(pm_same_shape(dy, x) && pm_same_shape(dx, x) && mean && invstd && gamma && sums, PM_EINVAL, "bn_bwd_apply(bf16): bad args");
  PM_REQUIRE(relu >= 0 && relu <= 2, PM_EINVAL, "bn_bwd_apply(bf16): relu mode %d (0 none, 1 mask from y, 2 mask rebuilt from x)", relu);
  PM_REQUIRE(relu != 1 || (y && pm_vec8(y) && pm_same_shape(y, x)), PM_EINVAL, "bn_bwd_apply(bf16): relu mode 1 needs the forward output");
  PM_REQUIRE(relu != 2 || beta, PM_EINVAL, "bn_bwd_apply(bf16): relu mode 2 needs beta");
  PM_REQUIRE(!dres || (pm_vec8(dres) && pm_same_shape(dres, x)), PM_EINVAL, "bn_bwd_apply(bf16): dres shape mismatch");
  const pm_bf16 *pd = (const pm_bf16*)dy->ptr, *po = relu == 1 ? (const pm_bf16*)y->ptr : nullptr, *px = (const pm_bf16*)x->ptr;
  pm_bf16 *pdx = (pm_bf16*)dx->ptr, *pdr = dres ? (pm_bf16*)dres->ptr : nullptr;
  const long a = dy->pitch, b = relu == 1 ? y->pitch : 0, c = x->pitch, d = dx->pitch, e2 = dres ? dres->pitch : 0;
  const bool from_x = relu == 2;
  const int C = x->c;
  const bool dev_count = !(count > 0.f);
  const float host_inv_n = dev_count ? 0.f : 1.f / count;
  static const int fixed_on = getenv("PM_BN16_FIXED") ? atoi(getenv("PM_BN16_FIXED")) : 1;
  if (fixed_on && fixed_ok(C) && pm_pixels(x) > 0) {
    const long P = pm_pixels(x);
    const dim3 grid(fixed_grid(P, C));
    const int cg = C / V;
#define PM16_BAPPLY(M, D)                                                                                                                                            \
  hipLaunchKernelGGL((bn16_bwd_apply_fixed_kernel<M, D>), grid, dim3(256), 0, st, pd, a, po, b, px, c, mean, invstd, gamma, beta, sums, host_inv_n, dev_count ? 1 : 0, C, pdx, d, \
                     pdr, e2, P, cg)
    if (relu == 0) { if (pdr) PM16_BAPPLY(0, true); else PM16_BAPPLY(0, false); }
    else if (relu == 1) { if (pdr) PM16_BAPPLY(1, true); else PM16_BAPPLY(1, false); }
    else { if (pdr) PM16_BAPPLY(2, true); else PM16_BAPPLY(2, false); }
#undef PM16_BAPPLY
    return pm_check_launch("bn_bwd_apply(bf16, fixed groups)");
  }
  return ew16_launch(pm_pixels(x), C, st, "bn_bwd_apply(bf16)", [=] __device__(long p, int ch) {
    const float inv_n = dev_count ? 1.f / sums[2 * C] : host_inv_n;
    float g[V], v[V], mu[V], is[V], ga[V], s1[V], s2[V], r[V];
    pm_ld8(pd + p * a + ch, g);
    if (po) {
      float o[V];
      pm_ld8(po + p * b + ch, o);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    pm_ld8(px + p * c + ch, v);
    ld8f(mean + ch, mu), ld8f(invstd + ch, is), ld8f(gamma + ch, ga);
    if (from_x) {
      float be[V];
      ld8f(beta + ch, be);
#pragma unroll
      for (int e = 0; e < V; ++e) g[e] = bn_affine(v[e], mu[e], is[e], ga[e], be[e]) > 0.f ? g[e] : 0.f;
    }
    if (pdr) pm_st8(pdr + p * e2 + ch, g);
    ld8f(sums + ch, s1), ld8f(sums + C + ch, s2);
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = (g[e] - s1[e] * inv_n - (v[e] - mu[e]) * is[e] * (s2[e] * inv_n)) * (is[e] * ga[e]);
    pm_st8(pdx + p * d + ch, r);
  });
}

int pm16_add_n(const pm_tensor* const* xs, int n, const pm_tensor* o, hipStream_t st) {
  PM_REQUIRE(xs && o && n >= 2 && n <= 8, PM_EINVAL, "add_n(bf16): 2..8 operands");
  AddN16 a;
  for (int i = 0; i < n; ++i) {
    PM_REQUIRE(xs[i] && pm_same_shape(xs[i], o) && pm_vec8(xs[i]), PM_EINVAL, "add_n(bf16): operand %d: shape / 16-byte bf16 view mismatch", i);
    a.p[i] = (const pm_bf16*)xs[i]->ptr, a.pitch[i] = xs[i]->pitch;
  }
  PM_REQUIRE(pm_vec8(o), PM_EINVAL, "add_n(bf16): output must be a 16-byte bf16 view");
  pm_bf16* po = (pm_bf16*)o->ptr;
  const long c = o->pitch;
  return ew16_launch(pm_pixels(o), o->c, st, "add_n(bf16)", [=] __device__(long p, int ch) {
    float s[V];
    pm_ld8(a.p[0] + p * a.pitch[0] + ch, s);
    for (int i = 1; i < n; ++i) {
      float v[V];
      pm_ld8(a.p[i] + p * a.pitch[i] + ch, v);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += v[e];
    }
    pm_st8(po + p * c + ch, s);
  });
}

int pm16_copy(const pm_tensor* x, const pm_tensor* o, hipStream_t st) {
  PM_REQUIRE(x && o && pm_same_shape(x, o) && pm_vec8(x) && pm_vec8(o), PM_EINVAL, "copy(bf16): shape / 16-byte bf16 view mismatch");
  const pm_bf16* px = (const pm_bf16*)x->ptr;
  pm_bf16* po = (pm_bf16*)o->ptr;
  const long a = x->pitch, c = o->pitch;
  return ew16_launch(pm_pixels(x), x->c, st, "copy(bf16)",
                     [=] __device__(long p, int ch) { *reinterpret_cast<uint4*>(po + p * c + ch) = *reinterpret_cast<const uint4*>(px + p * a + ch); });
}

int pm16_maxpool_fwd(const pm_tensor* x, const pm_tensor* y, uint8_t* argmax, hipStream_t st) {
  PM_REQUIRE(pm_vec8(x) && pm_vec8(y) && (reinterpret_cast<uintptr_t>(argmax) & 7u) == 0, PM_EINVAL, "maxpool_fwd(bf16): 16-byte bf16 views, 8-byte aligned argmax");
  const long total = pm_pixels(y) * (y->c / V);
  hipLaunchKernelGGL(maxpool16_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const pm_bf16*)x->ptr, (long)x->pitch, x->h, x->w, (pm_bf16*)y->ptr,
                     (long)y->pitch, y->h, y->w, y->c, total, argmax);
  return pm_check_launch("maxpool_fwd(bf16)");
}
int pm16_maxpool_bwd(const pm_tensor* dy, const uint8_t* argmax, const pm_tensor* dx, hipStream_t st) {
  PM_REQUIRE(pm_vec8(dy) && pm_vec8(dx) && (reinterpret_cast<uintptr_t>(argmax) & 7u) == 0, PM_EINVAL, "maxpool_bwd(bf16): 16-byte bf16 views, 8-byte aligned argmax");
  const long total = pm_pixels(dx) * (dx->c / V);
  hipLaunchKernelGGL(maxpool16_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const pm_bf16*)dy->ptr, (long)dy->pitch, dy->h, dy->w, argmax,
                     (pm_bf16*)dx->ptr, (long)dx->pitch, dx->h, dx->w, dx->c, total);
  return pm_check_launch("maxpool_bwd(bf16)");
}

int pm16_gap_fwd(const pm_tensor* x, const pm_tensor* y, hipStream_t st) {
  PM_REQUIRE(pm_vec8(x) && pm_is_bf16(y) && y->ptr, PM_EINVAL, "global_avgpool_fwd(bf16): bad args");
  hipLaunchKernelGGL(gap16_fwd_kernel, dim3(x->n, pm_cdiv(x->c, 128)), dim3(256), 0, st, (const pm_bf16*)x->ptr, (long)x->pitch, (long)x->h * x->w, x->c,
                     (pm_bf16*)y->ptr, (long)y->pitch, 1.f / (float)((long)x->h * x->w), 0);
  return pm_check_launch("global_avgpool_fwd(bf16)");
}
int pm16_gap_bwd(const pm_tensor* dy, const pm_tensor* dx, int accumulate, hipStream_t st) {
  PM_REQUIRE(pm_is_bf16(dy) && dy->ptr && pm_vec8(dx) && pm_aligned16(dy->ptr) && dy->pitch % 8 == 0, PM_EINVAL, "global_avgpool_bwd(bf16): bad args");
  const pm_bf16* pd = (const pm_bf16*)dy->ptr;
  pm_bf16* px = (pm_bf16*)dx->ptr;
  const long dp = dy->pitch, xp = dx->pitch, HW = (long)dx->h * dx->w;
  const float inv = 1.f / (float)HW;
  return ew16_launch(pm_pixels(dx), dx->c, st, "global_avgpool_bwd(bf16)", [=] __device__(long p, int ch) {
    const long n = p / HW;
    float o[V], d[V];
    if (accumulate) pm_ld8(px + p * xp + ch, o);
    else {
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = 0.f;
    }
    pm_ld8(pd + n * dp + ch, d);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] += d[e] * inv;
    pm_st8(px + p * xp + ch, o);
  });
}

int pm16_resize_fwd(const pm_tensor* x, const pm_tensor* y, hipStream_t st) {
  PM_REQUIRE(pm_vec8(x) && pm_vec8(y), PM_EINVAL, "resize_fwd(bf16): 16-byte bf16 views");
  const long total = pm_pixels(y) * (y->c / V);
  hipLaunchKernelGGL(resize16_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const pm_bf16*)x->ptr, (long)x->pitch, x->h, x->w, (pm_bf16*)y->ptr,
                     (long)y->pitch, y->h, y->w, y->c, total, pm_ac_scale(x->h, y->h), pm_ac_scale(x->w, y->w));
  return pm_check_launch("resize_fwd(bf16)");
}
int pm16_resize_bwd(const pm_tensor* dy, const pm_tensor* dx, int accumulate, hipStream_t st) {
  PM_REQUIRE(pm_vec8(dy) && pm_vec8(dx), PM_EINVAL, "resize_bwd(bf16): 16-byte bf16 views");
  if (dx->h == 1 && dx->w == 1) {      // 1x1 source (ASPP image feature): every output pixel has weight 1 -> a plain column sum
    hipLaunchKernelGGL(gap16_fwd_kernel, dim3(dy->n, pm_cdiv(dy->c, 128)), dim3(256), 0, st, (const pm_bf16*)dy->ptr, (long)dy->pitch, (long)dy->h * dy->w, dy->c,
                       (pm_bf16*)dx->ptr, (long)dx->pitch, 1.f, accumulate);
    return pm_check_launch("resize_bwd(1x1, bf16)");
  }
  const long total = pm_pixels(dx) * (dx->c / V);
  hipLaunchKernelGGL(resize16_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const pm_bf16*)dy->ptr, (long)dy->pitch, dy->h, dy->w, (pm_bf16*)dx->ptr,
                     (long)dx->pitch, dx->h, dx->w, dx->c, total, pm_ac_scale(dx->h, dy->h), pm_ac_scale(dx->w, dy->w), accumulate);
  return pm_check_launch("resize_bwd(bf16)");
}
size_t pm16_resize_bwd_workspace(const pm_tensor* dy, const pm_tensor* dx) {
  if (!pm_vec8(dy) || !pm_vec8(dx) || dx->h < 2 || dx->w < 2 || dy->h < 2 * dx->h || dy->w < 2 * dx->w) return 0;
  return pm_align_up((size_t)dy->n * dy->h * dx->w * dy->c * sizeof(float), 256);
}
int pm16_resize_bwd_separable(const pm_tensor* dy, const pm_tensor* dx, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  const size_t need = pm16_resize_bwd_workspace(dy, dx);
  PM_REQUIRE(need > 0, PM_EUNSUPPORTED, "resize_bwd_separable(bf16): needs 16-byte channel vectors and an up-sampling ratio >= 2 in both directions");
  PM_REQUIRE(ws && ws_bytes >= need, PM_EWORKSPACE, "resize_bwd_separable(bf16): workspace %zu < %zu", ws_bytes, need);
  const float sy = pm_ac_scale(dx->h, dy->h), sx = pm_ac_scale(dx->w, dy->w);
  const long t1 = (long)dy->n * dy->h * dx->w * (dy->c / V), t2 = pm_pixels(dx) * (dx->c / V);
  hipLaunchKernelGGL(resize16_bwd_cols_kernel, dim3(grid_for(t1)), dim3(256), 0, st, (const pm_bf16*)dy->ptr, (long)dy->pitch, dy->h, dy->w, (float*)ws, dx->w, dy->c,
                     t1, sx);
  hipLaunchKernelGGL(resize16_bwd_rows_kernel, dim3(grid_for(t2)), dim3(256), 0, st, (const float*)ws, dy->h, (pm_bf16*)dx->ptr, (long)dx->pitch, dx->h, dx->w, dx->c,
                     t2, sy, accumulate);
  return pm_check_launch("resize_bwd_separable(bf16)");
}

int pm16_to_f32(const pm_bf16* x, long pitch, int C, long P, float* out, long out_pitch, hipStream_t st) {
  const long total = P * ((C + V - 1) / V);
  if (total == 0) return PM_OK;
  hipLaunchKernelGGL(to_f32_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, pitch, C, P, out, out_pitch);
  return pm_check_launch("bf16_to_f32");
}
int pm16_pad_rows(const pm_bf16* x, long pitch, int C, int Cp, long P, pm_bf16* out, hipStream_t st) {
  const long total = P * (Cp / V);
  if (total == 0) return PM_OK;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(total)), dim3(256), 0, st, x, pitch, C, Cp, P, out);
  return pm_check_launch("bf16_pad_rows");
}

// dtype conversion between two views of the same shape (the edges of the tier: the memory module and the losses stay fp32)
extern "C" int pm_cast(const pm_tensor* x, const pm_tensor* y, void* stream) {
  PM_REQUIRE(x && y && x->ptr && y->ptr && pm_same_shape(x, y), PM_EINVAL, "cast: shape mismatch");
  hipStream_t st = (hipStream_t)stream;
  const long P = pm_pixels(x);
  if (pm_is_bf16(x) && pm_is_f32(y)) return pm16_to_f32((const pm_bf16*)x->ptr, x->pitch, x->c, P, (float*)y->ptr, y->pitch, st);
  if (pm_is_f32(x) && pm_is_bf16(y)) {
    const float* px = (const float*)x->ptr;
    pm_bf16* py = (pm_bf16*)y->ptr;
    const long a = x->pitch, c = y->pitch;
    const int C = x->c;
    if (C % 8 == 0 && pm_vec_ok(x) && y->pitch % 8 == 0 && pm_aligned16(y->ptr))      // 16-byte rows on both sides (any pitches: channel slices of wider buffers)
      return ew16_launch(P, C, st, "cast", [=] __device__(long p, int ch) {
        float v[V];
        ld8f(px + p * a + ch, v);
        pm_st8(py + p * c + ch, v);
      });
    return pm_ew_launch(false, P, C, st, "cast", [=] __device__(long p, int ch) { py[p * c + ch] = pm_f32_to_bf16(px[p * a + ch]); });      // odd shapes: element by element
  }
  PM_REQUIRE(false, PM_EUNSUPPORTED, "cast: dtype %d -> %d", x->dtype, y->dtype);
  return PM_OK;
}
