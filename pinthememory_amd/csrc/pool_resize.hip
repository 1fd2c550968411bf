// K3/K5: max-pool 3x3 s2 p1, global average pool, bilinear resize (align_corners=True) -- NHWC fp32, HBM-bound.
// Replaces nn.MaxPool2d(3,2,1) (/root/reference/network/Resnet.py:432), nn.AdaptiveAvgPool2d(1) (deepv3plus.py:85)
// and mynn.Upsample (mynn.py:57-62). Backward passes are gathers (no atomics): deterministic.
#include "pm_common.h"

namespace {

// ---------------- max pool ------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, long xp, int H, int W, float* __restrict__ y, long yp, int Ho,
                                                          int Wo, int C, long total, uint8_t* __restrict__ arg) {
  constexpr int V = VEC ? 4 : 1;
  const int cg = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long op = i / cg;
    const int ch = (int)(i - op * cg) * V;
    const int ox = (int)(op % Wo), oy = (int)((op / Wo) % Ho), n = (int)(op / ((long)Wo * Ho));
    float best[V];
    uint8_t bi[V];
#pragma unroll
    for (int v = 0; v < V; ++v) best[v] = -INFINITY, bi[v] = 0;
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float* p = x + ((long)(n * H + iy) * W + ix) * xp + ch;
        float v[V];
        if constexpr (VEC) {
          const float4 q = PM_LD4(p);
          v[0] = q.x;
          if (V > 1) v[1 % V] = q.y, v[2 % V] = q.z, v[3 % V] = q.w;
        } else {
          v[0] = *p;
        }
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (first || v[e] > best[e] || v[e] != v[e]) best[e] = v[e], bi[e] = (uint8_t)(ky * 3 + kx);
        first = false;
      }
    }
    float* o = y + op * yp + ch;
    uint8_t* a = arg + op * C + ch;
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = best[e], a[e] = bi[e];
  }
}

// float4 variant (C % 4 == 0): thread = (input pixel, 4 channels); the argmax bytes of 4 channels come as one 32-bit load
__global__ __launch_bounds__(256) void maxpool_bwd_vec_kernel(const float* __restrict__ dy, long dp, int Ho, int Wo, const uint8_t* __restrict__ arg,
                                                              float* __restrict__ dx, long xp, int H, int W, int C, long total) {
  const int cg = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long ip = i / cg;
    const int ch = (int)(i - ip * cg) * 4;
    const int ix = (int)(ip % W), iy = (int)((ip / W) % H), n = (int)(ip / ((long)W * H));
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    const int oy_hi = min((iy + 1) >> 1, Ho - 1), ox_hi = min((ix + 1) >> 1, Wo - 1);
    for (int oy = iy >> 1; oy <= oy_hi; ++oy) {
      const int ky = iy + 1 - 2 * oy;
      if (ky < 0 || ky > 2) continue;
      for (int ox = ix >> 1; ox <= ox_hi; ++ox) {
        const int kx = ix + 1 - 2 * ox;
        if (kx < 0 || kx > 2) continue;
        const long op = (long)(n * Ho + oy) * Wo + ox;
        const unsigned a = *reinterpret_cast<const unsigned*>(arg + op * C + ch), want = (unsigned)(ky * 3 + kx);
        const float4 d = PM_LD4(dy + op * dp + ch);
        g.x += (a & 255u) == want ? d.x : 0.f;
        g.y += ((a >> 8) & 255u) == want ? d.y : 0.f;
        g.z += ((a >> 16) & 255u) == want ? d.z : 0.f;
        g.w += (a >> 24) == want ? d.w : 0.f;
      }
    }
    PM_ST4(dx + ip * xp + ch, g);
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, long dp, int Ho, int Wo, const uint8_t* __restrict__ arg,
                                                          float* __restrict__ dx, long xp, int H, int W, int C, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long ip = i / C;
    const int ch = (int)(i - ip * C);
    const int ix = (int)(ip % W), iy = (int)((ip / W) % H), n = (int)(ip / ((long)W * H));
    float g = 0.f;
    const int oy_hi = min((iy + 1) >> 1, Ho - 1), ox_hi = min((ix + 1) >> 1, Wo - 1);
    for (int oy = max(iy >> 1, 0); oy <= oy_hi; ++oy) {
      const int ky = iy + 1 - 2 * oy;
      if (ky < 0 || ky > 2) continue;
      for (int ox = max(ix >> 1, 0); ox <= ox_hi; ++ox) {
        const int kx = ix + 1 - 2 * ox;
        if (kx < 0 || kx > 2) continue;
        const long op = (long)(n * Ho + oy) * Wo + ox;
        if (arg[op * C + ch] == (uint8_t)(ky * 3 + kx)) g += dy[op * dp + ch];
      }
    }
    dx[ip * xp + ch] = g;
  }
}

// ---------------- global average pool ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, long xp, long HW, int C, float* __restrict__ y, long yp, float scale,
                                                      int accumulate) {
  __shared__ float sm[16][64];
  const int g = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + g * 4, n = blockIdx.x;
  float s[4] = {0, 0, 0, 0};
  if (c < C) {   // four independent partial sums: four 16 B loads in flight per thread (one block per CU has little else to hide latency with)
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* base = x + (long)n * HW * xp + c;
    long p = r;
    for (; p + 48 < HW; p += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = PM_LD4(base + (p + 16 * u) * xp);
        a[u].x += v.x, a[u].y += v.y, a[u].z += v.z, a[u].w += v.w;
      }
    }
    for (; p < HW; p += 16) {
      const float4 v = PM_LD4(base + p * xp);
      a[0].x += v.x, a[0].y += v.y, a[0].z += v.z, a[0].w += v.w;
    }
    s[0] = (a[0].x + a[1].x) + (a[2].x + a[3].x), s[1] = (a[0].y + a[1].y) + (a[2].y + a[3].y);
    s[2] = (a[0].z + a[1].z) + (a[2].z + a[3].z), s[3] = (a[0].w + a[1].w) + (a[2].w + a[3].w);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sm[r][g * 4 + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sm[i][threadIdx.x];
    const int ch = blockIdx.y * 64 + threadIdx.x;
    if (ch < C) y[(long)n * yp + ch] = (accumulate ? y[(long)n * yp + ch] : 0.f) + t * scale;
  }
}

// ---------------- bilinear, align_corners=True -------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const float* __restrict__ x, long xp, int h, int w, float* __restrict__ y, long yp, int H, int W,
                                                         int C, long total, float sy, float sx) {
  constexpr int V = VEC ? 4 : 1;
  const int cg = C / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long op = i / cg;
    const int ch = (int)(i - op * cg) * V;
    const int X = (int)(op % W), Y = (int)((op / W) % H), n = (int)(op / ((long)W * H));
    const pm_lerp ly = pm_ac_lerp(sy, Y, h), lx = pm_ac_lerp(sx, X, w);
    const float* r0 = x + ((long)(n * h + ly.i0) * w) * xp + ch;
    const float* r1 = x + ((long)(n * h + ly.i1) * w) * xp + ch;
    if constexpr (VEC) {
      const float4 a = PM_LD4(r0 + lx.i0 * xp), b = PM_LD4(r0 + lx.i1 * xp), c = PM_LD4(r1 + lx.i0 * xp), d = PM_LD4(r1 + lx.i1 * xp);
      float4 o;
      o.x = ly.w0 * (lx.w0 * a.x + lx.w1 * b.x) + ly.w1 * (lx.w0 * c.x + lx.w1 * d.x);
      o.y = ly.w0 * (lx.w0 * a.y + lx.w1 * b.y) + ly.w1 * (lx.w0 * c.y + lx.w1 * d.y);
      o.z = ly.w0 * (lx.w0 * a.z + lx.w1 * b.z) + ly.w1 * (lx.w0 * c.z + lx.w1 * d.z);
      o.w = ly.w0 * (lx.w0 * a.w + lx.w1 * b.w) + ly.w1 * (lx.w0 * c.w + lx.w1 * d.w);
      PM_ST4(y + op * yp + ch, o);
    } else {
      y[op * yp + ch] = ly.w0 * (lx.w0 * r0[lx.i0 * xp] + lx.w1 * r0[lx.i1 * xp]) + ly.w1 * (lx.w0 * r1[lx.i0 * xp] + lx.w1 * r1[lx.i1 * xp]);
    }
  }
}

// Output rows whose taps may touch input row `i`: conservative [lo, hi] from the inverse map, each verified exactly.
__device__ __forceinline__ void support(float scale, int i, int out, int& lo, int& hi) {
  if (scale <= 0.f) {
    lo = 0, hi = out - 1;
    return;
  }
  const float inv = 1.f / scale;
  lo = max(0, (int)floorf(((float)i - 1.f) * inv) - 1);
  hi = min(out - 1, (int)ceilf(((float)i + 1.f) * inv) + 1);
}
__device__ __forceinline__ float tap_weight(const pm_lerp& l, int i) { return (l.i0 == i ? l.w0 : 0.f) + (l.i1 == i ? l.w1 : 0.f); }

// Gather formulation: each input pixel sums the output pixels whose bilinear taps touch it. The per-row / per-column tap
// weights of the (small) support window are computed once into registers (MAXT x 2), so the inner loops are pure
// load + FMA; windows larger than MAXT (e.g. the 1x1 -> HxW broadcast of the ASPP image feature) take the generic loop.
template <bool VEC>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dy, long dp, int H, int W, float* __restrict__ dx, long xp, int h, int w,
                                                         int C, long total, float sy, float sx, int accumulate) {
  constexpr int V = VEC ? 4 : 1;
  constexpr int MAXT = 14;
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
    if (yhi - ylo < MAXT && xhi - xlo < MAXT) {
      float wyv[MAXT], wxv[MAXT];
#pragma unroll
      for (int k = 0; k < MAXT; ++k) {
        wyv[k] = ylo + k <= yhi ? tap_weight(pm_ac_lerp(sy, ylo + k, h), y) : 0.f;
        wxv[k] = xlo + k <= xhi ? tap_weight(pm_ac_lerp(sx, xlo + k, w), x) : 0.f;
      }
      const float* base = dy + ((long)(n * H + ylo) * W + xlo) * dp + ch;
#pragma unroll
      for (int k = 0; k < MAXT; ++k) {
        if (wyv[k] == 0.f) continue;
#pragma unroll
        for (int l = 0; l < MAXT; ++l) {
          if (wxv[l] == 0.f) continue;
          const float* p = base + ((long)k * W + l) * dp;
          const float ww = wyv[k] * wxv[l];
          if constexpr (VEC) {
            const float4 q = PM_LD4(p);
            g[0] += ww * q.x;
            if (V > 1) g[1 % V] += ww * q.y, g[2 % V] += ww * q.z, g[3 % V] += ww * q.w;
          } else {
            g[0] += ww * *p;
          }
        }
      }
    } else {
      for (int Y = ylo; Y <= yhi; ++Y) {
        const float wy = tap_weight(pm_ac_lerp(sy, Y, h), y);
        if (wy == 0.f) continue;
        for (int X = xlo; X <= xhi; ++X) {
          const float wx = tap_weight(pm_ac_lerp(sx, X, w), x);
          if (wx == 0.f) continue;
          const float* p = dy + ((long)(n * H + Y) * W + X) * dp + ch;
          const float ww = wy * wx;
          if constexpr (VEC) {
            const float4 q = PM_LD4(p);
            g[0] += ww * q.x;
            if (V > 1) g[1 % V] += ww * q.y, g[2 % V] += ww * q.z, g[3 % V] += ww * q.w;
          } else {
            g[0] += ww * *p;
          }
        }
      }
    }
    float* o = dx + ip * xp + ch;
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = accumulate ? o[e] + g[e] : g[e];
  }
}

// F.interpolate(..., mode='bilinear') with align_corners=False as ATen computes it: scale = in/out (float),
// src = scale * (dst + 0.5) - 0.5 clamped at 0, i0 = floor(src), i1 = i0 + (i0 < in-1), lambda = src - i0.
__device__ __forceinline__ pm_lerp hp_lerp(float scale, int dst, int in) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  pm_lerp r;
  r.i0 = i0;
  r.i1 = i0 + (i0 < in - 1 ? 1 : 0);
  float l1 = src - (float)i0;
  l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
  r.w1 = l1;
  r.w0 = 1.f - l1;
  return r;
}
__global__ __launch_bounds__(256) void resize_hp_fwd_kernel(const float* __restrict__ x, long xp, int h, int w, float* __restrict__ y, long yp, int H, int W,
                                                            int C, long total, float sy, float sx, int flip_w) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long op = i / C;
    const int ch = (int)(i - op * C);
    const int X = (int)(op % W), Y = (int)((op / W) % H), n = (int)(op / ((long)W * H));
    const pm_lerp ly = hp_lerp(sy, Y, h), lx = hp_lerp(sx, X, w);
    const float* r0 = x + ((long)(n * h + ly.i0) * w) * xp + ch;
    const float* r1 = x + ((long)(n * h + ly.i1) * w) * xp + ch;
    const float v = ly.w0 * (lx.w0 * r0[lx.i0 * xp] + lx.w1 * r0[lx.i1 * xp]) + ly.w1 * (lx.w0 * r1[lx.i0 * xp] + lx.w1 * r1[lx.i1 * xp]);
    const int Xo = flip_w ? W - 1 - X : X;   // un-flip on the way out (eval.py:330 flip_tensor2(y, -1))
    y[((long)(n * H + Y) * W + Xo) * yp + ch] = v;
  }
}
// one thread per pixel: softmax over C <= 32 classes in fp32 (as torch.softmax on fp32 logits), running mean in fp64
__global__ __launch_bounds__(256) void softmax_mean_kernel(const float* __restrict__ lg, long lp, long pixels, int C, double* __restrict__ buf, double inv_cnt) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long)gridDim.x * 256) {
    float v[32];
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) v[c] = lg[p * lp + c], mx = fmaxf(mx, v[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) v[c] = expf(v[c] - mx), se += v[c];
    for (int c = 0; c < C; ++c) {
      const double pr = (double)(v[c] / se), b = buf[p * C + c];
      buf[p * C + c] = b + (pr - b) * inv_cnt;
    }
  }
}
__global__ __launch_bounds__(256) void argmax_f64_kernel(const double* __restrict__ buf, long pixels, int C, int64_t* __restrict__ cls, double* __restrict__ prob) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long)gridDim.x * 256) {
    double best = buf[p * C];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      const double v = buf[p * C + c];
      if (v > best) best = v, bi = c;             // first maximum wins, as torch.max
    }
    cls[p] = bi;
    if (prob) prob[p] = best;
  }
}

inline int grid_for(long work) { return (int)std::min<long>((work + 255) / 256, 256 * 32); }

// Separable backward of the align_corners bilinear resize for up-sampling ratios >= 2: the transposed operator factorises into a
// column pass and a row pass, so the large gradient dy is read ONCE (the gather above re-reads every hi-res pixel from the ~4 low-res
// pixels whose support covers it: 873 MB of traffic for the 302 MB decoder gradient).
//   pass 1  T[n, Y, x, c] = sum_X wx(X, x) * dy[n, Y, X, c]        (thread = (n, Y, x, 4 channels), <= 14 taps)
//   pass 2  dx[n, y, x, c] (+)= sum_Y wy(Y, y) * T[n, Y, x, c]
// Both sums run in ascending index order: deterministic.
__global__ __launch_bounds__(256) void resize_bwd_cols_kernel(const float* __restrict__ dy, long dp, int H, int W, float* __restrict__ T, int w, int C,
                                                             long total, float sx) {
  const int cg = C / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long ip = i / cg;
    const int ch = (int)(i - ip * cg) * 4;
    const int x = (int)(ip % w);
    const long row = ip / w;   // n * H + Y
    int xlo, xhi;
    support(sx, x, W, xlo, xhi);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* base = dy + (row * W) * dp + ch;
    for (int X = xlo; X <= xhi; ++X) {
      const float wx = tap_weight(pm_ac_lerp(sx, X, w), x);
      if (wx == 0.f) continue;
      const float4 q = PM_LD4(base + (long)X * dp);
      g.x += wx * q.x, g.y += wx * q.y, g.z += wx * q.z, g.w += wx * q.w;
    }
    PM_ST4(T + ip * C + ch, g);
  }
}
__global__ __launch_bounds__(256) void resize_bwd_rows_kernel(const float* __restrict__ T, int H, float* __restrict__ dx, long xp, int h, int w, int C,
                                                             long total, float sy, int accumulate) {
  const int cg = C / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long ip = i / cg;
    const int ch = (int)(i - ip * cg) * 4;
    const int x = (int)(ip % w), y = (int)((ip / w) % h), n = (int)(ip / ((long)w * h));
    int ylo, yhi;
    support(sy, y, H, ylo, yhi);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int Y = ylo; Y <= yhi; ++Y) {
      const float wy = tap_weight(pm_ac_lerp(sy, Y, h), y);
      if (wy == 0.f) continue;
      const float4 q = PM_LD4(T + (((long)n * H + Y) * w + x) * C + ch);
      g.x += wy * q.x, g.y += wy * q.y, g.z += wy * q.z, g.w += wy * q.w;
    }
    float* o = dx + ip * xp + ch;
    if (accumulate) {
      const float4 q = PM_LD4(o);
      g.x += q.x, g.y += q.y, g.z += q.z, g.w += q.w;
    }
    PM_ST4(o, g);
  }
}

}  // namespace

extern "C" int pm_maxpool3x3s2_fwd(const pm_tensor* x, const pm_tensor* y, uint8_t* argmax, void* stream) {
  PM_REQUIRE(x && y && argmax && x->ptr && y->ptr, PM_EINVAL, "maxpool_fwd: null");
  PM_REQUIRE(y->h == (x->h + 2 - 3) / 2 + 1 && y->w == (x->w + 2 - 3) / 2 + 1 && x->n == y->n && x->c == y->c, PM_EINVAL, "maxpool_fwd: shape mismatch");
  if (pm_is_bf16(x) && pm_is_bf16(y)) return pm16_maxpool_fwd(x, y, argmax, (hipStream_t)stream);
  PM_REQUIRE_F32(x, "maxpool_fwd"); PM_REQUIRE_F32(y, "maxpool_fwd");
  const bool v = pm_vec4(x) && pm_vec4(y);
  const long total = pm_pixels(y) * (v ? y->c / 4 : y->c);
  if (v)
    hipLaunchKernelGGL(maxpool_fwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, (long)x->pitch, x->h, x->w,
                       (float*)y->ptr, (long)y->pitch, y->h, y->w, y->c, total, argmax);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, (long)x->pitch, x->h, x->w,
                       (float*)y->ptr, (long)y->pitch, y->h, y->w, y->c, total, argmax);
  return pm_check_launch("maxpool_fwd");
}

extern "C" int pm_maxpool3x3s2_bwd(const pm_tensor* dy, const uint8_t* argmax, const pm_tensor* dx, void* stream) {
  PM_REQUIRE(dy && dx && argmax && dy->ptr && dx->ptr && dy->n == dx->n && dy->c == dx->c, PM_EINVAL, "maxpool_bwd: bad args");
  if (pm_is_bf16(dy) && pm_is_bf16(dx)) return pm16_maxpool_bwd(dy, argmax, dx, (hipStream_t)stream);
  PM_REQUIRE_F32(dy, "maxpool_bwd"); PM_REQUIRE_F32(dx, "maxpool_bwd");
  const long total = pm_pixels(dx) * dx->c;
  if (pm_vec4(dy) && pm_vec4(dx) && (reinterpret_cast<uintptr_t>(argmax) & 3u) == 0)
    hipLaunchKernelGGL(maxpool_bwd_vec_kernel, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)dy->ptr, (long)dy->pitch, dy->h, dy->w,
                       argmax, (float*)dx->ptr, (long)dx->pitch, dx->h, dx->w, dx->c, total / 4);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy->ptr, (long)dy->pitch, dy->h, dy->w, argmax,
                       (float*)dx->ptr, (long)dx->pitch, dx->h, dx->w, dx->c, total);
  return pm_check_launch("maxpool_bwd");
}

extern "C" int pm_global_avgpool_fwd(const pm_tensor* x, const pm_tensor* y, void* stream) {
  PM_REQUIRE(x && y && y->ptr && y->h == 1 && y->w == 1 && y->n == x->n && y->c == x->c, PM_EINVAL, "global_avgpool_fwd: bad args");
  if (pm_is_bf16(x) && pm_is_bf16(y)) return pm16_gap_fwd(x, y, (hipStream_t)stream);
  PM_REQUIRE_F32(x, "global_avgpool_fwd"); PM_REQUIRE_F32(y, "global_avgpool_fwd");
  PM_REQUIRE(pm_vec4(x), PM_EINVAL, "global_avgpool_fwd: bad args");
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(x->n, pm_cdiv(x->c, 64)), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, (long)x->pitch,
                     (long)x->h * x->w, x->c, (float*)y->ptr, (long)y->pitch, 1.f / (float)((long)x->h * x->w), 0);
  return pm_check_launch("global_avgpool_fwd");
}

extern "C" int pm_global_avgpool_bwd(const pm_tensor* dy, const pm_tensor* dx, int accumulate, void* stream) {
  PM_REQUIRE(dy && dx && dy->ptr && dy->h == 1 && dy->w == 1 && dy->n == dx->n && dy->c == dx->c, PM_EINVAL, "global_avgpool_bwd: bad args");
  if (pm_is_bf16(dy) && pm_is_bf16(dx)) return pm16_gap_bwd(dy, dx, accumulate, (hipStream_t)stream);
  PM_REQUIRE_F32(dy, "global_avgpool_bwd"); PM_REQUIRE_F32(dx, "global_avgpool_bwd");
  PM_REQUIRE(pm_vec4(dx), PM_EINVAL, "global_avgpool_bwd: bad args");
  const float* pd = (const float*)dy->ptr;
  float* px = (float*)dx->ptr;
  const long dp = dy->pitch, xp = dx->pitch, HW = (long)dx->h * dx->w;
  const float inv = 1.f / (float)HW;
  return pm_ew_launch(true, pm_pixels(dx), dx->c, (hipStream_t)stream, "global_avgpool_bwd", [=] __device__(long p, int ch) {
    const long n = p / HW;
    float4 o = accumulate ? PM_LD4(px + p * xp + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x += pd[n * dp + ch] * inv, o.y += pd[n * dp + ch + 1] * inv, o.z += pd[n * dp + ch + 2] * inv, o.w += pd[n * dp + ch + 3] * inv;
    PM_ST4(px + p * xp + ch, o);
  });
}

extern "C" int pm_resize_bilinear_fwd(const pm_tensor* x, const pm_tensor* y, void* stream) {
  PM_REQUIRE(x && y && x->ptr && y->ptr && x->n == y->n && x->c == y->c, PM_EINVAL, "resize_fwd: bad args");
  if (pm_is_bf16(x) && pm_is_bf16(y)) return pm16_resize_fwd(x, y, (hipStream_t)stream);
  PM_REQUIRE_F32(x, "resize_fwd"); PM_REQUIRE_F32(y, "resize_fwd");
  // channel counts that are not a multiple of 4 (the 19 class logits) on pitch-padded views: run the float4 path over the padded
  // width -- the pad lanes of the input are zero (kernels.new), so the pad lanes of the output are written as zero
  const int cv = (x->c + 3) & ~3;
  const bool v = pm_vec_ok(x) && pm_vec_ok(y) && (x->c % 4 == 0 || (x->pitch == cv && y->pitch == cv));   // padded lanes only when they are the views' own
  const int ceff = v ? cv : y->c;
  const long total = pm_pixels(y) * (v ? ceff / 4 : ceff);
  const float sy = pm_ac_scale(x->h, y->h), sx = pm_ac_scale(x->w, y->w);
  if (v)
    hipLaunchKernelGGL(resize_fwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, (long)x->pitch, x->h, x->w,
                       (float*)y->ptr, (long)y->pitch, y->h, y->w, ceff, total, sy, sx);
  else
    hipLaunchKernelGGL(resize_fwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, (long)x->pitch, x->h, x->w,
                       (float*)y->ptr, (long)y->pitch, y->h, y->w, y->c, total, sy, sx);
  return pm_check_launch("resize_fwd");
}

extern "C" int pm_resize_bilinear_bwd(const pm_tensor* dy, const pm_tensor* dx, int accumulate, void* stream) {
  PM_REQUIRE(dy && dx && dy->ptr && dx->ptr && dy->n == dx->n && dy->c == dx->c, PM_EINVAL, "resize_bwd: bad args");
  if (pm_is_bf16(dy) && pm_is_bf16(dx)) return pm16_resize_bwd(dy, dx, accumulate, (hipStream_t)stream);
  PM_REQUIRE_F32(dy, "resize_bwd"); PM_REQUIRE_F32(dx, "resize_bwd");
  const bool v = pm_vec4(dy) && pm_vec4(dx);
  if (dx->h == 1 && dx->w == 1 && v) {  // 1x1 source (ASPP image feature): every output pixel has weight 1 -> a plain column sum
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(dy->n, pm_cdiv(dy->c, 64)), dim3(256), 0, (hipStream_t)stream, (const float*)dy->ptr, (long)dy->pitch,
                       (long)dy->h * dy->w, dy->c, (float*)dx->ptr, (long)dx->pitch, 1.f, accumulate);
    return pm_check_launch("resize_bwd(1x1)");
  }
  const long total = pm_pixels(dx) * (v ? dx->c / 4 : dx->c);
  const float sy = pm_ac_scale(dx->h, dy->h), sx = pm_ac_scale(dx->w, dy->w);
  if (v)
    hipLaunchKernelGGL(resize_bwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy->ptr, (long)dy->pitch, dy->h, dy->w,
                       (float*)dx->ptr, (long)dx->pitch, dx->h, dx->w, dx->c, total, sy, sx, accumulate);
  else
    hipLaunchKernelGGL(resize_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy->ptr, (long)dy->pitch, dy->h, dy->w,
                       (float*)dx->ptr, (long)dx->pitch, dx->h, dx->w, dx->c, total, sy, sx, accumulate);
  return pm_check_launch("resize_bwd");
}

// workspace of the separable backward (0: shape not eligible -> use pm_resize_bilinear_bwd)
extern "C" size_t pm_resize_bilinear_bwd_workspace(const pm_tensor* dy, const pm_tensor* dx) {
  if (dy && dx && pm_is_bf16(dy) && pm_is_bf16(dx)) return pm16_resize_bwd_workspace(dy, dx);
  if (!dy || !dx || !pm_is_f32(dy) || !pm_is_f32(dx) || !pm_vec4(dy) || !pm_vec4(dx) || dx->h < 2 || dx->w < 2 || dy->h < 2 * dx->h || dy->w < 2 * dx->w) return 0;
  return pm_align_up((size_t)dy->n * dy->h * dx->w * dy->c * sizeof(float), 256);
}
extern "C" int pm_resize_bilinear_bwd_separable(const pm_tensor* dy, const pm_tensor* dx, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE(dy && dx && dy->ptr && dx->ptr && dy->n == dx->n && dy->c == dx->c, PM_EINVAL, "resize_bwd_separable: bad args");
  if (pm_is_bf16(dy) && pm_is_bf16(dx)) return pm16_resize_bwd_separable(dy, dx, accumulate, ws, ws_bytes, (hipStream_t)stream);
  PM_REQUIRE_F32(dy, "resize_bwd_separable"); PM_REQUIRE_F32(dx, "resize_bwd_separable");
  const size_t need = pm_resize_bilinear_bwd_workspace(dy, dx);
  PM_REQUIRE(need > 0, PM_EUNSUPPORTED, "resize_bwd_separable: needs 16-byte channel vectors and an up-sampling ratio >= 2 in both directions");
  PM_REQUIRE(ws && ws_bytes >= need, PM_EWORKSPACE, "resize_bwd_separable: workspace %zu < %zu", ws_bytes, need);
  const float sy = pm_ac_scale(dx->h, dy->h), sx = pm_ac_scale(dx->w, dy->w);
  const long t1 = (long)dy->n * dy->h * dx->w * (dy->c / 4), t2 = pm_pixels(dx) * (dx->c / 4);
  hipLaunchKernelGGL(resize_bwd_cols_kernel, dim3(grid_for(t1)), dim3(256), 0, (hipStream_t)stream, (const float*)dy->ptr, (long)dy->pitch, dy->h, dy->w, (float*)ws,
                     dx->w, dy->c, t1, sx);
  hipLaunchKernelGGL(resize_bwd_rows_kernel, dim3(grid_for(t2)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, dy->h, (float*)dx->ptr, (long)dx->pitch, dx->h,
                     dx->w, dx->c, t2, sy, accumulate);
  return pm_check_launch("resize_bwd_separable");
}

extern "C" int pm_resize_bilinear_hp_fwd(const pm_tensor* x, const pm_tensor* y, int flip_w, void* stream) {
  PM_REQUIRE(x && y && x->ptr && y->ptr && x->n == y->n && x->c == y->c, PM_EINVAL, "resize_hp_fwd: bad args");
  PM_REQUIRE_F32(x, "resize_hp_fwd"); PM_REQUIRE_F32(y, "resize_hp_fwd");
  const long total = pm_pixels(y) * y->c;
  hipLaunchKernelGGL(resize_hp_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, (long)x->pitch, x->h, x->w,
                     (float*)y->ptr, (long)y->pitch, y->h, y->w, y->c, total, (float)x->h / (float)y->h, (float)x->w / (float)y->w, flip_w);
  return pm_check_launch("resize_hp_fwd");
}
extern "C" int pm_softmax_mean_update(const pm_tensor* logits, double* buffer, int counter, void* stream) {
  PM_REQUIRE(logits && logits->ptr && buffer && counter >= 1 && logits->c >= 1 && logits->c <= 32, PM_EINVAL, "softmax_mean_update: bad args");
  PM_REQUIRE_F32(logits, "softmax_mean_update");
  const long pixels = pm_pixels(logits);
  hipLaunchKernelGGL(softmax_mean_kernel, dim3(grid_for(pixels)), dim3(256), 0, (hipStream_t)stream, (const float*)logits->ptr, (long)logits->pitch, pixels,
                     logits->c, buffer, 1.0 / (double)counter);
  return pm_check_launch("softmax_mean_update");
}
namespace {
// Sliding-window stitching (eval.py:210-274: add the tiles' logits, divide by the per-pixel tile count, un-flip; the reference does it per
// class in numpy threads on the host). One thread per output pixel: float64 sum over the covering tiles in tile order, / count, written to
// the class-major float64 accumulator at the (un-flipped) column; `accumulate` adds to what an earlier flip / scale left there.
constexpr int STITCH_MAX_TILES = 64;
struct StitchTiles {
  int n;
  int x1[STITCH_MAX_TILES], y1[STITCH_MAX_TILES], x2[STITCH_MAX_TILES], y2[STITCH_MAX_TILES];
};
__global__ __launch_bounds__(256) void sliding_stitch_kernel(const float* __restrict__ lg, long pitch, int th, int tw, int C, const StitchTiles tiles, int H, int W,
                                                             int flip, double* __restrict__ acc, int accumulate) {
  const long total = (long)H * W;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    const int xo = flip ? W - 1 - x : x;
    double cnt = 0.0;
    for (int t = 0; t < tiles.n; ++t)
      if (x >= tiles.x1[t] && x < tiles.x2[t] && y >= tiles.y1[t] && y < tiles.y2[t]) cnt += 1.0;
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int t = 0; t < tiles.n; ++t)
        if (x >= tiles.x1[t] && x < tiles.x2[t] && y >= tiles.y1[t] && y < tiles.y2[t])
          s += (double)lg[(((long)t * th + (y - tiles.y1[t])) * tw + (x - tiles.x1[t])) * pitch + c];
      const double v = s / cnt;          // uncovered pixels: 0 / 0 = NaN, as the host formulation gives
      double* dst = acc + ((long)c * H + y) * W + xo;
      *dst = accumulate ? *dst + v : v;
    }
  }
}

}  // namespace
extern "C" int pm_sliding_stitch(const pm_tensor* logits, const int32_t* tiles_xyxy, int ntiles, int H, int W, int flip_w, double* acc, int accumulate,
                                 void* stream) {
  PM_REQUIRE(logits && logits->ptr && tiles_xyxy && acc && ntiles >= 1 && ntiles <= STITCH_MAX_TILES && logits->n == ntiles, PM_EINVAL,
             "sliding_stitch: bad args (1..%d tiles, logits->n == ntiles)", STITCH_MAX_TILES);
  PM_REQUIRE_F32(logits, "sliding_stitch");
  StitchTiles t;
  t.n = ntiles;
  for (int i = 0; i < ntiles; ++i) {
    t.x1[i] = tiles_xyxy[4 * i], t.y1[i] = tiles_xyxy[4 * i + 1], t.x2[i] = tiles_xyxy[4 * i + 2], t.y2[i] = tiles_xyxy[4 * i + 3];
    PM_REQUIRE(t.x2[i] - t.x1[i] == logits->w && t.y2[i] - t.y1[i] == logits->h && t.x1[i] >= 0 && t.y1[i] >= 0 && t.x2[i] <= W && t.y2[i] <= H, PM_EINVAL,
               "sliding_stitch: tile %d does not match the logits' %dx%d", i, logits->h, logits->w);
  }
  hipLaunchKernelGGL(sliding_stitch_kernel, dim3(grid_for((long)H * W)), dim3(256), 0, (hipStream_t)stream, (const float*)logits->ptr, (long)logits->pitch, logits->h,
                     logits->w, logits->c, t, H, W, flip_w, acc, accumulate);
  return pm_check_launch("sliding_stitch");
}
extern "C" int pm_argmax_f64(const double* buffer, int n, int h, int w, int c, int64_t* out_cls, double* out_prob, void* stream) {
  PM_REQUIRE(buffer && out_cls && c >= 1, PM_EINVAL, "argmax_f64: bad args");
  const long pixels = (long)n * h * w;
  hipLaunchKernelGGL(argmax_f64_kernel, dim3(grid_for(pixels)), dim3(256), 0, (hipStream_t)stream, buffer, pixels, c, out_cls, out_prob);
  return pm_check_launch("argmax_f64");
}
