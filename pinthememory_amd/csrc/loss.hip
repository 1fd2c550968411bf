// K6: fused bilinear-upsample(align_corners=True) + CrossEntropy(ignore_index=255, reduction='mean').
// Replaces  Upsample(dec2, x_size) -> criterion  (/root/reference/network/deepv3plus.py:575-578), the aux loss
// (:589-595, same-size case) and the read loss  F.interpolate(score/T) -> celoss  (/root/reference/network/memory.py:173-176)
// WITHOUT materialising the [B,19,H,W] upsampled logits (358 MB at bs=8, 768^2): the low-res logits stay L2-resident
// and each hi-res pixel interpolates its 19 classes in registers. HBM-bound on the int64 labels (37.7 MB).
// Forward: block partial (sum, count) -> fixed-order final reduce (deterministic).
// Backward: GATHER per low-res pixel (one wave each, lanes over the hi-res support) -- no atomics, deterministic.
#include "pm_common.h"

namespace {

constexpr int MAXC = 32;

struct CEGeom {
  const float* logits;
  long lp;           // pitch
  int n, h, w, C;
  const int64_t* labels;
  int H, W;
  float sy, sx, inv_temp;
};

// interpolated logits of hi-res pixel (b, Y, X) into v[0..C)
template <int C_>
__device__ __forceinline__ void interp_logits(const CEGeom& g, int b, const pm_lerp& ly, const pm_lerp& lx, float* v) {
  const float* r0 = g.logits + ((long)(b * g.h + ly.i0) * g.w) * g.lp;
  const float* r1 = g.logits + ((long)(b * g.h + ly.i1) * g.w) * g.lp;
  const float* p00 = r0 + (long)lx.i0 * g.lp;
  const float* p01 = r0 + (long)lx.i1 * g.lp;
  const float* p10 = r1 + (long)lx.i0 * g.lp;
  const float* p11 = r1 + (long)lx.i1 * g.lp;
  const int C = C_ > 0 ? C_ : g.C;
#pragma unroll
  for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c) {
    if (c < C) {
      const float a = p00[c] * g.inv_temp, bq = p01[c] * g.inv_temp, cq = p10[c] * g.inv_temp, d = p11[c] * g.inv_temp;
      v[c] = ly.w0 * (lx.w0 * a + lx.w1 * bq) + ly.w1 * (lx.w0 * cq + lx.w1 * d);
    }
  }
}

template <int C_>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const CEGeom g, float* __restrict__ part) {
  const int C = C_ > 0 ? C_ : g.C;
  const long total = (long)g.n * g.H * g.W;
  float lsum = 0.f, lcnt = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int64_t lab = g.labels[i];
    if (lab == 255) continue;
    const int X = (int)(i % g.W), Y = (int)((i / g.W) % g.H), b = (int)(i / ((long)g.W * g.H));
    const pm_lerp ly = pm_ac_lerp(g.sy, Y, g.h), lx = pm_ac_lerp(g.sx, X, g.w);
    float v[C_ > 0 ? C_ : MAXC];
    interp_logits<C_>(g, b, ly, lx, v);
    float mx = -INFINITY, vl = 0.f;
#pragma unroll
    for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c)
      if (c < C) {
        mx = fmaxf(mx, v[c]);
        if (c == (int)lab) vl = v[c];
      }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c)
      if (c < C) se += expf(v[c] - mx);
    lsum += (mx + logf(se)) - vl;
    lcnt += 1.f;
  }
  __shared__ float sm[2][4];
  lsum = pm_wave_sum(lsum);
  lcnt = pm_wave_sum(lcnt);
  if ((threadIdx.x & 63) == 0) sm[0][threadIdx.x >> 6] = lsum, sm[1][threadIdx.x >> 6] = lcnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
    part[blockIdx.x * 2 + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
  }
}

__global__ void ce_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ double s[2][64];
  double a = 0.0, c = 0.0;
  for (int b = threadIdx.x; b < nb; b += 64) a += (double)part[b * 2], c += (double)part[b * 2 + 1];
  s[0][threadIdx.x] = a, s[1][threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tc = 0.0;
    for (int i = 0; i < 64; ++i) ta += s[0][i], tc += s[1][i];
    out[0] = (float)(ta / tc);  // 0/0 -> NaN like torch when every pixel is ignored
    out[1] = (float)tc;
  }
}

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

// one wave per low-res pixel; lanes sweep the hi-res support window; 19 partial sums per lane, wave-reduced at the end
template <int C_>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const CEGeom g, const float* __restrict__ loss_out, const float* __restrict__ gscale,
                                                     float* __restrict__ dl, long dlp, int accumulate) {
  const int C = C_ > 0 ? C_ : g.C;
  const int lane = threadIdx.x & 63;
  const long npix = (long)g.n * g.h * g.w;
  const float cnt = loss_out[1];
  const float gs = (gscale ? gscale[0] : 1.f) * g.inv_temp / cnt;
  for (long lp = (long)blockIdx.x * 4 + (threadIdx.x >> 6); lp < npix; lp += (long)gridDim.x * 4) {
    const int x = (int)(lp % g.w), y = (int)((lp / g.w) % g.h), b = (int)(lp / ((long)g.w * g.h));
    int ylo, yhi, xlo, xhi;
    support(g.sy, y, g.H, ylo, yhi);
    support(g.sx, x, g.W, xlo, xhi);
    const int wx = xhi - xlo + 1, wtot = (yhi - ylo + 1) * wx;
    float acc[C_ > 0 ? C_ : MAXC];
#pragma unroll
    for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c) acc[c] = 0.f;
    for (int j = lane; j < wtot; j += 64) {
      const int Y = ylo + j / wx, X = xlo + j % wx;
      const pm_lerp ly = pm_ac_lerp(g.sy, Y, g.h), lx = pm_ac_lerp(g.sx, X, g.w);
      const float wgt = tap_weight(ly, y) * tap_weight(lx, x);
      if (wgt == 0.f) continue;
      const int64_t lab = g.labels[((long)b * g.H + Y) * g.W + X];
      if (lab == 255) continue;
      float v[C_ > 0 ? C_ : MAXC];
      interp_logits<C_>(g, b, ly, lx, v);
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c)
        if (c < C) mx = fmaxf(mx, v[c]);
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c)
        if (c < C) v[c] = expf(v[c] - mx), se += v[c];
      const float inv = wgt / se;
#pragma unroll
      for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c)
        if (c < C) acc[c] += v[c] * inv - (c == (int)lab ? wgt : 0.f);
    }
#pragma unroll
    for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c)
      if (c < C) {
        const float s = pm_wave_sum(acc[c]) * gs;
        if (lane == 0) dl[lp * dlp + c] = accumulate ? dl[lp * dlp + c] + s : s;
      }
  }
}

inline int fwd_blocks(long total) { return (int)std::min<long>((total + 255) / 256, 4096); }

int fill(CEGeom& g, const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, const char* who) {
  PM_REQUIRE(logits && logits->ptr && labels && H > 0 && W > 0, PM_EINVAL, "%s: null/empty", who);
  PM_REQUIRE(logits->c >= 1 && logits->c <= MAXC, PM_EUNSUPPORTED, "%s: classes %d > %d", who, logits->c, MAXC);
  g.logits = (const float*)logits->ptr, g.lp = logits->pitch, g.n = logits->n, g.h = logits->h, g.w = logits->w, g.C = logits->c;
  g.labels = labels, g.H = H, g.W = W;
  g.sy = pm_ac_scale(logits->h, H), g.sx = pm_ac_scale(logits->w, W), g.inv_temp = inv_temp;
  return PM_OK;
}

}  // namespace

extern "C" size_t pm_upsample_ce_workspace(int n, int H, int W) { return pm_align_up((size_t)fwd_blocks((long)n * H * W) * 2 * sizeof(float), 256); }

extern "C" int pm_upsample_ce_fwd(const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, float* loss_out, void* ws, size_t ws_bytes,
                                  void* stream) {
  CEGeom g;
  if (int e = fill(g, logits, inv_temp, labels, H, W, "upsample_ce_fwd")) return e;
  PM_REQUIRE(loss_out && ws && ws_bytes >= pm_upsample_ce_workspace(g.n, H, W), PM_EWORKSPACE, "upsample_ce_fwd: workspace too small");
  const int nb = fwd_blocks((long)g.n * H * W);
  hipStream_t st = (hipStream_t)stream;
  if (g.C == 19) hipLaunchKernelGGL(ce_fwd_kernel<19>, dim3(nb), dim3(256), 0, st, g, (float*)ws);
  else hipLaunchKernelGGL(ce_fwd_kernel<0>, dim3(nb), dim3(256), 0, st, g, (float*)ws);
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, nb, loss_out);
  return pm_check_launch("upsample_ce_fwd");
}

extern "C" int pm_upsample_ce_bwd(const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, const float* loss_out, const float* gscale,
                                  const pm_tensor* dlogits, void* stream) {
  CEGeom g;
  if (int e = fill(g, logits, inv_temp, labels, H, W, "upsample_ce_bwd")) return e;
  PM_REQUIRE(loss_out && dlogits && dlogits->ptr && pm_same_shape(logits, dlogits), PM_EINVAL, "upsample_ce_bwd: bad args");
  const long npix = (long)g.n * g.h * g.w;
  const int nb = (int)std::min<long>((npix + 3) / 4, 256 * 64);
  hipStream_t st = (hipStream_t)stream;
  if (g.C == 19) hipLaunchKernelGGL(ce_bwd_kernel<19>, dim3(nb), dim3(256), 0, st, g, loss_out, gscale, (float*)dlogits->ptr, (long)dlogits->pitch, 0);
  else hipLaunchKernelGGL(ce_bwd_kernel<0>, dim3(nb), dim3(256), 0, st, g, loss_out, gscale, (float*)dlogits->ptr, (long)dlogits->pitch, 0);
  return pm_check_launch("upsample_ce_bwd");
}
