// K4: BatchNorm2d over NHWC fp32 (train statistics, apply + ReLU + residual, backward), HBM-bound.
// Replaces mynn.Norm2d (/root/reference/network/mynn.py:8-14) == nn.BatchNorm2d / SyncBatchNorm everywhere it is
// used (Resnet.py:146-151,405,456; deepv3plus.py:73,79,88,399,405,410,413,421; memory.py:76,105).
// Statistics are shifted sums (shift = first pixel of each channel) -> (mean, M2, count), combined in double in a
// fixed order: one pass over the activation, no catastrophic cancellation, deterministic, mergeable across ranks.
#include "pm_common.h"

namespace {

constexpr int CB = 64;    // channels per block
constexpr int RL = 16;    // row lanes per block (256 threads = 16 float4 groups x 16 rows)

// y = (x - mean) * invstd * gamma + beta, evaluated the same way in the forward pass and wherever the backward pass rebuilds the
// ReLU mask from x (two explicit FMAs: bit-identical in both places whatever the compiler would contract).
__device__ __forceinline__ float bn_affine(float v, float mu, float is, float ga, float be) {
  const float s = is * ga;
  return fmaf(v, s, fmaf(-mu, s, be));
}

inline int chunk_rows(long P, int C) {  // pixels per block: >= 2048 blocks over (pixel chunks x 64-channel groups), >= 64 rows each
  const long colblocks = std::max<long>(1, (C + CB - 1) / CB);
  const long want = std::max<long>(512, 2048 / colblocks);   // narrow tensors (64 channels) need more pixel chunks to fill the GPU
  long r = (P + want - 1) / want;
  r = std::max<long>(r, 64);
  return (int)((r + RL - 1) / RL * RL);
}

// partial[blk][c][2] : sum(x - K[c]), sum((x - K[c])^2) over the block's pixel chunk
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, long pitch, long P, int C, int rows, float* __restrict__ part) {
  __shared__ float sm[RL][CB][2];
  const int g = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = blockIdx.y * CB + g * 4;
  const long p0 = (long)blockIdx.x * rows, p1 = min(P, p0 + rows);
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (c < C) {
    const float4 k = PM_LD4(x + c);
    for (long p = p0 + r; p < p1; p += RL) {
      const float4 v = PM_LD4(x + p * pitch + c);
      const float d0 = v.x - k.x, d1 = v.y - k.y, d2 = v.z - k.z, d3 = v.w - k.w;
      s1[0] += d0, s1[1] += d1, s1[2] += d2, s1[3] += d3;
      s2[0] += d0 * d0, s2[1] += d1 * d1, s2[2] += d2 * d2, s2[3] += d3 * d3;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sm[r][g * 4 + j][0] = s1[j], sm[r][g * 4 + j][1] = s2[j];
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

// Second stage: 4 channels x 64 lanes per block; each lane sums a strided subset of the block partials in double, the 64
// lane sums are combined in fixed order (deterministic; a 512-long serial chain per channel would cost ~100 us of pure latency).
constexpr int FC = 4, FL = 64;
__device__ __forceinline__ void final_sums(const float* __restrict__ part, int nb, int C, int c, int lane, double& s1, double& s2) {
  __shared__ double red[FL][FC][2];
  double a = 0.0, b = 0.0;
  if (c < C)
#pragma unroll 8   // eight independent partial loads in flight per lane (the additions keep their order)
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

// FIN: local statistics (no process group) -- mean / invstd and the running-moment update of bn_finalize_kernel happen here, on the
// same float values the two-kernel path hands over through `moments` (bit-identical), one launch less per BN layer.
template <bool FIN>
__global__ __launch_bounds__(256) void bn_stats_final(const float* __restrict__ part, int nb, const float* __restrict__ x, long P, int C, float* __restrict__ moments,
                                                      float eps, float* __restrict__ mean, float* __restrict__ invstd, float* running_mean, float* running_var,
                                                      float momentum) {
  const int c = blockIdx.x * FC + (threadIdx.x & (FC - 1)), lane = threadIdx.x / FC;
  double s1, s2;
  final_sums(part, nb, C, c, lane, s1, s2);
  if (lane != 0 || c >= C) return;
  const double n = (double)P;
  const float m = (float)((double)x[c] + s1 / n), m2 = (float)fmax(s2 - s1 * s1 / n, 0.0), nf = (float)n;
  if constexpr (!FIN) {
    moments[c] = m;
    moments[C + c] = m2;
    moments[2 * C + c] = nf;
  } else {
    const float var = m2 / nf;
    mean[c] = m;
    invstd[c] = 1.f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    if (running_var && nf > 1.f) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / (nf - 1.f));   // n == 1: no unbiased estimate (0 / 0)
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ moments, int C, float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* running_mean, float* running_var, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float m = moments[c], m2 = moments[C + c], n = moments[2 * C + c];
  const float var = m2 / n;
  mean[c] = m;
  invstd[c] = 1.f / sqrtf(var + eps);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
  if (running_var && n > 1.f) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / (n - 1.f));
}


// Merge of the (mean, M2) slab partials a convolution epilogue emitted for its own output (conv_igemm.hip, staged epilogue): rows in slabs
// of 32, part[slab][c][2]. Block = 4 channels x 64 lanes; lane l accumulates, in double, A = sum n_b mean_b and B = sum (M2_b + n_b mean_b^2)
// over slabs l, l + 64, ... (two FMAs per slab, no division); the 64 lane pairs are added in lane order (fixed -> deterministic);
// mean = A / N, M2 = B - N mean^2 (the within-slab part is exact two-pass fp32; the between-slab part is a sum of squares in double:
// relative error 1e-16 (mean^2 / var), harmless below 1e8). MOM: write mean | M2 | count for the SyncBatchNorm exchange instead of finalising.
template <bool MOM>
__global__ __launch_bounds__(256) void bn_partials_final(const float* __restrict__ part, long rows, int C, float eps, float* __restrict__ mean,
                                                         float* __restrict__ invstd, float* running_mean, float* running_var, float momentum,
                                                         float* __restrict__ moments) {
  __shared__ double red[FL][FC][2];
  const int ci = threadIdx.x & (FC - 1), lane = threadIdx.x / FC;
  const int c = blockIdx.x * FC + ci;
  const long nslab = (rows + 31) / 32;
  double A = 0.0, B = 0.0;
  if (c < C)
#pragma unroll 8
    for (long sidx = lane; sidx < nslab; sidx += FL) {
      const float2 v = *reinterpret_cast<const float2*>(part + (sidx * C + c) * 2);
      const double nb = (double)min(32l, rows - sidx * 32), mb = (double)v.x;
      A = fma(nb, mb, A);
      B += fma(nb * mb, mb, (double)v.y);
    }
  red[lane][ci][0] = A, red[lane][ci][1] = B;
  __syncthreads();
  if (lane != 0 || c >= C) return;
  A = B = 0.0;
#pragma unroll 1
  for (int i = 0; i < FL; ++i) A += red[i][ci][0], B += red[i][ci][1];
  const double n = (double)rows, gm = A / n;
  const float m = (float)gm, m2 = (float)fmax(B - n * gm * gm, 0.0), nf = (float)n;
  if constexpr (MOM) {
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
// RELU 0: no activation. 1: mask = y > 0 read from the forward output (needed when a residual was added before the ReLU).
// 2: mask rebuilt from x (y = relu(bn(x)), no residual) -- one tensor less to read. GOUT: also store dyz (the gradient of the
// residual branch), so that the apply pass reads one tensor (dyz) instead of two (dy, y).
// 3: mask from the byte per float4 group bn_apply left behind (bit e = output e of the group was positive): 1 / 16 of the bytes of reading y.
template <int RELU, bool GOUT>
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* __restrict__ dy, long dpitch, const float* __restrict__ y, long ypitch,
                                                      const float* __restrict__ x, long xpitch, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ gout, long gpitch, long P, int C, int rows, float* __restrict__ part) {
  __shared__ float sm[RL][CB][2];
  const int g = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int c = blockIdx.y * CB + g * 4;
  const long p0 = (long)blockIdx.x * rows, p1 = min(P, p0 + rows);
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (c < C) {
    const float4 mu = PM_LD4(mean + c), is = PM_LD4(invstd + c);
    float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), be = ga;
    if (RELU == 2) ga = PM_LD4(gamma + c), be = PM_LD4(beta + c);
    for (long p = p0 + r; p < p1; p += RL) {
      float4 d = PM_LD4(dy + p * dpitch + c);
      const float4 v = PM_LD4(x + p * xpitch + c);
      if (RELU == 1) {
        const float4 o = PM_LD4(y + p * ypitch + c);
        d.x = o.x > 0.f ? d.x : 0.f, d.y = o.y > 0.f ? d.y : 0.f, d.z = o.z > 0.f ? d.z : 0.f, d.w = o.w > 0.f ? d.w : 0.f;
      } else if (RELU == 3) {
        const unsigned mb = reinterpret_cast<const unsigned char*>(y)[p * (C >> 2) + (c >> 2)];
        d.x = (mb & 1u) ? d.x : 0.f, d.y = (mb & 2u) ? d.y : 0.f, d.z = (mb & 4u) ? d.z : 0.f, d.w = (mb & 8u) ? d.w : 0.f;
      } else if (RELU == 2) {
        d.x = bn_affine(v.x, mu.x, is.x, ga.x, be.x) > 0.f ? d.x : 0.f, d.y = bn_affine(v.y, mu.y, is.y, ga.y, be.y) > 0.f ? d.y : 0.f;
        d.z = bn_affine(v.z, mu.z, is.z, ga.z, be.z) > 0.f ? d.z : 0.f, d.w = bn_affine(v.w, mu.w, is.w, ga.w, be.w) > 0.f ? d.w : 0.f;
      }
      if (GOUT) PM_ST4(gout + p * gpitch + c, d);
      s1[0] += d.x, s1[1] += d.y, s1[2] += d.z, s1[3] += d.w;
      s2[0] += d.x * ((v.x - mu.x) * is.x), s2[1] += d.y * ((v.y - mu.y) * is.y);
      s2[2] += d.z * ((v.z - mu.z) * is.z), s2[3] += d.w * ((v.w - mu.w) * is.w);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) sm[r][g * 4 + j][0] = s1[j], sm[r][g * 4 + j][1] = s2[j];
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
__global__ __launch_bounds__(256) void bn_bwd_final(const float* __restrict__ part, int nb, int C, float* __restrict__ sums) {
  const int c = blockIdx.x * FC + (threadIdx.x & (FC - 1)), lane = threadIdx.x / FC;
  double s1, s2;
  final_sums(part, nb, C, c, lane, s1, s2);
  if (lane != 0 || c >= C) return;
  sums[c] = (float)s1;
  sums[C + c] = (float)s2;
}

// Chan et al. merge of per-rank (mean | M2 | count) rows gathered over the process group: [W][3C] -> [3C], in double.
// FIN: mean / invstd and the running-moment update of bn_finalize_kernel in the same launch, on the float values the two-kernel path
// would hand over through `out` (bit-identical), so that SyncBatchNorm costs one launch less per layer.
template <bool FIN>
__global__ void bn_merge_kernel(const float* __restrict__ parts, int W, int C, float* __restrict__ out, float eps, float* __restrict__ mean,
                                float* __restrict__ invstd, float* running_mean, float* running_var, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double n = 0.0, s = 0.0;
  for (int r = 0; r < W; ++r) {
    const double cnt = parts[(long)r * 3 * C + 2 * C + c];
    n += cnt, s += cnt * (double)parts[(long)r * 3 * C + c];
  }
  const double gmean = s / n;
  double m2d = 0.0;
  for (int r = 0; r < W; ++r) {
    const double cnt = parts[(long)r * 3 * C + 2 * C + c], d = (double)parts[(long)r * 3 * C + c] - gmean;
    m2d += (double)parts[(long)r * 3 * C + C + c] + cnt * d * d;
  }
  const float m = (float)gmean, m2 = (float)m2d, nf = (float)n;
  if constexpr (!FIN) {
    out[c] = m, out[C + c] = m2, out[2 * C + c] = nf;
  } else {
    const float var = m2 / nf;
    mean[c] = m;
    invstd[c] = 1.f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    if (running_var && nf > 1.f) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / (nf - 1.f));   // n == 1: no unbiased estimate (0 / 0)
  }
}

// every BatchNorm of a network folded in one launch (eval-mode forward): table[i] = {gamma, beta, running_mean, running_var} device
// pointers of layer i, cs[i] its channel count, offs[i] its offset in the arena (scale at arena + offs[i], shift at arena + total + offs[i])
__global__ void bn_fold_multi_kernel(const unsigned long long* __restrict__ table, const int* __restrict__ cs, const int* __restrict__ offs, int total,
                                     float eps, float* __restrict__ arena) {
  const int i = blockIdx.x, C = cs[i];
  const float *g = (const float*)table[4 * i], *b = (const float*)table[4 * i + 1], *rm = (const float*)table[4 * i + 2], *rv = (const float*)table[4 * i + 3];
  float* scale = arena + offs[i];
  float* shift = arena + total + offs[i];
  for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < C; c += gridDim.y * blockDim.x) {
    const float s = g[c] / sqrtf(rv[c] + eps);      // the same expressions as bn_fold_kernel: identical values
    scale[c] = s;
    shift[c] = b[c] - rm[c] * s;
  }
}

__global__ void bn_fold_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ rm, const float* __restrict__ rv,
                               const float* __restrict__ cb, int C, float eps, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = g[c] / sqrtf(rv[c] + eps);
  scale[c] = s;
  shift[c] = b[c] - rm[c] * s + (cb ? cb[c] * s : 0.f);
}

int check_bn(const pm_tensor* x, const char* who) {
  PM_REQUIRE_F32(x, who);      // the bf16 forms are dispatched before this check (act16.hip); a mixed-type call ends here
  PM_REQUIRE(x && x->ptr && pm_vec4(x), PM_EINVAL, "%s: tensor must be 16B aligned, pitch %% 4 == 0 and C %% 4 == 0", who);
  return PM_OK;
}

}  // namespace

extern "C" size_t pm_bn_workspace(const pm_tensor* x) {
  if (pm_is_bf16(x)) return pm16_bn_workspace(x);
  const long P = pm_pixels(x);
  const int nb = pm_cdiv(P, chunk_rows(P, x->c));
  return pm_align_up((size_t)nb * x->c * 2 * sizeof(float), 256);
}

extern "C" int pm_bn_stats(const pm_tensor* x, float* moments, void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE(x && moments, PM_EINVAL, "bn_stats: null");
  if (pm_is_bf16(x)) return pm16_bn_stats(x, moments, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f, ws, ws_bytes, (hipStream_t)stream);
  if (int e = check_bn(x, "bn_stats")) return e;
  PM_REQUIRE(moments && ws && ws_bytes >= pm_bn_workspace(x), PM_EWORKSPACE, "bn_stats: workspace too small");
  const long P = pm_pixels(x);
  PM_REQUIRE(P > 0, PM_EINVAL, "bn_stats: empty tensor");
  const int rows = chunk_rows(P, x->c), nb = pm_cdiv(P, rows);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_stats_partial, dim3(nb, pm_cdiv(x->c, CB)), dim3(256), 0, st, (const float*)x->ptr, (long)x->pitch, P, x->c, rows, (float*)ws);
  hipLaunchKernelGGL(bn_stats_final<false>, dim3(pm_cdiv(x->c, FC)), dim3(256), 0, st, (const float*)ws, nb, (const float*)x->ptr, P, x->c, moments, 0.f,
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
  return pm_check_launch("bn_stats");
}

extern "C" int pm_bn_stats_finalize(const pm_tensor* x, float eps, float* mean, float* invstd, float* running_mean, float* running_var, float momentum,
                                    void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE(x && mean && invstd, PM_EINVAL, "bn_stats_finalize: bad args");
  if (pm_is_bf16(x)) return pm16_bn_stats(x, nullptr, eps, mean, invstd, running_mean, running_var, momentum, ws, ws_bytes, (hipStream_t)stream);
  if (int e = check_bn(x, "bn_stats_finalize")) return e;
  PM_REQUIRE(ws && ws_bytes >= pm_bn_workspace(x), PM_EWORKSPACE, "bn_stats_finalize: workspace too small");
  const long P = pm_pixels(x);
  PM_REQUIRE(P > 0, PM_EINVAL, "bn_stats_finalize: empty tensor");
  // torch.nn.BatchNorm2d in training mode: "Expected more than 1 value per channel when training" (B = 1 through ASPP's image-pooling branch)
  PM_REQUIRE(P > 1, PM_EINVAL, "bn_stats_finalize: expected more than 1 value per channel when training, got %ld", P);
  const int rows = chunk_rows(P, x->c), nb = pm_cdiv(P, rows);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_stats_partial, dim3(nb, pm_cdiv(x->c, CB)), dim3(256), 0, st, (const float*)x->ptr, (long)x->pitch, P, x->c, rows, (float*)ws);
  hipLaunchKernelGGL(bn_stats_final<true>, dim3(pm_cdiv(x->c, FC)), dim3(256), 0, st, (const float*)ws, nb, (const float*)x->ptr, P, x->c, (float*)nullptr, eps,
                     mean, invstd, running_mean, running_var, momentum);
  return pm_check_launch("bn_stats_finalize");
}

extern "C" int pm_bn_finalize(const float* moments, int c, float eps, float* mean, float* invstd, float* running_mean, float* running_var,
                              float momentum, void* stream) {
  PM_REQUIRE(moments && mean && invstd && c > 0, PM_EINVAL, "bn_finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(pm_cdiv(c, 64)), dim3(64), 0, (hipStream_t)stream, moments, c, eps, mean, invstd, running_mean, running_var,
                     momentum);
  return pm_check_launch("bn_finalize");
}

extern "C" int pm_bn_merge(const float* parts, int world, int c, float* moments, void* stream) {
  PM_REQUIRE(parts && moments && world >= 1 && c > 0, PM_EINVAL, "bn_merge: bad args");
  hipLaunchKernelGGL(bn_merge_kernel<false>, dim3(pm_cdiv(c, 64)), dim3(64), 0, (hipStream_t)stream, parts, world, c, moments, 0.f, (float*)nullptr,
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f);
  return pm_check_launch("bn_merge");
}

extern "C" int pm_bn_merge_finalize(const float* parts, int world, int c, float eps, float* mean, float* invstd, float* running_mean, float* running_var,
                                    float momentum, void* stream) {
  PM_REQUIRE(parts && mean && invstd && world >= 1 && c > 0, PM_EINVAL, "bn_merge_finalize: bad args");
  hipLaunchKernelGGL(bn_merge_kernel<true>, dim3(pm_cdiv(c, 64)), dim3(64), 0, (hipStream_t)stream, parts, world, c, (float*)nullptr, eps, mean, invstd,
                     running_mean, running_var, momentum);
  return pm_check_launch("bn_merge_finalize");
}

extern "C" int pm_bn_partials_finalize(const float* partials, int64_t pixels, int c, float eps, float* mean, float* invstd, float* running_mean,
                                       float* running_var, float momentum, float* moments, void* stream) {
  PM_REQUIRE(partials && pixels > 0 && c > 0 && (moments || (mean && invstd)), PM_EINVAL, "bn_partials_finalize: bad args");
  PM_REQUIRE(moments || pixels > 1, PM_EINVAL, "bn_partials_finalize: expected more than 1 value per channel when training, got %ld", (long)pixels);
  hipStream_t st = (hipStream_t)stream;
  if (moments)
    hipLaunchKernelGGL(bn_partials_final<true>, dim3(pm_cdiv(c, FC)), dim3(256), 0, st, partials, (long)pixels, c, eps, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, 0.f, moments);
  else
    hipLaunchKernelGGL(bn_partials_final<false>, dim3(pm_cdiv(c, FC)), dim3(256), 0, st, partials, (long)pixels, c, eps, mean, invstd, running_mean, running_var,
                       momentum, (float*)nullptr);
  return pm_check_launch("bn_partials_finalize");
}

extern "C" int pm_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, const float* conv_bias, int c, float eps, float* scale,
                          float* shift, void* stream) {
  PM_REQUIRE(gamma && beta && rm && rv && scale && shift && c > 0, PM_EINVAL, "bn_fold: bad args");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(pm_cdiv(c, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta, rm, rv, conv_bias, c, eps, scale, shift);
  return pm_check_launch("bn_fold");
}

extern "C" int pm_bn_fold_multi(const void* table, const int* cs, const int* offs, int n, int max_c, int total, float eps, float* arena, void* stream) {
  PM_REQUIRE(table && cs && offs && arena && n > 0 && max_c > 0 && total > 0, PM_EINVAL, "bn_fold_multi: bad args");
  hipLaunchKernelGGL(bn_fold_multi_kernel, dim3(n, pm_cdiv(max_c, 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)table, cs, offs, total, eps,
                     arena);
  return pm_check_launch("bn_fold_multi");
}

extern "C" int pm_bn_apply(const pm_tensor* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const pm_tensor* res,
                           int relu, const pm_tensor* y, void* stream) {
  return pm_bn_apply_mask(x, mean, invstd, gamma, beta, res, relu, y, nullptr, stream);
}

extern "C" int pm_bn_apply_mask(const pm_tensor* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const pm_tensor* res,
                                int relu, const pm_tensor* y, uint8_t* mask, void* stream) {
  PM_REQUIRE(x && y, PM_EINVAL, "bn_apply: null");
  if (pm_is_bf16(x)) return pm16_bn_apply_mask(x, mean, invstd, gamma, beta, res, relu, y, mask, (hipStream_t)stream);      // all tensors bf16; mask: one byte per 8 channels
  if (int e = check_bn(x, "bn_apply")) return e;
  if (int e = check_bn(y, "bn_apply")) return e;
  PM_REQUIRE(pm_same_shape(x, y) && mean && invstd && gamma && beta, PM_EINVAL, "bn_apply: bad args");
  if (res) {
    if (int e = check_bn(res, "bn_apply")) return e;
    PM_REQUIRE(pm_same_shape(x, res), PM_EINVAL, "bn_apply: residual shape mismatch");
  }
  const float *px = (const float*)x->ptr, *pr = res ? (const float*)res->ptr : nullptr;
  float* py = (float*)y->ptr;
  const long a = x->pitch, b = res ? res->pitch : 0, c = y->pitch, cq = x->c >> 2;
  return pm_ew_launch(true, pm_pixels(x), x->c, (hipStream_t)stream, "bn_apply", [=] __device__(long p, int ch) {
    const float4 v = PM_LD4(px + p * a + ch), mu = PM_LD4(mean + ch), is = PM_LD4(invstd + ch), ga = PM_LD4(gamma + ch), be = PM_LD4(beta + ch);
    float4 o = make_float4(bn_affine(v.x, mu.x, is.x, ga.x, be.x), bn_affine(v.y, mu.y, is.y, ga.y, be.y), bn_affine(v.z, mu.z, is.z, ga.z, be.z),
                           bn_affine(v.w, mu.w, is.w, ga.w, be.w));
    if (pr) {
      const float4 q = PM_LD4(pr + p * b + ch);
      o.x += q.x, o.y += q.y, o.z += q.z, o.w += q.w;
    }
    if (mask) mask[p * cq + (ch >> 2)] = (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
    if (relu) o.x = fmaxf(o.x, 0.f), o.y = fmaxf(o.y, 0.f), o.z = fmaxf(o.z, 0.f), o.w = fmaxf(o.w, 0.f);
    PM_ST4(py + p * c + ch, o);
  });
}

extern "C" int pm_bn_bwd_reduce(const pm_tensor* dy, const pm_tensor* y, const pm_tensor* x, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, int relu, const pm_tensor* gmask, float* sums, void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE(dy && x, PM_EINVAL, "bn_bwd_reduce: null");
  PM_REQUIRE(relu >= 0 && relu <= 2, PM_EINVAL, "bn_bwd_reduce: relu mode %d (0 none, 1 mask from y, 2 mask rebuilt from x)", relu);
  if (pm_is_bf16(x)) return pm16_bn_bwd_reduce(dy, y, nullptr, x, mean, invstd, gamma, beta, relu, gmask, sums, ws, ws_bytes, (hipStream_t)stream);
  if (int e = check_bn(dy, "bn_bwd_reduce")) return e;
  if (int e = check_bn(x, "bn_bwd_reduce")) return e;
  PM_REQUIRE(pm_same_shape(dy, x) && mean && invstd && sums, PM_EINVAL, "bn_bwd_reduce: bad args");
  PM_REQUIRE(relu >= 0 && relu <= 2, PM_EINVAL, "bn_bwd_reduce: relu mode %d (0 none, 1 mask from y, 2 mask rebuilt from x)", relu);
  PM_REQUIRE(relu != 1 || (y && pm_vec4(y) && pm_same_shape(y, x)), PM_EINVAL, "bn_bwd_reduce: relu mode 1 needs the forward output");
  PM_REQUIRE(relu != 2 || (gamma && beta), PM_EINVAL, "bn_bwd_reduce: relu mode 2 needs gamma and beta");
  PM_REQUIRE(!gmask || (relu != 0 && pm_vec4(gmask) && pm_same_shape(gmask, x)), PM_EINVAL, "bn_bwd_reduce: gmask needs a ReLU mode and the shape of x");
  PM_REQUIRE(ws && ws_bytes >= pm_bn_workspace(x), PM_EWORKSPACE, "bn_bwd_reduce: workspace too small");
  const long P = pm_pixels(x);
  const int rows = chunk_rows(P, x->c), nb = pm_cdiv(P, rows);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nb, pm_cdiv(x->c, CB));
  const float *pdy = (const float*)dy->ptr, *py = relu == 1 ? (const float*)y->ptr : nullptr, *px = (const float*)x->ptr;
  const long yp = relu == 1 ? y->pitch : 0;
  float* pg = gmask ? (float*)gmask->ptr : nullptr;
  const long gp = gmask ? gmask->pitch : 0;
#define PM_BN_BWD_PARTIAL(R, G)                                                                                                                         \
  hipLaunchKernelGGL((bn_bwd_partial<R, G>), grid, dim3(256), 0, st, pdy, (long)dy->pitch, py, yp, px, (long)x->pitch, mean, invstd, gamma, beta, pg, gp, P, \
                     x->c, rows, (float*)ws)
  if (relu == 0) PM_BN_BWD_PARTIAL(0, false);
  else if (relu == 1 && gmask) PM_BN_BWD_PARTIAL(1, true);
  else if (relu == 1) PM_BN_BWD_PARTIAL(1, false);
  else if (gmask) PM_BN_BWD_PARTIAL(2, true);
  else PM_BN_BWD_PARTIAL(2, false);
#undef PM_BN_BWD_PARTIAL
  hipLaunchKernelGGL(bn_bwd_final, dim3(pm_cdiv(x->c, FC)), dim3(256), 0, st, (const float*)ws, nb, x->c, sums);
  return pm_check_launch("bn_bwd_reduce");
}

// BN + residual + ReLU backward reduce with the ReLU mask taken from pm_bn_apply_mask's bytes instead of the forward output: sums and the masked
// gradient gmask (= the gradient of the residual branch), as pm_bn_bwd_reduce(relu = 1) gives them -- same values, 1 / 16 of the mask bytes.
extern "C" int pm_bn_bwd_reduce_mask(const pm_tensor* dy, const uint8_t* mask, const pm_tensor* x, const float* mean, const float* invstd,
                                     const pm_tensor* gmask, float* sums, void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE(dy && x && mask, PM_EINVAL, "bn_bwd_reduce_mask: null");
  if (pm_is_bf16(x)) return pm16_bn_bwd_reduce(dy, nullptr, mask, x, mean, invstd, nullptr, nullptr, 3, gmask, sums, ws, ws_bytes, (hipStream_t)stream);
  if (int e = check_bn(dy, "bn_bwd_reduce_mask")) return e;
  if (int e = check_bn(x, "bn_bwd_reduce_mask")) return e;
  PM_REQUIRE(pm_same_shape(dy, x) && mean && invstd && sums && mask, PM_EINVAL, "bn_bwd_reduce_mask: bad args");
  PM_REQUIRE(!gmask || (pm_vec4(gmask) && pm_same_shape(gmask, x)), PM_EINVAL, "bn_bwd_reduce_mask: gmask must have the shape of x");
  PM_REQUIRE(ws && ws_bytes >= pm_bn_workspace(x), PM_EWORKSPACE, "bn_bwd_reduce_mask: workspace too small");
  const long P = pm_pixels(x);
  const int rows = chunk_rows(P, x->c), nb = pm_cdiv(P, rows);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nb, pm_cdiv(x->c, CB));
  float* pg = gmask ? (float*)gmask->ptr : nullptr;
  const long gp = gmask ? gmask->pitch : 0;
  if (gmask)
    hipLaunchKernelGGL((bn_bwd_partial<3, true>), grid, dim3(256), 0, st, (const float*)dy->ptr, (long)dy->pitch, reinterpret_cast<const float*>(mask), 0l,
                       (const float*)x->ptr, (long)x->pitch, mean, invstd, (const float*)nullptr, (const float*)nullptr, pg, gp, P, x->c, rows, (float*)ws);
  else
    hipLaunchKernelGGL((bn_bwd_partial<3, false>), grid, dim3(256), 0, st, (const float*)dy->ptr, (long)dy->pitch, reinterpret_cast<const float*>(mask), 0l,
                       (const float*)x->ptr, (long)x->pitch, mean, invstd, (const float*)nullptr, (const float*)nullptr, pg, gp, P, x->c, rows, (float*)ws);
  hipLaunchKernelGGL(bn_bwd_final, dim3(pm_cdiv(x->c, FC)), dim3(256), 0, st, (const float*)ws, nb, x->c, sums);
  return pm_check_launch("bn_bwd_reduce_mask");
}

extern "C" int pm_bn_bwd_apply(const pm_tensor* dy, const pm_tensor* y, const pm_tensor* x, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, const float* sums, float count, int relu, const pm_tensor* dx, const pm_tensor* dres, void* stream) {
  PM_REQUIRE(dy && x && dx, PM_EINVAL, "bn_bwd_apply: null");
  if (pm_is_bf16(x)) return pm16_bn_bwd_apply(dy, y, x, mean, invstd, gamma, beta, sums, count, relu, dx, dres, (hipStream_t)stream);
  if (int e = check_bn(dy, "bn_bwd_apply")) return e;
  if (int e = check_bn(x, "bn_bwd_apply")) return e;
  if (int e = check_bn(dx, "bn_bwd_apply")) return e;
  PM_REQUIRE(pm_same_shape(dy, x) && pm_same_shape(dx, x) && mean && invstd && gamma && sums, PM_EINVAL, "bn_bwd_apply: bad args");
  PM_REQUIRE(relu >= 0 && relu <= 2, PM_EINVAL, "bn_bwd_apply: relu mode %d (0 none, 1 mask from y, 2 mask rebuilt from x)", relu);
  PM_REQUIRE(relu != 1 || (y && pm_vec4(y) && pm_same_shape(y, x)), PM_EINVAL, "bn_bwd_apply: relu mode 1 needs the forward output");
  PM_REQUIRE(relu != 2 || beta, PM_EINVAL, "bn_bwd_apply: relu mode 2 needs beta");
  PM_REQUIRE(!dres || (pm_vec4(dres) && pm_same_shape(dres, x)), PM_EINVAL, "bn_bwd_apply: dres shape mismatch");
  const float *pd = (const float*)dy->ptr, *po = relu == 1 ? (const float*)y->ptr : nullptr, *px = (const float*)x->ptr;
  float *pdx = (float*)dx->ptr, *pdr = dres ? (float*)dres->ptr : nullptr;
  const long a = dy->pitch, b = relu == 1 ? y->pitch : 0, c = x->pitch, d = dx->pitch, e2 = dres ? dres->pitch : 0;
  const bool from_x = relu == 2;
  const int C = x->c;
  // count <= 0: the element count lives on the device at sums[2 * C] -- SyncBatchNorm all-reduces it with the two sums, so ranks with
  // different batch sizes normalise by the true global count (torch.nn.SyncBatchNorm gathers the counts the same way)
  const bool dev_count = !(count > 0.f);
  const float host_inv_n = dev_count ? 0.f : 1.f / count;
  return pm_ew_launch(true, pm_pixels(x), C, (hipStream_t)stream, "bn_bwd_apply", [=] __device__(long p, int ch) {
    const float inv_n = dev_count ? 1.f / sums[2 * C] : host_inv_n;
    float4 g = PM_LD4(pd + p * a + ch);
    if (po) {
      const float4 o = PM_LD4(po + p * b + ch);
      g.x = o.x > 0.f ? g.x : 0.f, g.y = o.y > 0.f ? g.y : 0.f, g.z = o.z > 0.f ? g.z : 0.f, g.w = o.w > 0.f ? g.w : 0.f;
    }
    const float4 v = PM_LD4(px + p * c + ch), mu = PM_LD4(mean + ch), is = PM_LD4(invstd + ch), ga = PM_LD4(gamma + ch);
    if (from_x) {
      const float4 be = PM_LD4(beta + ch);
      g.x = bn_affine(v.x, mu.x, is.x, ga.x, be.x) > 0.f ? g.x : 0.f, g.y = bn_affine(v.y, mu.y, is.y, ga.y, be.y) > 0.f ? g.y : 0.f;
      g.z = bn_affine(v.z, mu.z, is.z, ga.z, be.z) > 0.f ? g.z : 0.f, g.w = bn_affine(v.w, mu.w, is.w, ga.w, be.w) > 0.f ? g.w : 0.f;
    }
    if (pdr) PM_ST4(pdr + p * e2 + ch, g);
    const float4 s1 = PM_LD4(sums + ch), s2 = PM_LD4(sums + C + ch);
    float4 r;
    r.x = (g.x - s1.x * inv_n - (v.x - mu.x) * is.x * (s2.x * inv_n)) * (is.x * ga.x);
    r.y = (g.y - s1.y * inv_n - (v.y - mu.y) * is.y * (s2.y * inv_n)) * (is.y * ga.y);
    r.z = (g.z - s1.z * inv_n - (v.z - mu.z) * is.z * (s2.z * inv_n)) * (is.z * ga.z);
    r.w = (g.w - s1.w * inv_n - (v.w - mu.w) * is.w * (s2.w * inv_n)) * (is.w * ga.w);
    PM_ST4(pdx + p * d + ch, r);
  });
}
