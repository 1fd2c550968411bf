// K7/K8: the categorical memory of "Pin the Memory" (/root/reference/network/memory.py) -- HBM-bound row kernels.
//   read   (memory.py:317-336 + get_score :167-189): qhat = x/max(|x|,1e-12); S = qhat.M^T; P = softmax_slots(S [+gumbel]);
//          R = P.M; writes [qhat | R] (the 2d-channel input of memory.output), S and P. Forward and the dx-only backward: four waves per
//          32 query rows, both products as MFMA tiles (mem_read_fwd_mfma_kernel / mem_read_bwd_mfma_kernel below); backward with dmem
//          (meta-test read through a written memory): one wave per row (d = 256 = 64 lanes x float4), per-slot dot products reduced
//          with wave shuffles.
//   write  (memory.py:206-239): 4-tap bilinear(align_corners) soft labels straight from the int64 mask (never the
//          755 MB one-hot), class-masked accumulation of nominator[20][256] / denominator[20] in per-wave LDS slabs,
//          fixed-order two-stage reduce; momentum update + renormalise with a device-side `den != 0` predicate
//          (the reference syncs the host 19 times per step, memory.py:234-237).
// All reductions are deterministic (no atomics).
#include "pm_common.h"

namespace {

constexpr int D = 256;     // feature dim (mem_dim); one wave = one row of 64 float4
constexpr int MAXM = 32;   // max slots (+1 for the ignore class in write)
constexpr float EPS = 1e-12f;
typedef float mr_f32x16 __attribute__((ext_vector_type(16)));   // one 32x32 fp32 MFMA result tile per wave

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// softmax over ALL rows per slot column (memory.py:186). Two launches over an L2-resident [N][m]: (1) every block folds its 128 rows
// into a per-column (max, sum exp) pair with the online-softmax merge, (2) every block merges all partials in block order (fixed,
// deterministic) and normalises its rows. Thread = (column c < 32, row lane rl < 8).
constexpr int CS_ROWS = 128;
struct MaxSum {
  float m, s;
};
__device__ __forceinline__ MaxSum cs_merge(MaxSum a, MaxSum b) {   // (max, sum of exp(v - max)) of the union
  const float m = fmaxf(a.m, b.m);
  MaxSum r;
  r.m = m;
  r.s = (a.m == -INFINITY ? 0.f : a.s * expf(a.m - m)) + (b.m == -INFINITY ? 0.f : b.s * expf(b.m - m));
  return r;
}
__global__ __launch_bounds__(256) void mem_colsoftmax_partial_kernel(const float* __restrict__ score, const float* __restrict__ noise, long rows, int M,
                                                                     float* __restrict__ part) {
  __shared__ MaxSum red[8][32];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const long r0 = (long)blockIdx.x * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
  MaxSum acc = {-INFINITY, 0.f};
  if (c < M)
    for (long r = r0 + rl; r < r1; r += 8) {
      const float v = score[r * M + c] + (noise ? noise[r * M + c] : 0.f);
      acc = cs_merge(acc, MaxSum{v, 1.f});
    }
  red[rl][c] = acc;
  __syncthreads();
  if (rl == 0 && c < M) {
#pragma unroll
    for (int i = 1; i < 8; ++i) acc = cs_merge(acc, red[i][c]);
    part[((long)blockIdx.x * M + c) * 2] = acc.m;
    part[((long)blockIdx.x * M + c) * 2 + 1] = acc.s;
  }
}
// Apply pass (round 4 form). What bounded the older one was the chain of dependent L2 round trips per thread (one partial load + merge after the other:
// 12 us for 144 partials, 35-44 us for the 576 32-row tiles the read kernel leaves), not arithmetic. Now a 1 024-thread block = (column c < 32) x (32 ranges
// of partials): a thread's <= 24 partials are loaded in one go into registers (all in flight at once), the global max per column is found first, then
// sum_b s_b exp(m_b - max) with independent terms; ranges are combined in order through LDS: deterministic. 256 rows per block.
constexpr int CSA_T = 1024, CSA_PER = 24, CSA_ROWS = 256;
__global__ __launch_bounds__(CSA_T) void mem_colsoftmax_apply_kernel(const float* __restrict__ score, const float* __restrict__ noise, long rows, int M,
                                                                     const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ float red[32][33];
  const int c = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int per = (nb + 31) / 32, b0 = rl * per, b1 = min(nb, b0 + per);
  const float2* p2 = reinterpret_cast<const float2*>(part);
  const bool regs = per <= CSA_PER;
  float2 pv[CSA_PER];
  float gm = -INFINITY;
  if (regs) {
#pragma unroll
    for (int i = 0; i < CSA_PER; ++i) {
      pv[i] = make_float2(-INFINITY, 0.f);
      if (c < M && b0 + i < b1) pv[i] = p2[(long)(b0 + i) * M + c];
    }
#pragma unroll
    for (int i = 0; i < CSA_PER; ++i) gm = fmaxf(gm, pv[i].x);
  } else if (c < M) {
    for (int b = b0; b < b1; ++b) gm = fmaxf(gm, p2[(long)b * M + c].x);
  }
  red[rl][c] = gm;
  __syncthreads();
  gm = red[0][c];
#pragma unroll
  for (int i = 1; i < 32; ++i) gm = fmaxf(gm, red[i][c]);
  __syncthreads();
  float gs = 0.f;
  if (regs) {
#pragma unroll
    for (int i = 0; i < CSA_PER; ++i) gs += pv[i].x == -INFINITY ? 0.f : pv[i].y * expf(pv[i].x - gm);
  } else if (c < M) {
    for (int b = b0; b < b1; ++b) {
      const float2 p = p2[(long)b * M + c];
      gs += p.x == -INFINITY ? 0.f : p.y * expf(p.x - gm);
    }
  }
  red[rl][c] = gs;
  __syncthreads();
  gs = red[0][c];
#pragma unroll
  for (int i = 1; i < 32; ++i) gs += red[i][c];
  if (c >= M) return;
  const long r0 = (long)blockIdx.x * CSA_ROWS, r1 = min(rows, r0 + CSA_ROWS);
  const float rgs = 1.f / gs;
#pragma unroll 8
  for (long r = r0 + rl; r < r1; r += 32) out[r * M + c] = expf(score[r * M + c] + (noise ? noise[r * M + c] : 0.f) - gm) * rgs;
}

// ---------------------------------------------------------------------------------------------------------------
// read backward: dx (and optionally per-block partials of dmem)
// ---------------------------------------------------------------------------------------------------------------
template <int M_, bool DMEM>
__global__ __launch_bounds__(256) void mem_read_bwd_kernel(const float* __restrict__ x, long xp, long rows, const float* __restrict__ mem, int m_rt,
                                                           const float* __restrict__ pm, const float* __restrict__ dqr, long dqp,
                                                           const float* __restrict__ dsx, float* __restrict__ dx, long dxp, float* __restrict__ dmem_part) {
  const int M = M_ > 0 ? M_ : m_rt;
  __shared__ __align__(16) float smem[MAXM * D];
  for (int i = threadIdx.x; i < M * D / 4; i += 256) reinterpret_cast<float4*>(smem)[i] = reinterpret_cast<const float4*>(mem)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float4 dm[DMEM ? (M_ > 0 ? M_ : MAXM) : 1];
#pragma unroll
  for (int j = 0; j < (DMEM ? (M_ > 0 ? M_ : MAXM) : 1); ++j) dm[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long r = (long)blockIdx.x * 4 + wv; r < rows; r += (long)gridDim.x * 4) {
    const float4 v = PM_LD4(x + r * xp + lane * 4);
    const float n0 = sqrtf(pm_wave_sum(dot4(v, v)));
    const float nrm = fmaxf(n0, EPS);
    const float4 q = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
    float4 dq = PM_LD4(dqr + r * dqp + lane * 4);
    const float4 dr = PM_LD4(dqr + r * dqp + D + lane * 4);
    float p[M_ > 0 ? M_ : MAXM], dp[M_ > 0 ? M_ : MAXM];
    float pdp = 0.f;
#pragma unroll
    for (int j = 0; j < (M_ > 0 ? M_ : MAXM); ++j)
      if (j < M) {
        p[j] = pm[r * M + j];
        dp[j] = pm_wave_sum(dot4(dr, reinterpret_cast<const float4*>(smem + j * D)[lane]));
        pdp += p[j] * dp[j];
      }
#pragma unroll
    for (int j = 0; j < (M_ > 0 ? M_ : MAXM); ++j)
      if (j < M) {
        const float ds = p[j] * (dp[j] - pdp) + (dsx ? dsx[r * M + j] : 0.f);
        const float4 mv = reinterpret_cast<const float4*>(smem + j * D)[lane];
        dq.x += ds * mv.x, dq.y += ds * mv.y, dq.z += ds * mv.z, dq.w += ds * mv.w;
        if (DMEM) {
          dm[j].x += ds * q.x + p[j] * dr.x, dm[j].y += ds * q.y + p[j] * dr.y;
          dm[j].z += ds * q.z + p[j] * dr.z, dm[j].w += ds * q.w + p[j] * dr.w;
        }
      }
    // through qhat = x / max(|x|, eps)
    float4 o;
    if (n0 >= EPS) {
      const float qd = pm_wave_sum(dot4(q, dq));
      o = make_float4((dq.x - q.x * qd) / nrm, (dq.y - q.y * qd) / nrm, (dq.z - q.z * qd) / nrm, (dq.w - q.w * qd) / nrm);
    } else {
      o = make_float4(dq.x / nrm, dq.y / nrm, dq.z / nrm, dq.w / nrm);
    }
    PM_ST4(dx + r * dxp + lane * 4, o);
  }
  if (DMEM) {  // cross-wave reduce through LDS (memory copy no longer needed), then one partial slab per block
    __syncthreads();
    float* red = smem;  // reuse: [4 waves] would need 4*M*D floats > MAXM*D; reduce slot by slot instead
#pragma unroll
    for (int j = 0; j < (M_ > 0 ? M_ : MAXM); ++j)
      if (j < M) {
        reinterpret_cast<float4*>(red + wv * D)[lane] = dm[j];
        __syncthreads();
        if (wv == 0) {
          float4 a = reinterpret_cast<float4*>(red)[lane], b = reinterpret_cast<float4*>(red + D)[lane];
          float4 c = reinterpret_cast<float4*>(red + 2 * D)[lane], d = reinterpret_cast<float4*>(red + 3 * D)[lane];
          float4 s = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
          PM_ST4(dmem_part + ((long)blockIdx.x * M + j) * D + lane * 4, s);
        }
        __syncthreads();
      }
  }
}

// out[i] = sum_b part[b][i]: 16 elements x 16 partial-lanes per block (lane z sums b = z, z + 16, ...; lanes combined in order)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nb, long n, float* __restrict__ out) {
  __shared__ float red[16][16];
  const int e = threadIdx.x & 15, z = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + e;
  float s = 0.f;
  if (i < n)
    for (int b = z; b < nb; b += 16) s += part[(long)b * n + i];
  red[z][e] = s;
  __syncthreads();
  if (z == 0 && i < n) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][e];
    out[i] = s;
  }
}

// the same sum (same association order: bit-identical) for slabs of whole float4 groups: thread = (float4 column, partial lane), 16-byte loads, four in flight
__global__ __launch_bounds__(256) void reduce_partials_vec4_kernel(const float4* __restrict__ part, int nb, long n4, float4* __restrict__ out) {
  __shared__ float4 red[16][16];
  const int e = threadIdx.x & 15, z = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + e;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
#pragma unroll 4
    for (int b = z; b < nb; b += 16) {
      const float4 v = part[(long)b * n4 + i];
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  }
  red[z][e] = s;
  __syncthreads();
  if (z == 0 && i < n4) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s.x += red[k][e].x, s.y += red[k][e].y, s.z += red[k][e].z, s.w += red[k][e].w;
    out[i] = s;
  }
}
inline void launch_reduce_partials(const float* part, int nb, long n, float* out, hipStream_t st) {
  if ((n & 3) == 0 && pm_aligned16(part) && pm_aligned16(out))
    hipLaunchKernelGGL(reduce_partials_vec4_kernel, dim3(pm_cdiv(n / 4, 16)), dim3(256), 0, st, reinterpret_cast<const float4*>(part), nb, n / 4,
                       reinterpret_cast<float4*>(out));
  else
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(pm_cdiv(n, 16)), dim3(256), 0, st, part, nb, n, out);
}

// ---------------------------------------------------------------------------------------------------------------
// write: accumulate nominator / denominator
// ---------------------------------------------------------------------------------------------------------------
struct Taps {
  int cls[4];
  float w[4];
};
__device__ __forceinline__ Taps soft_label_taps(const int64_t* __restrict__ lab, int b, int y, int x, int H, int W, int h, int w, float sy, float sx, int m) {
  const pm_lerp ly = pm_ac_lerp(sy, y, H), lx = pm_ac_lerp(sx, x, W);  // downsample: source grid is the H x W mask
  (void)h, (void)w;
  Taps t;
  const int64_t* base = lab + (long)b * H * W;
  const int ys[2] = {ly.i0, ly.i1}, xs[2] = {lx.i0, lx.i1};
  const float wy[2] = {ly.w0, ly.w1}, wx[2] = {lx.w0, lx.w1};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int64_t c = base[(long)ys[i] * W + xs[j]];
      if (c == 255) c = m;                       // memory.py:220
      t.cls[i * 2 + j] = (c >= 0 && c <= m) ? (int)c : -1;
      t.w[i * 2 + j] = wy[i] * wx[j];
    }
  return t;
}

constexpr int ACC_W = 4;   // waves per block
__global__ __launch_bounds__(256) void mem_write_accum_kernel(const float* __restrict__ z, long zp, int n, int h, int w, const int64_t* __restrict__ lab,
                                                              int H, int W, int m, int normalize, float sy, float sx, float* __restrict__ part) {
  extern __shared__ __align__(16) float sacc[];   // [ACC_W][(m+1)*D + MAXM]
  const int slab = (m + 1) * D + MAXM;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* mine = sacc + wv * slab;
  for (int i = lane; i < slab; i += 64) mine[i] = 0.f;
  const long rows = (long)n * h * w;
  const long stride = (long)gridDim.x * ACC_W;
  long r = (long)blockIdx.x * ACC_W + wv;
  auto fetch = [&](long rr, float4& v, Taps& t) {
    v = PM_LD4(z + rr * zp + lane * 4);
    t = soft_label_taps(lab, (int)(rr / ((long)w * h)), (int)((rr / w) % h), (int)(rr % w), H, W, h, w, sy, sx, m);
  };
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  Taps t{};
  if (r < rows) fetch(r, v, t);
  while (r < rows) {   // the next row's feature vector and label taps are in flight while this one is accumulated
    const long rn = r + stride;
    float4 vn = v;
    Taps tn = t;
    if (rn < rows) fetch(rn, vn, tn);
    if (normalize) {
      const float nrm = fmaxf(sqrtf(pm_wave_sum(dot4(v, v))), EPS);
      v = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (t.cls[k] < 0 || t.w[k] == 0.f) continue;   // wave-uniform
      float4* a = reinterpret_cast<float4*>(mine + t.cls[k] * D) + lane;
      float4 c = *a;
      c.x += t.w[k] * v.x, c.y += t.w[k] * v.y, c.z += t.w[k] * v.z, c.w += t.w[k] * v.w;
      *a = c;
      if (lane == 0) mine[(m + 1) * D + t.cls[k]] += t.w[k];
    }
    v = vn, t = tn, r = rn;
  }
  __syncthreads();
  const int nout = (m + 1) * D + (m + 1);
  for (int i = threadIdx.x; i < nout; i += 256) {
    const int src = i < (m + 1) * D ? i : (m + 1) * D + (i - (m + 1) * D);
    part[(long)blockIdx.x * nout + i] = (sacc[src] + sacc[slab + src]) + (sacc[2 * slab + src] + sacc[3 * slab + src]);
  }
}

// The same accumulation as a split-K product on the matrix cores: nominator[c][d] = sum_r Y[r][c] * zhat[r][d] with Y the (<= 4 non-zeros per row)
// soft-label matrix, i.e. a (32 padded classes) x (64-row chunk) x (256 channels) GEMM per chunk, K split over the blocks. What bounded the slab
// kernel above was not bytes but its dependent chain -- 18 rows per wave, each a global round trip plus four LDS read-modify-writes (44 us for
// 19 MB); here every row of a chunk is in flight at once (16 float4 per lane), and the scatter by class becomes the A operand:
//   * wave w = (channel half ch = w & 1, row half rh = w >> 1); lane (l31 = lane & 31, half = lane >> 5) loads, for s = 0..15, the float4 of row
//     32 rh + 2 s + half at channels 128 ch + 4 l31: half-waves read 512 contiguous bytes;
//   * row norms: per-lane partial sums of squares parked in LDS, four threads per row fold the 64 partials in fixed order; the 1 / max(norm, eps)
//     is folded into the class weights (A[c][r] = (sum of the row's tap weights of class c) / norm_r), so z itself is never rescaled;
//   * 4 MFMAs (32x32x2 f32) per loaded float4: B = component e of the float4 (output column l31 of tile e = channel 128 ch + 4 l31 + e), so each
//     lane ends up with float4s of consecutive channels per class -- partial rows leave as 512-byte runs;
//   * denominators are summed outside the product (plain tap weights, fixed order); the two row halves meet in LDS (fixed order).
// Deterministic, no atomics; partial layout = the slab kernel's, so reduce_partials_kernel finishes both.
constexpr int AM_ROWS = 64;                 // rows per chunk
constexpr int AM_SSP = 65;                  // padded row of partial squared norms
__global__ __launch_bounds__(256) void mem_write_accum_mfma_kernel(const float* __restrict__ z, long zp, int n, int h, int w, const int64_t* __restrict__ lab,
                                                                   int H, int W, int m, int normalize, float sy, float sx, float* __restrict__ part) {
  __shared__ __align__(16) float am_ss[MAXM * D];              // partial squared norms [64][65]; afterwards the row-half exchange tile [m + 1][256]
  __shared__ __align__(16) int am_cls[AM_ROWS][4];
  __shared__ __align__(16) float am_w[AM_ROWS][4];
  __shared__ float am_rn[AM_ROWS];
  __shared__ float am_den[8][32];
  static_assert(AM_ROWS * AM_SSP <= MAXM * D, "the squared-norm partials share the exchange tile");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int ch = wv & 1, rh = wv >> 1;
  const long rows = (long)n * h * w;
  const long nchunks = (rows + AM_ROWS - 1) / AM_ROWS;
  mr_f32x16 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[e][q] = 0.f;
  float den = 0.f;                                              // thread t < 32: denominator of class t
  for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const long r0 = chunk * AM_ROWS;
    if (chunk != (long)blockIdx.x) __syncthreads();             // the previous chunk is done with the tap tables
    float4 v[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = PM_LD4(z + min(r0 + 32 * rh + 2 * s + half, rows - 1) * zp + 128 * ch + 4 * l31);
    if (threadIdx.x < AM_ROWS) {                                // the row's four label taps (rows past the end: no class)
      const long rr = r0 + threadIdx.x;
      Taps t;
      if (rr < rows) {
        const unsigned r32 = (unsigned)rr, px = r32 / (unsigned)w;                  // rows < 2^31 (checked by the launcher): 32-bit divisions
        t = soft_label_taps(lab, (int)(px / (unsigned)h), (int)(px % (unsigned)h), (int)(r32 % (unsigned)w), H, W, h, w, sy, sx, m);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) t.cls[k] = -1, t.w[k] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) am_cls[threadIdx.x][k] = t.cls[k], am_w[threadIdx.x][k] = t.cls[k] < 0 ? 0.f : t.w[k];
    }
    if (normalize) {
#pragma unroll
      for (int s = 0; s < 16; ++s) am_ss[(32 * rh + 2 * s + half) * AM_SSP + 32 * ch + l31] = dot4(v[s], v[s]);
    }
    __syncthreads();
    {
      const int row = threadIdx.x >> 2, pq = threadIdx.x & 3;
      float s2 = 0.f;
      if (normalize) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s2 += am_ss[row * AM_SSP + 16 * pq + i];
        s2 += __shfl_xor(s2, 1, 64);
        s2 += __shfl_xor(s2, 2, 64);
      }
      if (pq == 0) am_rn[row] = normalize ? 1.f / fmaxf(sqrtf(s2), EPS) : 1.f;
      // denominators: class c = t & 31 over the 8 rows of group t >> 5, tap order
      const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) d += am_cls[8 * grp + i][k] == c ? am_w[8 * grp + i][k] : 0.f;
      am_den[grp][c] = d;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      float d = am_den[0][threadIdx.x];
#pragma unroll
      for (int g = 1; g < 8; ++g) d += am_den[g][threadIdx.x];
      den += d;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int rr = 32 * rh + 2 * s + half;
      const int4 tc = *reinterpret_cast<const int4*>(am_cls[rr]);
      const float4 tw = *reinterpret_cast<const float4*>(am_w[rr]);
      float a = (tc.x == l31 ? tw.x : 0.f);
      a += (tc.y == l31 ? tw.y : 0.f);
      a += (tc.z == l31 ? tw.z : 0.f);
      a += (tc.w == l31 ? tw.w : 0.f);
      a *= am_rn[rr];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[s].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[s].y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[s].z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[s].w, acc[3], 0, 0, 0);
    }
  }
  // the two row halves meet in LDS: rh = 1 parks, rh = 0 adds and stores. Register q of a lane = class (q & 3) + 8 (q >> 2) + 4 half.
  __syncthreads();
  float* xch = am_ss;                                           // [m + 1][256]
  if (rh == 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = (q & 3) + 8 * (q >> 2) + 4 * half;
      if (c <= m) *reinterpret_cast<float4*>(xch + c * D + 128 * ch + 4 * l31) = make_float4(acc[0][q], acc[1][q], acc[2][q], acc[3][q]);
    }
  }
  __syncthreads();
  const int nout = (m + 1) * D + (m + 1);
  float* mine = part + (long)blockIdx.x * nout;
  if (rh == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = (q & 3) + 8 * (q >> 2) + 4 * half;
      if (c <= m) {
        const float4 o = *reinterpret_cast<const float4*>(xch + c * D + 128 * ch + 4 * l31);
        float* dst = mine + c * D + 128 * ch + 4 * l31;         // 512-byte runs per half-wave
        const float4 r = make_float4(acc[0][q] + o.x, acc[1][q] + o.y, acc[2][q] + o.z, acc[3][q] + o.w);
        if ((nout & 3) == 0) PM_ST4(dst, r);                    // block partials are 16-byte aligned when (m + 1) % 4 == 0 (19 classes + ignore)
        else dst[0] = r.x, dst[1] = r.y, dst[2] = r.z, dst[3] = r.w;
      }
    }
  }
  if ((int)threadIdx.x <= m) mine[(m + 1) * D + threadIdx.x] = den;
}

__global__ __launch_bounds__(256) void mem_write_accum_bwd_kernel(const float* __restrict__ z, long zp, int n, int h, int w, const int64_t* __restrict__ lab,
                                                                  int H, int W, int m, int normalize, float sy, float sx, const float* __restrict__ dnom,
                                                                  float* __restrict__ dz, long dzp) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long rows = (long)n * h * w;
  for (long r = (long)blockIdx.x * 4 + wv; r < rows; r += (long)gridDim.x * 4) {
    const int x = (int)(r % w), y = (int)((r / w) % h), b = (int)(r / ((long)w * h));
    const Taps t = soft_label_taps(lab, b, y, x, H, W, h, w, sy, sx, m);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (t.cls[k] < 0 || t.w[k] == 0.f) continue;
      const float4 d = PM_LD4(dnom + (long)t.cls[k] * D + lane * 4);
      g.x += t.w[k] * d.x, g.y += t.w[k] * d.y, g.z += t.w[k] * d.z, g.w += t.w[k] * d.w;
    }
    if (normalize) {
      const float4 v = PM_LD4(z + r * zp + lane * 4);
      const float n0 = sqrtf(pm_wave_sum(dot4(v, v)));
      const float nrm = fmaxf(n0, EPS);
      if (n0 >= EPS) {
        const float4 q = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
        const float qd = pm_wave_sum(dot4(q, g));
        g = make_float4((g.x - q.x * qd) / nrm, (g.y - q.y * qd) / nrm, (g.z - q.z * qd) / nrm, (g.w - q.w * qd) / nrm);
      } else {
        g = make_float4(g.x / nrm, g.y / nrm, g.z / nrm, g.w / nrm);
      }
    }
    PM_ST4(dz + r * dzp + lane * 4, g);
  }
}

// update: one block (256 threads = D channels) per slot
__global__ __launch_bounds__(256) void mem_write_update_kernel(const float* __restrict__ mem, const float* __restrict__ nomden, int m, float mu,
                                                               float* __restrict__ out, float* __restrict__ u_out) {
  __shared__ float red[4];
  const int j = blockIdx.x, c = threadIdx.x;
  const float den = nomden[(m + 1) * D + j];
  const float old = mem[j * D + c];
  const float u = den != 0.f ? mu * old + (1.f - mu) * nomden[j * D + c] / den : old;
  float ss = pm_wave_sum(u * u);
  if ((c & 63) == 0) red[c >> 6] = ss;
  __syncthreads();
  const float nrm = fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), EPS);
  out[j * D + c] = u / nrm;
  if (u_out) u_out[j * D + c] = u;
}

__global__ __launch_bounds__(256) void mem_write_update_bwd_kernel(const float* __restrict__ u, const float* __restrict__ nomden, int m, float mu,
                                                                   const float* __restrict__ dout, float* __restrict__ dnom, float* __restrict__ dmem_in) {
  __shared__ float red[2][4];
  const int j = blockIdx.x, c = threadIdx.x;
  if (j == m) {  // ignore-class row never reaches the memory
    dnom[j * D + c] = 0.f;
    return;
  }
  const float uv = u[j * D + c], g = dout[j * D + c];
  const float ss = pm_wave_sum(uv * uv), ug = pm_wave_sum(uv * g);
  if ((c & 63) == 0) red[0][c >> 6] = ss, red[1][c >> 6] = ug;
  __syncthreads();
  const float n0 = sqrtf((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
  const float dot = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float nrm = fmaxf(n0, EPS);
  const float du = n0 >= EPS ? (g - (uv / nrm) * (dot / nrm)) / nrm : g / nrm;
  const float den = nomden[(m + 1) * D + j];
  dnom[j * D + c] = den != 0.f ? du * (1.f - mu) / den : 0.f;
  if (dmem_in) dmem_in[j * D + c] = den != 0.f ? du * mu : du;
}

// Memory read on the matrix cores: a 256-thread block (4 waves) owns 32 query rows, and x is fetched from HBM exactly once.
//   * the 32 rows are loaded as 32 fully coalesced 1 KB wave instructions (8 per wave, all in flight at once) into LDS (Xs, 33 KB);
//   * wave w owns the channels [64 w, 64 w + 64): it forms the PARTIAL score tile over its quarter of the reduction (32 MFMAs instead
//     of the 128 a lone wave would chain -- the single-wave version of this kernel was bound by exactly that latency: 208 dependent
//     MFMAs = 5.5 us per tile with 2.25 waves per CU), parks it in LDS, and after one barrier every wave adds the four partials in the
//     same fixed order, so all of them hold bit-identical scores;
//   * both products run TRANSPOSED (operands swapped): S^T = M X^T and agg^T = M^T P^T, so the MFMA result layout hands lane
//     (row = lane & 31, half = lane >> 5) 16 of its OWN row's 32 (padded) slots / four consecutive output channels per register quad:
//     the softmax over the slots is 16 in-lane values + one exchange with the partner half-lane, P is already the B operand of the
//     second product (each wave produces its own 64 channels of it);
//   * the 19 x 256 memory is read straight from L1 / L2 in fragment layout (19 KB, hot in every CU) and prefetched ahead of the x rows;
//   * qhat = x / ||x|| is produced from the LDS copy, 8 rows per wave; [agg] is transposed through the same LDS rows and leaves as
//     whole 1 KB rows as well.
// 50 KB of LDS per block: three blocks (12 waves) per CU, the 576 blocks of the flagship (18 432 rows) are resident at once.
constexpr int MR_LDK = D + 4;     // 1040-byte rows: conflict-free ds_read_b128 fragments (bank step 4 per row)
constexpr int MR_SP = 17 * 64;    // per wave: 16 partial-score registers + the partial squared norm, one float per lane each
template <int M_, bool COLPART = false>
__global__ __launch_bounds__(256, 3) void mem_read_fwd_mfma_kernel(const float* __restrict__ x, long xp, long rows, const float* __restrict__ mem, int m_rt,
                                                                const float* __restrict__ noise, float* __restrict__ qr, long qp,
                                                                float* __restrict__ score, float* __restrict__ pm,
                                                                const float* __restrict__ noise_q, float* __restrict__ colpart) {
  constexpr int LDK = MR_LDK;
  constexpr int MM = M_ > 0 ? M_ : MAXM;
  const int M = M_ > 0 ? M_ : m_rt;
  extern __shared__ __align__(16) float mr_smem[];
  float* Xs = mr_smem;                  // [32][LDK]
  float* Sp = mr_smem + 32 * LDK;       // [4][17][64]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int mslot = min(l31, M - 1);                       // lanes beyond the slots read a valid row and multiply by zero below
  const float mz = l31 < M ? 1.f : 0.f;
  // this wave's fragments of the memory: S^T operand (slot = lane & 31, 4 consecutive channels per 8-channel group of its 64) ...
  float4 mf[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    mf[g] = PM_LD4(mem + (long)mslot * D + 64 * w + 8 * g + 4 * half);
    mf[g].x *= mz, mf[g].y *= mz, mf[g].z *= mz, mf[g].w *= mz;
  }
  // ... and agg^T operand (channel = 64 w + 32 n + (lane & 31); step q multiplies slot (q&3) + 8 (q>>2) [+ 4 in the upper half-wave])
  float ma[2][16];
  auto load_ma = [&](int opaque) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if ((q & 3) + 8 * (q >> 2) >= MM) continue;             // both slots of this step are padding
      const float* bp = mem + (long)min((q & 3) + 8 * (q >> 2) + 4 * half, M - 1) * D + 64 * w + l31 + opaque;
      ma[0][q] = bp[0], ma[1][q] = bp[32];
    }
  };
  if constexpr (!COLPART) load_ma(0);     // with the column partials these 32 registers are fetched after them (the kernel sits exactly at its 168-register budget)
  for (long row0 = (long)blockIdx.x * 32; row0 < rows; row0 += (long)gridDim.x * 32) {
    if (row0 != (long)blockIdx.x * 32) __syncthreads();    // the previous tile's qhat pass is done with Xs
    {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = PM_LD4(x + min(row0 + 8 * w + u, rows - 1) * xp + lane * 4);   // rows past the end: a copy of the last row, never stored
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(Xs + (8 * w + u) * LDK + lane * 4) = v[u];
    }
    __syncthreads();
    const long myrow = min(row0 + l31, rows - 1);
    const bool live = row0 + l31 < rows;
    mr_f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float n2 = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(Xs + l31 * LDK + 64 * w + 8 * g + 4 * half);      // lane & 31 = query row
      n2 += dot4(a, a);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[g].x, a.x, acc, 0, 0, 0);                              // S^T[slot][row]: operands swapped
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[g].y, a.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[g].z, a.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[g].w, a.w, acc, 0, 0, 0);
    }
    n2 += __shfl_xor(n2, 32, 64);
#pragma unroll
    for (int q = 0; q < 16; ++q) Sp[w * MR_SP + q * 64 + lane] = acc[q];
    Sp[w * MR_SP + 16 * 64 + lane] = n2;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = ((Sp[q * 64 + lane] + Sp[MR_SP + q * 64 + lane]) + Sp[2 * MR_SP + q * 64 + lane]) + Sp[3 * MR_SP + q * 64 + lane];
    n2 = ((Sp[16 * 64 + lane] + Sp[MR_SP + 16 * 64 + lane]) + Sp[2 * MR_SP + 16 * 64 + lane]) + Sp[3 * MR_SP + 16 * 64 + lane];
    const float rnrm = 1.f / fmaxf(sqrtf(n2), EPS);      // one division per row; qhat and the scores multiply by it (<= 1 ulp from x / ||x||)
    // acc[q] = <x_row, m_slot> for slot s = (q & 3) + 8 (q >> 2) + 4 half of this lane's own row
    if constexpr (COLPART) {
      // per-slot (max, sum exp) of this tile's 32 rows for the softmax over ALL queries (memory.py:186): every wave holds the same scores, wave w takes the
      // register quads q with (q & 3) == w (at most three), rows are the 32 lanes of a half-wave -> butterfly within the half. Replaces the first of the
      // two column-softmax launches; the partial layout is mem_colsoftmax_partial_kernel's with 32-row instead of 128-row tiles.
      const long tile = row0 >> 5;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        if (8 * qq >= MM) continue;                                       // compile-time: register quads beyond the slots
        const float a = w == 0 ? acc[4 * qq] : (w == 1 ? acc[4 * qq + 1] : (w == 2 ? acc[4 * qq + 2] : acc[4 * qq + 3]));      // wave-uniform selects
        const int sl = w + 8 * qq + 4 * half;
        const bool ok = sl < M;
        float cv = -INFINITY;
        if (ok && live) cv = a * rnrm + (noise_q ? noise_q[myrow * M + sl] : 0.f);
        float cm = cv;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) cm = fmaxf(cm, __shfl_xor(cm, o, 64));
        float cs = cv == -INFINITY ? 0.f : expf(cv - cm);
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) cs += __shfl_xor(cs, o, 64);
        if (ok && l31 == 0) *reinterpret_cast<float2*>(colpart + (tile * M + sl) * 2) = make_float2(cm, cs);
      }
    }
    if constexpr (COLPART) {
      int opaque = 0;
      asm volatile("" : "+v"(opaque));      // keeps the loads below from being hoisted back in front of the tile loop
      load_ma(opaque);
    }
    float pr[16];
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int sl = (q & 3) + 8 * (q >> 2) + 4 * half;
      if ((q & 3) + 8 * (q >> 2) < MM && sl < M) {          // first test: compile-time pruning of register quads beyond the slots
        float sv = acc[q] * rnrm;
        if (live && w == 0) score[myrow * M + sl] = sv;
        if (noise) sv += noise[myrow * M + sl];
        pr[q] = sv;
        mx = fmaxf(mx, sv);
      } else {
        pr[q] = -INFINITY;
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float se = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) pr[q] = pr[q] == -INFINITY ? 0.f : expf(pr[q] - mx), se += pr[q];
    se += __shfl_xor(se, 32, 64);
    const float rse = 1.f / se;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int sl = (q & 3) + 8 * (q >> 2) + 4 * half;
      pr[q] = pr[q] * rse;
      if ((q & 3) + 8 * (q >> 2) < MM && sl < M && live && w == 1) pm[myrow * M + sl] = pr[q];
    }
    // qhat = x / ||x|| from the LDS copy, one row (1 KB) per wave instruction; the row's norm lives in lane (row & 31)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long r = row0 + 8 * w + u;
      const float nr = __shfl(rnrm, 8 * w + u, 64);
      const float4 v = *reinterpret_cast<const float4*>(Xs + (8 * w + u) * LDK + lane * 4);
      if (r < rows) PM_ST4(qr + r * qp + lane * 4, make_float4(v.x * nr, v.y * nr, v.z * nr, v.w * nr));
    }
    __syncthreads();                                       // every wave is done with the x rows: Xs becomes the [32][256] staging tile of agg
    // agg^T = M^T P^T for this wave's 64 channels; padding slots carry P = 0. The register quads (4 consecutive channels of the lane's own
    // row) are parked in LDS and leave as whole 1 KB rows, 8 rows per wave: full-line stores instead of 32-byte pieces of 32 rows.
    {
      mr_f32x16 ag[2];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) ag[n][q] = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if ((q & 3) + 8 * (q >> 2) >= MM) continue;
#pragma unroll
        for (int n = 0; n < 2; ++n) ag[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ma[n][q], pr[q], ag[n], 0, 0, 0);
      }
      float* dst = Xs + l31 * LDK + 64 * w + 4 * half;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<float4*>(dst + n * 32 + 8 * g4) = make_float4(ag[n][4 * g4], ag[n][4 * g4 + 1], ag[n][4 * g4 + 2], ag[n][4 * g4 + 3]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long r = row0 + 8 * w + u;
      const float4 v = *reinterpret_cast<const float4*>(Xs + (8 * w + u) * LDK + lane * 4);
      if (r < rows) PM_ST4(qr + r * qp + D + lane * 4, v);
    }
  }
}

// Memory read backward on the matrix cores (dx only; the dmem variant keeps the row kernel above). Same decomposition as the forward kernel:
// a 256-thread block owns 32 query rows, wave w the channels [64 w, 64 w + 64). Lane (row = lane & 31, half = lane >> 5) holds, for
// g = 0..7, the four channels 64 w + 8 g + 4 half + (0..3) of ITS row of x, dq and dr -- which is both the B-operand layout of
// dP^T = M dR^T (reduction split over the four waves, partials added in fixed order after one barrier) and the result layout of
// add^T = M^T dS^T, so everything between the two products (softmax backward over the slots, the extra score gradient) and after them
// (dq + add, the projection off q of the normalisation backward, whose row dot needs one more cross-wave sum) stays in registers.
// dx leaves as whole 1 KB rows through an LDS tile. 220 VGPRs (three row fragments of 32 floats + the memory fragment): two blocks per CU; capping
// the registers for three spills and is slower (30.5 vs 24.1 us). 78.3 MB in 24 us = 3.25 TB/s (49 us for the row kernel).
template <int M_>
__global__ __launch_bounds__(256, 2) void mem_read_bwd_mfma_kernel(const float* __restrict__ x, long xp, long rows, const float* __restrict__ mem, int m_rt,
                                                                   const float* __restrict__ pm, const float* __restrict__ dqr, long dqp,
                                                                   const float* __restrict__ dsx, float* __restrict__ dx, long dxp) {
  constexpr int LDK = MR_LDK;
  constexpr int MM = M_ > 0 ? M_ : MAXM;
  const int M = M_ > 0 ? M_ : m_rt;
  extern __shared__ __align__(16) float mr_smem[];
  float* Os = mr_smem;                  // [32][LDK]: dx tile
  float* Sp = mr_smem + 32 * LDK;       // [4][17][64]: partial dP quads + partial squared norm / partial row dot
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int mslot = min(l31, M - 1);
  const float mz = l31 < M ? 1.f : 0.f;
  float4 mf[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    mf[g] = PM_LD4(mem + (long)mslot * D + 64 * w + 8 * g + 4 * half);
    mf[g].x *= mz, mf[g].y *= mz, mf[g].z *= mz, mf[g].w *= mz;
  }
  for (long row0 = (long)blockIdx.x * 32; row0 < rows; row0 += (long)gridDim.x * 32) {
    if (row0 != (long)blockIdx.x * 32) __syncthreads();    // the previous tile's row stores are done with Os / Sp
    const long myrow = min(row0 + l31, rows - 1);
    const bool live = row0 + l31 < rows;
    float4 xv[8], dq[8], dr[8];
    {
      const float* xr = x + myrow * xp + 64 * w + 4 * half;
      const float* qr = dqr + myrow * dqp + 64 * w + 4 * half;
#pragma unroll
      for (int g = 0; g < 8; ++g) dr[g] = PM_LD4(qr + D + 8 * g);
#pragma unroll
      for (int g = 0; g < 8; ++g) xv[g] = PM_LD4(xr + 8 * g);
#pragma unroll
      for (int g = 0; g < 8; ++g) dq[g] = PM_LD4(qr + 8 * g);
    }
    mr_f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[g].x, dr[g].x, acc, 0, 0, 0);       // dP^T[slot][row]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[g].y, dr[g].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[g].z, dr[g].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(mf[g].w, dr[g].w, acc, 0, 0, 0);
    }
    float n2 = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) n2 += dot4(xv[g], xv[g]);
    n2 += __shfl_xor(n2, 32, 64);
#pragma unroll
    for (int q = 0; q < 16; ++q) Sp[w * MR_SP + q * 64 + lane] = acc[q];
    Sp[w * MR_SP + 16 * 64 + lane] = n2;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = ((Sp[q * 64 + lane] + Sp[MR_SP + q * 64 + lane]) + Sp[2 * MR_SP + q * 64 + lane]) + Sp[3 * MR_SP + q * 64 + lane];
    n2 = ((Sp[16 * 64 + lane] + Sp[MR_SP + 16 * 64 + lane]) + Sp[2 * MR_SP + 16 * 64 + lane]) + Sp[3 * MR_SP + 16 * 64 + lane];
    const float n0 = sqrtf(n2);
    const float rn = 1.f / fmaxf(n0, EPS);
    // softmax backward over the slots of the lane's own row: ds = p (dp - sum_j p_j dp_j) + dsx
    float ds[16], pv[16];
    float pdp = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int sl = (q & 3) + 8 * (q >> 2) + 4 * half;
      pv[q] = 0.f;
      if ((q & 3) + 8 * (q >> 2) < MM && sl < M) {
        pv[q] = pm[myrow * M + sl];
        pdp += pv[q] * acc[q];
      }
    }
    pdp += __shfl_xor(pdp, 32, 64);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int sl = (q & 3) + 8 * (q >> 2) + 4 * half;
      ds[q] = 0.f;
      if ((q & 3) + 8 * (q >> 2) < MM && sl < M) ds[q] = pv[q] * (acc[q] - pdp) + (dsx ? dsx[myrow * M + sl] : 0.f);
    }
    __syncthreads();                                       // every wave has read the partials: Sp is free for the row dots
    // dq += dS M for this wave's 64 channels
    float qd = 0.f;
    {
      mr_f32x16 ag[2];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) ag[n][q] = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if ((q & 3) + 8 * (q >> 2) >= MM) continue;
        const float* bp = mem + (long)min((q & 3) + 8 * (q >> 2) + 4 * half, M - 1) * D + 64 * w + l31;
#pragma unroll
        for (int n = 0; n < 2; ++n) ag[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bp[32 * n], ds[q], ag[n], 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        dq[g].x += ag[g >> 2][4 * (g & 3)], dq[g].y += ag[g >> 2][4 * (g & 3) + 1], dq[g].z += ag[g >> 2][4 * (g & 3) + 2], dq[g].w += ag[g >> 2][4 * (g & 3) + 3];
        qd += dot4(xv[g], dq[g]);
      }
    }
    qd *= rn;                                              // q . dq with q = x / ||x||
    qd += __shfl_xor(qd, 32, 64);
    Sp[w * 64 + lane] = qd;
    __syncthreads();
    qd = ((Sp[lane] + Sp[64 + lane]) + Sp[128 + lane]) + Sp[192 + lane];
    // through qhat = x / max(|x|, eps): o = (dq - q (q . dq)) / ||x||  (the clamp is inactive for every row with a norm >= eps)
    {
      const float c = n0 >= EPS ? qd * rn : 0.f;
      float* dst = Os + l31 * LDK + 64 * w + 4 * half;
#pragma unroll
      for (int g = 0; g < 8; ++g)
        *reinterpret_cast<float4*>(dst + 8 * g) = make_float4((dq[g].x - xv[g].x * c) * rn, (dq[g].y - xv[g].y * c) * rn, (dq[g].z - xv[g].z * c) * rn, (dq[g].w - xv[g].w * c) * rn);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long r = row0 + 8 * w + u;
      const float4 v = *reinterpret_cast<const float4*>(Os + (8 * w + u) * LDK + lane * 4);
      if (r < rows) PM_ST4(dx + r * dxp + lane * 4, v);
    }
    (void)live;
  }
}

inline int row_blocks(long rows) { return (int)std::min<long>((rows + 3) / 4, 256 * 8); }
inline int accum_blocks(long rows) { return (int)std::min<long>((rows + 15) / 16, 256); }   // slab kernel: one block (4 wave slabs in LDS) per CU
inline int accum_mfma_blocks(long rows) { return (int)std::min<long>((rows + AM_ROWS - 1) / AM_ROWS, 512); }   // MFMA kernel: two resident blocks per CU (141 + 68 registers)
inline bool accum_slab() {   // PM_MEM_ACCUM_SLAB=1: the round-2 LDS-slab kernel (A/B)
  static const bool v = [] { const char* e = getenv("PM_MEM_ACCUM_SLAB"); return e && e[0] == '1'; }();
  return v;
}

}  // namespace

namespace {
int mem_read_fwd_launch(const pm_tensor* x, const float* mem, int m, const float* noise, const float* noise_q, const pm_tensor* qr, float* score, float* p_mem,
                        float* colpart, void* stream);
}
extern "C" int pm_mem_read_fwd(const pm_tensor* x, const float* mem, int m, const float* noise, const pm_tensor* qr, float* score, float* p_mem,
                               void* stream) {
  return mem_read_fwd_launch(x, mem, m, noise, nullptr, qr, score, p_mem, nullptr, stream);
}
namespace {
int mem_read_fwd_launch(const pm_tensor* x, const float* mem, int m, const float* noise, const float* noise_q, const pm_tensor* qr, float* score, float* p_mem,
                        float* colpart, void* stream) {
  PM_REQUIRE_F32(x, "mem_read_fwd");
  PM_REQUIRE_F32(qr, "mem_read_fwd");
  PM_REQUIRE(x && qr && mem && score && p_mem && x->ptr && qr->ptr, PM_EINVAL, "mem_read_fwd: null");
  PM_REQUIRE(x->c == D && qr->c == 2 * D && pm_vec_ok(x) && pm_vec_ok(qr) && pm_aligned16(mem), PM_EUNSUPPORTED, "mem_read_fwd: needs d == %d, aligned views", D);
  PM_REQUIRE(m >= 1 && m <= MAXM && pm_pixels(x) == pm_pixels(qr), PM_EINVAL, "mem_read_fwd: bad slots/rows");
  const long rows = pm_pixels(x);
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)std::min<long>((rows + 31) / 32, 256 * 3 * 4);
  const size_t lds = (size_t)(32 * MR_LDK + 4 * MR_SP) * sizeof(float);
#define PM_MR_LAUNCH(MM, CP)                                                                                                                              \
  hipLaunchKernelGGL((mem_read_fwd_mfma_kernel<MM, CP>), dim3(nb), dim3(256), lds, st, (const float*)x->ptr, (long)x->pitch, rows, mem, m, noise, (float*)qr->ptr, \
                     (long)qr->pitch, score, p_mem, noise_q, colpart)
  if (m == 19) {
    if (colpart) PM_MR_LAUNCH(19, true);
    else PM_MR_LAUNCH(19, false);
  } else {
    if (colpart) PM_MR_LAUNCH(0, true);
    else PM_MR_LAUNCH(0, false);
  }
#undef PM_MR_LAUNCH
  return pm_check_launch("mem_read_fwd");
}
}  // namespace

// read + the softmax over all queries in two launches: the read kernel leaves the per-tile column partials, the apply kernel merges them and normalises
extern "C" size_t pm_mem_read_fwd_pq_workspace(int64_t rows, int m) { return pm_align_up((size_t)pm_cdiv(rows, 32) * m * 2 * sizeof(float), 256); }
extern "C" int pm_mem_read_fwd_pq(const pm_tensor* x, const float* mem, int m, const float* noise, const float* noise_q, const pm_tensor* qr, float* score,
                                  float* p_mem, float* p_query, void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE(x && p_query && m >= 1 && m <= MAXM, PM_EINVAL, "mem_read_fwd_pq: bad args");
  const long rows = pm_pixels(x);
  PM_REQUIRE(ws && ws_bytes >= pm_mem_read_fwd_pq_workspace(rows, m), PM_EWORKSPACE, "mem_read_fwd_pq: workspace too small");
  if (int e = mem_read_fwd_launch(x, mem, m, noise, noise_q, qr, score, p_mem, (float*)ws, stream)) return e;
  hipLaunchKernelGGL(mem_colsoftmax_apply_kernel, dim3(pm_cdiv(rows, CSA_ROWS)), dim3(CSA_T), 0, (hipStream_t)stream, (const float*)score, noise_q, rows, m,
                     (const float*)ws, (int)pm_cdiv(rows, 32), p_query);
  return pm_check_launch("mem_read_fwd_pq");
}

extern "C" size_t pm_mem_colsoftmax_workspace(int64_t rows, int m) { return pm_align_up((size_t)pm_cdiv(rows, CS_ROWS) * m * 2 * sizeof(float), 256); }
extern "C" int pm_mem_colsoftmax(const float* score, const float* noise, int64_t rows, int m, float* p_query, void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE(score && p_query && rows > 0 && m >= 1 && m <= MAXM, PM_EINVAL, "mem_colsoftmax: bad args");
  PM_REQUIRE(ws && ws_bytes >= pm_mem_colsoftmax_workspace(rows, m), PM_EWORKSPACE, "mem_colsoftmax: workspace too small");
  const int nb = pm_cdiv(rows, CS_ROWS);
  hipLaunchKernelGGL(mem_colsoftmax_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, score, noise, (long)rows, m, (float*)ws);
  hipLaunchKernelGGL(mem_colsoftmax_apply_kernel, dim3(pm_cdiv(rows, CSA_ROWS)), dim3(CSA_T), 0, (hipStream_t)stream, score, noise, (long)rows, m, (const float*)ws, nb,
                     p_query);
  return pm_check_launch("mem_colsoftmax");
}

extern "C" size_t pm_mem_read_bwd_workspace(int64_t rows, int m, int d) { return pm_align_up((size_t)row_blocks(rows) * m * d * sizeof(float), 256); }
extern "C" int pm_mem_read_bwd(const pm_tensor* x, const float* mem, int m, const float* p_mem, const pm_tensor* dqr, const float* dsx,
                               const pm_tensor* dx, float* dmem, void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE_F32(x, "mem_read_bwd");
  PM_REQUIRE_F32(dqr, "mem_read_bwd");
  PM_REQUIRE_F32(dx, "mem_read_bwd");
  PM_REQUIRE(x && dqr && dx && mem && p_mem && x->ptr && dqr->ptr && dx->ptr, PM_EINVAL, "mem_read_bwd: null");
  PM_REQUIRE(x->c == D && dqr->c == 2 * D && dx->c == D && pm_vec_ok(x) && pm_vec_ok(dqr) && pm_vec_ok(dx) && pm_aligned16(mem), PM_EUNSUPPORTED,
             "mem_read_bwd: needs d == %d, aligned views", D);
  PM_REQUIRE(m >= 1 && m <= MAXM, PM_EINVAL, "mem_read_bwd: bad slots");
  const long rows = pm_pixels(x);
  const int nb = row_blocks(rows);
  hipStream_t st = (hipStream_t)stream;
  if (dmem) {
    PM_REQUIRE(ws && ws_bytes >= pm_mem_read_bwd_workspace(rows, m, D), PM_EWORKSPACE, "mem_read_bwd: workspace too small");
    if (m == 19)
      hipLaunchKernelGGL((mem_read_bwd_kernel<19, true>), dim3(nb), dim3(256), 0, st, (const float*)x->ptr, (long)x->pitch, rows, mem, m, p_mem,
                         (const float*)dqr->ptr, (long)dqr->pitch, dsx, (float*)dx->ptr, (long)dx->pitch, (float*)ws);
    else
      hipLaunchKernelGGL((mem_read_bwd_kernel<0, true>), dim3(nb), dim3(256), 0, st, (const float*)x->ptr, (long)x->pitch, rows, mem, m, p_mem,
                         (const float*)dqr->ptr, (long)dqr->pitch, dsx, (float*)dx->ptr, (long)dx->pitch, (float*)ws);
    const long n = (long)m * D;
    launch_reduce_partials((const float*)ws, nb, n, dmem, st);
  } else if (getenv("PM_MEM_BWD_ROWS") == nullptr) {
    const int nt = (int)std::min<long>((rows + 31) / 32, 256 * 3 * 4);
    const size_t lds = (size_t)(32 * MR_LDK + 4 * MR_SP) * sizeof(float);
    if (m == 19)
      hipLaunchKernelGGL(mem_read_bwd_mfma_kernel<19>, dim3(nt), dim3(256), lds, st, (const float*)x->ptr, (long)x->pitch, rows, mem, m, p_mem,
                         (const float*)dqr->ptr, (long)dqr->pitch, dsx, (float*)dx->ptr, (long)dx->pitch);
    else
      hipLaunchKernelGGL(mem_read_bwd_mfma_kernel<0>, dim3(nt), dim3(256), lds, st, (const float*)x->ptr, (long)x->pitch, rows, mem, m, p_mem,
                         (const float*)dqr->ptr, (long)dqr->pitch, dsx, (float*)dx->ptr, (long)dx->pitch);
  } else {
    if (m == 19)
      hipLaunchKernelGGL((mem_read_bwd_kernel<19, false>), dim3(nb), dim3(256), 0, st, (const float*)x->ptr, (long)x->pitch, rows, mem, m, p_mem,
                         (const float*)dqr->ptr, (long)dqr->pitch, dsx, (float*)dx->ptr, (long)dx->pitch, (float*)nullptr);
    else
      hipLaunchKernelGGL((mem_read_bwd_kernel<0, false>), dim3(nb), dim3(256), 0, st, (const float*)x->ptr, (long)x->pitch, rows, mem, m, p_mem,
                         (const float*)dqr->ptr, (long)dqr->pitch, dsx, (float*)dx->ptr, (long)dx->pitch, (float*)nullptr);
  }
  return pm_check_launch("mem_read_bwd");
}

extern "C" size_t pm_mem_write_accum_workspace(const pm_tensor* z, int m) {
  if ((z && !pm_is_f32(z))) return 0;      // fp32 tensors only
  return pm_align_up((size_t)std::max(accum_blocks(pm_pixels(z)), accum_mfma_blocks(pm_pixels(z))) * ((m + 1) * (D + 1)) * sizeof(float), 256);
}
extern "C" int pm_mem_write_accum(const pm_tensor* z, const int64_t* labels, int H, int W, int m, int normalize, float* nomden, void* ws, size_t ws_bytes,
                                  void* stream) {
  PM_REQUIRE_F32(z, "mem_write_accum");
  PM_REQUIRE(z && z->ptr && labels && nomden, PM_EINVAL, "mem_write_accum: null");
  PM_REQUIRE(z->c == D && pm_vec_ok(z) && m >= 1 && m + 1 <= MAXM, PM_EUNSUPPORTED, "mem_write_accum: needs d == %d and <= %d slots", D, MAXM - 1);
  PM_REQUIRE(ws && ws_bytes >= pm_mem_write_accum_workspace(z, m), PM_EWORKSPACE, "mem_write_accum: workspace too small");
  const long rows = pm_pixels(z);
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)(m + 1) * (D + 1);
  if (!accum_slab() && rows < (1l << 31)) {
    const int nb = accum_mfma_blocks(rows);
    hipLaunchKernelGGL(mem_write_accum_mfma_kernel, dim3(nb), dim3(256), 0, st, (const float*)z->ptr, (long)z->pitch, z->n, z->h, z->w, labels, H, W, m, normalize,
                       pm_ac_scale(H, z->h), pm_ac_scale(W, z->w), (float*)ws);
    launch_reduce_partials((const float*)ws, nb, n, nomden, st);
    return pm_check_launch("mem_write_accum");
  }
  const int nb = accum_blocks(rows);
  const size_t lds = (size_t)ACC_W * ((m + 1) * D + MAXM) * sizeof(float);
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mem_write_accum_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL(mem_write_accum_kernel, dim3(nb), dim3(256), lds, st, (const float*)z->ptr, (long)z->pitch, z->n, z->h, z->w, labels, H, W, m, normalize,
                     pm_ac_scale(H, z->h), pm_ac_scale(W, z->w), (float*)ws);
  launch_reduce_partials((const float*)ws, nb, n, nomden, st);
  return pm_check_launch("mem_write_accum");
}

extern "C" int pm_mem_write_accum_bwd(const pm_tensor* z, const int64_t* labels, int H, int W, int m, int normalize, const float* dnom,
                                      const pm_tensor* dz, void* stream) {
  PM_REQUIRE_F32(z, "mem_write_accum_bwd");
  PM_REQUIRE_F32(dz, "mem_write_accum_bwd");
  PM_REQUIRE(z && dz && z->ptr && dz->ptr && labels && dnom && pm_aligned16(dnom), PM_EINVAL, "mem_write_accum_bwd: null/unaligned");
  PM_REQUIRE(z->c == D && dz->c == D && pm_vec_ok(z) && pm_vec_ok(dz) && m >= 1 && m + 1 <= MAXM, PM_EUNSUPPORTED, "mem_write_accum_bwd: needs d == %d", D);
  const long rows = pm_pixels(z);
  hipLaunchKernelGGL(mem_write_accum_bwd_kernel, dim3(row_blocks(rows)), dim3(256), 0, (hipStream_t)stream, (const float*)z->ptr, (long)z->pitch, z->n, z->h,
                     z->w, labels, H, W, m, normalize, pm_ac_scale(H, z->h), pm_ac_scale(W, z->w), dnom, (float*)dz->ptr, (long)dz->pitch);
  return pm_check_launch("mem_write_accum_bwd");
}

extern "C" int pm_mem_write_update(const float* mem, const float* nomden, int m, int d, float momentum, float* mem_out, float* u_out, void* stream) {
  PM_REQUIRE(mem && nomden && mem_out && d == D && m >= 1 && m + 1 <= MAXM, PM_EINVAL, "mem_write_update: bad args (d must be %d)", D);
  hipLaunchKernelGGL(mem_write_update_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, mem, nomden, m, momentum, mem_out, u_out);
  return pm_check_launch("mem_write_update");
}

extern "C" int pm_mem_write_update_bwd(const float* u, const float* nomden, int m, int d, float momentum, const float* dmem_out, float* dnom, float* dmem_in,
                                       void* stream) {
  PM_REQUIRE(u && nomden && dmem_out && dnom && d == D && m >= 1 && m + 1 <= MAXM, PM_EINVAL, "mem_write_update_bwd: bad args (d must be %d)", D);
  hipLaunchKernelGGL(mem_write_update_bwd_kernel, dim3(m + 1), dim3(256), 0, (hipStream_t)stream, u, nomden, m, momentum, dmem_out, dnom, dmem_in);
  return pm_check_launch("mem_write_update_bwd");
}
