// K6: fused bilinear-upsample(align_corners=True) + CrossEntropy(ignore_index=255, reduction='mean').
// Replaces  Upsample(dec2, x_size) -> criterion  (/root/reference/network/deepv3plus.py:575-578), the aux loss
// (:589-595, same-size case) and the read loss  F.interpolate(score/T) -> celoss  (/root/reference/network/memory.py:173-176)
// WITHOUT materialising the [B,19,H,W] upsampled logits (358 MB at bs=8, 768^2): the low-res logits stay L2-resident
// and each hi-res pixel interpolates its 19 classes in registers. HBM-bound on the int64 labels (37.7 MB).
// Forward: block partial (sum, count) -> fixed-order final reduce (deterministic).
// Backward: the transposed bilinear operator is separable, so the gradient is gathered in two passes without atomics
// (deterministic): (1) per hi-res row, softmax - onehot of every hi-res pixel once (staged in LDS) reduced over its columns to the
// low-res columns -> T[n][H][w][C]; (2) per low-res pixel, T reduced over the supporting hi-res rows.
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

// Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MB L2. With block id = hi-res row, every XCD would walk ALL low-res logit rows (the four
// hi-res rows between two low-res rows sit on four different XCDs): the 22 MB of main-loss logits were fetched 8 x per sweep (counters, round 4: 226 MB read for
// 60 MB of inputs). Bijective remap: consecutive hi-res rows -> one XCD, so a low-res row pair is fetched into one L2 and re-used there.
__device__ __forceinline__ int ce_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
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

// Row-staged forward: block = one hi-res row (Y, image b). The two low-res logit rows that row interpolates between are staged in LDS
// once (coalesced, pre-scaled by 1/T); every hi-res pixel then gathers its 4 x C taps from LDS instead of issuing 4 x C scattered global
// loads. The interpolation expression is the one of interp_logits / ce_bwd_rows_kernel, so the per-pixel logits carry the same bits.
template <int C_>
__global__ __launch_bounds__(256) void ce_fwd_rows_kernel(const CEGeom g, float* __restrict__ part) {
  extern __shared__ float L[];
  const int C = C_ > 0 ? C_ : g.C;
  const int CP = C | 1;
  const int bid = ce_xcd_remap(blockIdx.x, gridDim.x);
  const int Y = bid % g.H, b = bid / g.H;
  const pm_lerp ly = pm_ac_lerp(g.sy, Y, g.h);
  float* L0 = L;
  float* L1 = L + (size_t)g.w * CP;
  const int64_t* lrow = g.labels + ((long)b * g.H + Y) * g.W;
  int64_t labs[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) labs[u] = (int)threadIdx.x + 256 * u < g.W ? lrow[threadIdx.x + 256 * u] : 255;   // in flight while the rows are staged
  for (int i = threadIdx.x; i < g.w * C; i += 256) {
    const int xl = i / C, c = i - xl * C;
    L0[xl * CP + c] = g.logits[((long)(b * g.h + ly.i0) * g.w + xl) * g.lp + c] * g.inv_temp;
    L1[xl * CP + c] = g.logits[((long)(b * g.h + ly.i1) * g.w + xl) * g.lp + c] * g.inv_temp;
  }
  __syncthreads();
  float lsum = 0.f, lcnt = 0.f;
  for (int X = threadIdx.x, u = 0; X < g.W; X += 256, ++u) {
    const int64_t lab = u < 3 ? (u == 0 ? labs[0] : (u == 1 ? labs[1] : labs[2])) : lrow[X];
    if (lab == 255) continue;
    const pm_lerp lx = pm_ac_lerp(g.sx, X, g.w);
    const float *p00 = L0 + lx.i0 * CP, *p01 = L0 + lx.i1 * CP, *p10 = L1 + lx.i0 * CP, *p11 = L1 + lx.i1 * CP;
    float v[C_ > 0 ? C_ : MAXC];
    float mx = -INFINITY, vl = 0.f;
#pragma unroll
    for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c)
      if (c < C) {
        v[c] = ly.w0 * (lx.w0 * p00[c] + lx.w1 * p01[c]) + ly.w1 * (lx.w0 * p10[c] + lx.w1 * p11[c]);
        mx = fmaxf(mx, v[c]);
        if (c == (int)lab) vl = v[c];
      }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < (C_ > 0 ? C_ : MAXC); ++c)
      if (c < C) se += __expf(v[c] - mx);   // v_exp_f32 (~1 ulp per term, as in the backward pass): the libm expf was 2/3 of this kernel's ALU work
    lsum += (mx + logf(se)) - vl;
    lcnt += 1.f;
  }
  __shared__ float sm[2][4];
  lsum = pm_wave_sum(lsum);
  lcnt = pm_wave_sum(lcnt);
  if ((threadIdx.x & 63) == 0) sm[0][threadIdx.x >> 6] = lsum, sm[1][threadIdx.x >> 6] = lcnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[bid * 2] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
    part[bid * 2 + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
  }
}

// One block folds the per-block (loss sum, pixel count) partials in double: 1024 threads keep the chain of dependent loads short (the
// 64-thread version walked 96 L2 round trips per thread: 26 us for 50 KB), then a fixed binary tree in LDS.
constexpr int CE_FINAL_T = 1024;
__global__ __launch_bounds__(CE_FINAL_T) void ce_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  __shared__ double s[2][CE_FINAL_T];
  double a = 0.0, c = 0.0;
  for (int b = threadIdx.x; b < nb; b += CE_FINAL_T) {
    const float2 v = *reinterpret_cast<const float2*>(part + (long)b * 2);
    a += (double)v.x, c += (double)v.y;
  }
  s[0][threadIdx.x] = a, s[1][threadIdx.x] = c;
  __syncthreads();
  for (int w = CE_FINAL_T / 2; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) s[0][threadIdx.x] += s[0][threadIdx.x + w], s[1][threadIdx.x] += s[1][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(s[0][0] / s[1][0]);  // 0/0 -> NaN like torch when every pixel is ignored
    out[1] = (float)s[1][0];
  }
}

__host__ __device__ __forceinline__ void support(float scale, int i, int out, int& lo, int& hi) {   // hi-res indices that can touch low-res index i
  if (scale <= 0.f) {
    lo = 0, hi = out - 1;
    return;
  }
  const float inv = 1.f / scale;
  lo = max(0, (int)floorf(((float)i - 1.f) * inv) - 1);
  hi = min(out - 1, (int)ceilf(((float)i + 1.f) * inv) + 1);
}
__device__ __forceinline__ float tap_weight(const pm_lerp& l, int i) { return (l.i0 == i ? l.w0 : 0.f) + (l.i1 == i ? l.w1 : 0.f); }

// pass 2: thread per (low-res pixel, class): supporting hi-res rows in ascending order, then the loss scale
__global__ __launch_bounds__(256) void ce_bwd_cols_kernel(const CEGeom g, const float* __restrict__ T, const float* __restrict__ loss_out,
                                                          const float* __restrict__ gscale, float* __restrict__ dl, long dlp) {
  const int C = g.C;
  const long total = (long)g.n * g.h * g.w * C;
  const float gs = (gscale ? gscale[0] : 1.f) * g.inv_temp / loss_out[1];
  // consecutive logical blocks (neighbouring low-res rows, which share most of their supporting field rows) on one XCD's L2
  for (long i = (long)ce_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long q = i / C;
    const int x = (int)(q % g.w);
    q /= g.w;
    const int y = (int)(q % g.h), b = (int)(q / g.h);
    int lo, hi;
    support(g.sy, y, g.H, lo, hi);
    float acc = 0.f;
    for (int Y = lo; Y <= hi; ++Y) acc += tap_weight(pm_ac_lerp(g.sy, Y, g.h), y) * T[(((long)b * g.H + Y) * g.w + x) * C + c];
    dl[((long)(b * g.h + y) * g.w + x) * dlp + c] = acc * gs;
  }
}

// ---- forward and the first backward pass in ONE sweep over the labels ------------------------------------------------------------
// The forward already evaluates the softmax of every hi-res pixel; softmax - onehot is all the backward needs from the labels and the
// up-sampled logits, and the upstream scale is a scalar that can be applied last. So the training forward leaves the column-reduced field
//   T[b][Y][x][c] = sum over hi-res columns X of (softmax - onehot)(b, Y, X)[c] * (bilinear weight of X on low-res column x)
// behind, and the backward is the row pass (ce_bwd_cols_kernel) alone: the labels and the logits are read once per step instead of twice.
// Block = one hi-res row (Y, image b): the label row (as bytes) and the two low-res logit rows it interpolates between are staged in LDS.
// Lane group (low-res column x, part p of PARTS) walks its share of the hi-res pixels whose left tap is x -- every pixel is evaluated exactly
// once -- and keeps two weighted sums per class: A (left-tap weight, stays in column x) and B (right-tap weight, goes to column x + 1).
// The PARTS lanes of a column are combined by a fixed butterfly, columns exchange B through LDS, and T leaves as one contiguous row.
// Fixed association order everywhere: run-to-run deterministic.
__device__ __forceinline__ int ce_first_ge(float sx, int k, int w, int W) {   // first hi-res column whose left tap i0 is >= k (i0 is non-decreasing)
  if (k <= 0) return 0;
  if (k > w - 1 || sx <= 0.f) return W;
  int X = (int)ceilf((float)k / sx);
  X = max(0, min(W, X));
  while (X > 0 && pm_ac_lerp(sx, X - 1, w).i0 >= k) --X;
  while (X < W && pm_ac_lerp(sx, X, w).i0 < k) ++X;
  return X;
}

template <int C_, int PARTS, bool WITH_T>
__global__ __launch_bounds__(256) void ce_fused_rows_kernel(const CEGeom g, float* __restrict__ part, float* __restrict__ T) {
  extern __shared__ float L[];
  __shared__ float carry[MAXC];   // B of the last column of a round, owed to the first column of the next (rows wider than one round only)
  const int C = C_ > 0 ? C_ : g.C;
  const int CP = C | 1;
  const int nt = (int)blockDim.x, tid = (int)threadIdx.x;
  const int bid = ce_xcd_remap(blockIdx.x, gridDim.x);
  const int Y = bid % g.H, b = bid / g.H;
  const pm_lerp ly = pm_ac_lerp(g.sy, Y, g.h);
  float* L0 = L;
  float* L1 = L + (size_t)g.w * CP;                                   // (+ C floats of slack behind it: bufB holds one column more than a round)
  unsigned char* lab8 = reinterpret_cast<unsigned char*>(L1 + (size_t)g.w * CP + C);
  const int64_t* lrow = g.labels + ((long)b * g.H + Y) * g.W;
  for (int X = tid; X < g.W; X += nt) {
    const int64_t l = lrow[X];
    lab8[X] = (l >= 0 && l < 254) ? (unsigned char)l : (l == 255 ? 255 : 254);   // 254: a label no class matches (as the unfused kernels treat it)
  }
  auto stage_rows = [&]() {
    for (int i = tid; i < g.w * C; i += nt) {
      const int xl = i / C, c = i - xl * C;
      L0[xl * CP + c] = g.logits[((long)(b * g.h + ly.i0) * g.w + xl) * g.lp + c] * g.inv_temp;
      L1[xl * CP + c] = g.logits[((long)(b * g.h + ly.i1) * g.w + xl) * g.lp + c] * g.inv_temp;
    }
  };
  stage_rows();
  __syncthreads();
  constexpr int CR = C_ > 0 ? C_ : MAXC;
  float lsum = 0.f, lcnt = 0.f;
  const int R = nt / PARTS;                       // low-res columns per round
  const int rounds = (g.w + R - 1) / R;
  for (int rd = 0; rd < rounds; ++rd) {
    if (rd > 0) {                                 // the previous round recycled the staged rows as its exchange buffers
      __syncthreads();
      stage_rows();
      __syncthreads();
    }
    const int xr0 = rd * R, xr1 = min(g.w, xr0 + R);
    const int x = xr0 + tid / PARTS, p = tid % PARTS;
    const bool live = x < xr1;
    float A[CR], B[CR];
#pragma unroll
    for (int c = 0; c < CR; ++c) A[c] = B[c] = 0.f;
    if (live) {
      const int j0 = ce_first_ge(g.sx, x, g.w, g.W), j1 = ce_first_ge(g.sx, x + 1, g.w, g.W);
      const int per = (j1 - j0 + PARTS - 1) / PARTS;
      const int ja = j0 + p * per, jb = min(j1, ja + per);
      for (int X = ja; X < jb; ++X) {
        const int lab = lab8[X];
        if (lab == 255) continue;
        const pm_lerp lx = pm_ac_lerp(g.sx, X, g.w);
        const float *p00 = L0 + lx.i0 * CP, *p01 = L0 + lx.i1 * CP, *p10 = L1 + lx.i0 * CP, *p11 = L1 + lx.i1 * CP;
        float v[CR];
        float mx = -INFINITY, vl = 0.f;
#pragma unroll
        for (int c = 0; c < CR; ++c)
          if (c < C) {
            v[c] = ly.w0 * (lx.w0 * p00[c] + lx.w1 * p01[c]) + ly.w1 * (lx.w0 * p10[c] + lx.w1 * p11[c]);   // same expression as interp_logits
            mx = fmaxf(mx, v[c]);
            if (c == lab) vl = v[c];
          }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CR; ++c)
          if (c < C) v[c] = __expf(v[c] - mx), se += v[c];
        lsum += (mx + logf(se)) - vl;
        lcnt += 1.f;
        if constexpr (WITH_T) {
          const float inv = 1.f / se;
          const float wa = lx.i1 == lx.i0 ? lx.w0 + lx.w1 : lx.w0, wb = lx.i1 == lx.i0 ? 0.f : lx.w1;
#pragma unroll
          for (int c = 0; c < CR; ++c)
            if (c < C) {
              const float gq = v[c] * inv - (c == lab ? 1.f : 0.f);
              A[c] += wa * gq, B[c] += wb * gq;
            }
        }
      }
    }
    if constexpr (WITH_T) {
      if constexpr (PARTS > 1) {
#pragma unroll
        for (int o = 1; o < PARTS; o <<= 1)
#pragma unroll
          for (int c = 0; c < CR; ++c)
            if (c < C) A[c] += __shfl_xor(A[c], o, 64), B[c] += __shfl_xor(B[c], o, 64);
      }
      __syncthreads();                            // every lane is done with the staged rows: they become the exchange buffers
      float* bufA = L0;                           // A of column x         -> bufA[x - xr0]
      float* bufB = L1;                           // B of column x (owed to column x + 1) -> bufB[x - xr0 + 1]; bufB[0] = carry of the previous round
      if (live && p == 0) {
#pragma unroll
        for (int c = 0; c < CR; ++c)
          if (c < C) {
            bufA[(x - xr0) * C + c] = A[c];
            bufB[(x - xr0 + 1) * C + c] = B[c];
          }
      }
      if (tid < C) bufB[tid] = rd == 0 ? 0.f : carry[tid];
      __syncthreads();
      float* Trow = T + (((long)b * g.H + Y) * g.w + xr0) * C;
      for (int i = tid; i < (xr1 - xr0) * C; i += nt) Trow[i] = bufB[i] + bufA[i];
      if (tid < C) carry[tid] = bufB[(xr1 - xr0) * C + tid];
    }
  }
  __shared__ float sm[2][4];
  lsum = pm_wave_sum(lsum);
  lcnt = pm_wave_sum(lcnt);
  if ((tid & 63) == 0) sm[0][tid >> 6] = lsum, sm[1][tid >> 6] = lcnt;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, c = 0.f;
    for (int wv = 0; wv < (nt >> 6); ++wv) a += sm[0][wv], c += sm[1][wv];
    part[bid * 2] = a;
    part[bid * 2 + 1] = c;
  }
}

// ---- the same sweep with the ROW half of the transposed bilinear operator folded in (round 4, second session) ------------------------------------------------
// ce_fused_rows_kernel leaves one column-reduced row per HI-RES row (T[n][H][w][C]: 89.7 MB for the main loss) and the backward gathers ~8 of them per low-res
// pixel. Here a block owns a low-res row INTERVAL y -- every hi-res row whose upper tap is y (four of them at 192 -> 768) -- and folds their column-reduced rows with
// the two row weights straight away: FA[b][y] = sum_Y w0(Y) T_Y (stays in low-res row y), FB[b][y] = sum_Y w1(Y) T_Y (owed to row y + 1). The field is two
// low-res-sized rows per interval (2 x 22.4 MB) and the backward is dl[y] = (FA[y] + FB[y - 1]) x scale, elementwise.
//   * block = RG row groups x TPR threads; group gi takes rows Ya + gi, Ya + gi + RG, ...; lane x of a group = low-res column x (the form for modest up-sampling
//     ratios: one lane per column, PARTS == 1), walking the hi-res pixels whose left tap is x exactly as the kernel above (same expressions, same order);
//   * the right-tap sums B go to the neighbouring lane by a one-lane shuffle (wave edges through LDS) instead of an LDS round trip of the whole row;
//   * every round the groups fold their row into the block's LDS accumulators accA / accB in group order (one barrier per group: per-lane register accumulators
//     for all rounds put the 1 024-thread block 27 registers over its 128-register budget and the spills into the pixel loop: 190 instead of 120 us), loss
//     partials per block: all fixed-order, deterministic.
__host__ __device__ __forceinline__ int ce2_rows_per_interval(int h, int H) { return (H + h - 1) / h + 1; }
template <int C_>
__global__ __launch_bounds__(1024) void ce_fused_rows2_kernel(const CEGeom g, float* __restrict__ part, float* __restrict__ FA, float* __restrict__ FB, int tpr, int rg, int nseg, int ws) {
  extern __shared__ float L[];
  __shared__ float sm2[2][16];
  constexpr int CR = C_ > 0 ? C_ : MAXC;
  const int C = C_ > 0 ? C_ : g.C;
  const int CP = C | 1;
  const int tid = (int)threadIdx.x, nt = (int)blockDim.x;
  const int gi = tid / tpr, x = tid - gi * tpr, lane = tid & 63, wv = x >> 6;       // group, low-res column (lane of the group), wave of the group
  const int nwv = tpr >> 6;
  // block = (image b, interval y, column segment seg): the segment owns the low-res columns [c0, c0 + ncol); lane x of a group is column c0 - 1 + x, i.e. lane 0 is
  // the column LEFT of the segment -- it evaluates its hi-res pixels only for the right-tap sums B they owe to column c0 (4 % of redundant work at 96 columns per
  // segment; its loss terms and its own row entries belong to the neighbouring segment and are dropped here). Half the columns per block = half the LDS: four
  // resident blocks per CU instead of two.
  const int bid = ce_xcd_remap(blockIdx.x, gridDim.x);
  const int seg = bid % nseg, y = (bid / nseg) % g.h, b = bid / (nseg * g.h);
  const int y1 = min(y + 1, g.h - 1);
  const int c0 = seg * ws, ncol = min(ws, g.w - c0);
  const int cb = max(c0 - 1, 0), ce = min(c0 + ncol, g.w - 1), nst = ce - cb + 1;   // staged low-res columns [cb, ce]: the taps of the lanes' pixels
  const int Wp = (g.W + 15) & ~15;
  float* L0 = L;
  float* L1 = L + (size_t)(ws + 2) * CP;
  float* accA = L1 + (size_t)(ws + 2) * CP;                                         // [ws][C] x 2: the block's pieces of the FA / FB rows
  float* accB = accA + (size_t)ws * C;
  float* edge = accB + (size_t)ws * C;                                              // [rg][nwv][CR]: B of a wave's last lane, owed to the next wave's lane 0
  unsigned char* lab8 = reinterpret_cast<unsigned char*>(edge + (size_t)rg * nwv * CR);   // [rg][Wp]
  for (int i = tid; i < nst * C; i += nt) {
    const int xl = i / C, c = i - xl * C;
    L0[xl * CP + c] = g.logits[((long)(b * g.h + y) * g.w + cb + xl) * g.lp + c] * g.inv_temp;
    L1[xl * CP + c] = g.logits[((long)(b * g.h + y1) * g.w + cb + xl) * g.lp + c] * g.inv_temp;
  }
  for (int i = tid; i < ws * C; i += nt) accA[i] = 0.f, accB[i] = 0.f;
  const int Ya = ce_first_ge(g.sy, y, g.h, g.H), Yb = ce_first_ge(g.sy, y + 1, g.h, g.H);
  const int Xlo = ce_first_ge(g.sx, cb, g.w, g.W), Xhi = ce_first_ge(g.sx, c0 + ncol, g.w, g.W);   // hi-res columns whose left tap is one of the lanes' columns
  const int ov = nseg > 1 ? 1 : 0;                                                  // a single segment needs no neighbour lane
  const int xc = c0 - ov + x;                                                       // this lane's low-res column
  const bool own = x >= ov && x < ov + ncol;                                        // a column of the segment (with segments, lane 0: the left neighbour, B only)
  const bool colive = own || (ov && x == 0 && c0 > 0);
  int j0 = 0, j1 = 0;
  if (colive) j0 = ce_first_ge(g.sx, xc, g.w, g.W), j1 = ce_first_ge(g.sx, xc + 1, g.w, g.W);
  float lsum = 0.f, lcnt = 0.f;
  const int rounds = (Yb - Ya + rg - 1) / rg;
  for (int rd = 0; rd < rounds; ++rd) {
    const int Y = Ya + rd * rg + gi;
    const bool rowlive = Y < Yb;
    unsigned char* mylab = lab8 + (size_t)gi * Wp;
    if (rowlive) {
      const int64_t* lrow = g.labels + ((long)b * g.H + Y) * g.W;
      for (int X = Xlo + x; X < Xhi; X += tpr) {
        const int64_t l = lrow[X];
        mylab[X - Xlo] = (l >= 0 && l < 254) ? (unsigned char)l : (l == 255 ? 255 : 254);
      }
    }
    __syncthreads();                                 // labels (and, in the first round, the logit rows) are staged
    float A[CR], B[CR];
#pragma unroll
    for (int c = 0; c < CR; ++c) A[c] = B[c] = 0.f;
    pm_lerp ly = pm_ac_lerp(g.sy, rowlive ? Y : Ya, g.h);
    if (rowlive && colive) {
      for (int X = j0; X < j1; ++X) {
        const int lab = mylab[X - Xlo];
        if (lab == 255) continue;
        const pm_lerp lx = pm_ac_lerp(g.sx, X, g.w);
        const float *p00 = L0 + (lx.i0 - cb) * CP, *p01 = L0 + (lx.i1 - cb) * CP, *p10 = L1 + (lx.i0 - cb) * CP, *p11 = L1 + (lx.i1 - cb) * CP;
        float v[CR];
        float mx = -INFINITY, vl = 0.f;
#pragma unroll
        for (int c = 0; c < CR; ++c)
          if (c < C) {
            v[c] = ly.w0 * (lx.w0 * p00[c] + lx.w1 * p01[c]) + ly.w1 * (lx.w0 * p10[c] + lx.w1 * p11[c]);   // same expression as interp_logits
            mx = fmaxf(mx, v[c]);
            if (c == lab) vl = v[c];
          }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CR; ++c)
          if (c < C) v[c] = __expf(v[c] - mx), se += v[c];
        if (own) lsum += (mx + logf(se)) - vl, lcnt += 1.f;      // the left-neighbour lane's pixels are counted by the segment that owns them
        const float inv = 1.f / se;
        const float wa = lx.i1 == lx.i0 ? lx.w0 + lx.w1 : lx.w0, wb = lx.i1 == lx.i0 ? 0.f : lx.w1;
#pragma unroll
        for (int c = 0; c < CR; ++c)
          if (c < C) {
            const float gq = v[c] * inv - (c == lab ? 1.f : 0.f);
            A[c] += wa * gq, B[c] += wb * gq;
          }
      }
    }
    if (lane == 63) {
#pragma unroll
      for (int c = 0; c < CR; ++c)
        if (c < C) edge[((size_t)gi * nwv + wv) * CR + c] = B[c];
    }
    __syncthreads();                                 // wave-edge values are in LDS; every lane is done with this round's labels
    {
      const float wA = ly.i1 == ly.i0 ? ly.w0 + ly.w1 : ly.w0, wB = ly.i1 == ly.i0 ? 0.f : ly.w1;
#pragma unroll
      for (int c = 0; c < CR; ++c)
        if (c < C) {
          float bs = __shfl_up(B[c], 1, 64);
          if (lane == 0) bs = wv > 0 ? edge[((size_t)gi * nwv + wv - 1) * CR + c] : 0.f;
          A[c] = bs + A[c];                           // the column-reduced row entry T[Y][x][c] of the kernel above (same two terms)
        }
      for (int k = 0; k < rg; ++k) {                  // fold into the block's rows, group after group
        if (gi == k && rowlive && own) {
#pragma unroll
          for (int c = 0; c < CR; ++c)
            if (c < C) accA[(x - ov) * C + c] += wA * A[c], accB[(x - ov) * C + c] += wB * A[c];
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();                                    // (also the barrier between staging / zeroing and the stores when an interval has no row)
  float* outA = FA + (((long)b * g.h + y) * g.w + c0) * C;
  float* outB = FB + (((long)b * g.h + y) * g.w + c0) * C;
  for (int i = tid; i < ncol * C; i += nt) outA[i] = accA[i], outB[i] = accB[i];
  lsum = pm_wave_sum(lsum);
  lcnt = pm_wave_sum(lcnt);
  if ((tid & 63) == 0) sm2[0][tid >> 6] = lsum, sm2[1][tid >> 6] = lcnt;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, c = 0.f;
    for (int k = 0; k < (nt >> 6); ++k) a += sm2[0][k], c += sm2[1][k];
    part[bid * 2] = a;
    part[bid * 2 + 1] = c;
  }
}
// backward of the interval form: dl[b][y][x][c] = (FA[b][y][x][c] + FB[b][y - 1][x][c]) x upstream scale / valid pixels / T
__global__ __launch_bounds__(256) void ce_bwd_rows2_kernel(const CEGeom g, const float* __restrict__ FA, const float* __restrict__ FB, const float* __restrict__ loss_out,
                                                           const float* __restrict__ gscale, float* __restrict__ dl, long dlp) {
  const int C = g.C;
  const long rowsz = (long)g.w * C, total = (long)g.n * g.h * rowsz;
  const float gs = (gscale ? gscale[0] : 1.f) * g.inv_temp / loss_out[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / C;
    const int c = (int)(i - pix * C);
    const int y = (int)((pix / g.w) % g.h);
    const float v = FA[i] + (y > 0 ? FB[i - rowsz] : 0.f);
    dl[pix * dlp + c] = v * gs;
  }
}
constexpr size_t ROWS2_MAX_LDS = 120 * 1024;      // dynamic LDS of the interval form (opt-in beyond 64 KB, set per device at the first launch)
struct Rows2Plan {
  bool ok;
  int tpr, rg, nseg, ws;
  size_t lds;
};
inline Rows2Plan rows2_plan(const CEGeom& g) {
  static const bool off = [] { const char* e = getenv("PM_CE_ROWS2"); return e && e[0] == '0'; }();      // PM_CE_ROWS2=0: the one-row-per-hi-res-row field (A/B)
  Rows2Plan p{};
  const int ratio = std::max(1, g.W / std::max(1, g.w));
  static const int seg_cols = [] { const char* e = getenv("PM_CE_ROWS2_SEG"); const int v = e ? atoi(e) : 96; return std::max(16, v); }();
  p.nseg = std::max(1, (g.w + seg_cols - 1) / seg_cols);      // column segments per interval (<= 96 columns each: 31 KB of LDS at 19 classes)
  p.ws = (g.w + p.nseg - 1) / p.nseg;
  p.nseg = (g.w + p.ws - 1) / p.ws;
  p.tpr = std::min(320, (p.ws + (p.nseg > 1 ? 1 : 0) + 63) / 64 * 64);      // + the left-neighbour lane when there are segments
  static const int max_threads = [] { const char* e = getenv("PM_CE_ROWS2_THREADS"); const int v = e ? atoi(e) : 256; return std::min(1024, std::max(128, v)); }();      // 256: measured best (gpu_r6_ce.sh)
  p.rg = std::max(1, std::min(ce2_rows_per_interval(g.h, g.H), max_threads / std::max(64, p.tpr)));
  p.lds = ((size_t)2 * (p.ws + 2) * (g.C | 1) + (size_t)2 * p.ws * g.C + (size_t)p.rg * (p.tpr / 64) * MAXC) * sizeof(float) + (size_t)p.rg * ((g.W + 15) & ~15);
  // one lane per low-res column (modest up-sampling ratios; the 16-fold read loss keeps the PARTS form: its field is small), up-sampling only (block count <= hi-res rows:
  // the partial workspace is sized for those), everything in 64 KB of dynamic LDS
  p.ok = !off && g.C == 19 && g.w <= 256 && ratio < 8 && g.H >= g.h && g.h >= 1 && p.lds <= ROWS2_MAX_LDS && (long)g.n * g.h * p.nseg <= (1l << 30) && p.nseg <= 2 * std::max(1, g.H / std::max(1, g.h)) ;      // 19 classes: the
  // specialised instantiation (2 x 19 accumulators + 3 x 19 working registers sit at the 128-register budget of a 1 024-thread block; 32 classes would not)
  return p;
}

inline void rows2_launch(const CEGeom& g, const Rows2Plan& r, float* part, float* field, hipStream_t st) {
  static bool attr_set[64] = {};      // > 64 KB of dynamic LDS needs the opt-in, once per device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ce_fused_rows2_kernel<19>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROWS2_MAX_LDS) != hipSuccess)
      (void)hipGetLastError();        // the launch then fails with its own error for plans beyond 64 KB
    else
      attr_set[dev] = true;
  }
  const long half = (long)g.n * g.h * g.w * g.C;
  hipLaunchKernelGGL(ce_fused_rows2_kernel<19>, dim3(g.n * g.h * r.nseg), dim3(r.tpr * r.rg), r.lds, st, g, part, field, field + half, r.tpr, r.rg, r.nseg, r.ws);
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

extern "C" size_t pm_upsample_ce_workspace(int n, int H, int W) {   // one (sum, count) pair per block of either forward kernel
  return pm_align_up((size_t)std::max<long>(fwd_blocks((long)n * H * W), (long)n * H * 2) * 2 * sizeof(float), 256);      // interval form: <= 2 n H blocks (rows2_plan)
}

extern "C" int pm_upsample_ce_fwd(const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, float* loss_out, void* ws, size_t ws_bytes,
                                  void* stream) {
  PM_REQUIRE_F32(logits, "upsample_ce_fwd");
  CEGeom g;
  if (int e = fill(g, logits, inv_temp, labels, H, W, "upsample_ce_fwd")) return e;
  PM_REQUIRE(loss_out && ws && ws_bytes >= pm_upsample_ce_workspace(g.n, H, W), PM_EWORKSPACE, "upsample_ce_fwd: workspace too small");
  int nb = fwd_blocks((long)g.n * H * W);
  hipStream_t st = (hipStream_t)stream;
  const size_t row_lds = 2 * (size_t)g.w * (g.C | 1) * sizeof(float);     // the two staged low-res rows
  if (row_lds <= 48 * 1024 && (long)g.n * H <= (1 << 20) && W >= 32) {     // row-staged kernel: one block per hi-res row
    nb = g.n * H;
    if (g.C == 19) hipLaunchKernelGGL(ce_fwd_rows_kernel<19>, dim3(nb), dim3(256), row_lds, st, g, (float*)ws);
    else hipLaunchKernelGGL(ce_fwd_rows_kernel<0>, dim3(nb), dim3(256), row_lds, st, g, (float*)ws);
  } else if (g.C == 19) {
    hipLaunchKernelGGL(ce_fwd_kernel<19>, dim3(nb), dim3(256), 0, st, g, (float*)ws);
  } else {
    hipLaunchKernelGGL(ce_fwd_kernel<0>, dim3(nb), dim3(256), 0, st, g, (float*)ws);
  }
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(CE_FINAL_T), 0, st, (const float*)ws, nb, loss_out);
  return pm_check_launch("upsample_ce_fwd");
}

// ---- fused forward + first backward pass (training forward: the logits carry a graph) -------------------------------------------------
namespace {
struct FusedPlan {
  int parts, threads;
  size_t lds;
};
inline FusedPlan fused_plan(const CEGeom& g) {
  FusedPlan p;
  // ~4 hi-res pixels per lane: PARTS lanes share a low-res column when the up-sampling ratio is large (the read loss: 768 / 48 = 16)
  const int ratio = std::max(1, g.W / std::max(1, g.w));
  p.parts = 1;
  while (p.parts < 16 && p.parts * 8 <= ratio && g.w * p.parts * 2 <= 256) p.parts *= 2;
  p.threads = std::min(256, std::max(64, (g.w * p.parts + 63) / 64 * 64));
  p.lds = ((size_t)2 * g.w * (g.C | 1) + g.C) * sizeof(float) + (size_t)(g.W + 15) / 16 * 16;
  return p;
}
constexpr size_t FUSED_MAX_LDS = 159 * 1024;   // dynamic part; the kernel also declares 160 B of static LDS
template <int CC, int PP, bool WITH_T>
void fused_launch_one(const CEGeom& g, const FusedPlan& p, float* part, float* T, hipStream_t st) {
  // > 64 KB of dynamic LDS (logit rows wider than ~420 pixels x 19 classes) needs an explicit opt-in, once per kernel AND device (the guard
  // used to cover the first device of a multi-device process only, and dropped the call's status)
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ce_fused_rows_kernel<CC, PP, WITH_T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_MAX_LDS) != hipSuccess)
      (void)hipGetLastError();      // the launch below then fails with its own error for rows beyond 64 KB
    else
      attr_set[dev] = true;
  }
  hipLaunchKernelGGL((ce_fused_rows_kernel<CC, PP, WITH_T>), dim3(g.n * g.H), dim3(p.threads), p.lds, st, g, part, T);
}
template <bool WITH_T>
int fused_launch(const CEGeom& g, const FusedPlan& p, float* part, float* T, hipStream_t st) {
#define PM_CE_LAUNCH(CC, PP) fused_launch_one<CC, PP, WITH_T>(g, p, part, T, st)
#define PM_CE_PARTS(CC)                         \
  switch (p.parts) {                            \
    case 1: PM_CE_LAUNCH(CC, 1); break;         \
    case 2: PM_CE_LAUNCH(CC, 2); break;         \
    case 4: PM_CE_LAUNCH(CC, 4); break;         \
    case 8: PM_CE_LAUNCH(CC, 8); break;         \
    default: PM_CE_LAUNCH(CC, 16); break;       \
  }
  if (g.C == 19) {
    PM_CE_PARTS(19)
  } else {
    PM_CE_PARTS(0)
  }
#undef PM_CE_PARTS
#undef PM_CE_LAUNCH
  return pm_check_launch("upsample_ce_fused");
}
}  // namespace

extern "C" size_t pm_upsample_ce_field_bytes(const pm_tensor* logits, int H, int W) {
  if (!logits || !pm_is_f32(logits)) return 0;      // fp32 tensors only
  // 0 also when the fused row kernel cannot take the shape -- two low-res logit rows + the label row of one hi-res row must fit LDS (159 KB: w <= ~1000 low-res
  // columns at 19 classes) -- so that a caller routes such shapes elsewhere instead of meeting PM_EUNSUPPORTED (ops.upsample_ce composes resize + CE then)
  CEGeom g{};
  g.n = logits->n, g.h = logits->h, g.w = logits->w, g.H = H, g.W = W, g.C = logits->c;
  if (rows2_plan(g).ok) return (size_t)2 * logits->n * logits->h * logits->w * logits->c * sizeof(float);      // interval form: FA | FB
  if (fused_plan(g).lds > FUSED_MAX_LDS) return 0;
  return (size_t)logits->n * H * logits->w * logits->c * sizeof(float);
}

extern "C" int pm_upsample_ce_fwd_field(const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, float* loss_out, float* field,
                                        void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE_F32(logits, "upsample_ce_fwd_field");
  CEGeom g;
  if (int e = fill(g, logits, inv_temp, labels, H, W, "upsample_ce_fwd_field")) return e;
  PM_REQUIRE(loss_out && field && ws && ws_bytes >= pm_upsample_ce_workspace(g.n, H, W), PM_EWORKSPACE, "upsample_ce_fwd_field: workspace too small / null field");
  hipStream_t st = (hipStream_t)stream;
  if (const Rows2Plan r = rows2_plan(g); r.ok) {
    rows2_launch(g, r, (float*)ws, field, st);
    hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(CE_FINAL_T), 0, st, (const float*)ws, g.n * g.h * r.nseg, loss_out);
    return pm_check_launch("upsample_ce_fwd_field");
  }
  const FusedPlan p = fused_plan(g);
  PM_REQUIRE(p.lds <= FUSED_MAX_LDS && (long)g.n * H <= (1l << 30), PM_EUNSUPPORTED, "upsample_ce_fwd_field: logit rows of %d x %d classes do not fit LDS", g.w, g.C);
  if (int e = fused_launch<true>(g, p, (float*)ws, field, st)) return e;
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(CE_FINAL_T), 0, st, (const float*)ws, g.n * H, loss_out);
  return pm_check_launch("upsample_ce_fwd_field");
}

extern "C" int pm_upsample_ce_bwd_field(const pm_tensor* logits, float inv_temp, int H, int W, const float* loss_out, const float* gscale, const float* field,
                                        const pm_tensor* dlogits, void* stream) {
  PM_REQUIRE_F32(logits, "upsample_ce_bwd_field");
  PM_REQUIRE_F32(dlogits, "upsample_ce_bwd_field");
  CEGeom g;
  static const int64_t dummy = 0;
  if (int e = fill(g, logits, inv_temp, &dummy, H, W, "upsample_ce_bwd_field")) return e;      // the row pass reads neither labels nor logits
  PM_REQUIRE(loss_out && field && dlogits && dlogits->ptr && pm_same_shape(logits, dlogits), PM_EINVAL, "upsample_ce_bwd_field: bad args");
  const long total = (long)g.n * g.h * g.w * g.C;
  if (rows2_plan(g).ok) {
    hipLaunchKernelGGL(ce_bwd_rows2_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, g, field, field + total,
                       loss_out, gscale, (float*)dlogits->ptr, (long)dlogits->pitch);
    return pm_check_launch("upsample_ce_bwd_field");
  }
  hipLaunchKernelGGL(ce_bwd_cols_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, g, field, loss_out,
                     gscale, (float*)dlogits->ptr, (long)dlogits->pitch);
  return pm_check_launch("upsample_ce_bwd_field");
}

// Backward without a field from the forward (the caller ran pm_upsample_ce_fwd): the field is rebuilt into the workspace by the same fused sweep
// (its loss partials are discarded), then the row pass. Same results as pm_upsample_ce_fwd_field + pm_upsample_ce_bwd_field.
extern "C" size_t pm_upsample_ce_bwd_workspace(const pm_tensor* logits, int H, int W) {
  if ((logits && !pm_is_f32(logits))) return 0;      // fp32 tensors only
  return pm_align_up(pm_upsample_ce_field_bytes(logits, H, W), 256) + pm_upsample_ce_workspace(logits->n, H, W);
}

extern "C" int pm_upsample_ce_bwd(const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, const float* loss_out, const float* gscale,
                                  const pm_tensor* dlogits, void* ws, size_t ws_bytes, void* stream) {
  PM_REQUIRE_F32(logits, "upsample_ce_bwd");
  PM_REQUIRE_F32(dlogits, "upsample_ce_bwd");
  CEGeom g;
  if (int e = fill(g, logits, inv_temp, labels, H, W, "upsample_ce_bwd")) return e;
  PM_REQUIRE(loss_out && dlogits && dlogits->ptr && pm_same_shape(logits, dlogits), PM_EINVAL, "upsample_ce_bwd: bad args");
  PM_REQUIRE(ws && ws_bytes >= pm_upsample_ce_bwd_workspace(logits, H, W), PM_EWORKSPACE, "upsample_ce_bwd: workspace too small");
  float* field = (float*)ws;
  float* part = (float*)((char*)ws + pm_align_up(pm_upsample_ce_field_bytes(logits, H, W), 256));
  if (const Rows2Plan r = rows2_plan(g); r.ok) {
    rows2_launch(g, r, part, field, (hipStream_t)stream);
    if (int e = pm_check_launch("upsample_ce_bwd")) return e;
    return pm_upsample_ce_bwd_field(logits, inv_temp, H, W, loss_out, gscale, field, dlogits, stream);
  }
  const FusedPlan p = fused_plan(g);
  PM_REQUIRE(p.lds <= FUSED_MAX_LDS, PM_EUNSUPPORTED, "upsample_ce_bwd: logit rows of %d x %d classes do not fit LDS", g.w, g.C);
  if (int e = fused_launch<true>(g, p, part, field, (hipStream_t)stream)) return e;
  return pm_upsample_ce_bwd_field(logits, inv_temp, H, W, loss_out, gscale, field, dlogits, stream);
}

