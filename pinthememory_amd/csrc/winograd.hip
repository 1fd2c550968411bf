// Winograd F(m x m, 3x3) transforms, m = 2 or 4, for the stride-1 "same" 3x3 convolutions (pad == dilation) of the trunk / ASPP /
// decoder.
//
//   Y = At [ (G g Gt) .* (Bt d B) ] A          (Lavin & Gray; correlation form, fp32 throughout; points 0, +-1 (, +-2), inf)
//
// The element-wise product over channels is P = (m+2)^2 independent GEMMs [tiles x Cin] x [Cin x Cout], which run on the
// implicit-GEMM kernel of conv_igemm.hip as a batched 1x1 convolution (blockIdx.y = transform point): 16 multiplies per 2x2 output
// tile (2.25x fewer MFMA FLOPs than the direct algorithm) or 36 per 4x4 tile (4x fewer).  The transforms here are pure HBM
// streaming kernels:
//   input   x  [N,H,W,C]        -> V [P][tiles][Kp]   (Kp = C rounded up to 32 so that the GEMM's K-state stays wave-uniform)
//   filter  w  [Cout,3,3,Cin]   -> U [P][Cout][Kp]    (or the 180-degree-rotated, channel-transposed filter for the data gradient)
//   output  M  [P][tiles][Cout] -> y [N,H,W,Cout]     (+ the fused epilogue: bias / affine / residual / ReLU)
//   weight gradient: Z = A dY At [P][tiles][Cout];  dU[p] = Z[p]^T V[p] (batched wgrad GEMM, split-K slabs);  dw = Gt (sum dU) G
// A dilated convolution (d > 1) is d*d independent undilated convolutions on the sub-lattices (y % d, x % d); the tile index
// enumerates (image, sub-lattice, tile row, tile column) and the transforms address pixels as  r + d * (m t + a - 1).
#include "pm_common.h"

namespace {

// ---- transform matrices as compile-time functions (every use is fully unrolled, so zero coefficients vanish) ---------------------
enum { W_BT = 0, W_AT = 1, W_G = 2, W_A = 3, W_GT = 4 };   // W_A = At^T, W_GT = G^T

template <int MT>
__host__ __device__ constexpr float wino_bt(int i, int j) {
  if (MT == 2) {
    constexpr float t[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
    return t[i & 3][j & 3];
  }
  constexpr float t[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  return t[i][j];
}
template <int MT>
__host__ __device__ constexpr float wino_at(int i, int j) {
  if (MT == 2) {
    constexpr float t[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
    return t[i & 1][j & 3];
  }
  constexpr float t[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
  return t[i][j];
}
template <int MT>
__host__ __device__ constexpr float wino_g(int i, int j) {
  if (MT == 2) {
    constexpr float t[4][3] = {{1, 0, 0}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0, 0, 1}};
    return t[i & 3][j];
  }
  constexpr float t[6][3] = {{1.f / 4, 0, 0},          {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                             {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6},  {0, 0, 1}};
  return t[i][j];
}
template <int MT, int WHICH>
__host__ __device__ constexpr float wino_coef(int r, int c) {
  return WHICH == W_BT ? wino_bt<MT>(r, c) : WHICH == W_AT ? wino_at<MT>(r, c) : WHICH == W_G ? wino_g<MT>(r, c) : WHICH == W_A ? wino_at<MT>(c, r) : wino_g<MT>(c, r);
}

template <int W>
struct Vec {
  float v[W];
};
template <int W>
__device__ __forceinline__ Vec<W> vzero() {
  Vec<W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = 0.f;
  return r;
}
template <int W>
__device__ __forceinline__ Vec<W> vload(const float* p) {
  Vec<W> r;
  if constexpr (W == 4) {
    const float4 q = PM_LD4(p);
    r.v[0] = q.x, r.v[1] = q.y, r.v[2] = q.z, r.v[3] = q.w;
  } else if constexpr (W == 2) {
    const float2 q = *reinterpret_cast<const float2*>(p);
    r.v[0] = q.x, r.v[1] = q.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}
template <int W>
__device__ __forceinline__ void vstore(float* p, const Vec<W>& r) {
  if constexpr (W == 4) PM_ST4(p, make_float4(r.v[0], r.v[1], r.v[2], r.v[3]));
  else if constexpr (W == 2) *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
  else *p = r.v[0];
}

// out[r] = sum_c coef(r, c) * in[c], r < R, c < C: constants folded, +-1 become adds, zeros disappear
template <int MT, int WHICH, int R, int C, int W>
__device__ __forceinline__ void mat_apply(const Vec<W> (&in)[C], Vec<W> (&out)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    Vec<W> acc = vzero<W>();
    bool first = true;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float k = wino_coef<MT, WHICH>(r, c);
      if (k == 0.f) continue;
#pragma unroll
      for (int i = 0; i < W; ++i) {
        if (first) acc.v[i] = k == 1.f ? in[c].v[i] : (k == -1.f ? -in[c].v[i] : k * in[c].v[i]);
        else if (k == 1.f) acc.v[i] += in[c].v[i];
        else if (k == -1.f) acc.v[i] -= in[c].v[i];
        else acc.v[i] = fmaf(k, in[c].v[i], acc.v[i]);
      }
      first = false;
    }
    out[r] = acc;
  }
}

struct TileId {
  int n, ry, rx, ty, tx;
};
__device__ __forceinline__ TileId decode_tile(unsigned tile, const pm_wino_geom& g) {
  TileId t;
  t.tx = (int)(tile % (unsigned)g.TX);
  unsigned q = tile / (unsigned)g.TX;
  t.ty = (int)(q % (unsigned)g.TY);
  q /= (unsigned)g.TY;
  t.rx = (int)(q % (unsigned)g.d);
  q /= (unsigned)g.d;
  t.ry = (int)(q % (unsigned)g.d);
  t.n = (int)(q / (unsigned)g.d);
  return t;
}

// thread -> (tile, W channels); W = 4 floats for m = 2, 2 floats for m = 4 (36 live values per lane)
template <int MT, int W>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, long xp, int C, int Kp, const pm_wino_geom g, float* __restrict__ V) {
  constexpr int A = MT + 2;
  const unsigned kg = Kp / W;
  const unsigned total = (unsigned)g.tiles * kg;
  const long plane = g.tiles * (long)Kp;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned tile = i / kg;
    const int c = (int)(i - tile * kg) * W;
    const TileId t = decode_tile(tile, g);
    const bool cok = c < C;
    Vec<W> tt[A][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {   // Bt d, one patch column at a time
      const int xx = t.rx + g.d * (MT * t.tx + b - 1);
      const bool xok = cok && (unsigned)xx < (unsigned)g.W;
      Vec<W> col[A], o[A];
#pragma unroll
      for (int a = 0; a < A; ++a) {
        const int yy = t.ry + g.d * (MT * t.ty + a - 1);
        col[a] = (xok && (unsigned)yy < (unsigned)g.H) ? vload<W>(x + ((long)(t.n * g.H + yy) * g.W + xx) * xp + c) : vzero<W>();
      }
      mat_apply<MT, W_BT>(col, o);
#pragma unroll
      for (int a = 0; a < A; ++a) tt[a][b] = o[a];
    }
    float* out = V + (long)tile * Kp + c;
#pragma unroll
    for (int a = 0; a < A; ++a) {   // (Bt d) B
      Vec<W> o[A];
      mat_apply<MT, W_BT>(tt[a], o);
#pragma unroll
      for (int b = 0; b < A; ++b) vstore<W>(out + (a * A + b) * plane, o[b]);
    }
  }
}

template <int MT, int W>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, int Cout, const pm_wino_geom g, float* __restrict__ y, long yp,
                                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ residual, long rp, int relu) {
  constexpr int A = MT + 2;
  const unsigned cg = Cout / W;
  const unsigned total = (unsigned)g.tiles * cg;
  const long plane = g.tiles * (long)Cout;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned tile = i / cg;
    const int c = (int)(i - tile * cg) * W;
    const TileId t = decode_tile(tile, g);
    const float* in = M + (long)tile * Cout + c;
    Vec<W> s[MT][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {   // At m
      Vec<W> col[A], o[MT];
#pragma unroll
      for (int a = 0; a < A; ++a) col[a] = vload<W>(in + (a * A + b) * plane);
      mat_apply<MT, W_AT>(col, o);
#pragma unroll
      for (int a = 0; a < MT; ++a) s[a][b] = o[a];
    }
    Vec<W> bi = vzero<W>(), sc = vzero<W>(), sh = vzero<W>();
    if (bias) bi = vload<W>(bias + c);
    if (scale) sc = vload<W>(scale + c), sh = vload<W>(shift + c);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int yy = t.ry + g.d * (MT * t.ty + a);
      Vec<W> o[MT];
      mat_apply<MT, W_AT>(s[a], o);   // (At m) A
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        const int xx = t.rx + g.d * (MT * t.tx + b);
        if (yy >= g.H || xx >= g.W) continue;
        const long pix = (long)(t.n * g.H + yy) * g.W + xx;
        Vec<W> v = o[b];
        if (bias) {
#pragma unroll
          for (int q = 0; q < W; ++q) v.v[q] += bi.v[q];
        }
        if (scale) {
#pragma unroll
          for (int q = 0; q < W; ++q) v.v[q] = v.v[q] * sc.v[q] + sh.v[q];
        }
        if (residual) {
          const Vec<W> rr = vload<W>(residual + pix * rp + c);
#pragma unroll
          for (int q = 0; q < W; ++q) v.v[q] += rr.v[q];
        }
        if (relu) {
#pragma unroll
          for (int q = 0; q < W; ++q) v.v[q] = fmaxf(v.v[q], 0.f);
        }
        vstore<W>(y + pix * yp + c, v);
      }
    }
  }
}

// U = G g Gt for a 32 (rows of the GEMM's B) x 32 (its K) block of filters, staged through LDS so that both the KRSC read and the
// [P][rows][Kp] write are coalesced for either orientation:
//   forward   rows = Cout, k = Cin,  g(ky,kx) = w[row][ky][kx][k]
//   data grad rows = Cin,  k = Cout, g(ky,kx) = w[k][2-ky][2-kx][row]      (180-degree rotation, channels transposed)
template <int MT, bool DGRAD>
__device__ __forceinline__ void wino_filter_block(const float* __restrict__ w, int Cout, int Cin, int Kp, float* __restrict__ U, int bx, int by,
                                                  float (*sg)[32][33]) {   // sg: [tap][co_local][ci_local]
  constexpr int A = MT + 2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int R = DGRAD ? Cin : Cout;      // GEMM rows
  const int Kr = DGRAD ? Cout : Cin;     // real K extent (zero-filled up to Kp)
  const int r0 = by * 32, k0 = bx * 32;
  const int co0 = DGRAD ? k0 : r0, ci0 = DGRAD ? r0 : k0;
  for (int j = ty; j < 32; j += 8) {   // coalesced over ci
    const int co = co0 + j, ci = ci0 + tx;
    const bool ok = co < Cout && ci < Cin;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) sg[tap][j][tx] = ok ? w[((long)co * 9 + tap) * Cin + ci] : 0.f;
  }
  __syncthreads();
  const long plane = (long)R * Kp;
  // thread -> (row r0 + j, four consecutive k = k0 + 4 q): one float4 store per transform point (36 x 16 B instead of 144 x 4 B per thread)
  const int j = threadIdx.x >> 3, q = threadIdx.x & 7;
  const int row = r0 + j, k = k0 + 4 * q;
  if (row >= R || k >= Kp) return;
  Vec<4> gg[A][3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {   // G g
    Vec<4> col[3], o[A];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int e = 0; e < 4; ++e) col[ky].v[e] = DGRAD ? sg[(2 - ky) * 3 + (2 - kx)][4 * q + e][j] : sg[ky * 3 + kx][j][4 * q + e];
    mat_apply<MT, W_G>(col, o);
#pragma unroll
    for (int a = 0; a < A; ++a) gg[a][kx] = o[a];
  }
  float* out = U + (long)row * Kp + k;
#pragma unroll
  for (int a = 0; a < A; ++a) {      // (G g) Gt
    Vec<4> o[A];
    mat_apply<MT, W_G>(gg[a], o);
#pragma unroll
    for (int b = 0; b < A; ++b) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k + e >= Kr) o[b].v[e] = 0.f;   // zero fill of the K padding
      vstore<4>(out + (a * A + b) * plane, o[b]);
    }
  }
}
template <int MT, bool DGRAD>
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, int Cout, int Cin, int Kp, float* __restrict__ U) {
  __shared__ float sg[9][32][33];
  wino_filter_block<MT, DGRAD>(w, Cout, Cin, Kp, U, blockIdx.x, blockIdx.y, sg);
}

// Every kept forward transform of a model rewritten in ONE launch (pm_conv_wxf_refresh_f32, called behind the optimizer step): the 17-21 per-layer launches of
// wino_filter_kernel in front of the next forward pass are 4 ... 512 blocks each and mostly pure launch latency (17 us on average, 0.36 ms per step).
// Same block body, same values; block -> (job, tile) through a start table carried in the kernel arguments.
constexpr int WR_BATCH = 48;
struct WinoRefreshJobs {
  const float* w[WR_BATCH];
  float* U[WR_BATCH];
  int cout[WR_BATCH], cin[WR_BATCH], kp[WR_BATCH], m[WR_BATCH];
  int start[WR_BATCH + 1];
  int n;
};
__global__ __launch_bounds__(256) void wino_filter_multi_kernel(const WinoRefreshJobs t) {
  __shared__ float sg[9][32][33];
  int j = 0;
  while (j + 1 < t.n && (int)blockIdx.x >= t.start[j + 1]) ++j;      // <= 48 jobs: a scalar walk
  const int local = (int)blockIdx.x - t.start[j];
  const int nbx = t.kp[j] / 32;
  const int bx = local % nbx, by = local / nbx;
  if (t.m[j] == 4) wino_filter_block<4, false>(t.w[j], t.cout[j], t.cin[j], t.kp[j], t.U[j], bx, by, sg);
  else wino_filter_block<2, false>(t.w[j], t.cout[j], t.cin[j], t.kp[j], t.U[j], bx, by, sg);
}

// Weight gradient, step 1: Z = A dY At -- the m x m output-gradient tile scattered to the P transform points (pixels outside the
// image contribute 0).  dU[p][co][ci] = sum_tiles Z[p][tile][co] * V[p][tile][ci] is then a batched wgrad GEMM.
template <int MT, int W>
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, long yp, int Cout, const pm_wino_geom g, float* __restrict__ Z) {
  constexpr int A = MT + 2;
  const unsigned cg = Cout / W;
  const unsigned total = (unsigned)g.tiles * cg;
  const long plane = g.tiles * (long)Cout;
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned tile = i / cg;
    const int c = (int)(i - tile * cg) * W;
    const TileId t = decode_tile(tile, g);
    Vec<W> r[A][MT];
#pragma unroll
    for (int b = 0; b < MT; ++b) {   // A dY
      const int xx = t.rx + g.d * (MT * t.tx + b);
      Vec<W> col[MT], o[A];
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const int yy = t.ry + g.d * (MT * t.ty + a);
        col[a] = (yy < g.H && xx < g.W) ? vload<W>(dy + ((long)(t.n * g.H + yy) * g.W + xx) * yp + c) : vzero<W>();
      }
      mat_apply<MT, W_A>(col, o);
#pragma unroll
      for (int a = 0; a < A; ++a) r[a][b] = o[a];
    }
    float* out = Z + (long)tile * Cout + c;
#pragma unroll
    for (int a = 0; a < A; ++a) {   // (A dY) At
      Vec<W> o[A];
      mat_apply<MT, W_A>(r[a], o);
#pragma unroll
      for (int b = 0; b < A; ++b) vstore<W>(out + (a * A + b) * plane, o[b]);
    }
  }
}

// Weight gradient, step 3: fixed-order sum of the split-K slabs [ks][P][Cout][Kp] and dw = Gt dU G -> KRSC [Cout][3][3][Cin].
template <int MT>
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ slab, int ks, int Cout, int Cin, int Kp, float* __restrict__ dw) {
  constexpr int A = MT + 2, P = A * A;
  const long total = (long)Cout * Cin, plane = (long)Cout * Kp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int co = (int)(i / Cin), ci = (int)(i - (long)co * Cin);
    const float* in = slab + (long)co * Kp + ci;
    // slab sum with z outermost: the P loads of one slab are independent and all in flight together (the per-element order of the
    // additions over z is unchanged, so the result is bit-identical to a z-innermost loop)
    float du[P];
#pragma unroll
    for (int q = 0; q < P; ++q) du[q] = 0.f;
    for (int z = 0; z < ks; ++z) {
      float ld[P];
#pragma unroll
      for (int q = 0; q < P; ++q) ld[q] = in[((long)z * P + q) * plane];
#pragma unroll
      for (int q = 0; q < P; ++q) du[q] += ld[q];
    }
    Vec<1> h[3][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {   // Gt dU
      Vec<1> col[A], o[3];
#pragma unroll
      for (int a = 0; a < A; ++a) col[a].v[0] = du[a * A + b];
      mat_apply<MT, W_GT>(col, o);
#pragma unroll
      for (int k = 0; k < 3; ++k) h[k][b] = o[k];
    }
    float* out = dw + (long)co * 9 * Cin + ci;
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // (Gt dU) G
      Vec<1> o[3];
      mat_apply<MT, W_GT>(h[k], o);
#pragma unroll
      for (int l = 0; l < 3; ++l) out[(k * 3 + l) * (long)Cin] = o[l].v[0];
    }
  }
}

// ---- F(4x4,3x3): the 36 point GEMMs AND the output transform in one kernel -- M = V U^T never reaches HBM ------------------------------
// One block = 32 transform tiles x 32 output channels x ALL 36 points: 12 waves, wave w owns points 3w .. 3w + 2, one 32 x 32 MFMA tile
// (v_mfma_f32_32x32x2_f32, exact fp32 chain) per point = 48 accumulator registers per lane. K (input channels) runs outermost in slabs of 16:
// per slab the block stages 36 x (32 tile rows of V + 32 channel rows of U) x 64 B = 144 KB in LDS (16-byte chunks XOR-swizzled by (row >> 2) & 3:
// conflict-free ds_read_b128 without padding), next slab's rows prefetched to registers under the MFMAs. After the K loop the accumulators are
// parked in the same LDS as M[36][32][32 (+1)] and every (tile, channel) pair applies y = At M A + the fused epilogue (bias / affine / residual /
// ReLU, same expressions as wino_output_kernel), stores leave as 128-byte channel runs. Against GEMM + wino_output_kernel this removes the write
// and the read of M (2.25 x the output each) and one launch; the price is U / V re-read from L2 (Cout / 32 and tiles / 32 times).
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WinoFusedArgs {
  const float* V;      // [36][tiles][Kp]
  const float* U;      // [36][Cout][Kp]
  float* y;
  long yp;
  int Cout, Kp;
  pm_wino_geom g;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual;
  long rp;
  int relu;
  int ncb;             // channel blocks (Cout / 32, rounded up); gridDim.x = tile blocks * ncb
};

constexpr int WF_THREADS = 768, WF_TB = 32, WF_CB = 32, WF_BK = 16, WF_P = 36;
constexpr size_t WF_LDS = (size_t)WF_P * WF_TB * (WF_CB + 1) * sizeof(float);   // the parked accumulators (152 064 B) >= the operand stage (147 456 B)

__device__ __forceinline__ int wf_xcd_remap(int bid, int nwg) {   // consecutive logical ids (they share the V rows) on one XCD's L2
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__global__ __launch_bounds__(WF_THREADS) void wino_fused_f4_kernel(const WinoFusedArgs a) {
  extern __shared__ __align__(16) float4 lds4[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = wf_xcd_remap(blockIdx.x, gridDim.x);
  const int tb = lid / a.ncb, cb = lid - tb * a.ncb;
  const long tile0 = (long)tb * WF_TB;
  const int co0 = cb * WF_CB;

  // ---- loader: thread -> one 16-byte chunk of one row, the same (row, chunk) for the points pgrp, pgrp + 3, ... (12 per thread) ----------------
  const int rem = tid & 255, row = rem >> 2, chunk = rem & 3, pgrp = tid >> 8;
  const bool is_a = row < WF_TB;
  bool valid;
  const float* gp;
  long plane;
  if (is_a) {
    const long tile = tile0 + row;
    valid = tile < a.g.tiles;
    gp = a.V + (valid ? tile : 0) * a.Kp + chunk * 4;
    plane = a.g.tiles * (long)a.Kp;
  } else {
    const int co = co0 + row - WF_TB;
    valid = co < a.Cout;
    gp = a.U + (long)(valid ? co : 0) * a.Kp + chunk * 4;
    plane = (long)a.Cout * a.Kp;
  }
  gp += (long)pgrp * plane;
  const long pstep = 3 * plane;
  const int ldsw = pgrp * 256 + row * 4 + (chunk ^ ((row >> 2) & 3));
  float4 r[12];
  auto gload = [&](int k0) {
#pragma unroll
    for (int it = 0; it < 12; ++it) r[it] = valid ? PM_LD4(gp + it * pstep + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int it = 0; it < 12; ++it) lds4[it * 768 + ldsw] = r[it];
  };

  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
  const int l31 = lane & 31, half = lane >> 5;
  // fragment rows: A row l31, B row 32 + l31; this lane-half's 8 k are chunks 2 half, 2 half + 1 of the row (swizzled)
  const int sa = (l31 >> 2) & 3, sb = ((32 + l31) >> 2) & 3;
  const int ia0 = l31 * 4 + ((2 * half) ^ sa), ia1 = l31 * 4 + ((2 * half + 1) ^ sa);
  const int ib0 = (32 + l31) * 4 + ((2 * half) ^ sb), ib1 = (32 + l31) * 4 + ((2 * half + 1) ^ sb);
  auto compute = [&]() {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4* T = lds4 + (wave * 3 + j) * 256;
      const float4 a0 = T[ia0], a1 = T[ia1], b0 = T[ib0], b1 = T[ib1];
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[j], 0, 0, 0);
    }
  };

  const int nk = a.Kp / WF_BK;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * WF_BK);      // in flight under the 24 MFMAs of this slab
    compute();
    __syncthreads();                                // every wave is done reading the stage
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }

  // ---- epilogue: park the 36 products of every (tile, channel) pair in LDS, then y = At M A per pair -----------------------------------------
  float* Ms = reinterpret_cast<float*>(lds4);       // [36][32 tiles][33]
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int pnt = wave * 3 + j;
#pragma unroll
    for (int q = 0; q < 16; ++q) Ms[(pnt * WF_TB + (q & 3) + 8 * (q >> 2) + 4 * half) * (WF_CB + 1) + l31] = acc[j][q];
  }
  __syncthreads();
  constexpr int MT = 4, A = 6;
  for (int idx = tid; idx < WF_TB * WF_CB; idx += WF_THREADS) {
    const int t = idx >> 5, c = idx & 31;
    const long tile = tile0 + t;
    const int co = co0 + c;
    if (tile >= a.g.tiles || co >= a.Cout) continue;
    const TileId ti = decode_tile((unsigned)tile, a.g);
    Vec<1> sv[MT][A];
#pragma unroll
    for (int b = 0; b < A; ++b) {   // At m
      Vec<1> col[A], o[MT];
#pragma unroll
      for (int aa = 0; aa < A; ++aa) col[aa].v[0] = Ms[((aa * A + b) * WF_TB + t) * (WF_CB + 1) + c];
      mat_apply<MT, W_AT>(col, o);
#pragma unroll
      for (int aa = 0; aa < MT; ++aa) sv[aa][b] = o[aa];
    }
    const float bi = a.bias ? a.bias[co] : 0.f, sc = a.scale ? a.scale[co] : 0.f, sh = a.scale ? a.shift[co] : 0.f;
#pragma unroll
    for (int aa = 0; aa < MT; ++aa) {
      const int yy = ti.ry + a.g.d * (MT * ti.ty + aa);
      Vec<1> o[MT];
      mat_apply<MT, W_AT>(sv[aa], o);   // (At m) A
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        const int xx = ti.rx + a.g.d * (MT * ti.tx + b);
        if (yy >= a.g.H || xx >= a.g.W) continue;
        const long pix = (long)(ti.n * a.g.H + yy) * a.g.W + xx;
        float v = o[b].v[0];
        if (a.bias) v += bi;
        if (a.scale) v = v * sc + sh;
        if (a.residual) v += a.residual[pix * a.rp + co];
        if (a.relu) v = fmaxf(v, 0.f);
        a.y[pix * a.yp + co] = v;
      }
    }
  }
}

inline unsigned nblocks(long total) { return (unsigned)std::min<long>((total + 255) / 256, 1 << 20); }

}  // namespace

pm_wino_geom pm_wino_make_geom(int n, int h, int w, int d, int m) {
  pm_wino_geom g;
  g.N = n, g.H = h, g.W = w, g.d = d, g.m = m;
  g.TY = (pm_cdiv(h, d) + m - 1) / m, g.TX = (pm_cdiv(w, d) + m - 1) / m;
  g.tiles = (long)n * d * d * g.TY * g.TX;
  return g;
}

int pm_wino_input_xf(const float* x, long pitch, int C, int Kp, const pm_wino_geom& g, float* V, hipStream_t st) {
  if (g.m == 4) hipLaunchKernelGGL((wino_input_kernel<4, 2>), dim3(nblocks(g.tiles * (Kp / 2))), dim3(256), 0, st, x, pitch, C, Kp, g, V);
  else hipLaunchKernelGGL((wino_input_kernel<2, 4>), dim3(nblocks(g.tiles * (Kp / 4))), dim3(256), 0, st, x, pitch, C, Kp, g, V);
  return pm_check_launch("wino_input");
}

int pm_wino_filter_xf(const float* w, int Cout, int Cin, int Kp, bool dgrad, int m, float* U, hipStream_t st) {
  const int R = dgrad ? Cin : Cout;
  dim3 grid(Kp / 32, pm_cdiv(R, 32));
  if (m == 4) {
    if (dgrad) hipLaunchKernelGGL((wino_filter_kernel<4, true>), grid, dim3(256), 0, st, w, Cout, Cin, Kp, U);
    else hipLaunchKernelGGL((wino_filter_kernel<4, false>), grid, dim3(256), 0, st, w, Cout, Cin, Kp, U);
  } else {
    if (dgrad) hipLaunchKernelGGL((wino_filter_kernel<2, true>), grid, dim3(256), 0, st, w, Cout, Cin, Kp, U);
    else hipLaunchKernelGGL((wino_filter_kernel<2, false>), grid, dim3(256), 0, st, w, Cout, Cin, Kp, U);
  }
  return pm_check_launch("wino_filter");
}

// jobs: forward transforms only (w [Cout][3][3][Cin], U [P][Cout][Kp], m in {2, 4}); batches of WR_BATCH per launch
int pm_wino_filter_xf_multi(const float* const* w, float* const* U, const int* cout, const int* cin, const int* kp, const int* m, int n, hipStream_t st) {
  for (int i0 = 0; i0 < n; i0 += WR_BATCH) {
    WinoRefreshJobs t;
    t.n = std::min(WR_BATCH, n - i0);
    long blocks = 0;
    for (int i = 0; i < t.n; ++i) {
      t.w[i] = w[i0 + i], t.U[i] = U[i0 + i], t.cout[i] = cout[i0 + i], t.cin[i] = cin[i0 + i], t.kp[i] = kp[i0 + i], t.m[i] = m[i0 + i];
      t.start[i] = (int)blocks;
      blocks += (long)(kp[i0 + i] / 32) * pm_cdiv(cout[i0 + i], 32);
    }
    t.start[t.n] = (int)blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(wino_filter_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, t);
    if (int e = pm_check_launch("wino_filter_multi")) return e;
  }
  return PM_OK;
}

int pm_wino_output_xf(const float* M, int Cout, const pm_wino_geom& g, float* y, long ypitch, const float* bias, const float* scale, const float* shift,
                      const float* residual, long res_pitch, int relu, hipStream_t st) {
  if (g.m == 4)
    hipLaunchKernelGGL((wino_output_kernel<4, 2>), dim3(nblocks(g.tiles * (Cout / 2))), dim3(256), 0, st, M, Cout, g, y, ypitch, bias, scale, shift, residual,
                       res_pitch, relu);
  else
    hipLaunchKernelGGL((wino_output_kernel<2, 4>), dim3(nblocks(g.tiles * (Cout / 4))), dim3(256), 0, st, M, Cout, g, y, ypitch, bias, scale, shift, residual,
                       res_pitch, relu);
  return pm_check_launch("wino_output");
}

int pm_wino_dy_xf(const float* dy, long pitch, int Cout, const pm_wino_geom& g, float* Z, hipStream_t st) {
  if (g.m == 4) hipLaunchKernelGGL((wino_dy_kernel<4, 2>), dim3(nblocks(g.tiles * (Cout / 2))), dim3(256), 0, st, dy, pitch, Cout, g, Z);
  else hipLaunchKernelGGL((wino_dy_kernel<2, 4>), dim3(nblocks(g.tiles * (Cout / 4))), dim3(256), 0, st, dy, pitch, Cout, g, Z);
  return pm_check_launch("wino_dy");
}

int pm_wino_dw_xf(const float* slab, int ks, int Cout, int Cin, int Kp, int m, float* dw, hipStream_t st) {
  const unsigned nb = (unsigned)std::min<long>(((long)Cout * Cin + 255) / 256, 1 << 16);
  if (m == 4) hipLaunchKernelGGL(wino_dw_kernel<4>, dim3(nb), dim3(256), 0, st, slab, ks, Cout, Cin, Kp, dw);
  else hipLaunchKernelGGL(wino_dw_kernel<2>, dim3(nb), dim3(256), 0, st, slab, ks, Cout, Cin, Kp, dw);
  return pm_check_launch("wino_dw");
}

// GEMMs + output transform in one kernel (F(4x4) only): V [36][tiles][Kp] x U [36][Cout][Kp] -> y with the fused epilogue.
int pm_wino_fused_f4(const float* V, const float* U, int Cout, int Kp, const pm_wino_geom& g, float* y, long ypitch, const float* bias, const float* scale,
                     const float* shift, const float* residual, long res_pitch, int relu, hipStream_t st) {
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_fused_f4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WF_LDS);
    return true;
  }();
  (void)attr_set;
  WinoFusedArgs a;
  a.V = V, a.U = U, a.y = y, a.yp = ypitch, a.Cout = Cout, a.Kp = Kp, a.g = g;
  a.bias = bias, a.scale = scale, a.shift = shift, a.residual = residual, a.rp = res_pitch, a.relu = relu;
  a.ncb = (Cout + WF_CB - 1) / WF_CB;
  const long ntb = (g.tiles + WF_TB - 1) / WF_TB;
  hipLaunchKernelGGL(wino_fused_f4_kernel, dim3((unsigned)(ntb * a.ncb)), dim3(WF_THREADS), WF_LDS, st, a);
  return pm_check_launch("wino_fused_f4");
}

