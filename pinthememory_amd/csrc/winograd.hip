// Winograd F(2x2, 3x3) transforms for the stride-1 "same" 3x3 convolutions (pad == dilation) of the trunk / ASPP / decoder.
//
//   Y = At [ (G g Gt) .* (Bt d B) ] A          (Lavin & Gray; correlation form, fp32 throughout)
//
// The element-wise product over channels is 16 independent GEMMs [tiles x Cin] x [Cin x Cout], which run on the implicit-GEMM kernel
// of conv_igemm.hip as a batched 1x1 convolution (blockIdx.y = transform point): 16 multiplies per 2x2 output tile instead of 36,
// i.e. 2.25x fewer MFMA FLOPs than the direct algorithm.  The three transforms here are pure HBM streaming kernels:
//   input   x  [N,H,W,C]       -> V [16][tiles][Kp]   (Kp = C rounded up to 32 so that the GEMM's K-state stays wave-uniform)
//   filter  w  [Cout,3,3,Cin]  -> U [16][Cout][Kp]    (or the 180-degree-rotated, channel-transposed filter for the data gradient)
//   output  M  [16][tiles][Cout] -> y [N,H,W,Cout]    (+ the fused epilogue: bias / affine / residual / ReLU)
// A dilated convolution (d > 1) is d*d independent undilated convolutions on the sub-lattices (y % d, x % d); the tile index
// enumerates (image, sub-lattice, tile row, tile column) and the transforms address pixels as  r + d * (2 t + a - 1).
#include "pm_common.h"

namespace {

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

struct TileId {
  int n, ry, rx, ty, tx;
};
__device__ __forceinline__ TileId decode_tile(long tile, const pm_wino_geom& g) {
  TileId t;
  t.tx = (int)(tile % g.TX);
  long q = tile / g.TX;
  t.ty = (int)(q % g.TY);
  q /= g.TY;
  t.rx = (int)(q % g.d);
  q /= g.d;
  t.ry = (int)(q % g.d);
  t.n = (int)(q / g.d);
  return t;
}

__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, long xp, int C, int Kp, const pm_wino_geom g,
                                                         float* __restrict__ V) {
  const int kg = Kp >> 2;
  const long total = g.tiles * kg, plane = g.tiles * (long)Kp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long tile = i / kg;
    const int c = (int)(i - tile * kg) * 4;
    const TileId t = decode_tile(tile, g);
    const bool cok = c < C;
    float4 dm[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = t.ry + g.d * (2 * t.ty + a - 1);
      const bool yok = cok && (unsigned)yy < (unsigned)g.H;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = t.rx + g.d * (2 * t.tx + b - 1);
        dm[a][b] = (yok && (unsigned)xx < (unsigned)g.W) ? PM_LD4(x + ((long)(t.n * g.H + yy) * g.W + xx) * xp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float4 tt[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {   // Bt d
      tt[0][b] = f4sub(dm[0][b], dm[2][b]);
      tt[1][b] = f4add(dm[1][b], dm[2][b]);
      tt[2][b] = f4sub(dm[2][b], dm[1][b]);
      tt[3][b] = f4sub(dm[1][b], dm[3][b]);
    }
    float* out = V + tile * Kp + c;
#pragma unroll
    for (int a = 0; a < 4; ++a) {   // (Bt d) B
      PM_ST4(out + (a * 4 + 0) * plane, f4sub(tt[a][0], tt[a][2]));
      PM_ST4(out + (a * 4 + 1) * plane, f4add(tt[a][1], tt[a][2]));
      PM_ST4(out + (a * 4 + 2) * plane, f4sub(tt[a][2], tt[a][1]));
      PM_ST4(out + (a * 4 + 3) * plane, f4sub(tt[a][1], tt[a][3]));
    }
  }
}

__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, int Cout, const pm_wino_geom g, float* __restrict__ y, long yp,
                                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ residual, long rp, int relu) {
  const int cg = Cout >> 2;
  const long total = g.tiles * cg, plane = g.tiles * (long)Cout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long tile = i / cg;
    const int c = (int)(i - tile * cg) * 4;
    const TileId t = decode_tile(tile, g);
    const float* in = M + tile * Cout + c;
    float4 s[2][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {   // At m
      const float4 m0 = PM_LD4(in + (0 + b) * plane), m1 = PM_LD4(in + (4 + b) * plane), m2 = PM_LD4(in + (8 + b) * plane),
                   m3 = PM_LD4(in + (12 + b) * plane);
      s[0][b] = f4add(f4add(m0, m1), m2);
      s[1][b] = f4sub(f4sub(m1, m2), m3);
    }
    float4 bi = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bi;
    if (bias) bi = PM_LD4(bias + c);
    if (scale) sc = PM_LD4(scale + c), sh = PM_LD4(shift + c);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int yy = t.ry + g.d * (2 * t.ty + a);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int xx = t.rx + g.d * (2 * t.tx + b);
        if (yy >= g.H || xx >= g.W) continue;
        float4 v = b == 0 ? f4add(f4add(s[a][0], s[a][1]), s[a][2]) : f4sub(f4sub(s[a][1], s[a][2]), s[a][3]);   // (At m) A
        const long pix = (long)(t.n * g.H + yy) * g.W + xx;
        if (bias || scale) {
          v.x = (v.x + bi.x) * sc.x + sh.x, v.y = (v.y + bi.y) * sc.y + sh.y;
          v.z = (v.z + bi.z) * sc.z + sh.z, v.w = (v.w + bi.w) * sc.w + sh.w;
        }
        if (residual) v = f4add(v, PM_LD4(residual + pix * rp + c));
        if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
        PM_ST4(y + pix * yp + c, v);
      }
    }
  }
}

// U = G g Gt for a 32 (rows of the GEMM's B) x 32 (its K) block of filters, staged through LDS so that both the KRSC read and the
// [16][rows][Kp] write are coalesced for either orientation:
//   forward   rows = Cout, k = Cin,  g(ky,kx) = w[row][ky][kx][k]
//   data grad rows = Cin,  k = Cout, g(ky,kx) = w[k][2-ky][2-kx][row]      (180-degree rotation, channels transposed)
template <bool DGRAD>
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, int Cout, int Cin, int Kp, float* __restrict__ U) {
  __shared__ float sg[9][32][33];   // [tap][co_local][ci_local]
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int R = DGRAD ? Cin : Cout;      // GEMM rows
  const int Kr = DGRAD ? Cout : Cin;     // real K extent (zero-filled up to Kp)
  const int r0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int co0 = DGRAD ? k0 : r0, ci0 = DGRAD ? r0 : k0;
  for (int j = ty; j < 32; j += 8) {   // coalesced over ci
    const int co = co0 + j, ci = ci0 + tx;
    const bool ok = co < Cout && ci < Cin;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) sg[tap][j][tx] = ok ? w[((long)co * 9 + tap) * Cin + ci] : 0.f;
  }
  __syncthreads();
  const long plane = (long)R * Kp;
  for (int j = ty; j < 32; j += 8) {   // thread -> (row r0 + j, k = k0 + tx): coalesced over k
    const int row = r0 + j, k = k0 + tx;
    if (row >= R || k >= Kp) continue;
    float gq[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) gq[ky][kx] = DGRAD ? sg[(2 - ky) * 3 + (2 - kx)][tx][j] : sg[ky * 3 + kx][j][tx];
    float gg[4][3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {   // G g
      gg[0][kx] = gq[0][kx];
      gg[1][kx] = 0.5f * (gq[0][kx] + gq[1][kx] + gq[2][kx]);
      gg[2][kx] = 0.5f * (gq[0][kx] - gq[1][kx] + gq[2][kx]);
      gg[3][kx] = gq[2][kx];
    }
    float* out = U + (long)row * Kp + k;
    const bool kok = k < Kr;
#pragma unroll
    for (int a = 0; a < 4; ++a) {      // (G g) Gt
      out[(a * 4 + 0) * plane] = kok ? gg[a][0] : 0.f;
      out[(a * 4 + 1) * plane] = kok ? 0.5f * (gg[a][0] + gg[a][1] + gg[a][2]) : 0.f;
      out[(a * 4 + 2) * plane] = kok ? 0.5f * (gg[a][0] - gg[a][1] + gg[a][2]) : 0.f;
      out[(a * 4 + 3) * plane] = kok ? gg[a][2] : 0.f;
    }
  }
}

// Weight gradient, step 1: Z = A dY At -- the 2x2 output-gradient tile scattered to the 16 transform points (pixels outside the
// image contribute 0).  dU[p][co][ci] = sum_tiles Z[p][tile][co] * V[p][tile][ci] is then a batched wgrad GEMM.
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, long yp, int Cout, const pm_wino_geom g, float* __restrict__ Z) {
  const int cg = Cout >> 2;
  const long total = g.tiles * cg, plane = g.tiles * (long)Cout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long tile = i / cg;
    const int c = (int)(i - tile * cg) * 4;
    const TileId t = decode_tile(tile, g);
    float4 q[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int yy = t.ry + g.d * (2 * t.ty + a);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int xx = t.rx + g.d * (2 * t.tx + b);
        q[a][b] = (yy < g.H && xx < g.W) ? PM_LD4(dy + ((long)(t.n * g.H + yy) * g.W + xx) * yp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r[4][2];   // A dY: rows (y0, y0 + y1, y0 - y1, -y1)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      r[0][b] = q[0][b];
      r[1][b] = f4add(q[0][b], q[1][b]);
      r[2][b] = f4sub(q[0][b], q[1][b]);
      r[3][b] = f4sub(zero, q[1][b]);
    }
    float* out = Z + tile * Cout + c;
#pragma unroll
    for (int a = 0; a < 4; ++a) {   // (A dY) At
      PM_ST4(out + (a * 4 + 0) * plane, r[a][0]);
      PM_ST4(out + (a * 4 + 1) * plane, f4add(r[a][0], r[a][1]));
      PM_ST4(out + (a * 4 + 2) * plane, f4sub(r[a][0], r[a][1]));
      PM_ST4(out + (a * 4 + 3) * plane, f4sub(zero, r[a][1]));
    }
  }
}

// Weight gradient, step 3: fixed-order sum of the split-K slabs [ks][16][Cout][Kp] and dw = Gt dU G -> KRSC [Cout][3][3][Cin].
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ slab, int ks, int Cout, int Cin, int Kp, float* __restrict__ dw) {
  const long total = (long)Cout * Cin, plane = (long)Cout * Kp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int co = (int)(i / Cin), ci = (int)(i - (long)co * Cin);
    const float* in = slab + (long)co * Kp + ci;
    float u[4][4];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      float acc = 0.f;
      for (int z = 0; z < ks; ++z) acc += in[((long)z * 16 + p) * plane];
      u[p >> 2][p & 3] = acc;
    }
    float h[3][4];   // Gt dU
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      h[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
      h[1][b] = 0.5f * (u[1][b] - u[2][b]);
      h[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
    }
    float* out = dw + (long)co * 9 * Cin + ci;
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // (Gt dU) G
      out[(a * 3 + 0) * (long)Cin] = h[a][0] + 0.5f * (h[a][1] + h[a][2]);
      out[(a * 3 + 1) * (long)Cin] = 0.5f * (h[a][1] - h[a][2]);
      out[(a * 3 + 2) * (long)Cin] = 0.5f * (h[a][1] + h[a][2]) + h[a][3];
    }
  }
}

}  // namespace

int pm_wino_dy_xf(const float* dy, long pitch, int Cout, const pm_wino_geom& g, float* Z, hipStream_t st) {
  const long total = g.tiles * (Cout / 4);
  hipLaunchKernelGGL(wino_dy_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)), dim3(256), 0, st, dy, pitch, Cout, g, Z);
  return pm_check_launch("wino_dy");
}

int pm_wino_dw_xf(const float* slab, int ks, int Cout, int Cin, int Kp, float* dw, hipStream_t st) {
  const long total = (long)Cout * Cin;
  hipLaunchKernelGGL(wino_dw_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 16)), dim3(256), 0, st, slab, ks, Cout, Cin, Kp, dw);
  return pm_check_launch("wino_dw");
}

pm_wino_geom pm_wino_make_geom(int n, int h, int w, int d) {
  pm_wino_geom g;
  g.N = n, g.H = h, g.W = w, g.d = d;
  g.TY = (pm_cdiv(h, d) + 1) / 2, g.TX = (pm_cdiv(w, d) + 1) / 2;
  g.tiles = (long)n * d * d * g.TY * g.TX;
  return g;
}

int pm_wino_input_xf(const float* x, long pitch, int C, int Kp, const pm_wino_geom& g, float* V, hipStream_t st) {
  const long total = g.tiles * (Kp / 4);
  hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)), dim3(256), 0, st, x, pitch, C, Kp, g, V);
  return pm_check_launch("wino_input");
}

int pm_wino_filter_xf(const float* w, int Cout, int Cin, int Kp, bool dgrad, float* U, hipStream_t st) {
  const int R = dgrad ? Cin : Cout;
  dim3 grid(Kp / 32, pm_cdiv(R, 32));
  if (dgrad) hipLaunchKernelGGL(wino_filter_kernel<true>, grid, dim3(256), 0, st, w, Cout, Cin, Kp, U);
  else hipLaunchKernelGGL(wino_filter_kernel<false>, grid, dim3(256), 0, st, w, Cout, Cin, Kp, U);
  return pm_check_launch("wino_filter");
}

int pm_wino_output_xf(const float* M, int Cout, const pm_wino_geom& g, float* y, long ypitch, const float* bias, const float* scale, const float* shift,
                      const float* residual, long res_pitch, int relu, hipStream_t st) {
  const long total = g.tiles * (Cout / 4);
  hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 1 << 20)), dim3(256), 0, st, M, Cout, g, y, ypitch, bias, scale,
                     shift, residual, res_pitch, relu);
  return pm_check_launch("wino_output");
}
