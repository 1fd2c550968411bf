// K1/K2/K3: im2col-free implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TF peak).
//
// One kernel template, three gather modes over  C[M][N] = sum_k A[m][k] * B[k][n] :
//   FWD    M = n*ho*wo output pixels, N = Cout,      K = taps*Cin   A = gathered x rows (k-contiguous), B = w[Cout][K]
//   DGRAD  M = n*h*w   input  pixels, N = Cin,       K = taps*Cout  A = gathered dy rows,               B = w viewed [k][Cin]
//   WGRAD  M = Cout,                  N = taps*Cin,  K = pixels     A = dy[p][Cout] (m-contiguous),     B = gathered x rows
// NHWC keeps every gathered row a contiguous channel vector, so global loads are 16 B/lane and coalesced; dilation
// (6/12/18/24) costs nothing because taps are gathered from L2, never staged as a spatial halo.
// 256 threads = 4 waves; each wave owns TMxTN MFMA tiles of 32x32; LDS double-buffered, next K-slab prefetched to
// registers while the current one feeds the matrix pipe (one barrier per K-step).
// Replaces nn.Conv2d fwd/bwd of /root/reference/network/Resnet.py:145-150,404,453-457, deepv3plus.py:72-81,398-424.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>

#include "pm_common.h"

#include "conv_igemm_kernel.h"

namespace {

// split-K combine: C[row][col] = epilogue( sum_z ws[z][row][col] ), fixed z order (deterministic).
// Vectorised split-K reduce: thread -> 4 consecutive outputs, ZL lanes share the slabs of one output group (lane l sums
// z = l, l + ZL, ...; the ZL partial sums are combined in lane order through LDS), so that a 256-way split over a small dw is not
// a 256-long serial chain of dependent loads. Fixed association order -> deterministic.
template <int ZL, bool O16 = false>
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ ws, int ksplit, long slab, int M, int Nn,
                                                                float* __restrict__ C, long c_pitch, const float* bias,
                                                                const float* scale, const float* shift, const float* residual,
                                                                long res_pitch, int relu) {
  constexpr int GP = 256 / ZL;   // output groups per block
  __shared__ float4 red[ZL > 1 ? ZL : 1][GP];
  const int gl = threadIdx.x % GP, zl = threadIdx.x / GP;
  const long groups = (long)M * Nn / 4;
  for (long g0 = (long)blockIdx.x * GP; g0 < groups; g0 += (long)gridDim.x * GP) {
    const long grp = g0 + gl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (grp < groups)
#pragma unroll 8   // eight independent slab loads in flight; the additions stay in z order
      for (int z = zl; z < ksplit; z += ZL) {
        const float4 q = PM_LD4(ws + (long)z * slab + grp * 4);
        v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
      }
    if constexpr (ZL > 1) {
      red[zl][gl] = v;
      __syncthreads();
      if (zl == 0) {
#pragma unroll
        for (int l = 1; l < ZL; ++l) {
          const float4 q = red[l][gl];
          v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
        }
      }
    }
    if (zl == 0 && grp < groups) {
      const long i = grp * 4;
      const int row = (int)(i / Nn), col = (int)(i - (long)row * Nn);
      if (bias) {
        const float4 b = PM_LD4(bias + col);
        v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
      }
      if (scale) {
        const float4 sc = PM_LD4(scale + col), sh = PM_LD4(shift + col);
        v.x = v.x * sc.x + sh.x, v.y = v.y * sc.y + sh.y, v.z = v.z * sc.z + sh.z, v.w = v.w * sc.w + sh.w;
      }
      if (residual) {
        if constexpr (O16) {      // bf16 tier: residual and output are bf16 tensors (pitches in elements), 8-byte accesses
          const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const pm_bf16*>(residual) + (long)row * res_pitch + col);
          v.x += __uint_as_float(q.x << 16), v.y += __uint_as_float(q.x & 0xffff0000u), v.z += __uint_as_float(q.y << 16), v.w += __uint_as_float(q.y & 0xffff0000u);
        } else {
          const float4 r = PM_LD4(residual + (long)row * res_pitch + col);
          v.x += r.x, v.y += r.y, v.z += r.z, v.w += r.w;
        }
      }
      if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
      if constexpr (O16) *reinterpret_cast<uint2*>(reinterpret_cast<pm_bf16*>(C) + (long)row * c_pitch + col) = make_uint2(pm_pack_bf16(v.x, v.y), pm_pack_bf16(v.z, v.w));
      else PM_ST4(C + (long)row * c_pitch + col, v);
    }
    if constexpr (ZL > 1) __syncthreads();
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, long slab, int M, int Nn,
                                                            float* __restrict__ C, long c_pitch, const float* bias,
                                                            const float* scale, const float* shift, const float* residual,
                                                            long res_pitch, int relu) {
  const long total = (long)M * Nn;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int row = (int)(i / Nn), col = (int)(i - (long)row * Nn);
    float v = 0.f;
    for (int z = 0; z < ksplit; ++z) v += ws[(long)z * slab + i];
    if (bias) v += bias[col];
    if (scale) v = v * scale[col] + shift[col];
    if (residual) v += residual[(long)row * res_pitch + col];
    if (relu) v = fmaxf(v, 0.f);
    C[(long)row * c_pitch + col] = v;
  }
}

int splitk_reduce(const float* ws, int ksplit, long M, long Nn, float* C, long c_pitch, const float* bias, const float* scale, const float* shift,
                  const float* residual, long res_pitch, int relu, hipStream_t st, bool out16 = false) {
  const long total = M * Nn;
  if (out16) {      // bf16 tier: bf16 output / residual, four outputs = 8 bytes per thread
    PM_REQUIRE((Nn % 4 == 0) && (c_pitch % 4 == 0) && (res_pitch % 4 == 0) && pm_aligned16(ws) && (reinterpret_cast<uintptr_t>(C) & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(residual) & 7) == 0 && (!bias || pm_aligned16(bias)) && (!scale || (pm_aligned16(scale) && pm_aligned16(shift))),
               PM_EUNSUPPORTED, "splitk_reduce(bf16): unaligned output");
    const long groups = total / 4;
    const int zl = (ksplit >= 32 && groups < (1 << 18)) ? 16 : (ksplit >= 8 && groups < (1 << 20) ? 4 : 1);
    const int nb = (int)std::min<long>((groups + 256 / zl - 1) / (256 / zl), 8192);
    if (zl == 16)
      hipLaunchKernelGGL((splitk_reduce_vec_kernel<16, true>), dim3(nb), dim3(256), 0, st, ws, ksplit, total, (int)M, (int)Nn, C, c_pitch, bias, scale, shift, residual, res_pitch, relu);
    else if (zl == 4)
      hipLaunchKernelGGL((splitk_reduce_vec_kernel<4, true>), dim3(nb), dim3(256), 0, st, ws, ksplit, total, (int)M, (int)Nn, C, c_pitch, bias, scale, shift, residual, res_pitch, relu);
    else
      hipLaunchKernelGGL((splitk_reduce_vec_kernel<1, true>), dim3(nb), dim3(256), 0, st, ws, ksplit, total, (int)M, (int)Nn, C, c_pitch, bias, scale, shift, residual, res_pitch, relu);
    return pm_check_launch("splitk_reduce(bf16)");
  }
  const bool vec = (Nn % 4 == 0) && (c_pitch % 4 == 0) && (res_pitch % 4 == 0) && pm_aligned16(ws) && pm_aligned16(C) &&
                   (!bias || pm_aligned16(bias)) && (!scale || (pm_aligned16(scale) && pm_aligned16(shift))) && (!residual || pm_aligned16(residual));
  if (!vec) {
    const int nb = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, ws, ksplit, total, (int)M, (int)Nn, C, c_pitch, bias, scale, shift, residual,
                       res_pitch, relu);
  } else {
    const long groups = total / 4;
    // enough lanes per output group that the whole GPU has work: small outputs with many slabs take 16 lanes
    const int zl = (ksplit >= 32 && groups < (1 << 18)) ? 16 : (ksplit >= 8 && groups < (1 << 20) ? 4 : 1);
    const int gp = 256 / zl;
    const int nb = (int)std::min<long>((groups + gp - 1) / gp, 8192);
    if (zl == 16)
      hipLaunchKernelGGL(splitk_reduce_vec_kernel<16>, dim3(nb), dim3(256), 0, st, ws, ksplit, total, (int)M, (int)Nn, C, c_pitch, bias, scale, shift, residual, res_pitch, relu);
    else if (zl == 4)
      hipLaunchKernelGGL(splitk_reduce_vec_kernel<4>, dim3(nb), dim3(256), 0, st, ws, ksplit, total, (int)M, (int)Nn, C, c_pitch, bias, scale, shift, residual, res_pitch, relu);
    else
      hipLaunchKernelGGL(splitk_reduce_vec_kernel<1>, dim3(nb), dim3(256), 0, st, ws, ksplit, total, (int)M, (int)Nn, C, c_pitch, bias, scale, shift, residual, res_pitch, relu);
  }
  return pm_check_launch("splitk_reduce");
}

// pixel rows per block of the bias-gradient column sum: ~1024 blocks in flight, 64 ... 2048 rows each
inline int colsum_rows(long P, int C) {
  const long col_blocks = std::max<long>(1, (C + 63) / 64);
  const long want = col_blocks == 1 ? 256 : std::max<long>(1, 1024 / col_blocks);   // the final pass walks the partials serially per channel
  return (int)std::min<long>(2048, std::max<long>(64, (P + want - 1) / want));
}

// column sum of dy for the conv bias gradient: one block per 64 channels x pixel chunk, fixed-order second stage.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long pitch, long P, int C, int rows_per_block,
                                                             float* __restrict__ part) {
  __shared__ float sm[4][64];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const long p0 = (long)blockIdx.x * rows_per_block, p1 = min(P, p0 + rows_per_block);
  float s = 0.f;
  if (c < C)
    for (long p = p0 + r; p < p1; p += 4) s += x[p * pitch + c];
  sm[r][threadIdx.x & 63] = s;
  __syncthreads();
  if (r == 0 && c < C) part[(long)blockIdx.x * C + c] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}
// narrow tensors (pitch <= 64 floats, e.g. the 19 class logits at pitch 20): the rows are one contiguous stream of float4 groups;
// thread = (group g of the row, row lane), so a wave reads 1 KB of consecutive memory per instruction instead of one 76 B row
__global__ __launch_bounds__(256) void colsum_partial_narrow_kernel(const float* __restrict__ x, int p4, long P, int C, int rows_per_block,
                                                                    float* __restrict__ part) {
  __shared__ float4 sm[256];
  const int RL = 256 / p4, t = threadIdx.x;
  const int g = t % p4, rl = t / p4;
  const long p0 = (long)blockIdx.x * rows_per_block, p1 = min(P, p0 + rows_per_block);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < RL)
    for (long p = p0 + rl; p < p1; p += RL) {
      const float4 v = PM_LD4(x + (p * p4 + g) * 4);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
  sm[t] = s;
  __syncthreads();
  if (t < p4) {   // fixed-order sum over the row lanes of this group
    for (int k = 1; k < RL; ++k) {
      const float4 v = sm[k * p4 + t];
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    const float o[4] = {s.x, s.y, s.z, s.w};
    for (int j = 0; j < 4; ++j)
      if (t * 4 + j < C) part[(long)blockIdx.x * C + t * 4 + j] = o[j];
  }
}
// 64 channels x 16 lanes per block: lane l adds partials l, l + 16, ... (independent loads in flight), the 16 lane sums are combined in lane
// order (fixed -> deterministic). A single thread per channel walked ~1000 dependent L2 round trips: 38 us for a 76 KB array.
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, int nb, int C, float* __restrict__ out) {
  __shared__ float sm[16][64];
  const int cl = threadIdx.x & 63, lane = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < C)
#pragma unroll 4
    for (int b = lane; b < nb; b += 16) s += part[(long)b * C + c];
  sm[lane][cl] = s;
  __syncthreads();
  if (lane != 0 || c >= C) return;
  s = sm[0][cl];
#pragma unroll
  for (int i = 1; i < 16; ++i) s += sm[i][cl];
  out[c] = s;
}

// stride-2 dgrad: scatter the four per-class results (compact [class][n*Hc*Wc][Cin]) back to dx and fuse the optional add.
template <bool O16>
__global__ __launch_bounds__(256) void dgrad_s2_interleave_kernel(const float* __restrict__ tmp, long class_stride, int valid_mask,
                                                                  float* __restrict__ dx, long xp, int N, int H, int W, int C, const float* __restrict__ add,
                                                                  long add_pitch) {
  const int c4n = C / 4;
  const long total = (long)N * H * W * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long pix = i / c4n;
    const int c = (int)(i - pix * c4n) * 4;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    const int cy = iy & 1, cx = ix & 1, cls = cy * 2 + cx;
    const int Hc = (H - cy + 1) >> 1, Wc = (W - cx + 1) >> 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((valid_mask >> cls) & 1) v = PM_LD4(tmp + cls * class_stride + ((long)(n * Hc + (iy >> 1)) * Wc + (ix >> 1)) * C + c);
    if constexpr (O16) {      // bf16 tier: dx and the fused skip gradient are bf16 tensors (pitches in elements)
      if (add) {
        const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const pm_bf16*>(add) + pix * add_pitch + c);
        v.x += __uint_as_float(q.x << 16), v.y += __uint_as_float(q.x & 0xffff0000u), v.z += __uint_as_float(q.y << 16), v.w += __uint_as_float(q.y & 0xffff0000u);
      }
      *reinterpret_cast<uint2*>(reinterpret_cast<pm_bf16*>(dx) + pix * xp + c) = make_uint2(pm_pack_bf16(v.x, v.y), pm_pack_bf16(v.z, v.w));
    } else {
      if (add) {
        const float4 q = PM_LD4(add + pix * add_pitch + c);
        v.x += q.x, v.y += q.y, v.z += q.z, v.w += q.w;
      }
      PM_ST4(dx + pix * xp + c, v);
    }
  }
}

struct S2Class {   // one input-pixel parity class of a stride-2 dgrad
  int cy, cx, Hc, Wc, ky0, nky, ksy, kx0, nkx, ksx;
  long M;
};
// taps whose (c + pad - k*dil) is even: every second one from (c+pad)&1 when dil is odd; all or none when dil is even
inline void s2_taps(int c, int pad, int dil, int k, int& k0, int& nk, int& ks) {
  if (dil & 1) {
    k0 = (c + pad) & 1, ks = 2, nk = k0 < k ? (k - k0 + 1) / 2 : 0;
  } else {
    k0 = 0, ks = 1, nk = ((c + pad) & 1) ? 0 : k;
  }
}
inline S2Class s2_class(int cls, const pm_tensor* dx, const pm_conv_params* p) {
  S2Class c;
  c.cy = cls >> 1, c.cx = cls & 1;
  c.Hc = (dx->h - c.cy + 1) >> 1, c.Wc = (dx->w - c.cx + 1) >> 1;
  s2_taps(c.cy, p->pad, p->dil, p->kh, c.ky0, c.nky, c.ksy);
  s2_taps(c.cx, p->pad, p->dil, p->kw, c.kx0, c.nkx, c.ksx);
  c.M = (long)dx->n * c.Hc * c.Wc;
  return c;
}

struct Plan {
  int bm, bn;    // block tile: bm 128 / 64, bn 128 / 64 / 32
  int tiles_m, tiles_n, ksplit, kper;
  size_t ws_bytes;
};

// Split-K choice by a small cost model (units: one K-step of one block on an MFMA-bound CU ~ 4096 cycles ~ 2 us):
//   T(ks) = ceil(blocks / 256 CUs) * (K-steps per block + 2 for prologue/epilogue) + workspace round trip + reduce launch.
// It fills the 256 CUs when a problem has few output tiles (wgrad: Cout x taps*Cin) without paying for partial slabs
// when the tile count alone already does.
// ---- PREC 5 routing (round 6): which fp32 launches run as split-operand bf16 MFMA (conv_split.hip). pm_route.split: 0 never, 1 every eligible launch (default).
int g_split_nst = getenv("PM_SPLIT_NST") ? atoi(getenv("PM_SPLIT_NST")) : 1;
int g_split_min_k = getenv("PM_SPLIT_MIN_K") ? atoi(getenv("PM_SPLIT_MIN_K")) : 129;      // shorter forward / data-gradient reductions (64 -> 256 @192^2, 128 -> 512 @96^2) are bound by their
inline bool split_k_ok(int mode, long K) { return mode == MODE_WGRAD || K >= g_split_min_k; }      // output stream, not by the matrix pipe: measured slower on the split path (58 vs 70 TF)
int g_split_tile = getenv("PM_SPLIT_TILE") ? atoi(getenv("PM_SPLIT_TILE")) : 1;      // 1: the split path prefers the 128 x 128 tile (3.7 VALU per MFMA; 64 x 64: 7.3), 0: the fp32 kernel's tile choice
Plan make_plan(int mode, long M, long Nn, long K, bool bf16 = false) {
  Plan best_p{};
  double best = 1e30;
  static const int force_bm = getenv("PM_FORCE_BM") ? atoi(getenv("PM_FORCE_BM")) : 0;   // tuning runs only
  static const int force_bn = getenv("PM_FORCE_BN") ? atoi(getenv("PM_FORCE_BN")) : 0;
  // column tile: weight gradients keep 128-wide tiles; forward / data gradient run best as many small blocks (64 x 64: 70 VGPRs,
  // 18 KB single-stage LDS -> 7-8 resident per CU): -1.4 ms/step of kernel time against 64 x 128 in the per-shape A/B
  // (tools/conv_compare.py), within +-3 % on the few N >= 1024 shapes that preferred the wider tile. PM_WIDE_BN=1 restores 128.
  static const int wide_bn = getenv("PM_WIDE_BN") ? atoi(getenv("PM_WIDE_BN")) : 0;
  int bn = Nn > 64 ? 128 : ((Nn > 32 || bf16) ? 64 : 32);      // no bf16-operand instantiation of the 128 x 32 tile: narrow outputs (19 classes) pad to 64
  // bf16 operands (configs[2], direct algorithm everywhere): the MFMA phase is 16x shorter, the kernel is bound by staging its
  // operands through L2 / LDS, so the 128 x 128 tile (half the operand traffic per FLOP of 64 x 64) wins: 73.1 -> see DESIGN
  const bool big = bf16 || (pm_route.split && g_split_tile && split_k_ok(mode, K));
  if (mode != MODE_WGRAD && !wide_bn && !big && bn == 128) bn = 64;
  if (force_bn && mode != MODE_WGRAD) bn = force_bn;
  const long ksteps = (K + BK - 1) / BK;
  // candidate row tiles: 128 always; 64 halves the tile so that problems with few / awkward tile counts (the 48x48 maps: 144
  // row tiles of 128) spread evenly over the 256 CUs; a 64-row tile is ~8 % less efficient per FLOP (half the MFMAs per
  // fragment read and per barrier).
  for (int bm = 128; bm >= 64; bm -= 64) {
    if (bm == 64 && bn < 64) continue;                            // no 64x32 instantiation (4 waves need >= 2 tiles)
    if (force_bm && mode != MODE_WGRAD && bm != force_bm && !(force_bm == 64 && bn < 64)) continue;
    // forward / data gradient: 64-row blocks (64 x 128: 114 VGPRs, 27.6 KB single-stage LDS -> four per CU) beat 128-row ones (three
    // per CU) on the whole step by ~0.9 ms (same-box A/B), so the larger tile is not a candidate (PM_PREFER_BM64=0 restores it)
    static const int prefer64 = getenv("PM_PREFER_BM64") ? atoi(getenv("PM_PREFER_BM64")) : 1;
    if (prefer64 && !big && !force_bm && mode != MODE_WGRAD && bn >= 64 && bm == 128) continue;
    if (bm == 64 && mode == MODE_WGRAD && M > 64) continue;       // wgrad: 64 rows only for Cout <= 64
    if (bm == 128 && mode == MODE_WGRAD && M <= 64 && bn >= 64) continue;
    const int tiles_m = pm_cdiv(M, bm), tiles_n = pm_cdiv(Nn, bn);
    const long tiles = (long)tiles_m * tiles_n;
    // (a bf16 K-step is ~4 x shorter; a cost unit of 0.5 us for it was measured: it halves the splits of every weight gradient and loses 1.0 ms/step --
    //  r04c vs r04a, the side-stream weight gradients want the parallelism more than they mind the slabs -- so the unit is the same for both types)
    const double unit_us = 2.0 * (bm / 128.0) * (bn / 128.0) * (bm == 64 ? 1.08 : 1.0);
    const long ks_max = std::max<long>(1, std::min<long>(ksteps / 4, 512));
    static const int force_ks = getenv("PM_FORCE_KS") ? atoi(getenv("PM_FORCE_KS")) : 0;
    for (long ks = 1; ks <= ks_max; ++ks) {
      if (force_ks && mode != MODE_WGRAD && ks != force_ks && force_ks <= ks_max) continue;
      const long steps_per = (ksteps + ks - 1) / ks;
      if ((ksteps + steps_per - 1) / steps_per != ks) continue;
      const long blocks = tiles * ks;
      const long per_cu = (blocks + 255) / 256;
      double t = (double)per_cu * (double)(steps_per + 2) * unit_us * (per_cu == 1 ? 1.25 : 1.0);  // a lone block per CU cannot hide its own stalls
      if (ks > 1) t += 2.0 * (double)ks * (double)M * (double)Nn * 4.0 / 3.0e6 + 6.0;             // slab round trip at 3 TB/s + reduce launch
      if (t < best) {
        best = t;
        best_p.bm = bm, best_p.bn = bn, best_p.tiles_m = tiles_m, best_p.tiles_n = tiles_n;
        best_p.ksplit = (int)ks, best_p.kper = (int)(steps_per * BK);
      }
    }
  }
  best_p.ws_bytes = best_p.ksplit > 1 ? (size_t)best_p.ksplit * M * Nn * sizeof(float) : 0;
  return best_p;
}

template <int MODE, int BM, int BN, int WM, int WN, int KM, int PREC, int NST>
void launch_nst(const ConvK& k, dim3 grid, size_t smem, hipStream_t st) {
  static const bool attr_set = [] {  // > 64 KB of dynamic LDS needs an explicit opt-in, once per kernel
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<MODE, BM, BN, WM, WN, KM, PREC, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  const size_t ep_bytes = (size_t)4 * 32 * (BN / WN + 4) * sizeof(float);      // staged epilogue: four wave slabs
  if constexpr (MODE == MODE_FWD && PREC != 1 && BN >= 64) {
    if (k.stats) {
      static const bool attr_set2 = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<MODE, BM, BN, WM, WN, KM, PREC, NST, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
      }();
      (void)attr_set2;
      hipLaunchKernelGGL((conv_igemm_kernel<MODE, BM, BN, WM, WN, KM, PREC, NST, true>), grid, dim3(256), std::max(smem / (NST == 1 ? 2 : 1), ep_bytes), st, k);
      return;
    }
  }
  hipLaunchKernelGGL((conv_igemm_kernel<MODE, BM, BN, WM, WN, KM, PREC, NST>), grid, dim3(256), std::max(smem / (NST == 1 ? 2 : 1), ep_bytes), st, k);
}
// longest reduction (in K-steps per block) that takes the single-stage variant; PM_NST1_STEPS overrides it for tuning runs
inline int nst1_max_steps() {
  static const int v = [] {
    const char* e = getenv("PM_NST1_STEPS");
    return e ? atoi(e) : 64;
  }();
  return v;
}
// native-bf16 weight gradient on ONE LDS stage (PM_WGRAD_NST1=1, tuning): 40 instead of 80 KB per 128 x 128 block -> three resident blocks per CU instead of two
inline bool wgrad_nst1() {
  static const bool v = [] { const char* e = getenv("PM_WGRAD_NST1"); return e && e[0] == '1'; }();
  return v;
}
template <int MODE, int BM, int BN, int WM, int WN, int KM, int PREC>
void launch_prec(const ConvK& k, dim3 grid, size_t smem, hipStream_t st) {
  // single-stage variant only where it is used: fast-path fwd / dgrad with a short reduction per block
  if constexpr (KM == K_FAST && MODE != MODE_WGRAD && BN >= 64) {
    if (k.kper <= nst1_max_steps() * BK) return launch_nst<MODE, BM, BN, WM, WN, KM, PREC, 1>(k, grid, smem, st);
  }
  if constexpr (MODE == MODE_WGRAD && PREC == 4) {
    if (wgrad_nst1()) return launch_nst<MODE, BM, BN, WM, WN, KM, PREC, 1>(k, grid, smem, st);
  }
  launch_nst<MODE, BM, BN, WM, WN, KM, PREC, 2>(k, grid, smem, st);
}
template <int MODE, int BM, int BN, int WM, int WN, int KM>
void launch_inst(const ConvK& k, dim3 grid, size_t smem, hipStream_t st) {
  if constexpr (MODE == MODE_FWD && KM == K_FAST && BN >= 64) {
    if (k.prec == 2) return launch_prec<MODE, BM, BN, WM, WN, KM, 2>(k, grid, smem, st);
  }
  if constexpr (MODE == MODE_WGRAD && BN >= 64) {
    if (k.prec == 3) return launch_prec<MODE, BM, BN, WM, WN, KM, 3>(k, grid, smem, st);
    if (k.prec == 4) return launch_prec<MODE, BM, BN, WM, WN, KM, 4>(k, grid, smem, st);
  }
  if (k.prec != 0) launch_prec<MODE, BM, BN, WM, WN, KM, 1>(k, grid, smem, st);
  else launch_prec<MODE, BM, BN, WM, WN, KM, 0>(k, grid, smem, st);
}
template <int MODE, int BM, int BN, int WM, int WN>
void launch_one(const ConvK& k, dim3 grid, size_t smem, hipStream_t st) {
  if (k.kmode == K_SMALL) launch_inst<MODE, BM, BN, WM, WN, K_SMALL>(k, grid, smem, st);
  else if (k.kmode == K_FAST && MODE != MODE_WGRAD) launch_inst<MODE, BM, BN, WM, WN, (MODE != MODE_WGRAD ? K_FAST : K_MID)>(k, grid, smem, st);
  else launch_inst<MODE, BM, BN, WM, WN, K_MID>(k, grid, smem, st);
}

// ---- optional in-library timing of the implicit-GEMM kernel itself (HIP events on the launch stream) ----------------
struct ProfRec {
  hipEvent_t a, b;
  int mode, bm, bn, km, prec, nst;
  int M, Nn, K, batch, ksplit;
  double flops;
  double bytes;      // algorithmic HBM bytes of the launch: both operands once (their descriptor extents) + the output (split-K: every slab) -- what `traffic` is held against
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

inline bool split_takes(int mode, const Plan& p, const ConvK& k, int batch) {      // batched launches (the Winograd point products: their output stays in L2) at every K
  static const int modes = getenv("PM_SPLIT_MODES") ? atoi(getenv("PM_SPLIT_MODES")) : 7;      // bisecting knob: bit 0 forward form, 1 data gradient, 2 weight gradient
  static const int bsel = getenv("PM_SPLIT_BATCH") ? atoi(getenv("PM_SPLIT_BATCH")) : 0;      // bisecting knob: 1 only batched launches, 2 only unbatched ones
  static const int maxm = getenv("PM_SPLIT_MIN_M") ? atoi(getenv("PM_SPLIT_MIN_M")) : 0;
  if ((bsel == 1 && batch <= 1) || (bsel == 2 && batch > 1) || k.M < maxm) return false;
  return pm_route.split != 0 && ((modes >> mode) & 1) && p.bn >= 64 && !k.io16 && (batch > 1 || split_k_ok(mode, k.K));
}

template <int MODE>
int launch(const ConvK& k0, const Plan& p, hipStream_t st, int batch = 1, double flops = -1.0) {
  ConvK k = k0;
  k.tiles_m = p.tiles_m;
  k.tiles_n = p.tiles_n;
  k.ksplit = p.ksplit;
  k.kper = p.kper;
  dim3 grid(p.tiles_m * p.tiles_n, batch, p.ksplit);
  // staged epilogue: in situ +3 ... +6 % on the convolutions whose output streams to HBM (every unbatched launch), -2 ... -4 % on the
  // batched Winograd GEMMs whose product M[p] stays in L2 / Infinity Cache for the output transform (tools/gpu_env_ab2.sh PM_STAGE_EP)
  if (batch != 1 && !getenv("PM_STAGE_EP")) k.stage_ep = 0;
  if (k.prec == 0 && split_takes(MODE, p, k, batch)) k.prec = 5;
  {      // whole (batch, K-slice) units per XCD (conv_igemm_kernel.h): the unit count's remainder mod 8 must cut evenly into runs of tiles
    static const int batch_xcd = getenv("PM_BATCH_XCD") ? atoi(getenv("PM_BATCH_XCD")) : 3;      // bit 0: batched launches, bit 1: split-K launches
    const int units = batch * p.ksplit, rem = units & 7, ntile = p.tiles_m * p.tiles_n;
    const bool want = p.ksplit > 1 ? (batch_xcd & 2) != 0 : (batch_xcd & 1) != 0;
    k.batch_xcd = want && units >= 8 && (rem == 0 || ((rem == 1 || rem == 2 || rem == 4) && ntile % (8 / rem) == 0));
  }
  ProfRec rec{};
  if (g_prof_on) {
    (void)hipEventCreate(&rec.a), (void)hipEventCreate(&rec.b);
    rec.mode = MODE, rec.bm = p.bm, rec.bn = p.bn, rec.km = (MODE == MODE_WGRAD && k.kmode == K_FAST) ? K_MID : ((k.prec == 5 && MODE != MODE_WGRAD && k.kmode == K_FAST && k.kh * k.kw == 1 && k.stride == 1 && k.pad == 0 && !k.sub) ? K_PW : k.kmode), rec.prec = k.prec, rec.nst = (MODE != MODE_WGRAD && k.kmode == K_FAST && p.bn >= 64 && k.kper <= nst1_max_steps() * BK) ? 1 : 2, rec.M = k.M, rec.Nn = k.Nn, rec.K = k.K, rec.batch = batch, rec.ksplit = p.ksplit, rec.flops = flops >= 0.0 ? flops : 2.0 * (double)k.M * (double)k.Nn * (double)k.K * batch;
    rec.bytes = (double)batch * ((double)k.a_bytes + (double)k.b_bytes + ((k.io16 && p.ksplit == 1) ? 2.0 : 4.0) * (double)k.M * (double)k.Nn * (double)p.ksplit);
    (void)hipEventRecord(rec.a, st);
  }
  if constexpr (MODE != MODE_WGRAD) {      // short pointwise reductions of the fp32 tier stream wave by wave (pwstream.hip): the split path's arithmetic, no tiles
    if (pm_route.split && (k.prec == 0 || k.prec == 5) && !k.io16 && batch == 1 && p.ksplit == 1 && !k.stats && k.kh * k.kw == 1 && k.stride == 1 && k.pad == 0 && !k.sub) {      // fp32 operands only (prec 1 / 2: bf16-operand forms of the older tier)
      pm_gemm_pw g;
      g.A = k.A, g.B = k.B, g.C = k.C, g.bias = k.bias, g.scale = k.scale, g.shift = k.shift, g.residual = k.residual;
      g.a_pitch = MODE == MODE_FWD ? k.x_pitch : k.y_pitch, g.c_pitch = k.c_pitch, g.res_pitch = k.res_pitch;
      g.b_sn = MODE == MODE_FWD ? k.K : 1, g.b_sk = MODE == MODE_FWD ? 1 : k.Nn;
      g.M = k.M, g.Nn = k.Nn, g.K = k.K, g.relu = k.relu;
      if (pm_pwstream_ok(&g)) {
        if (g_prof_on) rec.bm = 32, rec.bn = 64, rec.km = 4, rec.prec = 5, rec.nst = 0;
        const int e = pm_pwstream_launch(&g, st);
        if (g_prof_on) {
          (void)hipEventRecord(rec.b, st);
          g_prof.push_back(rec);
        }
        return e;
      }
    }
  }
  constexpr bool akc = MODE != MODE_WGRAD, bkc = MODE == MODE_FWD;
  if (k.prec == 5) {      // fp32 operands, three-way bf16 split, six products on the bf16 matrix pipe (conv_split.hip)
    const size_t stage = pm_conv_split_stage_bytes(MODE, p.bm, p.bn);
    const bool nst1 = g_split_nst != 2 || 2 * stage > 160 * 1024;
    if (int e = pm_conv_split_launch(MODE, p.bm, p.bn, k, grid.x, grid.y, grid.z, stage, nst1, st)) return e;
    if (g_prof_on) {
      rec.nst = nst1 ? 1 : 2;
      (void)hipEventRecord(rec.b, st);
      g_prof.push_back(rec);
    }
    return pm_check_launch("conv_split");
  }
  const bool tr = MODE == MODE_WGRAD && (k.prec == 3 || k.prec == 4) && p.bn >= 64;      // bf16 [k][m + 32] tiles: 2 bytes per element
  auto smem = [&](int bm, int bn) {
    if (tr) return (size_t)2 * ((k.prec == 4 ? 2 : 1) * BK * (bm + 32) + (k.prec == 4 ? 2 : 1) * BK * (bn + 32)) * 2;
    return (size_t)2 * ((akc ? bm * LDK : BK * bm) + (bkc ? bn * LDK : BK * bn)) * sizeof(float);
  };
  if (p.bm == 64) {
    if (p.bn == 128) launch_one<MODE, 64, 128, 2, 2>(k, grid, smem(64, 128), st);
    else launch_one<MODE, 64, 64, 2, 2>(k, grid, smem(64, 64), st);
  } else if (p.bn == 128) {
    launch_one<MODE, 128, 128, 2, 2>(k, grid, smem(128, 128), st);
  } else if (p.bn == 64) {
    launch_one<MODE, 128, 64, 2, 2>(k, grid, smem(128, 64), st);
  } else {
    launch_one<MODE, 128, 32, 4, 1>(k, grid, smem(128, 32), st);
  }
  if (g_prof_on) {
    (void)hipEventRecord(rec.b, st);
    g_prof.push_back(rec);
  }
  return pm_check_launch("conv_igemm");
}

int check_common(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p) {
  PM_REQUIRE(x && y && p && x->ptr && y->ptr, PM_EINVAL, "conv: null tensor");
  PM_REQUIRE(p->struct_size == (int32_t)sizeof(pm_conv_params), PM_EINVAL, "conv: pm_conv_params.struct_size %d != %zu -- caller built against another pinmem_hip.h (library ABI %d)",
             p->struct_size, sizeof(pm_conv_params), PM_ABI_VERSION);
  PM_REQUIRE(p->kh >= 1 && p->kw >= 1 && p->dil >= 1 && p->pad >= 0, PM_EINVAL, "conv: bad geometry");
  PM_REQUIRE(p->stride == 1 || p->stride == 2, PM_EUNSUPPORTED, "conv: stride %d unsupported (1 or 2)", p->stride);
  const int ho = (x->h + 2 * p->pad - p->dil * (p->kh - 1) - 1) / p->stride + 1;
  const int wo = (x->w + 2 * p->pad - p->dil * (p->kw - 1) - 1) / p->stride + 1;
  PM_REQUIRE(ho == y->h && wo == y->w && x->n == y->n, PM_EINVAL, "conv: output %dx%d does not match geometry (%dx%d)", y->h, y->w, ho, wo);
  PM_REQUIRE((x->dtype == PM_F32 || x->dtype == PM_BF16) && (y->dtype == PM_F32 || y->dtype == PM_BF16), PM_EUNSUPPORTED, "conv: dtype %d / %d", x->dtype, y->dtype);
  for (const pm_tensor* t : {x, y}) {
    if (pm_is_bf16(t)) PM_REQUIRE(pm_vec8(t), PM_EINVAL, "conv: bf16 tensors must be 16B aligned with pitch %% 8 == 0 and channels %% 8 == 0");
    else PM_REQUIRE(pm_vec_ok(t), PM_EINVAL, "conv: tensors must be 16B aligned with pitch %% 4 == 0");
  }
  PM_REQUIRE(x->c % 4 == 0, PM_EUNSUPPORTED, "conv: Cin %% 4 != 0 unsupported (pad the input channels)");
  PM_REQUIRE((y->c % 4 == 0) || (p->kh * p->kw == 1), PM_EUNSUPPORTED, "conv: Cout %% 4 != 0 only for 1x1");
  PM_REQUIRE(x->pitch >= x->c && y->pitch >= y->c, PM_EINVAL, "conv: pitch < channels");
  PM_REQUIRE(pm_pixels(x) * x->pitch < (1ll << 29) && pm_pixels(y) * y->pitch < (1ll << 29) && (int64_t)y->c * x->c * p->kh * p->kw < (1ll << 29),
             PM_EUNSUPPORTED, "conv: tensor too large for 32-bit byte offsets (2 GiB per tensor)");
  return PM_OK;
}

void fill_geom(ConvK& k, const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p) {
  k.N = x->n, k.H = x->h, k.W = x->w, k.Cin = x->c, k.x_pitch = x->pitch;
  k.Ho = y->h, k.Wo = y->w, k.Cout = y->c, k.y_pitch = y->pitch;
  k.kh = p->kh, k.kw = p->kw, k.stride = p->stride, k.pad = p->pad, k.dil = p->dil;
  k.sshift = p->stride == 2 ? 1 : 0;
  k.prec = p->prec != 0 ? 1 : 0;   // prec 2 call sites that cannot take the bf16-operand route fall back to the staged-fp32 bf16 MFMA form
  k.T_eff = p->kh * p->kw, k.tk_w = p->kw, k.ky0 = k.kx0 = 0, k.ksy = k.ksx = 1;
  k.sub = k.sub_cy = k.sub_cx = 0, k.Hc = x->h, k.Wc = x->w;
  k.bias = k.scale = k.shift = k.residual = nullptr;
  k.res_pitch = 0, k.relu = 0, k.stats = nullptr, k.io16 = 0;
  static const int stage_ep = getenv("PM_STAGE_EP") ? atoi(getenv("PM_STAGE_EP")) : 1;
  k.stage_ep = stage_ep;
  static const int spl_prio = getenv("PM_SPLIT_PRIO") ? atoi(getenv("PM_SPLIT_PRIO")) : 1;
  k.spl_prio = spl_prio;
  static const int n_group = getenv("PM_N_GROUP") ? atoi(getenv("PM_N_GROUP")) : 8;
  k.n_group = n_group;
  k.batch_xcd = 0;
  k.a_bs = k.b_bs = k.c_bs = 0;
}

// ---- Winograd F(m x m, 3x3) route for the wide stride-1 "same" 3x3 convolutions (transforms: winograd.hip) ---------------------
// forward:  V = Bt x B ; M[p] = V[p] U[p]^T (P = (m+2)^2 batched GEMMs on the kernel above) ; y = At M A (+ epilogue)
// data grad: the same pipeline on dy with the rotated / transposed filter.
// Taken when the GEMMs are MFMA-bound (both channel counts >= 128: the expanded V / M streams would otherwise dominate) and the
// dilation sub-lattices tile the image without too much padding: the m in {4, 2} with the fewest multiplies per output, if that is
// below 0.6 of the direct algorithm's (m = 4 for d = 1, 2, 6, 12, 18 on the 48x48 maps and d = 1 on 192x192).
struct WinoPlan {
  bool use;
  pm_wino_geom g;
  int Kp, P;
  size_t v_bytes, m_bytes, u_bytes;
  Plan pl;
};
WinoPlan wino_plan(const pm_tensor* xin, int cout, const pm_conv_params* p, bool wgrad = false) {
  WinoPlan wp{};
  if (wgrad && (long)xin->c * cout < 256 * 256) return wp;   // two transforms + slabs per GEMM: pays from 256 x 256 channels up
  if (pm_route.winograd == 0 || p->kh != 3 || p->kw != 3 || p->stride != 1 || p->pad != p->dil || p->prec != 0) return wp;
  const int cin = xin->c;
  if (cin < 128 || cout < 128 || (cout & 3) || (cin & 3)) return wp;
  // multiplies per output relative to the direct algorithm: (m+2)^2 / (9 m^2) x the padding of the sub-lattices to whole tiles
  // (cover). F(4x4) on exactly tiling maps: 0.25; ASPP's d18 on 48x48 (sub-lattices of 3 or 2 rows in one 4-row tile): 0.56, still a
  // measured 1.3x over direct. Above 0.6 the transforms eat the gain.
  int m = 0;
  double best_ratio = 0.6;
  for (int cand = pm_route.winograd; cand >= 2; cand -= 2) {
    const pm_wino_geom g = pm_wino_make_geom(xin->n, xin->h, xin->w, p->dil, cand);
    const double cover = (double)(cand * g.TY * p->dil) * (double)(cand * g.TX * p->dil) / ((double)xin->h * xin->w);
    const double ratio = cover * (cand + 2) * (cand + 2) / (9.0 * cand * cand);
    if (ratio <= best_ratio) best_ratio = ratio, m = cand, wp.g = g;
  }
  if (!m) return wp;
  wp.P = (m + 2) * (m + 2);
  wp.Kp = (cin + BK - 1) / BK * BK;
  if (wp.g.tiles * (long)std::max(wp.Kp, cout) * 4 >= (1ll << 31)) return wp;
  wp.v_bytes = pm_align_up((size_t)wp.P * wp.g.tiles * wp.Kp * sizeof(float), 256);
  wp.m_bytes = pm_align_up((size_t)wp.P * wp.g.tiles * cout * sizeof(float), 256);
  wp.u_bytes = pm_align_up((size_t)wp.P * cout * wp.Kp * sizeof(float), 256);
  static const int wino_bm = getenv("PM_WINO_BM") ? atoi(getenv("PM_WINO_BM")) : 128;
  static const int wino_bn = getenv("PM_WINO_BN") ? atoi(getenv("PM_WINO_BN")) : 128;
  wp.pl.bm = wino_bm, wp.pl.bn = wino_bn;
  // column tile: 64 wide where 128-wide tiles would pad the channel extent by more than 12 % (final1's 304-channel data gradient:
  // 3 x 128 = 384 columns of MFMA work for 304 real ones, 5 x 64 = 320)
  if (wp.pl.bn == 128 && pm_cdiv(cout, 128) * 128 > cout * 1.12 && pm_cdiv(cout, 64) * 64 < pm_cdiv(cout, 128) * 128) wp.pl.bn = 64;
  // few transform tiles (the 48 x 48 maps with dilation: 1152 ... 4608 rows per GEMM): 128-row blocks leave the last round of the 256 CUs
  // half empty; 64 x 128 blocks measured -0.25 ms/step over those layers in situ (tools/gpu_tile_ab2.sh), nothing gained above
  // (the split path keeps 128 rows: 3.7 instead of 5.5 split instructions per MFMA outweigh the half-empty last round -- same box: 52.2 -> 51.7 ms/step, the deep ASPP
  //  products 160-178 -> 188-200 TF)
  if (wp.pl.bm == 128 && wp.g.tiles <= 4608 && !getenv("PM_WINO_BM") && !(pm_route.split && g_split_tile)) wp.pl.bm = 64;
  wp.pl.tiles_m = pm_cdiv(wp.g.tiles, wp.pl.bm), wp.pl.tiles_n = pm_cdiv(cout, wp.pl.bn);
  wp.pl.ksplit = 1, wp.pl.kper = wp.Kp, wp.pl.ws_bytes = 0;
  wp.use = true;
  return wp;
}
inline size_t wino_ws(const WinoPlan& wp) { return wp.v_bytes + wp.m_bytes + wp.u_bytes; }

int wino_conv(const pm_tensor* xin, const float* w, int w_cout, int w_cin, bool dgrad, const pm_tensor* yout, const WinoPlan& wp,
              const pm_conv_epilogue& ep, void* ws, hipStream_t st, float* v_keep = nullptr, float* u_ext = nullptr, bool u_valid = false) {
  float* V = v_keep ? v_keep : (float*)ws;   // forward of a training step: the transformed input is kept for the weight gradient
  float* Mo = (float*)((char*)ws + wp.v_bytes);
  float* U = u_ext ? u_ext : (float*)((char*)ws + wp.v_bytes + wp.m_bytes);   // caller-owned: survives the call (filter-transform cache)
  const int cout = yout->c;
  if (int e = pm_wino_input_xf((const float*)xin->ptr, xin->pitch, xin->c, wp.Kp, wp.g, V, st)) return e;
  if (!(u_ext && u_valid))
    if (int e = pm_wino_filter_xf(w, w_cout, w_cin, wp.Kp, dgrad, wp.g.m, U, st)) return e;
  const pm_tensor xv = {V, 1, 1, (int32_t)wp.g.tiles, wp.Kp, wp.Kp};
  const pm_tensor yv = {Mo, 1, 1, (int32_t)wp.g.tiles, cout, cout};
  const pm_conv_params p1 = {(int32_t)sizeof(pm_conv_params), 1, 1, 1, 0, 1, 0};
  ConvK k;
  fill_geom(k, &xv, &yv, &p1);
  k.A = V, k.B = U, k.C = Mo;
  k.M = (int)wp.g.tiles, k.Nn = cout, k.K = wp.Kp;
  k.a_bytes = (unsigned)(wp.g.tiles * wp.Kp * 4), k.b_bytes = (unsigned)((long)cout * wp.Kp * 4), k.kmode = K_FAST;
  k.c_pitch = cout, k.c_split = 0;
  k.a_bs = wp.g.tiles * wp.Kp, k.b_bs = (long)cout * wp.Kp, k.c_bs = wp.g.tiles * cout;
  // F(4x4): the 36 GEMMs and the output transform as ONE kernel -- M never reaches HBM (winograd.hip, wino_fused_f4_kernel). Correct (the Winograd
  // kernel tests pass on it) and SLOWER: 76.4 vs 64.2 ms/step, 1.59 vs 1.07 ms on final1.3, 1.28 vs 0.52 ms on ASPP d6 (same box, round 3). Holding all
  // 36 points of a tile in one block caps the block tile at 32 x 32 per point (registers), i.e. 8 FLOP per byte staged through LDS instead of the
  // 32 of the 128 x 128 GEMM tile: 144 KB per 16-k slab per CU, one LDS stage, the stage's write phase (~1900 cycles at 79 B/clk) not overlapped with its
  // 4600 MFMA cycles. Bounds measured on the unfused path: a GEMM that never stores M -3.3 ms/step, no output-transform pass either -5.6 ms/step.
  // Kept as an opt-in (PM_WINO_FUSED=1) with its tests; the default is the batched GEMM + wino_output_kernel.
  if (pm_route.winograd_fused && wp.g.m == 4 && wp.Kp % 16 == 0) {
    ProfRec rec{};
    if (g_prof_on) {
      (void)hipEventCreate(&rec.a), (void)hipEventCreate(&rec.b);
      rec.mode = 3, rec.bm = 32, rec.bn = 32, rec.km = 0, rec.prec = 0, rec.nst = 1, rec.M = (int)wp.g.tiles, rec.Nn = cout, rec.K = wp.Kp, rec.batch = wp.P, rec.ksplit = 1;
      rec.flops = 2.0 * wp.P * (double)wp.g.tiles * cout * xin->c;
      (void)hipEventRecord(rec.a, st);
    }
    const int e = pm_wino_fused_f4(V, U, cout, wp.Kp, wp.g, (float*)yout->ptr, yout->pitch, ep.bias, ep.scale, ep.shift, ep.residual, ep.residual_pitch, ep.relu, st);
    if (g_prof_on) {
      (void)hipEventRecord(rec.b, st);
      g_prof.push_back(rec);
    }
    return e;
  }
  if (int e = launch<MODE_FWD>(k, wp.pl, st, wp.P, 2.0 * wp.P * (double)wp.g.tiles * cout * xin->c)) return e;
  return pm_wino_output_xf(Mo, cout, wp.g, (float*)yout->ptr, yout->pitch, ep.bias, ep.scale, ep.shift, ep.residual, ep.residual_pitch, ep.relu, st);
}

// weight gradient: dU[p] = Z[p]^T V[p] (P batched wgrad GEMMs, K = tiles, split-K into slabs) ; dw = Gt (sum of slabs) G
struct WinoWgradPlan {
  Plan pl;
  size_t slab_bytes;
};
WinoWgradPlan wino_wgrad_plan(const WinoPlan& wp, int cout) {
  WinoWgradPlan q{};
  q.pl.bm = 128, q.pl.bn = 128;
  if (pm_cdiv(wp.Kp, 128) * 128 > wp.Kp * 1.12 && pm_cdiv(wp.Kp, 64) * 64 < pm_cdiv(wp.Kp, 128) * 128) q.pl.bn = 64;   // Kp = 320: 5 x 64, not 3 x 128
  q.pl.tiles_m = pm_cdiv(cout, 128), q.pl.tiles_n = pm_cdiv(wp.Kp, q.pl.bn);
  const long ksteps = (wp.g.tiles + BK - 1) / BK;
  const long per_split = (long)q.pl.tiles_m * q.pl.tiles_n * wp.P;
  long ks = std::max<long>(1, std::min<long>((768 + per_split / 2) / per_split, ksteps / 8));   // ~3 rounds of 256 CUs, >= 8 K-steps per block
  const long steps_per = (ksteps + ks - 1) / ks;
  ks = (ksteps + steps_per - 1) / steps_per;
  q.pl.ksplit = (int)ks, q.pl.kper = (int)(steps_per * BK), q.pl.ws_bytes = 0;
  q.slab_bytes = pm_align_up((size_t)ks * wp.P * cout * wp.Kp * sizeof(float), 256);
  return q;
}
inline size_t wino_wgrad_ws(const WinoPlan& wp, const WinoWgradPlan& q) { return wp.v_bytes + wp.m_bytes + q.slab_bytes; }

int wino_wgrad(const pm_tensor* x, const pm_tensor* dy, float* dw, const WinoPlan& wp, const WinoWgradPlan& q, void* ws, hipStream_t st,
               float* v_kept = nullptr) {
  float* V = v_kept ? v_kept : (float*)ws;
  float* Z = (float*)((char*)ws + wp.v_bytes);
  float* slab = (float*)((char*)ws + wp.v_bytes + wp.m_bytes);
  const int cout = dy->c;
  if (!v_kept)
    if (int e = pm_wino_input_xf((const float*)x->ptr, x->pitch, x->c, wp.Kp, wp.g, V, st)) return e;
  if (int e = pm_wino_dy_xf((const float*)dy->ptr, dy->pitch, cout, wp.g, Z, st)) return e;
  const pm_tensor xv = {V, 1, 1, (int32_t)wp.g.tiles, wp.Kp, wp.Kp};
  const pm_tensor zv = {Z, 1, 1, (int32_t)wp.g.tiles, cout, cout};
  const pm_conv_params p1 = {(int32_t)sizeof(pm_conv_params), 1, 1, 1, 0, 1, 0};
  ConvK k;
  fill_geom(k, &xv, &zv, &p1);
  k.A = Z, k.B = V, k.C = slab;
  k.M = cout, k.Nn = wp.Kp, k.K = (int)wp.g.tiles;
  k.a_bytes = (unsigned)(wp.g.tiles * cout * 4), k.b_bytes = (unsigned)(wp.g.tiles * wp.Kp * 4), k.kmode = K_MID;
  k.c_pitch = wp.Kp, k.c_split = (long)wp.P * cout * wp.Kp;
  k.a_bs = wp.g.tiles * cout, k.b_bs = wp.g.tiles * wp.Kp, k.c_bs = (long)cout * wp.Kp;
  if (int e = launch<MODE_WGRAD>(k, q.pl, st, wp.P, 2.0 * wp.P * (double)wp.g.tiles * cout * x->c)) return e;
  return pm_wino_dw_xf(slab, q.pl.ksplit, cout, x->c, wp.Kp, wp.g.m, dw, st);
}

// ---- prec = 2: bf16 operands in HBM (bf16.hip) + bf16 LDS tiles, forward and stride-1 data gradient -----------------------------
// The convolution is handed to the implicit-GEMM kernel as an fp32-typed problem with HALF the input channels: a float of the view is a
// pair of bf16 channels, a 32-float K-slab is 64 bf16 k-values. Input channels are padded to a multiple of 64 (zero-filled by the cast)
// so that a slab never straddles a tap and the wave-uniform K-state (K_FAST) applies to every layer.
struct Bf16Plan {
  bool use;
  bool c16;                  // the LDS-DMA kernel (conv16.hip) takes the call; k16 holds its geometry and plan (pointers / epilogue filled at launch)
  pm_conv16 k16;
  bool inplace;              // the gathered operand already is a bf16 tensor whose rows can be gathered where they lie (no cast / pad pass)
  int Cp;                    // padded input channels (bf16 elements)
  long M, Nn, Kf;            // GEMM extents, K in float units (= taps * Cp / 2)
  size_t xb_bytes, wb_bytes;
  Plan pl;
};
Bf16Plan bf16_plan(const pm_tensor* xin, const pm_tensor* yout, const pm_conv_params* p) {
  Bf16Plan b{};
  if ((p->prec != 2 && !pm_is_bf16(xin) && !pm_is_bf16(yout)) || xin->c < 32) return b;      // the 3(4)-channel stem would be padded 16x
  b.Cp = (xin->c + 63) / 64 * 64;
  const long T = (long)p->kh * p->kw;
  b.M = pm_pixels(yout), b.Nn = yout->c, b.Kf = T * b.Cp / 2;
  // round 4, bf16 activations: a bf16 tensor is gathered in place when a K-slab of 64 channels never straddles a tap, i.e. its channel count is a multiple of 64,
  // or the caller vouches for zero-filled pad channels inside the pitch (PM_TF_ZERO_PAD64: the decoder's 304-channel concat buffer)
  b.inplace = pm_is_bf16(xin) && (xin->c % 64 == 0 || ((xin->flags & PM_TF_ZERO_PAD64) && xin->pitch >= b.Cp));
  const size_t xb = b.inplace ? (size_t)pm_pixels(xin) * xin->pitch * 2 : (size_t)pm_pixels(xin) * b.Cp * 2, wb = (size_t)yout->c * T * b.Cp * 2;
  if (xb >= (1ull << 31) || wb >= (1ull << 31)) return b;         // 32-bit byte offsets
  b.xb_bytes = b.inplace ? 0 : pm_align_up(xb, 256), b.wb_bytes = pm_align_up(wb, 256);
  b.pl = make_plan(MODE_FWD, b.M, b.Nn, b.Kf, true);
  if (b.pl.bn < 64) return b;                                     // no bf16 instantiation of the 128 x 32 tile
  b.use = true;
  // round 4: the LDS-DMA kernel for everything but tiny row counts (the image-pooling branch: M = batch) -- PM_CONV16=0 keeps the register-staged kernel (A/B runs)
  const int c16_on = pm_route.conv16;
  static const int c16_min_m = getenv("PM_CONV16_MIN_M") ? atoi(getenv("PM_CONV16_MIN_M")) : 256;
  b.c16 = false;
  if (c16_on && b.M >= c16_min_m && p->stride >= 1) {
    pm_conv16& k = b.k16;
    k = pm_conv16{};
    k.N = xin->n, k.H = xin->h, k.W = xin->w, k.Ho = yout->h, k.Wo = yout->w;
    k.a_pitch = b.inplace ? xin->pitch : b.Cp, k.Cp = b.Cp;
    k.kh = p->kh, k.kw = p->kw, k.stride = p->stride, k.pad = p->pad, k.dil = p->dil;
    k.M = (int)b.M, k.Nn = (int)b.Nn, k.K = (int)(T * b.Cp), k.ksteps = k.K / 64;
    k.c_pitch = yout->pitch, k.c_f32 = pm_is_bf16(yout) ? 0 : 1;
    pm_conv16_plan(&k);
    // Where it wins (per-shape A/B, profiles/r04c_conv16_vs_regstaged.txt): 64-row tiles -- the 48 x 48 and 96 x 96 maps, 525 -> 655-700 TF on their 3x3s -- and the
    // single-K-step 1x1s (four blocks per CU). On 128 x 128 tiles the register-staged kernel's scheduled interleave is still ~10 % ahead (883 vs 982 TF on the
    // decoder's 3x3): those stay there. PM_CONV16=2 forces the LDS-DMA kernel everywhere.
    // (second form of the kernel, buffer-descriptor fetches: it also takes the long reductions the register-staged planner would run on 64-row tiles -- the
    //  3x3 512 -> 1024-wide data gradients at 48 x 48: 596 -> 877 TF on 128 x 128 -- while the short 1x1 reductions stay with the single-stage register-staged form)
    b.c16 = c16_on == 2 || k.wide || k.bm == 64 || k.ksteps_per == 1 || (b.pl.bm == 64 && k.ksteps >= 16);
  }
  return b;
}
inline size_t bf16_ws(const Bf16Plan& b) {
  return b.xb_bytes + b.wb_bytes + std::max(pm_align_up(b.pl.ws_bytes, 256), b.c16 ? pm_conv16_slab_bytes(&b.k16) : (size_t)0);
}

// xin: the tensor the GEMM gathers from (x for the forward pass, dy for the data gradient); `rotate`: w is read as the rotated /
// transposed filter; pe: geometry of the convolution actually run (the data gradient of a stride-1 convolution is a stride-1
// convolution of dy with pad' = dil (k - 1) - pad).
int conv_bf16(const pm_tensor* xin, const float* w, int w_cout, int w_cin, bool rotate, const pm_tensor* yout, const pm_conv_params* pe, const Bf16Plan& b,
              const pm_conv_epilogue& e0, void* ws, hipStream_t st, char* wb_ext = nullptr, bool wb_valid = false) {
  char* xb = b.inplace ? (char*)xin->ptr : (char*)ws;
  char* wb = wb_ext ? wb_ext : (char*)ws + b.xb_bytes;   // caller-owned: survives the call (weight-cast cache)
  float* slab = (float*)((char*)ws + b.xb_bytes + b.wb_bytes);
  const int T = pe->kh * pe->kw;
  if (!b.inplace) {
    if (pm_is_bf16(xin)) {      // bf16 rows whose channel count is not a multiple of 64 (48): copied once with zero pad channels
      if (int e = pm16_pad_rows((const pm_bf16*)xin->ptr, xin->pitch, xin->c, b.Cp, pm_pixels(xin), (pm_bf16*)xb, st)) return e;
    } else if (int e = pm_bf16_cast_rows((const float*)xin->ptr, xin->pitch, xin->c, b.Cp, pm_pixels(xin), xb, st)) return e;
  }
  if (!(wb_ext && wb_valid))
    if (int e = pm_bf16_cast_weights(w, w_cout, T, w_cin, b.Cp, rotate, wb, st)) return e;
  const long xpitch16 = b.inplace ? xin->pitch : b.Cp;                       // bf16 elements between pixels
  if (b.c16) {
    const bool o16 = pm_is_bf16(yout);
    const bool ep_any = e0.bias || e0.scale || e0.residual || e0.relu;
    // what the kernel's two epilogues cover: bf16 rows of whole 16-byte groups with the full fused epilogue, or fp32 rows with at most a bias; no statistics
    const bool ok = (!e0.bn_partials || o16) &&
                    (o16 ? ((yout->c | yout->pitch | (e0.residual ? e0.residual_pitch : 0)) & 7) == 0 && pm_aligned16(yout->ptr) && pm_aligned16(e0.residual)
                         : !(e0.scale || e0.residual || e0.relu));
    if (ok) {
      pm_conv16 k = b.k16;
      k.A = (const pm_bf16*)xb, k.B = (const pm_bf16*)wb;
      // EXECUTED FLOPs: the K-steps of filter rows that no row of a tile can see are skipped by the kernel (dilated ASPP branches) and are not counted
      const double fl = 2.0 * (double)b.M * (double)b.Nn * (double)T * (double)xin->c * (g_prof_on ? pm_conv16_executed_fraction(&k) : 1.0);
      ProfRec rec{};
      if (g_prof_on) {
        (void)hipEventCreate(&rec.a), (void)hipEventCreate(&rec.b);
        rec.mode = k.wide ? 5 : 4, rec.bm = k.bm, rec.bn = k.bn, rec.km = pm_conv16w_persistent(&k), rec.prec = 5, rec.nst = k.wide ? 3 : (k.ksteps_per == 1 ? 1 : 2), rec.M = k.M, rec.Nn = k.Nn, rec.K = k.K / 2, rec.batch = 1,
        rec.ksplit = k.ksplit, rec.flops = fl;
        (void)hipEventRecord(rec.a, st);
      }
      int e;
      if (k.ksplit > 1) {
        k.C = slab;
        e = pm_conv16_launch(&k, st);
      } else {
        k.C = yout->ptr;
        k.bias = e0.bias, k.scale = e0.scale, k.shift = e0.shift, k.residual = e0.residual, k.res_pitch = e0.residual_pitch, k.relu = e0.relu;
        k.stats = o16 ? e0.bn_partials : nullptr;
        e = pm_conv16_launch(&k, st);
      }
      if (g_prof_on) {
        (void)hipEventRecord(rec.b, st);
        g_prof.push_back(rec);
      }
      if (e) return e;
      if (k.ksplit > 1)
        return splitk_reduce(slab, k.ksplit, b.M, b.Nn, (float*)yout->ptr, (long)yout->pitch, e0.bias, e0.scale, e0.shift, e0.residual, (long)e0.residual_pitch, e0.relu,
                             st, o16);
      (void)ep_any;
      return PM_OK;
    }
  }
  const pm_tensor xv = {xb, xin->n, xin->h, xin->w, b.Cp / 2, xpitch16 / 2};   // fp32-typed view: one float = two bf16 channels
  ConvK k;
  fill_geom(k, &xv, yout, pe);
  k.prec = 2;
  k.A = (const float*)xb, k.B = (const float*)wb;
  k.M = (int)b.M, k.Nn = (int)b.Nn, k.K = (int)b.Kf;
  k.a_bytes = (unsigned)((size_t)pm_pixels(xin) * xpitch16 * 2), k.b_bytes = (unsigned)((size_t)yout->c * T * b.Cp * 2), k.kmode = K_FAST;
  const double flops = 2.0 * (double)b.M * (double)b.Nn * (double)T * (double)xin->c;
  k.io16 = pm_is_bf16(yout) ? 1 : 0;      // bf16 output (and bf16 residual / skip gradient: the caller checked the types)
  if (b.pl.ksplit > 1) {
    k.C = slab, k.c_pitch = b.Nn, k.c_split = b.M * b.Nn;
    if (int e = launch<MODE_FWD>(k, b.pl, st, 1, flops)) return e;
    return splitk_reduce(slab, b.pl.ksplit, b.M, b.Nn, (float*)yout->ptr, (long)yout->pitch, e0.bias, e0.scale, e0.shift, e0.residual, (long)e0.residual_pitch,
                         e0.relu, st, k.io16 != 0);
  }
  k.C = (float*)yout->ptr, k.c_pitch = yout->pitch, k.c_split = 0;
  k.bias = e0.bias, k.scale = e0.scale, k.shift = e0.shift, k.residual = e0.residual, k.res_pitch = e0.residual_pitch, k.relu = e0.relu;
  k.stats = e0.bn_partials;
  return launch<MODE_FWD>(k, b.pl, st, 1, flops);
}
inline pm_conv_params dgrad_as_fwd(const pm_conv_params* p) {
  pm_conv_params q = *p;
  q.stride = 1, q.pad = p->dil * (p->kh - 1) - p->pad;
  return q;
}
inline bool dgrad_bf16_ok(const pm_conv_params* p) { return p->prec == 2 && p->stride == 1 && p->kh == p->kw && p->dil * (p->kh - 1) - p->pad >= 0; }
// the bf16 tier is entered by tensor type as well as by pm_conv_params.prec: a call with a bf16 tensor always runs the bf16-MFMA forms
inline pm_conv_params tier_params(const pm_conv_params* p, const pm_tensor* a, const pm_tensor* b) {
  pm_conv_params q = *p;
  if (pm_is_bf16(a) || pm_is_bf16(b)) q.prec = 2;
  return q;
}
// dense fp32 copy of a bf16 tensor in the workspace (the few mixed-type call sites: stride-2 data gradients, stem / class-head weight gradients, bias gradients)
inline size_t upcast_bytes(const pm_tensor* t) { return pm_is_bf16(t) ? pm_align_up((size_t)pm_pixels(t) * t->c * sizeof(float), 256) : 0; }
inline int upcast(const pm_tensor* t, void* dst, pm_tensor* out, hipStream_t st) {
  *out = *t;
  if (!pm_is_bf16(t)) return PM_OK;
  out->ptr = dst, out->pitch = t->c, out->dtype = PM_F32, out->flags = 0;
  return pm16_to_f32((const pm_bf16*)t->ptr, t->pitch, t->c, pm_pixels(t), (float*)dst, t->c, st);
}

// prec = 2 weight gradient: dw[co][(t, ci)] = sum_p dyt[co][p] * xt[t][ci][p] over the output pixels p -- with both operands transposed to
// pixel-contiguous bf16 (pm_bf16_transpose_taps: per tap the input pixel each output pixel sees, zero outside the image) this is the same
// k-contiguous GEMM as the forward pass (M = Cout, N = taps * Cin, K = pixels), split over K with fp32 slabs and a fixed-order reduce.
// Off by default: the nine shifted copies of x a 3x3 layer needs cost more HBM time than the faster GEMM saves (bench.py --dtype bf16: GEMM
// time 26.8 -> 23.0 ms/step, step 45.7 -> 48.1 ms); pm_set_bf16_wgrad(1) enables it (kernel tests, and the day the producers emit the copies).
// ---- stride-2 data gradient on the bf16 tier, native (round 4): each input-pixel parity class is a stride-1 FORWARD convolution of the bf16 dy with the class'
// sub-filter (taps ky0 + 2 i, kx0 + 2 j, visited in flipped order), run by the LDS-DMA kernel on dy where it lies; the compact class results stay fp32 (the sum
// with the fused skip gradient is rounded ONCE, as in every other data gradient of the tier) and are interleaved into dx. Replaces: widening dy to fp32 and four
// fp32-row gathers with per-fragment rounding.
struct S2Native {
  bool ok;
  pm_conv16 k[4];
  bool valid[4];
  PmS2Classes cl;
  size_t wb_off[4], out_off[4], slab_off, total;
  long class_stride;      // floats between class buffers
};
S2Native s2_native_plan(const pm_tensor* dy, const pm_tensor* dx, const pm_conv_params* p) {
  S2Native s{};
  static const int on = getenv("PM_S2_NATIVE16") ? atoi(getenv("PM_S2_NATIVE16")) : 1;
  if (!on || !pm_route.conv16 || p->stride != 2 || p->dil != 1 || !pm_is_bf16(dy) || !pm_is_bf16(dx) || !pm_vec8(dy) || !pm_vec8(dx) || dy->c % 64 || dx->c % 8) return s;
  size_t off = 0, out_bytes = 0, slab = 0;
  for (int cls = 0; cls < 4; ++cls) {
    const S2Class c = s2_class(cls, dx, p);
    s.valid[cls] = c.M > 0 && c.nky * c.nkx > 0;
    s.cl.ky0[cls] = c.ky0, s.cl.nky[cls] = s.valid[cls] ? c.nky : 0, s.cl.kx0[cls] = c.kx0, s.cl.nkx[cls] = s.valid[cls] ? c.nkx : 0;
    out_bytes = std::max(out_bytes, (size_t)c.M * dx->c * sizeof(float));
    if (!s.valid[cls]) continue;
    // dy row of tap i' (flipped order) = py + (cy + pad - ky0) / 2 - (nky - 1) + i'  ->  the forward form's pad is (nky - 1) - (cy + pad - ky0) / 2, the same in x
    const int pady = (c.nky - 1) - (c.cy + p->pad - c.ky0) / 2, padx = (c.nkx - 1) - (c.cx + p->pad - c.kx0) / 2;
    if (pady != padx || pady < 0) return s;      // one pad for both axes in the kernel: 3x3 / pad 1 and 1x1 / pad 0 give 0 everywhere
    pm_conv16& k = s.k[cls];
    k = pm_conv16{};
    k.N = dy->n, k.H = dy->h, k.W = dy->w, k.Ho = c.Hc, k.Wo = c.Wc;
    k.a_pitch = dy->pitch, k.Cp = dy->c;
    k.kh = c.nky, k.kw = c.nkx, k.stride = 1, k.pad = pady, k.dil = 1;
    k.M = (int)c.M, k.Nn = dx->c, k.K = c.nky * c.nkx * dy->c, k.ksteps = k.K / 64;
    k.c_pitch = dx->c, k.c_f32 = 1;
    pm_conv16_plan(&k);
    s.wb_off[cls] = off;
    off += pm_align_up((size_t)dx->c * c.nky * c.nkx * dy->c * 2, 256);
    slab = std::max(slab, pm_conv16_slab_bytes(&k));
  }
  out_bytes = pm_align_up(out_bytes, 256);
  for (int cls = 0; cls < 4; ++cls) s.out_off[cls] = off + cls * out_bytes;
  s.class_stride = (long)(out_bytes / sizeof(float));
  s.slab_off = off + 4 * out_bytes;
  s.total = s.slab_off + slab + 256;
  s.ok = true;
  return s;
}
int dgrad_s2_bf16(const pm_tensor* dy, const float* w, const pm_tensor* dx, const pm_conv_params* p, const pm_tensor* add, S2Native& s, void* ws, hipStream_t st) {
  for (int cls = 0; cls < 4; ++cls) s.cl.out[cls] = (char*)ws + s.wb_off[cls];
  if (int e = pm_bf16_cast_weights_s2(w, dy->c, p->kh, p->kw, dx->c, dy->c, &s.cl, st)) return e;
  float* slab = (float*)((char*)ws + s.slab_off);
  for (int cls = 0; cls < 4; ++cls) {
    if (!s.valid[cls]) continue;
    pm_conv16 k = s.k[cls];
    k.A = (const pm_bf16*)dy->ptr, k.B = (const pm_bf16*)((char*)ws + s.wb_off[cls]);
    void* out = (char*)ws + s.out_off[cls];
    ProfRec rec{};
    if (g_prof_on) {
      (void)hipEventCreate(&rec.a), (void)hipEventCreate(&rec.b);
      rec.mode = 4, rec.bm = k.bm, rec.bn = k.bn, rec.km = 0, rec.prec = 5, rec.nst = k.ksteps_per == 1 ? 1 : 2, rec.M = k.M, rec.Nn = k.Nn, rec.K = k.K / 2, rec.batch = 1,
      rec.ksplit = k.ksplit, rec.flops = 2.0 * (double)k.M * (double)k.Nn * (double)k.K;
      (void)hipEventRecord(rec.a, st);
    }
    k.C = k.ksplit > 1 ? (void*)slab : out;
    const int e = pm_conv16_launch(&k, st);
    if (g_prof_on) {
      (void)hipEventRecord(rec.b, st);
      g_prof.push_back(rec);
    }
    if (e) return e;
    if (k.ksplit > 1)
      if (int e2 = splitk_reduce(slab, k.ksplit, k.M, k.Nn, (float*)out, (long)dx->c, nullptr, nullptr, nullptr, nullptr, 0l, 0, st, false)) return e2;
  }
  const int valid_mask = s.valid[0] | (s.valid[1] << 1) | (s.valid[2] << 2) | (s.valid[3] << 3);
  const long total = pm_pixels(dx) * (dx->c / 4);
  hipLaunchKernelGGL(dgrad_s2_interleave_kernel<true>, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st, (const float*)((char*)ws + s.out_off[0]),
                     s.class_stride, valid_mask, (float*)dx->ptr, (long)dx->pitch, dx->n, dx->h, dx->w, dx->c, add ? (const float*)add->ptr : nullptr,
                     add ? (long)add->pitch : 0l);
  return pm_check_launch("dgrad_s2_interleave(native bf16 classes)");
}

struct Bf16WgradPlan {
  bool use;
  long P, M, Nn, Kf;
  size_t dyt_bytes, xt_bytes;
  Plan pl;
};
Bf16WgradPlan bf16_wgrad_plan(const pm_tensor* x, const pm_tensor* dy, const pm_conv_params* p) {
  Bf16WgradPlan b{};
  if (p->prec != 2 || !pm_route.bf16_wgrad || x->c < 32 || dy->c < 32) return b;
  b.P = pm_pixels(dy);
  if (b.P % 64) return b;                                           // whole K-slabs of 64 pixels
  const long T = (long)p->kh * p->kw;
  b.M = dy->c, b.Nn = T * x->c, b.Kf = b.P / 2;
  const size_t dyt = (size_t)dy->c * b.P * 2, xt = (size_t)T * x->c * b.P * 2;
  if (dyt >= (1ull << 31) || xt >= (1ull << 31)) return b;
  b.dyt_bytes = pm_align_up(dyt, 256), b.xt_bytes = pm_align_up(xt, 256);
  b.pl = make_plan(MODE_FWD, b.M, b.Nn, b.Kf, true);
  if (b.pl.bn < 64) return b;
  b.use = true;
  return b;
}
inline size_t bf16_wgrad_ws(const Bf16WgradPlan& b) { return b.dyt_bytes + b.xt_bytes + pm_align_up(b.pl.ws_bytes, 256); }

int conv_wgrad_bf16(const pm_tensor* x, const pm_tensor* dy, float* dw, const pm_conv_params* p, const Bf16WgradPlan& b, void* ws, hipStream_t st) {
  char* dyt = (char*)ws;
  char* xt = dyt + b.dyt_bytes;
  float* slab = (float*)(xt + b.xt_bytes);
  // dy -> dyt[Cout][P]: the plain transpose (one "tap", stride 1, no padding, input grid = output grid)
  if (int e = pm_bf16_transpose_taps((const float*)dy->ptr, dy->pitch, dy->c, dy->n, dy->h, dy->w, dy->h, dy->w, 1, 1, 1, 0, 1, dyt, st)) return e;
  if (int e = pm_bf16_transpose_taps((const float*)x->ptr, x->pitch, x->c, x->n, x->h, x->w, dy->h, dy->w, p->kh, p->kw, p->stride, p->pad, p->dil, xt, st)) return e;
  const int Pf = (int)(b.P / 2);
  const pm_tensor av = {dyt, 1, 1, (int32_t)b.M, Pf, Pf};          // fp32-typed views: rows of P / 2 floats
  const pm_tensor cv = {dw, 1, 1, (int32_t)b.M, (int32_t)b.Nn, b.Nn};
  const pm_conv_params p1 = {(int32_t)sizeof(pm_conv_params), 1, 1, 1, 0, 1, 2};
  ConvK k;
  fill_geom(k, &av, &cv, &p1);
  k.prec = 2;
  k.A = (const float*)dyt, k.B = (const float*)xt;
  k.M = (int)b.M, k.Nn = (int)b.Nn, k.K = (int)b.Kf;
  k.a_bytes = (unsigned)((size_t)b.M * b.P * 2), k.b_bytes = (unsigned)((size_t)b.Nn * b.P * 2), k.kmode = K_FAST;
  const double flops = 2.0 * (double)b.M * (double)b.Nn * (double)b.P;
  if (b.pl.ksplit > 1) {
    k.C = slab, k.c_pitch = b.Nn, k.c_split = b.M * b.Nn;
    if (int e = launch<MODE_FWD>(k, b.pl, st, 1, flops)) return e;
    return splitk_reduce(slab, b.pl.ksplit, b.M, b.Nn, dw, b.Nn, nullptr, nullptr, nullptr, nullptr, 0l, 0, st);
  }
  k.C = dw, k.c_pitch = b.Nn, k.c_split = 0;
  return launch<MODE_FWD>(k, b.pl, st, 1, flops);
}

void gemm_dims(int which, const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p, long& M, long& Nn, long& K) {
  const long T = (long)p->kh * p->kw;
  if (which == MODE_FWD) M = pm_pixels(y), Nn = y->c, K = T * x->c;
  else if (which == MODE_DGRAD) M = pm_pixels(x), Nn = x->c, K = T * y->c;
  else M = y->c, Nn = T * x->c, K = pm_pixels(y);
}

}  // namespace

// ---- the library's routing state: ONE struct (include/pinmem_hip.h pm_routing), initialised from the PM_* environment at load, replaced as a whole by pm_routing_set.
// The pm_set_* entry points below are thin wrappers that change one field (kernel tests, A/B runs).
pm_routing pm_route = {
    (int32_t)sizeof(pm_routing),
    4,                                                                  // winograd: prefer F(4x4,3x3)
    getenv("PM_WINO_FUSED") ? atoi(getenv("PM_WINO_FUSED")) : 0,        // winograd_fused
    getenv("PM_CONV16") ? atoi(getenv("PM_CONV16")) : 1,                // conv16
    getenv("PM_C16W") ? atoi(getenv("PM_C16W")) : 1,                    // conv16_wide
    getenv("PM_C16P") ? atoi(getenv("PM_C16P")) : 1,                    // conv16_persistent
    getenv("PM_WGRAD16") ? atoi(getenv("PM_WGRAD16")) : 1,              // wgrad16
    0,                                                                  // bf16_wgrad
    getenv("PM_SPLIT") ? atoi(getenv("PM_SPLIT")) : 1,                  // split
};
extern "C" int pm_routing_get(pm_routing* out) {
  PM_REQUIRE(out && out->struct_size == (int32_t)sizeof(pm_routing), PM_EINVAL, "pm_routing_get: struct_size %d != %zu (library ABI %d)", out ? out->struct_size : -1,
             sizeof(pm_routing), PM_ABI_VERSION);
  *out = pm_route;
  return PM_OK;
}
extern "C" int pm_routing_set(const pm_routing* r) {
  PM_REQUIRE(r && r->struct_size == (int32_t)sizeof(pm_routing), PM_EINVAL, "pm_routing_set: struct_size %d != %zu (library ABI %d)", r ? r->struct_size : -1, sizeof(pm_routing),
             PM_ABI_VERSION);
  PM_REQUIRE(r->winograd == 0 || r->winograd == 2 || r->winograd == 4, PM_EINVAL, "pm_routing_set: winograd %d (0, 2 or 4)", r->winograd);
  PM_REQUIRE(r->conv16 >= 0 && r->conv16 <= 2 && r->conv16_wide >= 0 && r->conv16_wide <= 3, PM_EINVAL, "pm_routing_set: conv16 %d / conv16_wide %d", r->conv16, r->conv16_wide);
  pm_route = *r;
  return PM_OK;
}
extern "C" int pm_set_winograd(int mode) {
  PM_REQUIRE(mode == 0 || mode == 2 || mode == 4, PM_EINVAL, "pm_set_winograd: mode %d (0 off, 2 F(2x2,3x3), 4 prefer F(4x4,3x3))", mode);
  pm_route.winograd = mode;
  return PM_OK;
}
extern "C" int pm_set_conv16(int on) {
  PM_REQUIRE(on >= 0 && on <= 8, PM_EINVAL, "pm_set_conv16: %d (0 register-staged, 1 per shape, 2 LDS-DMA everywhere / narrow tiles only, 3 LDS-DMA everywhere / wide tiles "
             "wherever the shape allows, 4 per shape without the wide kernel, 5 / 6 = 1 (the streaming 1x1 kernel left the library in round 6), 7 as 3 with the 256 x 256 tile, "
             "8 as 3 with one block per tile instead of the persistent ring)", on);
  static const pm_routing defaults = pm_route;      // the environment's defaults, as read at load
  pm_route.conv16 = (on == 3 || on == 7 || on == 8) ? 2 : ((on == 4 || on == 5 || on == 6) ? 1 : on);
  pm_route.conv16_wide = (on == 3 || on == 8) ? 2 : (on == 7 ? 3 : ((on == 2 || on == 4) ? 0 : defaults.conv16_wide));
  pm_route.conv16_persistent = on == 8 ? 0 : (on == 3 ? 1 : defaults.conv16_persistent);
  return PM_OK;
}
extern "C" int pm_set_winograd_fused(int on) {
  pm_route.winograd_fused = on != 0;
  return PM_OK;
}
extern "C" int pm_set_wgrad16(int on) {
  pm_route.wgrad16 = on != 0;
  return PM_OK;
}
extern "C" int pm_set_split(int on) {
  pm_route.split = on ? 1 : 0;
  return PM_OK;
}
extern "C" int pm_set_bf16_wgrad(int on) {
  pm_route.bf16_wgrad = on != 0;
  return PM_OK;
}
extern "C" int pm_profile_enable(int on) {
  g_prof_on = on != 0;
  return PM_OK;
}
// Sums (and clears) the records of one kernel instantiation conv_igemm_kernel<mode, bm, bn, .., km>: mode 0 fwd / 1 dgrad /
// 2 wgrad, bm x bn the block tile, km the K-state variant, nst the LDS stage count. Negative values act as wildcards. Synchronises on the recorded events only.
extern "C" int pm_profile_read_prec(int mode, int bm, int bn, int km, int nst, int prec, double* total_ms, double* total_flops, int64_t* launches, int clear);
extern "C" int pm_profile_read(int mode, int bm, int bn, int km, int nst, double* total_ms, double* total_flops, int64_t* launches, int clear) {
  return pm_profile_read_prec(mode, bm, bn, km, nst, -1, total_ms, total_flops, launches, clear);
}
// the same with the operand form of the instantiation as a sixth key (prec: 0 fp32, 1 bf16 staged, 2 bf16 operands; negative = any)
extern "C" int pm_profile_read_prec(int mode, int bm, int bn, int km, int nst, int prec, double* total_ms, double* total_flops, int64_t* launches, int clear) {
  double ms = 0.0, fl = 0.0;
  int64_t n = 0;
  for (const ProfRec& r : g_prof) {
    if ((mode >= 0 && r.mode != mode) || (bm >= 0 && r.bm != bm) || (bn >= 0 && r.bn != bn) || (km >= 0 && r.km != km) || (nst >= 0 && r.nst != nst) ||
        (prec >= 0 && r.prec != prec))
      continue;
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms += t, fl += r.flops, ++n;
  }
  if (clear) {
    for (const ProfRec& r : g_prof) (void)hipEventDestroy(r.a), (void)hipEventDestroy(r.b);
    g_prof.clear();
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  return PM_OK;
}

// sum of the algorithmic bytes (operands once + output) of the recorded launches of one instantiation (negative key = any); for records other entry points file
// (the fused Winograd kernel, the bf16 tier's LDS-DMA kernels) the field is 0 unless they fill it
extern "C" int pm_profile_read_bytes(int mode, int bm, int bn, int km, int nst, int prec, double* total_bytes) {
  double b = 0.0;
  for (const ProfRec& r : g_prof) {
    if ((mode >= 0 && r.mode != mode) || (bm >= 0 && r.bm != bm) || (bn >= 0 && r.bn != bn) || (km >= 0 && r.km != km) || (nst >= 0 && r.nst != nst) ||
        (prec >= 0 && r.prec != prec))
      continue;
    b += r.bytes;
  }
  if (total_bytes) *total_bytes = b;
  return PM_OK;
}

// One CSV line per recorded launch (tuning aid): mode,bm,bn,km,nst,prec,M,N,K,batch,ksplit,ms,gflop
extern "C" int pm_profile_dump(const char* path) {
  PM_REQUIRE(path, PM_EINVAL, "pm_profile_dump: null path");
  FILE* f = fopen(path, "w");
  PM_REQUIRE(f, PM_EINVAL, "pm_profile_dump: cannot open %s", path);
  fprintf(f, "mode,bm,bn,km,nst,prec,M,N,K,batch,ksplit,ms,gflop\n");
  for (const ProfRec& r : g_prof) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.5f,%.4f\n", r.mode, r.bm, r.bn, r.km, r.nst, r.prec, r.M, r.Nn, r.K, r.batch, r.ksplit, t, r.flops * 1e-9);
  }
  fclose(f);
  return PM_OK;
}

extern "C" size_t pm_conv_winograd_v_bytes(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p0) {
  if (!x || !y || !p0) return 0;
  const pm_conv_params tp = tier_params(p0, x, y);
  const pm_conv_params* p = &tp;
  const WinoPlan f = wino_plan(x, y->c, p), g = wino_plan(x, y->c, p, true);   // forward and weight gradient both on the route
  return (f.use && g.use && f.g.m == g.g.m) ? f.v_bytes : 0;
}

extern "C" size_t pm_conv_wxf_bytes(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p0) {
  if (!x || !y || !p0) return 0;
  const pm_conv_params tp = tier_params(p0, x, y);
  const pm_conv_params* p = &tp;
  const WinoPlan f = wino_plan(x, y->c, p);
  if (f.use) return f.u_bytes;
  if (p->prec == 2) {
    const Bf16Plan b = bf16_plan(x, y, p);
    if (b.use) return b.wb_bytes;
  }
  return 0;
}

// fp32 tier: every kept Winograd forward transform U = G g Gt (what pm_conv_fwd writes into pm_conv_params.wxf when wxf_valid == 0) rewritten in one launch per 48
// filters, right after the optimizer moved the weights. A job's transform size is recovered from its buffer size (pm_conv_wxf_bytes of the call that filled it:
// (m + 2)^2 x cout x round32(cin) floats, m = 4 or 2); anything else is refused.
extern "C" int pm_conv_wxf_refresh_f32(const pm_wxf_job* jobs, int n, void* stream) {
  PM_REQUIRE(n >= 0 && (jobs || n == 0), PM_EINVAL, "conv_wxf_refresh_f32: bad job table");
  std::vector<const float*> w(n);
  std::vector<float*> U(n);
  std::vector<int> cout(n), cin(n), kp(n), m(n);
  for (int i = 0; i < n; ++i) {
    const pm_wxf_job& q = jobs[i];
    PM_REQUIRE(q.w && q.wxf && q.cout > 0 && q.cin > 0 && q.kh == 3 && q.kw == 3 && q.dgrad == 0, PM_EINVAL,
               "conv_wxf_refresh_f32: job %d: forward transforms of 3x3 filters only", i);
    PM_REQUIRE(pm_aligned16(q.wxf), PM_EINVAL, "conv_wxf_refresh_f32: job %d: buffer must be 16-byte aligned", i);
    const int Kp = (q.cin + BK - 1) / BK * BK;
    const size_t u4 = pm_align_up((size_t)36 * q.cout * Kp * sizeof(float), 256), u2 = pm_align_up((size_t)16 * q.cout * Kp * sizeof(float), 256);
    PM_REQUIRE((size_t)q.wxf_bytes == u4 || (size_t)q.wxf_bytes == u2, PM_EINVAL, "conv_wxf_refresh_f32: job %d: %ld bytes is neither the F(4x4) nor the F(2x2) transform of %d x %d",
               i, (long)q.wxf_bytes, q.cout, q.cin);
    w[i] = q.w, U[i] = (float*)q.wxf, cout[i] = q.cout, cin[i] = q.cin, kp[i] = Kp, m[i] = (size_t)q.wxf_bytes == u4 ? 4 : 2;
  }
  return pm_wino_filter_xf_multi(w.data(), U.data(), cout.data(), cin.data(), kp.data(), m.data(), n, (hipStream_t)stream);
}

// the same for pm_conv_bwd_data: bytes of the rotated / transposed bf16 filter a stride-1 data gradient of the bf16 tier derives from w (0: none)
extern "C" size_t pm_conv_wxf_bytes_dgrad(const pm_tensor* dy, const pm_tensor* dx, const pm_conv_params* p0) {
  if (!dy || !dx || !p0) return 0;
  const pm_conv_params tp = tier_params(p0, dy, dx);
  if (!dgrad_bf16_ok(&tp)) return 0;
  const pm_conv_params q = dgrad_as_fwd(&tp);
  const Bf16Plan b = bf16_plan(dy, dx, &q);
  return b.use ? b.wb_bytes : 0;
}

// Can this forward call hand the BatchNorm statistics of its output out of its own epilogue? Only the unbatched direct GEMM with one K
// split and a 16-byte-aligned output takes the staged epilogue; everything else (Winograd route, split-K, the 19-class heads) answers 0.
static bool bn_partials_route(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p) {
  if (!x || !y || !p || check_common(x, y, p) != PM_OK) return false;
  static const int on = getenv("PM_BN_EPILOGUE") ? atoi(getenv("PM_BN_EPILOGUE")) : 1;
  static const int stage_ep = getenv("PM_STAGE_EP") ? atoi(getenv("PM_STAGE_EP")) : 1;
  if (!on || !stage_ep) return false;
  if (pm_is_bf16(y)) {      // bf16 tier: both bf16 kernels carry the statistics in their 8-column staged epilogue (one K split, whole 16-byte groups)
    if (!pm_vec8(y)) return false;
    const Bf16Plan b = bf16_plan(x, y, p);
    return b.use && (b.c16 ? (b.k16.ksplit == 1 && !b.k16.wide) : b.pl.ksplit == 1);
  }
  if (pm_is_bf16(x)) return false;
  if ((y->c & 3) || (y->pitch & 3) || !pm_aligned16(y->ptr)) return false;
  if (p->prec == 2) {
    const Bf16Plan b = bf16_plan(x, y, p);
    if (b.use) return b.pl.ksplit == 1;
  }
  if (wino_plan(x, y->c, p).use) return false;
  if (p->prec != 0) return false;                 // staged-fp32 bf16 form: no statistics instantiation
  long M, Nn, K;
  gemm_dims(MODE_FWD, x, y, p, M, Nn, K);
  const Plan pl = make_plan(MODE_FWD, M, Nn, K, false);
  return pl.ksplit == 1 && pl.bn >= 64;
}
extern "C" size_t pm_conv_bn_partials_bytes(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p) {
  if (!bn_partials_route(x, y, p)) return 0;
  return pm_align_up((size_t)pm_cdiv(pm_pixels(y), 32) * y->c * 2 * sizeof(float), 256);
}

extern "C" size_t pm_conv_workspace(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p0, int which) {
  const pm_conv_params tp = tier_params(p0, x, y);
  const pm_conv_params* p = &tp;
  if (which == MODE_DGRAD && p->stride == 2) {   // four parity classes: compact results + the largest split-K slab set
    {
      const S2Native s = s2_native_plan(y, x, p);      // (x = dx, y = dy in this query)
      if (s.ok) return s.total;
    }
    size_t slab = 0, tmp = 0;      // (+ the fp32 copy of a bf16 dy below: on the bf16 tier the stride-2 data gradient still gathers fp32 rows)
    for (int cls = 0; cls < 4; ++cls) {
      const S2Class c = s2_class(cls, x, p);
      tmp = std::max(tmp, (size_t)c.M * x->c * sizeof(float));
      if (c.M > 0 && c.nky * c.nkx > 0) slab = std::max(slab, make_plan(MODE_DGRAD, c.M, x->c, (long)c.nky * c.nkx * y->c, p->prec != 0).ws_bytes);
    }
    return pm_align_up(4 * pm_align_up(tmp, 256) + slab + 256, 256) + upcast_bytes(y);
  }
  if (which == MODE_FWD && p->prec == 2) {
    const Bf16Plan b = bf16_plan(x, y, p);
    if (b.use) return bf16_ws(b);
  }
  if (which == MODE_DGRAD && dgrad_bf16_ok(p)) {
    const pm_conv_params q = dgrad_as_fwd(p);
    const Bf16Plan b = bf16_plan(y, x, &q);
    if (b.use) return bf16_ws(b);
  }
  if (which == MODE_FWD || which == MODE_DGRAD) {
    const WinoPlan wp = which == MODE_FWD ? wino_plan(x, y->c, p) : wino_plan(y, x->c, p);
    if (wp.use) return wino_ws(wp);
  }
  const size_t bias_part = pm_align_up((size_t)pm_cdiv(pm_pixels(y), colsum_rows(pm_pixels(y), y->c)) * y->c * sizeof(float), 256);
  if (which == MODE_WGRAD && (pm_is_bf16(x) || pm_is_bf16(y))) {      // bf16 tier: split-K slabs + bias partials + the fp32 copies of the mixed-type call sites
    long M, Nn, K;
    gemm_dims(which, x, y, p, M, Nn, K);
    return pm_align_up(make_plan(which, M, Nn, K, true).ws_bytes, 256) + bias_part + upcast_bytes(x) + upcast_bytes(y);
  }
  if (which == MODE_WGRAD && p->prec == 2) {
    const Bf16WgradPlan b = bf16_wgrad_plan(x, y, p);
    if (b.use) return pm_align_up(bf16_wgrad_ws(b), 256) + bias_part;
  }
  if (which == MODE_WGRAD) {
    const WinoPlan wp = wino_plan(x, y->c, p, true);
    if (wp.use) return pm_align_up(wino_wgrad_ws(wp, wino_wgrad_plan(wp, y->c)), 256) + bias_part;
  }
  long M, Nn, K;
  gemm_dims(which, x, y, p, M, Nn, K);
  size_t b = make_plan(which, M, Nn, K, p->prec != 0).ws_bytes;
  if (which == MODE_WGRAD) b += bias_part;
  return pm_align_up(b, 256);
}

extern "C" int pm_conv_fwd(const pm_tensor* x, const float* w, const pm_tensor* y, const pm_conv_params* p0,
                           const pm_conv_epilogue* ep, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_common(x, y, p0)) return e;
  const pm_conv_params tp = tier_params(p0, x, y);      // a bf16 tensor puts the call on the bf16 tier whatever prec says
  const pm_conv_params* p = &tp;
  PM_REQUIRE(!ep || ep->struct_size == (int64_t)sizeof(pm_conv_epilogue), PM_EINVAL,
             "conv_fwd: pm_conv_epilogue.struct_size %ld != %zu -- caller built against another pinmem_hip.h (library ABI %d)", ep ? (long)ep->struct_size : 0l,
             sizeof(pm_conv_epilogue), PM_ABI_VERSION);
  if (ep && ep->bn_partials) {
    PM_REQUIRE(!ep->relu && !ep->residual, PM_EINVAL, "conv_fwd: bn_partials are the statistics of the convolution output (no residual / ReLU)");
    const size_t need = pm_conv_bn_partials_bytes(x, y, p);
    PM_REQUIRE(need != 0 && (size_t)ep->bn_partials_bytes >= need, PM_EINVAL, "conv_fwd: this call cannot emit bn_partials (ask pm_conv_bn_partials_bytes): %zu needed", need);
  }
  PM_REQUIRE(w && pm_aligned16(w), PM_EINVAL, "conv_fwd: weight null or unaligned");
  {
    pm_conv_epilogue e1 = {(int64_t)sizeof(pm_conv_epilogue), nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (ep) e1 = *ep;
    PM_REQUIRE((e1.scale == nullptr) == (e1.shift == nullptr), PM_EINVAL, "conv_fwd: scale and shift go together");
    const WinoPlan wp = wino_plan(x, y->c, p);
    if (wp.use) {
      PM_REQUIRE(ws && ws_bytes >= wino_ws(wp), PM_EWORKSPACE, "conv_fwd(winograd): workspace %zu < %zu", ws_bytes, wino_ws(wp));
      float* keep = (p->wino_v && (size_t)p->wino_v_bytes >= wp.v_bytes) ? (float*)p->wino_v : nullptr;
      float* uext = (p->wxf && (size_t)p->wxf_bytes >= wp.u_bytes) ? (float*)p->wxf : nullptr;
      return wino_conv(x, w, y->c, x->c, false, y, wp, e1, ws, (hipStream_t)stream, keep, uext, p->wxf_valid != 0);
    }
  }
  if (p->prec == 2) {
    const Bf16Plan b = bf16_plan(x, y, p);
    if (b.use) {
      pm_conv_epilogue e2 = {(int64_t)sizeof(pm_conv_epilogue), nullptr, nullptr, nullptr, nullptr, 0, 0};
      if (ep) e2 = *ep;
      PM_REQUIRE((e2.scale == nullptr) == (e2.shift == nullptr), PM_EINVAL, "conv_fwd: scale and shift go together");
      PM_REQUIRE(ws && ws_bytes >= bf16_ws(b), PM_EWORKSPACE, "conv_fwd(bf16): workspace %zu < %zu", ws_bytes, bf16_ws(b));
      char* wext = (p->wxf && (size_t)p->wxf_bytes >= b.wb_bytes) ? (char*)p->wxf : nullptr;
      return conv_bf16(x, w, y->c, x->c, false, y, p, b, e2, ws, (hipStream_t)stream, wext, p->wxf_valid != 0);
    }
  }
  // what is left gathers fp32 rows: fp32 convolutions, and on the bf16 tier the 4-channel stem (fp32 image in, bf16 out)
  PM_REQUIRE(pm_is_f32(x), PM_EUNSUPPORTED, "conv_fwd: a bf16 input needs at least 32 channels (got %d)", x->c);
  long M, Nn, K;
  gemm_dims(MODE_FWD, x, y, p, M, Nn, K);
  Plan pl = make_plan(MODE_FWD, M, Nn, K, p->prec != 0);
  PM_REQUIRE(pl.ws_bytes <= ws_bytes && (pl.ws_bytes == 0 || ws), PM_EWORKSPACE, "conv_fwd: workspace %zu < %zu", ws_bytes, pl.ws_bytes);
  ConvK k;
  fill_geom(k, x, y, p);
  k.io16 = pm_is_bf16(y) ? 1 : 0;
  k.A = (const float*)x->ptr, k.B = w;
  k.M = (int)M, k.Nn = (int)Nn, k.K = (int)K;
  k.a_bytes = (unsigned)(pm_pixels(x) * x->pitch * 4), k.b_bytes = (unsigned)((long)y->c * K * 4), k.kmode = x->c % BK == 0 ? 0 : (x->c >= BK ? 1 : 2);
  pm_conv_epilogue e0 = {(int64_t)sizeof(pm_conv_epilogue), nullptr, nullptr, nullptr, nullptr, 0, 0};
  if (ep) e0 = *ep;
  PM_REQUIRE((e0.scale == nullptr) == (e0.shift == nullptr), PM_EINVAL, "conv_fwd: scale and shift go together");
  hipStream_t st = (hipStream_t)stream;
  if (pl.ksplit > 1) {
    k.C = (float*)ws, k.c_pitch = Nn, k.c_split = M * Nn;
    if (int e = launch<MODE_FWD>(k, pl, st)) return e;
    return splitk_reduce((const float*)ws, pl.ksplit, M, Nn, (float*)y->ptr, (long)y->pitch, e0.bias, e0.scale, e0.shift, e0.residual,
                         (long)e0.residual_pitch, e0.relu, st, k.io16 != 0);
  }
  k.C = (float*)y->ptr, k.c_pitch = y->pitch, k.c_split = 0;
  k.bias = e0.bias, k.scale = e0.scale, k.shift = e0.shift, k.residual = e0.residual, k.res_pitch = e0.residual_pitch, k.relu = e0.relu;
  k.stats = e0.bn_partials;
  return launch<MODE_FWD>(k, pl, st);
}

extern "C" int pm_conv_bwd_data(const pm_tensor* dy0, const float* w, const pm_tensor* dx, const pm_conv_params* p0, const pm_tensor* add,
                                void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_common(dx, dy0, p0)) return e;
  const pm_conv_params tp = tier_params(p0, dy0, dx);
  const pm_conv_params* p = &tp;
  const pm_tensor* dy = dy0;
  pm_tensor dy32;
  PM_REQUIRE(!add || (add->ptr && pm_same_shape(add, dx) && add->dtype == dx->dtype), PM_EINVAL, "conv_bwd_data: `add` must match dx (shape and dtype)");
  PM_REQUIRE(!add || !pm_is_bf16(add) || pm_vec8(add), PM_EINVAL, "conv_bwd_data: bf16 `add` must be 16B aligned with pitch %% 8 == 0");
  const float* addp = add ? (const float*)add->ptr : nullptr;
  const long add_pitch = add ? add->pitch : 0;
  PM_REQUIRE(w && pm_aligned16(w), PM_EINVAL, "conv_bwd_data: weight null or unaligned");
  hipStream_t st0 = (hipStream_t)stream;
  if (p->stride == 2) {
    // Only taps with (iy + pad - ky*dil) even reach an output pixel: split the input pixels into their four parity classes,
    // run a dense dgrad over each class with its matching tap subset (1/4 of the MFMA work of the masked formulation),
    // then interleave the compact results into dx.
    const size_t need = pm_conv_workspace(dx, dy, p, MODE_DGRAD);
    PM_REQUIRE(ws && ws_bytes >= need, PM_EWORKSPACE, "conv_bwd_data(stride 2): workspace %zu < %zu", ws_bytes, need);
    PM_REQUIRE(dx->c % 4 == 0, PM_EUNSUPPORTED, "conv_bwd_data(stride 2): Cin %% 4 != 0");
    {
      S2Native s = s2_native_plan(dy, dx, p);
      if (s.ok) return dgrad_s2_bf16(dy, w, dx, p, add, s, ws, st0);
    }
    if (pm_is_bf16(dy)) {      // bf16 tier, shapes the native form does not take: the parity-class gather reads fp32 rows -- dy (a quarter of dx's pixels) is widened once, at the end of the workspace
      if (int e = upcast(dy0, (char*)ws + need - upcast_bytes(dy0), &dy32, st0)) return e;
      dy = &dy32;
    }
    size_t tmp_bytes = 0;
    for (int cls = 0; cls < 4; ++cls) tmp_bytes = std::max(tmp_bytes, (size_t)s2_class(cls, dx, p).M * dx->c * sizeof(float));
    tmp_bytes = pm_align_up(tmp_bytes, 256);
    float* tmp = (float*)ws;
    float* slab = (float*)((char*)ws + 4 * tmp_bytes);
    unsigned char valid[4];
    for (int cls = 0; cls < 4; ++cls) {
      const S2Class c = s2_class(cls, dx, p);
      valid[cls] = c.M > 0 && c.nky * c.nkx > 0;
      if (!valid[cls]) continue;
      const long Kc = (long)c.nky * c.nkx * dy->c;
      const Plan pl = make_plan(MODE_DGRAD, c.M, dx->c, Kc, p->prec != 0);
      ConvK k;
      fill_geom(k, dx, dy, p);
      k.A = (const float*)dy->ptr, k.B = w;
      k.M = (int)c.M, k.Nn = dx->c, k.K = (int)Kc;
      k.a_bytes = (unsigned)(pm_pixels(dy) * dy->pitch * 4), k.b_bytes = (unsigned)((long)dy->c * p->kh * p->kw * dx->c * 4);
      k.kmode = dy->c >= BK ? K_MID : K_SMALL;
      k.T_eff = c.nky * c.nkx, k.tk_w = c.nkx, k.ky0 = c.ky0, k.kx0 = c.kx0, k.ksy = c.ksy, k.ksx = c.ksx;
      k.sub = 1, k.sub_cy = c.cy, k.sub_cx = c.cx, k.Hc = c.Hc, k.Wc = c.Wc;
      float* out = (float*)((char*)tmp + cls * tmp_bytes);
      if (pl.ksplit > 1) {
        k.C = slab, k.c_pitch = dx->c, k.c_split = c.M * dx->c;
        if (int e = launch<MODE_DGRAD>(k, pl, st0)) return e;
        if (int e = splitk_reduce(slab, pl.ksplit, c.M, dx->c, out, (long)dx->c, nullptr, nullptr, nullptr, nullptr, 0l, 0, st0)) return e;
      } else {
        k.C = out, k.c_pitch = dx->c, k.c_split = 0;
        if (int e = launch<MODE_DGRAD>(k, pl, st0)) return e;
      }
    }
    const int valid_mask = valid[0] | (valid[1] << 1) | (valid[2] << 2) | (valid[3] << 3);
    const long total = pm_pixels(dx) * (dx->c / 4);
    if (pm_is_bf16(dx))
      hipLaunchKernelGGL(dgrad_s2_interleave_kernel<true>, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st0, (const float*)tmp,
                         (long)(tmp_bytes / sizeof(float)), valid_mask, (float*)dx->ptr, (long)dx->pitch, dx->n, dx->h, dx->w, dx->c, addp, add_pitch);
    else
      hipLaunchKernelGGL(dgrad_s2_interleave_kernel<false>, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, st0, (const float*)tmp,
                         (long)(tmp_bytes / sizeof(float)), valid_mask, (float*)dx->ptr, (long)dx->pitch, dx->n, dx->h, dx->w, dx->c, addp, add_pitch);
    return pm_check_launch("dgrad_s2_interleave");
  }
  if (dgrad_bf16_ok(p)) {      // data gradient of a stride-1 convolution = forward convolution of dy with the rotated / transposed filter
    const pm_conv_params q = dgrad_as_fwd(p);
    const Bf16Plan b = bf16_plan(dy, dx, &q);
    if (b.use) {
      PM_REQUIRE(ws && ws_bytes >= bf16_ws(b), PM_EWORKSPACE, "conv_bwd_data(bf16): workspace %zu < %zu", ws_bytes, bf16_ws(b));
      const pm_conv_epilogue e1 = {(int64_t)sizeof(pm_conv_epilogue), nullptr, nullptr, nullptr, addp, add_pitch, 0};
      char* wext = (p->wxf && (size_t)p->wxf_bytes >= b.wb_bytes) ? (char*)p->wxf : nullptr;      // the rotated bf16 filter kept by the caller (pm_conv_wxf_bytes_dgrad)
      return conv_bf16(dy, w, dy->c, dx->c, true, dx, &q, b, e1, ws, st0, wext, p->wxf_valid != 0);
    }
  }
  {
    const WinoPlan wp = wino_plan(dy, dx->c, p);
    if (wp.use) {
      PM_REQUIRE(ws && ws_bytes >= wino_ws(wp), PM_EWORKSPACE, "conv_bwd_data(winograd): workspace %zu < %zu", ws_bytes, wino_ws(wp));
      const pm_conv_epilogue e1 = {(int64_t)sizeof(pm_conv_epilogue), nullptr, nullptr, nullptr, addp, add_pitch, 0};
      return wino_conv(dy, w, dy->c, dx->c, true, dx, wp, e1, ws, st0);
    }
  }
  PM_REQUIRE(pm_is_f32(dy), PM_EUNSUPPORTED, "conv_bwd_data: a bf16 dy needs at least 32 channels and a stride-1 geometry the forward form covers");
  long M, Nn, K;
  gemm_dims(MODE_DGRAD, dx, dy, p, M, Nn, K);
  Plan pl = make_plan(MODE_DGRAD, M, Nn, K, p->prec != 0);
  PM_REQUIRE(pl.ws_bytes <= ws_bytes && (pl.ws_bytes == 0 || ws), PM_EWORKSPACE, "conv_bwd_data: workspace %zu < %zu", ws_bytes, pl.ws_bytes);
  ConvK k;
  fill_geom(k, dx, dy, p);
  k.io16 = pm_is_bf16(dx) ? 1 : 0;
  k.A = (const float*)dy->ptr, k.B = w;
  k.M = (int)M, k.Nn = (int)Nn, k.K = (int)K;
  k.a_bytes = (unsigned)(pm_pixels(dy) * dy->pitch * 4), k.b_bytes = (unsigned)((long)dy->c * p->kh * p->kw * dx->c * 4), k.kmode = (dy->c % BK == 0 && p->stride == 1) ? 0 : (dy->c >= BK ? 1 : 2);
  hipStream_t st = (hipStream_t)stream;
  if (pl.ksplit > 1) {
    k.C = (float*)ws, k.c_pitch = Nn, k.c_split = M * Nn;
    if (int e = launch<MODE_DGRAD>(k, pl, st)) return e;
    return splitk_reduce((const float*)ws, pl.ksplit, M, Nn, (float*)dx->ptr, (long)dx->pitch, nullptr, nullptr, nullptr, addp, add_pitch, 0, st, k.io16 != 0);
  }
  k.C = (float*)dx->ptr, k.c_pitch = dx->pitch, k.c_split = 0;
  k.residual = addp, k.res_pitch = add_pitch;
  return launch<MODE_DGRAD>(k, pl, st);
}

extern "C" int pm_conv_bwd_weight(const pm_tensor* x0, const pm_tensor* dy0, float* dw, float* dbias, const pm_conv_params* p0, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (int e = check_common(x0, dy0, p0)) return e;
  const pm_conv_params tp = tier_params(p0, x0, dy0);
  const pm_conv_params* p = &tp;
  PM_REQUIRE(dw && pm_aligned16(dw), PM_EINVAL, "conv_bwd_weight: dw null or unaligned");
  hipStream_t st = (hipStream_t)stream;
  const size_t need = pm_conv_workspace(x0, dy0, p, MODE_WGRAD);
  PM_REQUIRE(need <= ws_bytes && ws, PM_EWORKSPACE, "conv_bwd_weight: workspace %zu < %zu", ws_bytes, need);
  const pm_tensor *x = x0, *dy = dy0;
  pm_tensor x32, dy32;
  // bf16 tier. Both operands bf16: the gather moves 16 bytes = eight channels per lane straight into the bf16 LDS tiles of the transpose-read form (PREC 4).
  // Mixed types (fp32 image x bf16 dy: the stem; bf16 x x fp32 dy: the 19-class heads) and a bias gradient over a bf16 dy: the bf16 side is widened to a
  // dense fp32 copy at the end of the workspace and the call proceeds as the fp32-gather form (PREC 3) -- small tensors, once per step.
  const bool tier = pm_is_bf16(x0) || pm_is_bf16(dy0);
  const bool native16 = pm_is_bf16(x0) && pm_is_bf16(dy0) && dy0->c >= 32;
  if (tier) {
    char* tail = (char*)ws + need;
    if (!native16 && pm_is_bf16(x0)) {
      tail -= upcast_bytes(x0);
      if (int e = upcast(x0, tail, &x32, st)) return e;
      x = &x32;
    }
    if (pm_is_bf16(dy0) && (!native16 || dbias)) {
      tail -= upcast_bytes(dy0);
      if (int e = upcast(dy0, tail, &dy32, st)) return e;
      if (!native16) dy = &dy32;
    }
  }
  long M, Nn, K;
  gemm_dims(MODE_WGRAD, x, dy, p, M, Nn, K);
  Plan pl = make_plan(MODE_WGRAD, M, Nn, K, p->prec != 0);
  const Bf16WgradPlan bw = tier ? Bf16WgradPlan{} : bf16_wgrad_plan(x, dy, p);
  if (bw.use) {
    if (int e = conv_wgrad_bf16(x, dy, dw, p, bw, ws, st)) return e;
    pl.ws_bytes = bf16_wgrad_ws(bw);       // the bias partials follow the bf16 buffers
  }
  const WinoPlan wp = wino_plan(x, dy->c, p, true);
  if (wp.use) {
    const WinoWgradPlan q = wino_wgrad_plan(wp, dy->c);
    float* kept = (p->wino_v && (size_t)p->wino_v_bytes >= wp.v_bytes) ? (float*)p->wino_v : nullptr;
    if (int e = wino_wgrad(x, dy, dw, wp, q, ws, st, kept)) return e;
    pl.ws_bytes = wino_wgrad_ws(wp, q);     // the bias partials follow the Winograd buffers
  }
  ConvK k;
  fill_geom(k, x, dy, p);
  // configs[2]: the weight gradient gathers the fp32 rows as before, rounds them to bf16 on the way into LDS and reads its fragments through the
  // transpose read (PREC 3): no transposed / per-tap copies of x in HBM. PM_BF16_WGRAD_TR=0 keeps the staged-fp32 form (A/B runs).
  static const int tr_on = getenv("PM_BF16_WGRAD_TR") ? atoi(getenv("PM_BF16_WGRAD_TR")) : 1;
  if (p->prec == 2 && tr_on) k.prec = 3;
  if (native16) {
    PM_REQUIRE(x->c % 8 == 0 && dy->c % 8 == 0 && pl.bn >= 64, PM_EUNSUPPORTED, "conv_bwd_weight(bf16): channels %% 8 != 0 or a tile narrower than 64");
    k.prec = 4;
    // the PREC 4 kernel steps 64 pixels at a time: the K range of a split must be a multiple of that (the slab count can only shrink, the workspace was sized for more)
    pl.kper = (pl.kper + 63) / 64 * 64;
    pl.ksplit = (int)((K + pl.kper - 1) / pl.kper);
  }
  const int esz = native16 ? 2 : 4;
  k.A = (const float*)dy->ptr, k.B = (const float*)x->ptr;
  k.M = (int)M, k.Nn = (int)Nn, k.K = (int)K;
  k.a_bytes = (unsigned)(pm_pixels(dy) * dy->pitch * esz), k.b_bytes = (unsigned)(pm_pixels(x) * x->pitch * esz), k.kmode = dy->w >= (native16 ? 2 * BK : BK) ? 1 : 2;
  // round 5, second session: both operands bf16 and Cin a multiple of the 128-channel block -> the LDS-DMA persistent ring of wgrad16.hip. Its units are
  // (256 x 128 tile, pixel range): the split is re-planned for one block per CU within the slab count the workspace was sized for.
  pm_wgrad16 w16{};
  bool use16 = false;
  if (native16 && !wp.use && !bw.use) {
    w16.X = (const pm_bf16*)x->ptr, w16.DY = (const pm_bf16*)dy->ptr;
    w16.N = x->n, w16.H = x->h, w16.W = x->w, w16.Ho = dy->h, w16.Wo = dy->w;
    w16.x_pitch = x->pitch, w16.dy_pitch = dy->pitch, w16.Cin = x->c, w16.Cout = dy->c;
    w16.kh = p->kh, w16.kw = p->kw, w16.stride = p->stride, w16.pad = p->pad, w16.dil = p->dil;
    w16.M = (int)M, w16.Nn = (int)Nn, w16.P = (int)K, w16.kper = 64, w16.c_split = M * Nn;
    w16.C = dw;
    use16 = K < (1l << 30) && pm_wgrad16_plan(&w16);
    if (use16) {
      const long tiles = (long)w16.tiles_m * w16.tiles_n, steps = (K + 63) / 64;
      int best_ks = 1;
      double best = 1e30;
      for (int ks = 1; ks <= pl.ksplit; ++ks) {
        const long per = (steps + ks - 1) / ks, kse = (steps + per - 1) / per;
        const long rounds = (tiles * kse + 255) / 256;
        // a unit costs its K-steps + ~6 steps of epilogue / hand-over; slabs: written and read back by the reduce (bytes / ~4 TB/s in K-steps of ~1 us)
        const double cost = (double)rounds * (per + 6) + (kse > 1 ? (double)kse * M * Nn * 8.0 / 4e12 / 1.0e-6 : 0.0);
        if (cost < best) best = cost, best_ks = (int)kse;
      }
      const long per = (steps + best_ks - 1) / best_ks;
      w16.kper = (int)per * 64, w16.ksplit = (int)((steps + per - 1) / per);
    }
  }
  if (use16) {
    ProfRec rec{};
    if (g_prof_on) {
      (void)hipEventCreate(&rec.a), (void)hipEventCreate(&rec.b);
      rec.mode = MODE_WGRAD, rec.bm = w16.bm, rec.bn = w16.bn, rec.km = 2, rec.prec = 4, rec.nst = 3, rec.M = (int)M, rec.Nn = (int)Nn, rec.K = (int)K, rec.batch = 1, rec.ksplit = w16.ksplit,
      rec.flops = 2.0 * (double)M * (double)Nn * (double)K;
      (void)hipEventRecord(rec.a, st);
    }
    if (w16.ksplit > 1) w16.C = (float*)ws;
    const int e = pm_wgrad16_launch(&w16, st);
    if (g_prof_on) {
      (void)hipEventRecord(rec.b, st);
      g_prof.push_back(rec);
    }
    if (e) return e;
    if (w16.ksplit > 1)
      if (int e2 = splitk_reduce((const float*)ws, w16.ksplit, M, Nn, dw, (long)Nn, nullptr, nullptr, nullptr, nullptr, 0l, 0, st)) return e2;
  } else if (wp.use || bw.use) {
  } else if (pl.ksplit > 1) {
    k.C = (float*)ws, k.c_pitch = Nn, k.c_split = M * Nn;
    if (int e = launch<MODE_WGRAD>(k, pl, st)) return e;
    if (int e = splitk_reduce((const float*)ws, pl.ksplit, M, Nn, dw, (long)Nn, nullptr, nullptr, nullptr, nullptr, 0l, 0, st)) return e;
  } else {
    k.C = dw, k.c_pitch = Nn, k.c_split = 0;
    if (int e = launch<MODE_WGRAD>(k, pl, st)) return e;
  }
  if (dbias) {
    const pm_tensor* dyb = pm_is_bf16(dy0) ? &dy32 : dy0;      // the bias gradient sums fp32 rows
    const long P = pm_pixels(dyb);
    const int rpb = colsum_rows(P, dyb->c), nb = pm_cdiv(P, rpb);
    float* part = (float*)((char*)ws + pm_align_up(pl.ws_bytes, 256));
    if (dyb->pitch == ((dyb->c + 3) & ~3) && dyb->pitch <= 64 && pm_aligned16(dyb->ptr))   // the tensor's own (pad-to-4) rows, not a channel slice
      hipLaunchKernelGGL(colsum_partial_narrow_kernel, dim3(nb), dim3(256), 0, st, (const float*)dyb->ptr, (int)(dyb->pitch / 4), P, dyb->c, rpb, part);
    else
      hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb, pm_cdiv(dyb->c, 64)), dim3(256), 0, st, (const float*)dyb->ptr, (long)dyb->pitch, P, dyb->c, rpb, part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(pm_cdiv(dyb->c, 64)), dim3(1024), 0, st, (const float*)part, nb, dyb->c, dbias);
    return pm_check_launch("conv_bias_grad");
  }
  return PM_OK;
}
