// BASELINE configs[2], round 4: the bf16 implicit-GEMM convolution built around LDS-DMA (the core of tools/micro/bf16_glds_lab.hip moved into the library and
// given the convolution gather). Forward and stride-1 data gradient (= forward convolution of dy with the rotated filter) of nn.Conv2d on bf16 activations:
//   C[M][N] = sum_k A[m][k] B[n][k],  M = output pixels, N = output channels, k = (tap, channel), channels padded to a multiple of 64 per tap
//   A = bf16 NHWC activations gathered per (row, tap): one K-step = one tap x 64 channels = 128 bytes of one pixel row
//   B = bf16 weights [N][taps][Cp] (pm_bf16_cast_weights), k-contiguous
// Staging: LDS-DMA (buffer_load_dwordx4 ... lds), 16 bytes per lane, straight from global memory into LDS -- no staging registers, no ds_write pass. The LDS image
// is lane-linear (row = 8 lanes x 16 B), so the XOR swizzle that makes the ds_read_b128 fragment reads conflict-free is applied on the SOURCE chunk a lane fetches
// (and again on the read). Both operands are addressed through buffer descriptors with 32-bit per-lane offsets: padding taps, rows beyond M and columns beyond N
// present an out-of-range offset and the fetch delivers zeros (the first form of the kernel used global_load_lds with 64-bit pointers and a zero page: 2.4 x the
// vector instructions per K-step, 566 vs 684 TF on the ASPP 3x3).
// 256 threads = 4 waves x (BM / WM) x (BN / WN) of v_mfma_f32_32x32x16_bf16 tiles, fp32 accumulation; NST LDS stages (2: the loads of K-step k + 1 fly under the
// MFMAs of step k; 1 for the single-step reductions of the 64-channel 1x1 convolutions, where four blocks share a CU instead of two).
// Epilogue staged through LDS: whole 16-byte row segments of eight bf16 (round to nearest even) with the fused bias / folded BatchNorm / residual / ReLU, or
// fp32 rows for the class logits and the split-K slabs.
// Replaces nn.Conv2d forward / input gradient of Resnet.py:145-150,195,453-457; deepv3plus.py:72-81,398-424; memory.py:75,104 on the bf16 tier.
#include <stdlib.h>
#include <algorithm>

#include "pm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BKB = 128;      // bytes per row and K-step (64 bf16)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// one 16-byte LDS-DMA fetch through a buffer descriptor: lane l of the wave lands at dst + 16 l (dst wave-uniform -> m0), out-of-range offsets deliver zeros.
// (A __device__ function of its own: with the builtin written into the kernel template the host pass silently drops the kernel's stub -- ROCm 7.2.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int WM, int WN, int NST, bool STATS>
__global__ __launch_bounds__(256, 2) void conv16_kernel(const pm_conv16 a) {
  constexpr int A_IT = BM / 32, B_IT = BN / 32;           // 16-byte fetches per lane and K-step: a 256-thread sweep covers 32 rows x 8 chunks
  constexpr int A_BYTES = BM * BKB, STAGE = (BM + BN) * BKB;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "bad tile config");
  extern __shared__ __align__(16) char lds[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, a.tiles_m * a.tiles_n);
  const int m0 = (lid / a.tiles_n) * BM, n0 = (lid % a.tiles_n) * BN;
  const int z = blockIdx.z;
  const int kt0 = z * a.ksteps_per, nk = min(a.ksteps - kt0, a.ksteps_per);      // this block's K-steps [kt0, kt0 + nk)
  const long pitchb = a.a_pitch * 2;

  // ---- per-lane staging constants: this lane's rows of the A and B tiles and the (swizzled) 16-byte chunk it fetches of each ----------------------------
  // Addressing (round 4, second form): both operands are fetched through BUFFER descriptors -- a 32-bit byte offset per lane, and a lane that must read zeros
  // (padding tap, row beyond M, column beyond N) simply presents an offset beyond the descriptor's range: the fetch returns 0 into LDS. The first form built a
  // 64-bit source pointer per fetch and selected a zero page for such lanes: 59 VALU + 56 SALU instructions per wave and K-step around 8 MFMAs.
  const __amdgpu_buffer_rsrc_t rA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.A), 0, (int)((long)a.N * a.H * a.W * pitchb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.B), 0, (int)((long)a.Nn * a.K * 2), 0x00020000);
  constexpr int OOB = 0x7fffffff;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // wave-uniform by construction: the LDS destination of a fetch lives on the scalar unit
  int aoff[A_IT], ay0[A_IT], ax0[A_IT];
  const bool pointwise = a.kh * a.kw == 1 && a.stride == 1 && a.pad == 0 && a.Ho == a.H && a.Wo == a.W;      // block-uniform
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int u = it * 256 + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
    const int m = m0 + row;
    if (m < a.M && pointwise) {      // 1x1 / stride 1 / no padding: output pixel m reads input pixel m -- no index decomposition (two integer divisions per row)
      ay0[it] = ax0[it] = 0;
      aoff[it] = (int)((long)m * pitchb) + ch * 16;
    } else if (m < a.M) {
      const int img = m / (a.Ho * a.Wo), rem = m - img * (a.Ho * a.Wo);
      const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
      ay0[it] = oy * a.stride - a.pad, ax0[it] = ox * a.stride - a.pad;
      aoff[it] = (int)(((long)(img * a.H + ay0[it]) * a.W + ax0[it]) * pitchb) + ch * 16;      // may be negative at the image border: only used with an in-range tap added
    } else {
      ay0[it] = ax0[it] = -(1 << 28);      // never inside the image: the row reads zeros
      aoff[it] = 0;
    }
  }
  int boff[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int u = it * 256 + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
    const int n = n0 + row;
    boff[it] = n < a.Nn ? n * a.K * 2 + ch * 16 : OOB;
  }
  // wave-uniform K-state: (tap, channel chunk) of the next K-step to stage
  const int cpc = a.Cp >> 6;      // 64-channel chunks per tap
  int s_tap = kt0 / cpc, s_ch = kt0 - s_tap * cpc, s_ky = s_tap / a.kw, s_kx = s_tap - s_ky * a.kw;
  s_tap = __builtin_amdgcn_readfirstlane(s_tap), s_ch = __builtin_amdgcn_readfirstlane(s_ch);
  s_ky = __builtin_amdgcn_readfirstlane(s_ky), s_kx = __builtin_amdgcn_readfirstlane(s_kx);
  int s_kb = kt0 * BKB;      // byte offset of the K-step inside a weight row (scalar offset of the B fetches)
  // Taps no row of this tile can see (round 5): a tile of BM output pixels covers image rows oy_a ... oy_b of ONE image (else: no skipping); a filter row ky whose
  // input rows oy * stride - pad + ky * dil lie outside the image for all of them contributes zeros only -- its K-steps are neither staged nor multiplied. On the
  // 48 x 48 maps the dilated ASPP branches (rates 6 / 12 / 18, deepv3plus.py:58-81) lose 8 / 17 / 25 % of their K-steps this way, nothing else is affected.
  unsigned ky_ok = ~0u;
  if (!pointwise && a.kh > 1) {
    const int hw = a.Ho * a.Wo, ml = min(m0 + BM, a.M) - 1;
    const int ia = m0 / hw, ib = ml / hw;
    if (ia == ib) {
      const int oy_a = (m0 - ia * hw) / a.Wo, oy_b = (ml - ib * hw) / a.Wo;
      ky_ok = 0;
      for (int ky = 0; ky < a.kh; ++ky)
        if (oy_b * a.stride - a.pad + ky * a.dil >= 0 && oy_a * a.stride - a.pad + ky * a.dil < a.H) ky_ok |= 1u << ky;
    }
  }
  if (ky_ok == (1u << a.kh) - 1u) ky_ok = ~0u;      // every filter row is visible (all interior tiles): no skipping, none of its per-tap bookkeeping, no K-step recount
  ky_ok = __builtin_amdgcn_readfirstlane(ky_ok);
  int nk_eff = nk;      // K-steps this block really runs: its range [kt0, kt0 + nk) minus the chunks of invisible taps
  if (ky_ok != ~0u) {
    nk_eff = 0;
    for (int kt = kt0; kt < kt0 + nk;) {
      const int tap = kt / cpc, run = min(kt0 + nk, (tap + 1) * cpc) - kt;
      if ((ky_ok >> (tap / a.kw)) & 1u) nk_eff += run;
      kt += run;
    }
    nk_eff = __builtin_amdgcn_readfirstlane(nk_eff);
  }
  auto skip_taps = [&]() {      // move the K-state past invisible taps (whole taps: the state sits at a tap start, or at the block's first K-step)
    while (s_ky < a.kh && !((ky_ok >> s_ky) & 1u)) {
      s_kb += (cpc - s_ch) * BKB;
      s_ch = 0;
      ++s_tap;
      if (++s_kx == a.kw) s_kx = 0, ++s_ky;
    }
  };
  if (ky_ok != ~0u) skip_taps();

  auto stage = [&](int buf) {
    char* la = lds + buf * STAGE;
    char* lb = la + A_BYTES;
    const int dy = s_ky * a.dil, dx = s_kx * a.dil;
    const int toff = (dy * a.W + dx) * (int)pitchb + s_ch * BKB;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const bool ok = ((unsigned)(ay0[it] + dy) < (unsigned)a.H) & ((unsigned)(ax0[it] + dx) < (unsigned)a.W);
      dma16(rA, la + (it * 256 + wave_u * 64) * 16, ok ? aoff[it] + toff : OOB, 0);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      dma16(rB, lb + (it * 256 + wave_u * 64) * 16, boff[it], s_kb);
    // advance the K-state (scalar unit)
    s_kb += BKB;
    if (++s_ch == cpc) {
      s_ch = 0;
      ++s_tap;
      if (++s_kx == a.kw) s_kx = 0, ++s_ky;
      if (ky_ok != ~0u) skip_taps();
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  auto compute = [&](int buf) {
    const char* la = lds + buf * STAGE;
    const char* lb = la + A_BYTES;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int c = kg * 2 + half;      // this lane-half's eight k of the 16-k block
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ra = wm * (BM / WM) + i * 32 + l31;
        fa[i] = *reinterpret_cast<const bf16x8*>(la + ra * BKB + ((c ^ ((ra >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int rb = wn * (BN / WN) + j * 32 + l31;
        fb[j] = *reinterpret_cast<const bf16x8*>(lb + rb * BKB + ((c ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  if (nk_eff > 0) {
    stage(0);
    __syncthreads();
    if constexpr (NST == 1) {
      for (int kt = 0; kt < nk_eff; ++kt) {
        compute(0);
        if (kt + 1 < nk_eff) {
          __syncthreads();      // every wave is done reading the single buffer
          stage(0);
          __syncthreads();
        }
      }
    } else {
      for (int kt = 0; kt < nk_eff; ++kt) {
        if (kt + 1 < nk_eff) stage((kt + 1) & 1);
        compute(kt & 1);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: a wave parks 32 rows of its tile in LDS (the stages are dead) and stores whole row segments ----------------------------------------
  constexpr int WC = BN / WN, LDC = WC + 4;
  __syncthreads();
  float* Ws = reinterpret_cast<float*>(lds) + wave * 32 * LDC;
  const bool slab = a.ksplit > 1;
  if (a.c_f32 || slab) {      // fp32 rows: split-K slabs (no epilogue) or the class logits (bias only; 19 columns at pitch 20: per-element stores)
    float* Cf = reinterpret_cast<float*>(a.C) + (slab ? (long)z * a.c_split : 0);
    const long cp = slab ? a.Nn : a.c_pitch;
    constexpr int LPR = WC / 4, RPI = 64 / LPR;
    const int rr0 = lane / LPR, cc = (lane % LPR) * 4;
    const int col = n0 + wn * WC + cc;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) Ws[((q & 3) + 8 * (q >> 2) + 4 * half) * LDC + n * 32 + l31] = acc[i][n][q];
#pragma unroll
      for (int r0 = 0; r0 < 32; r0 += RPI) {
        const int rr = r0 + rr0;
        const long row = m0 + wm * (BM / WM) + i * 32 + rr;
        const float4 v = *reinterpret_cast<const float4*>(Ws + rr * LDC + cc);
        if (row >= a.M) continue;
        const float e[4] = {v.x, v.y, v.z, v.w};
        if (((cp | a.Nn) & 3) == 0 && col + 4 <= a.Nn && !a.bias) PM_ST4(Cf + row * cp + col, v);
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (col + k < a.Nn) Cf[row * cp + col + k] = e[k] + ((a.bias && !slab) ? a.bias[col + k] : 0.f);
        }
      }
    }
    return;
  }
  {
    constexpr int LPR = WC / 8, RPI = 64 / LPR;
    static_assert(32 % RPI == 0, "row segments must tile the 32-row slab");
    const int rr0 = lane / LPR, cc = (lane % LPR) * 8;
    const int col = n0 + wn * WC + cc;
    const bool cok = col < a.Nn;      // Nn % 8 == 0: the group is all in or all out
    const bool aff = a.bias || a.scale, res = a.residual != nullptr, relu = a.relu != 0;
    float bi[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bi[e] = 0.f, sc[e] = 1.f, sh[e] = 0.f;
    if (aff && cok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (a.bias) bi[e] = a.bias[col + e];
        if (a.scale) sc[e] = a.scale[col + e], sh[e] = a.shift[col + e];
      }
    }
    pm_bf16* C16 = reinterpret_cast<pm_bf16*>(a.C);
    const pm_bf16* R16 = reinterpret_cast<const pm_bf16*>(a.residual);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) Ws[((q & 3) + 8 * (q >> 2) + 4 * half) * LDC + n * 32 + l31] = acc[i][n][q];
#pragma unroll
      for (int r0 = 0; r0 < 32; r0 += RPI) {
        const int rr = r0 + rr0;
        const long row = m0 + wm * (BM / WM) + i * 32 + rr;
        const float4 v0 = *reinterpret_cast<const float4*>(Ws + rr * LDC + cc), v1 = *reinterpret_cast<const float4*>(Ws + rr * LDC + cc + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (row < a.M && cok) {
          if (aff) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] + bi[e]) * sc[e] + sh[e];
          }
          if (res) {
            float q[8];
            pm_ld8(R16 + row * a.res_pitch + col, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += q[e];
          }
          if (relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          pm_st8(C16 + row * a.c_pitch + col, v);
        }
      }
      if constexpr (STATS)      // a separate instantiation: the kernels without statistics keep their register budget
        pm_slab_stats16<LDC, LPR, RPI>(Ws, rr0, cc, (long)m0 + wm * (BM / WM) + i * 32, a.M, a.Nn, col, cok, bi, sc, sh, a.stats);
    }
  }
}

template <int BM, int BN, int WM, int WN, int NST, bool STATS>
void launch_tile_s(const pm_conv16& k, dim3 grid, hipStream_t st) {
  constexpr size_t stage_bytes = (size_t)NST * (BM + BN) * BKB, ep_bytes = (size_t)4 * 32 * (BN / WN + 4) * sizeof(float);
  constexpr size_t smem = stage_bytes > ep_bytes ? stage_bytes : ep_bytes;
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv16_kernel<BM, BN, WM, WN, NST, STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL((conv16_kernel<BM, BN, WM, WN, NST, STATS>), grid, dim3(256), smem, st, k);
}
template <int BM, int BN, int WM, int WN, int NST>
void launch_tile(const pm_conv16& k, dim3 grid, hipStream_t st) {
  if (k.stats && k.ksplit == 1 && !k.c_f32) launch_tile_s<BM, BN, WM, WN, NST, true>(k, grid, st);
  else launch_tile_s<BM, BN, WM, WN, NST, false>(k, grid, st);
}

}  // namespace

// Tile / split choice. Rows in tiles of 128 (64 when that fills the 256 CUs better on the 48 x 48 maps), columns in tiles of 128 (64 for <= 64 output
// channels); split-K (fp32 slabs, fixed-order reduce by the caller) only where the output tiles alone leave most CUs idle AND the reduction is long.
// Round 5: the WIDE kernel (conv16w.hip: 256 x 128 / 128 x 256 tile, eight waves, ONE block per CU, three-stage ring) where a cost model of whole rounds of the 256 CUs
// says it wins: its K-step costs ~1.45 x a 64 x 128 step of the narrow kernel for 4 x the tile, but a partly filled last round costs a whole round, so the model picks
// the K-split that balances the tiles against the CUs. PM_C16W: 0 never, 1 by the model (default), 2 wherever the shape allows it (A/B runs, kernel tests).
namespace {
struct WidePick {
  bool ok;
  int bm, bn, ks;
  double cost;      // in units of one wide K-step on a full chip
};
WidePick wide_pick(const pm_conv16* k, int only_cfg = -1) {
  static const int force_ks = getenv("PM_C16W_KS") ? atoi(getenv("PM_C16W_KS")) : 0;
  static const double ovh = getenv("PM_C16W_OVH") ? atof(getenv("PM_C16W_OVH")) : 10.0;      // prologue + epilogue of a block, in K-steps
  WidePick best{false, 0, 0, 0, 1e30};
  const bool forced = pm_route.conv16_wide >= 2;      // kernel tests: every shape the kernel can express, ragged rows / columns and single K-steps included
  if (!forced && (k->Nn < 128 || k->M < 2048 || k->ksteps < 4)) return best;
  static const int env_cfg = getenv("PM_C16W_CFG") ? atoi(getenv("PM_C16W_CFG")) : -1;      // 0: 256 x 128, 1: 128 x 256, 2: 256 x 256 (two LDS stages)
  const int force_cfg = only_cfg >= 0 ? only_cfg : (pm_route.conv16_wide == 3 ? 2 : env_cfg);            // pm_set_conv16(7): the 256 x 256 tile on every shape (kernel tests)
  for (int cfg = 0; cfg < 3; ++cfg) {
    if (force_cfg >= 0 && cfg != force_cfg) continue;
    const int bm = cfg == 1 ? 128 : 256, bn = cfg == 0 ? 128 : 256;
    if (cfg == 2 && force_cfg != 2) continue;      // 256 x 256 only where asked for (the planner's rule below, PM_C16W_CFG, pm_set_conv16(7))
    const long tiles = (long)pm_cdiv(k->M, bm) * pm_cdiv(k->Nn, bn);
    const double fill = ((double)k->M * k->Nn) / ((double)tiles * bm * bn);      // padded rows / columns are wasted work
    for (int ks = 1; ks <= 16; ++ks) {
      if (force_ks && ks != force_ks) continue;
      if (ks > 1 && k->ksteps / ks < (forced ? 2 : 8)) break;
      const int per = pm_cdiv(k->ksteps, ks);
      const int kse = pm_cdiv(k->ksteps, per);
      const long blocks = tiles * kse;
      const long rounds = (blocks + 255) / 256;
      // split-K slabs: fp32 partial tiles written and read back by the reduce (bytes / ~4 TB/s, in wide K-steps of ~1.1 us measured on this kernel)
      const double slab = kse > 1 ? (double)kse * k->M * k->Nn * 8.0 / 4e12 / 1.1e-6 : 0.0;
      // a 256 x 256 step covers twice the area of a 256 x 128 one at ~1.3 x its rate (128 vs 85 FLOP per staged byte)
      const double cost = (double)rounds * (per + ovh) / fill * (cfg == 1 ? 1.03 : (cfg == 2 ? 1.5 : 1.0)) + slab;
      if (cost < best.cost) best = WidePick{true, bm, bn, kse, cost};
    }
  }
  return best;
}
}  // namespace
void pm_conv16_plan(pm_conv16* k) {
  static const int force_bm = getenv("PM_C16_BM") ? atoi(getenv("PM_C16_BM")) : 0;
  static const int force_ks = getenv("PM_C16_KS") ? atoi(getenv("PM_C16_KS")) : 0;
  k->wide = 0;
  k->bn = k->Nn > 64 ? 128 : 64;
  k->tiles_n = pm_cdiv(k->Nn, k->bn);
  const long t128 = (long)pm_cdiv(k->M, 128) * k->tiles_n, t64 = (long)pm_cdiv(k->M, 64) * k->tiles_n;
  // rounds of the 256 CUs x 2 resident blocks: take 64-row tiles when 128-row tiles would leave the last round under half full or not fill the chip once
  auto waste = [](long tiles) { const long r = (tiles + 511) / 512; return (double)(r * 512) / (double)tiles; };
  k->bm = (t128 < 512 || waste(t128) > 1.25 * waste(t64)) ? 64 : 128;
  if (force_bm) k->bm = force_bm;
  k->tiles_m = pm_cdiv(k->M, k->bm);
  const long tiles = (long)k->tiles_m * k->tiles_n;
  int ks = 1;
  if (tiles < 256 && k->ksteps >= 16) ks = (int)std::min<long>(std::min<long>(8, k->ksteps / 8), (512 + tiles - 1) / tiles);
  if (force_ks) ks = std::min(force_ks, k->ksteps);
  k->ksteps_per = pm_cdiv(k->ksteps, ks);
  k->ksplit = pm_cdiv(k->ksteps, k->ksteps_per);
  k->c_split = (long)k->M * k->Nn;
  // Round 5, measured and left OFF (PM_C16_FULLN=1 | 2 enables it for A/B runs): a 64 x 256 "full-N" tile for the HBM-bound 1x1 convolutions with a short reduction and a
  // wide output (64 -> 256 on the 192 x 192 maps, 128 -> 512 on 96 x 96, 256 -> 1024 on 48 x 48), so that the activation rows are fetched once per 256 output channels.
  // tools/conv16_probe.py, same box, alternated: 64 -> 256 253-260 -> 243-252 TF, 128 -> 512 288-297 -> 259-260, 256 -> 1024 360-369 -> 343-347: the second fetch of
  // the rows comes from L2 and was never the bound -- standalone the 64 -> 256 launch already moves its 189 MB in 37 us (5.1 TB/s: the output write), and the wider
  // tile only takes resident blocks away (80 KB of LDS stages per block instead of 48). The in-step figure of that shape (60 us) is not a tiling problem.
  static const int fulln = getenv("PM_C16_FULLN") ? atoi(getenv("PM_C16_FULLN")) : 0;
  if (!force_bm && k->Nn >= 256 && k->Nn % 256 == 0 && ((fulln == 1 && k->ksteps <= 4 && k->kh * k->kw == 1) || fulln >= 2) && ks == 1) {
    k->bm = 64, k->bn = 256;
    k->tiles_m = pm_cdiv(k->M, 64), k->tiles_n = k->Nn / 256;
    k->ksteps_per = k->ksteps, k->ksplit = 1;
  }
  if (pm_route.conv16_wide > 0) {
    // Where the wide tiles win (tools/conv16_probe.py on an MI355X, round 5, profiles/r05_conv16w_probe.txt), all with the 256 x 256 two-stage form (128 FLOP per staged
    // byte, full-N for the 256-channel outputs): the DEEP reductions -- the ASPP 3x3 2048 -> 256 on the 48 x 48 maps (288 K-steps: 690 TF narrow, 764 with skipped filter
    // rows, 733 on 256 x 128, **834-866**), the auxiliary head's 3x3 1024 -> 512 (660 -> 780) -- and the very wide outputs of a medium reduction (3x3 256 -> 2048
    // data-gradient form: 697 narrow, 725-769 on 256 x 128, **778-785**). Not the 72-K-step 3x3 512 -> 512 of layer4 (633 vs 650-700 narrow: 144 tiles), and on the
    // 192 x 192 maps every form sits at 915-958 TF (the decoder's 3x3s stay with the 128 x 128 register-staged kernel).
    const bool wins = k->ksteps >= 128 || (k->Nn >= 2048 && k->ksteps >= 32);
    // Second session: the PERSISTENT ring form (conv16w.hip conv16p_kernel: producer waves fetch ahead across tile boundaries). Same box, default routing -> persistent
    // (profiles/r05_conv16p_probe.txt): decoder 3x3 320 -> 256 @192 0.490 -> 0.423-0.438 ms (1030 TF), 256 -> 256 0.380 -> 0.348-0.355; ASPP 3x3 2048 -> 256 (256 x 256
    // form) 0.205 -> 0.191; its data-gradient form 256 -> 2048 0.225 -> 0.205. Not the 1024 -> 512 3x3 of the auxiliary head (0.222 on 256 x 256 vs 0.229-0.241), the
    // 72-K-step 3x3 512 -> 512 (0.126 narrow vs 0.138-0.145) or the short 1x1 reductions (the epilogue of a tile is not overlapped with anything).
    const bool ring_wins = pm_route.conv16_persistent && ((k->M >= 200000 && k->ksteps >= 32 && k->Nn >= 256) || (k->ksteps >= 128 && k->Nn <= 256) || (k->Nn >= 2048 && k->ksteps >= 32));
    const WidePick w = pm_route.conv16_wide >= 2 ? wide_pick(k) : (ring_wins ? wide_pick(k) : (wins ? wide_pick(k, 2) : WidePick{false, 0, 0, 0, 0}));
    if (w.ok) {
      k->wide = 1, k->bm = w.bm, k->bn = w.bn;
      k->tiles_m = pm_cdiv(k->M, k->bm), k->tiles_n = pm_cdiv(k->Nn, k->bn);
      k->ksteps_per = pm_cdiv(k->ksteps, w.ks);
      k->ksplit = pm_cdiv(k->ksteps, k->ksteps_per);
    }
  }
}
// Fraction of the K-steps the launch really runs: 1 minus the chunks of filter rows no row of a tile can see (the kernels' ky_ok rule, evaluated per row tile on the
// host). The in-library profile records EXECUTED FLOPs with it, so that a roofline fraction never counts multiplications that were skipped (round 5).
double pm_conv16_executed_fraction(const pm_conv16* k) {
  const bool pointwise = k->kh * k->kw == 1 && k->stride == 1 && k->pad == 0 && k->Ho == k->H && k->Wo == k->W;
  if (pointwise || k->kh <= 1) return 1.0;
  const long hw = (long)k->Ho * k->Wo;
  double run = 0.0;
  for (int tm = 0; tm < k->tiles_m; ++tm) {
    const long m0 = (long)tm * k->bm, ml = std::min<long>(m0 + k->bm, k->M) - 1;
    const long ia = m0 / hw, ib = ml / hw;
    int vis = k->kh;
    if (ia == ib) {
      const int oy_a = (int)((m0 - ia * hw) / k->Wo), oy_b = (int)((ml - ib * hw) / k->Wo);
      vis = 0;
      for (int ky = 0; ky < k->kh; ++ky)
        if (oy_b * k->stride - k->pad + ky * k->dil >= 0 && oy_a * k->stride - k->pad + ky * k->dil < k->H) ++vis;
    }
    run += (double)vis / k->kh;
  }
  return k->tiles_m > 0 ? run / k->tiles_m : 1.0;
}

size_t pm_conv16_slab_bytes(const pm_conv16* k) { return k->ksplit > 1 ? pm_align_up((size_t)k->ksplit * k->M * k->Nn * sizeof(float), 256) : 0; }

int pm_conv16_launch(const pm_conv16* k0, hipStream_t st) {
  if (k0->wide) {
    PM_REQUIRE(!k0->stats, PM_EUNSUPPORTED, "conv16w: no statistics epilogue (pm_conv_bn_partials_bytes answers 0 for this plan)");
    return pm_conv16w_launch(k0, st);
  }
  pm_conv16 k = *k0;
  dim3 grid(k.tiles_m * k.tiles_n, 1, k.ksplit);
  const bool one = k.ksteps_per == 1;      // a single K-step per block: one LDS stage, four blocks per CU
  if (k.bm == 64 && k.bn == 256) one ? launch_tile<64, 256, 1, 4, 1>(k, grid, st) : launch_tile<64, 256, 1, 4, 2>(k, grid, st);      // full-N tile of the short 1x1 reductions
  else if (k.bm == 128 && k.bn == 128) one ? launch_tile<128, 128, 2, 2, 1>(k, grid, st) : launch_tile<128, 128, 2, 2, 2>(k, grid, st);
  else if (k.bm == 64 && k.bn == 128) one ? launch_tile<64, 128, 2, 2, 1>(k, grid, st) : launch_tile<64, 128, 2, 2, 2>(k, grid, st);
  else if (k.bm == 128 && k.bn == 64) one ? launch_tile<128, 64, 2, 2, 1>(k, grid, st) : launch_tile<128, 64, 2, 2, 2>(k, grid, st);
  else if (k.bm == 64 && k.bn == 64) one ? launch_tile<64, 64, 2, 2, 1>(k, grid, st) : launch_tile<64, 64, 2, 2, 2>(k, grid, st);
  else {
    pm_set_error("conv16: no %d x %d tile", k.bm, k.bn);
    return PM_EUNSUPPORTED;
  }
  return pm_check_launch("conv16");
}
