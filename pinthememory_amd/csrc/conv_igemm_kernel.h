// The implicit-GEMM convolution kernel template of conv_igemm.hip, in a header so that the split-operand instantiations (PREC 5, conv_split.hip) compile as
// their own translation unit beside the fp32 / bf16 ones. See conv_igemm.hip for the description of the modes; the kernel parameter block is ConvK.
#pragma once
#include <stdint.h>
#include <type_traits>

#include "pm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// kernel parameter block (global namespace: conv_igemm.hip hands it to the PREC 5 launcher of conv_split.hip)
struct ConvK {
  const float* A;
  const float* B;
  float* C;
  int N, H, W, Cin;       // conv input tensor (x / dx)
  int Ho, Wo, Cout;       // conv output tensor (y / dy)
  long x_pitch, y_pitch;  // floats between pixels
  int kh, kw, stride, pad, dil, sshift;
  int M, Nn, K;           // GEMM extents
  int ksplit;             // gridDim.z
  int kper;               // K range per split (multiple of BK)
  long c_pitch;           // row pitch of C
  long c_split;           // floats between split-K slabs (ksplit > 1 -> C is the workspace)
  long a_bs, b_bs, c_bs;  // batched launch (gridDim.y > 1, the 16 Winograd points): floats between consecutive A / B / C operands
  int tiles_m, tiles_n;
  unsigned a_bytes, b_bytes;  // extents of the A / B buffers (buffer-descriptor range)
  int kmode;                  // K_FAST / K_MID / K_SMALL: how the gather's K-state advances (see the kernel)
  int prec;                   // 0 fp32 MFMA, 1 bf16 MFMA operands (fp32 storage and accumulation)
  // tap list actually iterated: rows ky0 + ksy*i (i < nky), cols kx0 + ksx*j (j < tk_w); T_eff = nky * tk_w. The full kernel
  // window for fwd / wgrad / stride-1 dgrad; the parity-matching subset for one input-pixel class of a stride-2 dgrad.
  int T_eff, tk_w, ky0, kx0, ksy, ksx;
  int sub, sub_cy, sub_cx, Hc, Wc;  // stride-2 dgrad: this launch covers input pixels (2*py + sub_cy, 2*px + sub_cx) only
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual;
  long res_pitch;
  int relu;
  float* stats;               // train-mode BN statistics of the output: (mean, M2) per 32-row slab and channel, or null
  int io16;                   // the bf16 tier: C and `residual` are bf16 tensors (pitches in elements); accumulation and the epilogue arithmetic stay fp32
  int stage_ep;               // 1: epilogue staged through LDS (16-byte row stores); 0: per-element stores (PM_STAGE_EP=0, A/B)
  int n_group;                // > 0: N tiles are walked in groups of n_group columns (tile order inside a group: M outer, N inner), so that one XCD's co-resident
                              // blocks cover a squarer patch of the output and the weight columns of the group stay in that XCD's L2 over the whole sweep of M
  int batch_xcd;              // 1: gridDim.y >= 8 batches are dealt to the XCDs whole (see the kernel); set by the launcher when the counts divide
  int spl_prio;               // PREC 5: 1 = waves in odd hardware wave slots run at s_setprio 2 (see the kernel), 0 = all equal
};

// conv_split.hip: launches the PREC 5 instantiation of the tile (bm x bn) / K-state mode of `k`; NST = 1 (one LDS stage) when nst1, else two
int pm_conv_split_launch(int mode, int bm, int bn, const ConvK& k, unsigned gx, unsigned gy, unsigned gz, size_t smem, bool nst1, hipStream_t st);
size_t pm_conv_split_stage_bytes(int mode, int bm, int bn);

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
constexpr int BK = 32;
constexpr int LDK = 36;  // k-contiguous LDS row stride (floats): 144 B rows -> conflict-free ds_read_b128


typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int OOB = 0x7fffffff;  // any byte offset beyond num_records makes a raw buffer load return zeros

// 16-byte load through a buffer descriptor: out-of-range lanes read 0 without a branch (padding taps, tile edges),
// and the address is a 32-bit byte offset (half the VALU work of 64-bit pointer arithmetic).
__device__ __forceinline__ float4 bload(__amdgpu_buffer_rsrc_t r, int off) {
  const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}

// XCD-aware bijective remap: consecutive logical ids (which share the A row-panel) land on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// KM selects how the running K-state of the gather advances by one K-step:
//   K_FAST  the channel extent is a multiple of BK (and stride 1 for dgrad): a K-slab never straddles a tap, so
//           (tap, channel) are wave-uniform -> they live in SGPRs and the per-row work is add / compare / select;
//   K_MID   channel extent (or Wo for wgrad) >= BK: at most one wrap per step, per-lane state, compare + select;
//   K_SMALL extent < BK (stem Cin = 4, Cout = 19, tiny test images): per-lane state, looping wrap.
//   K_PW    (round 6, the split path's instantiations) K_FAST on a pointwise problem -- 1x1 / stride 1 / no padding, and every Winograd point product: a row's validity does
//           not depend on the K-step, so it is folded into the row's base offset once and a gather costs ONE add per row (K_FAST: two compares, two adds, a select).
enum { K_FAST = 0, K_MID = 1, K_SMALL = 2, K_PW = 3 };

// Thread -> tile element mapping (256 threads, g = t & 7, r = t >> 3):
//   k-contiguous (KC) tiles [rows][LDK]: thread owns k-group g (4 floats) of rows r + 32*i   -> 1 K-state, static rows
//   m-contiguous (MC) tiles [BK][cols] : thread owns k-row r, column groups (g + 8*j) * 4      -> 1 K-state, static cols
// PREC 0: operands stay fp32 -> v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TF). PREC 1: the fp32 tiles staged in LDS are
// rounded to bf16 (RNE, v_cvt_pk_bf16_f32) as the fragments are read -> v_mfma_f32_32x32x16_bf16 with fp32 accumulation (2.5 PF):
// BASELINE configs[2]. Gathers, LDS layout and epilogue are shared; storage stays fp32.
// NST: LDS stages. 1 = single buffer for reductions of <= 64 K-steps (K <= 2048: every 1x1 convolution and Winograd GEMM of the
// flagship), 2 = double-buffered beyond that. A single stage halves the LDS footprint so that three blocks share a CU and cover each other's load / store phases.
// STATS: the staged epilogue also emits the BatchNorm statistics of the tile (a separate instantiation, so that the register budget and the
// occupancy of the kernel without statistics are untouched: 72 vs 74 VGPRs on the 64 x 64 forward tile = 7 vs 6 waves per SIMD).
template <int MODE, int BM, int BN, int WM, int WN, int KM, int PREC, int NST, bool STATS = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const ConvK a) {
  constexpr bool A_KC = (MODE != MODE_WGRAD);  // A tile stored [BM][LDK] (k contiguous) else [BK][BM]
  constexpr bool B_KC = (MODE == MODE_FWD);    // B tile stored [BN][LDK] else [BK][BN]
  constexpr bool FAST = (KM == K_FAST || KM == K_PW) && (MODE != MODE_WGRAD);
  constexpr bool PW = (KM == K_PW) && (MODE != MODE_WGRAD);
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  // PREC 3 (weight gradient of the bf16 tier): the pixel-major (m-contiguous) tiles are kept in LDS as bf16 [BK][BM + 32] -- rounded once, as the
  // gathered fp32 rows are stored -- and the MFMA fragments (8 consecutive k per lane) come out of them through the hardware transpose read
  // ds_read_b64_tr_b16. The 32-element pad puts the four k-rows one read touches on four disjoint bank quarters (row stride = 64 B mod 256 B).
  // PREC 4: the same with NATIVE bf16 operands (bf16 activations in HBM, BASELINE configs[2] round 4): a lane gathers 16 bytes = eight channels and stores them
  // to LDS as they are -- half the gather instructions, no conversion.
  constexpr bool TR = (PREC == 3 || PREC == 4);
  constexpr bool NAT16 = (PREC == 4);
  // PREC 5 (round 6): the fp32 tier on the bf16 matrix pipe. Every gathered fp32 element is split -- ONCE per block, as the gathered rows are stored to LDS -- into
  // three bf16 pieces by truncation: hi = x & 0xffff0000, r = x - hi (exact), mid = r & 0xffff0000, lo = r - mid (exact: at most eight significant bits are left),
  // so x == hi + mid + lo exactly. LDS holds three bf16 planes per operand tile; a K-step multiplies six cross products (hi hi, hi mid, mid hi, mid mid, hi lo,
  // lo hi; the dropped mid lo + lo mid + lo lo are <= 3 * 2^-24 of |a b|) on v_mfma_f32_32x32x16_bf16 into the same fp32 accumulators: 6 / 16 of the fp32 MFMA's
  // matrix-pipe cycles per K-step. Measured against fp64 (tools/micro/split_gemm.hip): the same error class as the fp32 fma chain of PREC 0.
  constexpr bool SPL = (PREC == 5);
  constexpr int LD5 = 52;      // floats per k-contiguous row: [hi | mid | lo] x 32 k x 2 B + 16 B -> 13 sixteen-byte granules per row: conflict-free ds_read_b128
  // K-step of this instantiation. PREC 4 takes 64 pixels per step (a thread gathers two pixel rows, r and r + 32): the bf16 MFMA phase of a 32-pixel step is 256
  // cycles per wave -- too short for one barrier pair and one round of address arithmetic; the forward kernels' steps are 64 deep as well (round 4: 465 -> see DESIGN)
  constexpr int KX = NAT16 ? 2 : 1, BKW = BK * KX;
  static_assert(!TR || MODE == MODE_WGRAD, "PREC 3 / 4 are the weight-gradient forms");
  constexpr int LDA_T = BM + 32, LDB_T = BN + 32;   // bf16 elements per k-row
  constexpr int PLANE_A = BK * LDA_T / 2, PLANE_B = BK * LDB_T / 2;   // PREC 5, m-contiguous operands: floats per bf16 plane [BK][m + 32]
  constexpr int A_FLOATS = SPL ? (A_KC ? BM * LD5 : 3 * PLANE_A) : (A_KC ? BM * LDK : (TR ? BKW * LDA_T / 2 : BK * BM));
  constexpr int B_FLOATS = SPL ? (B_KC ? BN * LD5 : 3 * PLANE_B) : (B_KC ? BN * LDK : (TR ? BKW * LDB_T / 2 : BK * BN));
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  constexpr int A_N = BM / 32, B_N = BN / 32;  // float4 per thread and tile
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BK == 32, "bad tile config");

  extern __shared__ __align__(16) float smem[];
  if constexpr (PREC == 5) {
    // Two blocks share a CU, i.e. two waves share each SIMD's VALU issue and matrix pipe. Equal waves that start together STAY together: both split (VALU) at the
    // same time, then both multiply (measured: VALU issue 44 % + matrix pipe 57 % of the kernel's cycles ~ 100 %, no overlap at all). A static priority for the wave
    // in the odd hardware slot lets it run ahead by one phase, after which the partner's VALU phase fills the leader's MFMA phase and vice versa.
    if (a.spl_prio && (__builtin_amdgcn_s_getreg(0x1804) & 1)) __builtin_amdgcn_s_setprio(2);      // hwreg(HW_REG_HW_ID, 0, 4): the wave's slot on its SIMD
  }

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int g = t & 7, r = t >> 3;

  const int ntile = a.tiles_m * a.tiles_n;
  // Batched / split-K launches (the 36 / 16 point GEMMs of a Winograd layer; the K slices of a weight gradient): whole (batch, K-slice) units go to one XCD (unit u -> XCD
  // u mod 8; a remainder of 1 / 2 / 4 units is cut into 8 / 4 / 2 runs of tiles), so that both operands of a unit pass through ONE L2 instead of the weight tile of a
  // point (the activation rows of a K slice) being fetched by every XCD that holds a few of its tiles. The linear workgroup id is dealt round-robin to the XCDs
  // (MI355X_MICROARCH.md "Workgroup dispatch"); a wrong guess costs locality only.
  int lid, by, z;
  if (a.batch_xcd) {
    const int L = blockIdx.x + ntile * (blockIdx.y + gridDim.y * blockIdx.z), xcd = L & 7, idx = L >> 3;
    const int units = (int)(gridDim.y * gridDim.z), whole = units >> 3, rem = units & 7;
    int u;
    if (idx < whole * ntile) {
      u = xcd + 8 * (idx / ntile), lid = idx % ntile;
    } else {
      const int share = 8 / rem, per = ntile / share;
      u = whole * 8 + xcd / share, lid = (xcd % share) * per + (idx - whole * ntile);
    }
    by = u % (int)gridDim.y, z = u / (int)gridDim.y;
  } else {
    lid = xcd_remap(blockIdx.x, ntile), by = blockIdx.y, z = blockIdx.z;
  }
  int tile_m, tile_n;
  if (a.n_group > 0 && a.tiles_n > a.n_group) {
    const int gsz = a.tiles_m * a.n_group, grp = lid / gsz, rem = lid - grp * gsz;
    const int gw = min(a.n_group, a.tiles_n - grp * a.n_group);      // the last group may be narrower
    tile_m = rem / gw, tile_n = grp * a.n_group + rem % gw;
  } else {
    tile_m = lid / a.tiles_n, tile_n = lid % a.tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int k_begin = z * a.kper;
  const int k_end = min(a.K, k_begin + a.kper);
  const int nk = (k_end - k_begin + BKW - 1) / BKW;
  const int T = a.T_eff;                 // taps iterated by this launch
  const int Treal = a.kh * a.kw;         // tap stride of the KRSC weight layout

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A + by * a.a_bs), 0, (int)a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B + by * a.b_bs), 0, (int)a.b_bytes, 0x00020000);
  const int xp4 = (int)a.x_pitch * (NAT16 ? 2 : 4), yp4 = (int)a.y_pitch * (NAT16 ? 2 : 4);      // bytes between pixels
  constexpr int A_NL = NAT16 ? (A_N + 1) / 2 : A_N, B_NL = NAT16 ? (B_N + 1) / 2 : B_N;            // 16-byte gathers per thread and tile
  constexpr int EPL = NAT16 ? 8 : 4;                                                               // elements per 16-byte gather

  // ---- gather state ---------------------------------------------------------------------------------------------
  // uniform K-state (FAST) / per-lane K-state (MID, SMALL) of the (tap, channel) decomposition of k
  int u_ch = 0, u_tap = 0, u_ky = 0, u_kx = 0;     // FAST: wave-uniform (SGPR)
  int a_ch = 0, a_tap = 0, a_ky = 0, a_kx = 0;     // MID/SMALL FWD+DGRAD A
  int b_co = 0, b_tap = 0;                         // MID/SMALL DGRAD B row (tap, co)
  int p_ox = 0, p_oy = 0, p_img = 0;               // WGRAD B: output pixel of this thread's k-row
  int q_ox = 0, q_oy = 0, q_img = 0;               // ... and of its second k-row (r + 32) when the K-step is 64 pixels (KX == 2)
  // static per-item constants
  int a_base[A_N], a_y0[A_N], a_x0[A_N];           // FWD/DGRAD A rows: byte offset of the window origin, window origin
  int a_col4[A_N];                                 // WGRAD A: byte offset of dy channel group j (OOB beyond Cout)
  int b_base[B_N];                                 // FWD: weight-row byte offset; DGRAD: ci group byte offset; WGRAD: tap+channel offset
  int b_dy[B_N], b_dx[B_N];                        // WGRAD B: tap displacement of column group j

  if constexpr (MODE == MODE_FWD || MODE == MODE_DGRAD) {
    const int rh = (MODE == MODE_FWD) ? a.Ho : (a.sub ? a.Hc : a.H), rw = (MODE == MODE_FWD) ? a.Wo : (a.sub ? a.Wc : a.W);
    const int cdim = (MODE == MODE_FWD) ? a.Cin : a.Cout;
    const bool pointwise = FAST && a.kh * a.kw == 1 && a.stride == 1 && a.pad == 0 && !a.sub;
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int m = m0 + r + 32 * i;
      if (PW) {          // the launcher vouches for a pointwise problem: rows beyond M carry an out-of-range base, nothing else is checked per K-step
        a_y0[i] = a_x0[i] = 0;
        a_base[i] = m < a.M ? m * (MODE == MODE_FWD ? xp4 : yp4) + g * 16 : OOB;
      } else if (pointwise) {   // 1x1 / stride 1 / no padding (and every Winograd GEMM): pixel m of the operand, no index decomposition
        a_y0[i] = a_x0[i] = m < a.M ? 0 : -(1 << 28);
        a_base[i] = m < a.M ? m * (MODE == MODE_FWD ? xp4 : yp4) + (FAST ? g * 16 : 0) : 0;
      } else if (m < a.M) {
        const int img = m / (rh * rw), rem = m - img * (rh * rw);
        const int py = rem / rw, px = rem - py * rw;
        if constexpr (MODE == MODE_FWD) {
          a_y0[i] = py * a.stride - a.pad;
          a_x0[i] = px * a.stride - a.pad;
          a_base[i] = (img * a.H * a.W + a_y0[i] * a.W + a_x0[i]) * xp4 + (FAST ? g * 16 : 0);
        } else {
          a_y0[i] = (a.sub ? 2 * py + a.sub_cy : py) + a.pad;
          a_x0[i] = (a.sub ? 2 * px + a.sub_cx : px) + a.pad;
          a_base[i] = FAST ? ((img * a.Ho + a_y0[i]) * a.Wo + a_x0[i]) * yp4 + g * 16 : img * a.Ho * a.Wo;
        }
      } else {
        a_base[i] = 0;
        a_y0[i] = -(1 << 28);
        a_x0[i] = -(1 << 28);
      }
    }
    if constexpr (FAST) {
      u_tap = __builtin_amdgcn_readfirstlane(k_begin / cdim);
      u_ch = k_begin - u_tap * cdim;
      u_ky = u_tap / a.kw;
      u_kx = u_tap - u_ky * a.kw;
    } else {
      const int k = k_begin + g * 4;
      a_tap = k / cdim;
      a_ch = k - a_tap * cdim;
      a_ky = a_tap / a.tk_w;
      a_kx = a_tap - a_ky * a.tk_w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < A_NL; ++j) {
      const int col = m0 + (g + 8 * j) * EPL;
      a_col4[j] = col < a.Cout ? col * (NAT16 ? 2 : 4) : OOB;
    }
  }
  if constexpr (MODE == MODE_FWD) {
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
      const int row = n0 + r + 32 * i;
      b_base[i] = row < a.Cout ? row * a.K * 4 + g * 16 : OOB;
    }
  } else if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
    for (int j = 0; j < B_N; ++j) {
      const int col = n0 + (g + 8 * j) * 4;
      b_base[j] = col < a.Cin ? col * 4 + (FAST ? r * Treal * a.Cin * 4 : 0) : OOB;
    }
    if constexpr (!FAST) {
      const int k = k_begin + r;
      b_tap = k / a.Cout;
      b_co = k - b_tap * a.Cout;
    }
  } else {
#pragma unroll
    for (int j = 0; j < B_NL; ++j) {
      const int n = n0 + (g + 8 * j) * EPL;
      const int tap = n < a.Nn ? n / a.Cin : 0;
      const int ky = tap / a.kw, kx = tap - ky * a.kw;
      b_dy[j] = ky * a.dil - a.pad;
      b_dx[j] = kx * a.dil - a.pad;
      b_base[j] = n < a.Nn ? (b_dy[j] * a.W + b_dx[j]) * xp4 + (n - tap * a.Cin) * (NAT16 ? 2 : 4) : OOB;
    }
    const int p = k_begin + r;
    p_img = p / (a.Ho * a.Wo);
    const int rem = p - p_img * (a.Ho * a.Wo);
    p_oy = rem / a.Wo;
    p_ox = rem - p_oy * a.Wo;
    if constexpr (KX == 2) {
      const int q = p + 32;
      q_img = q / (a.Ho * a.Wo);
      const int rq = q - q_img * (a.Ho * a.Wo);
      q_oy = rq / a.Wo;
      q_ox = rq - q_oy * a.Wo;
    }
  }

  float4 ra[A_N], rb[B_N];

  auto load_tiles = [&](int kt, int which = 0) {   // branch-free: every lane always issues its loads, invalid ones at offset OOB. which: 0 both operands, 1 A only, 2 B only
    const int kbase = k_begin + kt * BKW;
    const bool doA = which != 2, doB = which != 1;
    if constexpr (PW) {
      // wave-uniform offsets; a K-step beyond the reduction (the software pipeline's run-ahead) moves every offset out of the descriptor's range: no traffic
      const unsigned beyond = 0x80000000u;
      const bool tok = u_tap < T;
      if constexpr (MODE == MODE_FWD) {
        const unsigned toff = (unsigned)(u_ch * 4) + (tok ? 0u : beyond), kb4 = kbase < k_end ? (unsigned)(kbase * 4) : beyond;
        if (doA)
#pragma unroll
        for (int i = 0; i < A_N; ++i) ra[i] = bload(rA, (int)((unsigned)a_base[i] + toff));
        if (doB)
#pragma unroll
        for (int i = 0; i < B_N; ++i) rb[i] = bload(rB, (int)((unsigned)b_base[i] + kb4));
      } else {
        const unsigned toff = (unsigned)(u_ch * 4) + (tok ? 0u : beyond), uoff = tok ? (unsigned)((u_ch * Treal + u_tap) * a.Cin * 4) : beyond;
        if (doA)
#pragma unroll
        for (int i = 0; i < A_N; ++i) ra[i] = bload(rA, (int)((unsigned)a_base[i] + toff));
        if (doB)
#pragma unroll
        for (int j = 0; j < B_N; ++j) rb[j] = bload(rB, (int)((unsigned)b_base[j] + uoff));
      }
    } else if constexpr (MODE == MODE_FWD) {
      const int ky = FAST ? u_ky : a_ky, kx = FAST ? u_kx : a_kx, tap = FAST ? u_tap : a_tap, ch = FAST ? u_ch : a_ch;
      const int dy = ky * a.dil, dx = kx * a.dil;
      const int toff = (dy * a.W + dx) * xp4 + ch * 4;
      const bool tok = tap < T;
      if (doA)
#pragma unroll
      for (int i = 0; i < A_N; ++i) {
        const bool ok = ((unsigned)(a_y0[i] + dy) < (unsigned)a.H) & ((unsigned)(a_x0[i] + dx) < (unsigned)a.W) & tok;
        const int off = a_base[i] + toff;
        ra[i] = bload(rA, ok ? off : OOB);
      }
      const bool kok = FAST ? (kbase < k_end) : (kbase + g * 4 < k_end);
      if (doB)
#pragma unroll
      for (int i = 0; i < B_N; ++i) {
        const int off = b_base[i] + kbase * 4;   // b_base already carries this thread's k-group (g * 16 bytes)
        rb[i] = bload(rB, (kok & (b_base[i] != OOB)) ? off : OOB);
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      if constexpr (FAST) {   // stride 1, Cout % BK == 0: uniform (tap, co0)
        const int dy = u_ky * a.dil, dx = u_kx * a.dil;
        const int toff = (dy * a.Wo + dx) * yp4 - u_ch * 4;
        const bool tok = u_tap < T;
        if (doA)
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
          const bool ok = ((unsigned)(a_y0[i] - dy) < (unsigned)a.Ho) & ((unsigned)(a_x0[i] - dx) < (unsigned)a.Wo) & tok;
          const int off = a_base[i] - toff;
          ra[i] = bload(rA, ok ? off : OOB);
        }
        const int uoff = (u_ch * Treal + u_tap) * a.Cin * 4;
        if (doB)
#pragma unroll
        for (int j = 0; j < B_N; ++j) {
          const int off = b_base[j] + uoff;
          rb[j] = bload(rB, (tok & (b_base[j] != OOB)) ? off : OOB);
        }
      } else {
        const int dy = (a.ky0 + a.ksy * a_ky) * a.dil, dx = (a.kx0 + a.ksx * a_kx) * a.dil, smask = a.stride - 1;
        const bool kok = a_tap < T;
        if (doA)
#pragma unroll
        for (int i = 0; i < A_N; ++i) {
          const int ty = a_y0[i] - dy, tx = a_x0[i] - dx;
          const int oy = ty >> a.sshift, ox = tx >> a.sshift;
          const bool ok = kok & ((ty | tx) >= 0) & (((ty | tx) & smask) == 0) & (oy < a.Ho) & (ox < a.Wo);   // '&': no short-circuit branches
          const int off = (a_base[i] + oy * a.Wo + ox) * yp4 + a_ch * 4;   // computed unconditionally: keeps the loop body branch-free
          ra[i] = bload(rA, ok ? off : OOB);
        }
        const bool rok = (b_tap < T) & (kbase + r < k_end);
        const int b_tky = b_tap / a.tk_w, b_tkx = b_tap - b_tky * a.tk_w;
        const int roff = (b_co * Treal + (a.ky0 + a.ksy * b_tky) * a.kw + a.kx0 + a.ksx * b_tkx) * a.Cin * 4;
        if (doB)
#pragma unroll
        for (int j = 0; j < B_N; ++j) {
          const int off = roff + b_base[j];
          rb[j] = bload(rB, (rok & (b_base[j] != OOB)) ? off : OOB);
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < KX; ++kk) {      // KX == 2: the thread's second pixel row, 32 further on
        const int p = kbase + r + 32 * kk;
        const bool pok = p < k_end;
        const int poff = p * yp4;
        if (doA)
#pragma unroll
        for (int j = 0; j < A_NL; ++j) {
          const int off = poff + a_col4[j];
          ra[kk * A_NL + j] = bload(rA, (pok & (a_col4[j] != OOB)) ? off : OOB);
        }
        const int by = (kk ? q_oy : p_oy) * a.stride, bx = (kk ? q_ox : p_ox) * a.stride;
        const int rowbase = (((kk ? q_img : p_img) * a.H + by) * a.W + bx) * xp4;
        if (doB)
#pragma unroll
        for (int j = 0; j < B_NL; ++j) {
          const bool ok = pok & (b_base[j] != OOB) & ((unsigned)(by + b_dy[j]) < (unsigned)a.H) & ((unsigned)(bx + b_dx[j]) < (unsigned)a.W);
          const int off = rowbase + b_base[j];
          rb[kk * B_NL + j] = bload(rB, ok ? off : OOB);
        }
      }
    }
  };

  auto advance = [&]() {  // move the K-state one K-step (BK) forward
    if constexpr (MODE == MODE_FWD || MODE == MODE_DGRAD) {
      const int cdim = (MODE == MODE_FWD) ? a.Cin : a.Cout;
      if constexpr (FAST) {
        u_ch += BK;
        if (u_ch >= cdim) {            // uniform branch on SGPRs (scalar unit), never diverges
          u_ch = 0;
          ++u_tap;
          if (++u_kx == a.kw) u_kx = 0, ++u_ky;
        }
      } else {
        a_ch += BK;
        if constexpr (KM == K_SMALL) {
          while (a_ch >= cdim) {
            a_ch -= cdim;
            ++a_tap;
            if (++a_kx == a.tk_w) a_kx = 0, ++a_ky;
          }
        } else {
          const bool wrap = a_ch >= cdim;
          a_ch -= wrap ? cdim : 0;
          a_tap += wrap ? 1 : 0;
          const bool roww = wrap & (a_kx + 1 == a.tk_w);
          a_kx = roww ? 0 : a_kx + (wrap ? 1 : 0);
          a_ky += roww ? 1 : 0;
        }
        if constexpr (MODE == MODE_DGRAD) {
          b_co += BK;
          if constexpr (KM == K_SMALL) {
            while (b_co >= a.Cout) b_co -= a.Cout, ++b_tap;
          } else {
            const bool wrap = b_co >= a.Cout;
            b_co -= wrap ? a.Cout : 0;
            b_tap += wrap ? 1 : 0;
          }
        }
      }
    } else {
      auto step = [&](int& ox, int& oy, int& img) {
        ox += BKW;
        if constexpr (KM == K_SMALL) {
          while (ox >= a.Wo) {
            ox -= a.Wo;
            if (++oy == a.Ho) oy = 0, ++img;
          }
        } else {      // Wo >= BKW: at most one wrap per step
          const bool wrap = ox >= a.Wo;
          ox -= wrap ? a.Wo : 0;
          const bool imgw = wrap & (oy + 1 == a.Ho);
          oy = imgw ? 0 : oy + (wrap ? 1 : 0);
          img += imgw ? 1 : 0;
        }
      };
      step(p_ox, p_oy, p_img);
      if constexpr (KX == 2) step(q_ox, q_oy, q_img);
    }
  };

  auto pack4 = [](const float4& v) {   // four fp32 -> four bf16 (RNE), 8 bytes
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 o;
    o[0] = (__bf16)v.x, o[1] = (__bf16)v.y, o[2] = (__bf16)v.z, o[3] = (__bf16)v.w;
    return __builtin_bit_cast(float2, o);
  };
  auto split4 = [](const float4& v, float2& h, float2& m, float2& l) {   // four fp32 -> 3 x four bf16 (truncation split, exact): 16 VALU + 6 v_perm_b32
    const unsigned u0 = __float_as_uint(v.x), u1 = __float_as_uint(v.y), u2 = __float_as_uint(v.z), u3 = __float_as_uint(v.w);
    h.x = __uint_as_float(__builtin_amdgcn_perm(u1, u0, 0x07060302)), h.y = __uint_as_float(__builtin_amdgcn_perm(u3, u2, 0x07060302));
    const float r0 = v.x - __uint_as_float(u0 & 0xffff0000u), r1 = v.y - __uint_as_float(u1 & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(u2 & 0xffff0000u), r3 = v.w - __uint_as_float(u3 & 0xffff0000u);
    const unsigned q0 = __float_as_uint(r0), q1 = __float_as_uint(r1), q2 = __float_as_uint(r2), q3 = __float_as_uint(r3);
    m.x = __uint_as_float(__builtin_amdgcn_perm(q1, q0, 0x07060302)), m.y = __uint_as_float(__builtin_amdgcn_perm(q3, q2, 0x07060302));
    const float s0 = r0 - __uint_as_float(q0 & 0xffff0000u), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
    const float s2 = r2 - __uint_as_float(q2 & 0xffff0000u), s3 = r3 - __uint_as_float(q3 & 0xffff0000u);
    l.x = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302));
    l.y = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(s3), __float_as_uint(s2), 0x07060302));
  };
  // PREC 5: the split pieces of the gathered rows wait in registers (sa / sb) between the split -- interleaved with the MFMAs of the running K-step -- and the
  // LDS write behind the barrier
  float2 sa[SPL ? A_N : 1][3], sb[SPL ? B_N : 1][3];
  auto split_a = [&]() {
    if constexpr (SPL) {
#pragma unroll
      for (int i = 0; i < A_N; ++i) split4(ra[i], sa[i][0], sa[i][1], sa[i][2]);
    }
  };
  auto split_b = [&]() {
    if constexpr (SPL) {
#pragma unroll
      for (int i = 0; i < B_N; ++i) split4(rb[i], sb[i][0], sb[i][1], sb[i][2]);
    }
  };
  auto write_planes = [&](int buf) {
    if constexpr (SPL) {
      float* As = smem + buf * STAGE;
      float* Bs = As + A_FLOATS;
#pragma unroll
      for (int i = 0; i < A_N; ++i) {
        if constexpr (A_KC) {
          float* d = As + (r + 32 * i) * LD5 + g * 2;
          *reinterpret_cast<float2*>(d) = sa[i][0], *reinterpret_cast<float2*>(d + 16) = sa[i][1], *reinterpret_cast<float2*>(d + 32) = sa[i][2];
        } else {
          float* d = As + (r * LDA_T + (g + 8 * i) * 4) / 2;
          *reinterpret_cast<float2*>(d) = sa[i][0], *reinterpret_cast<float2*>(d + PLANE_A) = sa[i][1], *reinterpret_cast<float2*>(d + 2 * PLANE_A) = sa[i][2];
        }
      }
#pragma unroll
      for (int i = 0; i < B_N; ++i) {
        if constexpr (B_KC) {
          float* d = Bs + (r + 32 * i) * LD5 + g * 2;
          *reinterpret_cast<float2*>(d) = sb[i][0], *reinterpret_cast<float2*>(d + 16) = sb[i][1], *reinterpret_cast<float2*>(d + 32) = sb[i][2];
        } else {
          float* d = Bs + (r * LDB_T + (g + 8 * i) * 4) / 2;
          *reinterpret_cast<float2*>(d) = sb[i][0], *reinterpret_cast<float2*>(d + PLANE_B) = sb[i][1], *reinterpret_cast<float2*>(d + 2 * PLANE_B) = sb[i][2];
        }
      }
    }
  };
  auto store_tiles = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + A_FLOATS;
    if constexpr (SPL) {
      split_a();
      split_b();
      write_planes(buf);
      return;
    }
    if constexpr (NAT16) {      // two pixel rows per thread (r, r + 32), 16 bytes = eight channels per gather, stored as they are
#pragma unroll
      for (int kk = 0; kk < KX; ++kk) {
#pragma unroll
        for (int i = 0; i < A_NL; ++i)
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(As) + ((r + 32 * kk) * LDA_T + (g + 8 * i) * 8) * 2) = ra[kk * A_NL + i];
#pragma unroll
        for (int i = 0; i < B_NL; ++i)
          *reinterpret_cast<float4*>(reinterpret_cast<char*>(Bs) + ((r + 32 * kk) * LDB_T + (g + 8 * i) * 8) * 2) = rb[kk * B_NL + i];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < A_NL; ++i) {
      if constexpr (A_KC) *reinterpret_cast<float4*>(As + (r + 32 * i) * LDK + g * 4) = ra[i];
      else if constexpr (NAT16) *reinterpret_cast<float4*>(reinterpret_cast<char*>(As) + (r * LDA_T + (g + 8 * i) * 8) * 2) = ra[i];
      else if constexpr (TR) *reinterpret_cast<float2*>(reinterpret_cast<char*>(As) + (r * LDA_T + (g + 8 * i) * 4) * 2) = pack4(ra[i]);
      else *reinterpret_cast<float4*>(As + r * BM + (g + 8 * i) * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_NL; ++i) {
      if constexpr (B_KC) *reinterpret_cast<float4*>(Bs + (r + 32 * i) * LDK + g * 4) = rb[i];
      else if constexpr (NAT16) *reinterpret_cast<float4*>(reinterpret_cast<char*>(Bs) + (r * LDB_T + (g + 8 * i) * 8) * 2) = rb[i];
      else if constexpr (TR) *reinterpret_cast<float2*>(reinterpret_cast<char*>(Bs) + (r * LDB_T + (g + 8 * i) * 4) * 2) = pack4(rb[i]);
      else *reinterpret_cast<float4*>(Bs + r * BN + (g + 8 * i) * 4) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  auto compute_kg = [&](int buf, int kg) {   // one 8-k group (fp32) or one 16-k block (bf16: kg = 0, 2 cover the slab)
    const float* As = smem + buf * STAGE;
    const float* Bs = As + A_FLOATS;
    if constexpr (TR) {
      // bf16 [k][m] tiles, fragments through ds_read_b64_tr_b16: within a 16-lane group lane i hands in the address of row (i >> 2), column block
      // (i & 3) * 4 and receives column i of those four rows (tools/micro/tr_probe.hip) -- four consecutive k of its own m. Two reads = the lane's
      // eight k of v_mfma_f32_32x32x16_bf16 (lanes 32..63: the upper eight k of the 16-k block).
      if (kg & 1) return;                                   // two 16-k blocks per slab, issued on the even groups
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const int krow = (kg >> 1) * 16 + half * 8 + ((lane & 15) >> 2);
      const int cofs = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
      auto frag = [&](const float* T, int ld, int col0) {
        const short* base = reinterpret_cast<const short*>(T) + krow * ld + col0 + cofs;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * ld));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
      };
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = frag(As, LDA_T, wm * (BM / WM) + i * 32);
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[i] = frag(Bs, LDB_T, wn * (BN / WN) + i * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[n], acc[i][n], 0, 0, 0);
    } else if constexpr (SPL) {
      if (kg & 1) return;                                   // two 16-k blocks per slab, issued on the even groups
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const int krow = (kg >> 1) * 16 + half * 8 + ((lane & 15) >> 2);
      const int cofs = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
      auto frag_t = [&](const float* T, int ld, int col0) {      // m-contiguous plane [BK][ld] of bf16: the hardware transpose read, as PREC 3
        const short* base = reinterpret_cast<const short*>(T) + krow * ld + col0 + cofs;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * ld));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
      };
      const int kk = (kg >> 1) * 8 + half * 4;              // float offset of this lane-half's eight bf16 k inside a 64-byte plane row
      bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (A_KC) fa[p][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(As + (wm * (BM / WM) + i * 32 + l31) * LD5 + p * 16 + kk));
          else fa[p][i] = frag_t(As + p * PLANE_A, LDA_T, wm * (BM / WM) + i * 32);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          if constexpr (B_KC) fb[p][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(Bs + (wn * (BN / WN) + i * 32 + l31) * LD5 + p * 16 + kk));
          else fb[p][i] = frag_t(Bs + p * PLANE_B, LDB_T, wn * (BN / WN) + i * 32);
        }
      }
#define PM_SPL_PROD(PA, PB)                                                                                  \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int n = 0; n < TN; ++n) acc[i][n] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i], fb[PB][n], acc[i][n], 0, 0, 0);
      PM_SPL_PROD(2, 0) PM_SPL_PROD(0, 2) PM_SPL_PROD(1, 1) PM_SPL_PROD(1, 0) PM_SPL_PROD(0, 1) PM_SPL_PROD(0, 0)
#undef PM_SPL_PROD
    } else if constexpr (PREC == 2) {
      // bf16 operands in HBM and LDS (bf16.hip): the tensors enter as fp32-typed views with half the channels, so one float4 of a
      // k-contiguous LDS row is 8 consecutive bf16 k-values -- exactly this lane-half's operand of v_mfma_f32_32x32x16_bf16. Gather,
      // LDS layout (128-byte rows = 64 k, padded to 144 B) and epilogue are the fp32 code unchanged; 8 x fewer MFMA cycles per slab.
      static_assert(A_KC && B_KC, "bf16 tiles need k-contiguous operands (forward form)");
      const int kk = kg * 8 + half * 4;
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(As + (wm * (BM / WM) + i * 32 + l31) * LDK + kk));
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(Bs + (wn * (BN / WN) + i * 32 + l31) * LDK + kk));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[n], acc[i][n], 0, 0, 0);
    } else if constexpr (PREC == 0) {
      const int kk = kg * 8 + half * 4;  // this lane-half's 4 consecutive k of the 8-k group
      float fa[TM][4], fb[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / WM) + i * 32 + l31;
        if constexpr (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(As + row * LDK + kk);
          fa[i][0] = v.x, fa[i][1] = v.y, fa[i][2] = v.z, fa[i][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fa[i][j] = As[(kk + j) * BM + row];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int col = wn * (BN / WN) + i * 32 + l31;
        if constexpr (B_KC) {
          const float4 v = *reinterpret_cast<const float4*>(Bs + col * LDK + kk);
          fb[i][0] = v.x, fb[i][1] = v.y, fb[i][2] = v.z, fb[i][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[i][j] = Bs[(kk + j) * BN + col];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][j], fb[n][j], acc[i][n], 0, 0, 0);
    } else {
      if (kg & 1) return;                       // bf16: two 16-k blocks per slab, issued on the even groups
      const int kk = (kg >> 1) * 16 + half * 8;  // this lane-half's 8 consecutive k of the 16-k block
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / WM) + i * 32 + l31;
        float v[8];
        if constexpr (A_KC) {
          const float4 p = *reinterpret_cast<const float4*>(As + row * LDK + kk), q = *reinterpret_cast<const float4*>(As + row * LDK + kk + 4);
          v[0] = p.x, v[1] = p.y, v[2] = p.z, v[3] = p.w, v[4] = q.x, v[5] = q.y, v[6] = q.z, v[7] = q.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = As[(kk + j) * BM + row];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[i][j] = (__bf16)v[j];
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int col = wn * (BN / WN) + i * 32 + l31;
        float v[8];
        if constexpr (B_KC) {
          const float4 p = *reinterpret_cast<const float4*>(Bs + col * LDK + kk), q = *reinterpret_cast<const float4*>(Bs + col * LDK + kk + 4);
          v[0] = p.x, v[1] = p.y, v[2] = p.z, v[3] = p.w, v[4] = q.x, v[5] = q.y, v[6] = q.z, v[7] = q.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = Bs[(kk + j) * BN + col];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[i][j] = (__bf16)v[j];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[n], acc[i][n], 0, 0, 0);
    }
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int kg = 0; kg < BKW / 8; ++kg) compute_kg(buf, kg);
  };

  // ---- main loop: the gathers of slab kt+1 are issued (branch-free) ahead of the MFMAs of slab kt, whose 4096 matrix-pipe
  // cycles cover the load latency; LDS is double-buffered, one barrier per K-step ----------------------------------------
  if constexpr (SPL && NST == 1) {
    // PREC 5, one LDS stage, two blocks per CU. A wave's split (VALU) runs in the shadow of its OWN MFMAs -- measured: two equal waves on a SIMD do not fill each
    // other's phases (VALU issue 44 % + matrix pipe 57 % of the cycles, no overlap) -- so the K-step is software-pipelined inside the wave:
    //   first 16-k group's MFMAs  | split of the A rows of step kt + 1 (registers -> registers) | A gathers of step kt + 2 issued
    //   second 16-k group's MFMAs | split of the B rows of step kt + 1                          | B gathers of step kt + 2 issued
    //   barrier | write the planes of step kt + 1 | barrier
    // A gather has one K-step (~1.5 k cycles) to land before its split; a gather beyond the last K-step presents out-of-range offsets (no traffic).
    if (nk > 0) {
      load_tiles(0);
      advance();
      split_a(), split_b();
      write_planes(0);
      __syncthreads();
      load_tiles(1);
      advance();
      constexpr int NDS = 3 * (TM * (A_KC ? 1 : 2) + TN * (B_KC ? 1 : 2));      // fragment reads of one 16-k group
      constexpr int NMF = 6 * TM * TN;                                           // MFMAs of one 16-k group
      for (int kt = 0; kt < nk; ++kt) {
        compute_kg(0, 0);
        split_a();
        load_tiles(kt + 2, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        compute_kg(0, 2);
        split_b();
        load_tiles(kt + 2, 2);
        __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
#pragma unroll
        for (int q = 0; q < NMF; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();          // every wave is done reading the single buffer
        write_planes(0);          // unconditional (behind the last K-step: zeros nobody reads) and in the same basic block as the splits: a conditional write lets
        __syncthreads();          // the compiler sink the split arithmetic behind the barrier, out of the MFMAs' shadow
        advance();                // (its wave-uniform branches end the block: behind the write for the same reason)
      }
    }
  } else if constexpr (NST == 1) {
    if (nk > 0) {
      load_tiles(0);
      advance();
      store_tiles(0);
      __syncthreads();
      for (int kt = 0; kt < nk - 1; ++kt) {
        load_tiles(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        advance();
        compute(0);
        __syncthreads();          // every wave is done reading the single buffer
        store_tiles(0);
        __syncthreads();
      }
      compute(0);
    }
  } else if (nk > 0) {
    load_tiles(0);
    advance();
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk - 1; ++kt) {
      // Region 1: first k-group (fragment reads + TM*TN*4 MFMAs) with the whole gather of slab kt+1 (address VALU + the
      // buffer loads) interleaved into the 64-cycle MFMA shadows, so the matrix pipe restarts right after the barrier.
      compute_kg(kt & 1, 0);
      load_tiles(kt + 1);
      advance();
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN + (A_KC ? 0 : 3 * TM) + (B_KC ? 0 : 3 * TN), 0);  // fragment DS reads
#define PM_SG(I)                                                                   \
      if constexpr ((A_NL + B_NL) * KX > I) {                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  /* 2 MFMA            */ \
        __builtin_amdgcn_sched_group_barrier(0x006, 8, 0);  /* <= 8 VALU / SALU  */ \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  /* 1 buffer load     */ \
      }
      PM_SG(0) PM_SG(1) PM_SG(2) PM_SG(3) PM_SG(4) PM_SG(5) PM_SG(6) PM_SG(7)
#undef PM_SG
      __builtin_amdgcn_sched_barrier(0);
      // Region 2: the remaining three k-groups cover the load latency; then the LDS write of slab kt+1 and the barrier
#pragma unroll
      for (int kg = 1; kg < BKW / 8; ++kg) compute_kg(kt & 1, kg);
      store_tiles((kt + 1) & 1);
      __syncthreads();
    }
    compute((nk - 1) & 1);
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------
  // row form: a lane holds one output column and 16 rows of it (the MFMA's native C layout)
  float* Cb = a.C + by * a.c_bs + (a.ksplit > 1 ? (long)z * a.c_split : 0);
  const bool plain = a.ksplit > 1 || !(a.bias || a.scale || a.residual || a.relu);
  const bool full = m0 + BM <= a.M && n0 + BN <= a.Nn;
  const int rbase = m0 + wm * (BM / WM) + 4 * half, cbase = n0 + wn * (BN / WN) + l31;
  // Staged form (every 16-byte-aligned output): each wave parks 32 rows of its tile in LDS (the A / B stages are dead by now) and
  // stores WHOLE row segments -- (BN / WN) / 4 lanes per row, 16 bytes per lane, full 128-byte lines -- instead of 16 four-byte stores
  // per MFMA tile and lane. A finished tile is bound by the ISSUE of its stores, not by bandwidth: 64 -> 16 store instructions per
  // lane and 128 x 128 tile (tools/micro/gemm_lab.hip: +4 ... +18 % on the store-heavy shapes, never slower). The fused epilogue
  // operands are read as 16-byte vectors of the same row segments; values and their evaluation order are unchanged.
  // The bf16 tier (a.io16, split-K slabs excepted: they stay fp32): the same staging, a lane owns EIGHT columns of a row -- two 16-byte LDS reads, the fused
  // epilogue in fp32, one 16-byte store of eight bf16 (round to nearest even); the residual / skip-gradient operand is bf16 as well.
  const bool out16 = a.io16 && a.ksplit == 1;
  if (out16 && ((a.c_pitch | a.Nn | (a.residual ? a.res_pitch : 0)) & 7) == 0 &&
      ((reinterpret_cast<uintptr_t>(a.C) | reinterpret_cast<uintptr_t>(a.residual)) & 15) == 0) {
    constexpr int WC = BN / WN, LDC = WC + 4, LPR = WC / 8, RPI = 64 / LPR;
    static_assert(32 % RPI == 0, "row segments must tile the 32-row slab");
    __syncthreads();
    float* Ws = smem + wave * 32 * LDC;
    const int rr0 = lane / LPR, cc = (lane % LPR) * 8;
    const int col = n0 + wn * WC + cc;
    const bool cok = col < a.Nn;                      // Nn % 8 == 0: the group is all in or all out
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
    pm_bf16* C16 = reinterpret_cast<pm_bf16*>(a.C) + by * a.c_bs;
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
      if constexpr (STATS) {      // bf16 tier: statistics of the ROUNDED values of this 32-row slab (pm_common.h)
        if (a.stats) pm_slab_stats16<LDC, LPR, RPI>(Ws, rr0, cc, (long)m0 + wm * (BM / WM) + i * 32, a.M, a.Nn, col, cok, bi, sc, sh, a.stats);
      }
    }
    return;
  }
  if (out16) {      // unaligned / narrow bf16 outputs: per-element form
    pm_bf16* C16 = reinterpret_cast<pm_bf16*>(a.C) + by * a.c_bs;
    const pm_bf16* R16 = reinterpret_cast<const pm_bf16*>(a.residual);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int col = cbase + n * 32;
        const bool cok = col < a.Nn;
        float bi = 0.f, sc = 1.f, sh = 0.f;
        if (cok) {
          if (a.bias) bi = a.bias[col];
          if (a.scale) sc = a.scale[col], sh = a.shift[col];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = rbase + i * 32 + (q & 3) + 8 * (q >> 2);
          if (row < a.M && cok) {
            float v = (acc[i][n][q] + bi) * sc + sh;
            if (R16) v += pm_bf16_to_f32(R16[(long)row * a.res_pitch + col]);
            if (a.relu) v = fmaxf(v, 0.f);
            C16[(long)row * a.c_pitch + col] = pm_f32_to_bf16(v);
          }
        }
      }
    return;
  }
  const bool vec = a.stage_ep && ((a.c_pitch | a.Nn | (a.residual ? a.res_pitch : 0)) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(Cb) | reinterpret_cast<uintptr_t>(a.residual)) & 15) == 0;
  if (vec) {
    constexpr int WC = BN / WN, LDC = WC + 4, LPR = WC / 4, RPI = 64 / LPR;   // columns per wave, padded pitch, lanes per row, rows per instruction
    static_assert(32 % RPI == 0, "row segments must tile the 32-row slab");
    __syncthreads();                                  // every wave is done reading the last K-slab
    float* Ws = smem + wave * 32 * LDC;
    const int rr0 = lane / LPR, cc = (lane % LPR) * 4;
    const int col = n0 + wn * WC + cc;
    const bool cok = col < a.Nn;                      // Nn % 4 == 0: the quad is all in or all out
    const bool aff = !plain && (a.bias || a.scale), res = !plain && a.residual, relu = !plain && a.relu;
    float bi[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (aff && cok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.bias) bi[e] = a.bias[col + e];
        if (a.scale) sc[e] = a.scale[col + e], sh[e] = a.shift[col + e];
      }
    }
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
        float4 v = *reinterpret_cast<const float4*>(Ws + rr * LDC + cc);
        if (row < a.M && cok) {
          if (aff) v.x = (v.x + bi[0]) * sc[0] + sh[0], v.y = (v.y + bi[1]) * sc[1] + sh[1], v.z = (v.z + bi[2]) * sc[2] + sh[2], v.w = (v.w + bi[3]) * sc[3] + sh[3];
          if (res) {
            const float4 qv = PM_LD4(a.residual + row * a.res_pitch + col);
            v.x += qv.x, v.y += qv.y, v.z += qv.z, v.w += qv.w;
          }
          if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
          PM_ST4(Cb + row * a.c_pitch + col, v);
        }
      }
      if constexpr (STATS) {   // BatchNorm statistics of this 32-row slab (train mode: the epilogue is the convolution plus at most a
                       // bias), two passes over the slab still parked in LDS: per column the mean over the valid rows, then M2 around it;
                       // the RPI lanes sharing a column quad are combined by lane exchanges (fixed order). The slab is re-read rather than
                       // kept in registers so that the kernel's register budget (and occupancy) is the one without statistics.
        const long slab_row0 = m0 + wm * (BM / WM) + i * 32;
        const float cnt = (float)max(0l, min(32l, (long)a.M - slab_row0));
        auto slab_row = [&](int r0, bool& rok) {
          float4 v = *reinterpret_cast<const float4*>(Ws + (r0 + rr0) * LDC + cc);
          v.x = (v.x + bi[0]) * sc[0] + sh[0], v.y = (v.y + bi[1]) * sc[1] + sh[1], v.z = (v.z + bi[2]) * sc[2] + sh[2], v.w = (v.w + bi[3]) * sc[3] + sh[3];
          rok = slab_row0 + r0 + rr0 < a.M;
          return v;
        };
        float s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += RPI) {
          bool rok;
          const float4 v = slab_row(r0, rok);
          s1[0] += rok ? v.x : 0.f, s1[1] += rok ? v.y : 0.f, s1[2] += rok ? v.z : 0.f, s1[3] += rok ? v.w : 0.f;
        }
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) s1[e] += __shfl_xor(s1[e], o, 64);
        float mu[4], m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) mu[e] = cnt > 0.f ? s1[e] / cnt : 0.f;
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += RPI) {
          bool rok;
          const float4 v = slab_row(r0, rok);
          const float d0 = v.x - mu[0], d1 = v.y - mu[1], d2 = v.z - mu[2], d3 = v.w - mu[3];
          m2[0] += rok ? d0 * d0 : 0.f, m2[1] += rok ? d1 * d1 : 0.f, m2[2] += rok ? d2 * d2 : 0.f, m2[3] += rok ? d3 * d3 : 0.f;
        }
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
          for (int e = 0; e < 4; ++e) m2[e] += __shfl_xor(m2[e], o, 64);
        if (rr0 == 0 && cok && cnt > 0.f) {
          float* dst = a.stats + ((slab_row0 >> 5) * (long)a.Nn + col) * 2;
          PM_ST4(dst, make_float4(mu[0], m2[0], mu[1], m2[1]));
          PM_ST4(dst + 4, make_float4(mu[2], m2[2], mu[3], m2[3]));
        }
      }
    }
    return;
  }
  if (full) {  // interior tile: straight-line epilogue specialised on the (wave-uniform) fused operations, no per-element predicate
    auto run = [&](auto AFF, auto RES, auto RELU) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const int col = cbase + n * 32;
          float bi = 0.f, sc = 1.f, sh = 0.f;
          if constexpr (decltype(AFF)::value) {
            if (a.bias) bi = a.bias[col];
            if (a.scale) sc = a.scale[col], sh = a.shift[col];
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const long row = rbase + i * 32 + (q & 3) + 8 * (q >> 2);
            float v = acc[i][n][q];
            if constexpr (decltype(AFF)::value) v = (v + bi) * sc + sh;
            if constexpr (decltype(RES)::value) v += a.residual[row * a.res_pitch + col];
            if constexpr (decltype(RELU)::value) v = fmaxf(v, 0.f);
            Cb[row * a.c_pitch + col] = v;
          }
        }
    };
    using T1 = std::true_type;
    using T0 = std::false_type;
    const bool aff = !plain && (a.bias || a.scale), res = !plain && a.residual, relu = !plain && a.relu;
    if (!aff && !res && !relu) run(T0{}, T0{}, T0{});
    else if (!aff && res && !relu) run(T0{}, T1{}, T0{});       // dgrad + fused skip gradient
    else if (aff && !res && !relu) run(T1{}, T0{}, T0{});       // conv + bias
    else if (aff && !res && relu) run(T1{}, T0{}, T1{});        // eval: conv + folded BN + ReLU
    else if (aff && res && relu) run(T1{}, T1{}, T1{});         // eval: bottleneck tail
    else if (aff && res && !relu) run(T1{}, T1{}, T0{});
    else if (!aff && res && relu) run(T0{}, T1{}, T1{});
    else run(T0{}, T0{}, T1{});
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = cbase + n * 32;
      const bool cok = col < a.Nn;
      float bi = 0.f, sc = 1.f, sh = 0.f;
      if (!plain && cok) {
        if (a.bias) bi = a.bias[col];
        if (a.scale) sc = a.scale[col], sh = a.shift[col];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = rbase + i * 32 + (q & 3) + 8 * (q >> 2);
        if (row < a.M && cok) {
          float v = acc[i][n][q];
          if (!plain) {
            v = (v + bi) * sc + sh;
            if (a.residual) v += a.residual[(long)row * a.res_pitch + col];
            if (a.relu) v = fmaxf(v, 0.f);
          }
          Cb[(long)row * a.c_pitch + col] = v;
        }
      }
    }
  return;
}

}  // namespace
