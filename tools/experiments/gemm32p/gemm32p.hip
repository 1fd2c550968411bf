// fp32 tier, round 5 (second session): the GEMM-shaped fp32 launches -- the batched Winograd point products M[p] = V[p] U[p]^T (deepv3plus.py / Resnet.py 3x3 convolutions
// through winograd.hip) and the pointwise (1x1, stride 1) forward convolutions -- as a PERSISTENT producer / consumer kernel, the structure of conv16w.hip's conv16p_kernel:
// one block per CU walks (batch point, 256 x 128 tile) units; four producer waves issue the LDS-DMA fetches of a three-stage ring (128-byte = 32-float rows per K-step) that
// keeps running across unit boundaries; eight waves multiply 64 x 64 sub-tiles with v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: same arithmetic type as the shipped
// register-staged kernel, only the order of the K sum inside a 32-float step differs) and store the tile straight from the accumulators (a lane owns one output column: a
// store instruction writes two whole 128-byte row segments).
// Why persistent: an fp32 K-step is 4 096 MFMA clocks per wave, so fetch bandwidth is no issue at all (6 B/clk/CU); what the one-block-per-tile kernels lose is the prologue
// and the epilogue of every 8-64-step tile (MfmaUtil 0.69 with the waves ready to issue 70 % of the time). Here the next unit's first stages are in flight while this one
// is multiplied, and the stores are fire-and-forget.
#include <stdlib.h>
#include <algorithm>

#include "pm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

int g_gemm32p = getenv("PM_GEMM32P") ? atoi(getenv("PM_GEMM32P")) : 1;      // 0: the register-staged kernel everywhere (A/B), 1: this kernel where pm_gemm32p_plan accepts the shape

namespace {

constexpr int BKB = 128;      // bytes per row and K-step (32 floats)
constexpr int NP = 4, FT = NP * 64;

__device__ __forceinline__ int xcd_remap_f(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void dma16f(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm_f() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void ring_barrier_f() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// a store the compiler's wait bookkeeping does not see (conv16w.hip): the next unit's first fragment read must not wait for this unit's output
__device__ __forceinline__ void st4_untracked_f(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((8 + NP) * 64, (8 + NP) / 4) void gemm32p_kernel(const pm_gemm32 a) {
  static_assert(WM * WN == 8, "eight multiplying waves");
  constexpr int A_IT = BM * 8 / FT, B_IT = BN * 8 / FT, FETCH = A_IT + B_IT;
  constexpr int A_BYTES = BM * BKB, STAGE = (BM + BN) * BKB;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  extern __shared__ __align__(16) char lds[];

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ntiles = a.tiles_m * a.tiles_n, total = ntiles * a.batch, G = gridDim.x;
  const int nk = a.K >> 5;
  auto decode = [&](int v, int& b, int& m0, int& n0) {
    b = v / ntiles;
    const int lid = xcd_remap_f(v - b * ntiles, ntiles);
    m0 = (lid / a.tiles_n) * BM, n0 = (lid % a.tiles_n) * BN;
  };

  if (wave >= 8) {
    // ================================================== producer waves ==================================================
    const int wave_u = wave - 8, t = wave_u * 64 + lane;
    constexpr int OOB = 0x7fffffff;
    const int pitchb = (int)a.a_pitch * 4, kb = a.K * 4;
    int aoff[A_IT], boff[B_IT];
    __amdgpu_buffer_rsrc_t rA, rB;
    int s_kb = 0, left = 0;
    int vf = blockIdx.x;
    auto open_unit = [&]() {
      if (vf >= total) {
        left = 0;
        return;
      }
      int b, m0, n0;
      decode(vf, b, m0, n0);
      rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A + (long)b * a.a_bs), 0, (int)((long)a.M * pitchb), 0x00020000);
      rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B + (long)b * a.b_bs), 0, (int)((long)a.Nn * kb), 0x00020000);
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        const int u = it * FT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
        const int m = m0 + row;
        aoff[it] = m < a.M ? m * pitchb + ch * 16 : OOB;
      }
#pragma unroll
      for (int it = 0; it < B_IT; ++it) {
        const int u = it * FT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
        const int n = n0 + row;
        boff[it] = n < a.Nn ? n * kb + ch * 16 : OOB;
      }
      s_kb = 0, left = nk;
    };
    int slot = 0;
    auto issue = [&]() -> int {
      if (left == 0) {
        if (vf >= total) return 0;
        vf += G;
        open_unit();
        if (left == 0) return 0;
      }
      char* la = lds + slot * STAGE;
      char* lb = la + A_BYTES;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) dma16f(rA, la + (it * FT + wave_u * 64) * 16, aoff[it], s_kb);
#pragma unroll
      for (int it = 0; it < B_IT; ++it) dma16f(rB, lb + (it * FT + wave_u * 64) * 16, boff[it], s_kb);
      s_kb += BKB;
      --left;
      slot = slot == 2 ? 0 : slot + 1;
      return 1;
    };
    open_unit();
    int ahead = issue();
    ahead += issue();
    for (int vc = blockIdx.x; vc < total; vc += G) {      // mirror of the multiplying waves' barrier sequence: nk steps per unit
      for (int kt = 0; kt < nk; ++kt) {
        if (ahead >= 2) wait_vm_f<FETCH>();
        else wait_vm_f<0>();
        ring_barrier_f();
        ahead += issue() - 1;
      }
    }
    return;
  }

  // ================================================== multiplying waves ==================================================
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  // v_mfma_f32_32x32x2_f32: lane (l31, half) hands in A[row l31][k slot half] and B[col l31][k slot half]. The two slots of the 2 c-th and (2 c + 1)-th MFMA of a step
  // are floats 4 c + 2 half and 4 c + 2 half + 1 of the row: one 8-byte read per fragment and 16-byte chunk (any assignment of the step's 32 k-values to slots is a
  // valid order of the sum as long as both operands use it).
  int ra_off[TM], ra_key[TM], rb_off[TN], rb_key[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ra = wm * (BM / WM) + i * 32 + l31;
    ra_off[i] = ra * BKB + half * 8, ra_key[i] = (ra >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int rb = wn * (BN / WN) + j * 32 + l31;
    rb_off[j] = A_BYTES + rb * BKB + half * 8, rb_key[j] = (rb >> 1) & 7;
  }
  const bool aff = a.bias || a.scale, relu = a.relu != 0;      // (no residual: pm_gemm32p_plan leaves those launches to the tile kernel)
  int rd = 0;
  for (int vc = blockIdx.x; vc < total; vc += G) {
    int b, m0, n0;
    decode(vc, b, m0, n0);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      ring_barrier_f();
      const char* ls = lds + rd * STAGE;
      // fragments of chunk c + 1 are requested before the MFMAs of chunk c are issued (two register sets)
      f32x2 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x2*>(ls + ra_off[i] + ((0 ^ ra_key[i]) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x2*>(ls + rb_off[j] + ((0 ^ rb_key[j]) << 4));
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int cu = c & 1, nx = cu ^ 1;
        if (c + 1 < 8) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[nx][i] = *reinterpret_cast<const f32x2*>(ls + ra_off[i] + (((c + 1) ^ ra_key[i]) << 4));
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[nx][j] = *reinterpret_cast<const f32x2*>(ls + rb_off[j] + (((c + 1) ^ rb_key[j]) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);      // the requests stay in front of this chunk's MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu][i].x, fb[cu][j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu][i].y, fb[cu][j].y, acc[i][j], 0, 0, 0);
      }
      rd = rd == 2 ? 0 : rd + 1;
    }
    // ---- epilogue, straight from the accumulators: this lane's column of each 32 x 32 sub-tile, rows (q & 3) + 8 (q >> 2) + 4 half ----
    float* Cb = a.C + (long)b * a.c_bs;
    float bi[TN], sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {      // per-column constants before the first store (a register load issued behind a store would wait for it)
      const int col = n0 + wn * (BN / WN) + j * 32 + l31;
      bi[j] = 0.f, sc[j] = 1.f, sh[j] = 0.f;
      if (aff && col < a.Nn) {      // inline-asm loads with their own wait below: no vector-memory operation the compiler tracks is left in this path (see st4_untracked_f)
        if (a.bias) asm volatile("global_load_dword %0, %1, off" : "=v"(bi[j]) : "v"(a.bias + col) : "memory");
        if (a.scale) {
          asm volatile("global_load_dword %0, %1, off" : "=v"(sc[j]) : "v"(a.scale + col) : "memory");
          asm volatile("global_load_dword %0, %1, off" : "=v"(sh[j]) : "v"(a.shift + col) : "memory");
        }
      }
    }
    if (aff) {
      wait_vm_f<0>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 32 + l31;
      if (col >= a.Nn) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const long row = m0 + wm * (BM / WM) + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
          if (row >= a.M) continue;
          float v = acc[i][j][q];
          if (aff) v = (v + bi[j]) * sc[j] + sh[j];
          if (relu) v = fmaxf(v, 0.f);
          st4_untracked_f(Cb + row * a.c_pitch + col, v);
        }
    }
  }
}

template <int BM, int BN, int WM, int WN>
void launch_gemm32p(const pm_gemm32& k, hipStream_t st) {
  constexpr size_t smem = (size_t)3 * (BM + BN) * BKB;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static const int ncu = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm32p_kernel<BM, BN, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int total = k.tiles_m * k.tiles_n * k.batch;
  hipLaunchKernelGGL((gemm32p_kernel<BM, BN, WM, WN>), dim3(std::min(total, ncu)), dim3((8 + NP) * 64), smem, st, k);
}

}  // namespace

// Shapes the kernel takes: K in whole 32-float steps, 16-byte aligned rows, 32-bit byte offsets per batch point, enough tiles to fill the chip more than once (a persistent
// block pays nothing for many small units, but a launch with fewer units than CUs has nothing to pipeline).
bool pm_gemm32p_plan(pm_gemm32* k) {
  if (!g_gemm32p) return false;
  if (k->residual || k->K % 32 || k->K < 128 || k->a_pitch % 4 || k->c_pitch < k->Nn || k->Nn < 128 || k->M < 256) return false;
  if ((long)k->M * k->a_pitch * 4 >= (1l << 31) || (long)k->Nn * k->K * 4 >= (1l << 31)) return false;
  if (!pm_aligned16(k->A) || !pm_aligned16(k->B) || (k->a_bs | k->b_bs) % 4) return false;
  // 256 x 128, or 128 x 256 when that wastes fewer padded rows / columns
  const double f0 = (double)(pm_cdiv(k->M, 256) * 256) * (pm_cdiv(k->Nn, 128) * 128), f1 = (double)(pm_cdiv(k->M, 128) * 128) * (pm_cdiv(k->Nn, 256) * 256);
  if (f1 < f0) k->bm = 128, k->bn = 256;
  else k->bm = 256, k->bn = 128;
  k->tiles_m = pm_cdiv(k->M, k->bm), k->tiles_n = pm_cdiv(k->Nn, k->bn);
  static const int min_units = getenv("PM_GEMM32P_MIN_UNITS") ? atoi(getenv("PM_GEMM32P_MIN_UNITS")) : 512;
  if ((long)k->tiles_m * k->tiles_n * k->batch < min_units) return false;
  return true;
}

int pm_gemm32p_launch(const pm_gemm32* k, hipStream_t st) {
  if (k->bm == 256 && k->bn == 128) launch_gemm32p<256, 128, 4, 2>(*k, st);
  else if (k->bm == 128 && k->bn == 256) launch_gemm32p<128, 256, 2, 4>(*k, st);
  else {
    pm_set_error("gemm32p: no %d x %d tile", k->bm, k->bn);
    return PM_EUNSUPPORTED;
  }
  return pm_check_launch("gemm32p");
}
