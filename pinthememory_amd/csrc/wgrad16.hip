// BASELINE configs[2], round 5 (second session): the WEIGHT GRADIENT of the bf16 tier on LDS-DMA -- the persistent producer / consumer ring of conv16w.hip turned to
//   dw[cout][tap][cin] = sum over output pixels p of dy[p][cout] * x[p shifted by the tap][cin]          (nn.Conv2d backward of deepv3plus.py / Resnet.py on bf16 rows)
// Both operands lie pixel-major in HBM (NHWC), i.e. the reduction index is the SLOW one of both: a 16-byte fetch is eight channels of one pixel. The fetches go
// straight into LDS as [64 pixels][BM couts] and [64 pixels][BN cins] tiles (one tap, one block of input channels per tile), and the MFMA fragments -- eight
// consecutive pixels of one channel per lane -- come out of them through the hardware transpose read ds_read_b64_tr_b16 (the addressing of conv_igemm.hip's PREC 4
// form). No pad between the pixel rows: the 64-byte chunks of a row are XOR-ed with (pixel & 3) on the SOURCE side of the fetch, which puts the four rows one
// transpose read touches on four disjoint bank quarters.
// One block per CU walks (tile, pixel-range) units: four producer waves issue the fetches of a three-stage ring that runs on across unit boundaries and own all the
// index arithmetic (pixel -> image / row / column once per unit and piece, then stepped by 64 pixels); eight waves multiply 64 x 64 sub-tiles and write fp32 partial
// tiles (split-K slabs, reduced in fixed order by splitk_reduce as before) -- deterministic. Replaces the register-staged conv_igemm_kernel<2, 128, 128, .., 4, 2> where
// Cin fills the tile's 128-channel blocks at least 3 / 4 (every 3x3 / 1x1 of the backbone from 128 channels up, the decoder's 304-channel concat).
#include <stdlib.h>
#include <algorithm>

#include "pm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


namespace {

constexpr int BKP = 64;       // pixels per K-step
constexpr int NP = 4;         // producer waves
constexpr int FT = NP * 64;   // fetching threads

__device__ __forceinline__ int xcd_remap_g(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void dma16g(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm_g() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void ring_barrier_g() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// stores the compiler's wait bookkeeping does not see (conv16w.hip): the first fragment read of the next unit must not wait for this unit's partial tile
__device__ __forceinline__ void st16_untracked_g(void* p, u32x4 q) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(q) : "memory"); }
__device__ __forceinline__ void st4_untracked_g(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((8 + NP) * 64, (8 + NP) / 4) void wgrad16_kernel(const pm_wgrad16 a) {
  static_assert(WM * WN == 8, "eight multiplying waves");
  constexpr int A_ROWB = BM * 2, B_ROWB = BN * 2;                 // bytes per pixel row of the dy / x tile
  constexpr int A_BYTES = BKP * A_ROWB, STAGE = BKP * (A_ROWB + B_ROWB);
  constexpr int A_SL = BM / 8, B_SL = BN / 8;                     // 16-byte slots per pixel row
  constexpr int A_IT = BKP * A_SL / FT, B_IT = BKP * B_SL / FT;   // fetches per producer lane and K-step
  constexpr int FETCH = A_IT + B_IT;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int LDS_SUB = 36;
  static_assert(8 * 32 * LDS_SUB * 4 <= STAGE && A_IT >= 1 && B_IT >= 1 && TM >= 1 && TN >= 1, "bad tile config");
  extern __shared__ __align__(16) char lds[];

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ntiles = a.tiles_m * a.tiles_n, total = ntiles * a.ksplit, G = gridDim.x;
  const int cpt = (a.Cin + BN - 1) / BN;      // channel blocks per tap; the last one may be partial (304 / 320 channels): its missing channels are fetched as zeros and not stored
  auto decode = [&](int v, int& m0, int& tap, int& c0, int& z, int& p0, int& nk) {
    z = v / ntiles;
    const int lid = xcd_remap_g(v - z * ntiles, ntiles);
    const int tn = lid % a.tiles_n;
    m0 = (lid / a.tiles_n) * BM;
    tap = tn / cpt, c0 = (tn - tap * cpt) * BN;
    p0 = z * a.kper;
    nk = (min(a.P, p0 + a.kper) - p0 + BKP - 1) / BKP;
  };

  if (wave >= 8) {
    // ================================================== producer waves ==================================================
    const int wave_u = wave - 8, t = wave_u * 64 + lane;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.DY), 0, (int)((long)a.P * a.dy_pitch * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.X), 0, (int)((long)a.N * a.H * a.W * a.x_pitch * 2), 0x00020000);
    constexpr int OOB = 0x7fffffff;
    const int dyb = (int)a.dy_pitch * 2, xb = (int)a.x_pitch * 2;
    // fixed per lane: pixel row inside the stage and (swizzled) channel slot of every piece
    int a_row[A_IT], a_col[A_IT], b_row[B_IT], b_src[B_IT], b_col[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int u = it * FT + t, r = u / A_SL, s = (u % A_SL) ^ ((r & 3) << 2);
      a_row[it] = r, a_col[it] = s * 16;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int u = it * FT + t, r = u / B_SL, s = (u % B_SL) ^ ((r & 3) << 2);
      b_row[it] = r, b_src[it] = s * 16, b_col[it] = s * 16;
    }
    // state of the unit being fetched
    int a_off[A_IT];                          // byte offset of (pixel row, cout slot) relative to the step's first pixel, or OOB for couts beyond Cout
    int b_img[B_IT], b_oy[B_IT], b_ox[B_IT];  // output pixel of the piece's row at the current step
    int b_ch = 0, dy0 = 0, dx0 = 0;           // channel byte offset of the tile, tap displacement
    int pix = 0, pend = 0, left = 0;          // first pixel of the next step, end of the unit's pixel range, steps left
    int vf = blockIdx.x;
    auto open_unit = [&]() {
      for (; vf < total; vf += G) {
        int m0, tap, c0, z, p0, nk;
        decode(vf, m0, tap, c0, z, p0, nk);
        if (nk <= 0) continue;
        left = nk, pix = p0, pend = min(a.P, p0 + a.kper);
#pragma unroll
        for (int it = 0; it < A_IT; ++it) a_off[it] = (m0 * 2 + a_col[it] < a.Cout * 2) ? a_row[it] * dyb + m0 * 2 + a_col[it] : OOB;
        const int ky = tap / a.kw, kx = tap - ky * a.kw;
        dy0 = ky * a.dil - a.pad, dx0 = kx * a.dil - a.pad;
        b_ch = c0 * 2;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) b_col[it] = (c0 * 2 + b_src[it] < a.Cin * 2) ? b_src[it] : -1;      // -1: a channel group beyond Cin (partial last block) reads zeros
        // one index decomposition per unit (row 0 of the lane), the other pieces step on from it
        constexpr int RSTEP = FT / B_SL;
        int p = p0 + b_row[0];
        int img = p / (a.Ho * a.Wo);
        const int rem = p - img * (a.Ho * a.Wo);
        int oy = rem / a.Wo, ox = rem - oy * a.Wo;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
          b_img[it] = img, b_oy[it] = oy, b_ox[it] = ox;
          ox += RSTEP;
          while (ox >= a.Wo) {
            ox -= a.Wo;
            if (++oy == a.Ho) oy = 0, ++img;
          }
        }
        return;
      }
      left = 0;
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
      const int s_a = pix * dyb;      // scalar offset of the step's first pixel row (the range check of the descriptor covers the vector offset alone)
#pragma unroll
      for (int it = 0; it < A_IT; ++it) dma16g(rA, la + (it * FT + wave_u * 64) * 16, (pix + a_row[it] < pend) ? a_off[it] : OOB, s_a);
#pragma unroll
      for (int it = 0; it < B_IT; ++it) {
        const int iy = b_oy[it] * a.stride + dy0, ix = b_ox[it] * a.stride + dx0;
        const bool ok = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W) & (pix + b_row[it] < pend) & (b_col[it] >= 0);
        dma16g(rB, lb + (it * FT + wave_u * 64) * 16, ok ? ((b_img[it] * a.H + iy) * a.W + ix) * xb + b_ch + b_col[it] : OOB, 0);
        // this piece's pixel, 64 further on
        b_ox[it] += BKP;
        while (b_ox[it] >= a.Wo) {
          b_ox[it] -= a.Wo;
          if (++b_oy[it] == a.Ho) b_oy[it] = 0, ++b_img[it];
        }
      }
      pix += BKP;
      --left;
      slot = slot == 2 ? 0 : slot + 1;
      return 1;
    };
    open_unit();
    int ahead = issue();
    ahead += issue();
    for (int vc = blockIdx.x; vc < total; vc += G) {      // mirror of the multiplying waves' barrier sequence
      int m0, tap, c0, z, p0, nk;
      decode(vc, m0, tap, c0, z, p0, nk);
      for (int kt = 0; kt < nk; ++kt) {
        if (ahead >= 2) wait_vm_g<FETCH>();
        else wait_vm_g<0>();
        ring_barrier_g();
        ahead += issue() - 1;
      }
      ring_barrier_g();
    }
    return;
  }

  // ================================================== multiplying waves ==================================================
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  // transpose reads: within a 16-lane group lane i hands in the address of pixel row (i >> 2), 8-byte column block (i & 3) and receives column i of those four rows
  const int frow = half * 8 + ((lane & 15) >> 2);                      // pixel row inside a 16-pixel block (the second read: + 4)
  const int fcol = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;      // byte offset inside the 64-byte chunk of the lane's 32-channel group
  const int fswz = (frow & 3) << 6;                                    // the row's chunk swizzle (rows r and r + 4 share it)
  int fa_off[TM], fb_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa_off[i] = frow * A_ROWB + (((wm * (BM / WM) + i * 32) * 2 + fcol) ^ fswz);
#pragma unroll
  for (int j = 0; j < TN; ++j) fb_off[j] = A_BYTES + frow * B_ROWB + (((wn * (BN / WN) + j * 32) * 2 + fcol) ^ fswz);
  auto frag = [&](const char* p, int rowb) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * rowb));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  int rd = 0;
  for (int vc = blockIdx.x; vc < total; vc += G) {
    int m0, tap, c0, z, p0, nk;
    decode(vc, m0, tap, c0, z, p0, nk);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      ring_barrier_g();
      const char* ls = lds + rd * STAGE;
#pragma unroll
      for (int kb = 0; kb < BKP / 16; ++kb) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = frag(ls + kb * 16 * A_ROWB + fa_off[i], A_ROWB);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = frag(ls + kb * 16 * B_ROWB + fb_off[j], B_ROWB);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      rd = rd == 2 ? 0 : rd + 1;
    }
    // ---- epilogue: the fp32 partial tile, 32 x 32 slabs through the ring slot read last (conv16w.hip conv16p_kernel) ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ring_barrier_g();
    float* Ws = reinterpret_cast<float*>(lds + (rd == 0 ? 2 : rd - 1) * STAGE) + wave * 32 * LDS_SUB;
    float* Cf = a.C + (long)z * a.c_split;
    const long cp = a.Nn;
    const int rr0 = lane >> 3, cc = (lane & 7) * 4;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int cl = c0 + wn * (BN / WN) + n * 32 + cc;      // channel inside the tap: groups of four, Cin % 8 == 0 -- a group is whole or beyond Cin
      const int col = tap * a.Cin + cl;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) Ws[((q & 3) + 8 * (q >> 2) + 4 * half) * LDS_SUB + l31] = acc[i][n][q];
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += 8) {
          const int rr = r0 + rr0;
          const long row = m0 + wm * (BM / WM) + i * 32 + rr;
          const float4 v = *reinterpret_cast<const float4*>(Ws + rr * LDS_SUB + cc);
          if (row >= a.M || cl >= a.Cin) continue;
          if ((cp & 3) == 0) st16_untracked_g(Cf + row * cp + col, u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)});
          else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) st4_untracked_g(Cf + row * cp + col + k, e[k]);
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
void launch_wgrad16(const pm_wgrad16& k, hipStream_t st) {
  constexpr size_t smem = (size_t)3 * BKP * (BM + BN) * 2;
  static_assert(smem <= 160 * 1024, "LDS budget");
  const int ncu = pm_device_once([] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad16_kernel<BM, BN, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const int total = k.tiles_m * k.tiles_n * k.ksplit;
  hipLaunchKernelGGL((wgrad16_kernel<BM, BN, WM, WN>), dim3(std::min(total, ncu)), dim3((8 + NP) * 64), smem, st, k);
}

}  // namespace

// The shapes the kernel takes: both tensors bf16 with whole 16-byte channel groups, Cin in 128-channel blocks (at least three quarters full in total), at least 128 output
// channels, 32-bit byte offsets. Fills the tile plan; the pixel split (kper, ksplit) is the caller's (the slabs were sized for it).
bool pm_wgrad16_plan(pm_wgrad16* k) {
  if (!pm_route.wgrad16) return false;
  // Cin: 128-channel blocks at least three quarters full in total (the decoder's 304-channel concat: 2.4 blocks, 21 % of the multiplications on zeros)
  const int cblocks = pm_cdiv(k->Cin, 128);
  if (k->Cin % 8 || k->Cin < 128 || k->Cin * 4 < cblocks * 128 * 3 || k->Cout < 128 || (k->x_pitch | k->dy_pitch) % 8 || k->kper % BKP) return false;
  if ((long)k->N * k->H * k->W * k->x_pitch * 2 >= (1l << 31) || (long)k->P * k->dy_pitch * 2 >= (1l << 31)) return false;
  if (!pm_aligned16(k->X) || !pm_aligned16(k->DY) || !pm_aligned16(k->C)) return false;
  // 256 couts x 128 cins, or 128 x 256 when that wastes fewer rows (Cout = 128, 384, ...) and Cin allows it
  const int pad256 = pm_cdiv(k->Cout, 256) * 256 - k->Cout, pad128 = pm_cdiv(k->Cout, 128) * 128 - k->Cout;
  if (pad128 < pad256 && k->Cin % 256 == 0) k->bm = 128, k->bn = 256;      // (never with a partial channel block)
  else k->bm = 256, k->bn = 128;
  k->tiles_m = pm_cdiv(k->Cout, k->bm);
  k->tiles_n = k->kh * k->kw * pm_cdiv(k->Cin, k->bn);
  return true;
}

int pm_wgrad16_launch(const pm_wgrad16* k, hipStream_t st) {
  if (k->bm == 256 && k->bn == 128) launch_wgrad16<256, 128, 4, 2>(*k, st);
  else if (k->bm == 128 && k->bn == 256) launch_wgrad16<128, 256, 2, 4>(*k, st);
  else {
    pm_set_error("wgrad16: no %d x %d tile", k->bm, k->bn);
    return PM_EUNSUPPORTED;
  }
  return pm_check_launch("wgrad16");
}
