// BASELINE configs[2], round 5: the WIDE forms of the LDS-DMA bf16 implicit-GEMM convolution (conv16.hip) -- ONE large output tile per CU, eight waves:
//   * 256 x 256, two 64 KB LDS stages (128 FLOP per staged byte; wave tile 128 x 64: 0.75 KB of LDS reads per MFMA): the fetches of K-step k + 1 are issued behind the barrier
//     of step k and fly under its 32 MFMAs per wave. THE FORM THE PLANNER USES: the deep reductions (ASPP 3x3 2048 -> 256 on the 48 x 48 maps: 690 TF on the 64 x 128 tiles,
//     764 with skipped filter rows, 834-866 here; the auxiliary head's 3x3 1024 -> 512: 660 -> 780) and very wide outputs of a medium reduction (256 -> 2048: 697 -> 780);
//   * 256 x 128 / 128 x 256, a three-stage LDS ring filled with COUNTED waits (s_waitcnt vmcnt(N) + a bare s_barrier: the fetches of K-steps k + 1 and k + 2 stay in flight
//     while step k is multiplied; 85 FLOP per staged byte): 733-769 TF on the same shapes, level with the narrow tiles elsewhere -- kept for A/B (PM_C16W_CFG) and the tests.
// Why big tiles: every kernel of this family moves ~20-30 bytes per clock and CU from L2 into LDS whatever its tile (DESIGN section 7 "Round 5"), so TFLOP/s ~ FLOP per
// staged byte x that; the 64 x 128 / 128 x 128 tiles of conv16.hip stage 43 / 64. Why not everywhere: with one block per CU the tile count has to be balanced against the
// 256 CUs (split-K by the planner's cost model, conv16.hip pm_conv16_plan); the 48 x 48 maps give 72 x N / 256 tiles of 256 x 256.
// Same operands, same addressing (buffer descriptors, out-of-range offsets deliver zeros, source-side XOR swizzle), same epilogues, same invisible-filter-row skipping as
// conv16.hip. Replaces nn.Conv2d forward / input gradient of deepv3plus.py:72-81,419-425 (ASPP, dsn) on the bf16 tier where the planner picks it.
#include <stdlib.h>
#include <algorithm>

#include "pm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


namespace {

constexpr int BKB = 128;      // bytes per row and K-step (64 bf16)
constexpr int NT = 512;       // threads per block: eight waves

__device__ __forceinline__ int xcd_remap_w(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void dma16w(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt untouched): gfx9 encoding vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]
template <int N>
__device__ __forceinline__ void wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// NST = 3: the ring (256 x 128 / 128 x 256, 48 KB per stage). NST = 2: 256 x 256 (64 KB per stage, 128 FLOP per staged byte): the fetches of K-step k + 1 are issued behind
// the barrier of step k and fly under its 32 MFMAs per wave; the wait in front of the next barrier is a full one (nothing younger is in flight).
template <int BM, int BN, int WM, int WN, int NST>
__global__ __launch_bounds__(NT, 2) void conv16w_kernel(const pm_conv16 a) {
  static_assert(WM * WN == 8, "eight waves");
  constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;      // 16-byte fetches per lane and K-step: a 512-thread sweep covers 64 rows x 8 chunks
  constexpr int FETCH = A_IT + B_IT;                         // LDS-DMA instructions one wave issues per stage
  constexpr int A_BYTES = BM * BKB, STAGE = (BM + BN) * BKB;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1 && A_IT >= 1 && B_IT >= 1, "bad tile config");
  extern __shared__ __align__(16) char lds[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  const int lid = xcd_remap_w(blockIdx.x, a.tiles_m * a.tiles_n);
  const int m0 = (lid / a.tiles_n) * BM, n0 = (lid % a.tiles_n) * BN;
  const int z = blockIdx.z;
  const int kt0 = z * a.ksteps_per, nk = min(a.ksteps - kt0, a.ksteps_per);
  const long pitchb = a.a_pitch * 2;

  const __amdgpu_buffer_rsrc_t rA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.A), 0, (int)((long)a.N * a.H * a.W * pitchb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.B), 0, (int)((long)a.Nn * a.K * 2), 0x00020000);
  constexpr int OOB = 0x7fffffff;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int aoff[A_IT], ay0[A_IT], ax0[A_IT];
  const bool pointwise = a.kh * a.kw == 1 && a.stride == 1 && a.pad == 0 && a.Ho == a.H && a.Wo == a.W;
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int u = it * NT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
    const int m = m0 + row;
    if (m < a.M && pointwise) {
      ay0[it] = ax0[it] = 0;
      aoff[it] = (int)((long)m * pitchb) + ch * 16;
    } else if (m < a.M) {
      const int img = m / (a.Ho * a.Wo), rem = m - img * (a.Ho * a.Wo);
      const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
      ay0[it] = oy * a.stride - a.pad, ax0[it] = ox * a.stride - a.pad;
      aoff[it] = (int)(((long)(img * a.H + ay0[it]) * a.W + ax0[it]) * pitchb) + ch * 16;
    } else {
      ay0[it] = ax0[it] = -(1 << 28);
      aoff[it] = 0;
    }
  }
  int boff[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int u = it * NT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
    const int n = n0 + row;
    boff[it] = n < a.Nn ? n * a.K * 2 + ch * 16 : OOB;
  }
  const int cpc = a.Cp >> 6;
  int s_tap = kt0 / cpc, s_ch = kt0 - s_tap * cpc, s_ky = s_tap / a.kw, s_kx = s_tap - s_ky * a.kw;
  s_tap = __builtin_amdgcn_readfirstlane(s_tap), s_ch = __builtin_amdgcn_readfirstlane(s_ch);
  s_ky = __builtin_amdgcn_readfirstlane(s_ky), s_kx = __builtin_amdgcn_readfirstlane(s_kx);
  int s_kb = kt0 * BKB;
  // taps no row of this tile can see are neither staged nor multiplied (conv16.hip: the dilated ASPP branches on the 48 x 48 maps)
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
  int nk_eff = nk;
  if (ky_ok != ~0u) {
    nk_eff = 0;
    for (int kt = kt0; kt < kt0 + nk;) {
      const int tap = kt / cpc, run = min(kt0 + nk, (tap + 1) * cpc) - kt;
      if ((ky_ok >> (tap / a.kw)) & 1u) nk_eff += run;
      kt += run;
    }
    nk_eff = __builtin_amdgcn_readfirstlane(nk_eff);
  }
  auto skip_taps = [&]() {
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
      dma16w(rA, la + (it * NT + wave_u * 64) * 16, ok ? aoff[it] + toff : OOB, 0);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) dma16w(rB, lb + (it * NT + wave_u * 64) * 16, boff[it], s_kb);
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

  // this lane's fragment rows and their swizzle keys are K-step invariant: byte offsets inside a stage, one per 16-k group through the XOR
  int ra_off[TM], ra_key[TM], rb_off[TN], rb_key[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ra = wm * (BM / WM) + i * 32 + l31;
    ra_off[i] = ra * BKB, ra_key[i] = (ra >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int rb = wn * (BN / WN) + j * 32 + l31;
    rb_off[j] = A_BYTES + rb * BKB, rb_key[j] = (rb >> 1) & 7;
  }

  auto compute = [&](int buf) {
    const char* ls = lds + buf * STAGE;
    // (Measured and removed: all 16 fragment reads of the K-step issued ahead of the 16 MFMAs behind a sched_barrier -- 194 instead of 126 VGPRs, 9 % slower on
    //  every shape, two alternated runs: 828-833 -> 743-764 TF on the decoder's 3x3s. The compiler's read / MFMA pairing with two waves per SIMD is the better schedule.)
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int c = kg * 2 + half;
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ls + ra_off[i] + ((c ^ ra_key[i]) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(ls + rb_off[j] + ((c ^ rb_key[j]) << 4));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- the ring: stage k + 2 is issued behind the barrier of step k (its buffer was last read in step k - 1, which every wave has left); the wait in front of the
  // barrier leaves the FETCH instructions of stage k + 1 in flight (LDS-DMA fetches of a wave retire in order), so "my part of stage k has landed" + barrier = "stage k
  // has landed". No __syncthreads(): its fence would drain the DMA queue (vmcnt(0)) at every K-step.
  if (nk_eff > 0) {
    if constexpr (NST == 3) {
      stage(0);
      if (nk_eff > 1) stage(1);
      int cur = 0, nxt = 2;
      for (int kt = 0; kt < nk_eff; ++kt) {
        if (kt + 1 < nk_eff) wait_vm<FETCH>();
        else wait_vm<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk_eff) stage(nxt);
        compute(cur);
        cur = cur == NST - 1 ? 0 : cur + 1;
        nxt = nxt == NST - 1 ? 0 : nxt + 1;
      }
    } else {
      stage(0);
      for (int kt = 0; kt < nk_eff; ++kt) {
        wait_vm<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();      // stage kt has landed everywhere, and everybody has left the buffer stage kt + 1 goes into (read in step kt - 1)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk_eff) stage((kt + 1) & 1);
        compute(kt & 1);
      }
    }
  }

  // ---- epilogue: as conv16.hip -- a wave parks 32 rows of its tile in LDS (the ring is dead) and stores whole row segments ----------------------------------
  constexpr int WC = BN / WN, LDC = WC + 4;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float* Ws = reinterpret_cast<float*>(lds) + wave * 32 * LDC;
  const bool slab = a.ksplit > 1;
  if (a.c_f32 || slab) {
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
    const bool cok = col < a.Nn;
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
    }
  }
}

// ---- the PERSISTENT form (round 5, second session) ------------------------------------------------------------------------------------------------------------
// One block per CU walks a sequence of output tiles; four PRODUCER waves do nothing but issue the LDS-DMA fetches of a three-stage ring that runs on across tile
// boundaries (and own the index arithmetic of the A rows); the eight multiplying waves read fragments, issue MFMAs and write the tile out.
// Why (profiles/r05_decoder3x3_ta_tcp_sq_counters.txt): with one 144 KB block per CU nothing hides a tile's prologue -- every block asks for its first 96 KB at the
// same moment and waits ~9 k clocks for them, 18 % of the kernel on the decoder's 3x3 -- and a 1 KB fetch instruction holds its wave's issue for 60-185 clocks beside
// the MFMAs of the same wave. Here the first stages of tile i + 1 are in flight while tile i is multiplied and written out, and a stalled fetch blocks nobody else.
// Synchronisation is ONE s_barrier per K-step for all twelve waves, plus one per tile in front of the epilogue (which parks accumulators in the ring slot that was
// read last -- the producers refill it only behind the NEXT step's barrier):
//   step g:  producers wait until their part of stage g has landed (counted vmcnt: stage g + 1 stays in flight) | barrier | producers issue stage g + 2 into the slot
//            of stage g - 1, consumers multiply stage g.
// Same operands, addressing, tap skipping and epilogue arithmetic as the kernels above; split-K slabs and fp32 outputs included.
__device__ __forceinline__ int tile_ksteps(const pm_conv16& a, int m0, int BM, int kt0, int nk, int cpc, bool pointwise, unsigned& ky_ok) {
  ky_ok = ~0u;
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
  int nk_eff = nk;
  if (ky_ok != ~0u) {
    nk_eff = 0;
    for (int kt = kt0; kt < kt0 + nk;) {
      const int tap = kt / cpc, run = min(kt0 + nk, (tap + 1) * cpc) - kt;
      if ((ky_ok >> (tap / a.kw)) & 1u) nk_eff += run;
      kt += run;
    }
  }
  return __builtin_amdgcn_readfirstlane(nk_eff);
}

// 16-byte stores the compiler's wait bookkeeping does not see. Beside LDS-DMA hipcc puts s_waitcnt vmcnt(0) in front of an LDS read while ANY vector-memory operation of
// the wave is pending: with ordinary stores the first fragment read of the NEXT tile waits until this tile's output has been acknowledged by L2 (1-2 us per tile).
// Nothing reads these addresses again inside the kernel, and a wave's stores complete whatever it does next.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16_untracked(void* p, u32x4 q) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(q) : "memory"); }
__device__ __forceinline__ void st4_untracked(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8_bf16_untracked(pm_bf16* p, const float* v) {
  u32x4 q;
  q.x = pm_pack_bf16(v[0], v[1]), q.y = pm_pack_bf16(v[2], v[3]), q.z = pm_pack_bf16(v[4], v[5]), q.w = pm_pack_bf16(v[6], v[7]);
  st16_untracked(p, q);
}

__device__ __forceinline__ void ring_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int BM, int BN, int WM, int WN, int NP>
__global__ __launch_bounds__((8 + NP) * 64, (8 + NP) / 4) void conv16p_kernel(const pm_conv16 a) {
  static_assert(WM * WN == 8, "eight multiplying waves");
  constexpr int FT = NP * 64;                                // fetching threads
  constexpr int A_IT = BM * 8 / FT, B_IT = BN * 8 / FT;      // 16-byte fetches per producer lane and K-step
  constexpr int FETCH = A_IT + B_IT;
  constexpr int A_BYTES = BM * BKB, STAGE = (BM + BN) * BKB;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int LDS_SUB = 36;                                // floats per row of a 32 x 32 epilogue slab
  static_assert(8 * 32 * LDS_SUB * 4 <= STAGE, "the epilogue slabs of the eight waves live in one ring slot");
  extern __shared__ __align__(16) char lds[];

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ntiles = a.tiles_m * a.tiles_n, total = ntiles * a.ksplit, G = gridDim.x;
  const long pitchb = a.a_pitch * 2;
  const int cpc = a.Cp >> 6;
  const bool pointwise = a.kh * a.kw == 1 && a.stride == 1 && a.pad == 0 && a.Ho == a.H && a.Wo == a.W;
  auto decode = [&](int v, int& m0, int& n0, int& z, int& kt0, int& nk) {
    z = v / ntiles;
    const int lid = xcd_remap_w(v - z * ntiles, ntiles);
    m0 = (lid / a.tiles_n) * BM, n0 = (lid % a.tiles_n) * BN;
    kt0 = z * a.ksteps_per, nk = min(a.ksteps - kt0, a.ksteps_per);
  };

  if (wave >= 8) {
    // ================================================== producer waves ==================================================
    const int wave_u = wave - 8, t = wave_u * 64 + lane;
    const __amdgpu_buffer_rsrc_t rA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.A), 0, (int)((long)a.N * a.H * a.W * pitchb), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.B), 0, (int)((long)a.Nn * a.K * 2), 0x00020000);
    constexpr int OOB = 0x7fffffff;
    int aoff[A_IT], ay0[A_IT], ax0[A_IT], cur[A_IT], boff[B_IT];
    int s_ch = 0, s_ky = 0, s_kx = 0, s_kb = 0, s_fresh = 1, left = 0;
    unsigned ky_ok = ~0u;
    int vf = blockIdx.x;      // the tile being fetched (runs ahead of the tile being multiplied)
    auto skip_taps = [&]() {
      while (s_ky < a.kh && !((ky_ok >> s_ky) & 1u)) {
        s_kb += (cpc - s_ch) * BKB;
        s_ch = 0;
        if (++s_kx == a.kw) s_kx = 0, ++s_ky;
      }
    };
    auto open_tile = [&]() {      // fetch state of tile vf, or of the next one with something to fetch; left == 0 afterwards: no tile is left
      for (; vf < total; vf += G) {
        int m0, n0, z, kt0, nk;
        decode(vf, m0, n0, z, kt0, nk);
        left = tile_ksteps(a, m0, BM, kt0, nk, cpc, pointwise, ky_ok);
        if (left == 0) continue;
        // this lane's rows: piece `it` is row it * FT / 8 + (t >> 3) of the tile. ONE index decomposition (two integer divisions) per tile; the other pieces step on from it
        constexpr int RSTEP = FT / 8;
        int m = m0 + (t >> 3), img = 0, oy = 0, ox = 0;
        if (!pointwise) {
          img = m / (a.Ho * a.Wo);
          const int rem = m - img * (a.Ho * a.Wo);
          oy = rem / a.Wo, ox = rem - oy * a.Wo;
        }
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
          const int row = it * RSTEP + (t >> 3), ch = (t & 7) ^ ((row >> 1) & 7);
          if (m < a.M && pointwise) {
            ay0[it] = ax0[it] = 0;
            aoff[it] = (int)((long)m * pitchb) + ch * 16;
          } else if (m < a.M) {
            ay0[it] = oy * a.stride - a.pad, ax0[it] = ox * a.stride - a.pad;
            aoff[it] = (int)(((long)(img * a.H + ay0[it]) * a.W + ax0[it]) * pitchb) + ch * 16;      // may be negative at the border: only used with an in-range tap added
          } else {
            ay0[it] = ax0[it] = -(1 << 28);      // never inside the image: the row reads zeros
            aoff[it] = 0;
          }
          m += RSTEP;
          if (!pointwise) {
            ox += RSTEP;
            while (ox >= a.Wo) {
              ox -= a.Wo;
              if (++oy == a.Ho) oy = 0, ++img;
            }
          }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
          const int u = it * FT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
          const int n = n0 + row;
          boff[it] = n < a.Nn ? n * a.K * 2 + ch * 16 : OOB;
        }
        const int tap0 = kt0 / cpc;
        s_ch = __builtin_amdgcn_readfirstlane(kt0 - tap0 * cpc);
        s_ky = __builtin_amdgcn_readfirstlane(tap0 / a.kw);
        s_kx = __builtin_amdgcn_readfirstlane(tap0 - s_ky * a.kw);
        s_kb = kt0 * BKB;
        if (ky_ok != ~0u) skip_taps();
        s_fresh = 1;
        return;
      }
      left = 0;
    };
    int slot = 0;      // ring slot of the next stage
    auto issue = [&]() -> int {      // one more stage of the flat (tile, K-step) sequence; 0 when the sequence is over
      if (left == 0) {
        if (vf >= total) return 0;
        vf += G;
        open_tile();
        if (left == 0) return 0;
      }
      char* la = lds + slot * STAGE;
      char* lb = la + A_BYTES;
      if (s_fresh) {      // border test and pixel offset of the tap: once per tap; the 64-channel chunk rides in the scalar offset of the fetch
        const int dy = s_ky * a.dil, dx = s_kx * a.dil;
        const int toff = (dy * a.W + dx) * (int)pitchb;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
          const bool ok = ((unsigned)(ay0[it] + dy) < (unsigned)a.H) & ((unsigned)(ax0[it] + dx) < (unsigned)a.W);
          cur[it] = ok ? aoff[it] + toff : OOB;
        }
        s_fresh = 0;
      }
      const int s_cb = s_ch * BKB;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) dma16w(rA, la + (it * FT + wave_u * 64) * 16, cur[it], s_cb);
#pragma unroll
      for (int it = 0; it < B_IT; ++it) dma16w(rB, lb + (it * FT + wave_u * 64) * 16, boff[it], s_kb);
      s_kb += BKB;
      if (++s_ch == cpc) {
        s_ch = 0;
        if (++s_kx == a.kw) s_kx = 0, ++s_ky;
        if (ky_ok != ~0u) skip_taps();
        s_fresh = 1;
      }
      --left;
      slot = slot == 2 ? 0 : slot + 1;
      return 1;
    };
    open_tile();
    int ahead = issue();      // stages issued and not yet handed over
    ahead += issue();
    for (int vc = blockIdx.x; vc < total; vc += G) {      // mirror of the multiplying waves' barrier sequence
      int m0, n0, z, kt0, nk;
      unsigned dummy;
      decode(vc, m0, n0, z, kt0, nk);
      const int nk_c = tile_ksteps(a, m0, BM, kt0, nk, cpc, pointwise, dummy);
      for (int kt = 0; kt < nk_c; ++kt) {
        if (ahead >= 2) wait_vm<FETCH>();      // the oldest stage in flight has landed, the younger one may still fly
        else wait_vm<0>();
        ring_barrier();
        ahead += issue() - 1;
      }
      ring_barrier();      // the tile's epilogue barrier
    }
    return;
  }

  // ================================================== multiplying waves ==================================================
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  int ra_off[TM], ra_key[TM], rb_off[TN], rb_key[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ra = wm * (BM / WM) + i * 32 + l31;
    ra_off[i] = ra * BKB, ra_key[i] = (ra >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int rb = wn * (BN / WN) + j * 32 + l31;
    rb_off[j] = A_BYTES + rb * BKB, rb_key[j] = (rb >> 1) & 7;
  }
  int rd = 0;      // ring slot of the next stage to multiply
  for (int vc = blockIdx.x; vc < total; vc += G) {
    int m0, n0, z, kt0, nk;
    unsigned dummy;
    decode(vc, m0, n0, z, kt0, nk);
    const int nk_c = tile_ksteps(a, m0, BM, kt0, nk, cpc, pointwise, dummy);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    for (int kt = 0; kt < nk_c; ++kt) {
      ring_barrier();
      const char* ls = lds + rd * STAGE;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const int c = kg * 2 + half;
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ls + ra_off[i] + ((c ^ ra_key[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(ls + rb_off[j] + ((c ^ rb_key[j]) << 4));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      rd = rd == 2 ? 0 : rd + 1;
    }
    // ---- epilogue: 32 x 32 slabs of the wave's tile through the ring slot read last (free until the producers pass the next step's barrier) ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ring_barrier();
    float* Ws = reinterpret_cast<float*>(lds + (rd == 0 ? 2 : rd - 1) * STAGE) + wave * 32 * LDS_SUB;
    const bool slab = a.ksplit > 1;
    if (a.c_f32 || slab) {      // fp32 rows: split-K slabs (no epilogue arithmetic) or class logits (bias only)
      float* Cf = reinterpret_cast<float*>(a.C) + (slab ? (long)z * a.c_split : 0);
      const long cp = slab ? a.Nn : a.c_pitch;
      const int rr0 = lane >> 3, cc = (lane & 7) * 4;      // eight lanes per 32-column row, eight rows per sweep
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int col = n0 + wn * (BN / WN) + n * 32 + cc;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int q = 0; q < 16; ++q) Ws[((q & 3) + 8 * (q >> 2) + 4 * half) * LDS_SUB + l31] = acc[i][n][q];
#pragma unroll
          for (int r0 = 0; r0 < 32; r0 += 8) {
            const int rr = r0 + rr0;
            const long row = m0 + wm * (BM / WM) + i * 32 + rr;
            const float4 v = *reinterpret_cast<const float4*>(Ws + rr * LDS_SUB + cc);
            if (row >= a.M) continue;
            const float e[4] = {v.x, v.y, v.z, v.w};
            if (((cp | a.Nn) & 3) == 0 && col + 4 <= a.Nn && !a.bias) st16_untracked(Cf + row * cp + col, u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)});
            else {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (col + k < a.Nn) st4_untracked(Cf + row * cp + col + k, e[k] + ((a.bias && !slab) ? a.bias[col + k] : 0.f));
            }
          }
        }
      }
      continue;
    }
    {
      const int rr0 = lane >> 2, cc = (lane & 3) * 8;      // four lanes per 32-column row (8 channels = 16 bytes each), sixteen rows per sweep
      const bool aff = a.bias || a.scale, res = a.residual != nullptr, relu = a.relu != 0;
      pm_bf16* C16 = reinterpret_cast<pm_bf16*>(a.C);
      const pm_bf16* R16 = reinterpret_cast<const pm_bf16*>(a.residual);
      // the per-channel constants of BOTH 32-column groups are fetched before the first store of the tile: a register load issued behind a store would wait for it
      float bi[TN][8], sc[TN][8], sh[TN][8];
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int col = n0 + wn * (BN / WN) + n * 32 + cc;
#pragma unroll
        for (int e = 0; e < 8; ++e) bi[n][e] = 0.f, sc[n][e] = 1.f, sh[n][e] = 0.f;
        if (aff && col < a.Nn) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (a.bias) bi[n][e] = a.bias[col + e];
            if (a.scale) sc[n][e] = a.scale[col + e], sh[n][e] = a.shift[col + e];
          }
        }
      }
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int col = n0 + wn * (BN / WN) + n * 32 + cc;
        const bool cok = col < a.Nn;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int q = 0; q < 16; ++q) Ws[((q & 3) + 8 * (q >> 2) + 4 * half) * LDS_SUB + l31] = acc[i][n][q];
#pragma unroll
          for (int r0 = 0; r0 < 32; r0 += 16) {
            const int rr = r0 + rr0;
            const long row = m0 + wm * (BM / WM) + i * 32 + rr;
            const float4 v0 = *reinterpret_cast<const float4*>(Ws + rr * LDS_SUB + cc), v1 = *reinterpret_cast<const float4*>(Ws + rr * LDS_SUB + cc + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (row < a.M && cok) {
              if (aff) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] + bi[n][e]) * sc[n][e] + sh[n][e];
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
              st8_bf16_untracked(C16 + row * a.c_pitch + col, v);
            }
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int NP>
void launch_persistent(const pm_conv16& k, hipStream_t st) {
  constexpr size_t smem = (size_t)3 * (BM + BN) * BKB;
  static_assert(smem <= 160 * 1024, "LDS budget");
  const int ncu = pm_device_once([] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv16p_kernel<BM, BN, WM, WN, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const int total = k.tiles_m * k.tiles_n * k.ksplit;
  hipLaunchKernelGGL((conv16p_kernel<BM, BN, WM, WN, NP>), dim3(std::min(total, ncu)), dim3((8 + NP) * 64), smem, st, k);
}

template <int BM, int BN, int WM, int WN, int NST>
void launch_wide(const pm_conv16& k, dim3 grid, hipStream_t st) {
  constexpr size_t stage_bytes = (size_t)NST * (BM + BN) * BKB, ep_bytes = (size_t)8 * 32 * (BN / WN + 4) * sizeof(float);
  constexpr size_t smem = stage_bytes > ep_bytes ? stage_bytes : ep_bytes;
  static_assert(smem <= 160 * 1024, "LDS budget");
  (void)pm_device_once([] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv16w_kernel<BM, BN, WM, WN, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((conv16w_kernel<BM, BN, WM, WN, NST>), grid, dim3(NT), smem, st, k);
}

}  // namespace

// 1 when pm_conv16w_launch runs this plan in the persistent producer / consumer form (the ring tiles with PM_C16P != 0), 0 for one block per tile
int pm_conv16w_persistent(const pm_conv16* k) { return pm_route.conv16_persistent && k->wide && ((k->bm == 256 && k->bn == 128) || (k->bm == 128 && k->bn == 256)) ? 1 : 0; }

int pm_conv16w_launch(const pm_conv16* k0, hipStream_t st) {
  pm_conv16 k = *k0;
  dim3 grid(k.tiles_m * k.tiles_n, 1, k.ksplit);
  if (k.bm == 256 && k.bn == 128 && pm_route.conv16_persistent) launch_persistent<256, 128, 4, 2, 4>(k, st);
  else if (k.bm == 128 && k.bn == 256 && pm_route.conv16_persistent) launch_persistent<128, 256, 2, 4, 4>(k, st);
  else if (k.bm == 256 && k.bn == 128) launch_wide<256, 128, 4, 2, 3>(k, grid, st);
  else if (k.bm == 128 && k.bn == 256) launch_wide<128, 256, 2, 4, 3>(k, grid, st);
  else if (k.bm == 256 && k.bn == 256) launch_wide<256, 256, 2, 4, 2>(k, grid, st);
  else {
    pm_set_error("conv16w: no %d x %d tile", k.bm, k.bn);
    return PM_EUNSUPPORTED;
  }
  return pm_check_launch("conv16w");
}
