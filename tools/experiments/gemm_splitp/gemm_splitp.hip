// fp32 tier, round 6: the GEMM-shaped split-operand launches -- the batched Winograd point products M[p] = V[p] U[p]^T (winograd.hip) and the large pointwise (1x1, stride 1)
// forward convolutions -- as a PERSISTENT producer / consumer kernel. Same arithmetic as the PREC 5 path of conv_igemm_kernel.h (fp32 operands, every element split exactly
// into three bf16 pieces, six cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulation); what changes is WHO does the split and WHEN:
//   * four PRODUCER waves own the whole global -> registers -> split -> LDS path: they gather 16-byte pieces of the fp32 rows one K-step ahead, split them (4 VALU + 1.5
//     v_perm_b32 per element) and write the three bf16 planes of the step into the free slot of a two-slot LDS ring -- while
//   * eight MULTIPLYING waves (64 x 64 each of a 256 x 128 tile) do nothing but read plane fragments and issue MFMAs, and store the finished tile straight from the
//     accumulators (fire-and-forget stores the compiler's wait bookkeeping does not see).
// One s_barrier per K-step for all twelve waves: at barrier g the producers have written stage g and the multiplying waves have left stage g - 1, whose slot the producers
// then fill with stage g + 1. One block per CU walks (batch point, tile) units; the ring runs on across unit boundaries, so a unit's first stages are in LDS while the
// previous unit is still being written out.
// Why (profiles/r06_split_one_conv_sq_counters.txt): in the tile kernel every wave gathers, splits, writes, waits at the barrier and multiplies in turn; with two
// blocks per CU nothing is saturated (matrix pipe 0.49, VALU issue 0.44, LDS 0.45) -- the dependency chain of a K-step is the bound. Here the chain is cut in two halves that
// run concurrently on the same SIMDs: a SIMD's issue slots carry one producer's ~330 instructions and two multiplying waves' ~80 each per K-step, against 3 072 matrix-pipe
// cycles.
// Replaces nn.Conv2d forward / input gradient of /root/reference/network/Resnet.py:145-150,195 and deepv3plus.py:72-81,397-414 where the planner accepts the shape.
#include <stdlib.h>
#include <algorithm>

#include "pm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v4f_p __attribute__((ext_vector_type(4)));

namespace {

constexpr int ROWB = 208;      // bytes per LDS row and K-step: [hi | mid | lo] x 32 k x 2 B + 16 B pad (13 sixteen-byte granules: conflict-free ds_read_b128)
constexpr int NP = 4;      // producer waves

__device__ __forceinline__ int xcd_remap_s(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void ring_barrier_s() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wait_vm_s() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ float4 bload_s(__amdgpu_buffer_rsrc_t r, int off) {
  const v4f_p v = __builtin_bit_cast(v4f_p, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
// a store the compiler's wait bookkeeping does not see (conv16w.hip): the next unit's first fragment read must not wait for this unit's output
__device__ __forceinline__ void st4_untracked_s(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

// four fp32 -> 3 x four bf16 (truncation split, exact: hi + mid + lo == x), as conv_igemm_kernel.h split4
__device__ __forceinline__ void split4_s(const float4& v, float2& h, float2& m, float2& l) {
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
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__((8 + NP) * 64, (8 + NP) / 4) void gemm_splitp_kernel(const pm_gemm32 a) {
  static_assert(WM * WN == 8, "eight multiplying waves");
  constexpr int A_IT = BM / 32, B_IT = BN / 32;      // 16-byte gathers per producer lane and K-step: lane (g = t & 7, r = t >> 3) owns k-group g of rows r + 32 i
  constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  extern __shared__ __align__(16) char lds[];

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int ntiles = a.tiles_m * a.tiles_n, total = ntiles * a.batch, G = gridDim.x;
  const int nk = a.K >> 5;
  auto decode = [&](int v, int& b, int& m0, int& n0) {
    b = v / ntiles;
    const int lid = xcd_remap_s(v - b * ntiles, ntiles);
    m0 = (lid / a.tiles_n) * BM, n0 = (lid % a.tiles_n) * BN;
  };

  if (wave >= 8) {
    // ================================================== producer waves ==================================================
    const int t = (wave - 8) * 64 + lane, g = t & 7, r = t >> 3;
    constexpr int OOB = 0x7fffffff;
    const int pitchb = (int)a.a_pitch * 4, kb = a.K * 4;
    int aoff[A_IT], boff[B_IT];
    __amdgpu_buffer_rsrc_t rA, rB;
    float4 ra[A_IT], rb[B_IT];
    int s_kb = 0, left = 0;
    int vf = blockIdx.x;
    auto open_unit = [&]() {
      int b, m0, n0;
      decode(vf, b, m0, n0);
      rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A + (long)b * a.a_bs), 0, (int)((long)a.M * pitchb), 0x00020000);
      rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.B + (long)b * a.b_bs), 0, (int)((long)a.Nn * kb), 0x00020000);
#pragma unroll
      for (int i = 0; i < A_IT; ++i) aoff[i] = m0 + r + 32 * i < a.M ? (m0 + r + 32 * i) * pitchb + g * 16 : OOB;
#pragma unroll
      for (int i = 0; i < B_IT; ++i) boff[i] = n0 + r + 32 * i < a.Nn ? (n0 + r + 32 * i) * kb + g * 16 : OOB;
      s_kb = 0, left = nk;
    };
    auto fetch = [&]() -> int {      // the gathers of the next stage of the flat (unit, K-step) sequence into registers; 0 when the sequence is over
      if (left == 0) {
        vf += G;
        if (vf >= total) return 0;
        open_unit();
      }
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[i] = bload_s(rA, (int)((unsigned)aoff[i] + (unsigned)s_kb));
#pragma unroll
      for (int i = 0; i < B_IT; ++i) rb[i] = bload_s(rB, (int)((unsigned)boff[i] + (unsigned)s_kb));
      s_kb += 128;
      --left;
      return 1;
    };
    auto store = [&](int slot) {      // split the gathered rows and write the three planes of the stage
      char* la = lds + slot * STAGE + r * ROWB + g * 8;
      char* lb = la + A_BYTES;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        float2 h, m, l;
        if (a.dbg_nosplit) h = make_float2(ra[i].x, ra[i].y), m = make_float2(ra[i].z, ra[i].w), l = h;      // TIMING EXPERIMENT ONLY (PM_SPLITP_NOSPLIT=1): wrong results
        else split4_s(ra[i], h, m, l);
        char* d = la + i * 32 * ROWB;
        *reinterpret_cast<float2*>(d) = h, *reinterpret_cast<float2*>(d + 64) = m, *reinterpret_cast<float2*>(d + 128) = l;
      }
#pragma unroll
      for (int i = 0; i < B_IT; ++i) {
        float2 h, m, l;
        if (a.dbg_nosplit) h = make_float2(rb[i].x, rb[i].y), m = make_float2(rb[i].z, rb[i].w), l = h;
        else split4_s(rb[i], h, m, l);
        char* d = lb + i * 32 * ROWB;
        *reinterpret_cast<float2*>(d) = h, *reinterpret_cast<float2*>(d + 64) = m, *reinterpret_cast<float2*>(d + 128) = l;
      }
    };
    open_unit();
    int have = fetch();      // stage 0 (every block owns at least one unit: the grid is min(units, CUs))
    store(0);
    have = fetch();          // stage 1 in registers, if there is one
    int wr = 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int vc = blockIdx.x; vc < total; vc += G) {      // mirror of the multiplying waves' barrier sequence: nk steps per unit
      for (int kt = 0; kt < nk; ++kt) {
        ring_barrier_s();      // barrier g: stage g is complete, the slot of stage g - 1 is free
        if (have) {
          store(wr);
          wr ^= 1;
          have = fetch();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    return;
  }

  // ================================================== multiplying waves ==================================================
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  int ra_off[TM], rb_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) ra_off[i] = (wm * (BM / WM) + i * 32 + l31) * ROWB + half * 16;
#pragma unroll
  for (int j = 0; j < TN; ++j) rb_off[j] = A_BYTES + (wn * (BN / WN) + j * 32 + l31) * ROWB + half * 16;
  const bool aff = a.bias || a.scale, relu = a.relu != 0;      // (no residual: pm_gemm_splitp_plan leaves those launches to the tile kernel)
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
      ring_barrier_s();
      const char* ls = lds + rd * STAGE;
      // two 16-k groups per stage; this lane-half's eight k of plane p: 16 bytes at p * 64 + kg * 32 + half * 16 of its row. The fragments of the second group are requested
      // before the MFMAs of the first are issued (two register sets), so that their LDS latency runs under 24 MFMAs instead of in front of the next 24.
      bf16x8 fa[2][3][TM], fb[2][3][TN];
      auto frags = [&](int set, int kg) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[set][p][i] = *reinterpret_cast<const bf16x8*>(ls + ra_off[i] + p * 64 + kg * 32);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[set][p][j] = *reinterpret_cast<const bf16x8*>(ls + rb_off[j] + p * 64 + kg * 32);
        }
      };
#define PM_SP_PROD(S, PA, PB)                                                                                 \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][PA][i], fb[S][PB][j], acc[i][j], 0, 0, 0);
      frags(0, 0);
      frags(1, 1);
      __builtin_amdgcn_sched_barrier(0);      // both groups' requests stay in front of the first group's MFMAs
      PM_SP_PROD(0, 2, 0) PM_SP_PROD(0, 0, 2) PM_SP_PROD(0, 1, 1) PM_SP_PROD(0, 1, 0) PM_SP_PROD(0, 0, 1) PM_SP_PROD(0, 0, 0)
      PM_SP_PROD(1, 2, 0) PM_SP_PROD(1, 0, 2) PM_SP_PROD(1, 1, 1) PM_SP_PROD(1, 1, 0) PM_SP_PROD(1, 0, 1) PM_SP_PROD(1, 0, 0)
#undef PM_SP_PROD
      rd ^= 1;
    }
    // ---- epilogue, straight from the accumulators: this lane's column of each 32 x 32 sub-tile, rows (q & 3) + 8 (q >> 2) + 4 half ----
    float* Cb = a.C + (long)b * a.c_bs;
    float bi[TN], sc[TN], sh[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {      // per-column constants before the first store (a register load issued behind a store would wait for it)
      const int col = n0 + wn * (BN / WN) + j * 32 + l31;
      bi[j] = 0.f, sc[j] = 1.f, sh[j] = 0.f;
      if (aff && col < a.Nn) {      // inline-asm loads with their own wait below: no vector-memory operation the compiler tracks is left in this path
        if (a.bias) asm volatile("global_load_dword %0, %1, off" : "=v"(bi[j]) : "v"(a.bias + col) : "memory");
        if (a.scale) {
          asm volatile("global_load_dword %0, %1, off" : "=v"(sc[j]) : "v"(a.scale + col) : "memory");
          asm volatile("global_load_dword %0, %1, off" : "=v"(sh[j]) : "v"(a.shift + col) : "memory");
        }
      }
    }
    if (aff) {
      wait_vm_s<0>();
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
          st4_untracked_s(Cb + row * a.c_pitch + col, v);
        }
    }
  }
}

template <int BM, int BN, int WM, int WN>
void launch_splitp(const pm_gemm32& k, hipStream_t st) {
  constexpr size_t smem = (size_t)2 * (BM + BN) * ROWB;
  static_assert(smem <= 160 * 1024, "LDS budget");
  const int ncu = pm_device_once([] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_splitp_kernel<BM, BN, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const int total = k.tiles_m * k.tiles_n * k.batch;
  hipLaunchKernelGGL((gemm_splitp_kernel<BM, BN, WM, WN>), dim3(std::min(total, ncu)), dim3((8 + NP) * 64), smem, st, k);
}

}  // namespace

// Shapes the kernel takes: K in whole 32-float steps, 16-byte aligned rows, 32-bit byte offsets per batch point, enough units to fill the chip more than once (a persistent
// block pays nothing for many small units, but a launch with fewer units than CUs has nothing to pipeline). PM_SPLITP: 0 never, 1 by this rule (default), 2 wherever the
// shape can be expressed (kernel tests).
bool pm_gemm_splitp_plan(pm_gemm32* k) {
  static const int mode = getenv("PM_SPLITP") ? atoi(getenv("PM_SPLITP")) : 1;
  if (!mode) return false;
  if (k->residual || k->K % 32 || k->K < 64 || k->a_pitch % 4 || k->c_pitch < k->Nn || k->M < 1 || k->Nn < 1) return false;
  if ((long)k->M * k->a_pitch * 4 >= (1l << 31) || (long)k->Nn * k->K * 4 >= (1l << 31)) return false;
  if (!pm_aligned16(k->A) || !pm_aligned16(k->B) || (k->a_bs | k->b_bs) % 4) return false;
  // 256 x 128, or 128 x 256 when that wastes fewer padded rows / columns
  const double f0 = (double)(pm_cdiv(k->M, 256) * 256) * (pm_cdiv(k->Nn, 128) * 128), f1 = (double)(pm_cdiv(k->M, 128) * 128) * (pm_cdiv(k->Nn, 256) * 256);
  if (f1 < f0) k->bm = 128, k->bn = 256;
  else k->bm = 256, k->bn = 128;
  k->tiles_m = pm_cdiv(k->M, k->bm), k->tiles_n = pm_cdiv(k->Nn, k->bn);
  static const int nosplit = getenv("PM_SPLITP_NOSPLIT") ? atoi(getenv("PM_SPLITP_NOSPLIT")) : 0;
  k->dbg_nosplit = nosplit;
  if (mode >= 2) return true;
  static const int min_units = getenv("PM_SPLITP_MIN_UNITS") ? atoi(getenv("PM_SPLITP_MIN_UNITS")) : 512;
  if (k->Nn < 128 || k->M < 256 || k->K < 128) return false;
  if ((long)k->tiles_m * k->tiles_n * k->batch < min_units) return false;
  return true;
}

int pm_gemm_splitp_launch(const pm_gemm32* k, hipStream_t st) {
  if (k->bm == 256 && k->bn == 128) launch_splitp<256, 128, 4, 2>(*k, st);
  else if (k->bm == 128 && k->bn == 256) launch_splitp<128, 256, 2, 4>(*k, st);
  else {
    pm_set_error("gemm_splitp: no %d x %d tile", k->bm, k->bn);
    return PM_EUNSUPPORTED;
  }
  return pm_check_launch("gemm_splitp");
}
