// LAB KERNEL (round 6): an fp32 GEMM core on the bf16 matrix pipe.
//   C[b][M][N] (fp32) = A[b][M][K] . B[b][N][K]^T, fp32 operands, k-contiguous rows (the pointwise convolutions and the batched Winograd point products of the
//   fp32 tier). The fp32 tiles are staged as they are (LDS-DMA, 128-byte rows = 32 k, source-side XOR swizzle as conv16.hip); every fragment element is split
//   AFTER the LDS read into three bf16 pieces by truncation,
//       hi = x & 0xffff0000,  r = x - hi (exact),  mid = r & 0xffff0000,  lo = r - mid (exact, <= 8 significant bits)   =>   x == hi + mid + lo exactly,
//   and the cross products go to v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator: NPROD = 6 (hh, hm, mh, mm, hl, lh; dropped ml + lm + ll <= 3 * 2^-24
//   relative to |a b|), 9 (all), or 3 (hh, hm, mh: a different precision tier, for reference only).
// What it prints per shape and variant: TFLOP/s counted as 2 M N K (the fp32 work), and the error of sampled outputs against an fp64 dot product, next to the
// error of an fp32 fmaf chain over the same operands (= v_mfma_f32_32x32x2_f32, bit for bit: what the shipped fp32 kernel computes), both relative to sum |a b|.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 split_gemm.hip -o split_gemm ; run: ./split_gemm [variant filter]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                        \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int BKB = 128;      // bytes per row and K-step (32 fp32)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

struct Split8 {
  bf16x8 hi, mid, lo;
};
// eight fp32 -> three bf16x8 by truncation: 4 VALU per element (and, sub, and, sub) + 1.5 v_perm_b32 per element pair
template <int NPROD>
__device__ __forceinline__ Split8 split8(const float4& p, const float4& q) {
  const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
  u32x4 h, m, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned u0 = __float_as_uint(x[2 * j]), u1 = __float_as_uint(x[2 * j + 1]);
    h[j] = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x[2 * j] - __uint_as_float(u0 & 0xffff0000u), r1 = x[2 * j + 1] - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m[j] = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    if constexpr (NPROD > 3) {
      const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
      l[j] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
    } else {
      l[j] = 0;
    }
  }
  Split8 s;
  s.hi = __builtin_bit_cast(bf16x8, h), s.mid = __builtin_bit_cast(bf16x8, m), s.lo = __builtin_bit_cast(bf16x8, l);
  return s;
}

// NST LDS stages of (BM + BN) x 128 B; WM x WN waves, each a (BM / WM) x (BN / WN) tile of 32 x 32 MFMA blocks; every wave issues its share of the LDS-DMA fetches.
template <int BM, int BN, int WM, int WN, int NST, int NPROD, int MINB>
__global__ __launch_bounds__(WM* WN * 64, MINB) void split_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N,
                                                                      int K, long a_bs, long b_bs, long c_bs, int tiles_n) {
  constexpr int NT = WM * WN * 64;
  constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
  constexpr int FETCH = A_IT + B_IT;
  constexpr int A_BYTES = BM * BKB, STAGE = (BM + BN) * BKB;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1 && A_IT >= 1 && B_IT >= 1, "bad tile config");
  extern __shared__ __align__(16) char lds[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  const int by = blockIdx.y;
  const int nk = K / 32;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + by * a_bs), 0, (int)((long)M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B + by * b_bs), 0, (int)((long)N * K * 4), 0x00020000);
  constexpr int OOB = 0x7fffffff;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int aoff[A_IT], boff[B_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int u = it * NT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
    aoff[it] = m0 + row < M ? (m0 + row) * K * 4 + ch * 16 : OOB;
  }
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int u = it * NT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
    boff[it] = n0 + row < N ? (n0 + row) * K * 4 + ch * 16 : OOB;
  }
  int s_kb = 0;
  auto stage = [&](int buf) {
    char* la = lds + buf * STAGE;
    char* lb = la + A_BYTES;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) dma16(rA, la + (it * NT + wave_u * 64) * 16, aoff[it], s_kb);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) dma16(rB, lb + (it * NT + wave_u * 64) * 16, boff[it], s_kb);
    s_kb += BKB;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

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
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {      // two 16-k groups per 128-byte row; this lane-half's eight k = chunks c0, c0 + 1
      const int c0 = kg * 4 + half * 2;
      Split8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = split8<NPROD>(*reinterpret_cast<const float4*>(ls + ra_off[i] + ((c0 ^ ra_key[i]) << 4)), *reinterpret_cast<const float4*>(ls + ra_off[i] + (((c0 + 1) ^ ra_key[i]) << 4)));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = split8<NPROD>(*reinterpret_cast<const float4*>(ls + rb_off[j] + ((c0 ^ rb_key[j]) << 4)), *reinterpret_cast<const float4*>(ls + rb_off[j] + (((c0 + 1) ^ rb_key[j]) << 4)));
#define PM_PROD(X, Y)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = \
      __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i].X, fb[j].Y, acc[i][j], 0, 0, 0);
      if constexpr (NPROD == 9) {
        PM_PROD(lo, lo) PM_PROD(lo, mid) PM_PROD(mid, lo)
      }
      if constexpr (NPROD >= 6) {
        PM_PROD(lo, hi) PM_PROD(hi, lo) PM_PROD(mid, mid)
      }
      PM_PROD(mid, hi) PM_PROD(hi, mid) PM_PROD(hi, hi)
#undef PM_PROD
    }
  };

  if (nk > 0) {
    if constexpr (NST >= 3) {
      stage(0);
      if (nk > 1) stage(1);
      int cur = 0, nxt = 2;
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vm<FETCH>();
        else wait_vm<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) stage(nxt);
        compute(cur);
        cur = cur == NST - 1 ? 0 : cur + 1;
        nxt = nxt == NST - 1 ? 0 : nxt + 1;
      }
    } else {
      stage(0);
      for (int kt = 0; kt < nk; ++kt) {
        wait_vm<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage((kt + 1) & 1);
        compute(kt & 1);
      }
    }
  }

  // epilogue: a wave parks 32 x 32 slabs in LDS (the ring is dead) and stores 16-byte row segments
  constexpr int LDS_SUB = 36;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  float* Ws = reinterpret_cast<float*>(lds) + wave * 32 * LDS_SUB;
  float* Cf = C + by * c_bs;
  const int rr0 = lane >> 3, cc = (lane & 7) * 4;
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
        if (row < M && col + 4 <= N) *reinterpret_cast<float4*>(Cf + row * N + col) = v;
      }
    }
  }
}

struct Shape {
  const char* name;
  int batch, M, N, K;
};

template <int BM, int BN, int WM, int WN, int NST, int NPROD, int MINB>
float run(const Shape& s, const float* A, const float* B, float* C, int reps) {
  constexpr size_t smem = (size_t)NST * (BM + BN) * BKB;
  static_assert(smem <= 160 * 1024, "LDS");
  auto kern = split_gemm_kernel<BM, BN, WM, WN, NST, NPROD, MINB>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int tiles_m = (s.M + BM - 1) / BM, tiles_n = (s.N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n, s.batch);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, 0, A, B, C, s.M, s.N, s.K, (long)s.M * s.K, (long)s.N * s.K, (long)s.M * s.N, tiles_n);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, 0, A, B, C, s.M, s.N, s.K, (long)s.M * s.K, (long)s.N * s.K, (long)s.M * s.N, tiles_n);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

struct Err {
  double max_rel, rms_rel, mean_signed;
};

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : "";
  const Shape shapes[] = {
      {"wino 36x1152x256x256 (layer3 conv2 point GEMM)", 36, 1152, 256, 256},
      {"wino 36x18432x256x256 (final1.3 point GEMM)", 36, 18432, 256, 256},
      {"wino 36x1152x256x2048 (ASPP d6 point GEMM)", 36, 1152, 256, 2048},
      {"1x1 18432x2048x512 (layer4 conv3)", 1, 18432, 2048, 512},
      {"1x1 18432x512x2048 (layer4 conv1)", 1, 18432, 512, 2048},
      {"1x1 294912x256x64 (layer1 conv3)", 1, 294912, 256, 64},
      {"1x1 73728x512x128 (layer2 conv3)", 1, 73728, 512, 128},
      {"1x1 18432x1024x256 (layer3 conv3)", 1, 18432, 1024, 256},
  };
  size_t maxA = 0, maxB = 0, maxC = 0;
  for (const Shape& s : shapes) {
    maxA = std::max(maxA, (size_t)s.batch * s.M * s.K);
    maxB = std::max(maxB, (size_t)s.batch * s.N * s.K);
    maxC = std::max(maxC, (size_t)s.batch * s.M * s.N);
  }
  std::vector<float> hA(maxA), hB(maxB), hC(maxC);
  uint64_t seed = 0x9e3779b97f4a7c15ull;
  auto rnd = [&]() {
    seed ^= seed << 13, seed ^= seed >> 7, seed ^= seed << 17;
    return (float)((double)(seed >> 11) / 9007199254740992.0 * 2.0 - 1.0);
  };
  // activations: post-ReLU-like (half zeros, the rest half-normal-ish); weights: signed
  for (size_t i = 0; i < maxA; ++i) {
    const float v = rnd();
    hA[i] = v > 0.f ? v * 1.7f * (1.f + rnd()) : 0.f;
  }
  for (size_t i = 0; i < maxB; ++i) hB[i] = rnd() * 0.05f;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, maxA * 4));
  CK(hipMalloc(&dB, maxB * 4));
  CK(hipMalloc(&dC, maxC * 4));
  CK(hipMemcpy(dA, hA.data(), maxA * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), maxB * 4, hipMemcpyHostToDevice));

  auto check = [&](const Shape& s, Err& e_kernel, Err& e_f32) {
    CK(hipMemcpy(hC.data(), dC, (size_t)s.batch * s.M * s.N * 4, hipMemcpyDeviceToHost));
    uint64_t sd = 12345;
    auto ri = [&](int n) {
      sd = sd * 6364136223846793005ull + 1442695040888963407ull;
      return (int)((sd >> 33) % (uint64_t)n);
    };
    const int S = 3000;
    double mk = 0, sk = 0, bk = 0, mf = 0, sf = 0, bf = 0;
    for (int i = 0; i < S; ++i) {
      const int b = ri(s.batch), m = ri(s.M), n = ri(s.N);
      const float* a = hA.data() + ((size_t)b * s.M + m) * s.K;
      const float* w = hB.data() + ((size_t)b * s.N + n) * s.K;
      double ref = 0, mag = 0;
      float chain = 0.f;
      for (int k = 0; k < s.K; ++k) {
        ref += (double)a[k] * (double)w[k];
        mag += fabs((double)a[k] * (double)w[k]);
        chain = fmaf(a[k], w[k], chain);
      }
      if (mag == 0) continue;
      const double got = hC[((size_t)b * s.M + m) * s.N + n];
      const double ek = (got - ref) / mag, ef = ((double)chain - ref) / mag;
      mk = std::max(mk, fabs(ek)), sk += ek * ek, bk += ek;
      mf = std::max(mf, fabs(ef)), sf += ef * ef, bf += ef;
    }
    e_kernel = {mk, sqrt(sk / S), bk / S};
    e_f32 = {mf, sqrt(sf / S), bf / S};
  };

#define VARIANT(NAME, ...)                                                                                                                   \
  if (strstr(NAME, filter)) {                                                                                                                \
    CK(hipMemset(dC, 0xff, (size_t)s.batch * s.M * s.N * 4));                                                                                \
    const float ms = run<__VA_ARGS__>(s, dA, dB, dC, 20);                                                                                    \
    Err ek, ef;                                                                                                                              \
    check(s, ek, ef);                                                                                                                        \
    printf("  %-34s %8.3f ms %7.1f TF   err/sum|ab|: max %.2e rms %.2e bias %+.1e   (fp32 fma chain: max %.2e rms %.2e bias %+.1e)\n", NAME, ms, \
           2.0 * s.batch * s.M * s.N * s.K / ms * 1e-9, ek.max_rel, ek.rms_rel, ek.mean_signed, ef.max_rel, ef.rms_rel, ef.mean_signed);    \
    fflush(stdout);                                                                                                                          \
  }

  for (const Shape& s : shapes) {
    printf("%s\n", s.name);
    //        name                                  BM   BN  WM WN NST NPROD MINB
    VARIANT("128x128 4w(64x64) 2st p6 x2", 128, 128, 2, 2, 2, 6, 2)
    VARIANT("128x128 4w(64x64) 2st p9 x2", 128, 128, 2, 2, 2, 9, 2)
    VARIANT("128x128 4w(64x64) 2st p3 x2", 128, 128, 2, 2, 2, 3, 2)
    VARIANT("256x128 8w(64x64) 3st p6", 256, 128, 4, 2, 3, 6, 1)
    VARIANT("256x256 8w(128x64) 2st p6", 256, 256, 2, 4, 2, 6, 1)
    VARIANT("256x256 8w(128x64) 2st p9", 256, 256, 2, 4, 2, 9, 1)
    VARIANT("256x256 4w(128x128) 2st p6", 256, 256, 2, 2, 2, 6, 1)
    VARIANT("256x128 4w(128x64) 3st p6", 256, 128, 2, 2, 3, 6, 1)
    VARIANT("128x256 4w(64x128) 3st p6", 128, 256, 2, 2, 3, 6, 1)
  }
  return 0;
}
