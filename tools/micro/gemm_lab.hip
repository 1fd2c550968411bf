// GEMM laboratory (GPU box): C[b][M][N] = A[b][M][K] * B[b][N][K]^T, fp32 MFMA 32x32x2, the pointwise fast path of conv_igemm.hip
// (every 1x1 convolution and every batched Winograd GEMM) rebuilt with compile-time knobs so that the cost of each phase can be
// isolated on the real shapes:  hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_lab.hip -o gemm_lab && ./gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <string>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float4 bload(__amdgpu_buffer_rsrc_t r, int off) {
  const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// knobs
//  BM, BN      block tile (4 waves as 2 x 2, each wave (BM/2) x (BN/2))
//  BKK         K-slab per LDS stage (32 / 64)
//  NST         LDS stages (1: two barriers per slab, 2: one)
//  PERSIST     0: one tile per block; 1: persistent blocks, next tile's first slab prefetched under the epilogue
//  ABL         ablation: 0 full | 1 no global loads | 2 no epilogue stores | 3 no loads + no stores | 4 no MFMA
template <int BM, int BN, int BKK, int NST, int PERSIST, int ABL, int MINW, int TR = 0, int PF2 = 0, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, MINW) void gemm_lab(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                                      long a_bs, long b_bs, long c_bs, int tiles_m, int tiles_n, int batch) {
  constexpr int LDK = BKK + 4;
  constexpr int KG = BKK / 4;          // float4 groups per row
  constexpr int NT = WAVES * 64;
  constexpr int RP = NT / KG;          // rows per pass
  constexpr int A_N = BM / RP, B_N = BN / RP;
  constexpr int WSPLIT = WAVES == 4 ? 2 : 1;   // waves per tile dimension
  constexpr int TM = BM / WSPLIT / 32, TN = BN / WSPLIT / 32;
  constexpr int A_FLOATS = BM * LDK, B_FLOATS = BN * LDK, STAGE = A_FLOATS + B_FLOATS;
  extern __shared__ __align__(16) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = WAVES == 4 ? wave >> 1 : 0, wn = WAVES == 4 ? wave & 1 : 0, half = lane >> 5, l31 = lane & 31;
  const int g = t % KG, r = t / KG;
  const int ntile = tiles_m * tiles_n;
  const int total = ntile * batch;
  float4 ra[A_N], rb[B_N], ra2[A_N], rb2[B_N];
  f32x16 acc[TM][TN];

  auto tile_of = [&](int w, int& m0, int& n0, int& by) {
    by = w / ntile;
    const int lid = PERSIST ? (w - by * ntile) : xcd_remap(w - by * ntile, ntile);
    m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  };
  auto issue_into = [&](float4 (&ra)[A_N], float4 (&rb)[B_N], int m0, int n0, int by, int k0) {
    if (ABL == 1 || ABL == 3) return;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + by * a_bs), 0, M * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B + by * b_bs), 0, N * K * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_N; ++i) {
      const int m = m0 + r + RP * i;
      ra[i] = bload(rA, m < M ? (m * K + k0 + g * 4) * 4 : 0x7fffffff);
    }
#pragma unroll
    for (int i = 0; i < B_N; ++i) {
      const int n = n0 + r + RP * i;
      rb[i] = bload(rB, n < N ? (n * K + k0 + g * 4) * 4 : 0x7fffffff);
    }
  };
  auto issue = [&](int m0, int n0, int by, int k0) { issue_into(ra, rb, m0, n0, by, k0); };
  auto stash_from = [&](float4 (&ra)[A_N], float4 (&rb)[B_N], int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + A_FLOATS;
#pragma unroll
    for (int i = 0; i < A_N; ++i) *reinterpret_cast<float4*>(As + (r + RP * i) * LDK + g * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_N; ++i) *reinterpret_cast<float4*>(Bs + (r + RP * i) * LDK + g * 4) = rb[i];
  };
  auto stash = [&](int buf) { stash_from(ra, rb, buf); };
  auto compute = [&](int buf) {
    const float* As = smem + buf * STAGE;
    const float* Bs = As + A_FLOATS;
#pragma unroll
    for (int kg = 0; kg < BKK / 8; ++kg) {
      const int kk = kg * 8 + half * 4;
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(As + (wm * (BM / WSPLIT) + i * 32 + l31) * LDK + kk);
#pragma unroll
      for (int i = 0; i < TN; ++i) fb[i] = *reinterpret_cast<const float4*>(Bs + (wn * (BN / WSPLIT) + i * 32 + l31) * LDK + kk);
      if (ABL == 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i].x), "v"(fa[i].y), "v"(fa[i].z), "v"(fa[i].w));
#pragma unroll
        for (int i = 0; i < TN; ++i) asm volatile("" ::"v"(fb[i].x), "v"(fb[i].y), "v"(fb[i].z), "v"(fb[i].w));
        continue;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n) {
            const float a = j == 0 ? fa[i].x : (j == 1 ? fa[i].y : (j == 2 ? fa[i].z : fa[i].w));
            const float b = j == 0 ? fb[n].x : (j == 1 ? fb[n].y : (j == 2 ? fb[n].z : fb[n].w));
            acc[i][n] = TR == 1 ? __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[i][n], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][n], 0, 0, 0);
          }
    }
  };
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  };
  auto epilogue = [&](int m0, int n0, int by) {
    float* Cb = C + by * c_bs;
    if (TR == 2) {   // stage each wave's 32-row slab through LDS and store whole rows: 4 rows x 256 B per instruction (BN/2 = 64 columns)
      constexpr int WC = BN / WSPLIT, LDC = WC + 4;
      float* Ws = smem + wave * 32 * LDC;           // wave-private; the A / B stages are dead (caller synchronised)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int q = 0; q < 16; ++q) Ws[((q & 3) + 8 * (q >> 2) + 4 * half) * LDC + n * 32 + l31] = acc[i][n][q];
        // 64 lanes -> WC/4 lanes per row
        constexpr int LPR = WC / 4, RPI = 64 / LPR;   // lanes per row, rows per instruction
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += RPI) {
          const int rr = r0 + lane / LPR, cc = (lane % LPR) * 4;
          const float4 v = *reinterpret_cast<const float4*>(Ws + rr * LDC + cc);
          const int row = m0 + wm * (BM / WSPLIT) + i * 32 + rr, col = n0 + wn * WC + cc;
          if (ABL == 2 || ABL == 3) asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
          else if (row < M && col < N) *reinterpret_cast<float4*>(Cb + (long)row * N + col) = v;
        }
      }
      return;
    }
    if (TR == 1) {   // D' = (C tile)^T: lane l31 = row m, registers 4g..4g+3 = columns 8g + 4 half + (0..3): one 16-byte store each
      const int row0 = m0 + wm * (BM / WSPLIT) + l31, col0 = n0 + wn * (BN / WSPLIT) + 4 * half;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int row = row0 + i * 32, col = col0 + n * 32 + 8 * g4;
            const float4 v = make_float4(acc[i][n][4 * g4], acc[i][n][4 * g4 + 1], acc[i][n][4 * g4 + 2], acc[i][n][4 * g4 + 3]);
            if (ABL == 2 || ABL == 3) asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            else if (row < M && col < N) *reinterpret_cast<float4*>(Cb + (long)row * N + col) = v;
          }
      return;
    }
    const int rbase = m0 + wm * (BM / WSPLIT) + 4 * half, cbase = n0 + wn * (BN / WSPLIT) + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int col = cbase + n * 32;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = rbase + i * 32 + (q & 3) + 8 * (q >> 2);
          if (ABL == 2 || ABL == 3) asm volatile("" ::"v"(acc[i][n][q]));
          else if (row < M && col < N) Cb[(long)row * N + col] = acc[i][n][q];
        }
      }
  };

  // PF2 >= 2: raw s_barrier instead of __syncthreads(): the compiler's workgroup barrier carries s_waitcnt vmcnt(0), which drains the loads that
  // were issued two slabs ahead at EVERY barrier (so the deeper prefetch of PF2 == 1 never spans one); here only LDS traffic is waited for
  auto bar_reads_done = [&]() {
    if (PF2 >= 2) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } else __syncthreads();
  };
  auto bar_writes_visible = [&]() {
    if (PF2 >= 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } else __syncthreads();
  };
  const int nk = K / BKK;
  if (!PERSIST) {
    int m0, n0, by;
    tile_of(blockIdx.x, m0, n0, by);
    zero();
    issue(m0, n0, by, 0);
    stash(0);
    __syncthreads();
    if (NST == 1 && (PF2 == 1 || PF2 == 2)) {      // loads run two slabs ahead of the MFMAs (two register sets, loop unrolled by two)
      if (nk > 1) issue_into(ra2, rb2, m0, n0, by, BKK);
      int kt = 0;
      for (; kt + 2 < nk; kt += 2) {
        issue_into(ra, rb, m0, n0, by, (kt + 2) * BKK);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        bar_reads_done();
        stash_from(ra2, rb2, 0);
        bar_writes_visible();
        if (kt + 3 < nk) issue_into(ra2, rb2, m0, n0, by, (kt + 3) * BKK);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        bar_reads_done();
        stash_from(ra, rb, 0);
        bar_writes_visible();
      }
      if (kt + 1 < nk) {       // nk even: slab nk-1 is still in ra2
        compute(0);
        bar_reads_done();
        stash_from(ra2, rb2, 0);
        bar_writes_visible();
      }
      compute(0);
    } else if (NST == 1) {
      for (int kt = 0; kt < nk - 1; ++kt) {
        issue(m0, n0, by, (kt + 1) * BKK);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        bar_reads_done();
        stash(0);
        bar_writes_visible();
      }
      compute(0);
    } else {
      for (int kt = 0; kt < nk - 1; ++kt) {
        issue(m0, n0, by, (kt + 1) * BKK);
        __builtin_amdgcn_sched_barrier(0);
        compute(kt & 1);
        stash((kt + 1) & 1);
        __syncthreads();
      }
      compute((nk - 1) & 1);
    }
    if (TR == 2) __syncthreads();
    epilogue(m0, n0, by);
  } else {
    // persistent: block b walks tiles b, b + G, ...; consecutive blocks take consecutive tiles (same A row panel -> same XCD L2 is lost,
    // the tile order is XCD-strided instead); the first slab of the next tile is in flight while the epilogue stores
    int w = blockIdx.x;
    if (w >= total) return;
    int m0, n0, by;
    tile_of(w, m0, n0, by);
    issue(m0, n0, by, 0);
    while (true) {
      zero();
      stash(0);
      __syncthreads();
      for (int kt = 0; kt < nk - 1; ++kt) {
        issue(m0, n0, by, (kt + 1) * BKK);
        __builtin_amdgcn_sched_barrier(0);
        compute(NST == 1 ? 0 : (kt & 1));
        if (NST == 1) __syncthreads();
        stash(NST == 1 ? 0 : ((kt + 1) & 1));
        __syncthreads();
      }
      const int wn_ = w + gridDim.x;
      int m1 = 0, n1 = 0, b1 = 0;
      if (wn_ < total) {
        tile_of(wn_, m1, n1, b1);
        issue(m1, n1, b1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      compute(NST == 1 ? 0 : ((nk - 1) & 1));
      epilogue(m0, n0, by);
      if (wn_ >= total) break;
      __syncthreads();      // everyone is done reading the last slab before the next tile's first stash
      w = wn_, m0 = m1, n0 = n1, by = b1;
    }
  }
}

struct Shape { const char* name; int M, N, K, batch; };

template <int BM, int BN, int BKK, int NST, int PERSIST, int ABL, int MINW, int TR = 0, int PF2 = 0, int WAVES = 4>
double run(const Shape& s, const float* A, const float* B, float* C, int per_cu, int iters = 20, size_t extra_lds = 0) {
  const int tiles_m = (s.M + BM - 1) / BM, tiles_n = (s.N + BN - 1) / BN;
  const int total = tiles_m * tiles_n * s.batch;
  const size_t smem = (size_t)NST * (BM + BN) * (BKK + 4) * sizeof(float) + extra_lds;
  auto kern = gemm_lab<BM, BN, BKK, NST, PERSIST, ABL, MINW, TR, PF2, WAVES>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int grid = PERSIST ? std::min(total, 256 * per_cu) : total;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), smem, 0, A, B, C, s.M, s.N, s.K, (long)s.M * s.K, (long)s.N * s.K, (long)s.M * s.N, tiles_m, tiles_n, s.batch);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), smem, 0, A, B, C, s.M, s.N, s.K, (long)s.M * s.K, (long)s.N * s.K, (long)s.M * s.N, tiles_m, tiles_n, s.batch);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / iters * 1e3;   // us
}


// ---- direct-to-LDS ring: global -> LDS by buffer_load ... lds (no VGPR round trip, no ds_write phase), STAGES buffers, counted vmcnt so
// that the loads of the next STAGES - 1 slabs stay in flight across the one barrier per slab. LDS rows are 128 B (32 floats), unpadded;
// bank conflicts of the b128 fragment reads are removed by an XOR swizzle applied on the SOURCE side (the DMA writes lane-linear).
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef __attribute__((address_space(3))) void lds_void;
template <int BM, int BN, int STAGES, int EPS>
__global__ __launch_bounds__(256, 1) void gemm_glds(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                                    long a_bs, long b_bs, long c_bs, int tiles_m, int tiles_n, int batch) {
#if defined(__HIP_DEVICE_COMPILE__)      // the LDS-DMA builtin and the address-space casts exist on the device side only
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int STAGE_B = (BM + BN) * 128;                 // bytes per stage
  constexpr int AI = BM / 32, BI = BN / 32;                // DMA instructions per wave and stage (8 rows each)
  extern __shared__ __align__(16) float smem[];
  char* sb = reinterpret_cast<char*>(smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  const int ntile = tiles_m * tiles_n;
  const int by = blockIdx.x / ntile;
  const int lid = xcd_remap(blockIdx.x - by * ntile, ntile);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + by * a_bs), 0, M * K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B + by * b_bs), 0, N * K * 4, 0x00020000);
  // per-lane source offsets of this wave's DMA pieces (row R of the tile, swizzled 16-byte chunk)
  int offA[AI], offB[BI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int R = wave * (BM / 4) + i * 8 + (lane >> 3), kc = (lane & 7) ^ ((R >> 1) & 7);
    offA[i] = m0 + R < M ? ((m0 + R) * K) * 4 + kc * 16 : 0x7fffffff;
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int R = wave * (BN / 4) + i * 8 + (lane >> 3), kc = (lane & 7) ^ ((R >> 1) & 7);
    offB[i] = n0 + R < N ? ((n0 + R) * K) * 4 + kc * 16 : 0x7fffffff;
  }
  // The DMA is issued from inline asm: hipcc does not track it, so it neither drains vmcnt before the next ds_read (it does for the builtin:
  // every K-step waits for ALL loads in flight) nor reorders it; the counted waits below are the only ordering.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sb;
  const unsigned waveA = __builtin_amdgcn_readfirstlane(lds0 + wave * (BM / 4) * 128), waveB = __builtin_amdgcn_readfirstlane(lds0 + BM * 128 + wave * (BN / 4) * 128);
  auto dma16 = [&](__amdgpu_buffer_rsrc_t r, unsigned lds_addr, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(r), "s"(lds_addr) : "memory");
  };
  auto issue = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) dma16(rA, waveA + buf * STAGE_B + i * 8 * 128, offA[i] == 0x7fffffff ? 0x7fffffff : offA[i] + k0 * 4);
#pragma unroll
    for (int i = 0; i < BI; ++i) dma16(rB, waveB + buf * STAGE_B + i * 8 * 128, offB[i] == 0x7fffffff ? 0x7fffffff : offB[i] + k0 * 4);
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  auto compute = [&](int buf) {
    const char* As = sb + buf * STAGE_B;
    const char* Bs = As + BM * 128;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int kc = kg * 2 + half;
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / 2) + i * 32 + l31;
        fa[i] = *reinterpret_cast<const float4*>(As + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * (BN / 2) + i * 32 + l31;
        fb[i] = *reinterpret_cast<const float4*>(Bs + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int n = 0; n < TN; ++n) {
            const float a = j == 0 ? fa[i].x : (j == 1 ? fa[i].y : (j == 2 ? fa[i].z : fa[i].w));
            const float b = j == 0 ? fb[n].x : (j == 1 ? fb[n].y : (j == 2 ? fb[n].z : fb[n].w));
            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][n], 0, 0, 0);
          }
    }
  };
  const int nk = K / 32;
  constexpr int PER = AI + BI;                             // DMA instructions per wave and stage
  // prologue: STAGES - 1 slabs in flight
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s * 32);
  for (int kt = 0; kt < nk; ++kt) {
    // slab kt has landed once at most the later (STAGES - 2) slabs of this wave are outstanding
    if (kt + STAGES - 2 < nk) wait_vm<PER*(STAGES - 2)>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                          // every wave's pieces of slab kt are visible; slab kt - 1 is no longer read
    __builtin_amdgcn_sched_barrier(0);
    if (kt + STAGES - 1 < nk) issue((kt + STAGES - 1) % STAGES, (kt + STAGES - 1) * 32);
    compute(kt % STAGES);
    __builtin_amdgcn_sched_barrier(0);
  }
  float* Cb = C + by * c_bs;
  if (EPS) {   // staged row stores
    constexpr int WC = BN / 2, LDC = WC + 4, LPR = WC / 4, RPI = 64 / LPR;
    __syncthreads();
    float* Ws = smem + wave * 32 * LDC;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) Ws[((q & 3) + 8 * (q >> 2) + 4 * half) * LDC + n * 32 + l31] = acc[i][n][q];
#pragma unroll
      for (int r0 = 0; r0 < 32; r0 += RPI) {
        const int rr = r0 + lane / LPR, cc = (lane % LPR) * 4;
        const float4 v = *reinterpret_cast<const float4*>(Ws + rr * LDC + cc);
        const int row = m0 + wm * (BM / 2) + i * 32 + rr, col = n0 + wn * WC + cc;
        if (row < M && col < N) *reinterpret_cast<float4*>(Cb + (long)row * N + col) = v;
      }
    }
  } else {
    const int rbase = m0 + wm * (BM / 2) + 4 * half, cbase = n0 + wn * (BN / 2) + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = rbase + i * 32 + (q & 3) + 8 * (q >> 2), col = cbase + n * 32;
          if (row < M && col < N) Cb[(long)row * N + col] = acc[i][n][q];
        }
  }
#endif
}

template <int BM, int BN, int STAGES, int EPS>
double run_glds(const Shape& s, const float* A, const float* B, float* C, int iters = 20) {
  const int tiles_m = (s.M + BM - 1) / BM, tiles_n = (s.N + BN - 1) / BN;
  const int total = tiles_m * tiles_n * s.batch;
  const size_t smem = std::max<size_t>((size_t)STAGES * (BM + BN) * 128, (size_t)4 * 32 * (BN / 2 + 4) * 4);
  auto kern = gemm_glds<BM, BN, STAGES, EPS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL(kern, dim3(total), dim3(256), smem, 0, A, B, C, s.M, s.N, s.K, (long)s.M * s.K, (long)s.N * s.K, (long)s.M * s.N, tiles_m, tiles_n, s.batch);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL(kern, dim3(total), dim3(256), smem, 0, A, B, C, s.M, s.N, s.K, (long)s.M * s.K, (long)s.N * s.K, (long)s.M * s.N, tiles_m, tiles_n, s.batch);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / iters * 1e3;
}


// ---- wave-specialised variant: 4 consumer waves (LDS fragment reads + MFMA + epilogue only) and 4 producer waves (global loads, LDS writes
// only) per block, a ring of STAGES K-slabs in LDS, hand-off through two LDS counters per stage (full / empty, one ds_add per wave and use),
// persistent blocks: the producers run ahead into the next tile while the consumers store the finished one. Every poll loop is bounded and
// reports through *err instead of hanging the GPU.
__device__ __forceinline__ bool ws_wait(volatile unsigned* flag, unsigned target, int* err) {
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if ((int)(*flag - target) >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  *err = 1;
  return false;
}
template <int BM, int BN, int STAGES, int EPS, int MINB>
__global__ __launch_bounds__(512, MINB) void gemm_ws(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, long a_bs,
                                                  long b_bs, long c_bs, int tiles_m, int tiles_n, int batch, int* err) {
  constexpr int BKK = 32, LDK = BKK + 4, KG = BKK / 4;
  constexpr int RP = 256 / KG;                       // producer threads: rows per pass
  constexpr int A_N = BM / RP, B_N = BN / RP;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_FLOATS = BM * LDK, B_FLOATS = BN * LDK, STAGE = A_FLOATS + B_FLOATS;
  extern __shared__ __align__(16) float smem[];
  unsigned* flags = reinterpret_cast<unsigned*>(smem + STAGES * STAGE);      // full[STAGES], empty[STAGES]
  float* stage_c = smem + STAGES * STAGE + 16;                                // EPS: per consumer wave 32 x (BN/2 + 4) floats (flags: 16 words)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < 2 * STAGES) flags[t] = 0;
  __syncthreads();
  volatile unsigned* full = flags;
  volatile unsigned* empty = flags + STAGES;
  const int ntile = tiles_m * tiles_n, total = ntile * batch;
  const int nk = K / BKK;
  unsigned it = 0;                                   // K-slabs handled so far by this block (ring position), same sequence in every wave
  if (wave >= 4) {
    // ---------------- producers ----------------
    const int pt = t - 256, g = pt % KG, r = pt / KG;
    float4 cur[A_N + B_N], nxt[A_N + B_N];
    auto issue = [&](float4 (&dst)[A_N + B_N], int w, int kt) {
      const int by = w / ntile, lid = w - by * ntile;
      const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
      const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + by * a_bs), 0, M * K * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B + by * b_bs), 0, N * K * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < A_N; ++i) {
        const int m = m0 + r + RP * i;
        dst[i] = bload(rA, m < M ? (m * K + kt * BKK + g * 4) * 4 : 0x7fffffff);
      }
#pragma unroll
      for (int i = 0; i < B_N; ++i) {
        const int n = n0 + r + RP * i;
        dst[A_N + i] = bload(rB, n < N ? (n * K + kt * BKK + g * 4) * 4 : 0x7fffffff);
      }
    };
    int w = blockIdx.x, kt = 0;
    if (w < total) issue(cur, w, 0);
    while (w < total) {
      // the slab after this one (possibly the first of the next tile) is requested before this one is parked
      int w2 = w, kt2 = kt + 1;
      if (kt2 == nk) kt2 = 0, w2 = w + gridDim.x;
      if (w2 < total) issue(nxt, w2, kt2);
      const int st = it % STAGES;
      const unsigned use = it / STAGES;
      if (!ws_wait(&empty[st], 4 * use, err)) return;
      float* As = smem + st * STAGE;
      float* Bs = As + A_FLOATS;
#pragma unroll
      for (int i = 0; i < A_N; ++i) *reinterpret_cast<float4*>(As + (r + RP * i) * LDK + g * 4) = cur[i];
#pragma unroll
      for (int i = 0; i < B_N; ++i) *reinterpret_cast<float4*>(Bs + (r + RP * i) * LDK + g * 4) = cur[A_N + i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(const_cast<unsigned*>(&full[st]), 1u);
#pragma unroll
      for (int i = 0; i < A_N + B_N; ++i) cur[i] = nxt[i];
      ++it, w = w2, kt = kt2;
    }
    return;
  }
  // ---------------- consumers ----------------
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  unsigned seen = 0;
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    const int by = w / ntile, lid = w - by * ntile;
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    for (int kt = 0; kt < nk; ++kt, ++it) {
      const int st = it % STAGES;
      const unsigned use = it / STAGES;
      if ((int)(seen - 4 * (use + 1)) < 0)            // `seen`: this stage's counter as read under the previous slab's MFMAs
        if (!ws_wait(&full[st], 4 * (use + 1), err)) return;
      seen = full[(it + 1) % STAGES];                 // next stage's counter, in flight during this slab
      const float* As = smem + st * STAGE;
      const float* Bs = As + A_FLOATS;
#pragma unroll
      for (int kg = 0; kg < BKK / 8; ++kg) {
        const int kk = kg * 8 + half * 4;
        float4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4*>(As + (wm * (BM / 2) + i * 32 + l31) * LDK + kk);
#pragma unroll
        for (int i = 0; i < TN; ++i) fb[i] = *reinterpret_cast<const float4*>(Bs + (wn * (BN / 2) + i * 32 + l31) * LDK + kk);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
              const float a = j == 0 ? fa[i].x : (j == 1 ? fa[i].y : (j == 2 ? fa[i].z : fa[i].w));
              const float b = j == 0 ? fb[n].x : (j == 1 ? fb[n].y : (j == 2 ? fb[n].z : fb[n].w));
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][n], 0, 0, 0);
            }
      }
      // the fragment reads of this slab have returned (their values fed the MFMAs above): hand the stage back
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(const_cast<unsigned*>(&empty[st]), 1u);
    }
    float* Cb = C + by * c_bs;
    if (EPS) {
      constexpr int WC = BN / 2, LDC = WC + 4, LPR = WC / 4, RPI = 64 / LPR;
      float* Ws = stage_c + wave * 32 * LDC;          // wave-private staging rows (outside the ring)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int q = 0; q < 16; ++q) Ws[((q & 3) + 8 * (q >> 2) + 4 * half) * LDC + n * 32 + l31] = acc[i][n][q];
#pragma unroll
        for (int r0 = 0; r0 < 32; r0 += RPI) {
          const int rr = r0 + lane / LPR, cc = (lane % LPR) * 4;
          const float4 v = *reinterpret_cast<const float4*>(Ws + rr * LDC + cc);
          const int row = m0 + wm * (BM / 2) + i * 32 + rr, col = n0 + wn * WC + cc;
          if (row < M && col < N) *reinterpret_cast<float4*>(Cb + (long)row * N + col) = v;
        }
      }
    } else {
      const int rbase = m0 + wm * (BM / 2) + 4 * half, cbase = n0 + wn * (BN / 2) + l31;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const int col = cbase + n * 32;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = rbase + i * 32 + (q & 3) + 8 * (q >> 2);
            if (row < M && col < N) Cb[(long)row * N + col] = acc[i][n][q];
          }
        }
    }
  }
}
template <int BM, int BN, int STAGES, int EPS, int MINB = 1>
double run_ws(const Shape& s, const float* A, const float* B, float* C, int per_cu, int iters = 20) {
  const int tiles_m = (s.M + BM - 1) / BM, tiles_n = (s.N + BN - 1) / BN;
  const int total = tiles_m * tiles_n * s.batch;
  static_assert(2 * STAGES <= 16, "flag words");
  const size_t smem = ((size_t)STAGES * (BM + BN) * 36 + 16 + (EPS ? 4 * 32 * (BN / 2 + 4) : 0)) * sizeof(float);
  auto kern = gemm_ws<BM, BN, STAGES, EPS, MINB>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  static int* err = nullptr;
  if (!err) { CK(hipMalloc(&err, 4)); }
  CK(hipMemset(err, 0, 4));
  const int grid = std::min(total, 256 * per_cu);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, 0, A, B, C, s.M, s.N, s.K, (long)s.M * s.K, (long)s.N * s.K, (long)s.M * s.N, tiles_m, tiles_n, s.batch, err);
  CK(hipDeviceSynchronize());
  int herr = 0;
  CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  if (herr) { printf("   !! gemm_ws: a hand-off wait ran out (protocol bug), skipping\n"); return 1e9; }
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, 0, A, B, C, s.M, s.N, s.K, (long)s.M * s.K, (long)s.N * s.K, (long)s.M * s.N, tiles_m, tiles_n, s.batch, err);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / iters * 1e3;
}

static void fill(float* d, size_t n, unsigned seed) {
  std::vector<float> h(n);
  unsigned x = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 32768.f - 1.f; }
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
}

static double check(const Shape& s, const float* dA, const float* dB, const float* dC) {
  // spot check 64 outputs of batch 0 and the last batch against a double dot product
  std::vector<float> hA((size_t)s.M * s.K), hB((size_t)s.N * s.K), hC((size_t)s.M * s.N);
  double worst = 0;
  for (int b : {0, s.batch - 1}) {
    CK(hipMemcpy(hA.data(), dA + (size_t)b * s.M * s.K, hA.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hB.data(), dB + (size_t)b * s.N * s.K, hB.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hC.data(), dC + (size_t)b * s.M * s.N, hC.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 64; ++i) {
      const int m = (int)((i * 2654435761u) % (unsigned)s.M), n = (int)((i * 40503u + 7) % (unsigned)s.N);
      double ref = 0;
      for (int k = 0; k < s.K; ++k) ref += (double)hA[(size_t)m * s.K + k] * hB[(size_t)n * s.K + k];
      worst = std::max(worst, std::abs(ref - hC[(size_t)m * s.N + n]));
    }
  }
  return worst;
}

int main(int argc, char** argv) {
  const Shape shapes[] = {
      {"wino 256->256 @48 (b36)", 18432, 256, 256, 36},      // final1.3 / layer3 conv2 class
      {"wino 512->512 @48 d2 (b36)", 1152, 512, 512, 36},     // layer4 conv2
      {"wino 2048->256 (b36)", 1152, 256, 2048, 36},          // ASPP d6/d12
      {"1x1 512->2048 @48", 18432, 2048, 512, 1},             // layer4 conv3
      {"1x1 1024->256 @48", 18432, 256, 1024, 1},             // layer3 conv1
      {"1x1 64->256 @192", 294912, 256, 64, 1},               // layer1 conv3 (HBM-bound)
  };
  size_t maxA = 0, maxB = 0, maxC = 0;
  for (const Shape& s : shapes) {
    maxA = std::max(maxA, (size_t)s.M * s.K * s.batch), maxB = std::max(maxB, (size_t)s.N * s.K * s.batch), maxC = std::max(maxC, (size_t)s.M * s.N * s.batch);
  }
  float *A, *B, *C;
  CK(hipMalloc(&A, maxA * 4)); CK(hipMalloc(&B, maxB * 4)); CK(hipMalloc(&C, maxC * 4));
  fill(A, maxA, 1); fill(B, maxB, 2);
  printf("%-30s %-44s %9s %8s\n", "shape", "variant", "us", "TF");
  for (const Shape& s : shapes) {
    const double fl = 2.0 * s.M * s.N * s.K * s.batch;
    auto rep = [&](const char* v, double us) { printf("%-30s %-44s %9.1f %8.1f\n", s.name, v, us, fl / us * 1e-6); fflush(stdout); };
    if (argc > 1 && std::string(argv[1]) == "diag") {     // counter passes (rocprofv3 --pmc): the 1x1 256->1024 @48 class of the dominant kernel only
      if (std::string(s.name) != "1x1 512->2048 @48") continue;
      const Shape d = {"1x1 256->1024 @48", 18432, 1024, 256, 1};
      const double fd = 2.0 * d.M * d.N * d.K;
      auto rd = [&](const char* v, double us) { printf("%-30s %-44s %9.1f %8.1f\n", d.name, v, us, fd / us * 1e-6); fflush(stdout); };
      rd("64x64 staged stores (production)", run<64, 64, 32, 1, 0, 0, 2, 2>(d, A, B, C, 0, 5));
      rd("64x64 staged, no loads", run<64, 64, 32, 1, 0, 1, 2, 2>(d, A, B, C, 0, 5));
      rd("64x64 staged, no stores", run<64, 64, 32, 1, 0, 2, 2, 2>(d, A, B, C, 0, 5));
      rd("64x64 staged, core only", run<64, 64, 32, 1, 0, 3, 2, 2>(d, A, B, C, 0, 5));
      rd("64x64 staged, no MFMA", run<64, 64, 32, 1, 0, 4, 2, 2>(d, A, B, C, 0, 5));
      rd("128x128 staged stores", run<128, 128, 32, 1, 0, 0, 2, 2>(d, A, B, C, 0, 5));
      continue;
    }
    if (argc > 1 && std::string(argv[1]) == "wide") {      // 256-wide block tiles: 4 waves of 128x64 / 64x128 / 128x128 register tiles
      rep("128x128 staged", run<128, 128, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      rep("128x128 element", run<128, 128, 32, 1, 0, 0, 2>(s, A, B, C, 0));
      rep("256x128 staged", run<256, 128, 32, 1, 0, 0, 1, 2>(s, A, B, C, 0));
      printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
      rep("256x128 element", run<256, 128, 32, 1, 0, 0, 1>(s, A, B, C, 0));
      rep("128x256 staged", run<128, 256, 32, 1, 0, 0, 1, 2>(s, A, B, C, 0));
      rep("128x256 element", run<128, 256, 32, 1, 0, 0, 1>(s, A, B, C, 0));
      rep("256x256 element", run<256, 256, 32, 1, 0, 0, 1>(s, A, B, C, 0));
      rep("256x128 two LDS stages, element", run<256, 128, 32, 2, 0, 0, 1>(s, A, B, C, 0));
      rep("256x128 core only (no loads, no stores)", run<256, 128, 32, 1, 0, 3, 1>(s, A, B, C, 0));
      rep("256x64 staged (the vendor's tile for the K = 2048 GEMM)", run<256, 64, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      rep("256x64 element", run<256, 64, 32, 1, 0, 0, 2>(s, A, B, C, 0));
      rep("64x64 staged", run<64, 64, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      rep("128x64 staged", run<128, 64, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      rep("64x64 BK = 64 staged", run<64, 64, 64, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      continue;
    }
    if (argc > 1 && std::string(argv[1]) == "ws") {
      rep("128x128 nst1 element stores (baseline)", run<128, 128, 32, 1, 0, 0, 2>(s, A, B, C, 0));
      rep("128x128 nst1 staged stores (baseline)", run<128, 128, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      rep("64x64 nst1 staged stores (baseline)", run<64, 64, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      CK(hipMemset(C, 0, (size_t)s.M * s.N * s.batch * 4));
      rep("wave-specialised 128x128, 3 stages, 1 block/CU", run_ws<128, 128, 3, 0>(s, A, B, C, 1));
      printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
      rep("wave-specialised 128x128, 3 stages, staged stores", run_ws<128, 128, 3, 1>(s, A, B, C, 1));
      printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
      rep("wave-specialised 128x128, 2 stages, staged, 2 blocks/CU", run_ws<128, 128, 2, 1, 2>(s, A, B, C, 2));
      rep("wave-specialised 128x128, 2 stages, element, 2 blocks/CU", run_ws<128, 128, 2, 0, 2>(s, A, B, C, 2));
      rep("wave-specialised 128x128, 4 stages, element, 1 block/CU", run_ws<128, 128, 4, 0, 1>(s, A, B, C, 1));
      rep("wave-specialised 64x64, 3 stages, staged, 2 blocks/CU", run_ws<64, 64, 3, 1, 2>(s, A, B, C, 2));
      continue;
    }
    if (argc > 1 && std::string(argv[1]) == "rawbar") {
      rep("128x128 nst1", run<128, 128, 32, 1, 0, 0, 2>(s, A, B, C, 0));
      rep("128x128 nst1 raw barriers", run<128, 128, 32, 1, 0, 0, 2, 0, 3>(s, A, B, C, 0));
      printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
      rep("128x128 loads two slabs ahead", run<128, 128, 32, 1, 0, 0, 2, 0, 1>(s, A, B, C, 0));
      rep("128x128 loads two slabs ahead, raw barriers", run<128, 128, 32, 1, 0, 0, 2, 0, 2>(s, A, B, C, 0));
      printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
      rep("128x128 staged stores", run<128, 128, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      rep("128x128 staged stores, two ahead, raw barriers", run<128, 128, 32, 1, 0, 0, 2, 2, 2>(s, A, B, C, 0));
      rep("64x64 nst1", run<64, 64, 32, 1, 0, 0, 2>(s, A, B, C, 0));
      rep("64x64 nst1 raw barriers", run<64, 64, 32, 1, 0, 0, 2, 0, 3>(s, A, B, C, 0));
      rep("64x64 loads two slabs ahead", run<64, 64, 32, 1, 0, 0, 2, 0, 1>(s, A, B, C, 0));
      rep("64x64 loads two slabs ahead, raw barriers", run<64, 64, 32, 1, 0, 0, 2, 0, 2>(s, A, B, C, 0));
      printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
      rep("64x64 staged stores", run<64, 64, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      rep("64x64 staged stores, two ahead, raw barriers", run<64, 64, 32, 1, 0, 0, 2, 2, 2>(s, A, B, C, 0));
      rep("64x128 staged stores", run<64, 128, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
      rep("64x128 staged stores, two ahead, raw barriers", run<64, 128, 32, 1, 0, 0, 2, 2, 2>(s, A, B, C, 0));
      continue;
    }
    rep("128x128 bk32 nst1 (Winograd GEMM tile)", run<128, 128, 32, 1, 0, 0, 2>(s, A, B, C, 0));
    printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
    rep("  ablation: no global loads", run<128, 128, 32, 1, 0, 1, 2>(s, A, B, C, 0));
    rep("  ablation: no epilogue stores", run<128, 128, 32, 1, 0, 2, 2>(s, A, B, C, 0));
    rep("  ablation: no loads, no stores (LDS + MFMA core)", run<128, 128, 32, 1, 0, 3, 2>(s, A, B, C, 0));
    rep("  ablation: no MFMA (memory phases only)", run<128, 128, 32, 1, 0, 4, 2>(s, A, B, C, 0));
    rep("128x128 + LDS-staged row stores", run<128, 128, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
    rep("128x128 + transposed tiles, 16-byte stores", run<128, 128, 32, 1, 0, 0, 2, 1>(s, A, B, C, 0));
    rep("128x128 two LDS stages (one barrier per slab)", run<128, 128, 32, 2, 0, 0, 2>(s, A, B, C, 0));
    rep("128x128 BK = 64", run<128, 128, 64, 1, 0, 0, 2>(s, A, B, C, 0));
    rep("128x128 loads two slabs ahead", run<128, 128, 32, 1, 0, 0, 2, 0, 1>(s, A, B, C, 0));
    rep("128x128 persistent blocks x2/CU, next tile prefetched", run<128, 128, 32, 1, 1, 0, 2>(s, A, B, C, 2));
    rep("128x128 occupancy capped at 2 blocks/CU", run<128, 128, 32, 1, 0, 0, 2>(s, A, B, C, 0, 20, 20000));
    rep("64x64 bk32 nst1 (1x1 convolution tile)", run<64, 64, 32, 1, 0, 0, 2>(s, A, B, C, 0));
    rep("64x64 + LDS-staged row stores", run<64, 64, 32, 1, 0, 0, 2, 2>(s, A, B, C, 0));
    rep("64x128", run<64, 128, 32, 1, 0, 0, 2>(s, A, B, C, 0));
    rep("64x64, one wave per block (no cross-SIMD barrier)", run<64, 64, 32, 1, 0, 0, 1, 0, 0, 1>(s, A, B, C, 0));
    CK(hipMemset(C, 0, (size_t)s.M * s.N * s.batch * 4));
    rep("DMA ring 128x128 3 stages, staged stores", run_glds<128, 128, 3, 1>(s, A, B, C));
    printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
    rep("DMA ring 128x128 2 stages, staged stores", run_glds<128, 128, 2, 1>(s, A, B, C));
    rep("DMA ring 128x128 3 stages, element stores", run_glds<128, 128, 3, 0>(s, A, B, C));
    rep("DMA ring 64x64 3 stages, staged stores", run_glds<64, 64, 3, 1>(s, A, B, C));
    printf("   max |err| vs fp64 dot: %.3g\n", check(s, A, B, C));
    rep("DMA ring 64x64 4 stages, staged stores", run_glds<64, 64, 4, 1>(s, A, B, C));
    rep("DMA ring 64x128 3 stages, staged stores", run_glds<64, 128, 3, 1>(s, A, B, C));
    rep("DMA ring 128x128 4 stages, staged stores", run_glds<128, 128, 4, 1>(s, A, B, C));
  }
  return 0;
}
