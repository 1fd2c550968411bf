// LAB KERNEL (round 3, lead 1 of DESIGN.md "Leads after round 3"): how far does the simplest LDS-DMA pipeline carry the bf16 pointwise GEMM of the
// configs[2] tier?  C[M][N] (fp32) = A[M][K] . B[N][K]^T, both operands bf16, k-contiguous -- the 1x1 convolutions (and, with a gather on the source
// address, every convolution) of the tier. 128 x 128 block tile, BK = 64 (128-byte rows), 4 waves of 2 x 2 v_mfma_f32_32x32x16_bf16 tiles, operands
// staged by global_load_lds (16 B per lane, LDS image lane-linear) into two LDS buffers, XOR swizzle applied on the SOURCE chunk index so that the
// ds_read_b128 fragment reads are conflict-free. One barrier pair per K-step; hipcc drains vmcnt(0) at the barrier (two buffers: that is all it needs).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 bf16_glds_lab.hip -o bf16_glds_lab ; run: ./bf16_glds_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                        \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int BM = 128, BN = 128, BKB = 128;   // K-step in BYTES per row (64 bf16)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__global__ __launch_bounds__(256, 2) void gemm_glds(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  extern __shared__ __align__(16) char lds[];   // 2 buffers x (A 16 KB + B 16 KB)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int tiles_n = N / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  const int kb = K * 2;                          // row pitch in bytes
  // staging: 16-byte unit u = it * 256 + t of a 128 x 128 B tile -> LDS row u >> 3, LDS chunk u & 7; it reads SOURCE chunk (u & 7) ^ ((row >> 1) & 7)
  const char* ga[4];
  const char* gb[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int u = it * 256 + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
    ga[it] = (const char*)A + (long)(m0 + row) * kb + ch * 16;
    gb[it] = (const char*)B + (long)(n0 + row) * kb + ch * 16;
  }
  auto stage = [&](int buf, int kt) {
    char* la = lds + buf * 32768;
    char* lb = la + 16384;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      // LDS destination: wave-uniform base + lane * 16 (the instruction adds the lane part itself)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[it] + (long)kt * BKB),
                                       (__attribute__((address_space(3))) void*)(la + (it * 256 + wave * 64) * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[it] + (long)kt * BKB),
                                       (__attribute__((address_space(3))) void*)(lb + (it * 256 + wave * 64) * 16), 16, 0, 0);
    }
  };
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  auto compute = [&](int buf) {
    const char* la = lds + buf * 32768;
    const char* lb = la + 16384;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int c = kg * 2 + half;      // this lane-half's 8 k of the 16-k block
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wm * 64 + i * 32 + l31, rb = wn * 64 + i * 32 + l31;
        fa[i] = *reinterpret_cast<const bf16x8*>(la + ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4));
        fb[i] = *reinterpret_cast<const bf16x8*>(lb + rb * 128 + ((c ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  };
  const int nk = kb / BKB;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    compute(kt & 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const long row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        C[row * N + n0 + wn * 64 + j * 32 + l31] = acc[i][j][q];
      }
}

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main() {
  struct Shape {
    const char* name;
    int M, N, K;
  } shapes[] = {{"1x1 1024->256 @48^2", 18432, 256, 1024},  {"1x1 512->2048 @48^2", 18432, 2048, 512}, {"1x1 256->1024 @48^2", 18432, 1024, 256},
                {"1x1 64->256 @192^2", 294912, 256, 64},    {"1x1 256->64 pad128 @192^2", 294912, 128, 256}, {"square 4096^3", 4096, 4096, 4096}};
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  for (const Shape& s : shapes) {
    const size_t na = (size_t)s.M * s.K, nb = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
    std::vector<uint16_t> ha(na), hb(nb);
    srand(1);
    for (auto& v : ha) v = f2bf((rand() % 2001 - 1000) / 1000.f);
    for (auto& v : hb) v = f2bf((rand() % 2001 - 1000) / 1000.f);
    uint16_t *da, *db;
    float* dc;
    CK(hipMalloc(&da, na * 2)); CK(hipMalloc(&db, nb * 2)); CK(hipMalloc(&dc, nc * 4));
    CK(hipMemcpy(da, ha.data(), na * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), nb * 2, hipMemcpyHostToDevice));
    const int grid = (s.M / BM) * (s.N / BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_glds, dim3(grid), dim3(256), 65536, 0, da, db, dc, s.M, s.N, s.K);
    CK(hipDeviceSynchronize());
    const int iters = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_glds, dim3(grid), dim3(256), 65536, 0, da, db, dc, s.M, s.N, s.K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    // spot check: 64 random entries against a double sum of the bf16 values
    std::vector<float> hc(nc);
    CK(hipMemcpy(hc.data(), dc, nc * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int c = 0; c < 64; ++c) {
      const long m = rand() % s.M, n = rand() % s.N;
      double ref = 0;
      for (int k = 0; k < s.K; ++k) ref += (double)bf2f(ha[m * s.K + k]) * bf2f(hb[n * s.K + k]);
      worst = fmax(worst, fabs(ref - hc[m * s.N + n]) / (fabs(ref) + 1.0));
    }
    printf("%-28s M %6d N %5d K %5d  %8.3f ms  %7.1f TFLOP/s  max rel err %.2e\n", s.name, s.M, s.N, s.K, ms, 2.0 * s.M * s.N * s.K / ms / 1e9, worst);
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dc));
  }
  return 0;
}
