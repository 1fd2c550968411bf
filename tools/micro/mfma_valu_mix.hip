// LAB (round 6): how many plain VALU instructions hide in the shadow of one v_mfma_f32_32x32x16_bf16 -- the budget of the split-operand fp32 path (conv_split.hip), whose
// K-step carries ~5 VALU (v_and_b32 / v_sub_f32 / v_perm_b32) per MFMA. One block of 256 threads per CU slot (1 wave per SIMD) or two (2 waves per SIMD); every wave runs
// ITERS x 4 MFMAs on four accumulators with F independent VALU fillers behind each MFMA (sched_group_barrier pins the order). Prints ns per MFMA and per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu_mix.hip -o mfma_valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

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

template <int F, int KIND>
__global__ __launch_bounds__(256, 2) void mix_kernel(float* out, int iters, unsigned seed) {
  const int t = threadIdx.x;
  u32x4 ua = {seed + t, seed * 3 + t, seed * 5 + t, seed * 7 + t}, ub = {seed ^ t, seed + 2 * t, seed + 3 * t, seed + 5 * t};
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  f32x16 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (float)(t + j) * 1.37f + (float)seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        const int j = (m * F + f) & 7;
        if (KIND == 0) {      // the split's own mix: and / sub / perm on independent chains
          const unsigned u = __float_as_uint(x[j]);
          if (f % 3 == 0) x[j] = __uint_as_float(u & 0xffff0fffu | 0x1000u);
          else if (f % 3 == 1) x[j] = x[j] - 1.25f;
          else x[j] = __uint_as_float(__builtin_amdgcn_perm(u, __float_as_uint(x[(j + 1) & 7]), 0x07060302));
        } else if (KIND == 1) {      // fp32 adds only
          x[j] = x[j] + 1.25f;
        } else {      // integer ands only
          x[j] = __uint_as_float(__float_as_uint(x[j]) & (0xfffffff0u + f));
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (F > 0) __builtin_amdgcn_sched_group_barrier(0x002, F, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[m][q];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * 256 + t] = s;
}

template <int F, int KIND>
void run(float* out, int blocks_per_cu, int iters) {
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mix_kernel<F, KIND>), dim3(grid), dim3(256), 0, 0, out, iters, 1u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((mix_kernel<F, KIND>), dim3(grid), dim3(256), 0, 0, out, iters, 2u);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = (double)iters * 4 * blocks_per_cu;
  const double tf = 2.0 * 32 * 32 * 16 * (double)iters * 4 * 4 * grid / (ms * 1e-3) * 1e-12;
  printf("kind %d  F=%d fillers/MFMA  %d wave(s)/SIMD: %.3f ms  %.2f ns per MFMA and SIMD  (%.0f TF bf16)\n", KIND, F, blocks_per_cu, ms, ms * 1e6 / mfma_per_simd, tf);
}

int main() {
  float* out;
  CK(hipMalloc(&out, 256 * 2 * 256 * sizeof(float)));
  const int iters = 20000;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    run<0, 0>(out, bpc, iters);
    run<2, 0>(out, bpc, iters);
    run<4, 0>(out, bpc, iters);
    run<5, 0>(out, bpc, iters);
    run<6, 0>(out, bpc, iters);
    run<7, 0>(out, bpc, iters);
    run<8, 0>(out, bpc, iters);
    run<10, 0>(out, bpc, iters);
    run<5, 1>(out, bpc, iters);
    run<7, 1>(out, bpc, iters);
    run<5, 2>(out, bpc, iters);
    run<7, 2>(out, bpc, iters);
  }
  return 0;
}
