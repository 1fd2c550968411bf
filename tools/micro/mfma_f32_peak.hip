// micro-benchmark: the sustained rate of bare v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 streams (no memory traffic) on this box, random operands, 1-4 waves per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_peak.hip -o tools/micro/mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-result"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int NACC>
__global__ void k(int iters, const float* in, float* out) {
  float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  if (KIND == 3) {      // the fragment supply of a real K loop, without the ring: per eight MFMAs two 8-byte LDS reads per operand side, requested one group ahead
    __shared__ float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = in[128 + (i & 1023)];
    __syncthreads();
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int base = (threadIdx.x & 63) * 2 + (threadIdx.x >> 6) * 1024;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    f32x2 fa[2][2], fb[2][2];
    fa[0][0] = *(f32x2*)&sm[(base) & 8191], fa[0][1] = *(f32x2*)&sm[(base + 128) & 8191], fb[0][0] = *(f32x2*)&sm[(base + 256) & 8191], fb[0][1] = *(f32x2*)&sm[(base + 384) & 8191];
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int cu = c & 1, nx = cu ^ 1, o = base + (it + c + 1) * 512;
        fa[nx][0] = *(f32x2*)&sm[(o) & 8191], fa[nx][1] = *(f32x2*)&sm[(o + 128) & 8191], fb[nx][0] = *(f32x2*)&sm[(o + 256) & 8191], fb[nx][1] = *(f32x2*)&sm[(o + 384) & 8191];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu][i].x, fb[cu][j].x, acc[i * 2 + j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cu][i].y, fb[cu][j].y, acc[i * 2 + j], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
      for (int q = 0; q < 16; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
  } else if (KIND == 2) {      // 32x32x2 with operands that change from MFMA to MFMA (eight random registers per side): the data toggling of a real GEMM, still no memory traffic
    float av[8], bv[8];
    for (int r = 0; r < 8; ++r) av[r] = in[128 + r * 64 + (threadIdx.x & 63)], bv[r] = in[128 + 512 + r * 64 + (threadIdx.x & 63)];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; it += 8)
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(r + i) & 7], bv[(r + 3 * i) & 7], acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
      for (int q = 0; q < 16; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
  } else if (KIND == 0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
      for (int q = 0; q < 16; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
  } else {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
      for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
      for (int q = 0; q < 4; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
  }
}
template <int KIND, int NACC>
void run(const char* name, int waves_per_simd, const float* in, float* out) {
  const int iters = 20000, threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd, blocks = 256 * (256 * waves_per_simd / threads);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, NACC>), dim3(blocks), dim3(threads), 0, 0, iters, in, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double flop = (double)blocks * (threads / 64) * iters * NACC * (KIND == 1 ? 2048.0 : 4096.0);
  printf("%-28s acc/wave %d waves/SIMD %d: %.2f ms  %.1f TFLOP/s\n", name, NACC, waves_per_simd, best, flop / best / 1e9);
}
int main() {
  static float h[128 + 1024];
  for (int i = 0; i < 128; ++i) h[i] = 0.37f + 0.013f * i;
  srand(7);
  for (int i = 128; i < 128 + 1024; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *out;
  hipMalloc(&in, sizeof(h)), hipMalloc(&out, 4);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int w : {1, 2, 4}) {
    run<2, 4>("32x32x2, random rotating operands", w, in, out);
    run<3, 2>("32x32x2, operands from LDS", w, in, out);
    run<0, 1>("v_mfma_f32_32x32x2_f32", w, in, out);
    run<0, 4>("v_mfma_f32_32x32x2_f32", w, in, out);
    run<1, 1>("v_mfma_f32_16x16x4_f32", w, in, out);
    run<1, 4>("v_mfma_f32_16x16x4_f32", w, in, out);
    run<1, 8>("v_mfma_f32_16x16x4_f32", w, in, out);
  }
  return 0;
}
