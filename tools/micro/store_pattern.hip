// Micro-benchmark: HBM write rate of a [M x 256] fp32 matrix written 32x128 tile per wave, (a) in the MFMA 32x32 accumulator layout
// with 4-byte stores (what the GEMM epilogues do), (b) the same tile as 16-byte row-contiguous stores, (c) 4-byte stores but each
// instruction covering one full 256 B row segment pair.   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* C, long M, long pitch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const long tile = (long)blockIdx.x * 4 + wave;           // 32-row tile
  const int panel = blockIdx.y;                             // 128-col panel
  const long row0 = tile * 32;
  if (row0 >= M) return;
  float* base = C + row0 * pitch + panel * 128;
  const float v = (float)lane;
  if (MODE == 0) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) base[((q & 3) + 8 * (q >> 2) + 4 * half) * pitch + n * 32 + l31] = v + q;
  } else if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(base + (2 * i + half) * pitch + l31 * 4) = make_float4(v, v + 1, v + 2, v + i);
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) base[i * pitch + h * 64 + lane] = v + i;
  }
}
template <int MODE>
float run(float* C, long M, long pitch) {
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  dim3 grid((unsigned)((M / 32 + 3) / 4), 2);
  hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, C, M, pitch);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, C, M, pitch);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}
int main() {
  const long M = 294912, pitch = 256;
  float* C;
  hipMalloc(&C, M * pitch * 4 * 3);   // 3 buffers so that consecutive launches do not hit the same lines in the memory-side cache
  const double mb = M * pitch * 4 / 1e6;
  for (int rep = 0; rep < 2; ++rep) {
    float t0 = run<0>(C, M, pitch), t1 = run<1>(C + M * pitch, M, pitch), t2 = run<2>(C + 2 * M * pitch, M, pitch);
    printf("acc-layout 4B stores %.1f us (%.2f TB/s) | float4 rows %.1f us (%.2f TB/s) | 4B full-row %.1f us (%.2f TB/s)\n", t0 * 1e3, mb / t0 / 1e3,
           t1 * 1e3, mb / t1 / 1e3, t2 * 1e3, mb / t2 / 1e3);
  }
  return 0;
}
