// micro-benchmark: what does one s_barrier round cost a block of W waves (one block per CU: 144 KB of dynamic LDS)?   build: hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_cost.hip -o tools/micro/barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern __shared__ char lds[];
template <int MODE>
__global__ void k(int iters, int* out) {
  int acc = 0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (MODE == 2) acc += lds[(threadIdx.x * 16 + i) & 1023];      // one LDS read per step
    asm volatile("" ::: "memory");
  }
  if (acc == 12345) out[0] = acc;
}
int main() {
  int* out;
  hipMalloc(&out, 4);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int threads : {256, 512, 768, 1024})
    for (int mode = 0; mode < 3; ++mode)
      for (int iters : {0, 1000, 4000}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(a);
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 144 * 1024, 0, iters, out);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 144 * 1024, 0, iters, out);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 144 * 1024, 0, iters, out);
          hipEventRecord(b);
          hipEventSynchronize(b);
          float ms;
          hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        printf("threads %4d mode %d iters %4d: %.1f us\n", threads, mode, iters, best * 1e3f);
      }
  return 0;
}
