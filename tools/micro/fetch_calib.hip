// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 by access width (VERDICT r3 item 6 / MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half the
// bytes of a 16-B-per-lane streaming read; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern"). Each kernel streams a buffer of known size (512 MiB: past the 256 MiB Infinity Cache) exactly once with one access width:
//   rd16 / rd8 / rd4   16 / 8 / 4 bytes per lane, consecutive lanes consecutive addresses (8 B = the int64 label read of the CE kernels)
//   wr16 / wr8 / wr4   the same for stores (4 B = an MFMA accumulator-layout store: 32 lanes x 4 B per row segment)
//   wr4_rows           4-byte stores in the accumulator layout proper: lane l -> column l & 31 of row (l >> 5) + 2 i, 128-B row segments at a 1 KiB pitch
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 fetch_calib.hip -o fetch_calib ; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -- ./fetch_calib   and   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -- ./fetch_calib
// then tools/fetch_calib_summary.py out_f/..db out_w/..db  prints counter bytes / true bytes per kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                   \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) {                                     \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));     \
      exit(1);                                                  \
    }                                                           \
  } while (0)

constexpr size_t BYTES = 512ull << 20;

template <typename T>
__global__ __launch_bounds__(256) void rd_kernel(const T* __restrict__ p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = p[i];
    const unsigned* w = reinterpret_cast<const unsigned*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc ^= w[k];
  }
  if (acc == 0x12345678u) *sink = acc;      // keeps the loads alive
}
template <typename T>
__global__ __launch_bounds__(256) void wr_kernel(T* __restrict__ p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
// accumulator-layout 4-byte stores: a wave writes two 128-byte row segments per instruction, rows 1 KiB apart (a 256-column fp32 output tile)
__global__ __launch_bounds__(256) void wr4_rows_kernel(float* __restrict__ p, size_t rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 32; r0 < rows; r0 += (size_t)gridDim.x * 4 * 32)
    for (int cb = 0; cb < 8; ++cb)            // 8 column blocks of 32 floats cover the 256-float row
#pragma unroll
      for (int i = 0; i < 16; ++i) p[(r0 + (lane >> 5) + 2 * i) * 256 + cb * 32 + (lane & 31)] = 1.0f;
}

int main() {
  void* buf;
  unsigned* sink;
  CK(hipMalloc(&buf, BYTES));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, BYTES));
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(rd_kernel<uint4>, grid, block, 0, 0, (const uint4*)buf, BYTES / 16, sink);
    hipLaunchKernelGGL(rd_kernel<uint2>, grid, block, 0, 0, (const uint2*)buf, BYTES / 8, sink);
    hipLaunchKernelGGL(rd_kernel<unsigned>, grid, block, 0, 0, (const unsigned*)buf, BYTES / 4, sink);
    hipLaunchKernelGGL(wr_kernel<uint4>, grid, block, 0, 0, (uint4*)buf, BYTES / 16, make_uint4(1, 2, 3, 4));
    hipLaunchKernelGGL(wr_kernel<uint2>, grid, block, 0, 0, (uint2*)buf, BYTES / 8, make_uint2(1, 2));
    hipLaunchKernelGGL(wr_kernel<unsigned>, grid, block, 0, 0, (unsigned*)buf, BYTES / 4, 7u);
    hipLaunchKernelGGL(wr4_rows_kernel, grid, block, 0, 0, (float*)buf, BYTES / 1024);
    CK(hipDeviceSynchronize());
  }
  printf("true bytes per launch: %zu\n", BYTES);
  return 0;
}
