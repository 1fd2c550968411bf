// Probe of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4i16) on gfx950: which LDS elements does lane l receive for a given per-lane address pattern?
// LDS holds s[i] = i (16-bit). Pattern P: lane l supplies the address of element off(l); the four returned values are printed per lane.
// build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(int pattern, int rs, short* out) {
  __shared__ __attribute__((aligned(16))) short s[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  int off;
  if (pattern == 0) off = g * 1024 + (i & 3) * rs + (i >> 2) * 4;        // lane i of a 16-group: row i%4, column block i/4
  else if (pattern == 1) off = g * 1024 + (i >> 2) * rs + (i & 3) * 4;   // row i/4, column block i%4
  else off = l * 4;                                                      // plain: own 8 bytes
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(s + off));
  out[l * 4 + 0] = r.x, out[l * 4 + 1] = r.y, out[l * 4 + 2] = r.z, out[l * 4 + 3] = r.w;
}
int main() {
  short* d;
  hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int p = 0; p < 3; ++p) {
    const int rs = 64;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, rs, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d (row stride %d elements): lane -> 4 elements as (row, col) relative to the group's base\n", p, rs);
    for (int l = 0; l < 64; ++l) {
      printf("  l%2d:", l);
      for (int j = 0; j < 4; ++j) {
        int v = h[l * 4 + j] - (p < 2 ? (l >> 4) * 1024 : 0);
        printf(" (%d,%2d)", v / rs, v % rs);
      }
      if ((l & 3) == 3) printf("\n");
    }
  }
  return 0;
}
