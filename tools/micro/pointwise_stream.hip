// LAB KERNEL, NOT PART OF THE LIBRARY (round 3; it was wired into pm_conv_fwd / pm_conv_bwd_data at commit "Streaming kernel for the short-reduction
// 1x1 convolutions", passed the kernel tests there -- bit-identical to the tiled kernel: the same k pairing per MFMA -- and measured SLOWER on the
// same box: 64 -> 256 @192^2 0.140 vs 0.135 ms, 128 -> 512 @96^2 0.130 vs 0.114 ms, data gradient of 256 -> 64 0.149 vs 0.137 ms, step 65.5 vs 64.3 ms).
// These shapes sit where both rooflines meet (9.66 GFLOP = 61 us of fp32 MFMA; 370 MB = 62 us of HBM at 6 TB/s): the tiled kernel and this one both
// land at ~2.2 x that, with exactly algorithmic traffic (tools/gpu_shape_traffic.sh). Kept for the record next to gemm_lab.hip.
//
// Streaming kernel for the short-reduction 1x1 convolutions (Cin = 64 / 128 forward, Cout = 64 / 128 data gradient): layer1's 64 -> 256 and
// 64 -> 64 (Resnet.py:145-150 at 192 x 192), layer2's 128 -> 512, and the data gradients of the 256 -> 64 / 512 -> 128 reductions.
//   C[M][N] = A[M][K] . B[N][K]^T,  K in {64, 128}: 2 / 4 K-steps of the tiled implicit-GEMM kernel, whose blocks then live for one load phase, a
//   few hundred MFMA cycles and one store phase -- measured 70 ... 91 TFLOP/s with the HBM side at half its rate although the traffic is exactly
//   algorithmic (tools/gpu_shape_traffic.sh: 74 MB read, 295 MB written per 64 -> 256 launch): nothing overlaps inside such a block and the
//   co-resident blocks run in phase.
// Here the whole weight chunk B (NC columns x K, <= 68 KB) sits in LDS for the lifetime of a persistent block; every WAVE streams its own 32-row
// tiles of A through a private LDS stage -- no block barrier after the prologue, so the eight waves of a CU drift apart and one wave's MFMAs run
// under another's loads and stores: the next tile's rows are in flight (registers) during the current tile's MFMAs, a finished 32 x 32 tile goes
// through the wave's stage into whole 128-byte row segments. v_mfma_f32_32x32x2_f32, k ascending per lane pair: the same fp32 products as the
// tiled kernel, accumulated in a different (fixed) order.
#include "../../pinthememory_amd/csrc/pm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct PwArgs {
  const float* A;       // [M][lda]
  long lda;
  const float* B;       // forward: w [N][K]; data gradient: w [K][N] (TRANSB)
  float* C;             // [M][ldc]
  long ldc;
  long M;
  int N;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;     // residual / fused gradient add, [M][ldr]
  long ldr;
  int relu;
};

template <int K, int NC, int WAVES, bool TRANSB>
__global__ __launch_bounds__(WAVES * 64) void pw_stream_kernel(const PwArgs a) {
  constexpr int LD = K + 4;                 // row pitch (floats): 16-byte chunks of 16 consecutive rows on 16 distinct bank quads
  constexpr int TN = NC / 32;               // 32 x 32 MFMA tiles per row tile
  constexpr int AV = K / 8;                 // float4 per lane of one 32-row A tile (32 x K floats over 64 lanes)
  static_assert(K % 8 == 0 && NC % 32 == 0, "tile config");
  extern __shared__ __align__(16) float lds[];
  float* Bs = lds;                                            // [NC][LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* As = lds + NC * LD + wave * (32 * LD);               // this wave's stage [32][LD]
  const int n0 = blockIdx.y * NC;

  for (int i = tid; i < NC * (K / 4); i += WAVES * 64) {      // the weight chunk, once per block
    const int n = i / (K / 4), k4 = (i - n * (K / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 + n < a.N) {
      if constexpr (TRANSB) {
        const float* p = a.B + (long)k4 * a.N + n0 + n;
        v = make_float4(p[0], p[a.N], p[2 * (long)a.N], p[3 * (long)a.N]);
      } else {
        v = PM_LD4(a.B + (long)(n0 + n) * K + k4);
      }
    }
    *reinterpret_cast<float4*>(Bs + n * LD + k4) = v;
  }
  __syncthreads();

  const long ntiles = (a.M + 31) / 32;
  const long tstride = (long)gridDim.x * WAVES;
  long t = (long)blockIdx.x * WAVES + wave;
  const int l31 = lane & 31, half = lane >> 5;
  // loader: float4 index j * 64 + lane of the tile -> row (j * 64 + lane) / (K / 4), chunk (j * 64 + lane) % (K / 4): whole rows, coalesced
  float4 ra[AV];
  auto gload = [&](long tile) {
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      const int idx = j * 64 + lane, row = idx / (K / 4), c4 = idx - row * (K / 4);
      const long m = tile * 32 + row;
      ra[j] = (tile < ntiles && m < a.M) ? PM_LD4(a.A + m * a.lda + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      const int idx = j * 64 + lane, row = idx / (K / 4), c4 = idx - row * (K / 4);
      *reinterpret_cast<float4*>(As + row * LD + c4 * 4) = ra[j];
    }
  };
  // DS operations of one wave execute in program order; the fences below only keep the compiler from moving them across each other
  auto wave_fence = [] { __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  gload(t);
  for (; t < ntiles; t += tstride) {
    lstore();
    wave_fence();
    gload(t + tstride);                     // in flight under this tile's MFMAs
    f32x16 acc[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[n][q] = 0.f;
#pragma unroll
    for (int kk = 0; kk < K / 8; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(As + l31 * LD + kk * 8 + half * 4);
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const float4 bv = *reinterpret_cast<const float4*>(Bs + (n * 32 + l31) * LD + kk * 8 + half * 4);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[n], 0, 0, 0);
      }
    }
    wave_fence();                           // the stage is free: it now carries the finished tiles out, one 32 x 32 tile at a time
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
#pragma unroll
      for (int q = 0; q < 16; ++q) As[((q & 3) + 8 * (q >> 2) + 4 * half) * 36 + l31] = acc[n][q];
      wave_fence();
      const int col = n0 + n * 32 + ec;
      const bool cok = col < a.N;
      float bi[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
      if (cok && (a.bias || a.scale)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (a.bias) bi[e] = a.bias[col + e];
          if (a.scale) sc[e] = a.scale[col + e], sh[e] = a.shift[col + e];
        }
      }
#pragma unroll
      for (int r0 = 0; r0 < 32; r0 += 8) {
        const int rr = r0 + er;
        const long m = t * 32 + rr;
        float4 v = *reinterpret_cast<const float4*>(As + rr * 36 + ec);
        if (m < a.M && cok) {
          if (a.bias || a.scale) v.x = (v.x + bi[0]) * sc[0] + sh[0], v.y = (v.y + bi[1]) * sc[1] + sh[1], v.z = (v.z + bi[2]) * sc[2] + sh[2], v.w = (v.w + bi[3]) * sc[3] + sh[3];
          if (a.res) {
            const float4 qv = PM_LD4(a.res + m * a.ldr + col);
            v.x += qv.x, v.y += qv.y, v.z += qv.z, v.w += qv.w;
          }
          if (a.relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
          PM_ST4(a.C + m * a.ldc + col, v);
        }
      }
      wave_fence();
    }
  }
}

template <int K, int NC, int WAVES, bool TRANSB>
int pw_launch(const PwArgs& a, hipStream_t st) {
  constexpr size_t lds = (size_t)(NC * (K + 4) + WAVES * 32 * (K + 4)) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS");
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_stream_kernel<K, NC, WAVES, TRANSB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)attr_set;
  const long ntiles = (a.M + 31) / 32;
  const int nchunks = (a.N + NC - 1) / NC;
  // one persistent block per CU and column chunk (fewer for small problems: >= 2 tiles per wave)
  const int gx = (int)std::max<long>(1, std::min<long>(256 / std::min(nchunks, 4), (ntiles + 2 * WAVES - 1) / (2 * WAVES)));
  hipLaunchKernelGGL((pw_stream_kernel<K, NC, WAVES, TRANSB>), dim3(gx, nchunks), dim3(WAVES * 64), lds, st, a);
  return pm_check_launch("pw_stream");
}

}  // namespace

// C[M][N] = epilogue(A[M][K] . B^T): K = 64 or 128. transb: B is [K][N] (the data gradient reads the [Cout][Cin] weights as they lie). Returns
// PM_EUNSUPPORTED for shapes the kernel is not built for (the caller then takes the tiled kernel).
int pm_pointwise_stream(const float* A, long lda, const float* B, bool transb, float* C, long ldc, long M, int N, int K, const float* bias, const float* scale,
                        const float* shift, const float* res, long ldr, int relu, hipStream_t st) {
  if (!(K == 64 || K == 128) || (N & 31) || (lda & 3) || (ldc & 3) || (res && (ldr & 3)) || !pm_aligned16(A) || !pm_aligned16(B) || !pm_aligned16(C) ||
      (res && !pm_aligned16(res)))
    return PM_EUNSUPPORTED;
  PwArgs a{A, lda, B, C, ldc, M, N, bias, scale, shift, res, ldr, relu};
  if (K == 64) {
    if (N % 256 == 0) return transb ? pw_launch<64, 256, 8, true>(a, st) : pw_launch<64, 256, 8, false>(a, st);
    if (N % 64 == 0) return transb ? pw_launch<64, 64, 8, true>(a, st) : pw_launch<64, 64, 8, false>(a, st);
    return PM_EUNSUPPORTED;
  }
  if (N % 128 == 0) return transb ? pw_launch<128, 128, 4, true>(a, st) : pw_launch<128, 128, 4, false>(a, st);
  return PM_EUNSUPPORTED;
}
