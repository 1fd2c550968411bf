// BASELINE configs[2], round 5: the HBM-bound 1x1 convolutions of the bf16 tier as a STREAMING kernel.
//   y[M][N] = x[M][K] . w[N][K]^T,  K = 64 / 128 / 256 input channels, N = 128 ... 1024 output channels, M = 18 432 ... 294 912 pixels
// (conv3 / downsample of layer1-3 and the data gradients of their conv1: 64 -> 256 on the 192 x 192 maps, 128 -> 512 on 96 x 96, 256 -> 1024 on 48 x 48). These sit below
// the bf16 ridge (51 ... 100 FLOP per byte): what they cost is the activation rows in and the 4 x wider rows out. As tiles of the implicit-GEMM kernel (conv16.hip) they
// move 3.1 TB/s in the step (189 MB in 60 us for 64 -> 256) where the elementwise passes next to them reach 4.8-5.5 TB/s: every 64 / 128-row block pays its own
// prologue, weight fetch, barrier pair and LDS-staged epilogue for ~8 KB of input.
// Here a block is PERSISTENT over the pixel rows: the weights of its channel chunk (<= 64 KB) are fetched into LDS ONCE, the activation tiles stream through a four-stage
// LDS ring filled by LDS-DMA with counted waits (three tiles in flight per CU), and the output leaves straight from the accumulators as 16-byte stores: the MFMA runs
// with the WEIGHTS as its row operand, so a lane's accumulator registers hold channels of ONE pixel, and a half-wave exchange (v_permlane32_swap) turns the 4 + 4
// channel groups of the two half-waves into 8 consecutive bf16 channels per lane -- no LDS round trip, no barrier between compute and store.
// Fused epilogue as everywhere: (acc + bias) * scale + shift (+ residual) (ReLU), rounded to bf16 once. Deterministic (no atomics, no split reductions).
// Replaces nn.Conv2d forward / input gradient of Resnet.py:145-150,195 (conv3, downsample.0, dgrad of conv1) on the bf16 tier where the planner picks it.
#include <stdlib.h>
#include <algorithm>

#include "pm_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BKB = 128;      // bytes per row and K-step (64 bf16)
constexpr int NT = 256;       // four waves

__device__ __forceinline__ void dma16p(__amdgpu_buffer_rsrc_t r, char* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm_p() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}
// vmcnt(d * GX + st * GS + r * GR) for the few combinations a block meets. Every vector-memory operation of a wave -- LDS-DMA fetch, load, store -- retires in issue
// order on gfx9-class counters, so "stage i has landed" = "all but the operations issued after it have retired":
//   d  = activation stages issued after it (0 ... NA - 2: fewer at the end of a block's tile list),
//   st = store groups issued after it (min(i, 3): none before the block's first tile), r = residual-load groups issued after it (min(i, 2)).
// Issue order of a tile: [wait, barrier] residual loads of tile i -> stage i + 3 -> MFMAs -> stores of tile i.
template <int GX, int GS, int GR>
__device__ __forceinline__ void wait_stage(int d, int i) {
  switch (d * 4 + min(i, 3)) {
    case 0: wait_vm_p<0>(); break;
    case 1: wait_vm_p<GS + GR>(); break;
    case 2: wait_vm_p<2 * GS + 2 * GR>(); break;
    case 3: wait_vm_p<3 * GS + 2 * GR>(); break;
    case 4: wait_vm_p<GX>(); break;
    case 5: wait_vm_p<GX + GS + GR>(); break;
    case 6: wait_vm_p<GX + 2 * GS + 2 * GR>(); break;
    case 7: wait_vm_p<GX + 3 * GS + 2 * GR>(); break;
    case 8: wait_vm_p<2 * GX>(); break;
    case 9: wait_vm_p<2 * GX + GS + GR>(); break;
    case 10: wait_vm_p<2 * GX + 2 * GS + 2 * GR>(); break;
    default: wait_vm_p<2 * GX + 3 * GS + 2 * GR>(); break;
  }
}

struct pm_pw16 {
  const pm_bf16* X;      // [M][x_pitch] bf16 rows, K valid (zero-padded) channels
  const pm_bf16* W;      // [Nn][K] bf16 (pm_bf16_cast_weights of a 1x1 filter)
  pm_bf16* Y;            // [M][y_pitch]
  const pm_bf16* R;      // residual, same shape as Y, or null
  long x_pitch, y_pitch, r_pitch;
  int M, Nn, K;
  const float *bias, *scale, *shift;
  int relu, tiles_m;
};

// KS = K / 64; NC = output channels per block; BM = pixels per tile (64: two 32-pixel column blocks per wave, 32: one)
template <int KS, int NC, int BM, bool RES>
__global__ __launch_bounds__(NT, 1) void pw16_kernel(const pm_pw16 a) {
  constexpr int NA = 4;
  constexpr int TP = BM / 32, TC = NC / 4 / 32;
  constexpr int W_BYTES = KS * NC * BKB, X_STAGE = KS * BM * BKB;
  constexpr int GX = KS * BM * 8 / NT;              // LDS-DMA instructions of one wave per activation stage
  constexpr int GS = TP * TC * 2, GR = RES ? GS : 0;      // 16-byte stores / residual loads of one wave per tile
  constexpr int W_IT = NC * 8 / NT;
  static_assert(TC >= 1 && GX >= 1 && 2 * GX + 3 * GS + 2 * GR < 64, "tile config");
  extern __shared__ __align__(16) char lds[];
  char* lw = lds;                            // [KS][NC][128 B], chunk-swizzled
  char* lx = lds + W_BYTES;                  // NA stages of [KS][BM][128 B]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, half = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int n0 = blockIdx.y * NC;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.X), 0, (int)((long)a.M * a.x_pitch * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<pm_bf16*>(a.W), 0, (int)((long)a.Nn * a.K * 2), 0x00020000);
  constexpr int OOB = 0x7fffffff;

  // ---- once per block: the weights of this channel chunk and its epilogue constants ------------------------------------------------------------------
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int u = it * NT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
      const int n = n0 + row;
      dma16p(rW, lw + ks * NC * BKB + (it * NT + wave_u * 64) * 16, n < a.Nn ? n * a.K * 2 + ks * BKB + ch * 16 : OOB, 0);
    }
  // epilogue constants of this lane's channels, in REGISTERS (an LDS copy would be read behind the LDS-DMA queue: the compiler orders every such read behind
  // vmcnt(0), which drains the ring): after the half-wave exchange a lane owns channels cb + 8 k + 8 half ... + 7 for k = 0, 2 of each of its TC channel blocks
  const bool aff = a.bias || a.scale;
  float kbi[TC][2][8], ksc[TC][2][8], ksh[TC][2][8];
#pragma unroll
  for (int ci = 0; ci < TC; ++ci)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int col = n0 + wave * (NC / 4) + ci * 32 + 16 * kk + 8 * half;
      const bool ok = aff && col < a.Nn;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        kbi[ci][kk][e] = (ok && a.bias) ? a.bias[col + e] : 0.f;
        ksc[ci][kk][e] = (ok && a.scale) ? a.scale[col + e] : 1.f;
        ksh[ci][kk][e] = (ok && a.scale) ? a.shift[col + e] : 0.f;
      }
    }

  // the one-time fetches (weights, constants) are complete before the ring starts: from here on every wait is a counted one and the compiler's own bookkeeping has
  // nothing older to wait for inside the loop
  wait_vm_p<0>();
  asm volatile("" ::: "memory");

  // this lane's rows of an activation stage (K-step image ks: + ks * BM * 128): source offset of tile 0, advanced by the tile stride
  int xoff[GX / KS];
#pragma unroll
  for (int it = 0; it < GX / KS; ++it) {
    const int u = it * NT + t, row = u >> 3, ch = (u & 7) ^ ((row >> 1) & 7);
    xoff[it] = row * (int)(a.x_pitch * 2) + ch * 16;      // + tile * BM * pitch bytes; rows beyond M: see stage()
  }
  const int pitchb = (int)(a.x_pitch * 2);
  auto stage = [&](int tile, int slot) {
    char* ls = lx + slot * X_STAGE;
    const int base = tile * BM * pitchb;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int it = 0; it < GX / KS; ++it) {
        const int row = (it * NT + t) >> 3;
        dma16p(rX, ls + ks * BM * BKB + (it * NT + wave_u * 64) * 16, tile * BM + row < a.M ? base + xoff[it] + ks * BKB : OOB, 0);
      }
  };

  // fragment rows: weights = the MFMA's row operand (channels), activations = its column operand (pixels)
  int w_off[TC], w_key[TC], x_off[TP], x_key[TP];
#pragma unroll
  for (int i = 0; i < TC; ++i) {
    const int r = wave * (NC / 4) + i * 32 + l31;
    w_off[i] = r * BKB, w_key[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TP; ++j) {
    const int r = j * 32 + l31;
    x_off[j] = r * BKB, x_key[j] = (r >> 1) & 7;
  }

  const int first = blockIdx.x, step = gridDim.x;
  const int ntiles = first < a.tiles_m ? (a.tiles_m - first + step - 1) / step : 0;      // tiles of this block: first, first + step, ...
#pragma unroll
  for (int p = 0; p < NA - 1; ++p)
    if (p < ntiles) stage(first + p * step, p);

  const long ypb = a.y_pitch * 2, rpb = a.r_pitch * 2;
  for (int i = 0; i < ntiles; ++i) {
    const int tile = first + i * step, slot = i & (NA - 1);
    // stage i has landed when all but the younger operations have retired: the stages issued after it and the epilogue groups issued after it
    wait_stage<GX, GS, GR>(min(NA - 2, ntiles - 1 - i), i);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // The residual rows of this tile, issued BEFORE the next stage and hidden from the compiler's wait bookkeeping (inline asm): next to LDS-DMA traffic hipcc waits
    // vmcnt(0) for every ordinary register load -- the whole ring would drain once per tile. Their own counted wait sits in front of the epilogue (all but stage i + 3).
    u32x4 rq[TP][TC][2];
    if constexpr (RES) {
#pragma unroll
      for (int p = 0; p < TP; ++p) {
        const long m = (long)tile * BM + p * 32 + l31;
#pragma unroll
        for (int ci = 0; ci < TC; ++ci)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int col = n0 + wave * (NC / 4) + ci * 32 + 16 * kk + 8 * half;
            const bool ok = m < a.M && col < a.Nn;      // every lane issues the load (a counted operation of the wave); lanes out of range read row 0 and drop it
            const char* src = reinterpret_cast<const char*>(a.R) + (ok ? m * rpb + col * 2 : 0);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rq[p][ci][kk]) : "v"(src) : "memory");
          }
      }
    }
    if (i + NA - 1 < ntiles) stage(first + (i + NA - 1) * step, (i + NA - 1) & (NA - 1));      // into the slot tile i - 1 was read from: every wave has left it

    f32x16 acc[TC][TP];
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
      for (int p = 0; p < TP; ++p)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][p][q] = 0.f;
    const char* ls = lx + slot * X_STAGE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const int c = kg * 2 + half;
        bf16x8 fw[TC], fx[TP];
#pragma unroll
        for (int ci = 0; ci < TC; ++ci) fw[ci] = *reinterpret_cast<const bf16x8*>(lw + ks * NC * BKB + w_off[ci] + ((c ^ w_key[ci]) << 4));
#pragma unroll
        for (int p = 0; p < TP; ++p) fx[p] = *reinterpret_cast<const bf16x8*>(ls + ks * BM * BKB + x_off[p] + ((c ^ x_key[p]) << 4));
#pragma unroll
        for (int ci = 0; ci < TC; ++ci)
#pragma unroll
          for (int p = 0; p < TP; ++p) acc[ci][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ci], fx[p], acc[ci][p], 0, 0, 0);
      }

    if constexpr (RES) {      // the residual loads have retired (in issue order: everything but the stage issued behind them)
      if (i + NA - 1 < ntiles) wait_vm_p<GX>();
      else wait_vm_p<0>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue from the accumulators: acc[q] = channel (q & 3) + 8 (q >> 2) + 4 half of the 32-channel block, pixel l31. One half-wave exchange per channel-group
    // pair (k, k + 1): afterwards the lower half-wave holds channels 8k ... 8k + 7, the upper one 8k + 8 ... 8k + 15 (k = 0, 2) of its pixel.
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const long m = (long)tile * BM + p * 32 + l31;
      const bool rok = m < a.M;
#pragma unroll
      for (int ci = 0; ci < TC; ++ci) {
        const int cb = wave * (NC / 4) + ci * 32;      // channel block inside the chunk
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // (the elements go through named floats: __builtin_bit_cast applied to an ext-vector ELEMENT expression reads element 0 whatever the index -- hipcc 7.2)
            const float lo = acc[ci][p][4 * k + e], hi = acc[ci][p][4 * (k + 1) + e];
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
            v[e] = __uint_as_float(r[0]), v[4 + e] = __uint_as_float(r[1]);
          }
          const int cl = cb + 8 * k + 8 * half;      // first of this lane's eight channels, inside the chunk
          const int col = n0 + cl;
          if (aff) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] + kbi[ci][k >> 1][e]) * ksc[ci][k >> 1][e] + ksh[ci][k >> 1][e];
          }
          const bool ok = rok && col < a.Nn;      // Nn % 8 == 0: a group is all in or all out
          if constexpr (RES) {
            const u32x4 q = rq[p][ci][k >> 1];
            v[0] += __uint_as_float(q.x << 16), v[1] += __uint_as_float(q.x & 0xffff0000u), v[2] += __uint_as_float(q.y << 16), v[3] += __uint_as_float(q.y & 0xffff0000u);
            v[4] += __uint_as_float(q.z << 16), v[5] += __uint_as_float(q.z & 0xffff0000u), v[6] += __uint_as_float(q.w << 16), v[7] += __uint_as_float(q.w & 0xffff0000u);
          }
          if (a.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (ok) pm_st8(reinterpret_cast<pm_bf16*>(reinterpret_cast<char*>(a.Y) + m * ypb + col * 2), v);
        }
      }
    }
  }
}

template <int KS, int NC, int BM, bool RES>
int launch_pw(const pm_pw16& k, hipStream_t st) {
  constexpr size_t smem = (size_t)KS * NC * BKB + 4 * (size_t)KS * BM * BKB;
  static_assert(smem <= 160 * 1024, "LDS budget");
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw16_kernel<KS, NC, BM, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  const int chunks = pm_cdiv(k.Nn, NC);
  static const int per_cu = getenv("PM_PW16_BLOCKS") ? atoi(getenv("PM_PW16_BLOCKS")) : 1;
  const int gx = std::min(k.tiles_m, std::max(1, 256 * per_cu / chunks));
  hipLaunchKernelGGL((pw16_kernel<KS, NC, BM, RES>), dim3(gx, chunks), dim3(NT), smem, st, k);
  return pm_check_launch("pw16");
}

}  // namespace

// Does the streaming kernel take this 1x1 convolution? (bf16 rows in place, bf16 out, K = 64 / 128 / 256, whole 8-channel output groups)
// MEASURED, LEFT OFF BY DEFAULT (round 5; PM_PW16=1 enables it by size, pm_set_conv16(5) on every eligible call -- the kernel tests run it that way): same box, alternated,
// tools/conv16_probe.py: 64 -> 256 @192^2 35-36 us (tile kernel) vs 39-40 us (this kernel), 128 -> 512 @96^2 30 vs 30-31, 256 -> 1024 @48^2 26 vs 25-26; bs=8 768^2
// step 25.90 / 25.92 ms without, 25.85 / 25.93 with. Both kernels move the 189 MB of the 64 -> 256 launch at ~5 TB/s standalone; what the step adds on top (60 us per
// launch there) is not the kernel's structure but its place in a chain of HBM-bound passes: its input was written by the kernel before it and leaves that kernel's L2s
// at the boundary, its 151 MB of output leave its own. A streaming structure has nothing to win back from that. Kept as a tested alternative.
int g_pw16 = getenv("PM_PW16") ? atoi(getenv("PM_PW16")) : 0;      // 0 off (default), 1 by size, 2 every eligible 1x1 (kernel tests); pm_set_conv16(5 / 6)
bool pm_pw16_ok(long M, int Nn, int K, long x_pitch, long y_pitch, long r_pitch, const void* x, const void* y, const void* r) {
  const int on = g_pw16;
  static const long min_m = getenv("PM_PW16_MIN_M") ? atol(getenv("PM_PW16_MIN_M")) : 4096;
  if (!on || (K != 64 && K != 128 && K != 256) || Nn < 128 || (Nn & 7) || M < (on >= 2 ? 1 : min_m)) return false;
  if ((x_pitch & 7) || (y_pitch & 7) || (r && (r_pitch & 7)) || !pm_aligned16(x) || !pm_aligned16(y) || !pm_aligned16(r)) return false;
  if ((long)M * x_pitch * 2 >= (1l << 31) || (long)Nn * K * 2 >= (1l << 31)) return false;
  return true;
}

int pm_pw16_launch(const pm_bf16* X, long x_pitch, const pm_bf16* W, pm_bf16* Y, long y_pitch, const pm_bf16* R, long r_pitch, long M, int Nn, int K, const float* bias,
                   const float* scale, const float* shift, int relu, hipStream_t st) {
  pm_pw16 k{};
  k.X = X, k.W = W, k.Y = Y, k.R = R, k.x_pitch = x_pitch, k.y_pitch = y_pitch, k.r_pitch = r_pitch, k.M = (int)M, k.Nn = Nn, k.K = K;
  k.bias = bias, k.scale = scale, k.shift = shift, k.relu = relu;
  const bool res = R != nullptr;
  if (K == 64) {
    k.tiles_m = pm_cdiv(M, 64);
    return res ? launch_pw<1, 256, 64, true>(k, st) : launch_pw<1, 256, 64, false>(k, st);
  }
  if (K == 128) {
    k.tiles_m = pm_cdiv(M, 64);
    return res ? launch_pw<2, 256, 64, true>(k, st) : launch_pw<2, 256, 64, false>(k, st);
  }
  k.tiles_m = pm_cdiv(M, 32);
  return res ? launch_pw<4, 128, 32, true>(k, st) : launch_pw<4, 128, 32, false>(k, st);
}
