// Streaming pointwise GEMM for the short reductions of the fp32 tier (round 6): C[M x N] = A[M x K] . B^T with K = 64 or 128 -- the 1x1 convolutions of layer1 / layer2
// (64 -> 256 @192^2, 128 -> 512 @96^2, Resnet.py:145-150) and their stride-1 data gradients. These launches move 4 - 5 bytes per FLOP-pair more than the tile kernels
// were built for: 75 MB in and 302 MB out at 9.7 GFLOP. The tile kernels reach 2.5 - 2.8 TB/s on them (a block loads, then multiplies, then stores; two to seven
// blocks per CU are not enough to keep all three going), a plain read-modify-write stream reaches 6 TB/s on the same bytes.
//
// Here every WAVE is its own pipeline and there is no barrier after the prologue:
//   * the weight slab of a block (64 columns x K) is split once into the three bf16 planes (x == hi + mid + lo exactly, conv_igemm_kernel.h) and parked in LDS in MFMA
//     B-fragment order (one conflict-free ds_read_b128 per fragment);
//   * a wave walks 32-row tiles of A. Each lane loads its row's K floats straight into the A-fragment registers (16 bytes per load, two loads per 16-k group -- the k
//     order inside a group is permuted the same way on both operands, which a dot product does not see), for the NEXT tile while the current one is multiplied;
//   * the split to bf16 happens in registers (5.5 VALU per element), six products per fragment pair on v_mfma_f32_32x32x16_bf16 into fp32 accumulators -- the same
//     arithmetic and product order as the split tile kernel;
//   * the weights are the FIRST MFMA operand, so a lane's accumulator quads are four consecutive columns of its row; a column block takes a turn through the wave's
//     own LDS patch and leaves as 16-byte stores of whole 128-byte lines (dword stores straight from the accumulators: 105 us instead of 94 on 64 -> 256 @192^2).
// The blocks of the column slabs of one row range sit on one XCD (block b -> XCD b mod 8), so A passes through one L2.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "pm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct PwsK {
  const float* A;            // rows of K floats, a_pitch floats apart
  const float* B;            // element (n, k) at B[n * b_sn + k * b_sk]
  float* C;                  // rows of Nn floats, c_pitch floats apart
  const float *bias, *scale, *shift, *residual;      // epilogue of the tile kernels: v = (acc + bias) * scale + shift (+ residual) (relu)
  long a_pitch, c_pitch, res_pitch;
  unsigned a_bytes, c_bytes, res_bytes;               // buffer extents: rows >= M read zeros / are not stored
  int b_sn, b_sk;
  int M, Nn, row_tiles, relu;
};

__device__ __forceinline__ v4f pws_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

// four fp32 -> 3 x four bf16 by truncation (exact: hi + mid + lo == x), as in the split tile kernel
__device__ __forceinline__ void pws_split4(const v4f& v, unsigned* hi, unsigned* mid, unsigned* lo) {
  const unsigned u0 = __float_as_uint(v.x), u1 = __float_as_uint(v.y), u2 = __float_as_uint(v.z), u3 = __float_as_uint(v.w);
  hi[0] = __builtin_amdgcn_perm(u1, u0, 0x07060302), hi[1] = __builtin_amdgcn_perm(u3, u2, 0x07060302);
  const float r0 = v.x - __uint_as_float(u0 & 0xffff0000u), r1 = v.y - __uint_as_float(u1 & 0xffff0000u);
  const float r2 = v.z - __uint_as_float(u2 & 0xffff0000u), r3 = v.w - __uint_as_float(u3 & 0xffff0000u);
  const unsigned q0 = __float_as_uint(r0), q1 = __float_as_uint(r1), q2 = __float_as_uint(r2), q3 = __float_as_uint(r3);
  mid[0] = __builtin_amdgcn_perm(q1, q0, 0x07060302), mid[1] = __builtin_amdgcn_perm(q3, q2, 0x07060302);
  const float s0 = r0 - __uint_as_float(q0 & 0xffff0000u), s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
  const float s2 = r2 - __uint_as_float(q2 & 0xffff0000u), s3 = r3 - __uint_as_float(q3 & 0xffff0000u);
  lo[0] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
  lo[1] = __builtin_amdgcn_perm(__float_as_uint(s3), __float_as_uint(s2), 0x07060302);
}

// KG: 16-k groups (K = 16 * KG). Lane (l31, h) of a fragment holds, for group m and element i, k = 16 m + 8 (i >> 2) + 4 h + (i & 3): the two 16-byte loads
// (k offsets 16 m + 4 h and 16 m + 8 + 4 h) of its row -- on both operands.
// NB: register sets of A rows per wave: the set being multiplied and NB - 1 tiles in flight behind it.
template <int KG, int NB>
__global__ __launch_bounds__(256, 2) void pwstream_kernel(PwsK a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];      // [KG][2 column blocks][3 planes][64 lanes] x 16 bytes
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;      // (wave: uniform -> the tile walk is scalar)
  const int slabs = a.Nn >> 6;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int slab = j % slabs, grp = xcd + 8 * (j / slabs), groups = 8 * ((int)(gridDim.x >> 3) / slabs);
  const int n0 = slab * 64;

  // ---- prologue: this block's 64 weight columns, split, in fragment order ----
  {
    const int cb = (t >> 6) & 1, n = n0 + cb * 32 + l31;
#pragma unroll
    for (int e = 0; e < KG / 2; ++e) {
      const int m = (t >> 7) + 2 * e;
      v4f w0, w1;
      const float* bp = a.B + (long)n * a.b_sn + (long)(16 * m + 4 * h) * a.b_sk;
      w0.x = bp[0], w0.y = bp[a.b_sk], w0.z = bp[2 * a.b_sk], w0.w = bp[3 * a.b_sk];
      bp += 8 * a.b_sk;
      w1.x = bp[0], w1.y = bp[a.b_sk], w1.z = bp[2 * a.b_sk], w1.w = bp[3 * a.b_sk];
      unsigned hi[4], mid[4], lo[4];
      pws_split4(w0, hi, mid, lo), pws_split4(w1, hi + 2, mid + 2, lo + 2);
      char* dst = lds + ((m * 2 + cb) * 3) * 1024 + lane * 16;
      *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
      *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
  }
  // the slab's epilogue constants behind the planes: bias | scale | shift, 64 floats each
  float* epi = reinterpret_cast<float*>(lds + KG * 2 * 3 * 1024);
  if (t < 64) epi[t] = a.bias ? a.bias[n0 + t] : 0.f;
  else if (t < 128) epi[t] = a.scale ? a.scale[n0 + t - 64] : 1.f;
  else if (t < 192) epi[t] = a.scale ? a.shift[n0 + t - 128] : 0.f;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A), 0, (int)a.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, (int)a.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.residual ? a.residual : a.A), 0, (int)(a.residual ? a.res_bytes : 0u), 0x00020000);
  const unsigned a_row = (unsigned)(a.a_pitch * 4), c_row = (unsigned)(a.c_pitch * 4), r_row = (unsigned)(a.res_pitch * 4);
  const bool aff = a.bias || a.scale, res = a.residual != nullptr, relu = a.relu != 0;
  const int first = grp * 4 + wave, stride = groups * 4;
  v4f raw[NB][2 * KG];
  auto load = [&](v4f* dst, int tile) {
    const unsigned off = (unsigned)(tile * 32 + l31) * a_row + (unsigned)h * 16u;
#pragma unroll
    for (int q = 0; q < 2 * KG; ++q) dst[q] = pws_load(rA, off + (unsigned)q * 32u);
  };
  auto tile_out = [&](const v4f* src, int tile) {
    asm volatile("" ::: "memory");      // the weight fragments are re-read from LDS per tile: hoisted out of the tile loop they would pin 24 KG registers
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
    for (int m = 0; m < KG; ++m) {
      if (m) asm volatile("" ::: "memory");      // ... and the fragments of one 16-k group at a time: requested here, they land under this group's split arithmetic
      unsigned ah[4], am[4], al[4];
      pws_split4(src[2 * m], ah, am, al), pws_split4(src[2 * m + 1], ah + 2, am + 2, al + 2);
      const bf16x8 fah = __builtin_bit_cast(bf16x8, make_uint4(ah[0], ah[1], ah[2], ah[3]));
      const bf16x8 fam = __builtin_bit_cast(bf16x8, make_uint4(am[0], am[1], am[2], am[3]));
      const bf16x8 fal = __builtin_bit_cast(bf16x8, make_uint4(al[0], al[1], al[2], al[3]));
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const char* src_b = lds + ((m * 2 + cb) * 3) * 1024 + lane * 16;
        const bf16x8 fbh = *reinterpret_cast<const bf16x8*>(src_b);
        const bf16x8 fbm = *reinterpret_cast<const bf16x8*>(src_b + 1024);
        const bf16x8 fbl = *reinterpret_cast<const bf16x8*>(src_b + 2048);
        // weights as the first operand: the accumulator then holds, per lane, FOUR CONSECUTIVE COLUMNS of the lane's row in every register quad -> 16-byte stores.
        // smallest terms first, as in the tile kernel: lo hi, hi lo, mid mid, mid hi, hi mid, hi hi (a = activation piece, b = weight piece)
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbh, fal, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbl, fah, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbm, fam, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbh, fam, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbm, fah, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbh, fah, acc[cb], 0, 0, 0);
      }
    }
    // accumulator registers 4 j ... 4 j + 3 of lane (l31, h): row l31 of the tile, columns 8 j + 4 h ... + 3 of the 32-column block. Straight from there a 16-byte
    // store would touch 32 rows x 32 bytes (measured: slower than dword stores), so each column block takes a turn through this wave's own 32 x 144-byte LDS patch
    // (no barrier: a wave's LDS operations execute in order) and leaves as 16-byte stores of 8 rows x 128 contiguous bytes -- whole cache lines, a quarter of the
    // store instructions of the dword form (105 -> 94 us on 64 -> 256 @192^2; without any store the kernel runs in 54 us: the write stream is the bound). Rows beyond M (a last, partial
    // tile) fall outside the buffer extents: read as zero, not stored.
    char* stg = lds + KG * 2 * 3 * 1024 + 3 * 64 * 4 + wave * (32 * 144);
    const int rr = lane >> 3, cg = lane & 7;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<v4f*>(stg + l31 * 144 + (8 * j + 4 * h) * 4) = v4f{acc[cb][4 * j], acc[cb][4 * j + 1], acc[cb][4 * j + 2], acc[cb][4 * j + 3]};
      const int c = cb * 32 + cg * 4;
      v4f bi, sc, sh;
      if (aff) bi = *reinterpret_cast<const v4f*>(epi + c), sc = *reinterpret_cast<const v4f*>(epi + 64 + c), sh = *reinterpret_cast<const v4f*>(epi + 128 + c);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const unsigned row = (unsigned)(tile * 32 + rr + 8 * ps);
        v4f v = *reinterpret_cast<const v4f*>(stg + (rr + 8 * ps) * 144 + cg * 16);
        if (aff) v = (v + bi) * sc + sh;
        if (res) v += __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)(row * r_row + (unsigned)(n0 + c) * 4u), 0, 0));
        if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rC, (int)(row * c_row + (unsigned)(n0 + c) * 4u), 0, 0);
      }
    }
  };

  int tile = first;
#pragma unroll
  for (int b = 0; b < NB - 1; ++b)
    if (tile + b * stride < a.row_tiles) load(raw[b], tile + b * stride);
  while (tile < a.row_tiles) {      // NB tiles per trip: the register sets rotate roles without a copy
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int ahead = tile + (NB - 1) * stride;
      if (ahead < a.row_tiles) load(raw[(b + NB - 1) % NB], ahead);
      tile_out(raw[b], tile);
      tile += stride;
      if (tile >= a.row_tiles) break;
    }
  }
}

template <int KG, int NB>
void pws_launch(const PwsK& k, int grid, hipStream_t st) {
  const size_t smem = (size_t)KG * 2 * 3 * 1024 + 3 * 64 * sizeof(float) + 4 * 32 * 144;      // weight planes | epilogue constants | one staging patch per wave
  pm_device_once([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pwstream_kernel<KG, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); });      // K = 128: 67 KB
  hipLaunchKernelGGL((pwstream_kernel<KG, NB>), dim3(grid), dim3(256), smem, st, k);
}

}  // namespace

// Does this pointwise GEMM take the streaming kernel? K = 64 / 128 floats per (contiguous, 16-byte aligned) row, N a multiple of 64 with N / 64 in {1, 2, 4, 8}, enough rows
// to give every wave of the 512 blocks a few tiles, 32-bit byte offsets.
bool pm_pwstream_ok(const pm_gemm_pw* g) {
  static const int on = getenv("PM_PWSTREAM") ? atoi(getenv("PM_PWSTREAM")) : 1;
  static const long min_rows = getenv("PM_PWSTREAM_MIN_ROWS") ? atol(getenv("PM_PWSTREAM_MIN_ROWS")) : 65536;
  if (!on || (g->K != 64 && g->K != 128)) return false;
  const int slabs = g->Nn / 64;
  if (g->Nn % 64 || (slabs != 1 && slabs != 2 && slabs != 4 && slabs != 8)) return false;
  if (g->M < min_rows || (g->a_pitch & 3) || ((uintptr_t)g->A & 15) || (g->c_pitch & 3) || ((uintptr_t)g->C & 15)) return false;
  if (g->residual && ((g->res_pitch & 3) || ((uintptr_t)g->residual & 15))) return false;
  if ((double)g->M * (double)g->a_pitch * 4.0 >= 4.0e9 || (double)g->M * (double)g->c_pitch * 4.0 >= 4.0e9) return false;
  if (g->residual && (double)g->M * (double)g->res_pitch * 4.0 >= 4.0e9) return false;
  return true;
}

int pm_pwstream_launch(const pm_gemm_pw* g, hipStream_t st) {
  PwsK k;
  k.A = g->A, k.B = g->B, k.C = g->C, k.bias = g->bias, k.scale = g->scale, k.shift = g->shift, k.residual = g->residual;
  k.a_pitch = g->a_pitch, k.c_pitch = g->c_pitch, k.res_pitch = g->res_pitch;
  k.a_bytes = (unsigned)((g->M - 1) * g->a_pitch * 4 + (long)g->K * 4);
  k.c_bytes = (unsigned)((g->M - 1) * g->c_pitch * 4 + (long)g->Nn * 4);
  k.res_bytes = g->residual ? (unsigned)((g->M - 1) * g->res_pitch * 4 + (long)g->Nn * 4) : 0u;
  k.b_sn = g->b_sn, k.b_sk = g->b_sk, k.M = (int)g->M, k.Nn = g->Nn, k.row_tiles = (int)((g->M + 31) / 32), k.relu = g->relu;

  // 512 blocks = two per CU, one tile in flight behind the one being multiplied: three / four blocks per CU and two / three tiles in flight measured level
  // (94 - 102 us on 64 -> 256 @192^2; profiles/README.md round 6) -- the kernel is bound by its write stream, not by load latency
  const int slabs = g->Nn / 64, grid = 8 * slabs * std::max(1, 64 / slabs);
  if (g->K == 64) pws_launch<4, 2>(k, grid, st);
  else pws_launch<8, 2>(k, grid, st);
  return pm_check_launch("pwstream");
}
