// Round 6: the fp32 tier's convolutions on the bf16 matrix pipe -- the PREC 5 instantiations of the implicit-GEMM kernel (conv_igemm_kernel.h), compiled as their
// own translation unit. fp32 operands in HBM, fp32 accumulation and epilogue, fp32 results; inside a block every gathered element is split exactly into three bf16
// pieces (hi + mid + lo == x) as it is stored to LDS and a K-step multiplies six cross products on v_mfma_f32_32x32x16_bf16 (2.5 PF peak: 417 TF of fp32-equivalent
// work at six products, against the 157 TF of v_mfma_f32_32x32x2_f32). Gathers, K-state, split-K, epilogues, BatchNorm statistics: the kernel's own, unchanged.
// Replaces nn.Conv2d forward / input gradient / weight gradient of /root/reference/network/Resnet.py:145-150, deepv3plus.py:72-81,397-414 on the fp32 tier.
#include <stdlib.h>
#include <algorithm>

#include "conv_igemm_kernel.h"

namespace {

template <int MODE, int BM, int BN, int KM, int NST>
void launch_split(const ConvK& k, dim3 grid, size_t smem, hipStream_t st) {
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<MODE, BM, BN, 2, 2, KM, 5, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return true;
  }();
  (void)attr_set;
  const size_t ep_bytes = (size_t)4 * 32 * (BN / 2 + 4) * sizeof(float);      // staged epilogue: four wave slabs
  const size_t bytes = std::max(smem * NST, ep_bytes);
  if constexpr (MODE == MODE_FWD) {
    if (k.stats) {
      static const bool attr_set2 = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<MODE, BM, BN, 2, 2, KM, 5, NST, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        return true;
      }();
      (void)attr_set2;
      hipLaunchKernelGGL((conv_igemm_kernel<MODE, BM, BN, 2, 2, KM, 5, NST, true>), grid, dim3(256), bytes, st, k);
      return;
    }
  }
  hipLaunchKernelGGL((conv_igemm_kernel<MODE, BM, BN, 2, 2, KM, 5, NST>), grid, dim3(256), bytes, st, k);
}

template <int MODE, int BM, int BN, int KM>
void launch_nst(const ConvK& k, dim3 grid, size_t smem, bool nst1, hipStream_t st) {
  if (nst1) launch_split<MODE, BM, BN, KM, 1>(k, grid, smem, st);
  else launch_split<MODE, BM, BN, KM, 2>(k, grid, smem, st);
}

template <int MODE, int BM, int BN>
void launch_km(const ConvK& k, dim3 grid, size_t smem, bool nst1, hipStream_t st) {
  if (k.kmode == K_SMALL) return launch_nst<MODE, BM, BN, K_SMALL>(k, grid, smem, nst1, st);
  if constexpr (MODE != MODE_WGRAD) {
    if (k.kmode == K_FAST && k.kh * k.kw == 1 && k.stride == 1 && k.pad == 0 && !k.sub) return launch_nst<MODE, BM, BN, K_PW>(k, grid, smem, nst1, st);
    if (k.kmode == K_FAST) return launch_nst<MODE, BM, BN, K_FAST>(k, grid, smem, nst1, st);
  }
  launch_nst<MODE, BM, BN, K_MID>(k, grid, smem, nst1, st);
}

template <int MODE>
int launch_tile(int bm, int bn, const ConvK& k, dim3 grid, size_t smem, bool nst1, hipStream_t st) {
  if (bm == 128 && bn == 128) launch_km<MODE, 128, 128>(k, grid, smem, nst1, st);
  else if (bm == 128 && bn == 64) launch_km<MODE, 128, 64>(k, grid, smem, nst1, st);
  else if (bm == 64 && bn == 128) launch_km<MODE, 64, 128>(k, grid, smem, nst1, st);
  else if (bm == 64 && bn == 64) launch_km<MODE, 64, 64>(k, grid, smem, nst1, st);
  else {
    pm_set_error("conv_split: no %d x %d tile", bm, bn);
    return PM_EUNSUPPORTED;
  }
  return PM_OK;
}

}  // namespace

// bytes of ONE LDS stage of the (bm x bn) tile: k-contiguous operands as rows of three 64-byte planes + 16 B, m-contiguous ones as three [32][m + 32] bf16 planes
size_t pm_conv_split_stage_bytes(int mode, int bm, int bn) {
  const bool akc = mode != MODE_WGRAD, bkc = mode == MODE_FWD;
  const size_t a = akc ? (size_t)bm * 52 * 4 : (size_t)3 * BK * (bm + 32) * 2;
  const size_t b = bkc ? (size_t)bn * 52 * 4 : (size_t)3 * BK * (bn + 32) * 2;
  return a + b;
}

int pm_conv_split_launch(int mode, int bm, int bn, const ConvK& k, unsigned gx, unsigned gy, unsigned gz, size_t smem, bool nst1, hipStream_t st) {
  const dim3 grid(gx, gy, gz);
  if (mode == MODE_FWD) return launch_tile<MODE_FWD>(bm, bn, k, grid, smem, nst1, st);
  if (mode == MODE_DGRAD) return launch_tile<MODE_DGRAD>(bm, bn, k, grid, smem, nst1, st);
  return launch_tile<MODE_WGRAD>(bm, bn, k, grid, smem, nst1, st);
}
