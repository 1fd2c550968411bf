// Shared helpers for libpinmem_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pinmem_hip.h"

void pm_set_error(const char* fmt, ...);

#define PM_REQUIRE(cond, code, ...)      \
  do {                                   \
    if (!(cond)) {                       \
      pm_set_error(__VA_ARGS__);         \
      return (code);                     \
    }                                    \
  } while (0)

static inline int pm_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return PM_ELAUNCH;
  }
  return PM_OK;
}

static inline bool pm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t pm_pixels(const pm_tensor* t) { return (int64_t)t->n * t->h * t->w; }
static inline int pm_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t pm_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Vectorisable NHWC view: base 16B-aligned, pitch % 4 == 0.
static inline bool pm_vec_ok(const pm_tensor* t) { return pm_aligned16(t->ptr) && (t->pitch % 4) == 0; }
static inline bool pm_is_f32(const pm_tensor* t) { return t->dtype == PM_F32; }
static inline bool pm_is_bf16(const pm_tensor* t) { return t->dtype == PM_BF16; }
// every fp32-only entry point starts with this: a bf16 tensor handed to a kernel that would read it as floats must fail loudly
#define PM_REQUIRE_F32(t, who) PM_REQUIRE(!(t) || pm_is_f32(t), PM_EUNSUPPORTED, "%s: fp32 tensors only (got dtype %d)", who, (t)->dtype)
// bf16 views move 16 bytes = 8 channels per lane: base 16B-aligned, pitch % 8 == 0, c % 8 == 0
static inline bool pm_vec8(const pm_tensor* t) { return pm_is_bf16(t) && pm_aligned16(t->ptr) && (t->pitch % 8) == 0 && (t->c % 8) == 0; }

// ---- Winograd F(m x m, 3x3) transforms, m = 2 / 4 (winograd.hip), driven by the conv entry points in conv_igemm.hip ------------
struct pm_wino_geom {
  int N, H, W, d, m, TY, TX;   // tiles per image = d*d sub-lattices x TY x TX tiles of m x m outputs
  long tiles;
};
pm_wino_geom pm_wino_make_geom(int n, int h, int w, int d, int m);
int pm_wino_input_xf(const float* x, long pitch, int C, int Kp, const pm_wino_geom& g, float* V, hipStream_t st);
int pm_wino_filter_xf(const float* w, int Cout, int Cin, int Kp, bool dgrad, int m, float* U, hipStream_t st);
int pm_wino_filter_xf_multi(const float* const* w, float* const* U, const int* cout, const int* cin, const int* kp, const int* m, int n, hipStream_t st);
int pm_wino_output_xf(const float* M, int Cout, const pm_wino_geom& g, float* y, long ypitch, const float* bias, const float* scale, const float* shift,
                      const float* residual, long res_pitch, int relu, hipStream_t st);
int pm_wino_fused_f4(const float* V, const float* U, int Cout, int Kp, const pm_wino_geom& g, float* y, long ypitch, const float* bias, const float* scale,
                     const float* shift, const float* residual, long res_pitch, int relu, hipStream_t st);
int pm_wino_dy_xf(const float* dy, long pitch, int Cout, const pm_wino_geom& g, float* Z, hipStream_t st);
int pm_wino_dw_xf(const float* slab, int ks, int Cout, int Cin, int Kp, int m, float* dw, hipStream_t st);

// ---- bf16 operand preparation of the convolution kernels (bf16.hip; prec = 2) ----------------------------------------------------------
int pm_bf16_cast_rows(const float* x, long pitch, int C, int Cp, long P, void* out, hipStream_t st);
int pm_bf16_cast_weights(const float* w, int Cout, int T, int Cin, int Cp, bool rotate, void* out, hipStream_t st);
struct PmS2Classes {      // the four input-pixel parity classes of a stride-2 data gradient: tap subset (ky0 + 2 i, kx0 + 2 j) and the class' bf16 sub-filter buffer
  int ky0[4], nky[4], kx0[4], nkx[4];
  void* out[4];
};
int pm_bf16_cast_weights_s2(const float* w, int Cout, int kh, int kw, int Cin, int Cp, const PmS2Classes* cl, hipStream_t st);
int pm_bf16_transpose_taps(const float* x, long pitch, int C, int N, int H, int W, int Ho, int Wo, int kh, int kw, int stride, int pad, int dil, void* out,
                           hipStream_t st);

// ---- bf16 activation tier (act16.hip): the bf16 forms of the elementwise / reduction entry points; the extern "C" functions dispatch on dtype ------
typedef unsigned short pm_bf16;
int pm16_bn_stats(const pm_tensor* x, float* moments, float eps, float* mean, float* invstd, float* running_mean, float* running_var, float momentum, void* ws,
                  size_t ws_bytes, hipStream_t st);      // moments != NULL: mean | M2 | count ; else finalise
size_t pm16_bn_workspace(const pm_tensor* x);
int pm16_bn_apply_mask(const pm_tensor* x, const float* mean, const float* invstd, const float* gamma, const float* beta, const pm_tensor* res, int relu,
                       const pm_tensor* y, uint8_t* mask, hipStream_t st);
int pm16_bn_bwd_reduce(const pm_tensor* dy, const pm_tensor* y, const uint8_t* mask, const pm_tensor* x, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, int relu, const pm_tensor* gmask, float* sums, void* ws, size_t ws_bytes, hipStream_t st);
int pm16_bn_bwd_apply(const pm_tensor* dy, const pm_tensor* y, const pm_tensor* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* sums, float count, int relu, const pm_tensor* dx, const pm_tensor* dres, hipStream_t st);
int pm16_add_n(const pm_tensor* const* xs, int n, const pm_tensor* y, hipStream_t st);
int pm16_copy(const pm_tensor* x, const pm_tensor* y, hipStream_t st);
int pm16_maxpool_fwd(const pm_tensor* x, const pm_tensor* y, uint8_t* argmax, hipStream_t st);
int pm16_maxpool_bwd(const pm_tensor* dy, const uint8_t* argmax, const pm_tensor* dx, hipStream_t st);
int pm16_gap_fwd(const pm_tensor* x, const pm_tensor* y, hipStream_t st);
int pm16_gap_bwd(const pm_tensor* dy, const pm_tensor* dx, int accumulate, hipStream_t st);
int pm16_resize_fwd(const pm_tensor* x, const pm_tensor* y, hipStream_t st);
int pm16_resize_bwd(const pm_tensor* dy, const pm_tensor* dx, int accumulate, hipStream_t st);
size_t pm16_resize_bwd_workspace(const pm_tensor* dy, const pm_tensor* dx);
int pm16_resize_bwd_separable(const pm_tensor* dy, const pm_tensor* dx, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
// dense conversions used by the convolution entry points for the few mixed-type call sites of the tier (fp32 logits / image next to bf16 activations)
int pm16_to_f32(const pm_bf16* x, long pitch, int C, long P, float* out, long out_pitch, hipStream_t st);
int pm16_pad_rows(const pm_bf16* x, long pitch, int C, int Cp, long P, pm_bf16* out, hipStream_t st);

// ---- the LDS-DMA bf16 implicit-GEMM convolution (conv16.hip), driven by the conv entry points in conv_igemm.hip ----------------------------------------
struct pm_conv16 {
  const pm_bf16* A;          // bf16 NHWC activations (x, or dy for the data gradient), gathered in place
  const pm_bf16* B;          // bf16 weights [Nn][taps][Cp] (pm_bf16_cast_weights; rotated / transposed for the data gradient)
  void* C;                   // bf16 [M][c_pitch] (c_f32 == 0), fp32 [M][c_pitch] (c_f32 != 0), or the fp32 split-K slabs [ksplit][M][Nn] (ksplit > 1)
  int N, H, W, Ho, Wo;       // input and output pixel grids
  long a_pitch;              // bf16 elements between input pixels
  int Cp;                    // channels per tap, a multiple of 64 (zero pad channels inside the pitch)
  int kh, kw, stride, pad, dil;
  int M, Nn, K, ksteps;      // GEMM extents; K = taps * Cp, ksteps = K / 64
  long c_pitch;
  int c_f32;
  const float *bias, *scale, *shift;
  const void* residual;      // bf16, same shape as C
  long res_pitch;
  int relu;
  float* stats;              // train-mode BatchNorm statistics of the bf16 output: (mean, M2) per 32-row slab and channel (pm_conv_epilogue.bn_partials), or null
  int bm, bn, tiles_m, tiles_n, ksplit, ksteps_per;      // pm_conv16_plan
  long c_split;
  int wide;                  // pm_conv16_plan: 1 = a kernel of conv16w.hip takes the call (one block per CU): 256 x 256 two-stage, or the 256 x 128 / 128 x 256 ring (persistent)
};
void pm_conv16_plan(pm_conv16* k);
int pm_conv16w_launch(const pm_conv16* k, hipStream_t st);
int pm_conv16w_persistent(const pm_conv16* k);      // 1: the plan runs in the persistent producer / consumer form of conv16w.hip
size_t pm_conv16_slab_bytes(const pm_conv16* k);
double pm_conv16_executed_fraction(const pm_conv16* k);
int pm_conv16_launch(const pm_conv16* k, hipStream_t st);

// ---- the weight gradient of the bf16 tier on LDS-DMA (wgrad16.hip): persistent producer / consumer ring over (tile, pixel-range) units ---------------------------------
struct pm_wgrad16 {
  const pm_bf16* X;          // bf16 NHWC input of the convolution
  const pm_bf16* DY;         // bf16 NHWC output gradient
  float* C;                  // fp32 dw [Cout][taps * Cin] (ksplit == 1) or the split-K slabs [ksplit][Cout][taps * Cin]
  int N, H, W, Ho, Wo;
  long x_pitch, dy_pitch;    // bf16 elements between pixels
  int Cin, Cout, kh, kw, stride, pad, dil;
  int M, Nn, P;              // GEMM extents: Cout, taps * Cin, output pixels (the reduction)
  int kper, ksplit;          // pixels per split (a multiple of 64) and number of splits
  long c_split;              // floats between slabs
  int bm, bn, tiles_m, tiles_n;      // pm_wgrad16_plan
};
bool pm_wgrad16_plan(pm_wgrad16* k);      // false: the shape stays with the register-staged kernel
int pm_wgrad16_launch(const pm_wgrad16* k, hipStream_t st);

// pwstream.hip: a pointwise fp32 GEMM C[M x Nn] = A[M x K] . B^T with a short reduction (K = 64 / 128), streamed wave by wave (split-operand bf16 MFMA arithmetic)
struct pm_gemm_pw {
  const float* A;            // rows of K floats, a_pitch floats apart
  const float* B;            // element (n, k) at B[n * b_sn + k * b_sk]
  float* C;                  // rows of Nn floats, c_pitch floats apart
  const float *bias, *scale, *shift, *residual;      // v = (acc + bias) * scale + shift (+ residual) (relu): the tile kernels' epilogue
  long a_pitch, c_pitch, res_pitch;
  int b_sn, b_sk;
  long M;
  int Nn, K, relu;
};
bool pm_pwstream_ok(const pm_gemm_pw* g);
int pm_pwstream_launch(const pm_gemm_pw* g, hipStream_t st);

// One-time setup per DEVICE of a kernel that needs it (the > 64 KB dynamic-LDS opt-in) and the CU count of the current device (persistent kernels size their grid by
// it). ADVICE r5: a `static` per process served the device that happened to be current at the first call -- a later launch on another device asked for 144 KB of LDS
// without the attribute and sized its grid by device 0. The static lives in this template, i.e. once per call site (each passes its own lambda type).
template <typename F>
inline int pm_device_once(F&& setup) {
  static int ncu[32] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
  if (!ncu[dev]) {
    setup();
    int n = 256;
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    ncu[dev] = n > 0 ? n : 256;
  }
  return ncu[dev];
}

// the library's routing state (conv_igemm.hip): one struct, see include/pinmem_hip.h pm_routing
extern pm_routing pm_route;


// ---- device helpers -------------------------------------------------------------------------------------------
// eight bf16 channels (16 bytes) <-> eight floats. Round to nearest even on the way out (v_cvt_pk_bf16_f32).
__device__ __forceinline__ void pm_ld8(const pm_bf16* p, float* v) {
  const uint4 q = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(q.x << 16), v[1] = __uint_as_float(q.x & 0xffff0000u), v[2] = __uint_as_float(q.y << 16), v[3] = __uint_as_float(q.y & 0xffff0000u);
  v[4] = __uint_as_float(q.z << 16), v[5] = __uint_as_float(q.z & 0xffff0000u), v[6] = __uint_as_float(q.w << 16), v[7] = __uint_as_float(q.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned pm_pack_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  bf16x2_t o;
  o[0] = (__bf16)a, o[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, o);
}
__device__ __forceinline__ void pm_st8(pm_bf16* p, const float* v) {
  uint4 q;
  q.x = pm_pack_bf16(v[0], v[1]), q.y = pm_pack_bf16(v[2], v[3]), q.z = pm_pack_bf16(v[4], v[5]), q.w = pm_pack_bf16(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = q;
}
__device__ __forceinline__ float pm_bf16_to_f32(pm_bf16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ pm_bf16 pm_f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

// BatchNorm statistics of one 32-row slab of a bf16 convolution output, taken where the slab is still parked in LDS as fp32 (the staged epilogues of conv16.hip
// and conv_igemm.hip): the values are first rounded to bf16 exactly as they are stored -- the statistics are those of the tensor the BatchNorm will read -- then,
// per column, the mean over the valid rows and M2 around it (two passes over LDS, as the fp32 statistics epilogue). A lane owns eight columns (cc ... cc + 7) of rows
// rr0 + k * RPI; the RPI... 64 / LPR lanes sharing a column group are combined by lane exchanges in a fixed order. Output layout = pm_bn_partials_finalize's:
// stats[(slab * Nn + col) * 2 + {0: mean, 1: M2}]. bi / sc / sh: the fused affine of the epilogue (identity in train mode unless the convolution has a bias).
template <int LDC, int LPR, int RPI>
__device__ __forceinline__ void pm_slab_stats16(const float* Ws, int rr0, int cc, long slab_row0, long M, long Nn, int col, bool cok, const float* bi, const float* sc,
                                                const float* sh, float* stats) {
  const float cnt = (float)max(0l, min(32l, M - slab_row0));
  float s1[8], mu[8], m2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = 0.f, m2[e] = 0.f;
  auto slab_row = [&](int r0, float* v) {
    const float4 v0 = *reinterpret_cast<const float4*>(Ws + (r0 + rr0) * LDC + cc), v1 = *reinterpret_cast<const float4*>(Ws + (r0 + rr0) * LDC + cc + 4);
    const float t[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = pm_bf16_to_f32(pm_f32_to_bf16((t[e] + bi[e]) * sc[e] + sh[e]));
    return slab_row0 + r0 + rr0 < M;
  };
#pragma unroll
  for (int r0 = 0; r0 < 32; r0 += RPI) {
    float v[8];
    const bool rok = slab_row(r0, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] += rok ? v[e] : 0.f;
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] += __shfl_xor(s1[e], o, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) mu[e] = cnt > 0.f ? s1[e] / cnt : 0.f;
#pragma unroll
  for (int r0 = 0; r0 < 32; r0 += RPI) {
    float v[8];
    const bool rok = slab_row(r0, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[e] - mu[e];
      m2[e] += rok ? d * d : 0.f;
    }
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) m2[e] += __shfl_xor(m2[e], o, 64);
  if (rr0 == 0 && cok && cnt > 0.f) {
    float* dst = stats + ((slab_row0 >> 5) * Nn + col) * 2;
#pragma unroll
    for (int e = 0; e < 8; e += 2) *reinterpret_cast<float4*>(dst + 2 * e) = make_float4(mu[e], m2[e], mu[e + 1], m2[e + 1]);
  }
}

__device__ __forceinline__ float pm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float pm_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ATen's align_corners=True source-index math, in fp32 exactly as area_pixel_compute_scale /
// area_pixel_compute_source_index do it (scale = (in-1)/(out-1) in float, src = scale*dst).
struct pm_lerp {
  int i0, i1;
  float w0, w1;
};
__host__ __device__ __forceinline__ float pm_ac_scale(int in, int out) {
  return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
}
__device__ __forceinline__ pm_lerp pm_ac_lerp(float scale, int dst, int in) {
  float src = scale * (float)dst;
  int i0 = (int)src;                       // src >= 0 -> truncation == floor
  if (i0 > in - 1) i0 = in - 1;
  int i1 = i0 + (i0 < in - 1 ? 1 : 0);
  float l1 = src - (float)i0;
  if (l1 < 0.f) l1 = 0.f;
  if (l1 > 1.f) l1 = 1.f;
  pm_lerp r;
  r.i0 = i0;
  r.i1 = i1;
  r.w1 = l1;
  r.w0 = 1.f - l1;
  return r;
}

// ---- generic NHWC elementwise driver: thread -> (pixel, 4 channels) ----------------------------------------------
// VEC=true needs 16B-aligned views with c % 4 == 0 (pm_vec4). f(pixel, channel) is a __device__ lambda.
template <bool VEC, typename F>
__global__ __launch_bounds__(256) void pm_ew_kernel(long pixels, int c, F f) {
  const int cg = VEC ? c / 4 : c;
  const long total = pixels * cg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long p = i / cg;
    const int ch = (int)(i - p * cg) * (VEC ? 4 : 1);
    f(p, ch);
  }
}
static inline int pm_ew_grid(long work) { return (int)std::min<long>((work + 255) / 256, 256 * 16); }
template <typename F>
static inline int pm_ew_launch(bool vec, long pixels, int c, hipStream_t st, const char* name, F f) {
  if (pixels * c == 0) return PM_OK;
  if (vec) hipLaunchKernelGGL((pm_ew_kernel<true, F>), dim3(pm_ew_grid(pixels * c / 4)), dim3(256), 0, st, pixels, c, f);
  else hipLaunchKernelGGL((pm_ew_kernel<false, F>), dim3(pm_ew_grid(pixels * c)), dim3(256), 0, st, pixels, c, f);
  return pm_check_launch(name);
}
static inline bool pm_same_shape(const pm_tensor* a, const pm_tensor* b) { return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c; }
static inline bool pm_vec4(const pm_tensor* t) { return pm_vec_ok(t) && t->c % 4 == 0; }
#define PM_LD4(ptr) (*reinterpret_cast<const float4*>(ptr))
#define PM_ST4(ptr, v) (*reinterpret_cast<float4*>(ptr) = (v))
