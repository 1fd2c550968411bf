// Error reporting, elementwise helpers, layout edges and the flat-arena SGD step.
#include <algorithm>

#include "pm_common.h"

static thread_local char g_err[512] = "";

void pm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* pm_last_error(void) { return g_err; }
extern "C" int pm_version(void) { return PM_ABI_VERSION; }

#define LD4 PM_LD4
#define ST4 PM_ST4
#define ew_launch pm_ew_launch
#define same_shape pm_same_shape
#define vec4 pm_vec4

extern "C" int pm_relu_bwd(const pm_tensor* dy, const pm_tensor* y, const pm_tensor* dx, void* stream) {
  PM_REQUIRE(dy && y && dx && same_shape(dy, y) && same_shape(dy, dx), PM_EINVAL, "relu_bwd: shape mismatch");
  PM_REQUIRE_F32(dy, "relu_bwd"); PM_REQUIRE_F32(y, "relu_bwd"); PM_REQUIRE_F32(dx, "relu_bwd");
  const float *pdy = (const float*)dy->ptr, *py = (const float*)y->ptr;
  float* pdx = (float*)dx->ptr;
  const long a = dy->pitch, b = y->pitch, c = dx->pitch;
  const bool v = vec4(dy) && vec4(y) && vec4(dx);
  if (v)
    return ew_launch(true, pm_pixels(dy), dy->c, (hipStream_t)stream, "relu_bwd", [=] __device__(long p, int ch) {
      float4 g = LD4(pdy + p * a + ch), o = LD4(py + p * b + ch);
      g.x = o.x > 0.f ? g.x : 0.f, g.y = o.y > 0.f ? g.y : 0.f, g.z = o.z > 0.f ? g.z : 0.f, g.w = o.w > 0.f ? g.w : 0.f;
      ST4(pdx + p * c + ch, g);
    });
  return ew_launch(false, pm_pixels(dy), dy->c, (hipStream_t)stream, "relu_bwd",
                   [=] __device__(long p, int ch) { pdx[p * c + ch] = py[p * b + ch] > 0.f ? pdy[p * a + ch] : 0.f; });
}

extern "C" int pm_add(const pm_tensor* x, const pm_tensor* y, const pm_tensor* o, void* stream) {
  PM_REQUIRE(x && y && o && same_shape(x, y) && same_shape(x, o), PM_EINVAL, "add: shape mismatch");
  if (pm_is_bf16(o)) {
    const pm_tensor* two[2] = {x, y};
    return pm16_add_n(two, 2, o, (hipStream_t)stream);
  }
  PM_REQUIRE_F32(x, "add"); PM_REQUIRE_F32(y, "add");
  const float *px = (const float*)x->ptr, *py = (const float*)y->ptr;
  float* po = (float*)o->ptr;
  const long a = x->pitch, b = y->pitch, c = o->pitch;
  if (vec4(x) && vec4(y) && vec4(o))
    return ew_launch(true, pm_pixels(x), x->c, (hipStream_t)stream, "add", [=] __device__(long p, int ch) {
      float4 u = LD4(px + p * a + ch), w = LD4(py + p * b + ch);
      ST4(po + p * c + ch, make_float4(u.x + w.x, u.y + w.y, u.z + w.z, u.w + w.w));
    });
  return ew_launch(false, pm_pixels(x), x->c, (hipStream_t)stream, "add",
                   [=] __device__(long p, int ch) { po[p * c + ch] = px[p * a + ch] + py[p * b + ch]; });
}

// o = x[0] + x[1] + ... + x[n-1] (2 <= n <= 8, same shapes, float4 views), summed left to right in one pass: the gradient of a
// tensor with several consumers (the ASPP input feeds five branches) without a chain of two-operand adds
struct AddN {
  const float* p[8];
  long pitch[8];
};
extern "C" int pm_add_n(const pm_tensor* const* xs, int n, const pm_tensor* o, void* stream) {
  PM_REQUIRE(xs && o && n >= 2 && n <= 8, PM_EINVAL, "add_n: 2..8 operands");
  if (pm_is_bf16(o)) return pm16_add_n(xs, n, o, (hipStream_t)stream);
  AddN a;
  for (int i = 0; i < n; ++i) {
    PM_REQUIRE_F32(xs[i], "add_n");
    PM_REQUIRE(xs[i] && same_shape(xs[i], o) && vec4(xs[i]), PM_EINVAL, "add_n: operand %d: shape / float4 view mismatch", i);
    a.p[i] = (const float*)xs[i]->ptr, a.pitch[i] = xs[i]->pitch;
  }
  PM_REQUIRE(vec4(o), PM_EINVAL, "add_n: output must be a float4 view");
  float* po = (float*)o->ptr;
  const long c = o->pitch;
  return ew_launch(true, pm_pixels(o), o->c, (hipStream_t)stream, "add_n", [=] __device__(long p, int ch) {
    float4 s = LD4(a.p[0] + p * a.pitch[0] + ch);
    for (int i = 1; i < n; ++i) {
      const float4 v = LD4(a.p[i] + p * a.pitch[i] + ch);
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    ST4(po + p * c + ch, s);
  });
}

extern "C" int pm_copy(const pm_tensor* x, const pm_tensor* o, void* stream) {
  PM_REQUIRE(x && o && same_shape(x, o), PM_EINVAL, "copy: shape mismatch");
  if (pm_is_bf16(x) && pm_is_bf16(o)) return pm16_copy(x, o, (hipStream_t)stream);
  PM_REQUIRE_F32(x, "copy"); PM_REQUIRE_F32(o, "copy");
  const float* px = (const float*)x->ptr;
  float* po = (float*)o->ptr;
  const long a = x->pitch, c = o->pitch;
  if (vec4(x) && vec4(o))
    return ew_launch(true, pm_pixels(x), x->c, (hipStream_t)stream, "copy",
                     [=] __device__(long p, int ch) { ST4(po + p * c + ch, LD4(px + p * a + ch)); });
  return ew_launch(false, pm_pixels(x), x->c, (hipStream_t)stream, "copy", [=] __device__(long p, int ch) { po[p * c + ch] = px[p * a + ch]; });
}

// y = relu?(x*scale[c] + shift[c] + residual?)  -- eval-mode BN fold applied to an existing tensor
extern "C" int pm_scale_shift_act(const pm_tensor* x, const float* scale, const float* shift, const pm_tensor* res, int relu,
                                  const pm_tensor* o, void* stream) {
  PM_REQUIRE(x && o && scale && shift && same_shape(x, o) && (!res || same_shape(x, res)), PM_EINVAL, "scale_shift_act: bad args");
  PM_REQUIRE_F32(x, "scale_shift_act"); PM_REQUIRE_F32(o, "scale_shift_act"); PM_REQUIRE_F32(res, "scale_shift_act");
  const float *px = (const float*)x->ptr, *pr = res ? (const float*)res->ptr : nullptr;
  float* po = (float*)o->ptr;
  const long a = x->pitch, b = res ? res->pitch : 0, c = o->pitch;
  return ew_launch(false, pm_pixels(x), x->c, (hipStream_t)stream, "scale_shift_act", [=] __device__(long p, int ch) {
    float v = px[p * a + ch] * scale[ch] + shift[ch];
    if (pr) v += pr[p * b + ch];
    po[p * c + ch] = relu ? fmaxf(v, 0.f) : v;
  });
}

// ---- layout edges ----------------------------------------------------------------------------------------------
namespace {
// 32x32 LDS-tiled transpose between [C][P] planes and [P][pitch] rows (per image), coalesced on both sides.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int csrc, long P, float* __restrict__ y, int c, long pitch) {
  __shared__ float tile[32][33];
  const int img = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int ch = c0 + j;
    const long p = p0 + tx;
    tile[j][tx] = (ch < csrc && p < P) ? x[((long)img * csrc + ch) * P + p] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const long p = p0 + j;
    const int ch = c0 + tx;
    if (p < P && ch < c) y[((long)img * P + p) * pitch + ch] = tile[tx][j];
  }
}
// The image edge (3 planes -> 4-channel rows, zero pad): thread = pixel, one coalesced 4-byte read per plane, one 16-byte row out. The tiled transpose above
// spends a 32 x 32 tile on 3 useful plane rows: 128 us for the 8 x 3 x 768^2 batch of the flagship step (33 MB: ~10 us at HBM rate).
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, int csrc, long P, long total, float* __restrict__ y, int c, long pitch) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long img = i / P, p = i - img * P;
    const float* src = x + img * csrc * P + p;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = k < csrc ? src[(long)k * P] : 0.f;
    float* dst = y + i * pitch;
    if (c == 4) {
      PM_ST4(dst, make_float4(v[0], v[1], v[2], v[3]));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < c) dst[k] = v[k];
    }
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, int c, long pitch, long P, float* __restrict__ y) {
  __shared__ float tile[32][33];
  const int img = blockIdx.z;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const long p = p0 + j;
    const int ch = c0 + tx;
    tile[j][tx] = (p < P && ch < c) ? x[((long)img * P + p) * pitch + ch] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int ch = c0 + j;
    const long p = p0 + tx;
    if (ch < c && p < P) y[((long)img * c + ch) * P + p] = tile[tx][j];
  }
}
__global__ void label_nearest_kernel(const int64_t* __restrict__ lab, int n, int H, int W, int64_t* __restrict__ out, int h, int w, float sy, float sx) {
  const long total = (long)n * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long)w * h));
    // ATen nearest: src = min(floor(dst * scale), in-1), scale = in/out in float
    int yy = min((int)floorf((float)y * sy), H - 1), xx = min((int)floorf((float)x * sx), W - 1);
    out[i] = lab[((long)b * H + yy) * W + xx];
  }
}
}  // namespace

extern "C" int pm_nchw_to_nhwc(const float* x, int c_src, const pm_tensor* y, void* stream) {
  PM_REQUIRE(x && y && y->ptr && c_src <= y->c, PM_EINVAL, "nchw_to_nhwc: bad args");
  PM_REQUIRE_F32(y, "nchw_to_nhwc");
  const long P = (long)y->h * y->w;
  if (y->c <= 4 && c_src <= 4 && (y->c != 4 || (y->pitch % 4 == 0 && pm_aligned16(y->ptr)))) {      // image edge: thread per pixel
    const long total = P * y->n;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, x, c_src, P, total,
                       (float*)y->ptr, y->c, (long)y->pitch);
    return pm_check_launch("nchw_to_nhwc");
  }
  dim3 grid(pm_cdiv(P, 32), pm_cdiv(y->c, 32), y->n);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, c_src, P, (float*)y->ptr, y->c, (long)y->pitch);
  return pm_check_launch("nchw_to_nhwc");
}
extern "C" int pm_nhwc_to_nchw(const pm_tensor* x, float* y, void* stream) {
  PM_REQUIRE(x && y && x->ptr, PM_EINVAL, "nhwc_to_nchw: bad args");
  PM_REQUIRE_F32(x, "nhwc_to_nchw");
  const long P = (long)x->h * x->w;
  dim3 grid(pm_cdiv(P, 32), pm_cdiv(x->c, 32), x->n);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x->ptr, x->c, (long)x->pitch, P, y);
  return pm_check_launch("nhwc_to_nchw");
}
extern "C" int pm_label_nearest(const int64_t* lab, int n, int H, int W, int64_t* out, int h, int w, void* stream) {
  PM_REQUIRE(lab && out && h > 0 && w > 0, PM_EINVAL, "label_nearest: bad args");
  const long total = (long)n * h * w;
  hipLaunchKernelGGL(label_nearest_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, lab, n, H, W, out, h, w,
                     (float)H / (float)h, (float)W / (float)w);
  return pm_check_launch("label_nearest");
}

// ---- SGD with momentum over a flat arena (torch.optim.SGD semantics, optimizer.py:21-25) --------------------------
namespace {
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, long n, float lr, float mom,
                                                  float wd, int first) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float w = p[i];
    const float d = g[i] + wd * w;
    const float b = first ? d : mom * m[i] + d;
    m[i] = b;
    p[i] = w - lr * b;
  }
}
// Multi-tensor form: a handful of launches update every parameter of the network (161 tensors, 45.3 M floats for R50-DeepLabV3+).
// The tensor table travels BY VALUE in the kernel arguments (32 records per launch): no device table, no host-to-device copy, nothing
// to keep alive -- gradients are fresh tensors after every zero_grad(set_to_none=True). A block takes chunks of SGD_CHUNK elements;
// chunk_start[t] is the first chunk of record t (prefix sum), found by bisection.
constexpr int SGD_CHUNK = 4096, SGD_BATCH = 32;
// torch.optim.SGD's roundings exactly: d = fma(wd, p, g) [grad.add(p, alpha=wd)]; buf = round(momentum * buf) + d [buf.mul_(m).add_(d)];
// p = fma(-lr, buf, p) [p.add_(buf, alpha=-lr)] -- pinned with the _rn intrinsics so that -ffp-contract cannot merge the middle pair
__device__ __forceinline__ float sgd_buf(float mom, float b0, float g, float wd, float w) { return __fadd_rn(__fmul_rn(mom, b0), __fmaf_rn(wd, w, g)); }
struct SgdBatch {
  pm_sgd_entry e[SGD_BATCH];
  int chunk_start[SGD_BATCH + 1];
  int n;
};
__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdBatch tab, float lr, float mom, float wd, const float* __restrict__ lr_dev) {
  if (lr_dev) lr = *lr_dev;      // learning rate from device memory: a captured (hipGraph) step follows the schedule without being re-captured
  const int total_chunks = tab.chunk_start[tab.n];
  for (int c = blockIdx.x; c < total_chunks; c += gridDim.x) {
    int lo = 0, hi = tab.n;                        // largest t with chunk_start[t] <= c
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (tab.chunk_start[mid] <= c) lo = mid;
      else hi = mid;
    }
    float* P = tab.e[lo].param;
    const float* G = tab.e[lo].grad;
    float* Mb = tab.e[lo].momentum_buffer;
    const long n = tab.e[lo].numel;
    const long i0 = (long)(c - tab.chunk_start[lo]) * SGD_CHUNK, i1 = min(n, i0 + SGD_CHUNK);
    const bool v4 = ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Mb)) & 15) == 0;
    if (v4 && i1 - i0 == SGD_CHUNK) {
#pragma unroll
      for (int u = 0; u < SGD_CHUNK / 1024; ++u) {
        const long i = i0 + u * 1024 + threadIdx.x * 4;
        const float4 w = LD4(P + i), g = LD4(G + i), b0 = LD4(Mb + i);
        float4 b, o;
        b.x = sgd_buf(mom, b0.x, g.x, wd, w.x), b.y = sgd_buf(mom, b0.y, g.y, wd, w.y), b.z = sgd_buf(mom, b0.z, g.z, wd, w.z), b.w = sgd_buf(mom, b0.w, g.w, wd, w.w);
        o.x = __fmaf_rn(-lr, b.x, w.x), o.y = __fmaf_rn(-lr, b.y, w.y), o.z = __fmaf_rn(-lr, b.z, w.z), o.w = __fmaf_rn(-lr, b.w, w.w);
        ST4(Mb + i, b);
        ST4(P + i, o);
      }
    } else {
      for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float w = P[i];
        const float b = sgd_buf(mom, Mb[i], G[i], wd, w);
        Mb[i] = b;
        P[i] = __fmaf_rn(-lr, b, w);
      }
    }
  }
}
}  // namespace
extern "C" int pm_sgd_momentum_multi_dev(const pm_sgd_entry* entries, int n, float lr, const float* lr_dev, float momentum, float wd, void* stream);
extern "C" int pm_sgd_momentum_multi(const pm_sgd_entry* entries, int n, float lr, float momentum, float wd, void* stream) {
  return pm_sgd_momentum_multi_dev(entries, n, lr, nullptr, momentum, wd, stream);
}
extern "C" int pm_sgd_momentum_multi_dev(const pm_sgd_entry* entries, int n, float lr, const float* lr_dev, float momentum, float wd, void* stream) {
  PM_REQUIRE(entries && n >= 0, PM_EINVAL, "sgd_multi: bad args");
  for (int base = 0; base < n; base += SGD_BATCH) {
    SgdBatch b;
    b.n = std::min(SGD_BATCH, n - base);
    b.chunk_start[0] = 0;
    for (int t = 0; t < b.n; ++t) {
      const pm_sgd_entry& e = entries[base + t];
      PM_REQUIRE(e.param && e.grad && e.momentum_buffer && e.numel >= 0, PM_EINVAL, "sgd_multi: null tensor in record %d", base + t);
      b.e[t] = e;
      b.chunk_start[t + 1] = b.chunk_start[t] + (int)((e.numel + SGD_CHUNK - 1) / SGD_CHUNK);
    }
    const int total = b.chunk_start[b.n];
    if (total == 0) continue;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(std::min(total, 256 * 16)), dim3(256), 0, (hipStream_t)stream, b, lr, momentum, wd, lr_dev);
  }
  return pm_check_launch("sgd_momentum_multi");
}
extern "C" int pm_sgd_momentum(float* param, const float* grad, float* mbuf, int64_t n, float lr, float momentum, float wd, int first_step,
                               void* stream) {
  PM_REQUIRE(param && grad && mbuf && n >= 0, PM_EINVAL, "sgd: bad args");
  if (n == 0) return PM_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3((int)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, param, grad, mbuf, (long)n, lr,
                     momentum, wd, first_step);
  return pm_check_launch("sgd_momentum");
}

// ---- input edge: uint8 images / labels -> the layouts the path consumes ---------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void image_u8_kernel(const uint8_t* __restrict__ img, long pixels, float m0, float m1, float m2, float s0, float s1,
                                                       float s2, float* __restrict__ out) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long)gridDim.x * 256) {
    const uint8_t* q = img + p * 3;
    // ToTensor: /255 ; Normalize: (x - mean) / std  -- same operation order as torchvision
    const float4 o = make_float4(((float)q[0] / 255.f - m0) / s0, ((float)q[1] / 255.f - m1) / s1, ((float)q[2] / 255.f - m2) / s2, 0.f);
    ST4(out + p * 4, o);
  }
}
__global__ __launch_bounds__(256) void labels_u8_kernel(const uint8_t* __restrict__ lab, long n, int64_t* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (int64_t)lab[i];
}
}  // namespace
extern "C" int pm_image_u8_to_nhwc4(const uint8_t* img, int64_t pixels, const float* mean3, const float* std3, float* out, void* stream) {
  PM_REQUIRE(img && mean3 && std3 && out && pixels >= 0 && pm_aligned16(out), PM_EINVAL, "image_u8_to_nhwc4: bad args");
  if (pixels == 0) return PM_OK;
  hipLaunchKernelGGL(image_u8_kernel, dim3((int)std::min<long>((pixels + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, img, (long)pixels, mean3[0],
                     mean3[1], mean3[2], std3[0], std3[1], std3[2], out);
  return pm_check_launch("image_u8_to_nhwc4");
}
extern "C" int pm_labels_u8_to_i64(const uint8_t* lab, int64_t n, int64_t* out, void* stream) {
  PM_REQUIRE(lab && out && n >= 0, PM_EINVAL, "labels_u8_to_i64: bad args");
  if (n == 0) return PM_OK;
  hipLaunchKernelGGL(labels_u8_kernel, dim3((int)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, lab, (long)n, out);
  return pm_check_launch("labels_u8_to_i64");
}
