/* pinmem_hip.h -- C ABI of libpinmem_hip.so: hand-written gfx950 (MI355X) kernels for the dense hot path of
 * Genie-Kim/PintheMemory (DeepLabV3+/V2 over ResNet, plus the categorical memory read/update).
 *
 * The reference is pure Python on stock torch ops: no FFI exists there. Each entry point below therefore names
 * the reference op sequence (file:line under /root/reference) it replaces; INTEGRATION.md shows the ctypes stub
 * that binds it. Conventions:
 *   - every pointer is a BORROWED device pointer (torch owns the memory); the library never allocates or frees
 *     tensor memory; scratch comes from the caller through (ws, ws_bytes);
 *   - activations are NHWC fp32: pixel-major rows of `c` channels, `pitch` floats between pixels
 *     (pitch >= c lets a kernel write into a channel slice of a wider concat buffer);
 *   - conv weights are KRSC fp32 ([Cout][kh][kw][Cin] == a torch channels_last [Cout,Cin,kh,kw] tensor);
 *   - labels are int64 as the reference's callers provide them (transforms/transforms.py:95-97), 255 = ignore;
 *   - all work is enqueued on `stream` (a hipStream_t), no host synchronisation inside;
 *   - return value: 0 = PM_OK, negative = pm_status; pm_last_error() gives the message (thread-local). */
#ifndef PINMEM_HIP_H
#define PINMEM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pm_status { PM_OK = 0, PM_EINVAL = -1, PM_EWORKSPACE = -2, PM_ELAUNCH = -3, PM_EUNSUPPORTED = -4 } pm_status;

/* Element type of an activation tensor. PM_F32: BASELINE configs[1], the parity path (every entry point). PM_BF16: BASELINE configs[2], the bf16 tier --
 * activations and their gradients are STORED as bf16 between layers (round to nearest even), every reduction / statistic / accumulator stays fp32.
 * Entry points say which they take; all tensor arguments of one call share the type unless stated otherwise; anything else returns PM_EUNSUPPORTED. */
typedef enum pm_dtype { PM_F32 = 0, PM_BF16 = 1 } pm_dtype;
/* pm_tensor.flags */
#define PM_TF_ZERO_PAD64 1      /* caller's promise: channels c .. roundup(c, 64) - 1 of every pixel lie inside `pitch` and hold zeros (a bf16 convolution
                                   input whose channel count is not a multiple of 64 is then gathered in place instead of being copied to a padded buffer) */

typedef struct pm_tensor {      /* NHWC activation view */
  void* ptr;
  int32_t n, h, w, c;
  int64_t pitch;                /* ELEMENTS between consecutive pixels (>= c) */
  int32_t dtype;                /* pm_dtype */
  int32_t flags;                /* PM_TF_* */
} pm_tensor;

/* ABI version of this header (pm_version() returns the library's). The two structs below carry their own size as first member: an entry point that
 * receives a struct built against another header returns PM_EINVAL instead of reading past the caller's object (they grew in rounds 2 and 3).
 * 400 (round 4): pm_tensor carries dtype + flags (bf16 activation tier). */
#define PM_ABI_VERSION 400

typedef struct pm_conv_params { /* nn.Conv2d geometry (square kernels/strides as used by the reference) */
  int32_t struct_size;          /* = sizeof(pm_conv_params) */
  int32_t kh, kw, stride, pad, dil;
  int32_t prec;                 /* 0: fp32 MFMA (exact fp32 chain, parity path, BASELINE configs[1]);
                                   1: fp32 tiles staged in LDS, rounded to bf16 per fragment for v_mfma_f32_32x32x16_bf16, fp32 accumulate and storage;
                                   2: BASELINE configs[2] -- operands converted to bf16 in HBM (one streaming pass) and bf16 tiles in LDS, fp32
                                      accumulate and storage; call sites the form does not cover (3-channel stem, 19-class heads, stride-2 data
                                      gradients, weight gradients) run as prec 1 */
  void* wino_v;                 /* optional caller-owned buffer for the Winograd-transformed input of this convolution (NULL: none).
                                   pm_conv_fwd writes it there instead of its workspace; pm_conv_bwd_weight then reads it instead of
                                   transforming x again. Size from pm_conv_winograd_v_bytes (0 = the layer does not take the route). */
  int64_t wino_v_bytes;
  void* wxf;                    /* optional caller-owned buffer for the TRANSFORMED FILTER pm_conv_fwd derives from w (NULL: none), size from
                                   pm_conv_wxf_bytes: the Winograd U = G w G^T of a layer on that route, or the bf16 copy of the weights with
                                   prec == 2. wxf_valid == 0: the call writes it there instead of its workspace; != 0: the call trusts the
                                   content and skips the transform -- for a caller that knows the weights are unchanged since the call that
                                   filled it (eval-mode forward of step t and training forward of step t + 1). */
  int64_t wxf_bytes;
  int32_t wxf_valid;
} pm_conv_params;

typedef struct pm_conv_epilogue { /* optional fused epilogue of pm_conv_fwd; all pointers may be NULL */
  int64_t struct_size;          /* = sizeof(pm_conv_epilogue) */
  const float* bias;            /* [Cout]  conv bias (deepv3plus.py:417,420,424) */
  const float* scale;           /* [Cout]  folded eval-mode BN: y = conv*scale + shift (mynn.py:8-14 in .eval()) */
  const float* shift;           /* [Cout] */
  const float* residual;        /* NHWC, same n,h,w,c AND dtype as y (a bf16 y takes a bf16 residual: cast the pointer); += (Resnet.py:207) */
  int64_t residual_pitch;       /* elements */
  int32_t relu;                 /* max(.,0) last (Resnet.py:216) */
  float* bn_partials;           /* optional (NULL: none): train-mode BatchNorm statistics of y produced by the epilogue itself -- per 32-row slab of the
                                   GEMM and per channel (mean, M2), float[ceil(pixels / 32)][Cout][2], two-pass inside the slab -- so that no
                                   separate pass re-reads y; merged by pm_bn_partials_finalize. Size from pm_conv_bn_partials_bytes
                                   (0 = this call does not take the route: Winograd, split-K, unaligned / narrow outputs). Needs relu == 0, no residual. */
  int64_t bn_partials_bytes;
} pm_conv_epilogue;

const char* pm_last_error(void);
int pm_version(void);

/* ---- K1/K2/K3 convolution, implicit GEMM on v_mfma_f32_32x32x2_f32 -------------------------------------------
 * Replaces nn.Conv2d forward/backward at Resnet.py:145-150,195,404,453-457; deepv3plus.py:72-81,87,398-424;
 * deepv2.py:44-51,138-151; memory.py:75,104.  y: [n,ho,wo,cout], x: [n,h,w,cin], cin % 4 == 0.
 * dtypes: all PM_F32 = the parity path. The bf16 tier (BASELINE configs[2]) is entered by prec = 2 or by any PM_BF16 tensor: x / y / dy / dx / add may each be
 * bf16 (c % 8 == 0, pitch % 8 == 0) or fp32 (the image in, the class logits and their gradient out); weights, dw, dbias, bias / scale / shift stay fp32,
 * every product is accumulated in fp32 and rounded once on the way out. A bf16 input whose channel count is not a multiple of 64 is gathered in place
 * only with PM_TF_ZERO_PAD64, else it is copied to a zero-padded workspace buffer first. */
size_t pm_conv_workspace(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p, int which /*0 fwd,1 dgrad,2 wgrad*/);
size_t pm_conv_winograd_v_bytes(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p);
size_t pm_conv_wxf_bytes(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p);   /* 0: pm_conv_fwd keeps no transformed filter for this call */
/* the same for pm_conv_bwd_data on the bf16 tier: bytes of the rotated / transposed bf16 filter a stride-1 data gradient derives from w (0: none). The caller
 * passes the buffer in pm_conv_params.wxf / wxf_bytes / wxf_valid exactly as for pm_conv_fwd. */
size_t pm_conv_wxf_bytes_dgrad(const pm_tensor* dy, const pm_tensor* dx, const pm_conv_params* p);
/* bf16 tier: rewrite MANY kept bf16 filters (the buffers of pm_conv_wxf_bytes / _dgrad with prec == 2) from their fp32 weights in one or two launches, e.g.
 * right after the optimizer moved the weights -- the next pm_conv_fwd / pm_conv_bwd_data calls then arrive with wxf_valid = 1 and derive nothing (142 small
 * casts per training step otherwise). Layout written = the one those calls read: forward [cout][kh*kw][round64(cin)], dgrad (rotated / transposed)
 * [cin][kh*kw][round64(cout)] with tap t <- kh*kw-1-t; pad channels zero. wxf_bytes is checked against that size. */
typedef struct pm_wxf_job {
  const float* w;      /* KRSC fp32 weights [cout][kh][kw][cin] */
  void* wxf;           /* the kept buffer */
  int64_t wxf_bytes;
  int32_t cout, kh, kw, cin;
  int32_t dgrad;       /* 0: forward copy, 1: rotated / transposed copy of the data gradient */
  int32_t reserved;
} pm_wxf_job;
int pm_conv_wxf_refresh_bf16(const pm_wxf_job* jobs, int n, void* stream);
/* fp32 tier: the kept Winograd forward transforms (U = G g Gt, what pm_conv_fwd writes into wxf for the wide stride-1 3x3 layers) of all jobs in one launch per
 * 48 filters; kh = kw = 3, dgrad = 0, wxf_bytes = pm_conv_wxf_bytes of the call that owns the buffer (it tells F(4x4) from F(2x2)). Same values as the per-call
 * transform; replaces ~20 latency-bound launches per training step (network/deepv3plus.py:72-81,398-404; Resnet.py:195). */
int pm_conv_wxf_refresh_f32(const pm_wxf_job* jobs, int n, void* stream);
/* bytes of pm_conv_epilogue.bn_partials for this forward call, 0 if the call cannot emit them (then run pm_bn_stats* on y as before) */
size_t pm_conv_bn_partials_bytes(const pm_tensor* x, const pm_tensor* y, const pm_conv_params* p);
int pm_conv_fwd(const pm_tensor* x, const float* w_krsc, const pm_tensor* y, const pm_conv_params* p,
                const pm_conv_epilogue* ep, void* ws, size_t ws_bytes, void* stream);
/* dx = dgrad(dy) [+ add]: `add` (nullable, same shape as dx) fuses the sum with a second gradient path, e.g. the
 * identity/downsample branch of a Bottleneck (Resnet.py:207). */
int pm_conv_bwd_data(const pm_tensor* dy, const float* w_krsc, const pm_tensor* dx, const pm_conv_params* p,
                     const pm_tensor* add, void* ws, size_t ws_bytes, void* stream);
/* dw_krsc [Cout][kh][kw][Cin]; dbias [Cout] or NULL. Deterministic (split-K partials reduced in fixed order). */
int pm_conv_bwd_weight(const pm_tensor* x, const pm_tensor* dy, float* dw_krsc, float* dbias, const pm_conv_params* p,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- ROUTING (round 6): the library's ONLY mutable global state is one pm_routing value -- which kernel takes a call; never WHAT is computed (every route is parity-tested
 * against the same oracle). It is initialised from the PM_* environment when the library is loaded (the defaults below) and replaced as a whole by pm_routing_set: a
 * production caller sets it once before the first pm_conv_* call, or never, and the library is then re-entrant with an immutable kernel table. The pm_set_*
 * entry points further down are thin wrappers that change ONE field -- for kernel tests that must reach a specific kernel and same-box A/B runs. Contract for both: call from
 * one thread while no other thread is inside a pm_conv_* entry point; repeat size queries (pm_conv_workspace, pm_conv_wxf_bytes, ...) after a change. */
typedef struct pm_routing {
  int32_t struct_size;        /* = sizeof(pm_routing): guards against a caller built with another header */
  int32_t winograd;           /* fp32 tier, wide stride-1 3x3s: 4 prefer F(4x4,3x3) (default), 2 F(2x2,3x3) only, 0 direct */
  int32_t winograd_fused;     /* F(4x4) point products + output transform in one kernel: 0 (default; PM_WINO_FUSED), 1 */
  int32_t conv16;             /* bf16 tier, forward / stride-1 data gradient: 1 per shape (default; PM_CONV16), 2 LDS-DMA kernels everywhere, 0 register-staged everywhere */
  int32_t conv16_wide;        /* the wide LDS-DMA forms (conv16w.hip): 1 by the planner's cost model (default; PM_C16W), 0 never, 2 wherever the shape allows, 3 = 2 with the 256 x 256 tile */
  int32_t conv16_persistent;  /* the ring tiles run as the persistent producer / consumer kernel: 1 (default; PM_C16P), 0 one block per tile */
  int32_t wgrad16;            /* bf16 tier, weight gradient on the LDS-DMA persistent ring (wgrad16.hip): 1 (default; PM_WGRAD16), 0 register-staged */
  int32_t bf16_wgrad;         /* prec = 2 weight gradients on pixel-contiguous bf16 copies: 0 (default), 1 */
  int32_t split;              /* fp32 tier on the bf16 matrix pipe (conv_split.hip: fp32 operands, 3-way exact bf16 split, 6 products, fp32 accumulate): 1 (default; PM_SPLIT), 0 */
} pm_routing;
int pm_routing_get(pm_routing* out);            /* out->struct_size must be set by the caller */
int pm_routing_set(const pm_routing* r);

/* ---- single-field wrappers over pm_routing (kernel tests, A/B runs) -------------------------------------------------------------------------------------------------------
 * pm_set_winograd, pm_set_winograd_fused, pm_set_conv16, pm_set_wgrad16, pm_set_split, pm_set_bf16_wgrad and pm_profile_enable flip process-wide ROUTING / MEASUREMENT switches. They exist for
 * same-box A/B runs, kernel tests that must reach a specific kernel, and bench.py's roofline leg; they never change WHAT is computed (every route is parity-tested
 * against the same oracle), only which kernel computes it or whether launches are timed. Contract: call them from ONE thread while no other thread is inside a
 * pm_conv_* entry point (the switches are plain ints read at plan time: a concurrent flip is a benign race between two valid routes for an in-flight PLAN, but the
 * workspace size a caller asked for with the old setting may not fit the new route -> PM_EWORKSPACE, never a wrong result); size queries (pm_conv_workspace,
 * pm_conv_wxf_bytes, ...) must be repeated after a flip (pinthememory_amd/hip/kernels.py keys its size cache on a generation counter for that).
 * A production caller leaves all of them at their defaults and the library is then re-entrant with an immutable kernel table, as the boundary contract says. ------- */

/* Wide stride-1 3x3 convolutions with pad == dilation (Resnet.py:195 conv2 of layer2/3/4, deepv3plus.py:72-81 ASPP branches,
 * :398-404 final1, :455 dsn) run as Winograd convolutions in fp32: F(4x4,3x3) (4x fewer MFMA FLOPs) where the image tiles by
 * 4*dilation, else F(2x2,3x3) (2.25x fewer), else direct. Results stay within fp32 rounding of an fp64 convolution (max error
 * ~3e-6 of the output scale for F(4x4), ~4e-7 for F(2x2); the direct MFMA chain: ~5e-7).
 * mode: 0 = direct implicit GEMM everywhere, 2 = F(2x2) only, 4 = prefer F(4x4) (default). */
int pm_set_winograd(int mode);
/* F(4x4) layers: the 36 point GEMMs and the output transform in ONE kernel, so that the products M = V U^T never reach HBM (one block = 32 tiles x
 * 32 channels x all 36 points). Same results to fp32 round-off; measured slower than GEMM + output-transform pass on every flagship layer
 * (DESIGN.md section 7), hence off by default: 0 off, 1 on. Process-wide like pm_set_winograd. */
int pm_set_winograd_fused(int on);
/* bf16 tier, forward and stride-1 data gradient: which kernel takes a call. 1 (default) = per shape: the LDS-DMA kernel (csrc/conv16.hip: global_load_lds
 * staging, swizzled LDS image) on 64-row tiles and single-K-step reductions, the register-staged kernel of rounds 2-3 on 128 x 128 tiles; 2 = LDS-DMA everywhere;
 * 0 = register-staged everywhere (A/B runs and tests). Round 5: the wide form of the LDS-DMA kernel (csrc/conv16w.hip: 256 x 128 / 128 x 256 tile, eight waves, one block
 * per CU, three-stage LDS ring with counted waits) is part of "per shape" (the planner's cost model decides); 2 = LDS-DMA everywhere on the NARROW tiles only,
 * 3 = LDS-DMA everywhere with the wide tile wherever the shape allows it (kernel tests: the PERSISTENT ring -- one block per CU walks the tiles, four producer waves
 * fetch ahead across tile boundaries, eight waves multiply; 7 = the same with the 256 x 256 two-stage form, 8 = the ring with one block per tile), 4 = per shape
 * without the wide kernel (A/B). PM_C16P=0 in the environment keeps the ring out of "per shape" and runs it one block per tile. The HBM-bound 1x1 convolutions
 * with K = 64 / 128 / 256 had a streaming kernel of their own in round 5 (measured level with the tile kernel, never the default): it left the library in round 6
 * (tools/experiments/pw16/); modes 5 and 6 are accepted and mean 1.
 * Process-wide like pm_set_winograd. */
int pm_set_conv16(int on);
/* prec = 2 weight gradients on pixel-contiguous bf16 copies of x (one per tap) and dy instead of the staged-fp32 form: 0 off (default: the
 * copies cost more HBM time than the GEMM saves), 1 on. Process-wide like pm_set_winograd. */
int pm_set_bf16_wgrad(int on);
/* bf16 tier, weight gradient with both operands bf16 (nn.Conv2d backward of Resnet.py / deepv3plus.py on bf16 rows): 1 (default) = the LDS-DMA persistent ring of
 * csrc/wgrad16.hip wherever Cin is a multiple of 128 and Cout >= 128 (one block per CU walks (256 x 128 tile, pixel range) units, producer waves fetch ahead, fragments
 * through the transpose read), 0 = the register-staged implicit-GEMM kernel everywhere (A/B runs, kernel tests). Same split-K slabs, same fixed-order reduce.
 * Process-wide like pm_set_winograd; PM_WGRAD16=0 in the environment sets the default. */
int pm_set_wgrad16(int on);
/* fp32 tier (round 6): which fp32 convolution / Winograd-GEMM launches run on the bf16 matrix pipe -- fp32 operands, every element split exactly into three bf16
 * pieces inside the kernel (hi + mid + lo == x), six cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 results (csrc/conv_split.hip; replaces the
 * nn.Conv2d of Resnet.py:145-150 / deepv3plus.py:72-81,397-414 like the fp32-MFMA kernel it stands in for). 1 (default; PM_SPLIT in the environment) = every eligible
 * launch (output tiles >= 64 columns wide; forward / data-gradient reductions of >= 129, batched Winograd products of any length), 0 = v_mfma_f32_32x32x2_f32 everywhere
 * (A/B runs, accuracy tests). Same results to fp32 round-off (tests/test_hip_kernels.py::test_split_path_accuracy_vs_fp64). Process-wide like pm_set_winograd. */
int pm_set_split(int on);

/* In-library HIP-event timing of the implicit-GEMM kernel (bench.py's roofline leg). While enabled every conv launch is
 * bracketed by two events on its stream; pm_profile_read sums duration and algorithmic FLOPs (2*M*N*K) of one
 * instantiation conv_igemm_kernel<mode, bm, bn, .., km> (mode 0 fwd / 1 dgrad / 2 wgrad, block tile bm x bn, K-state variant
 * km 0 fast / 1 mid / 2 small, nst LDS stages 1 / 2; negative = any) and optionally clears the records. */
int pm_profile_enable(int on);
int pm_profile_dump(const char* csv_path);   /* one line per recorded launch: mode,bm,bn,km,nst,prec,M,N,K,batch,ksplit,ms,gflop */
int pm_profile_read(int mode, int bm, int bn, int km, int nst, double* total_ms, double* total_flops, int64_t* launches, int clear);
/* the same, additionally keyed by the operand form of the instantiation (prec 0 / 1 / 2 of pm_conv_params; negative = any) */
int pm_profile_read_prec(int mode, int bm, int bn, int km, int nst, int prec, double* total_ms, double* total_flops, int64_t* launches, int clear);
/* Sum of the ALGORITHMIC HBM bytes of the recorded launches of one implicit-GEMM instantiation (same keys, negative = any): both operands once + the output (every slab of a
 * split-K launch) -- the figure bench.py's roofline.traffic (PMC counters) is held against. Call before the pm_profile_read* that clears the records. */
int pm_profile_read_bytes(int mode, int bm, int bn, int km, int nst, int prec, double* total_bytes);

/* ---- K4 BatchNorm2d (mynn.py:8-14 -> nn.BatchNorm2d / SyncBatchNorm, eps 1e-5, momentum 0.1) -------------------
 * stats: per-channel shifted sums -> (count, mean, M2) so that ranks can be merged exactly (SyncBN, train.py:95).
 * moments layout: float[3*C] = mean[C] | m2[C] | count (replicated [C]). */
size_t pm_bn_workspace(const pm_tensor* x);
int pm_bn_stats(const pm_tensor* x, float* moments, void* ws, size_t ws_bytes, void* stream);
/* pm_bn_stats + pm_bn_finalize in one call for local (non-synchronised) statistics: same values, one launch less per BN layer.
 * PM_EINVAL for a single value per channel (torch: "Expected more than 1 value per channel when training"). */
int pm_bn_stats_finalize(const pm_tensor* x, float eps, float* mean, float* invstd, float* running_mean /*nullable*/, float* running_var /*nullable*/,
                         float momentum, void* ws, size_t ws_bytes, void* stream);
/* SyncBatchNorm (train.py:95): exact merge of the per-rank moments gathered over the process group, parts = float[world][3*C] */
int pm_bn_merge(const float* parts, int world, int c, float* moments, void* stream);
/* pm_bn_merge + pm_bn_finalize in one launch (the same values): SyncBatchNorm forward = stats, all-gather, this, apply */
int pm_bn_merge_finalize(const float* parts, int world, int c, float eps, float* mean, float* invstd, float* running_mean /*nullable*/,
                         float* running_var /*nullable*/, float momentum, void* stream);
/* mean/var(biased) -> invstd; optionally updates running stats (unbiased var), momentum as torch. */
int pm_bn_finalize(const float* moments, int c, float eps, float* mean, float* invstd,
                   float* running_mean, float* running_var, float momentum, void* stream);
/* y = relu?( (x-mean)*invstd*gamma + beta + residual? ) */
int pm_bn_apply(const pm_tensor* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                const pm_tensor* residual /*nullable*/, int relu, const pm_tensor* y, void* stream);
/* the same, and (mask != NULL) one byte per float4 channel group of y: bit e set <=> element e of the group is positive BEFORE the ReLU clamp, i.e. the
 * ReLU passes its gradient. Dense [pixels][c / 4] bytes whatever the pitches. pm_bn_bwd_reduce_mask then masks the incoming gradient from these bytes
 * instead of re-reading the forward output (BN + residual + ReLU, Resnet.py:207-216): 1 / 16 of the bytes. */
int pm_bn_apply_mask(const pm_tensor* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const pm_tensor* residual /*nullable*/, int relu, const pm_tensor* y, uint8_t* mask /*nullable*/, void* stream);
int pm_bn_bwd_reduce_mask(const pm_tensor* dy, const uint8_t* mask, const pm_tensor* x, const float* mean, const float* invstd,
                          const pm_tensor* gmask /*nullable*/, float* sums, void* ws, size_t ws_bytes, void* stream);
/* backward: dyz = dy * mask, sums[2*C] = sum(dyz) | sum(dyz * xhat).  relu: 0 no activation (mask = 1); 1 mask = y > 0 read from the
 * forward output (BN + residual + ReLU, Resnet.py:207-216); 2 mask rebuilt from x with gamma / beta (BN + ReLU without a residual:
 * one tensor less to read).  gmask (nullable, relu != 0): dyz is also stored there -- it is the gradient of the residual branch, and
 * pm_bn_bwd_apply can then take it as `dy` with relu = 0 instead of re-reading dy and y. */
int pm_bn_bwd_reduce(const pm_tensor* dy, const pm_tensor* y /*relu == 1 only*/, const pm_tensor* x, const float* mean, const float* invstd,
                     const float* gamma /*relu == 2 only*/, const float* beta /*relu == 2 only*/, int relu, const pm_tensor* gmask /*nullable*/,
                     float* sums, void* ws, size_t ws_bytes, void* stream);
/* dx = gamma*invstd*(dyz - sum_dy/count - xhat*sum_dy_xhat/count); dres = dyz (nullable); count = global element count.
 * count <= 0: the count is read from the device at sums[2*C] (SyncBatchNorm all-reduces [sums | local count] in one exchange, so ranks
 * with uneven batches normalise by the true global count, as torch.nn.SyncBatchNorm does). */
int pm_bn_bwd_apply(const pm_tensor* dy, const pm_tensor* y /*relu == 1 only*/, const pm_tensor* x, const float* mean, const float* invstd,
                    const float* gamma, const float* beta /*relu == 2 only*/, const float* sums, float count, int relu, const pm_tensor* dx,
                    const pm_tensor* dres /*nullable*/, void* stream);
/* eval-mode fold: scale = gamma/sqrt(running_var+eps); shift = beta - running_mean*scale + (conv_bias ? conv_bias*scale : 0) */
/* Chan merge (double, fixed order) of the (mean, M2) slab partials a convolution epilogue emitted for its own output (pm_conv_epilogue.bn_partials,
 * `pixels` rows in slabs of 32). moments == NULL: finalise like pm_bn_stats_finalize (mean, invstd, running moments); moments != NULL: write
 * mean[C] | M2[C] | count[C] for the SyncBatchNorm exchange instead (mean / invstd / running_* are ignored). */
int pm_bn_partials_finalize(const float* partials, int64_t pixels, int c, float eps, float* mean, float* invstd, float* running_mean /*nullable*/,
                            float* running_var /*nullable*/, float momentum, float* moments /*nullable*/, void* stream);
int pm_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, const float* conv_bias,
               int c, float eps, float* scale, float* shift, void* stream);
/* the same fold for n BatchNorm layers in one launch (eval-mode forward of a whole network): table = n x {gamma, beta, running_mean,
 * running_var} device pointers (uint64), cs[n] channel counts, offs[n] offsets into arena; scale_i = arena + offs[i], shift_i = arena + total + offs[i] */
int pm_bn_fold_multi(const void* table, const int* cs, const int* offs, int n, int max_c, int total, float eps, float* arena, void* stream);
/* eval-mode / frozen-stat backward helper and plain elementwise ops */
int pm_relu_bwd(const pm_tensor* dy, const pm_tensor* y, const pm_tensor* dx, void* stream);
int pm_add(const pm_tensor* a, const pm_tensor* b, const pm_tensor* y, void* stream);
/* y = xs[0] + ... + xs[n-1], 2 <= n <= 8, one pass (gradient accumulation of a multi-consumer tensor, e.g. the ASPP input,
 * deepv3plus.py:72-95: autograd would chain n-1 two-operand adds) */
int pm_add_n(const pm_tensor* const* xs, int n, const pm_tensor* y, void* stream);
int pm_copy(const pm_tensor* src, const pm_tensor* dst, void* stream);
int pm_scale_shift_act(const pm_tensor* x, const float* scale, const float* shift, const pm_tensor* residual, int relu,
                       const pm_tensor* y, void* stream);

/* dtype conversion between two views of the same shape (PM_F32 <-> PM_BF16, round to nearest even): the edges of the bf16 tier -- the memory module
 * (memory.py:167-239) and the losses stay fp32. Only the c channels of a pixel are written (pad lanes of a wider pitch are left alone). */
int pm_cast(const pm_tensor* x, const pm_tensor* y, void* stream);

/* ---- K3 pooling (Resnet.py:432 MaxPool2d(3,2,1); deepv3plus.py:85 AdaptiveAvgPool2d(1)) ------------------------- */
int pm_maxpool3x3s2_fwd(const pm_tensor* x, const pm_tensor* y, uint8_t* argmax, void* stream);
int pm_maxpool3x3s2_bwd(const pm_tensor* dy, const uint8_t* argmax, const pm_tensor* dx, void* stream);
int pm_global_avgpool_fwd(const pm_tensor* x, const pm_tensor* y /*n,1,1,c*/, void* stream);
int pm_global_avgpool_bwd(const pm_tensor* dy, const pm_tensor* dx, int accumulate, void* stream);

/* ---- K5 bilinear resize, align_corners=True (mynn.py:57-62), fp32 index math as ATen ---------------------------- */
int pm_resize_bilinear_fwd(const pm_tensor* x, const pm_tensor* y, void* stream);
int pm_resize_bilinear_bwd(const pm_tensor* dy, const pm_tensor* dx, int accumulate, void* stream);
/* the same gradient as a column pass + a row pass through a [n, H, w, c] workspace: dy is read once (up-sampling ratio >= 2 both ways,
 * c % 4 == 0; pm_resize_bilinear_bwd_workspace returns 0 for shapes that must take the gather above) */
size_t pm_resize_bilinear_bwd_workspace(const pm_tensor* dy, const pm_tensor* dx);
int pm_resize_bilinear_bwd_separable(const pm_tensor* dy, const pm_tensor* dx, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ---- pooled multi-scale / flip evaluation (eval.py:133-145,277-337) -------------------------------------------------
 * half-pixel bilinear (F.interpolate(mode='bilinear') default, align_corners=False) of the logits to the original size;
 * MeanFusion: buffer += (softmax(logits) - buffer) / counter in float64 (buffer NHWC double); argmax over channels. */
int pm_resize_bilinear_hp_fwd(const pm_tensor* x, const pm_tensor* y, int flip_w, void* stream);
int pm_softmax_mean_update(const pm_tensor* logits, double* buffer, int counter, void* stream);
/* Sliding-window stitching (eval.py:210-274,340-405): logits [ntiles, th, tw, C] of the tiles (x1,y1,x2,y2)[ntiles] (HOST array) -> float64 sum over
 * the covering tiles (tile order) / tile count, written to acc[C][H][W] at the un-flipped column (flip_w); accumulate != 0 adds to acc. */
int pm_sliding_stitch(const pm_tensor* logits, const int32_t* tiles_xyxy, int ntiles, int H, int W, int flip_w, double* acc, int accumulate, void* stream);
int pm_argmax_f64(const double* buffer, int n, int h, int w, int c, int64_t* out_cls, double* out_prob /*nullable*/, void* stream);

/* ---- input edge: ToTensor + Normalize(ImageNet) (datasets/gtav.py:284-289) and MaskToTensor
 * (transforms/transforms.py:95-97) on the GPU: uint8 HWC images -> normalised NHWC4 fp32 (zero 4th channel, the stem's layout),
 * uint8 label maps -> int64. Cuts the host->device traffic of a batch 4x (images) / 8x (labels). */
int pm_image_u8_to_nhwc4(const uint8_t* img_nhw3, int64_t pixels, const float* mean3, const float* std3, float* out_nhwc4, void* stream);
int pm_labels_u8_to_i64(const uint8_t* lab, int64_t n, int64_t* out, void* stream);

/* ---- layout edges ------------------------------------------------------------------------------------------------ */
int pm_nchw_to_nhwc(const float* x_nchw, int c_src, const pm_tensor* y, void* stream);   /* zero-fills y.c > c_src */
int pm_nhwc_to_nchw(const pm_tensor* x, float* y_nchw, void* stream);
int pm_label_nearest(const int64_t* lab, int n, int H, int W, int64_t* out, int h, int w, void* stream); /* deepv3plus.py:592-594 */

/* ---- K6 fused bilinear-upsample(align_corners) + CrossEntropy(ignore_index=255, mean) ---------------------------
 * Replaces Upsample + criterion (deepv3plus.py:575-578) and the read loss (memory.py:173-176) without materialising
 * the [B,19,H,W] logits. logits: NHWC [n,h,w,C<=32]; labels int64 [n,H,W]. loss_out[0]=mean loss, [1]=valid count. */
size_t pm_upsample_ce_workspace(int n, int H, int W);
int pm_upsample_ce_fwd(const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, float* loss_out,
                       void* ws, size_t ws_bytes, void* stream);
/* dlogits = gscale * d(mean CE)/dlogits ; gscale is a device scalar pointer (upstream grad), may be NULL (=1).
 * Two separable gather passes through a [n,H,w,C] fp32 workspace (no atomics, deterministic). */
size_t pm_upsample_ce_bwd_workspace(const pm_tensor* logits, int H, int W);
int pm_upsample_ce_bwd(const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, const float* loss_out,
                       const float* gscale, const pm_tensor* dlogits, void* ws, size_t ws_bytes, void* stream);

/* Training forward (the logits carry a graph): the same loss, and in the same sweep over the labels the column-reduced gradient field
 *   field[n][H][w][C] = sum over hi-res columns X of (softmax - onehot)(n, Y, X)[c] * (bilinear weight of X on low-res column x)
 * (float, pm_upsample_ce_field_bytes) -- everything the backward needs from labels and up-sampled logits; the upstream scale is a scalar applied
 * last. pm_upsample_ce_bwd_field is then the row pass alone: dlogits = gscale / valid * inv_temp * (field reduced over the supporting hi-res rows).
 * Labels and logits are read once per step instead of twice (replaces autograd through deepv3plus.py:575-578 / memory.py:173-176). */
size_t pm_upsample_ce_field_bytes(const pm_tensor* logits, int H, int W);
int pm_upsample_ce_fwd_field(const pm_tensor* logits, float inv_temp, const int64_t* labels, int H, int W, float* loss_out, float* field,
                             void* ws /* pm_upsample_ce_workspace */, size_t ws_bytes, void* stream);
int pm_upsample_ce_bwd_field(const pm_tensor* logits /* shape only */, float inv_temp, int H, int W, const float* loss_out, const float* gscale,
                             const float* field, const pm_tensor* dlogits, void* stream);

/* ---- K7 memory read (memory.py:317-336 + get_score :167-189) ---------------------------------------------------
 * x: [N rows of d=256] (NHWC feature map); mem [m<=32][d]; writes qr = [qhat | P_m.M] (2d channels, input of
 * memory.output), score S [N][m] (raw cosine scores), P_m [N][m] (softmax over slots, or gumbel if noise given). */
int pm_mem_read_fwd(const pm_tensor* x, const float* mem, int m, const float* gumbel_noise /*nullable [N][m]*/,
                    const pm_tensor* qr, float* score, float* p_mem, void* stream);
/* softmax over all N queries per slot (memory.py:186); two-pass column reduce, deterministic */
size_t pm_mem_colsoftmax_workspace(int64_t rows, int m);
int pm_mem_colsoftmax(const float* score, const float* noise /*nullable*/, int64_t rows, int m, float* p_query,
                      void* ws, size_t ws_bytes, void* stream);
/* read + softmax over all queries (memory.py:317-336 with get_score :183-189 complete) in two launches: the read kernel also leaves
 * per-tile column (max, sum exp) partials of score + noise_q, the second launch merges them in fixed order and writes p_query [N][m].
 * noise / noise_q: the two independent gumbel draws of F.gumbel_softmax(dim=1) / (dim=0); both nullable. */
size_t pm_mem_read_fwd_pq_workspace(int64_t rows, int m);
int pm_mem_read_fwd_pq(const pm_tensor* x, const float* mem, int m, const float* gumbel_noise, const float* gumbel_noise_q,
                       const pm_tensor* qr, float* score, float* p_mem, float* p_query, void* ws, size_t ws_bytes, void* stream);
/* backward into x (and into mem when dmem != NULL): dqr [N][2d], dscore_extra [N][m] (from the read loss; nullable) */
size_t pm_mem_read_bwd_workspace(int64_t rows, int m, int d);
int pm_mem_read_bwd(const pm_tensor* x, const float* mem, int m, const float* p_mem, const pm_tensor* dqr,
                    const float* dscore_extra, const pm_tensor* dx, float* dmem /*nullable [m][d]*/,
                    void* ws, size_t ws_bytes, void* stream);

/* ---- K8 memory write (memory.py:206-239) -----------------------------------------------------------------------
 * accum: zhat = z/max(|z|,1e-12); 4-tap bilinear(align_corners) soft labels of the H x W mask at each h x w pixel
 * (== one_hot(20) -> F.interpolate, memory.py:220-223, without the one-hot); nomden[(m+1)*(d+1)] =
 * nominator[m+1][d] | denominator[m+1], summed over batch and pixels. Deterministic two-stage reduce. */
size_t pm_mem_write_accum_workspace(const pm_tensor* z, int m);
int pm_mem_write_accum(const pm_tensor* z, const int64_t* labels, int H, int W, int m, int normalize,
                       float* nomden, void* ws, size_t ws_bytes, void* stream);
/* dz from dnom [m+1][d] (label weights are constants) through the row normalisation */
int pm_mem_write_accum_bwd(const pm_tensor* z, const int64_t* labels, int H, int W, int m, int normalize,
                           const float* dnom, const pm_tensor* dz, void* stream);
/* update: U = den!=0 ? mu*M + (1-mu)*nom/den : M ; M' = U/max(|U|,1e-12) (memory.py:233-239), device-side predicate */
int pm_mem_write_update(const float* mem, const float* nomden, int m, int d, float momentum, float* mem_out,
                        float* u_out /*nullable, saved for bwd*/, void* stream);
int pm_mem_write_update_bwd(const float* u, const float* nomden, int m, int d, float momentum, const float* dmem_out,
                            float* dnom /*[m+1][d]*/, float* dmem_in /*nullable [m][d]*/, void* stream);

/* ---- optimizer (optimizer.py:21-25: SGD momentum 0.9, wd 5e-4, no nesterov) on a flat parameter arena -------------- */
int pm_sgd_momentum(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                    float weight_decay, int first_step, void* stream);
/* The same update for EVERY parameter tensor (multi-tensor apply): `entries` is a HOST array of n records; the library passes them to
 * the device by value in the kernel arguments (32 records per launch), so nothing is copied or has to outlive the call. A fresh momentum buffer
 * is all zeros (round(m * 0) + d == d: torch's first step). Roundings are torch.optim.SGD's: d = fma(wd, p, g); buf = round(m * buf) + d;
 * p = fma(-lr, buf, p). Replaces torch.optim.SGD.step of optimizer.py:21-25 on the training step. */
typedef struct pm_sgd_entry {
  float* param;
  const float* grad;
  float* momentum_buffer;
  int64_t numel;
} pm_sgd_entry;
int pm_sgd_momentum_multi(const pm_sgd_entry* entries, int n, float lr, float momentum, float weight_decay, void* stream);
/* the same with the learning rate read from device memory when lr_dev != NULL (one float; `lr` is then ignored): a training step captured in a hipGraph follows
 * the LR schedule by a 4-byte write before each replay instead of a re-capture (harness.GraphedAggStep). */
int pm_sgd_momentum_multi_dev(const pm_sgd_entry* entries, int n, float lr, const float* lr_dev, float momentum, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif
