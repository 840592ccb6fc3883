/*
 * rubiks_hip.h -- C ABI of librubiks_hip.so, the MI355X (gfx950) implementation of
 * the RubiksShift hot path.  This is the drop-in boundary: these entry points are
 * what the reference's extension module `rubiksnet_cuda`
 * (cuda_src/rubiks.cpp:384-396, built by setup.py:41-52) binds, restated as plain C
 * (raw device pointers + sizes, no torch types).  INTEGRATION.md shows the
 * reference-side ctypes binding (`rubiksnet_cuda.py`) a maintainer would add.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer on the device that is current when the call
 *     is made; `stream` is a hipStream_t of that device (NULL = the null stream).
 *     The reference launches on the legacy default stream of whatever device is
 *     current and reads device attributes from device 0 (rubiks3d_kernels.cu:975-998);
 *     here the caller names the stream and the library caches nothing per device.
 *   - tensors are dense/contiguous in the reference's layouts:
 *       3D: x [N,T,C,H,W], y / gy [N,To,C,Ho,Wo], shift / gshift [3,C] rows = (T,H,W)
 *           (rubiks.cpp:197-207, :243-244)
 *       2D: x [N,C,H,W],  y / gy [N,C,Ho,Wo],     shift / gshift [2,C] rows = (H,W)
 *           (rubiks.cpp:61-63)
 *     out = (in + 2*pad - 1) / stride + 1   (rubiks.cpp:166 -- not the conv formula);
 *     rk_out_len() exposes it.
 *   - the caller owns every buffer including outputs and the workspace (the reference
 *     allocates scratch inside, rubiks.cpp:127-132,295-299); nothing is allocated,
 *     freed or synchronised inside the library, so calls are graph-capturable.
 *   - 3D kernels write EVERY element of y / gx / gshift (no pre-zeroing needed).
 *     2D with quantize != 0 reproduces the reference quirk of leaving out-of-range
 *     elements of y / gx untouched (rubiks2d_kernels.cu:116-121, :294-309): pre-zero
 *     those two buffers in that case, as rubiksnet/utils.py:26 does.
 *   - return value: RK_OK (0) or a negative RK_ERR_* code; never exit()s
 *     (the reference's gpuAssert does, rubiks3d_kernels.cu:963-971).  The reference's
 *     Python asserts `ret == 0` (rubiks3d/primitive.py:79,139).
 *   - re-entrant and thread-safe: any thread may call any entry point on any device (autograd worker threads,
 *     nn.DataParallel's one-thread-per-device backward).  The only process-wide mutable state is atomic and
 *     order-independent: the launch-tag counter and poll budget of the in-launch finalizers (rk_dma.hpp), and, PER
 *     DEVICE, the cached CU count and the "dynamic-LDS ceiling already raised" bit of each kernel instantiation
 *     (rk_common.hpp: raise_dynamic_lds -- hipFuncSetAttribute is per device).  Calls act on the CURRENT device of the
 *     calling thread (hipGetDevice); the Python layer sets it from the tensors' device.
 *   - ONE environment switch, read once per process: RK_SHIFT_KERNELS = auto | column | generic selects which
 *     kernel families the shift operators may use (rk_common.hpp; every family is bit-identical for y and d(x),
 *     tests/test_fallback_paths_gpu.py).  RK_FORCE_GENERIC=1 is the older spelling of `generic`.
 */
#ifndef RUBIKS_HIP_H_
#define RUBIKS_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rk_stream_t; /* hipStream_t */

enum {
    RK_OK = 0,
    RK_ERR_NULL_POINTER = -1,  /* a required pointer is NULL */
    RK_ERR_BAD_DIMS = -2,      /* a dimension is <= 0 or element count overflows int32 */
    RK_ERR_BAD_STRIDE = -3,    /* stride <= 0 or padding < 0 (reference only prints, rubiks.cpp:162-164) */
    RK_ERR_WORKSPACE = -4,     /* workspace NULL or smaller than *_workspace_bytes() */
    RK_ERR_LAUNCH = -5,        /* hipGetLastError() after a launch was not hipSuccess */
    RK_ERR_NO_DEVICE = -6,     /* no usable HIP device */
    RK_ERR_UNSUPPORTED = -7    /* the fused entry point has no kernel for this configuration: nothing was launched, use
                                  the unfused entry points (only the *_bn_* entry points return it) */
};

int rk_version(void);                 /* 1000*major + minor */
const char* rk_error_string(int code);
int rk_out_len(int in, int stride, int pad); /* rubiks.cpp:14-30, :161-178 */
int rk_device_count(void);            /* hipGetDeviceCount, 0 when none */
/* test hook (no reference counterpart): the tag the next backward launch with an in-launch row-sum will stamp its
 * workspace granules with; lets the parity tests pre-fill a workspace with adversarial near-miss patterns */
unsigned rk_debug_peek_launch_tag(void);
/* test hooks for the in-launch finalizers' give-up path (tests/test_finalizer_gpu.py): the poll budget after which a
 * finalizer wave stops waiting for partials and writes NaN (default ~2 s; spins <= 0 restores it; returns the previous
 * value), and a launch of ONLY the 3-D finalizer waves over a workspace nothing publishes to. */
int rk_debug_set_finalize_spins(int spins);
int rk3d_debug_finalize_only_f32(void* ws, size_t ws_bytes, int C, int partials, float* gshift, int normalize_grad,
                                 float t_factor, rk_stream_t stream);

/* ------------------------------------------------------------------------- 3D
 * Replaces rubiks_shift_3d_forward<T>  (cuda_src/rubiks.cpp:181-253) + functor
 * RubiksShift3DForward (cuda_src/rubiks3d.h:13-28) + K1 (rubiks3d_kernels.cu:15-205).
 * pybind names: rubiks_shift_3d_forward_float / _double (rubiks.cpp:392-393). */
int rk3d_forward_f32(const float* x, const float* shift, float* y,
                     int N, int T, int C, int H, int W,
                     int stride_T, int stride_H, int stride_W,
                     int pad_T, int pad_H, int pad_W,
                     int quantize, rk_stream_t stream);
int rk3d_forward_f64(const double* x, const double* shift, double* y,
                     int N, int T, int C, int H, int W,
                     int stride_T, int stride_H, int stride_W,
                     int pad_T, int pad_H, int pad_W,
                     int quantize, rk_stream_t stream);

/* Bytes of scratch rk3d_backward_* needs (elem_size = 4 or 8).  Replaces the
 * zeros[3C,Ho,Wo] + ones[Ho,Wo] allocations of rubiks.cpp:294-299.  The scratch needs NO initialisation and may be
 * reused by the next call on the same stream: the streaming fp32 kernels keep their partials there as 8-byte
 * {value, launch tag} granules and run the row-sum + K5 inside the backward launch. */
size_t rk3d_backward_workspace_bytes(int N, int T, int C, int H, int W,
                                     int stride_T, int stride_H, int stride_W,
                                     int pad_T, int pad_H, int pad_W, int elem_size);

/* Replaces rubiks_shift_3d_backward<T> (cuda_src/rubiks.cpp:256-379): K2 partials +
 * addmv row-sum (:344-345) + K5 normalise (:352-358) + K3/K4 input grad (:363-376),
 * functors cuda_src/rubiks3d.h:31-81.  pybind names rubiks_shift_3d_backward_float /
 * _double (rubiks.cpp:394-395).  Both gradients are always produced, as in the
 * reference; gx or gshift may be NULL to skip that half (extension).
 * `x` is read only for gshift (the reference passes it to K3/K4 but never reads it). */
int rk3d_backward_f32(const float* x, const float* shift, const float* gy,
                      float* gx, float* gshift,
                      int N, int T, int C, int H, int W,
                      int stride_T, int stride_H, int stride_W,
                      int pad_T, int pad_H, int pad_W,
                      int normalize_grad, float normalize_t_factor, int quantize,
                      void* workspace, size_t workspace_bytes, rk_stream_t stream);
int rk3d_backward_f64(const double* x, const double* shift, const double* gy,
                      double* gx, double* gshift,
                      int N, int T, int C, int H, int W,
                      int stride_T, int stride_H, int stride_W,
                      int pad_T, int pad_H, int pad_W,
                      int normalize_grad, double normalize_t_factor, int quantize,
                      void* workspace, size_t workspace_bytes, rk_stream_t stream);

/* Two-phase form of rk3d_backward_f32, the phases of the reference's own host glue (rubiks.cpp:324-376):
 * _partials = K2 + K3/K4: writes gx (or skips it when NULL) and the per-channel partial sums workspace[C][3][P],
 * P returned through *partials; _finalize = addmv row-sum (:344-345) + K5 normalise (:352-358) into gshift[3][C].
 * rk3d_backward_f32 computes exactly what _partials followed by _finalize compute, bit for bit (same partials,
 * same summation order); on the streaming shapes it does so in ONE launch (bench.py times both forms).          */
int rk3d_backward_partials_f32(const float* x, const float* shift, const float* gy, float* gx,
                               int N, int T, int C, int H, int W,
                               int stride_T, int stride_H, int stride_W, int pad_T, int pad_H, int pad_W,
                               int quantize, void* workspace, size_t workspace_bytes, int* partials,
                               rk_stream_t stream);
int rk3d_backward_finalize_f32(const void* workspace, int C, int partials, float* gshift,
                               int normalize_grad, float normalize_t_factor, rk_stream_t stream);

/* ------------------------------------------------------------------------- 2D
 * Replaces rubiks2d_forward (cuda_src/rubiks.cpp:44-67) + rubiks2d_forward_cuda
 * (rubiks2d_kernels.cu:408-432) + K6 (:94-145).  The reference dispatches
 * float/double/half (AT_DISPATCH_FLOATING_TYPES_AND_HALF); bf16 is an addition.
 * f16/bf16 compute in fp32 and round once on store. */
#define RK_DECL_2D(SFX, TYPE)                                                              \
    int rk2d_forward_##SFX(const TYPE* x, const TYPE* shift, TYPE* y,                      \
                           int N, int C, int H, int W,                                     \
                           int stride_H, int stride_W, int pad_H, int pad_W,               \
                           int quantize, rk_stream_t stream);                              \
    int rk2d_backward_##SFX(const TYPE* gy, const TYPE* x, const TYPE* shift,              \
                            TYPE* gx, TYPE* gshift,                                        \
                            int N, int C, int H, int W,                                    \
                            int stride_H, int stride_W, int pad_H, int pad_W,              \
                            int normalize_grad, int enable_shift_grad, int quantize,       \
                            void* workspace, size_t workspace_bytes, rk_stream_t stream);

/* rk2d_backward_* replaces rubiks2d_backward (cuda_src/rubiks.cpp:94-155): K7 partials
 * (rubiks2d_kernels.cu:147-266) + addmv row-sum (rubiks.cpp:140-143) + K9 normalise
 * (:381-397) when enable_shift_grad, then K8 (:269-379).  With enable_shift_grad == 0
 * gshift is left untouched, as in the reference.  f16 / bf16 pointers are uint16_t
 * bit patterns, passed as void*. */
RK_DECL_2D(f32, float)
RK_DECL_2D(f64, double)
RK_DECL_2D(f16, void)
RK_DECL_2D(bf16, void)
#undef RK_DECL_2D

/* 16-bit activations next to an fp32 shift table (what torch.autocast hands the operator: bf16 / f16 activations,
 * fp32 nn.Parameter).  The reference instantiates K6-K9 at ONE scalar type (rubiks2d_kernels.cu:113-114 reads the
 * shift in the tensor's type), so an autocast caller of it must round the parameter to 16 bits first; these entry
 * points keep the shift, everything derived from it (floor / remainder, the 1e-7 integer test of :189, the quantize
 * positions of :117-118) and d(shift) in fp32, exactly as rk2d_*_f32 evaluate them.  Same workspace as rk2d_backward_*. */
#define RK_DECL_2D_SF32(SFX)                                                                \
    int rk2d_forward_##SFX##_sf32(const void* x, const float* shift, void* y,              \
                                  int N, int C, int H, int W,                              \
                                  int stride_H, int stride_W, int pad_H, int pad_W,        \
                                  int quantize, rk_stream_t stream);                       \
    int rk2d_backward_##SFX##_sf32(const void* gy, const void* x, const float* shift,      \
                                   void* gx, float* gshift,                                \
                                   int N, int C, int H, int W,                             \
                                   int stride_H, int stride_W, int pad_H, int pad_W,       \
                                   int normalize_grad, int enable_shift_grad, int quantize, \
                                   void* workspace, size_t workspace_bytes, rk_stream_t stream);
RK_DECL_2D_SF32(f16)
RK_DECL_2D_SF32(bf16)
#undef RK_DECL_2D_SF32

size_t rk2d_backward_workspace_bytes(int N, int C, int H, int W,
                                     int stride_H, int stride_W, int pad_H, int pad_W,
                                     int elem_size);

/* Training fusion for the 2-D operator (round 5; the -aq blocks of rubiksnet/backbone.py:129-131 with models.py:71-79's
 * RubiksShift2D): the shift applied to relu(bn2(z)) = max(a z + b, 0) without the activation being stored.
 *   forward : y = shift2d(round(max(a z + b, 0))); ab [2][C] = the affine map of bn2's training forward
 *             (rk_bn_stats_finish_* / rk_bn_finish_tiles_f32).
 *   backward: dz = d(bn2's output) = d(x) of the shift masked by the ReLU; gshift [2][C] (K9 applied when normalize_grad);
 *             bn2's backward constants k12 [2][C] = (sum dz, sum dz zhat) / count, d(gamma), d(beta): what
 *             rk_bn_bwd_dx_pre_* needs to finish bn2's d(x) in one pass.  abmi [C][4] = (a, b, mean, invstd).
 * Shift table and d(shift) in fp32 (as rk2d_*_sf32).  RK_ERR_UNSUPPORTED (nothing launched) when no fused kernel takes the
 * configuration: quantize on, or fp32 planes the LDS-DMA kernels stream (stride 1, pad 0, W % 4 == 0 other than 14 x 14 --
 * normalise pass + streaming kernel is the faster pair there), or a shape only the column kernels take with RK_COLUMN=0;
 * the caller then normalises with rk_bn_apply_affine_* and calls the plain entry points.  Fused today: 14 x 14 planes
 * (fp32 / bf16), bf16 56 x 56 / 112 x 112 (raw-plane kernels), bf16 28 x 28 (register-staged), and every other stride /
 * padding / plane through the column kernels (the stride-2 layers, 7 x 7). */
/* 1 when a fused kernel takes this configuration at this storage size (4 = fp32, 2 = bf16), else 0 */
int rk2d_bn_fused_shape(int N, int C, int H, int W, int sH, int sW, int pH, int pW, int elem_size);
size_t rk2d_backward_bn_workspace_bytes(int N, int C, int H, int W, int stride_H, int stride_W, int pad_H, int pad_W);
int rk2d_forward_bn_f32(const float* z, const float* ab, const float* shift, float* y, int N, int C, int H, int W,
                        int stride_H, int stride_W, int pad_H, int pad_W, int quantize, rk_stream_t stream);
int rk2d_forward_bn_bf16_sf32(const void* z, const float* ab, const float* shift, void* y, int N, int C, int H, int W,
                              int stride_H, int stride_W, int pad_H, int pad_W, int quantize, rk_stream_t stream);
int rk2d_backward_bn_f32(const float* gy, const float* z, const float* abmi, const float* shift, float* dz, float* gshift,
                         float* k12, float* dgamma, float* dbeta, int N, int C, int H, int W, int stride_H, int stride_W,
                         int pad_H, int pad_W, int normalize_grad, int quantize, void* workspace, size_t workspace_bytes,
                         rk_stream_t stream);
int rk2d_backward_bn_bf16_sf32(const void* gy, const void* z, const float* abmi, const float* shift, void* dz, float* gshift,
                               float* k12, float* dgamma, float* dbeta, int N, int C, int H, int W, int stride_H,
                               int stride_W, int pad_H, int pad_W, int normalize_grad, int quantize, void* workspace,
                               size_t workspace_bytes, rk_stream_t stream);

/* ------------------------------------------------------------ temporal 3-tap
 * The device half of AttentionShift (rubiksnet/attention_shift.py:32-39): the
 * per-channel 3-tap temporal filter the reference runs as transpose -> grouped
 * conv1d(groups=C*H*W) -> transpose -> contiguous.  `taps` is the [C,3] tensor (fp32;
 * fp64 for the _f64 entry points) of already soft-maxed weights (attention_shift.py:29-30, computed on the host side
 * in PyTorch so autograd owns std/softmax); x, y are [NT, C, H, W] with
 * NT = n_batch * n_segment; zero padding in t.
 *   y[n,t] = taps[c,0]*x[n,t-1] + taps[c,1]*x[n,t] + taps[c,2]*x[n,t+1]
 * backward: gx = adjoint; gtaps[C,3] (same type as taps) = sum over n,t,h,w of gy * x[t-1+k]. */
#define RK_DECL_TAP(SFX, TYPE, TAPT)                                                       \
    int rk_tshift3_forward_##SFX(const TYPE* x, const TAPT* taps, TYPE* y,                 \
                                 int NT, int n_segment, int C, int HW, rk_stream_t stream);\
    int rk_tshift3_backward_##SFX(const TYPE* gy, const TYPE* x, const TAPT* taps,         \
                                  TYPE* gx, TAPT* gtaps,                                   \
                                  int NT, int n_segment, int C, int HW,                    \
                                  void* workspace, size_t workspace_bytes,                 \
                                  rk_stream_t stream);
RK_DECL_TAP(f32, float, float)
RK_DECL_TAP(f64, double, double)
RK_DECL_TAP(f16, void, float)
RK_DECL_TAP(bf16, void, float)
#undef RK_DECL_TAP

size_t rk_tshift3_backward_workspace_bytes(int NT, int n_segment, int C, int HW);
/* The -aq block's training-mode bn1 + ReLU folded into its temporal filter (backbone.py:129 `out = relu(bn1(x))` feeding
 * attention_shift.py:29-39): y = tshift3(relu(a[c] x + b[c])), ab = [2][C] from rk_bn_stats_finish_*; backward: gy -> dz =
 * d(relu(a x + b)) masked by the ReLU (x = the block input BEFORE bn1), gtaps, and bred [C][NT / n_segment] = (sum dz,
 * sum dz xhat) per (channel, clip) for rk_bn_bwd_finish_tiles_f32 (tiles = NT / n_segment) + rk_bn_bwd_dx_pre_*. */
int rk_tshift3_bn_forward_f32(const float* x, const float* taps, const float* ab, float* y, int NT, int S, int C, int HW,
                              rk_stream_t stream);
int rk_tshift3_bn_forward_bf16(const void* x, const float* taps, const float* ab, void* y, int NT, int S, int C, int HW,
                               rk_stream_t stream);
int rk_tshift3_bn_backward_f32(const float* gy, const float* x, const float* taps, const float* ab, const float* save_mean,
                               const float* save_invstd, float* dz, float* gtaps, void* bred, int NT, int S, int C, int HW,
                               void* ws, size_t ws_bytes, rk_stream_t stream);
int rk_tshift3_bn_backward_bf16(const void* gy, const void* x, const float* taps, const float* ab, const float* save_mean,
                                const float* save_invstd, void* dz, float* gtaps, void* bred, int NT, int S, int C, int HW,
                                void* ws, size_t ws_bytes, rk_stream_t stream);

/* The same with BatchNorm's backward constants finished inside the launch (the two sums travel as granules next to the tap
 * sums): k12 [2][C] = (sum dz, sum dz xhat) / (NT HW), dgamma / dbeta [C] -- what rk_bn_bwd_finish_tiles_f32 made of bred. */
size_t rk_tshift3_bn_backward_fin_workspace_bytes(int NT, int n_segment, int C, int HW);
int rk_tshift3_bn_backward_fin_f32(const float* gy, const float* x, const float* taps, const float* ab, const float* save_mean,
                                   const float* save_invstd, float* dz, float* gtaps, float* k12, float* dgamma, float* dbeta,
                                   int NT, int S, int C, int HW, void* ws, size_t ws_bytes, rk_stream_t stream);
int rk_tshift3_bn_backward_fin_bf16(const void* gy, const void* x, const float* taps, const float* ab, const float* save_mean,
                                    const float* save_invstd, void* dz, float* gtaps, float* k12, float* dgamma, float* dbeta,
                                    int NT, int S, int C, int HW, void* ws, size_t ws_bytes, rk_stream_t stream);

/* Downsampling -aq blocks (the activation also feeds the stride-2 projecting shortcut, backbone.py:98-104): the shortcut reads
 * rk_bn_relu_gather2_* (relu(a x + b) at the even pixels, ab = [2][C]); its gradient gsmall [NT, C, H/2, W/2] joins
 * d(activation) inside the temporal filter's backward, before the ReLU mask and BatchNorm's sums. */
int rk_bn_relu_gather2_f32(const float* x, const float* ab, float* xs, int F, int C, int H, int W, rk_stream_t stream);
int rk_bn_relu_gather2_bf16(const void* x, const float* ab, void* xs, int F, int C, int H, int W, rk_stream_t stream);
int rk_tshift3_bn_backward_fork_f32(const float* gy, const float* x, const float* taps, const float* ab, const float* save_mean,
                                    const float* save_invstd, const float* gsmall, float* dz, float* gtaps, float* k12,
                                    float* dgamma, float* dbeta, int NT, int S, int C, int H, int W, void* ws, size_t ws_bytes,
                                    rk_stream_t stream);
int rk_tshift3_bn_backward_fork_bf16(const void* gy, const void* x, const float* taps, const float* ab, const float* save_mean,
                                     const float* save_invstd, const void* gsmall, void* dz, float* gtaps, float* k12,
                                     float* dgamma, float* dbeta, int NT, int S, int C, int H, int W, void* ws, size_t ws_bytes,
                                     rk_stream_t stream);

/* The [C,3] half of AttentionShift (attention_shift.py:29-30): taps = softmax((weight / (std(weight, dim=1) + 1e-6)) / T)
 * over the three taps of a channel (std unbiased), and its backward (gweight from gtaps).  fp32; T is the module's
 * one-element device tensor (no host read); one launch each instead of ~15 + ~25 PyTorch kernels per layer. */
int rk_soft_taps_forward_f32(const float* weight, const float* T, float* taps, int C, rk_stream_t stream);
int rk_soft_taps_backward_f32(const float* weight, const float* T, const float* taps, const float* gtaps, float* gweight,
                              int C, rk_stream_t stream);
/* Every AttentionShift layer of a network in one launch each way.  jobs: device array of n records {const float* weight;
 * const float* T; int64 off; int32 C; int32 pad} (32 bytes); layer i's taps / gtaps / gweight are rows off .. off + C of
 * concatenated [sum C][3] fp32 buffers; max_c = the largest C.  Same arithmetic as the one-layer calls (bit-identical). */
int rk_soft_taps_many_forward_f32(const void* jobs, int n, float* taps, int max_c, rk_stream_t stream);
int rk_soft_taps_many_backward_f32(const void* jobs, int n, const float* taps, const float* gtaps, float* gweight, int max_c,
                                   rk_stream_t stream);

/* ---- BatchNorm2d (+ ReLU) of the backbone blocks -- widening row f3 of SURVEY 8(f) ----------------
 * Replaces the reference's nn.BatchNorm2d followed by nn.ReLU(inplace=True)
 * (rubiksnet/backbone.py:50-53 BN2d; :129 relu(bn1(x)), :131 relu(bn2(conv2(.))), :196 relu(bn_last(x))),
 * i.e. torch.nn.functional.batch_norm + relu on an NCHW tensor.
 *   x, y, dy, dx : [F, C, P] contiguous (F = N*T frames, P = H*W), f32 or bf16; parameters / statistics are f32.
 *   training != 0: batch statistics (biased variance for the normalisation); save_mean / save_invstd [C] are
 *                  written for the backward; running_mean / running_var (both or neither) are updated as
 *                  r = (1 - momentum) r + momentum * stat, with the UNBIASED variance, as torch does.
 *                  Needs ws of rk_bn_workspace_bytes(F, C, P) bytes.
 *   training == 0: y = relu?(gamma (x - running_mean) / sqrt(running_var + eps) + beta); save_*, ws unused.
 *   backward     : gradients of the training-mode forward; the ReLU mask is recomputed from x.  dskip (or NULL): a
 *                  gradient of the same shape added into dx -- the branch of the block's identity shortcut
 *                  (backbone.py:130), which autograd would otherwise sum with a separate elementwise pass.
 *   relu != 0 fuses the ReLU (and its backward).                                                         */
size_t rk_bn_workspace_bytes(int F, int C, int P);
#define RK_DECL_BN(SFX, CTYPE)                                                                                   \
    int rk_bn_relu_forward_##SFX(const CTYPE* x, const float* gamma, const float* beta, float* running_mean,     \
                                 float* running_var, float* save_mean, float* save_invstd, CTYPE* y, int F,      \
                                 int C, int P, float eps, float momentum, int relu, int training, void* ws,     \
                                 size_t ws_bytes, rk_stream_t stream);                                           \
    /* training forward that also does nn.BatchNorm2d's `num_batches_tracked += 1` (int64 device scalar, may be   \
     * NULL) inside the launch instead of a kernel of its own */                                                  \
    int rk_bn_relu_forward_counted_##SFX(const CTYPE* x, const float* gamma, const float* beta,                  \
                                         float* running_mean, float* running_var, float* save_mean,              \
                                         float* save_invstd, CTYPE* y, int F, int C, int P, float eps,           \
                                         float momentum, int relu, long long* num_batches_tracked, void* ws,    \
                                         size_t ws_bytes, rk_stream_t stream);                                   \
    int rk_bn_relu_backward_##SFX(const CTYPE* dy, const CTYPE* x, const float* gamma, const float* beta,        \
                                  const float* save_mean, const float* save_invstd, const CTYPE* dskip,          \
                                  CTYPE* dx, float* dgamma, float* dbeta, int F, int C, int P, int relu,         \
                                  void* ws, size_t ws_bytes, rk_stream_t stream);
RK_DECL_BN(f32, float)
RK_DECL_BN(bf16, void)
#undef RK_DECL_BN

/* ---- 1x1 convolution on NCHW activations -- widening row f1 of SURVEY 8(f), unfused half -----------
 * Replaces torch.nn.functional.conv2d with a 1x1 kernel, stride 1, no bias (rubiksnet/backbone.py:44-45
 * Conv1x1; conv2 / conv3 / stride-1 shortcuts of every block, :87-104) and its two gradients.
 *   rk_pw_gemm_*  : Y[f] = A X[f] (+ R[f]), X [F,K,P], Y / R [F,M,P] fp32 or bf16 storage (fp32 arithmetic),
 *                   A always fp32 (under bf16 autocast the weight is used as it is, no cast), P % 4 == 0.
 *                   forward: A = weight [M=Cout][K=Cin], a_is_mk = 1; R = the block's shortcut
 *                   (backbone.py:134 `out += shortcut`) or NULL;
 *                   d(input): A = weight read as [K=Cout][M=Cin], a_is_mk = 0, X = d(output), R = NULL.
 *   rk_pw_wgrad_* : d(weight)[M][K] (fp32) = sum_f dY[f] X[f]^T, dY [F,M,P], X [F,K,P]; ws of
 *                   rk_pw_wgrad_workspace_bytes() bytes holds per-chunk partials (summed in a fixed order).   */
int rk_pw_gemm_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P,
                   int a_is_mk, rk_stream_t stream);
int rk_pw_gemm_bf16(const float* A, const void* X, const void* R, void* Y, int F, int K, int M, int P,
                    int a_is_mk, rk_stream_t stream);
/*   bf16 activations, second generation (rk_pw16.hip): the weight is packed ONCE per version into bf16 MFMA-fragment
 *   order and the GEMM streams X by LDS-DMA, every row of a 128-pixel tile in one workgroup (X crosses HBM once).
 *   rk_pw_packed_bytes(rows, depth): bytes of a packed operand;
 *   rk_pw_pack_bf16: W [Cout][Cin] fp32 -> `fwd` (rows Cout, depth Cin: forward) and / or `bwd` (rows Cin, depth
 *                   Cout: W^T for d(input)); either may be NULL;
 *   rk_pw_gemm_packed_bf16: Y[f] = A X[f] (+ R[f]) with A packed (M rows, depth K); R may be NULL or Y itself.  R [F,M,P]
 *                   is streamed through the same LDS-DMA ring as X (identity-weighted chunks after X's: exact), so both
 *                   F*K*P*2 and F*M*P*2 must stay below 2^31 bytes (RK_ERR_BAD_DIMS otherwise). */
size_t rk_pw_packed_bytes(int rows, int depth);
int rk_pw_pack_bf16(const float* W, int Cout, int Cin, void* fwd, void* bwd, rk_stream_t stream);
/* The same for n weights in one launch (once per train step, pointwise.prepacked): `jobs` = device array of n 40-byte
 * records {const float* W; int64_t fwd_off, bwd_off; int32_t Cout, Cin, nf, nb}: the two images go to base + fwd_off /
 * base + bwd_off (multiples of 16), nf / nb = rk_pw_packed_bytes(..) / 16 of each; max_units = max over jobs of nf + nb. */
int rk_pw_pack_many_bf16(const void* jobs, int n, void* base, int max_units, rk_stream_t stream);
/* bf16 activations on planes without a 16-byte unit in a row (P % 4 != 0, P <= 64: the 7x7 planes of layer4, backbone.py:44-45
 * on [NT, 576, 7, 7]; rk_pw16_odd.hip): a workgroup per frame (a whole frame [K][P] IS contiguous and aligned).
 *   rk_pw_odd16_supported(F, K, M, P): 1 when the GEMM takes [F, K -> M, P] (K % 32 == 0, M % 8 == 0, P <= 64, LDS);
 *   rk_pw_gemm_packed_odd_bf16: Y[f] = A X[f] (+ R[f]), A packed by rk_pw_pack_bf16 (M rows, depth K); tensors 16-byte aligned;
 *   rk_pw_wgrad_odd16_bf16: dW [M][K] fp32 = sum_f dY[f] X[f]^T (K % 8 == M % 8 == 0); ws of .._workspace_bytes(). */
/* The 3x3 / stride-2 / pad-1 stem (backbone.py:154) under bf16 autocast (rk_stem16.hip): fp32 clip X [F, 3, Hin, Win] and fp32
 * weight W [Cout][3][3][3], both rounded to bf16 as autocast rounds them, Y / dY [F, Cout, Hin/2, Win/2] bf16, dW fp32.
 * Win % 32 == 0, Hin % 16 == 0, Cout <= 128 (rk_stem16_supported). */
int rk_stem16_supported(int F, int Cin, int Cout, int Hin, int Win);
/* out [planes][H][W] bf16 = main (NULL: zeros) + small [planes][H/2][W/2] scattered to the even (h, w) -- the gradient of the
 * stride-2 gather in front of a projecting shortcut (backbone.py:98-104) joined to the other consumer's gradient in one pass. */
int rk_scatter2x2_add_bf16(const void* main, const void* small, void* out, long long planes, int H, int W, rk_stream_t stream);
int rk_stem_conv3x3s2_bf16out(const float* W, const float* X, void* Y, int F, int Cin, int Cout, int Hin, int Win, rk_stream_t stream);
size_t rk_stem_wgrad16_workspace_bytes(int F, int Cin, int Cout, int Hin, int Win);
int rk_stem_wgrad3x3s2_bf16(const void* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win, void* workspace,
                            size_t workspace_bytes, rk_stream_t stream);
int rk_pw_odd16_supported(int F, int K, int M, int P);
int rk_pw_gemm_packed_odd_bf16(const void* Apk, const void* X, const void* R, void* Y, int F, int K, int M, int P, rk_stream_t stream);
size_t rk_pw_wgrad_odd16_workspace_bytes(int F, int K, int M, int P);
int rk_pw_wgrad_odd16_bf16(const void* dY, const void* X, float* dW, int F, int K, int M, int P, void* workspace,
                           size_t workspace_bytes, rk_stream_t stream);
int rk_pw_gemm_packed_bf16(const void* Apk, const void* X, const void* R, void* Y, int F, int K, int M, int P,
                           rk_stream_t stream);
/* training (round 5): the same GEMM + the tile statistics of Y for the BatchNorm that consumes it (backbone.py:50-53 after
 * :44-45 under autocast): stats float4 [M][tiles] = (pivot, sum(y - pivot), sum((y - pivot)^2), columns) per 64 columns of
 * the values as stored, tiles = rk_pw16_stat_tiles(F, P); finished by rk_bn_finish_tiles_f32. */
int rk_pw16_stat_tiles(int F, int P);
int rk_pw_gemm_packed_stats_bf16(const void* Apk, const void* X, const void* R, void* Y, int F, int K, int M, int P,
                                 void* stats, int tiles, rk_stream_t stream);
/*   rk_pw_wgrad16_bf16: d(weight)[M][K] (fp32) = sum_f dY[f] X[f]^T for bf16 dY [F,M,P], X [F,K,P] (P % 4 == 0, P >= 8):
 *                   both operands DMA'd fragment-wise, output tiles of up to 160 x 160 per workgroup; ws of
 *                   rk_pw_wgrad16_workspace_bytes() bytes (per-split partial matrices, summed in a fixed order). */
size_t rk_pw_wgrad16_workspace_bytes(int F, int K, int M, int P);
int rk_pw_wgrad16_bf16(const void* dY, const void* X, float* dW, int F, int K, int M, int P, void* ws, size_t ws_bytes,
                       rk_stream_t stream);
/*   rk_pw_gemm_fused_f32: inference form, Y[f] = epi(A pro(X[f])) (+ R[f]) with
 *                   pro: x' = relu?(ka[k] x + kb[k]) per input channel  (relu(bn1(x)) feeding conv2, backbone.py:129-131)
 *                   epi: y  = relu?(ma[m] y + mb[m]) per output channel (relu(bn2(conv2(.))), BatchNorm in eval mode:
 *                   a = gamma / sqrt(running_var + eps), b = beta - running_mean * a).  NULL pairs switch a stage off. */
/* eval-mode BatchNorm2d folded to y = a x + b per channel (a = gamma / sqrt(running_var + eps), b = beta -
 * running_mean a): the (ka, kb) / (ma, mb) arrays of the fused entry points below, in one launch. */
int rk_bn_fold_f32(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                   float* a, float* b, int C, rk_stream_t stream);
/* the same fold for n BatchNorm layers in one launch: jobs = device array of {gamma*, beta*, running_mean*, running_var*,
 * int64 off, int32 C, float eps} (48 bytes); ab = [2][total] fp32, layer i's a / b at [off, off + C) of each half */
int rk_bn_fold_many_f32(const void* jobs, int n, float* ab, long long total, int max_c, rk_stream_t stream);
int rk_pw_gemm_fused_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P,
                         int a_is_mk, const float* ka, const float* kb, int relu_in, const float* ma,
                         const float* mb, int relu_out, rk_stream_t stream);
/*   rk_stem_conv3x3s2_f32: the backbone's first layer (backbone.py:154, Conv3x3(3, width, stride=2), no bias) on the
 *                   same GEMM with the im2col gathered on the fly: W [Cout][Cin][3][3], X [F,Cin,Hin,Win],
 *                   Y [F,Cout,Hin/2,Win/2]; Hin even, Win % 8 == 0, 9 Cin <= 64.                              */
int rk_stem_conv3x3s2_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                          rk_stream_t stream);
/* d(weight) of that convolution: dW [Cout][Cin][3][3] (fp32) from dY [F, Cout, Hin/2, Win/2] and X [F, Cin, Hin, Win];
 * workspace of rk_pw_wgrad_workspace_bytes(F, 9 * Cin, Cout, (Hin/2) * (Win/2)) bytes.  Replaces the d(weight) half of
 * conv2d's backward for the stem (backbone.py:154); the stem needs no d(input). */
int rk_stem_wgrad3x3s2_f32(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win,
                           void* workspace, size_t workspace_bytes, rk_stream_t stream);
/* 1x1 / stride-2 / no-bias convolution (the projecting shortcut of a downsampling block, backbone.py:98-104) on the
 * same GEMM kernels: forward (the streamed operand read at stride 2), d(input) (results scattered to the even
 * positions, zeros elsewhere: every element of dX is written) and d(weight).  W [Cout][Cin] fp32, X / dX
 * [F, Cin, Hin, Win], Y / dY [F, Cout, Hin/2, Win/2]; Hin even, Win % 8 == 0, Cin and Cout even; the d(weight)
 * workspace is rk_pw_wgrad_workspace_bytes(F, Cin, Cout, (Hin/2) * (Win/2)) bytes. */
int rk_pw_s2_forward_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                         rk_stream_t stream);
int rk_pw_s2_forward_fused_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                               const float* ka, const float* kb, int relu_in, rk_stream_t stream);   /* inference: x' =
                               relu?(ka[k] x + kb[k]) on the streamed operand, as rk_pw_gemm_fused_f32's prologue */
int rk_pw_s2_dgrad_f32(const float* W, const float* dY, float* dX, int F, int Cin, int Cout, int Hin, int Win,
                       rk_stream_t stream);
int rk_pw_s2_wgrad_f32(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win,
                       void* workspace, size_t workspace_bytes, rk_stream_t stream);
size_t rk_pw_wgrad_workspace_bytes(int F, int K, int M, int P);
int rk_pw_wgrad_f32(const float* dY, const float* X, float* dW, int F, int K, int M, int P, void* ws,
                    size_t ws_bytes, rk_stream_t stream);
int rk_pw_wgrad_bf16(const void* dY, const void* X, float* dW, int F, int K, int M, int P, void* ws,
                     size_t ws_bytes, rk_stream_t stream);

/* 1x1 convolutions on planes with H * W % 4 != 0 -- the 7x7 planes of layer4 (rubiksnet/backbone.py:164), which the
 * kernels above cannot take (a row of 49 floats is not 16-byte aligned): the streamed operand goes through LDS as whole
 * 16-channel frame chunks (contiguous and aligned), see k_pw_gemm_odd.  37 <= P <= 64, K % 4 == 0, fp32.
 *   rk_pw_gemm_odd_f32     : Y[f] = A X[f] (+ R[f]), forward (a_is_mk = 1) / d(input) (a_is_mk = 0)
 *   rk_pw_wgrad_odd_f32    : d(weight); workspace rk_pw_wgrad_odd_workspace_bytes(F, K, M, P)
 *   rk_pw_s2_*_odd_f32     : the 1x1 / stride-2 projecting shortcut onto such planes (14x14 -> 7x7, backbone.py:98-104):
 *                            forward, d(input) (every element of dX written) and d(weight); Hin, Win even. */
int rk_pw_gemm_odd_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P,
                       int a_is_mk, rk_stream_t stream);
size_t rk_pw_wgrad_odd_workspace_bytes(int F, int K, int M, int P);
int rk_pw_wgrad_odd_f32(const float* dY, const float* X, float* dW, int F, int K, int M, int P, void* ws,
                        size_t ws_bytes, rk_stream_t stream);
int rk_pw_s2_forward_odd_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                             rk_stream_t stream);
int rk_pw_s2_dgrad_odd_f32(const float* W, const float* dY, float* dX, int F, int Cin, int Cout, int Hin, int Win,
                           rk_stream_t stream);
int rk_pw_s2_wgrad_odd_f32(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win,
                           void* ws, size_t ws_bytes, rk_stream_t stream);

/* ---- training-mode fusion of the block's BatchNorms into the 1x1 GEMMs -- rows f1 / f3 of SURVEY 8(f) ----------
 * The reference block (rubiksnet/backbone.py:123-135) is
 *     a1 = relu(bn1(x)); z = conv2(a1); a2 = relu(bn2(z)); s = as3(a2); out = conv3(s) + shortcut
 * with nn.BatchNorm2d in training mode (:50-53).  Unfused, every BatchNorm is a statistics pass + a normalise pass
 * forward and a reduction pass + a d(x) pass backward.  Here
 *   - the STATISTICS pass rides on the epilogue of the GEMM that produces the tensor (rk_pw_gemm_stats_f32; the stem:
 *     rk_stem_conv3x3s2_stats_f32): one float4 (pivot, sum(y - pivot), sum((y - pivot)^2), -) per (channel, 128-column
 *     wave tile), [M][rk_pw_tiles(F, P)], finished by rk_bn_finish_tiles_f32 into save_mean / save_invstd, the affine map
 *     (a, b) of y = a x + b, and nn.BatchNorm2d's running statistics / num_batches_tracked;
 *   - relu(bn1(x)) is never stored: it is the PROLOGUE (ka, kb, relu_in) of conv2's forward, of the strided shortcut's
 *     forward (rk_pw_s2_forward_fused_f32) and of their d(weight) kernels (rk_pw_wgrad_pro_f32, rk_pw_s2_wgrad_pro_f32);
 *   - the backward REDUCTION pass of bn1 rides on conv2's d(input) GEMM (rk_pw_gemm_bnbwd_f32): the result tile is masked
 *     with [a x + b > 0] and its (sum dz, sum dz xhat) go out as float2 per (channel, tile); rk_bn_bwd_finish_tiles_f32
 *     turns them into k1, k2, d(gamma), d(beta); rk_bn_bwd_dx_pre_f32 is the remaining d(x) pass (+ the identity
 *     shortcut's gradient);
 *   - rk_bn_apply_affine_f32: y = relu?(a x + b) with the finished map (bn2 in front of the shift);
 *   - rk_bn_tile_stats_f32: the same tile statistics for a tensor no GEMM epilogue produced them for.
 * All fp32, P % 4 == 0; results match torch.nn.functional.batch_norm + relu (+ conv2d) and their autograd gradients. */
/*   - relu(bn2(z)) is never stored either (f1 "and/or the preceding ReLU(BN2(.))"): the shift kernels read z and normalise
 *     the landed planes in LDS (rk3d_forward_bn_f32); the backward (rk3d_backward_bn_f32) reads z for the x operand,
 *     masks d(activation) with the ReLU on the way out (dz) and reduces bn2's sum(dz), sum(dz zhat) next to the d(shift)
 *     partials: k12 [2][C], d(gamma), d(beta) come out of the same launch and rk_bn_bwd_dx_pre_f32 finishes d(z).
 *     abmi: [C][4] = (a, b, mean, invstd) as rk_bn_finish_tiles_f32 packs it.  Stride 1 / pad 0 planes with W % 4 == 0
 *     (56x56, 28x28, 112x112); RK_ERR_UNSUPPORTED otherwise (the caller normalises with rk_bn_apply_affine_f32). */
int rk3d_forward_bn_f32(const float* z, const float* abmi, const float* shift, float* y, int N, int T, int C, int H,
                        int W, int stride_T, int stride_H, int stride_W, int pad_T, int pad_H, int pad_W, int quantize,
                        rk_stream_t stream);
size_t rk3d_backward_bn_workspace_bytes(int N, int T, int C, int H, int W, int stride_T, int stride_H, int stride_W,
                                        int pad_T, int pad_H, int pad_W);
int rk3d_backward_bn_f32(const float* z, const float* abmi, const float* shift, const float* gy, float* dz, float* gshift,
                         float* k12, float* dgamma, float* dbeta, int N, int T, int C, int H, int W, int stride_T,
                         int stride_H, int stride_W, int pad_T, int pad_H, int pad_W, int normalize_grad, float t_factor,
                         int quantize, void* workspace, size_t workspace_bytes, rk_stream_t stream);
int rk_pw_tiles(int F, int P);
/* tiles of the training epilogues of ONE rk_pw_gemm_stats_f32 / rk_pw_gemm_bnbwd_f32 call with these arguments: the
 * second-generation fp32 kernels (rk_pw2.hip: v_mfma_f32_16x16x4_f32, no LDS) write one partial per 64 columns, the first
 * generation one per 128; rk_bn_finish_tiles_f32 tells the two apart from (tiles, count). */
int rk_pw_gemm_tiles(const float* A, int F, int K, int M, int P, int a_is_mk);
/* tuning / test hooks of the second-generation fp32 1x1 kernels: the operations of rk_pw_gemm_fused_f32 (prologue only) and
 * rk_pw_wgrad_pro_f32 with the kernel configuration given explicitly (<= 0 / < 0: the planner's choice).  rb: 16-row blocks
 * per wave (3, 4, 5); amode: 0 = A [M][K] by 16-byte loads, 1 = A [K][M] by dword loads, 2 = LDS image; inst: d(weight)
 * tile instance (rk_pw2.hip kWInst), stages: LDS ring depth (2, 3), splits: pixel splits.  RK_ERR_UNSUPPORTED: no instance. */
int rk_pw2_gemm_cfg_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk,
                        const float* ka, const float* kb, int relu_in, int rb, int amode, int ct, rk_stream_t stream);
/* rk_pw4.hip: the streaming GEMM of the shallow layers (54 -> 54 / 72 -> 72 channels; operand in registers, everything else
 * through a per-wave LDS-DMA record ring) regardless of the size threshold of the dispatch -- test / probe hook.
 * epi 0: Y = A relu?(ka x + kb)(X) (+ R); 1: + statistics tiles (stats [M][tiles] float4, as rk_pw_gemm_stats_f32);
 * 2: the BatchNorm-backward epilogue of rk_pw_gemm_bnbwd_f32 (bx, bpack [M][4], bred [M][tiles] float2); tiles =
 * ceil(F P / 64).  RK_ERR_UNSUPPORTED: no instance for (K, M). */
int rk_pw4_gemm_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk,
                    const float* ka, const float* kb, int relu_in, int epi, void* stats, const float* bx, const float* bpack,
                    void* bred, int tiles, rk_stream_t stream);
size_t rk_pw2_wgrad_workspace_bytes(int F, int K, int M, int P);
int rk_pw2_wgrad_cfg_f32(const float* dY, const float* X, float* dW, int F, int K, int M, int P, void* ws, size_t ws_bytes,
                         const float* ka, const float* kb, int relu_in, int inst, int stages, int splits, rk_stream_t stream);
/* Operand layout of the two training entry points: rk_pw_gemm_stats_f32 is the FORWARD of conv2 / conv3 and is instantiated
 * for a_is_mk = 1 (A = the weight [M][K]); rk_pw_gemm_bnbwd_f32 is conv2's D(INPUT) and is instantiated for a_is_mk = 0 (A =
 * the weight read as [K][M]).  Those are the layouts rubiksnet_amd/train_block.py uses; the other two combinations have
 * instances only in the first-generation kernel and return RK_ERR_UNSUPPORTED for shapes the later generations take
 * (M <= 224 rows, or the 257..288-row layers): not a layout the training block can produce. */
int rk_pw_gemm_stats_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P,
                         int a_is_mk, const float* ka, const float* kb, int relu_in, void* stats, int tiles,
                         rk_stream_t stream);
int rk_stem_conv3x3s2_stats_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                                void* stats, int tiles, rk_stream_t stream);
int rk_pw_gemm_bnbwd_f32(const float* A, const float* dY, const float* R, float* dZ, int F, int K, int M, int P,
                         int a_is_mk, const float* x, const float* abmi /* [M][4] = (a, b, mean, invstd) */, void* bred,
                         int tiles, rk_stream_t stream);
int rk_pw_wgrad_pro_f32(const float* dY, const float* X, float* dW, int F, int K, int M, int P, const float* ka,
                        const float* kb, int relu_in, void* ws, size_t ws_bytes, rk_stream_t stream);
int rk_pw_s2_wgrad_pro_f32(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win,
                           const float* ka, const float* kb, int relu_in, void* ws, size_t ws_bytes, rk_stream_t stream);
int rk_bn_finish_tiles_f32(const void* stats, int tiles, long long count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* a,
                           float* b, float* abmi /* [C][4] packed copy, may be NULL */, int C, float eps, float momentum,
                           long long* num_batches_tracked, rk_stream_t stream);
int rk_bn_tile_stats_f32(const float* x, void* stats, int F, int C, int P, rk_stream_t stream);
int rk_bn_apply_affine_bf16(const void* x, const float* a, const float* b, void* y, int F, int C, int P, int relu,
                            rk_stream_t stream);
int rk_bn_apply_affine_f32(const float* x, const float* a, const float* b, float* y, int F, int C, int P, int relu,
                           rk_stream_t stream);
int rk_bn_bwd_finish_tiles_f32(const void* bred, int tiles, long long count, float* k12, float* dgamma, float* dbeta,
                               int C, rk_stream_t stream);
int rk_bn_bwd_dx_pre_f32(const float* dz, const float* x, const float* gamma, const float* save_mean,
                         const float* save_invstd, const float* k12, const float* skip, float* dx, int F, int C, int P,
                         rk_stream_t stream);
int rk_bn_bwd_dx_pre_bf16(const void* dz, const void* x, const float* gamma, const float* save_mean,
                          const float* save_invstd, const float* k12, const void* skip, void* dx, int F, int C, int P,
                          rk_stream_t stream);
/* the statistics half of the training forward alone: save_mean / save_invstd / ab [2][C] (y = a x + b), running statistics
 * and *num_batches_tracked as nn.BatchNorm2d's forward; ws of rk_bn_workspace_bytes() bytes */
int rk_bn_stats_finish_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float* save_mean, float* save_invstd, float* ab, int F, int C, int P, float eps, float momentum,
                           long long* num_batches_tracked, void* ws, size_t ws_bytes, rk_stream_t stream);
int rk_bn_stats_finish_bf16(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float* save_mean, float* save_invstd, float* ab, int F, int C, int P, float eps, float momentum,
                            long long* num_batches_tracked, void* ws, size_t ws_bytes, rk_stream_t stream);
/* ... also leaving abmi [C][4] = (a, b, mean, invstd), the packed record rk2d_backward_bn_* reads (16-byte aligned) */
int rk_bn_stats_finish_abmi_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float* save_mean, float* save_invstd, float* ab, float* abmi, int F, int C, int P, float eps,
                                float momentum, long long* num_batches_tracked, void* ws, size_t ws_bytes, rk_stream_t stream);
int rk_bn_stats_finish_abmi_bf16(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float* save_mean, float* save_invstd, float* ab, float* abmi, int F, int C, int P, float eps,
                                 float momentum, long long* num_batches_tracked, void* ws, size_t ws_bytes, rk_stream_t stream);

/* ---- input side of the network on the device -- widening row f4 of SURVEY 8(f) ----------------------
 * Replaces, per batch instead of per sample on CPU workers, the reference's transform tail
 * Stack -> ToTorchFormatTensor(div=True) -> GroupNormalize (rubiksnet/transforms.py:329-363, :66-79; wired up
 * in scripts/test_models.py:136-143): hwc [nclips, H, W, CS] uint8 (CS = 3 * frames, channel-interleaved RGB
 * frames as Stack(roll=False) lays them out) -> chw [nclips, CS, H, W],
 * ((v / 255) - mean3[c % 3]) / std3[c % 3], every step rounded in fp32 as the reference's tensor ops round it
 * (the f32 result is bit-identical); H * W * CS % 16 == 0, hwc 16-byte aligned.                              */
int rk_clip_u8_to_chw_f32(const unsigned char* hwc, const float* mean3, const float* std3, float* chw, int nclips,
                          int H, int W, int CS, rk_stream_t stream);
int rk_clip_u8_to_chw_bf16(const unsigned char* hwc, const float* mean3, const float* std3, void* chw, int nclips,
                           int H, int W, int CS, rk_stream_t stream);

/* ---- squeeze-and-excitation gate of RubiksNet-Small -- widening row f3 of SURVEY 8(f) --------------
 * SELayer (rubiksnet/backbone.py:56-71): y = x * sigmoid(W2 relu(W1 mean_hw(x))).  x, y, dy, dx [F, C, P];
 * mean / gate / dgate [F, C] f32.  squeeze: mean over P; scale: y = x * gate; scale_backward: dx = dy * gate and
 * dgate = sum_p dy * x in one pass (the caller adds the squeeze's share dmean / P to dx through autograd).      */
#define RK_DECL_SE(SFX)                                                                                       \
    int rk_se_squeeze_##SFX(const void* x, float* mean, int F, int C, int P, rk_stream_t stream);             \
    int rk_se_scale_##SFX(const void* x, const float* gate, void* y, int F, int C, int P, rk_stream_t stream); \
    int rk_se_scale_backward_##SFX(const void* dy, const void* x, const float* gate, void* dx, float* dgate,  \
                                   int F, int C, int P, rk_stream_t stream);
RK_DECL_SE(f32)
RK_DECL_SE(bf16)
#undef RK_DECL_SE
/* The SE backward of the fused training block in 4 tensor passes instead of 5: dgate[f, c] = sum_p dy * x alone, and -- once
 * the two Linear layers' backward produced d(mean) [F, C] -- dx = dy * gate[f, c] + add[f, c] * add_scale (add_scale = 1 / P
 * is the squeeze's share, SELayer backward of rubiksnet/backbone.py:56-71).  fp32. */
/* The gate's two bias-free Linear layers on the squeezed vector, fused (one workgroup per frame): q [F, C], W1 [Cr, C], W2 [C, Cr]
 * (nn.Linear layout) -> h = relu(q W1^T) [F, Cr], g = sigmoid(h W2^T) [F, C]; backward: dgate [F, C] -> dq [F, C], dW1, dW2
 * (dpre2 [F, C], dpre1 [F, Cr]: caller-owned scratch).  RK_ERR_UNSUPPORTED for C > 2048 or Cr > 128. */
int rk_se_mlp_forward_f32(const float* q, const float* W1, const float* W2, float* h, float* g, int F, int C, int Cr,
                          rk_stream_t stream);
int rk_se_mlp_backward_f32(const float* dgate, const float* g, const float* h, const float* q, const float* W1, const float* W2,
                           float* dpre2, float* dpre1, float* dq, float* dW1, float* dW2, int F, int C, int Cr,
                           rk_stream_t stream);
int rk_se_dgate_f32(const float* dy, const float* x, float* dgate, int F, int C, int P, rk_stream_t stream);
int rk_se_scale_add_f32(const float* x, const float* gate, const float* add, float add_scale, float* y, int F, int C, int P,
                        rk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RUBIKS_HIP_H_ */
