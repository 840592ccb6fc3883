// rk_stem16.hip -- the backbone's first layer under bf16 autocast: Conv3x3(3, width, stride 2, pad 1, no bias)
// (rubiksnet/backbone.py:154) on the fp32 clip, bf16 out -- forward and d(weight) (the clip needs no gradient).
// MIOpen ran it as cast + NCHW <-> NHWC transposes + implicit GEMM: 0.89 ms forward + 0.56 ms d(weight) at
// [256, 3, 224, 224] -> 72 channels (tools/stem_probe.py), for a layer that writes 462 MB (58 us at 8 TB/s) and is 12.5 GFLOP.
//
// K = 27 taps fit ONE v_mfma_f32_16x16x32_bf16 step (5 zero columns), so there is no K loop:
//   * forward (k_stem16_forward): a workgroup owns 4 output rows of a frame, a wave one of them.  The 9 input rows x 3 channels
//     they touch are staged in LDS as bf16 (the rounding autocast applies to the clip), left halo included; a B fragment (16
//     output pixels x 27 taps) is 8 two-byte LDS reads per lane at tap offsets computed once; the weight fragments (<= 8 row
//     blocks) sit in registers for the whole kernel.  Results are transposed through LDS so that every output channel row leaves
//     as 16-byte pieces (the layer is bound by its 462 MB of stores).
//   * d(weight) (k_stem16_wgrad): dW[co][tap] = sum over pixels of dY[co][pixel] * im2col[tap][pixel]: the reduction index is
//     the pixel, contiguous in dY (A fragments: 16-byte global loads) and a stride-2 walk along a staged input row for the B
//     fragment.  A workgroup walks bands of 8 output rows (persistent grid), its waves the 32-pixel groups of a band; the per-
//     workgroup partial matrices are summed in a fixed order by k_stem16_reduce.
// Arithmetic: bf16 operands (weight and clip rounded as autocast rounds them), fp32 accumulation, one rounding of the forward.
#include <type_traits>
#include "rk_common.hpp"

namespace rk {
namespace stem16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTaps = 27;

struct SDims {
    int F, C, H, W, Ho, Wo;
    int pitch;                 // elements per staged input row: W + 4 (column c at index c + 2; index 1 = the zero left halo)
    int units;                 // F * bands
    int bands;                 // bands of output rows per frame
};

__device__ __forceinline__ unsigned bf16_bits(float f) {
    return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(f));
}
__device__ __forceinline__ unsigned lds_u16(const char* p) { return (unsigned)*reinterpret_cast<const unsigned short*>(p); }

// input rows 2 ho0 - 1 .. 2 ho0 - 2 + NR of the 3 channels of frame f -> xs[ci * NR + r][pitch] as bf16
template <int NR>
__device__ __forceinline__ void stage_rows(const float* __restrict__ X, char* xs, const SDims& d, int f, int ho0) {
    const int q = d.W / 4;                                            // 16-byte pieces per row
    const int total = 3 * NR * q;
    for (int t = threadIdx.x; t < total; t += kBlock) {
        const int rr = t / q, j = t - rr * q;
        const int ci = rr / NR, r = rr - ci * NR;
        const int hi = 2 * ho0 - 1 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hi >= 0 && hi < d.H) v = *reinterpret_cast<const float4*>(X + (((size_t)f * 3 + ci) * d.H + hi) * d.W + 4 * j);
        unsigned* dst = reinterpret_cast<unsigned*>(xs + ((size_t)rr * d.pitch + 4 * j + 2) * 2);
        dst[0] = bf16_bits(v.x) | (bf16_bits(v.y) << 16);
        dst[1] = bf16_bits(v.z) | (bf16_bits(v.w) << 16);
    }
    for (int t = threadIdx.x; t < 3 * NR; t += kBlock) *reinterpret_cast<unsigned*>(xs + (size_t)t * d.pitch * 2) = 0u;   // indices 0, 1
}

// weight [C][27] fp32 -> the lane's A fragments (rows 16 rb + m, taps 8 g .. 8 g + 7; zeros outside)
template <int RBN>
__device__ __forceinline__ void weight_frags(const float* __restrict__ Wt, int C, int lane, bf16x8 (&a)[RBN]) {
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rb = 0; rb < RBN; ++rb) {
        const int row = 16 * rb + m;
        u32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k0 = 8 * g + 2 * j, k1 = k0 + 1;
            const float v0 = (row < C && k0 < kTaps) ? Wt[row * kTaps + k0] : 0.f;
            const float v1 = (row < C && k1 < kTaps) ? Wt[row * kTaps + k1] : 0.f;
            t[j] = bf16_bits(v0) | (bf16_bits(v1) << 16);
        }
        a[rb] = __builtin_bit_cast(bf16x8, t);
    }
}

// ---------------------------------------------------------------------------------------------
template <int RBN>
__global__ __launch_bounds__(kBlock, 2) void k_stem16_forward(const float* __restrict__ Wt, const float* __restrict__ X,
                                                              __hip_bfloat16* __restrict__ Y, SDims d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NR = 9;                                            // input rows of 4 output rows
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    char* xs = lds;
    const int xbytes = (3 * NR * d.pitch * 2 + 15) & ~15;
    char* ys = lds + xbytes + (size_t)wave * d.C * d.Wo * 2;          // this wave's [C][Wo] bf16

    bf16x8 a[RBN];
    weight_frags<RBN>(Wt, d.C, lane, a);
    // this lane's 8 taps: byte offset of (ci, 2 r + kh, kw + 1) with r = wave; taps >= 27 read tap 0 and are zeroed
    int off[8];
    bool live[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        live[j] = k < kTaps;
        const int kk = live[j] ? k : 0;
        const int ci = kk / 9, kh = (kk - 9 * ci) / 3, kw = kk - 9 * ci - 3 * kh;
        off[j] = ((ci * NR + 2 * wave + kh) * d.pitch + kw + 1) * 2;
    }
    const int ncb = d.Wo / 16;
    const int ppr = d.Wo / 8;                                         // 16-byte pieces per channel row
    const int pieces = d.C * ppr;
    // persistent: the weight fragments and tap offsets are set up once per workgroup (a workgroup per band paid ~3 us of
    // dependent loads for 5 us of work)
    for (int u = blockIdx.x; u < d.units; u += gridDim.x) {
        const int band = u % d.bands, f = u / d.bands;
        const int ho0 = 4 * band;
        __syncthreads();                                             // (the previous band's readers are done)
        stage_rows<NR>(X, xs, d, f, ho0);
        __syncthreads();
        const int ho = ho0 + wave;
        if (ho >= d.Ho) continue;                                    // (wave-uniform; the barriers above are reached by every wave)
        for (int cb = 0; cb < ncb; ++cb) {
            const char* px = xs + 4 * (16 * cb + n);
            u32x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned lo = live[2 * j] ? lds_u16(px + off[2 * j]) : 0u, hi = live[2 * j + 1] ? lds_u16(px + off[2 * j + 1]) : 0u;
                t[j] = lo | (hi << 16);
            }
            const bf16x8 b = __builtin_bit_cast(bf16x8, t);
#pragma unroll
            for (int rb = 0; rb < RBN; ++rb) {
                const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = 16 * rb + 4 * g + i;
                    if (ch < d.C) *reinterpret_cast<unsigned short*>(ys + ((size_t)ch * d.Wo + 16 * cb + n) * 2) = (unsigned short)bf16_bits(acc[i]);
                }
            }
        }
        // the wave's [C][Wo] block -> Y[f][ch][ho][*]: 16-byte pieces (LDS operations of one wave execute in order)
        __hip_bfloat16* yrow = Y + ((size_t)f * d.C * d.Ho + ho) * d.Wo;
        for (int t = lane; t < pieces; t += kWave) {
            const int ch = t / ppr, pj = t - ch * ppr;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ys + ((size_t)ch * d.Wo + 8 * pj) * 2);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(yrow + (size_t)ch * d.Ho * d.Wo + 8 * pj));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// partial dW of workgroup b: ws[b][C][27] fp32
template <int RBN>
__global__ __launch_bounds__(kBlock, 2) void k_stem16_wgrad(const __hip_bfloat16* __restrict__ dY, const float* __restrict__ X,
                                                            float* __restrict__ ws, SDims d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NR = 17;                                           // input rows of 8 output rows
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int m = lane & 15, g = lane >> 4;
    char* xs = lds;

    f32x4 acc[RBN][2];
#pragma unroll
    for (int rb = 0; rb < RBN; ++rb) { acc[rb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // this lane's two taps (column blocks 0, 1 of the im2col operand): byte offset of (ci, kh, kw + 1) in the staged rows
    int toff[2];
    bool tlive[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int k = m + 16 * c;
        tlive[c] = k < kTaps;
        const int kk = tlive[c] ? k : 0;
        const int ci = kk / 9, kh = (kk - 9 * ci) / 3, kw = kk - 9 * ci - 3 * kh;
        toff[c] = ((ci * NR + kh) * d.pitch + kw + 1) * 2;
    }
    int arow[RBN];
#pragma unroll
    for (int rb = 0; rb < RBN; ++rb) { const int row = 16 * rb + m; arow[rb] = row < d.C ? row : d.C - 1; }   // (rows past C: copies, not stored)
    const int ngroups = 8 * d.Wo / 32;                               // 32-pixel groups of a band
    const size_t plane = (size_t)d.Ho * d.Wo;

    for (int u = blockIdx.x; u < d.units; u += gridDim.x) {
        const int band = u % d.bands, f = u / d.bands;
        const int ho0 = 8 * band;
        __syncthreads();                                             // (the previous band's readers are done)
        stage_rows<NR>(X, xs, d, f, ho0);
        __syncthreads();
        const __hip_bfloat16* dyf = dY + (size_t)f * d.C * plane + (size_t)ho0 * d.Wo;
        for (int grp = wave; grp < ngroups; grp += 4) {
            const int p0 = 32 * grp + 8 * g;                         // this lane's 8 consecutive pixels: one output row (Wo % 8 == 0)
            const int r = p0 / d.Wo, wo0 = p0 - r * d.Wo;
            bf16x8 a[RBN];
#pragma unroll
            for (int rb = 0; rb < RBN; ++rb) a[rb] = *reinterpret_cast<const bf16x8*>(dyf + (size_t)arow[rb] * plane + p0);
            const char* px = xs + (2 * r * d.pitch) * 2 + 4 * wo0;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u32x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned lo = lds_u16(px + toff[c] + 4 * (2 * j)), hi = lds_u16(px + toff[c] + 4 * (2 * j + 1));
                    t[j] = tlive[c] ? (lo | (hi << 16)) : 0u;
                }
                const bf16x8 b = __builtin_bit_cast(bf16x8, t);
#pragma unroll
                for (int rb = 0; rb < RBN; ++rb) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rb], b, acc[rb][c], 0, 0, 0);
            }
        }
    }
    // the 4 waves' tiles -> LDS [wave][16 RBN][32], summed in wave order
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int rb = 0; rb < RBN; ++rb)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(wave * 16 * RBN + 16 * rb + 4 * g + i) * 32 + 16 * c + m] = acc[rb][c][i];
    __syncthreads();
    float* out = ws + (size_t)blockIdx.x * d.C * kTaps;
    for (int t = threadIdx.x; t < d.C * kTaps; t += kBlock) {
        const int row = t / kTaps, k = t - row * kTaps;
        const int at = row * 32 + k;
        out[t] = ((red[at] + red[16 * RBN * 32 + at]) + red[2 * 16 * RBN * 32 + at]) + red[3 * 16 * RBN * 32 + at];
    }
}

// dW[i] = sum of the B partials in a fixed order: 16 slices of the partial range per output (thread (slice, output): front to
// back), then the slice sums in slice order
__global__ __launch_bounds__(kBlock) void k_stem16_reduce(const float* __restrict__ ws, float* __restrict__ dW, int n, int B) {
    __shared__ float part[16][16];
    const int o = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + o;
    const int per = (B + 15) / 16, b0 = slice * per, b1 = (b0 + per) < B ? (b0 + per) : B;
    float s = 0.f;
    if (i < n) {
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += ws[(size_t)b * n + i];
    }
    part[slice][o] = s;
    __syncthreads();
    if (slice == 0 && i < n) {
        float t = part[0][o];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += part[k][o];
        dW[i] = t;
    }
}

inline int make_sdims(SDims& d, int F, int Cin, int C, int H, int W, int rows_per_band) {
    if (F <= 0 || C <= 0 || H <= 0 || W <= 0) return RK_ERR_BAD_DIMS;
    if (Cin != 3 || C > 128 || H % 2 || W % 32) return RK_ERR_UNSUPPORTED;       // (Wo % 16 == 0: whole column blocks)
    d.F = F; d.C = C; d.H = H; d.W = W; d.Ho = H / 2; d.Wo = W / 2;
    d.pitch = W + 4;
    d.bands = (d.Ho + rows_per_band - 1) / rows_per_band;
    d.units = F * d.bands;
    if ((long long)F * C * d.Ho * d.Wo >= (1ll << 31)) return RK_ERR_UNSUPPORTED;
    return RK_OK;
}
constexpr int kWgradGroups = 512;
inline int wgrad_groups(const SDims& d) { return d.units < kWgradGroups ? d.units : kWgradGroups; }

template <typename K> int raise_lds(K kernel, size_t lds, DynLdsRaised& raised) {      // per instance and device (rk_common.hpp)
    return raise_dynamic_lds(reinterpret_cast<const void*>(kernel), lds, raised);
}

template <int RBN>
int launch_forward(const float* Wt, const float* X, __hip_bfloat16* Y, const SDims& d, hipStream_t stream) {
    const size_t lds = ((size_t)(3 * 9 * d.pitch * 2 + 15) & ~(size_t)15) + (size_t)4 * d.C * d.Wo * 2;
    static DynLdsRaised raised;
    if (int rc = raise_lds(&k_stem16_forward<RBN>, lds, raised)) return rc;
    const int groups = d.units < 1024 ? d.units : 1024;              // two resident workgroups per CU, two rounds
    hipLaunchKernelGGL((k_stem16_forward<RBN>), dim3((unsigned)groups), dim3(kBlock), lds, stream, Wt, X, Y, d);
    return launch_status();
}
template <int RBN>
int launch_wgrad(const __hip_bfloat16* dY, const float* X, float* ws, const SDims& d, hipStream_t stream) {
    size_t lds = (size_t)3 * 17 * d.pitch * 2;
    const size_t red = (size_t)4 * 16 * RBN * 32 * 4;
    lds = lds > red ? lds : red;
    static DynLdsRaised raised;
    if (int rc = raise_lds(&k_stem16_wgrad<RBN>, lds, raised)) return rc;
    hipLaunchKernelGGL((k_stem16_wgrad<RBN>), dim3((unsigned)wgrad_groups(d)), dim3(kBlock), lds, stream, dY, X, ws, d);
    return launch_status();
}

}  // namespace stem16
}  // namespace rk

using namespace rk;
using namespace rk::stem16;

extern "C" {

// 1 when the bf16-output stem kernels take the layer
int rk_stem16_supported(int F, int Cin, int Cout, int Hin, int Win) {
    SDims d;
    if (make_sdims(d, F, Cin, Cout, Hin, Win, 4)) return 0;
    return (Hin / 2) % 8 == 0 ? 1 : 0;                              // (d(weight) walks bands of 8 output rows)
}
// Y [F, Cout, Hin/2, Win/2] bf16 = conv3x3 / stride 2 / pad 1 of the fp32 clip X [F, 3, Hin, Win] with W [Cout][3][3][3] fp32
// (both rounded to bf16 as autocast rounds them; fp32 accumulation)
int rk_stem_conv3x3s2_bf16out(const float* W, const float* X, void* Y, int F, int Cin, int Cout, int Hin, int Win,
                              rk_stream_t stream_) {
    if (!W || !X || !Y) return RK_ERR_NULL_POINTER;
    SDims d;
    if (int rc = make_sdims(d, F, Cin, Cout, Hin, Win, 4)) return rc;
    if (((uintptr_t)X & 15) || ((uintptr_t)Y & 15)) return RK_ERR_BAD_DIMS;
    hipStream_t stream = (hipStream_t)stream_;
    __hip_bfloat16* y = (__hip_bfloat16*)Y;
    if (Cout <= 64) return launch_forward<4>(W, X, y, d, stream);
    if (Cout <= 80) return launch_forward<5>(W, X, y, d, stream);
    return launch_forward<8>(W, X, y, d, stream);
}
size_t rk_stem_wgrad16_workspace_bytes(int F, int Cin, int Cout, int Hin, int Win) {
    SDims d;
    if (make_sdims(d, F, Cin, Cout, Hin, Win, 8)) return 0;
    return (size_t)wgrad_groups(d) * Cout * kTaps * sizeof(float);
}
// dW [Cout][3][3][3] fp32 = sum over frames and pixels of dY (bf16) x im2col(X rounded to bf16)
int rk_stem_wgrad3x3s2_bf16(const void* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win, void* ws,
                            size_t ws_bytes, rk_stream_t stream_) {
    if (!dY || !X || !dW) return RK_ERR_NULL_POINTER;
    SDims d;
    if (int rc = make_sdims(d, F, Cin, Cout, Hin, Win, 8)) return rc;
    if (d.Ho % 8) return RK_ERR_UNSUPPORTED;
    if (((uintptr_t)X & 15) || ((uintptr_t)dY & 15)) return RK_ERR_BAD_DIMS;
    if (!ws || ws_bytes < rk_stem_wgrad16_workspace_bytes(F, Cin, Cout, Hin, Win)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const __hip_bfloat16* dy = (const __hip_bfloat16*)dY;
    int rc;
    if (Cout <= 64) rc = launch_wgrad<4>(dy, X, (float*)ws, d, stream);
    else if (Cout <= 80) rc = launch_wgrad<5>(dy, X, (float*)ws, d, stream);
    else rc = launch_wgrad<8>(dy, X, (float*)ws, d, stream);
    if (rc) return rc;
    const int n = Cout * kTaps;
    hipLaunchKernelGGL(k_stem16_reduce, dim3((n + 15) / 16), dim3(kBlock), 0, stream, (const float*)ws, dW, n, wgrad_groups(d));
    return launch_status();
}

}  // extern "C"
