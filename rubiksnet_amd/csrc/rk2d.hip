// rk2d.hip -- RubiksShift2D for gfx950: kernels + C-ABI entry points (include/rubiks_hip.h).
//
// Semantics follow the reference's K6-K9 (cuda_src/rubiks2d_kernels.cu:94-397) and its
// host glue (cuda_src/rubiks.cpp:44-155).  Layout [N,C,H,W], shift [2,C] = (H,W).
// Same mapping idea as rk3d_generic.hpp: one (n, c) plane per group of E = 64/128/256
// threads, per-channel quantities hoisted, no per-element division, no float atomics:
// d(shift) goes wave-shuffle -> LDS -> one partial per plane -> fixed-order fp64 finalize.
// f16 / bf16 tensors are computed in fp32 and rounded once on store; only the quantize
// position arithmetic is done in the storage type, because it decides WHICH element is
// gathered (the reference instantiates the whole kernel at c10::Half).
#include "rk_common.hpp"

using namespace rk;

namespace {

struct Dims2 {
    int N, C, H, W, Ho, Wo;
    int sH, sW, pH, pW;
    int E, logE;
};

// rubiks2d_kernels.cu:69-73
template <typename CT> __device__ __forceinline__ int floor_fast(CT v) {
    const int iv = (int)v;
    return iv - (v < (CT)iv ? 1 : 0);
}

// Position arithmetic of the quantize branch, rubiks2d_kernels.cu:117-118 / :295-296 with
// round_fast (:76-82), in the tensor's own arithmetic: every operation rounds to T.
template <typename T> struct QuantPos {
    using CT = typename Compute<T>::type;
    __device__ static __forceinline__ int nearest(int base, CT off) {
        const CT v = (CT)base + off;
        return (v < (CT)0.0f) ? (int)(v - (CT)0.5f) : (int)(v + (CT)0.5f);
    }
};
template <> struct QuantPos<__half> {
    __device__ static __forceinline__ float rnd(float v) { return __half2float(__float2half(v)); }
    __device__ static __forceinline__ int nearest(int base, float off) {
        const float v = rnd(rnd((float)base) + off);
        return (v < 0.0f) ? (int)rnd(v - 0.5f) : (int)rnd(v + 0.5f);
    }
};
template <> struct QuantPos<__hip_bfloat16> {
    __device__ static __forceinline__ float rnd(float v) { return __bfloat162float(__float2bfloat16(v)); }
    __device__ static __forceinline__ int nearest(int base, float off) {
        const float v = rnd(rnd((float)base) + off);
        return (v < 0.0f) ? (int)rnd(v - 0.5f) : (int)rnd(v + 0.5f);
    }
};

// rubiks2d_kernels.cu:60-66
template <typename CT> __device__ __forceinline__ CT interp2d(CT p00, CT p01, CT p10, CT p11, CT rH, CT rW) {
    return p00 * (1 - rH) * (1 - rW) + p01 * (1 - rH) * rW + p10 * rH * (1 - rW) + p11 * rH * rW;
}

__device__ __forceinline__ bool my_plane2(const Dims2& d, int& n, int& c, int& e) {
    const int sub = threadIdx.x >> d.logE;
    e = threadIdx.x & (d.E - 1);
    const long long plane = (long long)blockIdx.x * (kBlock >> d.logE) + sub;
    const bool valid = plane < (long long)d.N * d.C;
    const long long q = valid ? plane : 0;
    c = (int)(q % d.C);
    n = (int)(q / d.C);
    return valid;
}

// rubiks2d_kernels.cu:298-300 / :360-362: the in-kernel stride is uint32, so a negative
// position wraps; the wrapped value always fails the divisibility or the bounds test.
// Net effect: negatives are skipped -- which signed C remainder/division also gives.
__device__ __forceinline__ int unmap2(int p, int s, int lim) {
    if (p < 0 || p % s != 0) return -1;
    const int q = p / s;
    return q < lim ? q : -1;
}

// ------------------------------------------------------------------------------ K6
template <typename T, bool QUANT>
__global__ __launch_bounds__(kBlock) void k2d_forward(const T* __restrict__ x, const T* __restrict__ shift,
                                                      T* __restrict__ y, Dims2 d) {
    using CT = typename Compute<T>::type;
    int n, c, e;
    if (!my_plane2(d, n, c, e)) return;
    const CT offH = ld(shift + c), offW = ld(shift + d.C + c);
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
    const T* xp = x + ((size_t)n * d.C + c) * HW;
    T* yp = y + ((size_t)n * d.C + c) * HWo;
    int ho = e / d.Wo, wo = e - ho * d.Wo;
    const int dh = d.E / d.Wo, dw = d.E - dh * d.Wo;
    const int iH = floor_fast(offH), iW = floor_fast(offW);
    const CT rH = offH - (CT)iH, rW = offW - (CT)iW;
    for (int i = e; i < HWo; i += d.E) {
        const int bH = ho * d.sH - d.pH, bW = wo * d.sW - d.pW;
        if (QUANT) {  // out-of-range source: y is left untouched (rubiks2d_kernels.cu:116-121)
            const int th = QuantPos<T>::nearest(bH, offH), tw = QuantPos<T>::nearest(bW, offW);
            if (th >= 0 && th < d.H && tw >= 0 && tw < d.W) yp[i] = xp[th * d.W + tw];
        } else {
            const int h0 = bH + iH, w0 = bW + iW;
            const bool mh0 = h0 >= 0 && h0 < d.H, mh1 = h0 + 1 >= 0 && h0 + 1 < d.H;
            const bool mw0 = w0 >= 0 && w0 < d.W, mw1 = w0 + 1 >= 0 && w0 + 1 < d.W;
            const int o = h0 * d.W + w0;
            CT p00 = 0, p01 = 0, p10 = 0, p11 = 0;
            if (mh0 && mw0) p00 = ld(xp + o);
            if (mh0 && mw1) p01 = ld(xp + o + 1);
            if (mh1 && mw0) p10 = ld(xp + o + d.W);
            if (mh1 && mw1) p11 = ld(xp + o + d.W + 1);
            st(yp + i, interp2d(p00, p01, p10, p11, rH, rW));
        }
        wo += dw; ho += dh;
        if (wo >= d.Wo) { wo -= d.Wo; ++ho; }
    }
}

// ------------------------------------------------------------------------------ K7
// partials part[c][2][P], P = N, p = n
template <typename T>
__global__ __launch_bounds__(kBlock) void k2d_backward_shift(const T* __restrict__ gy, const T* __restrict__ x,
                                                             const T* __restrict__ shift,
                                                             typename Compute<T>::type* __restrict__ part, Dims2 d) {
    using CT = typename Compute<T>::type;
    __shared__ CT red[2][kBlock / kWave];
    int n, c, e;
    const bool valid = my_plane2(d, n, c, e);
    CT aH = 0, aW = 0;
    if (valid) {
        const CT offH = ld(shift + c), offW = ld(shift + d.C + c);
        const int iH = floor_fast(offH), iW = floor_fast(offW);
        CT rH = offH - (CT)iH, rW = offW - (CT)iW;
        const CT tol = (CT)1e-7f;                         // rubiks2d_kernels.cu:189-200
        bool hint = false, wint = false;
        if (tol > rH && rH > -tol) { hint = true; rH = 0; }
        if (tol > rW && rW > -tol) { wint = true; rW = 0; }
        const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
        const T* xp = x + ((size_t)n * d.C + c) * HW;
        const T* gp = gy + ((size_t)n * d.C + c) * HWo;
        int ho = e / d.Wo, wo = e - ho * d.Wo;
        const int dh = d.E / d.Wo, dw = d.E - dh * d.Wo;
        auto at = [&](int h, int w) -> CT {
            return (h >= 0 && h < d.H && w >= 0 && w < d.W) ? ld(xp + h * d.W + w) : (CT)0;
        };
        for (int i = e; i < HWo; i += d.E) {
            const int h0 = ho * d.sH - d.pH + iH, w0 = wo * d.sW - d.pW + iW;
            const CT p00 = at(h0, w0), p01 = at(h0, w0 + 1), p10 = at(h0 + 1, w0), p11 = at(h0 + 1, w0 + 1);
            CT dH = (1 - rW) * (p10 - p00) + rW * (p11 - p01);          // :215-221
            CT dW = (1 - rH) * (p01 - p00) + rH * (p11 - p10);
            if (hint || wint) {                                          // :224-253, 3x3 around (h0, w0)
                if (hint)
                    dH = (CT)0.5f * ((1 - rW) * (at(h0 + 1, w0) - at(h0 - 1, w0)) +
                                     rW * (at(h0 + 1, w0 + 1) - at(h0 - 1, w0 + 1)));
                if (wint)
                    dW = (CT)0.5f * ((1 - rH) * (at(h0, w0 + 1) - at(h0, w0 - 1)) +
                                     rH * (at(h0 + 1, w0 + 1) - at(h0 + 1, w0 - 1)));
            }
            const CT og = ld(gp + i);
            aH += dH * og;
            aW += dW * og;
            wo += dw; ho += dh;
            if (wo >= d.Wo) { wo -= d.Wo; ++ho; }
        }
    }
    aH = group_sum(aH, d.E, red[0]);
    aW = group_sum(aW, d.E, red[1]);
    if (valid && e == 0) {
        CT* o = part + (size_t)c * 2 * d.N + n;
        o[0] = aH;
        o[d.N] = aW;
    }
}

// row-sum (rubiks.cpp:140-143) + K9 (rubiks2d_kernels.cu:381-397), one workgroup per channel
template <typename T>
__global__ __launch_bounds__(kBlock) void k2d_finalize(const typename Compute<T>::type* __restrict__ part,
                                                       T* __restrict__ gshift, int C, int P, int normalize) {
    using CT = typename Compute<T>::type;
    __shared__ double red[2][kBlock / kWave];
    const int c = blockIdx.x;
    const CT* p = part + (size_t)c * 2 * P;
    double s[2] = {0, 0};
    for (int k = 0; k < 2; ++k)
        for (int i = threadIdx.x; i < P; i += kBlock) s[k] += (double)p[(size_t)k * P + i];
    for (int k = 0; k < 2; ++k) s[k] = group_sum(s[k], kBlock, red[k]);
    if (threadIdx.x == 0) {
        CT gH = (CT)s[0], gW = (CT)s[1];
        if (normalize) {
            const CT mag = sqrt(gH * gH + gW * gW);
            if (mag > 0) { gH = gH / mag; gW = gW / mag; }
        }
        st(gshift + c, gH);
        st(gshift + C + c, gW);
    }
}

// ------------------------------------------------------------------------------ K8
template <typename T, bool QUANT>
__global__ __launch_bounds__(kBlock) void k2d_backward_input(const T* __restrict__ gy, const T* __restrict__ shift,
                                                             T* __restrict__ gx, Dims2 d) {
    using CT = typename Compute<T>::type;
    int n, c, e;
    if (!my_plane2(d, n, c, e)) return;
    const CT nH = -ld(shift + c), nW = -ld(shift + d.C + c);
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
    const T* gp = gy + ((size_t)n * d.C + c) * HWo;
    T* xp = gx + ((size_t)n * d.C + c) * HW;
    int h = e / d.W, w = e - h * d.W;
    const int dh = d.E / d.W, dw = d.E - dh * d.W;
    const int flH = floor_fast(nH), flW = floor_fast(nW);
    const CT rH = nH - (CT)flH, rW = nW - (CT)flW;
    const bool zero = (nW == 0 && nH == 0);                              // rubiks2d_kernels.cu:322
    auto gat = [&](int ph, int pw) -> CT {
        const int a = unmap2(ph, d.sH, d.Ho), b = unmap2(pw, d.sW, d.Wo);
        return (a >= 0 && b >= 0) ? ld(gp + a * d.Wo + b) : (CT)0;
    };
    for (int i = e; i < HW; i += d.E) {
        const int oH = h + d.pH, oW = w + d.pW;
        if (QUANT) {   // skipped positions leave gx untouched (rubiks2d_kernels.cu:294-309)
            const int a = unmap2(QuantPos<T>::nearest(oH, nH), d.sH, d.Ho);
            const int b = unmap2(QuantPos<T>::nearest(oW, nW), d.sW, d.Wo);
            if (a >= 0 && b >= 0) xp[i] = gp[a * d.Wo + b];
        } else if (zero) {
            st(xp + i, gat(oH, oW));
        } else {
            st(xp + i, interp2d(gat(oH + flH, oW + flW), gat(oH + flH, oW + flW + 1), gat(oH + flH + 1, oW + flW),
                                gat(oH + flH + 1, oW + flW + 1), rH, rW));
        }
        w += dw; h += dh;
        if (w >= d.W) { w -= d.W; ++h; }
    }
}

int make_dims2(Dims2& d, int N, int C, int H, int W, int sH, int sW, int pH, int pW) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return RK_ERR_BAD_DIMS;
    if (sH <= 0 || sW <= 0 || pH < 0 || pW < 0) return RK_ERR_BAD_STRIDE;
    d.N = N; d.C = C; d.H = H; d.W = W; d.sH = sH; d.sW = sW; d.pH = pH; d.pW = pW;
    d.Ho = out_len(H, sH, pH); d.Wo = out_len(W, sW, pW);
    if (d.Ho <= 0 || d.Wo <= 0) return RK_ERR_BAD_DIMS;
    // the reference's accessors index with uint32 (utils_cuda.h:12-13)
    if ((long long)N * C * H * W > 0x7fffffffLL || (long long)N * C * d.Ho * d.Wo > 0x7fffffffLL)
        return RK_ERR_BAD_DIMS;
    return RK_OK;
}

void set_group2(Dims2& d, int plane_elems) {
    d.E = pow2_at_least(plane_elems, kWave, kBlock);
    d.logE = (d.E == 64) ? 6 : (d.E == 128 ? 7 : 8);
}

unsigned grid2(const Dims2& d) {
    const int per_block = kBlock / d.E;
    return (unsigned)(((long long)d.N * d.C + per_block - 1) / per_block);
}

template <typename T>
int forward2(const void* x_, const void* shift_, void* y_, int N, int C, int H, int W, int sH, int sW, int pH,
             int pW, int quantize, rk_stream_t stream_) {
    const T* x = (const T*)x_; const T* shift = (const T*)shift_; T* y = (T*)y_;
    if (!x || !shift || !y) return RK_ERR_NULL_POINTER;
    Dims2 d;
    if (int rc = make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    set_group2(d, d.Ho * d.Wo);
    if (quantize)
        hipLaunchKernelGGL((k2d_forward<T, true>), dim3(grid2(d)), dim3(kBlock), 0, stream, x, shift, y, d);
    else
        hipLaunchKernelGGL((k2d_forward<T, false>), dim3(grid2(d)), dim3(kBlock), 0, stream, x, shift, y, d);
    return launch_status();
}

template <typename T>
int backward2(const void* gy_, const void* x_, const void* shift_, void* gx_, void* gshift_, int N, int C, int H,
              int W, int sH, int sW, int pH, int pW, int normalize_grad, int enable_shift_grad, int quantize,
              void* ws, size_t ws_bytes, rk_stream_t stream_) {
    using CT = typename Compute<T>::type;
    const T* gy = (const T*)gy_; const T* x = (const T*)x_; const T* shift = (const T*)shift_;
    T* gx = (T*)gx_; T* gshift = (T*)gshift_;
    if (!gy || !shift || !gx) return RK_ERR_NULL_POINTER;
    if (enable_shift_grad && (!x || !gshift)) return RK_ERR_NULL_POINTER;
    Dims2 d;
    if (int rc = make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (enable_shift_grad) {                                              // rubiks.cpp:126-149
        const size_t need = (size_t)C * 2 * N * sizeof(CT);
        if (!ws || ws_bytes < need) return RK_ERR_WORKSPACE;
        CT* part = (CT*)ws;
        set_group2(d, d.Ho * d.Wo);
        hipLaunchKernelGGL((k2d_backward_shift<T>), dim3(grid2(d)), dim3(kBlock), 0, stream, gy, x, shift, part, d);
        hipLaunchKernelGGL((k2d_finalize<T>), dim3(C), dim3(kBlock), 0, stream, (const CT*)part, gshift, C, N,
                           normalize_grad);
    }
    set_group2(d, d.H * d.W);                                             // rubiks.cpp:151-153
    if (quantize)
        hipLaunchKernelGGL((k2d_backward_input<T, true>), dim3(grid2(d)), dim3(kBlock), 0, stream, gy, shift, gx, d);
    else
        hipLaunchKernelGGL((k2d_backward_input<T, false>), dim3(grid2(d)), dim3(kBlock), 0, stream, gy, shift, gx, d);
    return launch_status();
}

}  // namespace

extern "C" {

size_t rk2d_backward_workspace_bytes(int N, int C, int H, int W, int sH, int sW, int pH, int pW, int elem_size) {
    (void)H; (void)W; (void)sH; (void)sW; (void)pH; (void)pW;
    if (N <= 0 || C <= 0) return 0;
    return (size_t)C * 2 * N * (size_t)(elem_size == 8 ? 8 : 4);   // partials are fp32 (fp64 for f64)
}

#define RK_DEF_2D(SFX, TYPE, CTYPE)                                                                              \
    int rk2d_forward_##SFX(const CTYPE* x, const CTYPE* shift, CTYPE* y, int N, int C, int H, int W, int sH,     \
                           int sW, int pH, int pW, int quantize, rk_stream_t stream) {                           \
        return forward2<TYPE>(x, shift, y, N, C, H, W, sH, sW, pH, pW, quantize, stream);                        \
    }                                                                                                            \
    int rk2d_backward_##SFX(const CTYPE* gy, const CTYPE* x, const CTYPE* shift, CTYPE* gx, CTYPE* gshift,       \
                            int N, int C, int H, int W, int sH, int sW, int pH, int pW, int normalize_grad,     \
                            int enable_shift_grad, int quantize, void* ws, size_t ws_bytes, rk_stream_t stream) { \
        return backward2<TYPE>(gy, x, shift, gx, gshift, N, C, H, W, sH, sW, pH, pW, normalize_grad,             \
                               enable_shift_grad, quantize, ws, ws_bytes, stream);                               \
    }
RK_DEF_2D(f32, float, float)
RK_DEF_2D(f64, double, double)
RK_DEF_2D(f16, __half, void)
RK_DEF_2D(bf16, __hip_bfloat16, void)
#undef RK_DEF_2D

}  // extern "C"
