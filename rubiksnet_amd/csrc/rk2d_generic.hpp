// rk2d_generic.hpp -- RubiksShift2D "any configuration" kernels for gfx950 (K6-K9 of
// cuda_src/rubiks2d_kernels.cu:94-397): any stride / padding, quantize, integer-shift branch, fp32 / fp64 /
// fp16 / bf16.  Same mapping idea as rk3d_generic.hpp: one (n, c) plane per group of E = 64/128/256 threads,
// per-channel quantities hoisted, no per-element division, no float atomics.  f16 / bf16 tensors are computed
// in fp32 and rounded once on store; only the quantize position arithmetic is done in the storage type,
// because it decides WHICH element is gathered (the reference instantiates the whole kernel at c10::Half).
#pragma once
#include "rk_common.hpp"

namespace rk {
namespace g2d {

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

// the type the quantize position arithmetic runs in: the tensor's own (the reference), unless the shift table is wider
template <typename T, typename S> struct PosType { using type = T; };
template <> struct PosType<__half, float> { using type = float; };
template <> struct PosType<__hip_bfloat16, float> { using type = float; };

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
template <typename T, typename S, bool QUANT>
__global__ __launch_bounds__(kBlock) void k2d_forward(const T* __restrict__ x, const S* __restrict__ shift,
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
            const int th = QuantPos<typename PosType<T, S>::type>::nearest(bH, offH), tw = QuantPos<typename PosType<T, S>::type>::nearest(bW, offW);
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
// this thread's share of the d(shift) terms of one (n, c) plane (E cooperating threads)
// act: what an element of x stands for (identity; the training fusion passes relu(bn2(.)), rk2d_column.hpp)
struct SameValue { template <typename V> __device__ __forceinline__ V operator()(V v) const { return v; } };
template <typename T, typename S, typename Act = SameValue>
__device__ __forceinline__ void shift_grad_plane2(const T* __restrict__ gy, const T* __restrict__ x,
                                                  const S* __restrict__ shift, const Dims2& d, int n, int c, int e,
                                                  int E, typename Compute<T>::type& aH,
                                                  typename Compute<T>::type& aW, Act act = Act()) {
    using CT = typename Compute<T>::type;
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
    const int dh = E / d.Wo, dw = E - dh * d.Wo;
    auto at = [&](int h, int w) -> CT {
        return (h >= 0 && h < d.H && w >= 0 && w < d.W) ? act(ld(xp + h * d.W + w)) : (CT)0;
    };
    for (int i = e; i < HWo; i += E) {
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

// partials part[c][2][P], P = N, p = n
template <typename T, typename S>
__global__ __launch_bounds__(kBlock) void k2d_backward_shift(const T* __restrict__ gy, const T* __restrict__ x,
                                                             const S* __restrict__ shift,
                                                             typename Compute<T>::type* __restrict__ part, Dims2 d) {
    using CT = typename Compute<T>::type;
    __shared__ CT red[2][kBlock / kWave];
    int n, c, e;
    const bool valid = my_plane2(d, n, c, e);
    CT aH = 0, aW = 0;
    if (valid) shift_grad_plane2<T>(gy, x, shift, d, n, c, e, d.E, aH, aW);
    aH = group_sum(aH, d.E, red[0]);
    aW = group_sum(aW, d.E, red[1]);
    if (valid && e == 0) {
        CT* o = part + (size_t)c * 2 * d.N + n;
        o[0] = aH;
        o[d.N] = aW;
    }
}

// row-sum (rubiks.cpp:140-143) + K9 (rubiks2d_kernels.cu:381-397), one workgroup per channel
template <typename T, typename S>
__global__ __launch_bounds__(kBlock) void k2d_finalize(const typename Compute<T>::type* __restrict__ part,
                                                       S* __restrict__ gshift, int C, int P, int normalize) {
    using CT = typename Compute<T>::type;
    __shared__ double red[2][kBlock / kWave];
    const int c = blockIdx.x;
    const CT* p = part + (size_t)c * 2 * P;
    double s[2] = {0, 0};
    for (int k = 0; k < 2; ++k)
        for (int i = threadIdx.x; i < P; i += blockDim.x) s[k] += (double)p[(size_t)k * P + i];
    for (int k = 0; k < 2; ++k) s[k] = group_sum(s[k], (int)blockDim.x, red[k]);
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
// one (n, c) plane of d(x), computed by the E threads that call it
template <typename T, bool QUANT, typename S>
__device__ __forceinline__ void backward_input_plane2(const T* __restrict__ gy, const S* __restrict__ shift,
                                                      T* __restrict__ gx, const Dims2& d, int n, int c, int e, int E) {
    using CT = typename Compute<T>::type;
    const CT nH = -ld(shift + c), nW = -ld(shift + d.C + c);
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
    const T* gp = gy + ((size_t)n * d.C + c) * HWo;
    T* xp = gx + ((size_t)n * d.C + c) * HW;
    int h = e / d.W, w = e - h * d.W;
    const int dh = E / d.W, dw = E - dh * d.W;
    const int flH = floor_fast(nH), flW = floor_fast(nW);
    const CT rH = nH - (CT)flH, rW = nW - (CT)flW;
    const bool zero = (nW == 0 && nH == 0);                              // rubiks2d_kernels.cu:322
    auto gat = [&](int ph, int pw) -> CT {
        const int a = unmap2(ph, d.sH, d.Ho), b = unmap2(pw, d.sW, d.Wo);
        return (a >= 0 && b >= 0) ? ld(gp + a * d.Wo + b) : (CT)0;
    };
    for (int i = e; i < HW; i += E) {
        const int oH = h + d.pH, oW = w + d.pW;
        if (QUANT) {   // skipped positions leave gx untouched (rubiks2d_kernels.cu:294-309)
            const int a = unmap2(QuantPos<typename PosType<T, S>::type>::nearest(oH, nH), d.sH, d.Ho);
            const int b = unmap2(QuantPos<typename PosType<T, S>::type>::nearest(oW, nW), d.sW, d.Wo);
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

template <typename T, typename S, bool QUANT>
__global__ __launch_bounds__(kBlock) void k2d_backward_input(const T* __restrict__ gy, const S* __restrict__ shift,
                                                             T* __restrict__ gx, Dims2 d) {
    int n, c, e;
    if (!my_plane2(d, n, c, e)) return;
    backward_input_plane2<T, QUANT>(gy, shift, gx, d, n, c, e, d.E);
}

}  // namespace g2d
}  // namespace rk
