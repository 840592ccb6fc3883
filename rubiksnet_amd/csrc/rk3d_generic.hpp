// rk3d_generic.hpp -- RubiksShift3D "any configuration" kernels for gfx950.
//
// These cover every case the reference's K1-K5 cover (cuda_src/rubiks3d_kernels.cu:15-960):
// arbitrary stride / padding, quantize, exactly-integer shifts (the d(shift) lowering
// quirk), fp32 and fp64.  The LDS-fed streaming kernels (rk3d_plane / rk3d_dma / rk3d_tile / rk3d_stride2 .hpp) take over
// for the shapes that dominate the networks; these stay as the complete fallback and as
// the per-channel slow path for integer shifts.
//
// Mapping (differs from the reference's flat grid-stride loop with 8 div/mod per element):
// one output plane (n, to, c) is owned by a group of E = 64/128/256 consecutive threads of a
// 256-thread workgroup (small 7x7 / 14x14 planes pack 4 / 1 per workgroup), so the channel's
// shift, its floor/remainder and the two source-plane pointers are computed once per thread
// and (ho, wo) advance incrementally -- no per-element division.  Consecutive lanes read
// consecutive addresses of one source plane (coalesced modulo the per-channel offset).
// d(shift) is reduced wave-shuffle -> LDS -> one partial per plane (no atomics), then
// k3d_finalize sums the partials in a fixed order in fp64 and applies K5.
#pragma once
#include "rk_common.hpp"

namespace rk {

struct Dims3 {
    int N, T, C, H, W;      // input
    int To, Ho, Wo;         // output
    int sT, sH, sW, pT, pH, pW;
    int E, logE;            // threads per plane (power of two, 64..256) and its log2
};

// floor/remainder of one shift component exactly as rubiks3d_kernels.cu:65-74 does it:
// floorf() in fp32 whatever T is, remainder in T.
template <typename T> struct Frac { int fl; T r; };
template <typename T> __device__ __forceinline__ Frac<T> split_shift(T s) {
    Frac<T> f;
    f.fl = (int)floorf((float)s);
    f.r = s - (T)f.fl;
    return f;
}

// rubiks3d_kernels.cu:193-203 (same expression tree => same rounding with contraction off)
template <typename T>
__device__ __forceinline__ T trilerp(T q000, T q001, T q010, T q011, T q100, T q101, T q110, T q111,
                                     T rT, T rH, T rW) {
    return (1 - rT) * ((1 - rH) * (q000 * (1 - rW) + q001 * rW) + rH * (q010 * (1 - rW) + q011 * rW)) +
           rT * ((1 - rH) * (q100 * (1 - rW) + q101 * rW) + rH * (q110 * (1 - rW) + q111 * rW));
}

// rubiks3d_kernels.cu:208-215
template <typename T> __device__ __forceinline__ T interp2(T p11, T p12, T p21, T p22, T d1, T d2) {
    return p11 * (1 - d1) * (1 - d2) + p12 * (1 - d1) * d2 + p21 * d1 * (1 - d2) + p22 * d1 * d2;
}

struct PlaneId { int n, t, c; bool valid; };

// Decompose this thread's plane (group of E threads) and element lane.
__device__ __forceinline__ PlaneId my_plane(const Dims3& d, int planes_t /* To or T */, int& lane_e) {
    const int sub = threadIdx.x >> d.logE;
    lane_e = threadIdx.x & (d.E - 1);
    const long long plane = (long long)blockIdx.x * (kBlock >> d.logE) + sub;
    PlaneId p;
    p.valid = plane < (long long)d.N * planes_t * d.C;
    const long long q = p.valid ? plane : 0;
    p.c = (int)(q % d.C);
    const long long nt = q / d.C;
    p.t = (int)(nt % planes_t);
    p.n = (int)(nt / planes_t);
    return p;
}

// ------------------------------------------------------------------------------ K1
template <typename T, bool QUANT>
__global__ __launch_bounds__(kBlock) void k3d_forward_generic(const T* __restrict__ x,
                                                              const T* __restrict__ shift,
                                                              T* __restrict__ y, Dims3 d) {
    int e;
    const PlaneId pl = my_plane(d, d.To, e);
    if (!pl.valid) return;
    const Frac<T> fT = split_shift(shift[pl.c]);
    const Frac<T> fH = split_shift(shift[d.C + pl.c]);
    const Frac<T> fW = split_shift(shift[2 * d.C + pl.c]);
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
    const size_t tstride = (size_t)d.C * HW;
    const T* xc = x + ((size_t)pl.n * d.T * d.C + pl.c) * HW;      // (n, t=0, c)
    T* yp = y + (((size_t)pl.n * d.To + pl.t) * d.C + pl.c) * HWo;
    const int bT = pl.t * d.sT - d.pT;

    int ho = e / d.Wo, wo = e - ho * d.Wo;
    const int dh = d.E / d.Wo, dw = d.E - dh * d.Wo;

    if (QUANT) {  // rubiks3d_kernels.cu:76-93 -- a pure gather
        const int tt = bT + ((fT.r < 0.5f) ? fT.fl : fT.fl + 1);
        const int qH = (fH.r < 0.5f) ? fH.fl : fH.fl + 1;
        const int qW = (fW.r < 0.5f) ? fW.fl : fW.fl + 1;
        const bool vt = tt >= 0 && tt < d.T;
        const T* p = xc + (vt ? (size_t)tt * tstride : 0);
        for (int i = e; i < HWo; i += d.E) {
            const int h = ho * d.sH - d.pH + qH, w = wo * d.sW - d.pW + qW;
            T v = 0;
            if (vt && h >= 0 && h < d.H && w >= 0 && w < d.W) v = p[h * d.W + w];
            yp[i] = v;
            wo += dw; ho += dh;
            if (wo >= d.Wo) { wo -= d.Wo; ++ho; }
        }
        return;
    }

    const int t0 = bT + fT.fl;
    const bool v0 = t0 >= 0 && t0 < d.T, v1 = t0 + 1 >= 0 && t0 + 1 < d.T;
    const T* p0 = xc + (v0 ? (size_t)t0 * tstride : 0);
    const T* p1 = xc + (v1 ? (size_t)(t0 + 1) * tstride : 0);
    for (int i = e; i < HWo; i += d.E) {
        const int h0 = ho * d.sH - d.pH + fH.fl, w0 = wo * d.sW - d.pW + fW.fl;
        const bool mh0 = h0 >= 0 && h0 < d.H, mh1 = h0 + 1 >= 0 && h0 + 1 < d.H;
        const bool mw0 = w0 >= 0 && w0 < d.W, mw1 = w0 + 1 >= 0 && w0 + 1 < d.W;
        const int o00 = h0 * d.W + w0;
        T q000 = 0, q001 = 0, q010 = 0, q011 = 0, q100 = 0, q101 = 0, q110 = 0, q111 = 0;
        if (v0) {
            if (mh0 && mw0) q000 = p0[o00];
            if (mh0 && mw1) q001 = p0[o00 + 1];
            if (mh1 && mw0) q010 = p0[o00 + d.W];
            if (mh1 && mw1) q011 = p0[o00 + d.W + 1];
        }
        if (v1) {
            if (mh0 && mw0) q100 = p1[o00];
            if (mh0 && mw1) q101 = p1[o00 + 1];
            if (mh1 && mw0) q110 = p1[o00 + d.W];
            if (mh1 && mw1) q111 = p1[o00 + d.W + 1];
        }
        yp[i] = trilerp(q000, q001, q010, q011, q100, q101, q110, q111, fT.r, fH.r, fW.r);
        wo += dw; ho += dh;
        if (wo >= d.Wo) { wo -= d.Wo; ++ho; }
    }
}

// ------------------------------------------------------------------------- K3 / K4
// d(x): gather of gy at the NEGATED shift through the stride/pad un-mapping
// (rubiks3d_kernels.cu:455-723; :726-929 is the same with stride 1 / pad 0).
__device__ __forceinline__ int unmap(int p, int s, int lim) {
    // rubiks3d_kernels.cu:586-589: C remainder (negative non-multiples are != 0), then bounds
    if (p % s != 0) return -1;
    const int q = p / s;
    return (q >= 0 && q < lim) ? q : -1;
}

// One input plane (n, t, c) of d(x), computed by the E threads e = 0..E-1 that call it.
template <typename T, bool QUANT>
__device__ __forceinline__ void backward_input_plane(const T* __restrict__ shift, const T* __restrict__ gy,
                                                     T* __restrict__ gx, const Dims3& d, int n, int t, int c,
                                                     int e, int E, int lo = 0, int hi = 0x7fffffff) {
    const T nT = -shift[c], nH = -shift[d.C + c], nW = -shift[2 * d.C + c];
    const Frac<T> fT = split_shift(nT), fH = split_shift(nH), fW = split_shift(nW);
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
    const size_t tstride = (size_t)d.C * HWo;
    const T* gc = gy + ((size_t)n * d.To * d.C + c) * HWo;   // (n, to=0, c)
    T* gp = gx + (((size_t)n * d.T + t) * d.C + c) * HW;
    const int oT = t + d.pT;

    // elements lo + e, lo + e + E, ... below min(hi, HW): the whole plane by default, a sub-range for rk3d_slab.hpp
    e += lo;
    const int HWe = hi < HW ? hi : HW;
    int h = e / d.W, w = e - h * d.W;
    const int dh = E / d.W, dw = E - dh * d.W;

    // 0 = single tap at the nearest position (quantize), 1 = all shifts exactly zero
    // (rubiks3d_kernels.cu:561-576), 2 = trilinear
    const int mode = QUANT ? 0 : ((nT == 0 && nH == 0 && nW == 0) ? 1 : 2);
    if (mode != 2) {
        const int aT = QUANT ? ((fT.r < 0.5f) ? fT.fl : fT.fl + 1) : 0;
        const int aH = QUANT ? ((fH.r < 0.5f) ? fH.fl : fH.fl + 1) : 0;
        const int aW = QUANT ? ((fW.r < 0.5f) ? fW.fl : fW.fl + 1) : 0;
        const int tt = unmap(oT + aT, d.sT, d.To);
        const T* p = gc + (tt >= 0 ? (size_t)tt * tstride : 0);
        for (int i = e; i < HWe; i += E) {
            const int hh = unmap(h + d.pH + aH, d.sH, d.Ho), ww = unmap(w + d.pW + aW, d.sW, d.Wo);
            T v = 0;
            if (tt >= 0 && hh >= 0 && ww >= 0) v = p[hh * d.Wo + ww];
            gp[i] = v;
            w += dw; h += dh;
            if (w >= d.W) { w -= d.W; ++h; }
        }
        return;
    }

    const int t0 = unmap(oT + fT.fl, d.sT, d.To), t1 = unmap(oT + fT.fl + 1, d.sT, d.To);
    const T* p0 = gc + (t0 >= 0 ? (size_t)t0 * tstride : 0);
    const T* p1 = gc + (t1 >= 0 ? (size_t)t1 * tstride : 0);
    for (int i = e; i < HWe; i += E) {
        const int h0 = unmap(h + d.pH + fH.fl, d.sH, d.Ho), h1 = unmap(h + d.pH + fH.fl + 1, d.sH, d.Ho);
        const int w0 = unmap(w + d.pW + fW.fl, d.sW, d.Wo), w1 = unmap(w + d.pW + fW.fl + 1, d.sW, d.Wo);
        T q000 = 0, q001 = 0, q010 = 0, q011 = 0, q100 = 0, q101 = 0, q110 = 0, q111 = 0;
        if (t0 >= 0) {
            if (h0 >= 0 && w0 >= 0) q000 = p0[h0 * d.Wo + w0];
            if (h0 >= 0 && w1 >= 0) q001 = p0[h0 * d.Wo + w1];
            if (h1 >= 0 && w0 >= 0) q010 = p0[h1 * d.Wo + w0];
            if (h1 >= 0 && w1 >= 0) q011 = p0[h1 * d.Wo + w1];
        }
        if (t1 >= 0) {
            if (h0 >= 0 && w0 >= 0) q100 = p1[h0 * d.Wo + w0];
            if (h0 >= 0 && w1 >= 0) q101 = p1[h0 * d.Wo + w1];
            if (h1 >= 0 && w0 >= 0) q110 = p1[h1 * d.Wo + w0];
            if (h1 >= 0 && w1 >= 0) q111 = p1[h1 * d.Wo + w1];
        }
        gp[i] = trilerp(q000, q001, q010, q011, q100, q101, q110, q111, fT.r, fH.r, fW.r);
        w += dw; h += dh;
        if (w >= d.W) { w -= d.W; ++h; }
    }
}

template <typename T, bool QUANT>
__global__ __launch_bounds__(kBlock) void k3d_backward_input_generic(const T* __restrict__ shift,
                                                                     const T* __restrict__ gy,
                                                                     T* __restrict__ gx, Dims3 d) {
    int e;
    const PlaneId pl = my_plane(d, d.T, e);   // planes of the INPUT
    if (!pl.valid) return;
    backward_input_plane<T, QUANT>(shift, gy, gx, d, pl.n, pl.t, pl.c, e, d.E);
}

// ------------------------------------------------------------------------------ K2
// d(shift) partials.  For every output element the reference forms, per dimension, the
// difference between the bilinear interpolation of the "large" face and of the "small"
// face of the 2x2x2 tap cube (rubiks3d_kernels.cu:432-446).  In a dimension whose remainder
// is EXACTLY zero the small index is lowered by one (:290-298) and that lowered index is
// used by every tap on that face (:359-431) -- reproduced through `lo`.
//
// Partials layout: part[c][3][P], P = N*To, p = n*To + to.
// This thread's share of the d(shift) terms of one output plane (n, to, c) (E cooperating threads).
// what a tap of x means: the value itself, or -- training fusion, x = z -- relu(bn(z)) = max(a z + b, 0)
struct NoAct { template <typename T> __device__ __forceinline__ T operator()(T v) const { return v; } };
struct BnAct {
    float a, b;
    __device__ __forceinline__ float operator()(float v) const { return fmaxf(fmaf(a, v, b), 0.f); }
};
template <typename T, typename Act = NoAct>
__device__ __forceinline__ void shift_grad_plane(const T* __restrict__ x, const T* __restrict__ shift,
                                                 const T* __restrict__ gy, const Dims3& d, int n, int to, int c,
                                                 int e, int E, T& aT, T& aH, T& aW, const Act act = Act(),
                                                 int lo = 0, int hi = 0x7fffffff) {
    const Frac<T> fT = split_shift(shift[c]);
    const Frac<T> fH = split_shift(shift[d.C + c]);
    const Frac<T> fW = split_shift(shift[2 * d.C + c]);
    const int zT = (fT.r == 0) ? 1 : 0, zH = (fH.r == 0) ? 1 : 0, zW = (fW.r == 0) ? 1 : 0;
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
    const size_t tstride = (size_t)d.C * HW;
    const T* xc = x + ((size_t)n * d.T * d.C + c) * HW;
    const T* gp = gy + (((size_t)n * d.To + to) * d.C + c) * HWo;
    const int bT = to * d.sT - d.pT;
    const int t0 = bT + fT.fl - zT, t1 = bT + fT.fl + 1;
    const bool v0 = t0 >= 0 && t0 < d.T, v1 = t1 >= 0 && t1 < d.T;
    const T* p0 = xc + (v0 ? (size_t)t0 * tstride : 0);
    const T* p1 = xc + (v1 ? (size_t)t1 * tstride : 0);

    e += lo;                                              // output elements lo + e, lo + e + E, ... below min(hi, HWo)
    const int HWe = hi < HWo ? hi : HWo;
    int ho = e / d.Wo, wo = e - ho * d.Wo;
    const int dh = E / d.Wo, dw = E - dh * d.Wo;
    for (int i = e; i < HWe; i += E) {
        const int hb = ho * d.sH - d.pH + fH.fl, wb = wo * d.sW - d.pW + fW.fl;
        const int h0 = hb - zH, h1 = hb + 1, w0 = wb - zW, w1 = wb + 1;
        const bool mh0 = h0 >= 0 && h0 < d.H, mh1 = h1 >= 0 && h1 < d.H;
        const bool mw0 = w0 >= 0 && w0 < d.W, mw1 = w1 >= 0 && w1 < d.W;
        T q000 = 0, q001 = 0, q010 = 0, q011 = 0, q100 = 0, q101 = 0, q110 = 0, q111 = 0;
        if (v0) {
            if (mh0 && mw0) q000 = act(p0[h0 * d.W + w0]);
            if (mh0 && mw1) q001 = act(p0[h0 * d.W + w1]);
            if (mh1 && mw0) q010 = act(p0[h1 * d.W + w0]);
            if (mh1 && mw1) q011 = act(p0[h1 * d.W + w1]);
        }
        if (v1) {
            if (mh0 && mw0) q100 = act(p1[h0 * d.W + w0]);
            if (mh0 && mw1) q101 = act(p1[h0 * d.W + w1]);
            if (mh1 && mw0) q110 = act(p1[h1 * d.W + w0]);
            if (mh1 && mw1) q111 = act(p1[h1 * d.W + w1]);
        }
        const T Ts = interp2(q000, q001, q010, q011, fH.r, fW.r);
        const T Tl = interp2(q100, q101, q110, q111, fH.r, fW.r);
        const T Hs = interp2(q000, q001, q100, q101, fT.r, fW.r);
        const T Hl = interp2(q010, q011, q110, q111, fT.r, fW.r);
        const T Ws = interp2(q000, q010, q100, q110, fT.r, fH.r);
        const T Wl = interp2(q001, q011, q101, q111, fT.r, fH.r);
        const T up = gp[i];
        aT += (-Ts + Tl) * up;
        aH += (-Hs + Hl) * up;
        aW += (-Ws + Wl) * up;
        wo += dw; ho += dh;
        if (wo >= d.Wo) { wo -= d.Wo; ++ho; }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k3d_backward_shift_generic(const T* __restrict__ x,
                                                                     const T* __restrict__ shift,
                                                                     const T* __restrict__ gy,
                                                                     T* __restrict__ part, Dims3 d) {
    __shared__ T red[3][kBlock / kWave];
    int e;
    const PlaneId pl = my_plane(d, d.To, e);
    T aT = 0, aH = 0, aW = 0;
    if (pl.valid) shift_grad_plane<T>(x, shift, gy, d, pl.n, pl.t, pl.c, e, d.E, aT, aH, aW);
    aT = group_sum(aT, d.E, red[0]);
    aH = group_sum(aH, d.E, red[1]);
    aW = group_sum(aW, d.E, red[2]);
    if (pl.valid && e == 0) {
        const int P = d.N * d.To;
        T* o = part + (size_t)pl.c * 3 * P + (size_t)pl.n * d.To + pl.t;
        o[0] = aT;
        o[P] = aH;
        o[2 * P] = aW;
    }
}

// ------------------------------------------------------------- row-sum + K5 (fused)
// One workgroup per channel (finalize_block(P) threads: a single wave -- no barriers -- when P <= 64):
// fixed-order fp64 sum of the P partials of each component
// (replaces the addmv_ row-sum, rubiks.cpp:344-345), then rubiks3d_kernels.cu:932-960.
template <typename T>
__global__ __launch_bounds__(kBlock) void k3d_finalize(const T* __restrict__ part, T* __restrict__ gshift,
                                                       int C, int P, int normalize, T t_factor) {
    __shared__ double red[3][kBlock / kWave];
    const int c = blockIdx.x;
    const T* p = part + (size_t)c * 3 * P;
    double s[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
        for (int i = threadIdx.x; i < P; i += blockDim.x) s[k] += (double)p[(size_t)k * P + i];
    for (int k = 0; k < 3; ++k) s[k] = group_sum(s[k], (int)blockDim.x, red[k]);
    if (threadIdx.x == 0) {
        T gT = (T)s[0], gH = (T)s[1], gW = (T)s[2];
        if (normalize) {
            T a, b, w;
            if (t_factor < 0) { a = gT; b = 0; w = 0; }
            else { a = gT * t_factor; b = gH; w = gW; }
            const T mag = sqrt(a * a + b * b + w * w);
            if (mag > 0) { gT = a / mag; gH = b / mag; gW = w / mag; }
        }
        gshift[c] = gT;
        gshift[C + c] = gH;
        gshift[2 * C + c] = gW;
    }
}

}  // namespace rk
