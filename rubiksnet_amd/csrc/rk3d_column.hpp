// rk3d_column.hpp -- RubiksShift3D "column" kernels: any spatial stride / padding, any H x W,
// fp32 and fp64, temporal stride 1 / pad 0, quantize off.
//
// These take every layer of the networks that the LDS-DMA kernels (rk3d_dma.hpp) do not:
// the four stride-(1,2,2) down-sampling layers of each net, and the 14x14 / 7x7 planes
// (W % 4 != 0), where the per-plane generic kernels spend their time on set-up (one 196-element
// plane per workgroup: 6-15 % of the HBM roofline at [32,8,288,14,14]).
//
// Same T-walk as the streaming kernels -- a group of E threads owns one (n, c) column (or one
// 1024-element chunk of it) and walks t, so the channel's shift, the per-element tap offsets and
// validity masks are computed once, and the H/W-interpolated field of a source plane is reused by
// the two outputs it feeds (registers).  The taps are per-element global loads served by L1/L2
// (consecutive lanes -> consecutive addresses of one plane), so no LDS tile, no alignment or
// width restrictions.
//   forward : y[to] = (1-rT) B(to+flT) + rT B(to+flT+1),  B = bilinear of the 4 taps of a plane
//             -- the reference's expression tree (rubiks3d_kernels.cu:193-203), bit-identical.
//   backward: adjoint form on the INPUT side with the negated shift (fl', r'), tap (j,k) of an
//             input element exists iff (h+pH+fl'H+j) % sH == 0 etc. (rubiks3d_kernels.cu:586-589):
//             gx[t] = (1-r'T) Q(t+fl'T) + r'T Q(t+fl'T+1) (tree of :709-719, bit-identical), and
//             gT = sum x (Q(t0) - Q(t0+1)),  gH = sum x ((1-r'T) QH(t0) + r'T QH(t0+1)),  gW alike,
//             with QH / QW the row / column differences of the same taps.  One partial per
//             (n, c, chunk): part[c][3][P], P = N * nchunks; k3d_finalize sums them (no atomics).
// Channels with an exactly-integer shift component take the per-element reference formulation
// (shared with the generic kernels) for the whole column.
#pragma once
#include <type_traits>

#include "rk3d_generic.hpp"
#include "rk3d_dma.hpp"

namespace rk {
namespace col3d {

// elements per thread M: 1 for planes that fit one group of threads (14x14, 7x7), else 4 (chunk = E * M elements)

struct CDims {
    Dims3 d;
    int E, logE;          // threads per column group (64/128/256)
    int nchunks;          // chunks per plane
    int M;                // elements per thread
};

struct ColId { int n, c, chunk; bool valid; };
constexpr int kDeep = 9;     // steps of a walk held in registers at once: T <= 8

__device__ __forceinline__ ColId my_column(const CDims& cd, int& e) {
    const int sub = threadIdx.x >> cd.logE;
    e = threadIdx.x & (cd.E - 1);
    const long long id = (long long)blockIdx.x * (kBlock >> cd.logE) + sub;     // (n, c, chunk) flattened
    ColId r;
    r.valid = id < (long long)cd.d.N * cd.d.C * cd.nchunks;
    const long long q = r.valid ? id : 0;
    r.chunk = (int)(q % cd.nchunks);
    const long long col = q / cd.nchunks;
    r.c = (int)(col % cd.d.C);
    r.n = (int)(col / cd.d.C);
    return r;
}

// ------------------------------------------------------------------------------------ forward
template <typename T, int kM>
__global__ __launch_bounds__(kBlock) void k3d_forward_column(const T* __restrict__ x, const T* __restrict__ shift,
                                                             T* __restrict__ y, CDims cd) {
    const Dims3& d = cd.d;
    int e;
    const ColId id = my_column(cd, e);
    if (!id.valid) return;
    const Frac<T> fT = split_shift(shift[id.c]);
    const Frac<T> fH = split_shift(shift[d.C + id.c]);
    const Frac<T> fW = split_shift(shift[2 * d.C + id.c]);
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
    const size_t tsi = (size_t)d.C * HW, tso = (size_t)d.C * HWo;
    const T* xc = x + ((size_t)id.n * d.T * d.C + id.c) * HW;        // (n, t = 0, c)
    T* yc = y + ((size_t)id.n * d.To * d.C + id.c) * HWo;

    // per element: flat offset of tap (0,0) in a source plane and validity of the 4 taps
    int o00[kM], oidx[kM];
    unsigned mask[kM];
#pragma unroll
    for (int m = 0; m < kM; ++m) {
        const int i = id.chunk * cd.E * kM + m * cd.E + e;
        oidx[m] = i < HWo ? i : -1;
        const int ii = i < HWo ? i : 0;
        const int ho = ii / d.Wo, wo = ii - ho * d.Wo;
        const int h0 = ho * d.sH - d.pH + fH.fl, w0 = wo * d.sW - d.pW + fW.fl;
        const bool mh0 = h0 >= 0 && h0 < d.H, mh1 = h0 + 1 >= 0 && h0 + 1 < d.H;
        const bool mw0 = w0 >= 0 && w0 < d.W, mw1 = w0 + 1 >= 0 && w0 + 1 < d.W;
        o00[m] = h0 * d.W + w0;
        mask[m] = (i < HWo) ? ((mh0 && mw0 ? 1u : 0u) | (mh0 && mw1 ? 2u : 0u) | (mh1 && mw0 ? 4u : 0u) |
                               (mh1 && mw1 ? 8u : 0u)) : 0u;
    }
    const T rT = fT.r, rH = fH.r, rW = fW.r;
    T Bprev[kM];
#pragma unroll
    for (int m = 0; m < kM; ++m) Bprev[m] = 0;

    const int t_first = fT.fl, t_last = d.T + fT.fl;                 // source plane index; T+1 steps
    for (int t = t_first; t <= t_last; ++t) {
        const bool valid = t >= 0 && t < d.T;
        const T* p = xc + (valid ? (size_t)t * tsi : 0);
        const int to = t - fT.fl - 1;
        const bool emit = to >= 0 && to < d.To;
        T* out = yc + (emit ? (size_t)to * tso : 0);
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            T q00 = 0, q01 = 0, q10 = 0, q11 = 0;
            if (valid) {
                const unsigned mk = mask[m];
                if (mk & 1u) q00 = p[o00[m]];
                if (mk & 2u) q01 = p[o00[m] + 1];
                if (mk & 4u) q10 = p[o00[m] + d.W];
                if (mk & 8u) q11 = p[o00[m] + d.W + 1];
            }
            const T B = (1 - rH) * (q00 * (1 - rW) + q01 * rW) + rH * (q10 * (1 - rW) + q11 * rW);
            if (emit && oidx[m] >= 0) out[oidx[m]] = (1 - rT) * Bprev[m] + rT * B;
            Bprev[m] = B;
        }
    }
}

// ----------------------------------------------------------------------------------- backward
// SINGLE: both spatial strides >= 2.  Then an input element has at most ONE gy tap -- (h+pH+fl'H+j) % sH == 0 holds
// for one j of {0, 1} at most, likewise k -- and the reference's tree collapses, exactly (the other three taps
// are zeros: 0*w and x+0 are exact), to Q = wj (v wk), QH = +-(v wk), QW = +-(wj v): one load per element and
// plane instead of four predicated ones (the four stride-(1,2,2) layers of each network run here).
// VEC (kM == 4, fp32, input planes of a multiple of 4 elements, 16-byte aligned tensors): a thread owns 4 CONSECUTIVE
// input elements, so the two big streams of a strided layer's backward -- x and gx, 4x the size of gy -- move as
// 16-byte accesses instead of 4-byte ones; the gy taps stay scalar (L1-served; neighbouring elements share them).
// Same arithmetic per element, bit-identical: [32,8,54,112,112] stride (1,2,2) 585 -> 373 us, [32,8,108,56,56]
// 365 -> 240 us.  (The forward is the other way round -- its big stream is the GATHERED one, and consecutive
// outputs per thread spread a wave's tap loads over 8x the cache lines: 213 -> 330 us; it stays element-strided.)
// FUSED (fp32): the row-sum over the partials and K5 run inside this launch -- the partials are published as granule
// pairs (rk_dma.hpp, fin_publish) and the C extra workgroups at the end of the grid finish a channel each -- so the 7x7
// planes and the 28 -> 14 / 14 -> 7 strided layers no longer pay a second launch (4.3 us + a dependent-launch boundary
// on a 28 us kernel at [32,8,576,7,7]).
template <typename T, bool WRITE_GX, int kM, bool SINGLE, bool VEC = false, bool FUSED = false>
__global__ __launch_bounds__(kBlock) void k3d_backward_column(const T* __restrict__ x, const T* __restrict__ shift,
                                                              const T* __restrict__ gy, T* __restrict__ gx,
                                                              T* __restrict__ part, CDims cd, dma3d::Fin3 fin = dma3d::Fin3{}) {
    if constexpr (FUSED) {
        if ((int)blockIdx.x >= fin.f.producers) {
            if (threadIdx.x < kWave) dma3d::finalizer_wave<3>(fin, (int)blockIdx.x - fin.f.producers, cd.d.C, cd.d.N * cd.nchunks);
            return;
        }
    }
    __shared__ T red[3][kBlock / kWave];
    const Dims3& d = cd.d;
    int e;
    const ColId id = my_column(cd, e);
    T accT = 0, accH = 0, accW = 0;
    if (id.valid) {
        const T s0 = shift[id.c], s1 = shift[d.C + id.c], s2 = shift[2 * d.C + id.c];
        if (split_shift(s0).r == 0 || split_shift(s1).r == 0 || split_shift(s2).r == 0) {
            // integer component: the reference's per-element formulation; chunk 0 does the whole column
            if (id.chunk == 0) {
                if (WRITE_GX)
                    for (int t = 0; t < d.T; ++t) backward_input_plane<T, false>(shift, gy, gx, d, id.n, t, id.c, e, cd.E);
                for (int to = 0; to < d.To; ++to)
                    shift_grad_plane<T>(x, shift, gy, d, id.n, to, id.c, e, cd.E, accT, accH, accW);
            }
        } else {
            const Frac<T> fT = split_shift(-s0), fH = split_shift(-s1), fW = split_shift(-s2);   // fl', r'
            const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
            const size_t tsi = (size_t)d.C * HW, tso = (size_t)d.C * HWo;
            const T* xc = x + ((size_t)id.n * d.T * d.C + id.c) * HW;
            const T* gc = gy + ((size_t)id.n * d.To * d.C + id.c) * HWo;
            T* oc = WRITE_GX ? gx + ((size_t)id.n * d.T * d.C + id.c) * HW : nullptr;

            const T rT = fT.r, rH = fH.r, rW = fW.r;
            T xa[kM], xb[kM], Qprev[kM];
            T sT = 0, sH = 0, sW = 0;
            bool t_done = false;
            int iidx[kM];
            // per INPUT element: gy offsets of its taps (row-major in the output plane), -1 = no such tap
            int tap[kM][SINGLE ? 1 : 4];
            T wj[kM], wk[kM], sj[kM], sk[kM];                         // SINGLE: the tap's H / W weights and signs
#pragma unroll
            for (int m = 0; m < kM; ++m) {
                const int i = VEC ? id.chunk * cd.E * kM + e * kM + m : id.chunk * cd.E * kM + m * cd.E + e;
                iidx[m] = i < HW ? i : -1;
                const int ii = i < HW ? i : 0;
                const int h = ii / d.W, w = ii - h * d.W;
                const int r0 = unmap(h + d.pH + fH.fl, d.sH, d.Ho), r1 = unmap(h + d.pH + fH.fl + 1, d.sH, d.Ho);
                const int c0 = unmap(w + d.pW + fW.fl, d.sW, d.Wo), c1 = unmap(w + d.pW + fW.fl + 1, d.sW, d.Wo);
                const bool live = i < HW;
                if (SINGLE) {
                    const int r = r0 >= 0 ? r0 : r1, c = c0 >= 0 ? c0 : c1;       // at most one of each exists
                    tap[m][0] = (live && r >= 0 && c >= 0) ? r * d.Wo + c : -1;
                    wj[m] = r0 >= 0 ? 1 - rH : rH;  sj[m] = r0 >= 0 ? (T)1 : (T)-1;
                    wk[m] = c0 >= 0 ? 1 - rW : rW;  sk[m] = c0 >= 0 ? (T)1 : (T)-1;
                } else {
                    tap[m][0] = (live && r0 >= 0 && c0 >= 0) ? r0 * d.Wo + c0 : -1;
                    tap[m][SINGLE ? 0 : 1] = (live && r0 >= 0 && c1 >= 0) ? r0 * d.Wo + c1 : -1;
                    tap[m][SINGLE ? 0 : 2] = (live && r1 >= 0 && c0 >= 0) ? r1 * d.Wo + c0 : -1;
                    tap[m][SINGLE ? 0 : 3] = (live && r1 >= 0 && c1 >= 0) ? r1 * d.Wo + c1 : -1;
                }
            }
            auto load4 = [&](const T* plane, T (&dst)[kM], bool on) {    // my elements of one x plane
                if constexpr (VEC) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (on && iidx[0] >= 0) v = *reinterpret_cast<const float4*>(plane + iidx[0]);
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                } else {
#pragma unroll
                    for (int m = 0; m < kM; ++m) dst[m] = (on && iidx[m] >= 0) ? plane[iidx[m]] : (T)0;
                }
            };
#pragma unroll
            for (int m = 0; m < kM; ++m) { xa[m] = 0; Qprev[m] = 0; }
            load4(xc, xb, true);                                       // x[0]
            // step on gy plane tg; to = tg - fl'T - 1 is the input plane whose gx is completed
            const int t_first = fT.fl, t_last = d.T + fT.fl;
            constexpr int NTAP = SINGLE ? 1 : 4;
            if constexpr (kM == 1) {
                if (d.T < kDeep) {
                    // Small planes (one element per thread: 7x7, 14x14 inputs): every gy tap and every x plane of the walk
                    // is requested before the first one is used (addresses clamped into the column, values masked at
                    // use) -- with a load -> use chain per step the walk paid T + 1 memory latencies
                    // ([32,8,576,14,14] stride (1,2,2): 130 -> 90 us, [32,8,576,7,7]: 33 -> 28 us).  Same arithmetic,
                    // same order.  (The forward and the 4-elements-per-thread backward did not gain: enough waves.)
                    T gq[kDeep][kM][NTAP], xq[kDeep][kM];
#pragma unroll
                    for (int s = 0; s < kDeep; ++s) {
                        const int tg = t_first + s;
                        const bool valid = s <= d.T && tg >= 0 && tg < d.To;
                        const T* p = gc + (valid ? (size_t)tg * tso : 0);
#pragma unroll
                        for (int m = 0; m < kM; ++m)
#pragma unroll
                            for (int j = 0; j < NTAP; ++j) gq[s][m][j] = p[tap[m][j] >= 0 ? tap[m][j] : 0];
                        const bool has_next = s + 1 < d.T;               // x[s + 1] enters the window at step s
                        const T* xn = xc + (has_next ? (size_t)(s + 1) * tsi : 0);
                        if constexpr (VEC) {
                            const float4 v = *reinterpret_cast<const float4*>(xn + (iidx[0] >= 0 ? iidx[0] : 0));
                            xq[s][0] = v.x; xq[s][1] = v.y; xq[s][2] = v.z; xq[s][3] = v.w;
                        } else {
#pragma unroll
                            for (int m = 0; m < kM; ++m) xq[s][m] = xn[iidx[m] >= 0 ? iidx[m] : 0];
                        }
                    }
#pragma unroll
                    for (int s = 0; s < kDeep; ++s) {
                        if (s > d.T) continue;
                        const int tg = t_first + s;
                        const bool valid = tg >= 0 && tg < d.To;
                        const int to = s - 1;
                        const bool emit = WRITE_GX && to >= 0;
                        const bool has_next = s + 1 < d.T;
                        T* out = WRITE_GX ? oc + (emit ? (size_t)to * tsi : 0) : nullptr;
                        T res[kM];
#pragma unroll
                        for (int m = 0; m < kM; ++m) {
                            T Q, QH, QW;
                            if (SINGLE) {
                                const T v = (valid && tap[m][0] >= 0) ? gq[s][m][0] : (T)0;
                                const T vk = v * wk[m];
                                Q = wj[m] * vk;
                                QH = sj[m] * vk;
                                QW = sk[m] * (wj[m] * v);
                            } else {
                                const T q00 = (valid && tap[m][0] >= 0) ? gq[s][m][0] : (T)0;
                                const T q01 = (valid && tap[m][NTAP > 1 ? 1 : 0] >= 0) ? gq[s][m][NTAP > 1 ? 1 : 0] : (T)0;
                                const T q10 = (valid && tap[m][NTAP > 1 ? 2 : 0] >= 0) ? gq[s][m][NTAP > 1 ? 2 : 0] : (T)0;
                                const T q11 = (valid && tap[m][NTAP > 1 ? 3 : 0] >= 0) ? gq[s][m][NTAP > 1 ? 3 : 0] : (T)0;
                                const T la = q00 * (1 - rW) + q01 * rW, lb = q10 * (1 - rW) + q11 * rW;
                                Q = (1 - rH) * la + rH * lb;
                                QH = la - lb;
                                QW = ((1 - rH) * q00 + rH * q10) - ((1 - rH) * q01 + rH * q11);
                            }
                            const T dx = xb[m] - xa[m];
                            const T mx = (1 - rT) * xb[m] + rT * xa[m];
                            sT += Q * dx;
                            sH += QH * mx;
                            sW += QW * mx;
                            if (WRITE_GX) {
                                res[m] = (1 - rT) * Qprev[m] + rT * Q;
                                if (!VEC && emit && iidx[m] >= 0) out[iidx[m]] = res[m];
                                Qprev[m] = Q;
                            }
                            xa[m] = xb[m];
                            xb[m] = (has_next && iidx[m] >= 0) ? xq[s][m] : (T)0;
                        }
                        if constexpr (VEC && WRITE_GX) {
                            if (emit && iidx[0] >= 0)
                                *reinterpret_cast<float4*>(out + iidx[0]) = make_float4(res[0], res[1], res[2], res[3]);
                        }
                    }
                    t_done = true;
                }
            }
            for (int tg = t_first; !t_done && tg <= t_last; ++tg) {
                const bool valid = tg >= 0 && tg < d.To;
                const T* p = gc + (valid ? (size_t)tg * tso : 0);
                const int to = tg - fT.fl - 1;                           // -1 .. T-1
                const bool emit = WRITE_GX && to >= 0;
                T* out = WRITE_GX ? oc + (emit ? (size_t)to * tsi : 0) : nullptr;
                const bool has_next = to + 2 < d.T;
                const T* xn = xc + (has_next ? (size_t)(to + 2) * tsi : 0);
                T xnext[kM], res[kM];
                load4(xn, xnext, has_next);
#pragma unroll
                for (int m = 0; m < kM; ++m) {
                    T Q, QH, QW;
                    if (SINGLE) {
                        T v = 0;
                        if (valid && tap[m][0] >= 0) v = p[tap[m][0]];
                        const T vk = v * wk[m];
                        Q = wj[m] * vk;                                  // = the reference's tree with three zero taps
                        QH = sj[m] * vk;
                        QW = sk[m] * (wj[m] * v);
                    } else {
                        T q00 = 0, q01 = 0, q10 = 0, q11 = 0;
                        if (valid) {
                            if (tap[m][0] >= 0) q00 = p[tap[m][0]];
                            if (tap[m][SINGLE ? 0 : 1] >= 0) q01 = p[tap[m][SINGLE ? 0 : 1]];
                            if (tap[m][SINGLE ? 0 : 2] >= 0) q10 = p[tap[m][SINGLE ? 0 : 2]];
                            if (tap[m][SINGLE ? 0 : 3] >= 0) q11 = p[tap[m][SINGLE ? 0 : 3]];
                        }
                        const T la = q00 * (1 - rW) + q01 * rW, lb = q10 * (1 - rW) + q11 * rW;
                        Q = (1 - rH) * la + rH * lb;                     // the reference's tree, contraction off
                        QH = la - lb;
                        QW = ((1 - rH) * q00 + rH * q10) - ((1 - rH) * q01 + rH * q11);
                    }
                    const T dx = xb[m] - xa[m];
                    const T mx = (1 - rT) * xb[m] + rT * xa[m];
                    sT += Q * dx;
                    sH += QH * mx;
                    sW += QW * mx;
                    if (WRITE_GX) {
                        res[m] = (1 - rT) * Qprev[m] + rT * Q;
                        if (!VEC && emit && iidx[m] >= 0) out[iidx[m]] = res[m];
                        Qprev[m] = Q;
                    }
                    xa[m] = xb[m];
                    xb[m] = xnext[m];
                }
                if constexpr (VEC && WRITE_GX) {
                    if (emit && iidx[0] >= 0)
                        *reinterpret_cast<float4*>(out + iidx[0]) = make_float4(res[0], res[1], res[2], res[3]);
                }
            }
            accT = sT; accH = sH; accW = sW;
        }
    }
    accT = group_sum(accT, cd.E, red[0]);
    accH = group_sum(accH, cd.E, red[1]);
    accW = group_sum(accW, cd.E, red[2]);
    if (id.valid && e == 0) {
        const int P = d.N * cd.nchunks;
        const size_t at = (size_t)id.c * 3 * P + (size_t)id.n * cd.nchunks + id.chunk;
        if constexpr (FUSED && std::is_same<T, float>::value) {
            dma::fin_publish(fin.f, at, accT); dma::fin_publish(fin.f, at + P, accH); dma::fin_publish(fin.f, at + 2 * (size_t)P, accW);
        } else {
            T* o = part + at;
            o[0] = accT;
            o[P] = accH;
            o[2 * P] = accW;
        }
    }
}

// ----------------------------------------------------------------------------------- host side
inline bool supported(const Dims3& d, int quantize) {
    return column_kernels_on() && !quantize && d.sT == 1 && d.pT == 0;
}

// plane_elems: the plane the threads index (output plane for forward, input plane for backward)
inline CDims make_cdims(const Dims3& d, int plane_elems) {
    CDims cd;
    cd.d = d;
    // <= 64 elements: one wave, 1 element per thread; <= 256: one wave, up to 4 elements per thread (4 independent
    // chains per thread, wave-only reduction, 4x fewer threads to set up); larger: 256 threads x 4 elements per chunk
    if (plane_elems <= kWave) { cd.E = kWave; cd.M = 1; }
    else if (plane_elems <= kBlock) { cd.E = kBlock; cd.M = 1; }
    else { cd.E = kBlock; cd.M = 4; }
    cd.logE = (cd.E == 64) ? 6 : (cd.E == 128 ? 7 : 8);
    cd.nchunks = (plane_elems + cd.E * cd.M - 1) / (cd.E * cd.M);
    return cd;
}
inline unsigned grid_of(const CDims& cd) {
    const long long groups = (long long)cd.d.N * cd.d.C * cd.nchunks;
    const int per_block = kBlock / cd.E;
    return (unsigned)((groups + per_block - 1) / per_block);
}

// 16-byte accesses on the streamed planes: fp32, plane a multiple of 4 elements, 16-byte aligned tensors
template <typename T> inline bool vec_ok(int plane_elems, const void* a, const void* b) {
    return std::is_same<T, float>::value && plane_elems % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0;
}

template <typename T>
inline int launch_forward(const T* x, const T* shift, T* y, const Dims3& d, hipStream_t stream) {
    const CDims cd = make_cdims(d, d.Ho * d.Wo);
    if (cd.M == 1)
        hipLaunchKernelGGL((k3d_forward_column<T, 1>), dim3(grid_of(cd)), dim3(kBlock), 0, stream, x, shift, y, cd);
    else
        hipLaunchKernelGGL((k3d_forward_column<T, 4>), dim3(grid_of(cd)), dim3(kBlock), 0, stream, x, shift, y, cd);
    return launch_status();
}

// returns P (partials per channel).  gshift != nullptr (fp32): row-sum + K5 inside the launch (ws = granule pairs);
// nullptr: plain partials ws[C][3][P] for k3d_finalize / the two-phase ABI
template <typename T>
inline int launch_backward(const T* x, const T* shift, const T* gy, T* gx, T* ws, const Dims3& d,
                           hipStream_t stream, T* gshift = nullptr, int normalize = 0, T t_factor = 1) {
    const CDims cd = make_cdims(d, d.H * d.W);
    const bool single = d.sH >= 2 && d.sW >= 2;
    const bool vec = cd.M == 4 && vec_ok<T>(d.H * d.W, x, gx);
    dma3d::Fin3 fin{};
    bool fused = false;
    if constexpr (std::is_same<T, float>::value) {
        if (gshift && streaming_kernels_on()) {
            fused = true;
            fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
            dma::fin_arm(fin.f);
            fin.f.producers = (int)grid_of(cd);
            fin.gshift = gshift;
            fin.normalize = normalize;
            fin.t_factor = t_factor;
        }
    }
    const unsigned grid = grid_of(cd) + (fused ? (unsigned)d.C : 0u);
#define RK_COL_BWD(GX, MM, SG, VC) do { \
        if constexpr (std::is_same<T, float>::value) { \
            if (fused) { hipLaunchKernelGGL((k3d_backward_column<T, GX, MM, SG, VC, true>), dim3(grid), dim3(kBlock), 0, stream, x, shift, gy, gx, ws, cd, fin); break; } \
        } \
        hipLaunchKernelGGL((k3d_backward_column<T, GX, MM, SG, VC>), dim3(grid), dim3(kBlock), 0, stream, x, shift, gy, gx, ws, cd, fin); } while (0)
#define RK_COL_SG(GX, MM, VC) do { if (single) RK_COL_BWD(GX, MM, true, VC); else RK_COL_BWD(GX, MM, false, VC); } while (0)
    if (gx) { if (cd.M == 1) RK_COL_SG(true, 1, false); else if (vec) RK_COL_SG(true, 4, true); else RK_COL_SG(true, 4, false); }
    else { if (cd.M == 1) RK_COL_SG(false, 1, false); else if (vec) RK_COL_SG(false, 4, true); else RK_COL_SG(false, 4, false); }
#undef RK_COL_SG
#undef RK_COL_BWD
    return fused ? -(d.N * cd.nchunks) : d.N * cd.nchunks;       // negative: finished inside the launch
}

}  // namespace col3d
}  // namespace rk
