// rk3d_translate.hpp -- RubiksShift3D with quantize=True, stride 1 / pad 0, fp32, W % 4 == 0: the forward
// (rubiks3d_kernels.cu:76-93) and d(x) (:819-827 ff., the "single tap at the nearest position" branch) are pure
// plane translations -- out[n,t,c,h,w] = src[n, t+aT, c, h+aH, w+aW] or 0 -- with a = the shift (forward) or the
// negated shift (d(x)) rounded as the reference rounds it (floor, +1 when the remainder is >= 0.5).
//
// One group of E = 64 / 128 / 256 threads per output plane (the grouping of rk3d_generic.hpp), a thread owns cells
// of 4 consecutive outputs: 4 scalar gathers (the source run is only 4-byte aligned; the four instructions of a wave
// share their cache lines) and ONE 16-byte non-temporal store; (h, w) advance incrementally.  The per-plane generic
// kernel did the same with 4-byte stores: [32,8,64,56,56] forward 103 us.
#pragma once
#include "rk3d_generic.hpp"
#include "rk_dma.hpp"

namespace rk {
namespace xlate3d {

template <bool NEGATE>
__global__ __launch_bounds__(kBlock) void k3d_translate(const float* __restrict__ src, const float* __restrict__ shift,
                                                        float* __restrict__ dst, Dims3 d) {
    int e;
    const PlaneId pl = my_plane(d, d.T, e);
    if (!pl.valid) return;
    float s0 = shift[pl.c], s1 = shift[d.C + pl.c], s2 = shift[2 * d.C + pl.c];
    if (NEGATE) { s0 = -s0; s1 = -s1; s2 = -s2; }
    const Frac<float> fT = split_shift(s0), fH = split_shift(s1), fW = split_shift(s2);
    const int aT = (fT.r < 0.5f) ? fT.fl : fT.fl + 1;
    const int aH = (fH.r < 0.5f) ? fH.fl : fH.fl + 1;
    const int aW = (fW.r < 0.5f) ? fW.fl : fW.fl + 1;

    const int HW = d.H * d.W, W4 = d.W >> 2, cells = HW >> 2;
    const int ts = pl.t + aT;
    const bool vt = ts >= 0 && ts < d.T;
    const float* sp = src + (((size_t)pl.n * d.T + (vt ? ts : 0)) * d.C + pl.c) * HW;
    float4* dp = reinterpret_cast<float4*>(dst + (((size_t)pl.n * d.T + pl.t) * d.C + pl.c) * HW);

    int h = e / W4, w4 = e - h * W4;
    const int dh = d.E / W4, dw4 = d.E - dh * W4;
    for (int cell = e; cell < cells; cell += d.E) {
        const int hs = h + aH, w0 = 4 * w4 + aW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vt && hs >= 0 && hs < d.H) {
            const float* row = sp + hs * d.W;
            if (w0 >= 0 && w0 < d.W) v.x = row[w0];
            if (w0 + 1 >= 0 && w0 + 1 < d.W) v.y = row[w0 + 1];
            if (w0 + 2 >= 0 && w0 + 2 < d.W) v.z = row[w0 + 2];
            if (w0 + 3 >= 0 && w0 + 3 < d.W) v.w = row[w0 + 3];
        }
        dma::stream_store(dp + cell, v);
        w4 += dw4; h += dh;
        if (w4 >= W4) { w4 -= W4; ++h; }
    }
}

// false = not handled here (stride / padding / W % 4 / alignment)
template <bool NEGATE>
inline bool launch(const float* src, const float* shift, float* dst, Dims3 d, hipStream_t stream) {
    const bool s1p0 = d.sT == 1 && d.sH == 1 && d.sW == 1 && d.pT == 0 && d.pH == 0 && d.pW == 0;
    if (!s1p0 || d.W % 4 != 0 || !dma::aligned16(dst) || !streaming_kernels_on()) return false;
    d.E = pow2_at_least(d.H * d.W / 4, kWave, kBlock);
    d.logE = (d.E == 64) ? 6 : (d.E == 128 ? 7 : 8);
    const long long planes = (long long)d.N * d.T * d.C;
    const int per_block = kBlock / d.E;
    hipLaunchKernelGGL((k3d_translate<NEGATE>), dim3((unsigned)((planes + per_block - 1) / per_block)), dim3(kBlock), 0,
                       stream, src, shift, dst, d);
    return true;
}

}  // namespace xlate3d
}  // namespace rk
