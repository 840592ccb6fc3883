// rk2d_dma.hpp -- RubiksShift2D streaming kernels fed by LDS-DMA, fp32, stride 1 / pad 0, W % 4 == 0:
// the configuration every RubiksShift2D of the -aq networks runs in.
//
// x is [F, C, H, W] (F = N*T frames) and the shift is per channel, so every frame of a channel is
// translated and blended by the SAME (flH, rH, flW, rW).  One 256-thread workgroup owns
// (channel, row band, group of FG consecutive frames) and walks the frames with the ring / counted-vmcnt
// machinery of rk3d_dma.hpp (cells, tap slots, zero cell, compile-time tap offset): per-channel and
// per-thread geometry is computed once per FG planes, every LDS access is an aligned b128, loads and
// stores are non-temporal.  Unlike the 3-D operator there is no coupling between planes, so each step
// reads one plane and writes one.
//
// Arithmetic: the reference's own expression trees -- interp2d (rubiks2d_kernels.cu:60-66) for the
// forward (K6, :94-146) and for d(x) (K8, :276-378), evaluated with contraction off => bit-identical
// to the oracle.  d(shift) (K7, :164-273) uses the adjoint form: with the negated shift (fl', r'),
//     dL/dsH = sum_x x * (lerpW'(gy row fl'H) - lerpW'(gy row fl'H+1)),
//     dL/dsW = sum_x x * (col'[k] - col'[k+1]),   col'[k] = (1-r'H) gy(rowA, k) + r'H gy(rowB, k),
// so only gy needs taps and x is read at the thread's own cells: one pass, 12 B per element.
// Channels whose shift is within 1e-7 of an integer in H or W take the reference's central-difference
// branch (:189-253).  A central difference is the mean of the one-sided differences on either side, and
// in the adjoint form a one-sided difference at an integer shift is the SAME streaming sum evaluated with
// remainder 0 and floor fl' (one side) or fl' - 1 (the other).  So such a channel simply walks its
// frames two or three times with different floors (d(x) is stored on the first walk) -- about 2-3x the
// cost of an ordinary channel, instead of a per-element fallback that would serialise ~200 dependent
// loads per thread per plane in one workgroup (measured: ONE integer channel took [256,64,56,56] bf16
// from 81 us to 843 us).  ShiftNet-style "group" initialisation makes every channel integer.
#pragma once
#include "rk2d_generic.hpp"
#include "rk_dma.hpp"

namespace rk {
namespace dma2d {

using namespace dma;
using g2d::Dims2;

struct FDims {
    BDims b;                 // b.N = 1, b.T = frames
    int frames, FG, ngroups; // FG frames per workgroup
};

// ---------------------------------------------------------------------------------------------
// Forward (src = x, unprimed shift) and d(x) alone (src = gy, negated shift).
template <int ROUNDS, int D, int OFF>
__device__ __forceinline__ void interp2_loop(const float* __restrict__ sp, float* __restrict__ dp, float4* ring,
                                             const BDims& d, const Band& b, const Frac<float>& fH,
                                             const Frac<float>& fW, size_t fstride, int nf) {
    constexpr int R = D + 1;
    const int slot_f4 = b.cells_in + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 4);
    init_tap_slots<ROUNDS>(ring, R, slot_f4, b, cs);

    const float rH = fH.r, rW = fW.r;
    const float uH = 1 - rH, uW = 1 - rW;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const unsigned slot_bytes = (unsigned)slot_f4 * 16u;
    const float* src0 = sp + (ptrdiff_t)b.src0 * 4;
    float* out0 = dp + (size_t)b.out0 * 4;

    int issued = 0;
    auto feed = [&](int k, int s) {
        if (k < nf) {
            dma_taps<ROUNDS>(src0 + (size_t)k * fstride, ring_addr + s * slot_bytes, cs);
            issued += cs.n_tap_wave;
        }
    };
    int mark[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { feed(j, j); mark[j] = issued; }

    auto round = [&](int i, const float4* cur, float4* out, bool store) {
        const float4 qa0 = lds_b128(cur + cs.a0[i]), qa1 = lds_b128(cur + cs.a1[i]);
        const float4 qb0 = lds_b128(cur + cs.b0[i]), qb1 = lds_b128(cur + cs.b1[i]);
        float q[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)                               // interp2d, rubiks2d_kernels.cu:60-66
            q[m] = tap<OFF>(qa0, qa1, m) * uH * uW + tap<OFF>(qa0, qa1, m + 1) * uH * rW +
                   tap<OFF>(qb0, qb1, m) * rH * uW + tap<OFF>(qb0, qb1, m + 1) * rH * rW;
        if (store)
            stream_store(reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + cs.off0 + 4096 * i),
                         make_float4(q[0], q[1], q[2], q[3]));
    };

    int slot = 0;
#pragma nounroll
    for (int k = 0; k < nf; ++k) {
        wait_vmcnt(issued - mark[0]);                              // my pieces of frame k have landed
        __syncthreads();                                           // everyone's have; frame k-1 is retired
        {
            int sn = slot + D; if (sn >= R) sn -= R;
            feed(k + D, sn);
#pragma unroll
            for (int j = 0; j + 1 < D; ++j) mark[j] = mark[j + 1];
            mark[D - 1] = issued;
        }
        const float4* cur = ring + slot * slot_f4;
        float4* out = reinterpret_cast<float4*>(out0 + (size_t)k * fstride);
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, out, true);
        if (cs.tail_on) round(ROUNDS - 1, cur, out, cs.tail_live);
        issued += cs.n_out_wave;
        if (++slot == R) slot = 0;
    }
}

template <bool NEGATE, int ROUNDS, int D>
__global__ __launch_bounds__(kBlock) void k2d_dma_interp(const float* __restrict__ src,
                                                         const float* __restrict__ shift,
                                                         float* __restrict__ dst, FDims fd) {
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    const BDims& d = fd.b;
    const int band = blockIdx.x % d.nbands, col = blockIdx.x / d.nbands;
    const int c = col % d.C, g = col / d.C;
    float sH = shift[c], sW = shift[d.C + c];
    if (NEGATE) { sH = -sH; sW = -sW; }
    const Frac<float> fH = split_shift(sH), fW = split_shift(sW);
    const int HW = d.H * d.W;
    const size_t fstride = (size_t)d.C * HW;
    const int f0 = g * fd.FG;
    const int nf = min(fd.FG, fd.frames - f0);
    const float* sp = src + ((size_t)f0 * d.C + c) * HW;
    float* dp = dst + ((size_t)f0 * d.C + c) * HW;
    const Band b = make_band(d, band, fH.fl);

    if (NEGATE && sH == 0 && sW == 0) {                            // rubiks2d_kernels.cu:322-329: plain copy
        for (int k = 0; k < nf; ++k)
            for (int cell = threadIdx.x; cell < b.cells_out; cell += kBlock)
                reinterpret_cast<float4*>(dp + (size_t)k * fstride)[b.out0 + cell] =
                    reinterpret_cast<const float4*>(sp + (size_t)k * fstride)[b.out0 + cell];
        return;
    }
    switch (((fW.fl % 4) + 4) % 4) {                               // wave-uniform
        case 0: interp2_loop<ROUNDS, D, 0>(sp, dp, ring, d, b, fH, fW, fstride, nf); break;
        case 1: interp2_loop<ROUNDS, D, 1>(sp, dp, ring, d, b, fH, fW, fstride, nf); break;
        case 2: interp2_loop<ROUNDS, D, 2>(sp, dp, ring, d, b, fH, fW, fstride, nf); break;
        default: interp2_loop<ROUNDS, D, 3>(sp, dp, ring, d, b, fH, fW, fstride, nf); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Which walks over the frames a channel needs (wave-uniform).  (sH, sW): negated shift as d(shift) sees it --
// a component within 1e-7 above an integer i counts as exactly i (rubiks2d_kernels.cu:189-200), i.e.
// (fl', r') = (-i, 0).  Walk 0 uses (sH, sW); walk 1 (H integer) lowers the H floor by one, walk 2 (W integer)
// the W floor: the mean of walk 0 and walk 1 / 2 is the reference's central difference.  (gH, gW): the negated
// shift as d(x) sees it (K8 has no tolerance); when it differs from (sH, sW) d(x) gets a walk of its own.
struct IntegerPlan {
    Frac<float> sH, sW, gH, gW;
    bool hint, wint, separate_gx;
    __device__ __forceinline__ bool walk_on(int walk) const { return walk == 0 || (walk == 1 ? hint : wint); }
};
__device__ __forceinline__ IntegerPlan plan_walks(float s0, float s1) {
    IntegerPlan p;
    const Frac<float> u0 = split_shift(s0), u1 = split_shift(s1);
    p.hint = u0.r < 1e-7f;
    p.wint = u1.r < 1e-7f;
    p.gH = split_shift(-s0);
    p.gW = split_shift(-s1);
    p.sH = p.gH;
    p.sW = p.gW;
    if (p.hint) { p.sH.fl = -u0.fl; p.sH.r = 0.f; }
    if (p.wint) { p.sW.fl = -u1.fl; p.sW.r = 0.f; }
    p.separate_gx = p.sH.fl != p.gH.fl || p.sH.r != p.gH.r || p.sW.fl != p.gW.fl || p.sW.r != p.gW.r;
    return p;
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) + d(shift) partials in one pass.  part[c][2][P], P = ngroups * nbands.
template <int ROUNDS, int DG, int DX, int OFF, bool WRITE_GX>
__device__ __forceinline__ void backward2_loop(const float* __restrict__ xp, const float* __restrict__ gp,
                                               float* __restrict__ op, float4* ring, const BDims& d, const Band& b,
                                               const Frac<float>& fH, const Frac<float>& fW, size_t fstride, int nf,
                                               float& accH, float& accW) {
    static_assert(DG >= DX && DX >= 1, "gy runs at least as far ahead as x");
    __syncthreads();                                              // a previous walk may still be reading the ring
    constexpr int RG = DG + 1, RX = DX;
    const int gslot_f4 = b.cells_in + 1, xslot_f4 = b.cells_out + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 4);
    float4* const gring = ring;
    float4* const xring = ring + RG * gslot_f4;
    init_tap_slots<ROUNDS>(gring, RG, gslot_f4, b, cs);
    if (threadIdx.x < RX) xring[threadIdx.x * xslot_f4 + b.cells_out] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float rH = fH.r, rW = fW.r;
    const float uH = 1 - rH, uW = 1 - rW;
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(gring));
    const unsigned xaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(xring));
    const unsigned gslot_bytes = (unsigned)gslot_f4 * 16u, xslot_bytes = (unsigned)xslot_f4 * 16u;
    const float* gsrc0 = gp + (ptrdiff_t)b.src0 * 4;
    const float* xsrc0 = xp + (size_t)b.out0 * 4;
    float* out0 = op + (size_t)b.out0 * 4;

    float sH = 0.f, sW = 0.f;
    int issued = 0;
    auto feed_g = [&](int k, int s) {
        if (k < nf) {
            dma_taps<ROUNDS>(gsrc0 + (size_t)k * fstride, gaddr + s * gslot_bytes, cs);
            issued += cs.n_tap_wave;
        }
    };
    auto feed_x = [&](int k, int s) {
        if (k < nf) {
            dma_own<ROUNDS>(xsrc0 + (size_t)k * fstride, xaddr + s * xslot_bytes, cs);
            issued += cs.n_out_wave;
        }
    };
    int mark[DX];
#pragma unroll
    for (int j = 0; j < DG; ++j) {
        feed_g(j, j);
        if (j < DX) { feed_x(j, j); mark[j] = issued; }
    }

    auto round = [&](int i, const float4* cur, const float4& xv4, float4* out, bool store) {
        const float4 qa0 = lds_b128(cur + cs.a0[i]), qa1 = lds_b128(cur + cs.a1[i]);
        const float4 qb0 = lds_b128(cur + cs.b0[i]), qb1 = lds_b128(cur + cs.b1[i]);
        const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
        float col[5], q[4];
#pragma unroll
        for (int m = 0; m < 5; ++m) col[m] = fmaf(uH, tap<OFF>(qa0, qa1, m), rH * tap<OFF>(qb0, qb1, m));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float a0 = tap<OFF>(qa0, qa1, m), a1 = tap<OFF>(qa0, qa1, m + 1);
            const float b0 = tap<OFF>(qb0, qb1, m), b1 = tap<OFF>(qb0, qb1, m + 1);
            q[m] = a0 * uH * uW + a1 * uH * rW + b0 * rH * uW + b1 * rH * rW;     // K8: interp2d, contraction off
            const float la = fmaf(a0, uW, a1 * rW), lb = fmaf(b0, uW, b1 * rW);
            sH = fmaf(la - lb, xv[m], sH);
            sW = fmaf(col[m] - col[m + 1], xv[m], sW);
        }
        if (store)
            stream_store(reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + cs.off0 + 4096 * i),
                         make_float4(q[0], q[1], q[2], q[3]));
    };
    const int n_store_wave = WRITE_GX ? cs.n_out_wave : 0;

    int gslot = 0, xslot = 0;
#pragma nounroll
    for (int k = 0; k < nf; ++k) {
        wait_vmcnt(issued - mark[0]);                             // my pieces of gy[k] and x[k] have landed
        __syncthreads();                                          // everyone's gy pieces have; step k-1 retired
        const char* xs = reinterpret_cast<const char*>(xring + xslot * xslot_f4);
        float4 xv[ROUNDS];
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) xv[i] = *reinterpret_cast<const float4*>(xs + cs.off0 + 4096 * i);
        xv[ROUNDS - 1] = reinterpret_cast<const float4*>(xs)[cs.xown];
        {
            int gs = gslot + DG; if (gs >= RG) gs -= RG;
            feed_g(k + DG, gs);
            feed_x(k + DX, xslot);                                // (the DMA waits for the LDS reads above)
#pragma unroll
            for (int j = 0; j + 1 < DX; ++j) mark[j] = mark[j + 1];
            mark[DX - 1] = issued;
        }
        const float4* cur = gring + gslot * gslot_f4;
        float4* out = reinterpret_cast<float4*>(out0 + (size_t)k * fstride);
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, xv[i], out, WRITE_GX);
        if (cs.tail_on) round(ROUNDS - 1, cur, xv[ROUNDS - 1], out, WRITE_GX && cs.tail_live);
        issued += n_store_wave;
        if (++gslot == RG) gslot = 0;
        if (++xslot == RX) xslot = 0;
    }
    accH = sH; accW = sW;
}

// K9 (rubiks2d_kernels.cu:381-397) on the row sums, written in the storage type of gshift
template <typename S> struct Fin2 {
    Fin f;
    S* gshift;                    // [2][C]
    int normalize;
};
template <typename S>
__device__ __forceinline__ void finalizer_wave2(const Fin2<S>& fin, int c, int C, int P) {
    double s[2];
    const bool ok = fin_collect<2>(fin.f, c, P, s);
    if (threadIdx.x == 0) {
        float gH = (float)s[0], gW = (float)s[1];
        if (fin.normalize) {
            const float mag = sqrtf(gH * gH + gW * gW);
            if (mag > 0) { gH = gH / mag; gW = gW / mag; }
        }
        if (!ok) gH = gW = __uint_as_float(0x7fc00000u);
        st(fin.gshift + c, gH);
        st(fin.gshift + C + c, gW);
    }
}

// training fusion (bn2 + ReLU inside the 2-D shift, round 5): what the backward needs of bn2 and where bn2's constants go
struct BnFuse2 {
    const float4* abmi;           // [C] (a, b, mean, invstd)
    float* k12;                   // [2][C]
    float* dgamma; float* dbeta;  // [C]
    float inv_count;              // 1 / (F H W)
};
template <typename S>
__device__ __forceinline__ void finalizer_wave2_bn(const Fin2<S>& fin, int c, int C, int P, const BnFuse2& bn) {
    double s[4];
    const bool ok = fin_collect<4>(fin.f, c, P, s);
    if (threadIdx.x == 0) {
        const float nanv = __uint_as_float(0x7fc00000u);
        float gH = (float)s[0], gW = (float)s[1];
        if (fin.normalize) {
            const float mag = sqrtf(gH * gH + gW * gW);
            if (mag > 0) { gH = gH / mag; gW = gW / mag; }
        }
        if (!ok) gH = gW = nanv;
        st(fin.gshift + c, gH);
        st(fin.gshift + C + c, gW);
        bn.dbeta[c] = ok ? (float)s[2] : nanv;
        bn.dgamma[c] = ok ? (float)s[3] : nanv;
        bn.k12[c] = ok ? (float)(s[2] * (double)bn.inv_count) : nanv;
        bn.k12[C + c] = ok ? (float)(s[3] * (double)bn.inv_count) : nanv;
    }
}
template <int ROUNDS, int DG, int DX>
__global__ __launch_bounds__(kBlock) void k2d_dma_backward(const float* __restrict__ gy,
                                                           const float* __restrict__ x,
                                                           const float* __restrict__ shift,
                                                           float* __restrict__ gx, FDims fd, Dims2 gd, Fin2<float> fin) {
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    __shared__ float red[2][kBlock / kWave];
    (void)gd;
    const BDims& d = fd.b;
    if ((int)blockIdx.x >= fin.f.producers) {                         // row-sum + K9 inside the launch (rk_dma.hpp)
        if (threadIdx.x < kWave) finalizer_wave2(fin, (int)blockIdx.x - fin.f.producers, d.C, fd.ngroups * d.nbands);
        return;
    }
    const int band = blockIdx.x % d.nbands, col = blockIdx.x / d.nbands;
    const int c = col % d.C, g = col / d.C;
    const float s0 = shift[c], s1 = shift[d.C + c];
    const int f0 = g * fd.FG;
    const int nf = min(fd.FG, fd.frames - f0);

    const int HW = d.H * d.W;
    const size_t fstride = (size_t)d.C * HW;
    const size_t base = ((size_t)f0 * d.C + c) * HW;
    const IntegerPlan plan = plan_walks(s0, s1);
    if (plan.separate_gx) {                                       // |r| < 1e-7 but not 0: K8 keeps the true remainder
        const Band b = make_band(d, band, plan.gH.fl);
        switch (((plan.gW.fl % 4) + 4) % 4) {
            case 0: interp2_loop<ROUNDS, 1, 0>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf); break;
            case 1: interp2_loop<ROUNDS, 1, 1>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf); break;
            case 2: interp2_loop<ROUNDS, 1, 2>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf); break;
            default: interp2_loop<ROUNDS, 1, 3>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf); break;
        }
    }
    float sumH0 = 0.f, sumW0 = 0.f, sumH1 = 0.f, sumW2 = 0.f;
    if (!plan.separate_gx) {                                      // walk 0 with d(x): every ordinary channel ends here
        const Frac<float> fH = plan.sH, fW = plan.sW;
        const Band b = make_band(d, band, fH.fl);
        float aH = 0.f, aW = 0.f;
        switch (((fW.fl % 4) + 4) % 4) {
            case 0: backward2_loop<ROUNDS, DG, DX, 0, true>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW); break;
            case 1: backward2_loop<ROUNDS, DG, DX, 1, true>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW); break;
            case 2: backward2_loop<ROUNDS, DG, DX, 2, true>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW); break;
            default: backward2_loop<ROUNDS, DG, DX, 3, true>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW); break;
        }
        sumH0 = aH; sumW0 = aW;
    }
    if (plan.separate_gx || plan.hint || plan.wint) {
#pragma nounroll
        for (int walk = plan.separate_gx ? 0 : 1; walk < 3; ++walk) {   // sums only
            if (!plan.walk_on(walk)) continue;
            Frac<float> fH = plan.sH, fW = plan.sW;
            if (walk == 1) fH.fl -= 1;
            if (walk == 2) fW.fl -= 1;
            const Band b = make_band(d, band, fH.fl);
            float aH = 0.f, aW = 0.f;
            switch (((fW.fl % 4) + 4) % 4) {
                case 0: backward2_loop<ROUNDS, DG, DX, 0, false>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW); break;
                case 1: backward2_loop<ROUNDS, DG, DX, 1, false>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW); break;
                case 2: backward2_loop<ROUNDS, DG, DX, 2, false>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW); break;
                default: backward2_loop<ROUNDS, DG, DX, 3, false>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW); break;
            }
            if (walk == 0) { sumH0 = aH; sumW0 = aW; }
            else if (walk == 1) sumH1 = aH;
            else sumW2 = aW;
        }
    }
    float accH = plan.hint ? 0.5f * (sumH0 + sumH1) : sumH0;
    float accW = plan.wint ? 0.5f * (sumW0 + sumW2) : sumW0;

    accH = group_sum(accH, kBlock, red[0]);
    accW = group_sum(accW, kBlock, red[1]);
    if (threadIdx.x == 0) {
        const int P = fd.ngroups * d.nbands;
        const size_t at = (size_t)c * 2 * P + (size_t)g * d.nbands + band;
        fin_publish(fin.f, at, accH);
        fin_publish(fin.f, at + P, accW);
    }
}

// ---------------------------------------------------------------------------------------------
// Host side.
// false = shape not handled here (stride / padding / W % 4 / RK_SHIFT_KERNELS)
inline bool make_fdims(FDims& f, const Dims2& d, int frames_per_group, bool backward = false) {
    const bool s1p0 = d.sH == 1 && d.sW == 1 && d.pH == 0 && d.pW == 0;
    if (!s1p0 || d.W % 4 != 0 || d.W < 4 || !streaming_kernels_on()) return false;
    BDims& b = f.b;
    b.N = 1; b.T = d.N; b.C = d.C; b.H = d.H; b.W = d.W; b.W4 = d.W / 4;
    if (!choose_bands(b, backward)) return false;      // (backward: two-round bands, rk_dma.hpp)
    f.frames = d.N;
    f.FG = frames_per_group < d.N ? frames_per_group : d.N;
    f.ngroups = (d.N + f.FG - 1) / f.FG;
    return true;
}
// Frames per workgroup, measured on [256,64,56,56] (fwd / bwd us): fp32 DMA kernels 1: 72/133, 2: 70/108,
// 4: 73/111, 8: 76/114, 16: 73/115 -- many short-lived workgroups keep the read/write mix of the chip even;
// the register-staged 16-bit kernels (deeper per-workgroup prologue) 2: 49/96, 4: 45/85, 8: 46/81, 16: 44/81.
constexpr int kFramesF32 = 2, kFrames16 = 8;

template <bool NEGATE>
inline bool launch_interp2(const float* src, const float* shift, float* dst, const Dims2& d, hipStream_t stream) {
    constexpr int D = 2;
    FDims f;
    if (!make_fdims(f, d, kFramesF32) || !aligned16(src) || !aligned16(dst)) return false;
    const size_t lds = interp_ring_bytes(f.b, D);
    if (lds > 64 * 1024) return false;
    const dim3 grid((unsigned)(f.ngroups * f.b.C * f.b.nbands)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_dma_interp<NEGATE, 1, D>), grid, block, lds, stream, src, shift, dst, f); break;
        case 2: hipLaunchKernelGGL((k2d_dma_interp<NEGATE, 2, D>), grid, block, lds, stream, src, shift, dst, f); break;
        case 3: hipLaunchKernelGGL((k2d_dma_interp<NEGATE, 3, D>), grid, block, lds, stream, src, shift, dst, f); break;
        default: hipLaunchKernelGGL((k2d_dma_interp<NEGATE, 4, D>), grid, block, lds, stream, src, shift, dst, f); break;
    }
    return true;
}

// partials per channel the fused backward writes for this shape (0 = shape not handled here)
inline int backward2_partials(const Dims2& d, int frames_per_group) {
    FDims f;
    return make_fdims(f, d, frames_per_group, true) ? f.ngroups * f.b.nbands : 0;
}

// d(x) + d(shift) (row-sum + K9 inside the launch: ws holds granules [C][2][P]); false = not handled here
inline bool launch_backward2(const float* gy, const float* x, const float* shift, float* gx, float* gshift, void* ws,
                             int normalize, const Dims2& d, hipStream_t stream) {
    constexpr int DG = 1, DX = 1;
    FDims f;
    if (!make_fdims(f, d, kFramesF32, true) || !aligned16(gy) || !aligned16(x) || !aligned16(gx)) return false;
    const size_t lds = bwd_ring_bytes(f.b, DG, DX);
    if (lds > 64 * 1024) return false;
    Fin2<float> fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = f.ngroups * f.b.C * f.b.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    const dim3 grid((unsigned)(fin.f.producers + f.b.C)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_dma_backward<1, DG, DX>), grid, block, lds, stream, gy, x, shift, gx, f, d, fin); break;
        case 2: hipLaunchKernelGGL((k2d_dma_backward<2, DG, DX>), grid, block, lds, stream, gy, x, shift, gx, f, d, fin); break;
        case 3: hipLaunchKernelGGL((k2d_dma_backward<3, DG, DX>), grid, block, lds, stream, gy, x, shift, gx, f, d, fin); break;
        default: hipLaunchKernelGGL((k2d_dma_backward<4, DG, DX>), grid, block, lds, stream, gy, x, shift, gx, f, d, fin); break;
    }
    return true;
}

}  // namespace dma2d
}  // namespace rk
