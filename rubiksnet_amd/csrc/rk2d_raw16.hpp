// rk2d_raw16.hpp -- RubiksShift2D streaming kernels for the 16-bit storage types (f16, bf16) on planes with
// W % 8 == 0 (stride 1 / pad 0): the 56x56 and 112x112 layers of the -aq networks under autocast.
//
// rk2d_stage.hpp widens a plane to fp32 on the way INTO LDS, so per element it executes what the fp32 kernels do
// (cells of 4, four b128 tap reads per cell, a b128 deposit) on half the bytes, and that instruction / LDS side --
// hidden under the memory time in fp32 -- is what bounds it.  Here the planes stay RAW in LDS:
//   * a cell is 16 bytes = 8 elements, so the whole band / cell / tap-slot / zero-cell geometry of rk_dma.hpp is
//     reused unchanged with "cells per row" = W / 8, and the feed is the same LDS-DMA (global_load_lds_dwordx4,
//     counted vmcnt) as the fp32 kernels': no register staging, no deposit;
//   * a thread produces 8 outputs from four b128 reads (rows A and B x an aligned 16-element window) -- half the LDS
//     instructions per element, LDS footprint halved -- and widens on the LDS -> register side: the 9 consecutive
//     elements starting OFF = flW mod 8 (compile time, 8 copies of the loop) of each window are one shift / and each;
//   * outputs are rounded once (v_cvt_pk_bf16_f32) and leave as 16-byte nt stores; x (backward) is DMA'd into a
//     one-slot ring and read back at the thread's own cell.
// Arithmetic: the same fp32 expression trees as rk2d_dma.hpp / rk2d_stage.hpp (interp2d rubiks2d_kernels.cu:60-66
// for K6 :94-146 and K8 :276-378; adjoint-form d(shift) for K7 :164-273; integer-shift walks as in rk2d_dma.hpp),
// contraction off, one rounding on store => bit-identical to those kernels and to the oracle on the widened inputs.
#pragma once
#include "rk2d_stage.hpp"

namespace rk {
namespace raw16 {

using namespace dma;
using dma2d::FDims;
using dma2d::Fin2;
using g2d::Dims2;

// element `half` (0 = low 16 bits) of a 32-bit word, widened
template <typename T> struct Wide;
template <> struct Wide<__hip_bfloat16> {
    __device__ static __forceinline__ float get(unsigned w, int half) {
        return __uint_as_float(half ? (w & 0xffff0000u) : (w << 16));
    }
};
template <> struct Wide<__half> {
    __device__ static __forceinline__ float get(unsigned w, int half) {
        return __half2float(__builtin_bit_cast(__half, (unsigned short)(half ? (w >> 16) : (w & 0xffffu))));
    }
};

struct Raw8 { unsigned w[4]; };
__device__ __forceinline__ Raw8 lds_raw(const float4* p) {
    const float4 v = lds_b128(p);
    return Raw8{{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}};
}
// the 9 consecutive elements starting OFF elements into the aligned 16-element window (q0, q1); OFF + m is constant
// after unrolling
template <typename T, int OFF> __device__ __forceinline__ void taps9(const Raw8& q0, const Raw8& q1, float (&t)[9]) {
#pragma unroll
    for (int m = 0; m < 9; ++m) {
        const int j = OFF + m;
        t[m] = Wide<T>::get(j < 8 ? q0.w[j >> 1] : q1.w[(j - 8) >> 1], j & 1);
    }
}
template <typename T> __device__ __forceinline__ void widen8(const Raw8& q, float (&t)[8]) {
#pragma unroll
    for (int m = 0; m < 8; ++m) t[m] = Wide<T>::get(q.w[m >> 1], m & 1);
}
// a value as the storage type holds it
template <typename T> __device__ __forceinline__ float round16(float v) { return Wide<T>::get(stage2d::Cell4<T>::bits(v), 0); }
// Training fusion (bn2 + ReLU inside the shift, fused_bn.bn_relu_shift2d): the planes a kernel reads are z = conv2's output
// and the shift applies to max(a z + b, 0) rounded to the storage type.  The wave that DMA'd a cell transforms it in place
// once it has landed (its own counted vmcnt wait) and before the step's barrier -- the slot's DMA'd cells only, so rows
// outside the plane and the zero cell stay zero (the 16-bit twin of rk_dma.hpp's bn_taps).
template <typename T, int ROUNDS>
__device__ __forceinline__ void bn_taps16(float4* slot, const BCells<ROUNDS>& cs, float a, float b) {
    char* base = reinterpret_cast<char*>(slot) + cs.off0;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
        if (cs.in_act[i]) {
            uint4* p = reinterpret_cast<uint4*>(base + 4096 * i);
            const uint4 w = *p;
            const unsigned in[4] = {w.x, w.y, w.z, w.w};
            unsigned out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v0 = fmaxf(fmaf(a, Wide<T>::get(in[j], 0), b), 0.f), v1 = fmaxf(fmaf(a, Wide<T>::get(in[j], 1), b), 0.f);
                out[j] = stage2d::Cell4<T>::bits(v0) | (stage2d::Cell4<T>::bits(v1) << 16);
            }
            *p = make_uint4(out[0], out[1], out[2], out[3]);
        }
}
template <typename T> __device__ __forceinline__ void store8(void* p, const float (&q)[8]) {
    using C4 = stage2d::Cell4<T>;
    const uint2 lo = C4::narrow(q[0], q[1], q[2], q[3]), hi = C4::narrow(q[4], q[5], q[6], q[7]);
    stream_store(reinterpret_cast<float4*>(p), make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y),
                                                            __uint_as_float(hi.x), __uint_as_float(hi.y)));
}

// ---------------------------------------------------------------------------------------------
// Forward (src = x) and d(x) alone (src = gy, negated shift).  sp / dp: frame 0 of the group, this channel.
template <typename T, int ROUNDS, int D, int OFF, bool BN = false>
__device__ __forceinline__ void interp2_loop(const T* __restrict__ sp, T* __restrict__ dp, float4* ring,
                                             const BDims& d, const Band& b, const Frac<float>& fH,
                                             const Frac<float>& fW, size_t fstride, int nf, float bn_a = 1.f, float bn_b = 0.f) {
    constexpr int R = D + 1;
    __syncthreads();                                              // (a backward walk may have used the ring before)
    const int slot_f4 = b.cells_in + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 8);
    init_tap_slots<ROUNDS>(ring, R, slot_f4, b, cs);

    const float rH = fH.r, rW = fW.r;
    const float uH = 1 - rH, uW = 1 - rW;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const unsigned slot_bytes = (unsigned)slot_f4 * 16u;
    const T* src0 = sp + (ptrdiff_t)b.src0 * 8;
    T* out0 = dp + (size_t)b.out0 * 8;

    int issued = 0;
    auto feed = [&](int k, int s) {
        if (k < nf) {
            dma_taps<ROUNDS>(reinterpret_cast<const float*>(src0 + (size_t)k * fstride), ring_addr + s * slot_bytes, cs);
            issued += cs.n_tap_wave;
        }
    };
    int mark[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { feed(j, j); mark[j] = issued; }

    auto round = [&](int i, const float4* cur, T* out, bool store) {
        const Raw8 qa0 = lds_raw(cur + cs.a0[i]), qa1 = lds_raw(cur + cs.a1[i]);
        const Raw8 qb0 = lds_raw(cur + cs.b0[i]), qb1 = lds_raw(cur + cs.b1[i]);
        float ta[9], tb[9], q[8];
        taps9<T, OFF>(qa0, qa1, ta);
        taps9<T, OFF>(qb0, qb1, tb);
#pragma unroll
        for (int m = 0; m < 8; ++m)                               // interp2d, rubiks2d_kernels.cu:60-66
            q[m] = ta[m] * uH * uW + ta[m + 1] * uH * rW + tb[m] * rH * uW + tb[m + 1] * rH * rW;
        if (store) store8<T>(reinterpret_cast<char*>(out) + cs.off0 + 4096 * i, q);
    };

    int slot = 0;
#pragma nounroll
    for (int k = 0; k < nf; ++k) {
        wait_vmcnt(issued - mark[0]);                              // my pieces of frame k have landed
        if (BN) bn_taps16<T, ROUNDS>(ring + slot * slot_f4, cs, bn_a, bn_b);
        __syncthreads();                                           // everyone's have; frame k-1 is retired
        {
            int sn = slot + D; if (sn >= R) sn -= R;
            feed(k + D, sn);
#pragma unroll
            for (int j = 0; j + 1 < D; ++j) mark[j] = mark[j + 1];
            mark[D - 1] = issued;
        }
        const float4* cur = ring + slot * slot_f4;
        T* out = out0 + (size_t)k * fstride;
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, out, true);
        if (cs.tail_on) round(ROUNDS - 1, cur, out, cs.tail_live);
        issued += cs.n_out_wave;
        if (++slot == R) slot = 0;
    }
}

#define RK_OFF8_SWITCH(off, CALL)                                                                                  \
    switch (off) {                                                                                                 \
        case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break;            \
        case 4: CALL(4); break; case 5: CALL(5); break; case 6: CALL(6); break; default: CALL(7); break;           \
    }
__device__ __forceinline__ int off8(int fl) { return ((fl % 8) + 8) % 8; }

template <typename T, typename S, bool NEGATE, int ROUNDS, int D, bool BN = false>
__global__ __launch_bounds__(kBlock) void k2d_raw16_interp(const T* __restrict__ src, const S* __restrict__ shift,
                                                           T* __restrict__ dst, FDims fd, const float* __restrict__ ab = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    const BDims& d = fd.b;
    const int band = blockIdx.x % d.nbands, col = blockIdx.x / d.nbands;
    const int c = col % d.C, g = col / d.C;
    float sH = ld(shift + c), sW = ld(shift + d.C + c);
    if (NEGATE) { sH = -sH; sW = -sW; }
    const Frac<float> fH = split_shift(sH), fW = split_shift(sW);
    const int HW = d.H * d.W;
    const size_t fstride = (size_t)d.C * HW;
    const int f0 = g * fd.FG;
    const int nf = min(fd.FG, fd.frames - f0);
    const T* sp = src + ((size_t)f0 * d.C + c) * HW;
    T* dp = dst + ((size_t)f0 * d.C + c) * HW;
    const Band b = make_band(d, band, fH.fl);

    if (NEGATE && sH == 0 && sW == 0) {                            // rubiks2d_kernels.cu:322-329: plain copy
        for (int k = 0; k < nf; ++k)
            for (int cell = threadIdx.x; cell < b.cells_out; cell += kBlock)
                reinterpret_cast<uint4*>(dp + (size_t)k * fstride)[b.out0 + cell] =
                    reinterpret_cast<const uint4*>(sp + (size_t)k * fstride)[b.out0 + cell];
        return;
    }
    const float bn_a = BN ? ab[c] : 1.f, bn_b = BN ? ab[d.C + c] : 0.f;
#define RK_CALL(O) interp2_loop<T, ROUNDS, D, O, BN>(sp, dp, ring, d, b, fH, fW, fstride, nf, bn_a, bn_b)
    RK_OFF8_SWITCH(off8(fW.fl), RK_CALL)                           // wave-uniform
#undef RK_CALL
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) + d(shift) partials in one pass (adjoint form, see rk2d_dma.hpp).
// BN: x holds z = bn2's input; the activation max(a z + b, 0) (rounded: what the forward shifted) is recomputed at the thread's
// own cell, d(x) leaves ReLU-masked and bn2's two sums ride along (accB1 = sum dz, accB2 = sum dz zhat; bnp = a, b, mean, invstd)
template <typename T, int ROUNDS, int OFF, bool WRITE_GX, bool BN = false>
__device__ __forceinline__ void backward2_loop(const T* __restrict__ xp, const T* __restrict__ gp,
                                               T* __restrict__ op, float4* ring, const BDims& d, const Band& b,
                                               const Frac<float>& fH, const Frac<float>& fW, size_t fstride, int nf,
                                               float& accH, float& accW, float4 bnp = make_float4(1.f, 0.f, 0.f, 1.f),
                                               float* accB1 = nullptr, float* accB2 = nullptr) {
    constexpr int DG = 1, DX = 1, RG = DG + 1, RX = DX;
    float sB1 = 0.f, sB2 = 0.f;
    __syncthreads();                                              // a previous walk may still be reading the ring
    const int gslot_f4 = b.cells_in + 1, xslot_f4 = b.cells_out + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 8);
    float4* const gring = ring;
    float4* const xring = ring + RG * gslot_f4;
    init_tap_slots<ROUNDS>(gring, RG, gslot_f4, b, cs);
    if (threadIdx.x < RX) xring[threadIdx.x * xslot_f4 + b.cells_out] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float rH = fH.r, rW = fW.r;
    const float uH = 1 - rH, uW = 1 - rW;
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(gring));
    const unsigned xaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(xring));
    const unsigned gslot_bytes = (unsigned)gslot_f4 * 16u, xslot_bytes = (unsigned)xslot_f4 * 16u;
    const T* gsrc0 = gp + (ptrdiff_t)b.src0 * 8;
    const T* xsrc0 = xp + (size_t)b.out0 * 8;
    T* out0 = op + (size_t)b.out0 * 8;

    float sH = 0.f, sW = 0.f;
    int issued = 0;
    auto feed_g = [&](int k, int s) {
        if (k < nf) {
            dma_taps<ROUNDS>(reinterpret_cast<const float*>(gsrc0 + (size_t)k * fstride), gaddr + s * gslot_bytes, cs);
            issued += cs.n_tap_wave;
        }
    };
    auto feed_x = [&](int k, int s) {
        if (k < nf) {
            dma_own<ROUNDS>(reinterpret_cast<const float*>(xsrc0 + (size_t)k * fstride), xaddr + s * xslot_bytes, cs);
            issued += cs.n_out_wave;
        }
    };
    int mark[DX];
#pragma unroll
    for (int j = 0; j < DG; ++j) {
        feed_g(j, j);
        if (j < DX) { feed_x(j, j); mark[j] = issued; }
    }

    auto round = [&](int i, const float4* cur, const Raw8& xraw, T* out, bool store, bool live) {
        const Raw8 qa0 = lds_raw(cur + cs.a0[i]), qa1 = lds_raw(cur + cs.a1[i]);
        const Raw8 qb0 = lds_raw(cur + cs.b0[i]), qb1 = lds_raw(cur + cs.b1[i]);
        float ta[9], tb[9], xv[8], col[9], q[8], zv[8];
        taps9<T, OFF>(qa0, qa1, ta);
        taps9<T, OFF>(qb0, qb1, tb);
        widen8<T>(xraw, xv);
        if (BN) {                                                  // (a lane without a cell reads the zero cell: relu(b) is not 0)
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                zv[m] = xv[m];
                xv[m] = live ? round16<T>(fmaxf(fmaf(bnp.x, zv[m], bnp.y), 0.f)) : 0.f;
            }
        }
#pragma unroll
        for (int m = 0; m < 9; ++m) col[m] = fmaf(uH, ta[m], rH * tb[m]);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            q[m] = ta[m] * uH * uW + ta[m + 1] * uH * rW + tb[m] * rH * uW + tb[m + 1] * rH * rW;   // K8: interp2d
            const float la = fmaf(ta[m], uW, ta[m + 1] * rW), lb = fmaf(tb[m], uW, tb[m + 1] * rW);
            sH = fmaf(la - lb, xv[m], sH);
            sW = fmaf(col[m] - col[m + 1], xv[m], sW);
            if (BN && WRITE_GX) {                                  // d(bn2's output): as stored, ReLU-masked; bn2's sums
                q[m] = xv[m] > 0.f ? round16<T>(q[m]) : 0.f;
                sB1 += q[m];
                sB2 = fmaf(q[m], (zv[m] - bnp.z) * bnp.w, sB2);
            }
        }
        if (store) store8<T>(reinterpret_cast<char*>(out) + cs.off0 + 4096 * i, q);
    };
    const int n_store_wave = WRITE_GX ? cs.n_out_wave : 0;

    int gslot = 0;
#pragma nounroll
    for (int k = 0; k < nf; ++k) {
        wait_vmcnt(issued - mark[0]);                             // my pieces of gy[k] and x[k] have landed
        __syncthreads();                                          // everyone's gy pieces have; step k-1 retired
        const char* xs = reinterpret_cast<const char*>(xring);
        Raw8 xv[ROUNDS];
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) xv[i] = lds_raw(reinterpret_cast<const float4*>(xs + cs.off0 + 4096 * i));
        xv[ROUNDS - 1] = lds_raw(reinterpret_cast<const float4*>(xs) + cs.xown);
        {
            int gs = gslot + DG; if (gs >= RG) gs -= RG;
            feed_g(k + DG, gs);
            feed_x(k + DX, 0);                                    // (the DMA waits for the LDS reads above)
            mark[0] = issued;
        }
        const float4* cur = gring + gslot * gslot_f4;
        T* out = out0 + (size_t)k * fstride;
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, xv[i], out, WRITE_GX, true);
        if (cs.tail_on) round(ROUNDS - 1, cur, xv[ROUNDS - 1], out, WRITE_GX && cs.tail_live, cs.tail_live);
        issued += n_store_wave;
        if (++gslot == RG) gslot = 0;
    }
    accH = sH; accW = sW;
    if (BN && WRITE_GX) { *accB1 = sB1; *accB2 = sB2; }
}

template <typename T, typename S, int ROUNDS, bool BN = false>
__global__ __launch_bounds__(kBlock) void k2d_raw16_backward(const T* __restrict__ gy, const T* __restrict__ x,
                                                             const S* __restrict__ shift, T* __restrict__ gx,
                                                             FDims fd, Fin2<S> fin, dma2d::BnFuse2 bn = dma2d::BnFuse2{}) {
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    __shared__ float red[4][kBlock / kWave];
    constexpr int ND = BN ? 4 : 2;
    const BDims& d = fd.b;
    if ((int)blockIdx.x >= fin.f.producers) {                         // row-sum + K9 inside the launch (rk_dma.hpp)
        if (threadIdx.x < kWave) {
            if (BN) dma2d::finalizer_wave2_bn(fin, (int)blockIdx.x - fin.f.producers, d.C, fd.ngroups * d.nbands, bn);
            else dma2d::finalizer_wave2(fin, (int)blockIdx.x - fin.f.producers, d.C, fd.ngroups * d.nbands);
        }
        return;
    }
    const int band = blockIdx.x % d.nbands, col = blockIdx.x / d.nbands;
    const int c = col % d.C, g = col / d.C;
    const float s0 = ld(shift + c), s1 = ld(shift + d.C + c);
    const int f0 = g * fd.FG;
    const int nf = min(fd.FG, fd.frames - f0);

    const int HW = d.H * d.W;
    const size_t fstride = (size_t)d.C * HW;
    const size_t base = ((size_t)f0 * d.C + c) * HW;
    const dma2d::IntegerPlan plan = dma2d::plan_walks(s0, s1);   // integer shifts: see rk2d_dma.hpp
    float4 bnp = make_float4(1.f, 0.f, 0.f, 1.f);
    if (BN) bnp = bn.abmi[c];
    float sumB1 = 0.f, sumB2 = 0.f;
    if (plan.separate_gx) {
        const Band b = make_band(d, band, plan.gH.fl);
#define RK_CALL(O) interp2_loop<T, ROUNDS, 1, O>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf)
        RK_OFF8_SWITCH(off8(plan.gW.fl), RK_CALL)
#undef RK_CALL
        if (BN) {
            // (a remainder below 1e-7 that is not 0: never a trained parameter.)  d(x) was written unmasked by the d(x)-only
            // walk: mask it in place -- every thread re-reads the cells the workgroup wrote -- and collect bn2's sums
            __syncthreads();
            for (int k = 0; k < nf; ++k) {
                T* gp = gx + base + (size_t)k * fstride + (size_t)b.out0 * 8;
                const T* zp = x + base + (size_t)k * fstride + (size_t)b.out0 * 8;
                for (int cell = threadIdx.x; cell < b.cells_out; cell += kBlock) {
                    const uint4 gw = reinterpret_cast<const uint4*>(gp)[cell], zw = reinterpret_cast<const uint4*>(zp)[cell];
                    const Raw8 gr{{gw.x, gw.y, gw.z, gw.w}}, zr{{zw.x, zw.y, zw.z, zw.w}};
                    float gv[8], zv[8];
                    widen8<T>(gr, gv); widen8<T>(zr, zv);
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const float act = round16<T>(fmaxf(fmaf(bnp.x, zv[m], bnp.y), 0.f));
                        gv[m] = act > 0.f ? gv[m] : 0.f;
                        sumB1 += gv[m];
                        sumB2 = fmaf(gv[m], (zv[m] - bnp.z) * bnp.w, sumB2);
                    }
                    store8<T>(reinterpret_cast<char*>(gp) + 16 * cell, gv);
                }
            }
        }
    }
    float sumH0 = 0.f, sumW0 = 0.f, sumH1 = 0.f, sumW2 = 0.f;
    if (!plan.separate_gx) {                                      // walk 0 with d(x): every ordinary channel ends here
        const Frac<float> fH = plan.sH, fW = plan.sW;
        const Band b = make_band(d, band, fH.fl);
        float aH = 0.f, aW = 0.f;
#define RK_CALL(O) backward2_loop<T, ROUNDS, O, true, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp, &sumB1, &sumB2)
        RK_OFF8_SWITCH(off8(fW.fl), RK_CALL)
#undef RK_CALL
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
#define RK_CALL(O) backward2_loop<T, ROUNDS, O, false, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp)
            RK_OFF8_SWITCH(off8(fW.fl), RK_CALL)
#undef RK_CALL
            if (walk == 0) { sumH0 = aH; sumW0 = aW; }
            else if (walk == 1) sumH1 = aH;
            else sumW2 = aW;
        }
    }
    float accH = plan.hint ? 0.5f * (sumH0 + sumH1) : sumH0;
    float accW = plan.wint ? 0.5f * (sumW0 + sumW2) : sumW0;

    accH = group_sum(accH, kBlock, red[0]);
    accW = group_sum(accW, kBlock, red[1]);
    if (BN) { sumB1 = group_sum(sumB1, kBlock, red[2]); sumB2 = group_sum(sumB2, kBlock, red[3]); }
    if (threadIdx.x == 0) {
        const int P = fd.ngroups * d.nbands;
        const size_t at = (size_t)c * ND * P + (size_t)g * d.nbands + band;
        fin_publish(fin.f, at, accH);
        fin_publish(fin.f, at + P, accW);
        if (BN) { fin_publish(fin.f, at + 2 * (size_t)P, sumB1); fin_publish(fin.f, at + 3 * (size_t)P, sumB2); }
    }
}
#undef RK_OFF8_SWITCH

// ---------------------------------------------------------------------------------------------
// Host side.  false = shape not handled here (stride / padding / W % 8 / RK_SHIFT_KERNELS): rk2d_stage.hpp takes it.
// In BDims "W4" is cells per row = W / 8 here.
// Frames per workgroup and band count do not matter here ([256,64,56,56] bf16, fwd / bwd us: 1 frame 37.6 / 57.3,
// 2: 37.1 / 56.3, 4: 37.4 / 57.5, 8: 37.6 / 58.1, 16: 36.9 / 57.1; two 28-row bands: the same within 1 us).
// frames per workgroup.  Measured on [256, 64, 56, 56] bf16 (round 4): forward 40.1 / 35.7 / 36.9 / 38.4 / 40.0 us with
// 1 / 2 / 3 / 4 / 8 frames, backward 65.9 / 59.8 / 57.7 / 59.0 / 60.0 / 58.5 us with 2 / 3 / 4 / 5 / 6 / 8
constexpr int kFramesRaw16 = 4;                   // backward
constexpr int kFramesRaw16Fwd = 2;                // forward / d(x) alone
inline bool make_fdims8(FDims& f, const Dims2& d, int frames_per_group) {
    const bool s1p0 = d.sH == 1 && d.sW == 1 && d.pH == 0 && d.pW == 0;
    if (!s1p0 || d.W % 8 != 0 || !streaming_kernels_on()) return false;
    BDims& b = f.b;
    b.N = 1; b.T = d.N; b.C = d.C; b.H = d.H; b.W = d.W; b.W4 = d.W / 8;
    if (!choose_bands(b)) return false;
    f.frames = d.N;
    f.FG = frames_per_group < d.N ? frames_per_group : d.N;
    f.ngroups = (d.N + f.FG - 1) / f.FG;
    return true;
}
inline int backward2_partials(const Dims2& d) {
    FDims f;
    return make_fdims8(f, d, kFramesRaw16) ? f.ngroups * f.b.nbands : 0;
}

template <typename T, bool NEGATE, typename S>
inline bool launch_interp2(const T* src, const S* shift, T* dst, const Dims2& d, hipStream_t stream) {
    constexpr int D = 2;
    FDims f;
    if (!make_fdims8(f, d, kFramesRaw16Fwd) || !aligned16(src) || !aligned16(dst)) return false;
    const size_t lds = interp_ring_bytes(f.b, D);
    if (lds > 64 * 1024) return false;
    const dim3 grid((unsigned)(f.ngroups * f.b.C * f.b.nbands)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_raw16_interp<T, S, NEGATE, 1, D>), grid, block, lds, stream, src, shift, dst, f); break;
        case 2: hipLaunchKernelGGL((k2d_raw16_interp<T, S, NEGATE, 2, D>), grid, block, lds, stream, src, shift, dst, f); break;
        case 3: hipLaunchKernelGGL((k2d_raw16_interp<T, S, NEGATE, 3, D>), grid, block, lds, stream, src, shift, dst, f); break;
        default: hipLaunchKernelGGL((k2d_raw16_interp<T, S, NEGATE, 4, D>), grid, block, lds, stream, src, shift, dst, f); break;
    }
    return true;
}

// d(x) + d(shift) (row-sum + K9 inside the launch: ws holds granules [C][2][P]); false = not handled here
template <typename T, typename S>
inline bool launch_backward2(const T* gy, const T* x, const S* shift, T* gx, S* gshift, void* ws, int normalize,
                             const Dims2& d, hipStream_t stream) {
    FDims f;
    if (!make_fdims8(f, d, kFramesRaw16) || !aligned16(gy) || !aligned16(x) || !aligned16(gx)) return false;
    const size_t lds = bwd_ring_bytes(f.b, 1, 1);
    if (lds > 64 * 1024) return false;
    Fin2<S> fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = f.ngroups * f.b.C * f.b.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    const dim3 grid((unsigned)(fin.f.producers + f.b.C)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_raw16_backward<T, S, 1>), grid, block, lds, stream, gy, x, shift, gx, f, fin); break;
        case 2: hipLaunchKernelGGL((k2d_raw16_backward<T, S, 2>), grid, block, lds, stream, gy, x, shift, gx, f, fin); break;
        case 3: hipLaunchKernelGGL((k2d_raw16_backward<T, S, 3>), grid, block, lds, stream, gy, x, shift, gx, f, fin); break;
        default: hipLaunchKernelGGL((k2d_raw16_backward<T, S, 4>), grid, block, lds, stream, gy, x, shift, gx, f, fin); break;
    }
    return true;
}

// training fusion: forward of relu(bn2(z)) (ab [2][C]) and its backward; false = not handled here
template <typename T, typename S>
inline bool launch_forward2_bn(const T* z, const float* ab, const S* shift, T* y, const Dims2& d, hipStream_t stream) {
    constexpr int D = 2;
    FDims f;
    if (!make_fdims8(f, d, kFramesRaw16Fwd) || !aligned16(z) || !aligned16(y)) return false;
    const size_t lds = interp_ring_bytes(f.b, D);
    if (lds > 64 * 1024) return false;
    const dim3 grid((unsigned)(f.ngroups * f.b.C * f.b.nbands)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_raw16_interp<T, S, false, 1, D, true>), grid, block, lds, stream, z, shift, y, f, ab); break;
        case 2: hipLaunchKernelGGL((k2d_raw16_interp<T, S, false, 2, D, true>), grid, block, lds, stream, z, shift, y, f, ab); break;
        case 3: hipLaunchKernelGGL((k2d_raw16_interp<T, S, false, 3, D, true>), grid, block, lds, stream, z, shift, y, f, ab); break;
        default: hipLaunchKernelGGL((k2d_raw16_interp<T, S, false, 4, D, true>), grid, block, lds, stream, z, shift, y, f, ab); break;
    }
    return true;
}
template <typename T, typename S>
inline bool launch_backward2_bn(const T* gy, const T* z, const S* shift, T* dz, S* gshift, void* ws, int normalize,
                                const dma2d::BnFuse2& bn, const Dims2& d, hipStream_t stream) {
    FDims f;
    if (!make_fdims8(f, d, kFramesRaw16) || !aligned16(gy) || !aligned16(z) || !aligned16(dz) || !aligned16(bn.abmi)) return false;
    const size_t lds = bwd_ring_bytes(f.b, 1, 1);
    if (lds > 64 * 1024) return false;
    Fin2<S> fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = f.ngroups * f.b.C * f.b.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    const dim3 grid((unsigned)(fin.f.producers + f.b.C)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_raw16_backward<T, S, 1, true>), grid, block, lds, stream, gy, z, shift, dz, f, fin, bn); break;
        case 2: hipLaunchKernelGGL((k2d_raw16_backward<T, S, 2, true>), grid, block, lds, stream, gy, z, shift, dz, f, fin, bn); break;
        case 3: hipLaunchKernelGGL((k2d_raw16_backward<T, S, 3, true>), grid, block, lds, stream, gy, z, shift, dz, f, fin, bn); break;
        default: hipLaunchKernelGGL((k2d_raw16_backward<T, S, 4, true>), grid, block, lds, stream, gy, z, shift, dz, f, fin, bn); break;
    }
    return true;
}

}  // namespace raw16
}  // namespace rk
