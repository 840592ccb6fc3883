// rk2d_stage.hpp -- RubiksShift2D streaming kernels for the 16-bit storage types (f16, bf16), stride 1 /
// pad 0, W % 4 == 0.  Same decomposition and the same fp32 arithmetic as rk2d_dma.hpp (workgroup = channel
// x row band x group of frames; cells of 4 elements; fp32 tap slots in LDS with a shared zero cell;
// compile-time tap offset), but the planes cannot be DMA'd: they have to be widened to fp32 on the way
// into LDS.  So the feed is register-staged -- each thread loads the 8-byte cells it is responsible for
// one plane ahead (two for gy), and deposits them, converted, into the other tap slot right after the
// step's single barrier -- and outputs are rounded once to the storage type and stored as 8-byte cells.
// x (needed at the thread's own cells only) never touches LDS.
//
// Results are bit-identical to the per-element kernels of rk2d_generic.hpp for these types: identical
// fp32 expression trees (contraction off), one rounding on store.
#pragma once
#include "rk2d_generic.hpp"
#include "rk2d_dma.hpp"

namespace rk {
namespace stage2d {

using namespace dma;
using dma2d::FDims;
using g2d::Dims2;

// 4 consecutive 16-bit elements <-> float4
template <typename T> struct Cell4;
template <> struct Cell4<__hip_bfloat16> {
    __device__ static __forceinline__ float4 widen(const uint2& r) {
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                           __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    }
    __device__ static __forceinline__ unsigned bits(float v) {
        return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(v));
    }
    __device__ static __forceinline__ uint2 narrow(float a, float b, float c, float d) {
        return make_uint2(bits(a) | (bits(b) << 16), bits(c) | (bits(d) << 16));
    }
};
template <> struct Cell4<__half> {
    __device__ static __forceinline__ float4 widen(const uint2& r) {
        const float2 lo = __half22float2(__builtin_bit_cast(__half2, r.x));
        const float2 hi = __half22float2(__builtin_bit_cast(__half2, r.y));
        return make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    __device__ static __forceinline__ unsigned bits(float v) {
        return (unsigned)__builtin_bit_cast(unsigned short, __float2half(materialise(v)));
    }
    __device__ static __forceinline__ uint2 narrow(float a, float b, float c, float d) {
        return make_uint2(bits(a) | (bits(b) << 16), bits(c) | (bits(d) << 16));
    }
};

__device__ __forceinline__ uint2 load_cell(const void* p) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void store_cell(void* p, const uint2& v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x2*>(p));
}

// ---------------------------------------------------------------------------------------------
// Forward (src = x) and d(x) alone (src = gy, negated shift).  sp / dp: frame 0 of the group, this channel.
// the activation of the training fusion (fused_bn.bn_relu_shift2d): max(a z + b, 0) as the storage type holds it
template <typename T> __device__ __forceinline__ float4 bn_act4(const float4& z, float a, float b) {
    return Cell4<T>::widen(Cell4<T>::narrow(fmaxf(fmaf(a, z.x, b), 0.f), fmaxf(fmaf(a, z.y, b), 0.f), fmaxf(fmaf(a, z.z, b), 0.f),
                                            fmaxf(fmaf(a, z.w, b), 0.f)));
}
// BN: src holds z = conv2's output and the shift applies to relu(bn2(z)): the cell is transformed on its way into LDS
template <typename T, int ROUNDS, int OFF, bool BN = false>
__device__ __forceinline__ void interp2_loop(const T* __restrict__ sp, T* __restrict__ dp, float4* ring,
                                             const BDims& d, const Band& b, const Frac<float>& fH,
                                             const Frac<float>& fW, size_t fstride, int nf, float bn_a = 1.f, float bn_b = 0.f) {
    const int slot_f4 = b.cells_in + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 4);
    init_tap_slots<ROUNDS>(ring, 2, slot_f4, b, cs);   // (first use of the ring in a kernel: no barrier needed before)

    const float rH = fH.r, rW = fW.r;
    const float uH = 1 - rH, uW = 1 - rW;
    // byte offset of this thread's cell `tid` (8 bytes per cell); round i adds 2048 i
    const char* src0 = reinterpret_cast<const char*>(sp + (ptrdiff_t)b.src0 * 4) + threadIdx.x * 8;
    char* out0 = reinterpret_cast<char*>(dp + (size_t)b.out0 * 4) + threadIdx.x * 8;
    const size_t fbytes = fstride * sizeof(T);

    uint2 stash[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) stash[i] = make_uint2(0u, 0u);
    auto fetch = [&](int k) {
        const char* p = src0 + (size_t)k * fbytes;
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i)
            if (cs.in_act[i]) stash[i] = load_cell(p + 2048 * i);
    };
    auto deposit = [&](float4* slot) {
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i)
            if (cs.in_act[i]) {
                const float4 v = Cell4<T>::widen(stash[i]);
                slot[threadIdx.x + kBlock * i] = BN ? bn_act4<T>(v, bn_a, bn_b) : v;
            }
    };
    auto round = [&](int i, const float4* cur, char* out, bool store) {
        const float4 qa0 = lds_b128(cur + cs.a0[i]), qa1 = lds_b128(cur + cs.a1[i]);
        const float4 qb0 = lds_b128(cur + cs.b0[i]), qb1 = lds_b128(cur + cs.b1[i]);
        float q[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)                               // interp2d, rubiks2d_kernels.cu:60-66
            q[m] = tap<OFF>(qa0, qa1, m) * uH * uW + tap<OFF>(qa0, qa1, m + 1) * uH * rW +
                   tap<OFF>(qb0, qb1, m) * rH * uW + tap<OFF>(qb0, qb1, m + 1) * rH * rW;
        if (store) store_cell(out + 2048 * i, Cell4<T>::narrow(q[0], q[1], q[2], q[3]));
    };

    fetch(0);
    deposit(ring);
    if (nf > 1) fetch(1);
#pragma nounroll
    for (int k = 0; k < nf; ++k) {
        __syncthreads();                                           // frame k deposited; frame k-1 retired
        float4* cur = ring + (k & 1) * slot_f4;
        if (k + 1 < nf) deposit(ring + ((k + 1) & 1) * slot_f4);
        if (k + 2 < nf) fetch(k + 2);
        char* out = out0 + (size_t)k * fbytes;
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, out, true);
        if (cs.tail_on) round(ROUNDS - 1, cur, out, cs.tail_live);
    }
}

template <typename T, typename S, bool NEGATE, int ROUNDS, bool BN = false>
__global__ __launch_bounds__(kBlock) void k2d_stage_interp(const T* __restrict__ src, const S* __restrict__ shift,
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
                reinterpret_cast<uint2*>(dp + (size_t)k * fstride)[b.out0 + cell] =
                    reinterpret_cast<const uint2*>(sp + (size_t)k * fstride)[b.out0 + cell];
        return;
    }
    const float bn_a = BN ? ab[c] : 1.f, bn_b = BN ? ab[d.C + c] : 0.f;
    switch (((fW.fl % 4) + 4) % 4) {                               // wave-uniform
        case 0: interp2_loop<T, ROUNDS, 0, BN>(sp, dp, ring, d, b, fH, fW, fstride, nf, bn_a, bn_b); break;
        case 1: interp2_loop<T, ROUNDS, 1, BN>(sp, dp, ring, d, b, fH, fW, fstride, nf, bn_a, bn_b); break;
        case 2: interp2_loop<T, ROUNDS, 2, BN>(sp, dp, ring, d, b, fH, fW, fstride, nf, bn_a, bn_b); break;
        default: interp2_loop<T, ROUNDS, 3, BN>(sp, dp, ring, d, b, fH, fW, fstride, nf, bn_a, bn_b); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) + d(shift) partials in one pass (adjoint form, see rk2d_dma.hpp).
// BN: x holds z = bn2's input: the activation is recomputed at the thread's own cell, d(x) leaves ReLU-masked and bn2's two sums
// ride along (rk2d_raw16.hpp, rk2d_tile.hpp: the same scheme)
template <typename T, int ROUNDS, int OFF, bool WRITE_GX, bool BN = false>
__device__ __forceinline__ void backward2_loop(const T* __restrict__ xp, const T* __restrict__ gp,
                                               T* __restrict__ op, float4* ring, const BDims& d, const Band& b,
                                               const Frac<float>& fH, const Frac<float>& fW, size_t fstride, int nf,
                                               float& accH, float& accW, float4 bnp = make_float4(1.f, 0.f, 0.f, 1.f),
                                               float* accB1 = nullptr, float* accB2 = nullptr) {
    float sB1 = 0.f, sB2 = 0.f;
    __syncthreads();                                              // a previous walk may still be reading the ring
    const int slot_f4 = b.cells_in + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 4);
    init_tap_slots<ROUNDS>(ring, 2, slot_f4, b, cs);

    const float rH = fH.r, rW = fW.r;
    const float uH = 1 - rH, uW = 1 - rW;
    const char* gsrc0 = reinterpret_cast<const char*>(gp + (ptrdiff_t)b.src0 * 4) + threadIdx.x * 8;
    // own cells; the last round's dead lanes re-read their round-0 cell and discard it
    const char* xsrc0 = reinterpret_cast<const char*>(xp + (size_t)b.out0 * 4) + threadIdx.x * 8;
    const int xtail = cs.tail_live ? 2048 * (ROUNDS - 1) : 0;
    char* out0 = reinterpret_cast<char*>(op + (size_t)b.out0 * 4) + threadIdx.x * 8;
    const size_t fbytes = fstride * sizeof(T);

    uint2 gstash[ROUNDS], xstash[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) gstash[i] = xstash[i] = make_uint2(0u, 0u);
    auto fetch_g = [&](int k) {
        const char* p = gsrc0 + (size_t)k * fbytes;
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i)
            if (cs.in_act[i]) gstash[i] = load_cell(p + 2048 * i);
    };
    auto fetch_x = [&](int k) {
        const char* p = xsrc0 + (size_t)k * fbytes;
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) xstash[i] = load_cell(p + 2048 * i);
        if (cs.tail_on) xstash[ROUNDS - 1] = load_cell(p + xtail);
    };
    auto deposit = [&](float4* slot) {
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i)
            if (cs.in_act[i]) slot[threadIdx.x + kBlock * i] = Cell4<T>::widen(gstash[i]);
    };

    float sH = 0.f, sW = 0.f;
    auto round = [&](int i, const float4* cur, const float4& xv4, char* out, bool live, bool alive) {
        const float4 qa0 = lds_b128(cur + cs.a0[i]), qa1 = lds_b128(cur + cs.a1[i]);
        const float4 qb0 = lds_b128(cur + cs.b0[i]), qb1 = lds_b128(cur + cs.b1[i]);
        float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
        const float zv[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
        if (BN) {                                                  // (a lane without a cell must contribute nothing: relu(b) is not 0)
            const float4 act = bn_act4<T>(xv4, bnp.x, bnp.y);
            xv[0] = alive ? act.x : 0.f; xv[1] = alive ? act.y : 0.f; xv[2] = alive ? act.z : 0.f; xv[3] = alive ? act.w : 0.f;
        }
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
        if (BN && WRITE_GX) {                                      // d(bn2's output): as stored, ReLU-masked; bn2's sums
            const float4 qr = Cell4<T>::widen(Cell4<T>::narrow(q[0], q[1], q[2], q[3]));
            const float r4[4] = {qr.x, qr.y, qr.z, qr.w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                q[m] = xv[m] > 0.f ? r4[m] : 0.f;
                sB1 += q[m];
                sB2 = fmaf(q[m], (zv[m] - bnp.z) * bnp.w, sB2);
            }
        }
        if (live) store_cell(out + 2048 * i, Cell4<T>::narrow(q[0], q[1], q[2], q[3]));
    };

    fetch_g(0);
    fetch_x(0);
    deposit(ring);
    if (nf > 1) fetch_g(1);
#pragma nounroll
    for (int k = 0; k < nf; ++k) {
        __syncthreads();                                           // gy[k] deposited; step k-1 retired
        float4* cur = ring + (k & 1) * slot_f4;
        float4 xv[ROUNDS];
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) xv[i] = Cell4<T>::widen(xstash[i]);
        if (!cs.tail_live) xv[ROUNDS - 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 1 < nf) { deposit(ring + ((k + 1) & 1) * slot_f4); fetch_x(k + 1); }
        if (k + 2 < nf) fetch_g(k + 2);
        char* out = out0 + (size_t)k * fbytes;
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, xv[i], out, WRITE_GX, true);
        if (cs.tail_on) round(ROUNDS - 1, cur, xv[ROUNDS - 1], out, WRITE_GX && cs.tail_live, cs.tail_live);
    }
    accH = sH; accW = sW;
    if (BN && WRITE_GX) { *accB1 = sB1; *accB2 = sB2; }
}

template <typename T, typename S, int ROUNDS, bool BN = false>
__global__ __launch_bounds__(kBlock) void k2d_stage_backward(const T* __restrict__ gy, const T* __restrict__ x,
                                                             const S* __restrict__ shift, T* __restrict__ gx,
                                                             FDims fd, Dims2 gd, dma2d::Fin2<S> fin,
                                                             dma2d::BnFuse2 bn = dma2d::BnFuse2{}) {
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
    (void)gd;

    const int HW = d.H * d.W;
    const size_t fstride = (size_t)d.C * HW;
    const size_t base = ((size_t)f0 * d.C + c) * HW;
    const dma2d::IntegerPlan plan = dma2d::plan_walks(s0, s1);   // integer shifts: see rk2d_dma.hpp
    if (plan.separate_gx) {
        const Band b = make_band(d, band, plan.gH.fl);
        switch (((plan.gW.fl % 4) + 4) % 4) {
            case 0: interp2_loop<T, ROUNDS, 0>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf); break;
            case 1: interp2_loop<T, ROUNDS, 1>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf); break;
            case 2: interp2_loop<T, ROUNDS, 2>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf); break;
            default: interp2_loop<T, ROUNDS, 3>(gy + base, gx + base, ring, d, b, plan.gH, plan.gW, fstride, nf); break;
        }
    }
    float4 bnp = make_float4(1.f, 0.f, 0.f, 1.f);
    if (BN) bnp = bn.abmi[c];
    float sumB1 = 0.f, sumB2 = 0.f;
    if (BN && plan.separate_gx) {
        // (a remainder below 1e-7 that is not 0: never a trained parameter.)  d(x) was written unmasked by the d(x)-only
        // walk: mask it in place and collect bn2's sums
        const Band b = make_band(d, band, plan.gH.fl);
        __syncthreads();
        for (int k = 0; k < nf; ++k) {
            char* gp = reinterpret_cast<char*>(gx + base + (size_t)k * fstride + (size_t)b.out0 * 4);
            const char* zp = reinterpret_cast<const char*>(x + base + (size_t)k * fstride + (size_t)b.out0 * 4);
            for (int cell = threadIdx.x; cell < b.cells_out; cell += kBlock) {
                const float4 g4 = Cell4<T>::widen(load_cell(gp + 8 * cell)), z4 = Cell4<T>::widen(load_cell(zp + 8 * cell));
                const float4 a4 = bn_act4<T>(z4, bnp.x, bnp.y);
                float gv[4] = {g4.x, g4.y, g4.z, g4.w};
                const float av[4] = {a4.x, a4.y, a4.z, a4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    gv[m] = av[m] > 0.f ? gv[m] : 0.f;
                    sumB1 += gv[m];
                    sumB2 = fmaf(gv[m], (zv[m] - bnp.z) * bnp.w, sumB2);
                }
                store_cell(gp + 8 * cell, Cell4<T>::narrow(gv[0], gv[1], gv[2], gv[3]));
            }
        }
    }
    float sumH0 = 0.f, sumW0 = 0.f, sumH1 = 0.f, sumW2 = 0.f;
    if (!plan.separate_gx) {                                      // walk 0 with d(x): every ordinary channel ends here
        const Frac<float> fH = plan.sH, fW = plan.sW;
        const Band b = make_band(d, band, fH.fl);
        float aH = 0.f, aW = 0.f;
        switch (((fW.fl % 4) + 4) % 4) {
            case 0: backward2_loop<T, ROUNDS, 0, true, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp, &sumB1, &sumB2); break;
            case 1: backward2_loop<T, ROUNDS, 1, true, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp, &sumB1, &sumB2); break;
            case 2: backward2_loop<T, ROUNDS, 2, true, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp, &sumB1, &sumB2); break;
            default: backward2_loop<T, ROUNDS, 3, true, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp, &sumB1, &sumB2); break;
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
                case 0: backward2_loop<T, ROUNDS, 0, false, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp); break;
                case 1: backward2_loop<T, ROUNDS, 1, false, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp); break;
                case 2: backward2_loop<T, ROUNDS, 2, false, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp); break;
                default: backward2_loop<T, ROUNDS, 3, false, BN>(x + base, gy + base, gx + base, ring, d, b, fH, fW, fstride, nf, aH, aW, bnp); break;
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
    if (BN) { sumB1 = group_sum(sumB1, kBlock, red[2]); sumB2 = group_sum(sumB2, kBlock, red[3]); }
    if (threadIdx.x == 0) {
        const int P = fd.ngroups * d.nbands;
        const size_t at = (size_t)c * ND * P + (size_t)g * d.nbands + band;
        fin_publish(fin.f, at, accH);
        fin_publish(fin.f, at + P, accW);
        if (BN) { fin_publish(fin.f, at + 2 * (size_t)P, sumB1); fin_publish(fin.f, at + 3 * (size_t)P, sumB2); }
    }
}

// ---------------------------------------------------------------------------------------------
inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }
inline size_t ring_bytes(const BDims& b) { return (size_t)2 * ((b.BH + 1) * b.W4 + 1) * 16; }

template <typename T, bool NEGATE, typename S>
inline bool launch_interp2(const T* src, const S* shift, T* dst, const Dims2& d, hipStream_t stream) {
    FDims f;
    if (!dma2d::make_fdims(f, d, dma2d::kFrames16) || !aligned8(src) || !aligned8(dst)) return false;
    const size_t lds = ring_bytes(f.b);
    const dim3 grid((unsigned)(f.ngroups * f.b.C * f.b.nbands)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_stage_interp<T, S, NEGATE, 1>), grid, block, lds, stream, src, shift, dst, f); break;
        case 2: hipLaunchKernelGGL((k2d_stage_interp<T, S, NEGATE, 2>), grid, block, lds, stream, src, shift, dst, f); break;
        case 3: hipLaunchKernelGGL((k2d_stage_interp<T, S, NEGATE, 3>), grid, block, lds, stream, src, shift, dst, f); break;
        default: hipLaunchKernelGGL((k2d_stage_interp<T, S, NEGATE, 4>), grid, block, lds, stream, src, shift, dst, f); break;
    }
    return true;
}

// d(x) + d(shift) (row-sum + K9 inside the launch: ws holds granules [C][2][P]); false = not handled here
template <typename T, typename S>
inline bool launch_backward2(const T* gy, const T* x, const S* shift, T* gx, S* gshift, void* ws, int normalize,
                             const Dims2& d, hipStream_t stream) {
    FDims f;
    if (!dma2d::make_fdims(f, d, dma2d::kFrames16) || !aligned8(gy) || !aligned8(x) || !aligned8(gx)) return false;
    const size_t lds = ring_bytes(f.b);
    dma2d::Fin2<S> fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = f.ngroups * f.b.C * f.b.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    const dim3 grid((unsigned)(fin.f.producers + f.b.C)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_stage_backward<T, S, 1>), grid, block, lds, stream, gy, x, shift, gx, f, d, fin); break;
        case 2: hipLaunchKernelGGL((k2d_stage_backward<T, S, 2>), grid, block, lds, stream, gy, x, shift, gx, f, d, fin); break;
        case 3: hipLaunchKernelGGL((k2d_stage_backward<T, S, 3>), grid, block, lds, stream, gy, x, shift, gx, f, d, fin); break;
        default: hipLaunchKernelGGL((k2d_stage_backward<T, S, 4>), grid, block, lds, stream, gy, x, shift, gx, f, d, fin); break;
    }
    return true;
}

// training fusion: forward of relu(bn2(z)) (ab [2][C]) and its backward; false = not handled here
template <typename T, typename S>
inline bool launch_forward2_bn(const T* z, const float* ab, const S* shift, T* y, const Dims2& d, hipStream_t stream) {
    FDims f;
    if (!dma2d::make_fdims(f, d, dma2d::kFrames16) || !aligned8(z) || !aligned8(y)) return false;
    const size_t lds = ring_bytes(f.b);
    const dim3 grid((unsigned)(f.ngroups * f.b.C * f.b.nbands)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_stage_interp<T, S, false, 1, true>), grid, block, lds, stream, z, shift, y, f, ab); break;
        case 2: hipLaunchKernelGGL((k2d_stage_interp<T, S, false, 2, true>), grid, block, lds, stream, z, shift, y, f, ab); break;
        case 3: hipLaunchKernelGGL((k2d_stage_interp<T, S, false, 3, true>), grid, block, lds, stream, z, shift, y, f, ab); break;
        default: hipLaunchKernelGGL((k2d_stage_interp<T, S, false, 4, true>), grid, block, lds, stream, z, shift, y, f, ab); break;
    }
    return true;
}
template <typename T, typename S>
inline bool launch_backward2_bn(const T* gy, const T* z, const S* shift, T* dz, S* gshift, void* ws, int normalize,
                                const dma2d::BnFuse2& bn, const Dims2& d, hipStream_t stream) {
    FDims f;
    if (!dma2d::make_fdims(f, d, dma2d::kFrames16) || !aligned8(gy) || !aligned8(z) || !aligned8(dz) || !aligned16(bn.abmi)) return false;
    const size_t lds = ring_bytes(f.b);
    dma2d::Fin2<S> fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = f.ngroups * f.b.C * f.b.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    const dim3 grid((unsigned)(fin.f.producers + f.b.C)), block(kBlock);
    switch (rounds_of(f.b)) {
        case 1: hipLaunchKernelGGL((k2d_stage_backward<T, S, 1, true>), grid, block, lds, stream, gy, z, shift, dz, f, d, fin, bn); break;
        case 2: hipLaunchKernelGGL((k2d_stage_backward<T, S, 2, true>), grid, block, lds, stream, gy, z, shift, dz, f, d, fin, bn); break;
        case 3: hipLaunchKernelGGL((k2d_stage_backward<T, S, 3, true>), grid, block, lds, stream, gy, z, shift, dz, f, d, fin, bn); break;
        default: hipLaunchKernelGGL((k2d_stage_backward<T, S, 4, true>), grid, block, lds, stream, gy, z, shift, dz, f, d, fin, bn); break;
    }
    return true;
}

}  // namespace stage2d
}  // namespace rk
