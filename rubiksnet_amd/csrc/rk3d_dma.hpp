// rk3d_dma.hpp -- RubiksShift3D streaming kernels fed by LDS-DMA (gfx950 global_load_lds_dwordx4),
// fp32, stride 1 / pad 0, W % 4 == 0.  These are the kernels the benchmark shape and the
// stride-1 layers of the networks run on (56x56, 28x28 whole planes; 112x112 in 4 row bands).
//
// Maths: the shift is constant per channel, so an output
// plane is a fixed 2-D translate-and-blend of two input planes and the H/W-interpolated field
// B(t) of an input plane is shared by the two outputs that touch it,
//     y[to] = (1-rT) B(to+flT) + rT B(to+flT+1),  B = (1-rH) lerpW(row) + rH lerpW(row+1),
// evaluated in the reference's own expression tree (rubiks3d_kernels.cu:193-203; contraction off)
// => bit-identical to the oracle.  The backward uses the adjoint form: only gy needs taps, x is
// read at the thread's own cells, d(x) and the d(shift) partials come out of ONE pass (12 B/elem).
//
// Work decomposition: one 256-thread workgroup owns one (n, c, row band) and walks t (T+1 steps).
// A band is BH output rows x W4 float4 "cells"; thread `tid` owns cells tid + 256 i.  The launcher
// picks equal bands with (BH+1) * W4 <= 1024, so rounds 0..ROUNDS-2 are full (no liveness tests)
// and only the last round is ragged.
//
// Data movement: each wave DMAs its own 1 KiB chunks of a plane's band straight into a ring of
// linear LDS slots, D planes ahead, with `global_load_lds_dwordx4 ... nt` (inline asm, M0 saved
// and restored in the same statement), and waits -- with a COUNTED s_waitcnt vmcnt(N), never a
// drain -- only for the plane it is about to read.  hipcc neither sees nor waits for these loads,
// so they stay in flight across the single barrier per step.  Bookkeeping per wave (all uniform):
// `issued` counts every VMEM instruction the wave has issued (DMA pieces + 16 B output stores),
// mark[j] is `issued` right after the DMAs feeding step k+j.  VMEM retires in order, so "plane k
// landed" <=> outstanding <= issued - mark[0].  `issued` is never over-counted (under-counting
// only makes the wait stricter).
//
// Tap slot layout: rows_out+1 input rows of the band (rows outside the plane are zero-filled once)
// x W4 float4 in flat order -- so the DMA destination is lane-linear -- then ONE zero float4 that
// every out-of-range column group is redirected to.  All LDS accesses are aligned b128.
// An out-of-range PLANE is a slot of zeros (plain LDS stores where its DMA would have been
// issued), so the step body carries no validity branches; the tap offset flW mod 4 is a template
// parameter (the kernel switches once into one of four copies of the loop).  The first version
// of the backward loop was instruction-issue bound: 308 of ~650 VALU instructions per step were
// v_mov from run-time tap selection, validity merges and masks.
//
// Loads and stores are non-temporal: every plane is read once by one CU and every output written
// once.  Measured on [32,8,64,56,56]: nt stores + nt loads 197 us fwd+bwd vs 216 us without.
#pragma once
#include "rk_dma.hpp"

namespace rk {
namespace dma3d {

using namespace dma;

__device__ __forceinline__ float tap4(const float4& v, int m) { return m == 0 ? v.x : m == 1 ? v.y : m == 2 ? v.z : v.w; }

// ---------------------------------------------------------------------------------------------
// Forward (NEGATE = false: src = x, dst = y) and d(x) alone (NEGATE = true: src = gy, dst = gx).
template <bool NEGATE, int ROUNDS, int D, int OFF, bool BN = false>
__device__ __forceinline__ void dma_interp_loop(const float* __restrict__ sp, float* __restrict__ dp, float4* ring,
                                                const BDims& d, const Band& b, const Frac<float>& fT,
                                                const Frac<float>& fH, const Frac<float>& fW, size_t tstride,
                                                float bn_a = 0.f, float bn_b = 0.f) {
    constexpr int R = D + 1;
    const int slot_f4 = b.cells_in + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 4);
    init_tap_slots<ROUNDS>(ring, R, slot_f4, b, cs);

    const float rT = fT.r, rH = fH.r, rW = fW.r;
    const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const unsigned slot_bytes = (unsigned)slot_f4 * 16u;
    const float* src0 = sp + (ptrdiff_t)b.src0 * 4;               // source of slot cell 0 at t = 0
    float* out0 = dp + (size_t)b.out0 * 4;

    float4 Bprev[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) Bprev[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int t_first = fT.fl, steps = d.T + 1;                   // plane of step k is t_first + k
    auto in_range = [&](int t) { return t >= 0 && t < d.T; };
    int issued = 0;
    auto feed = [&](int t, int s) {
        if (in_range(t)) {
            dma_taps<ROUNDS>(src0 + (ptrdiff_t)t * (ptrdiff_t)tstride, ring_addr + s * slot_bytes, cs);
            issued += cs.n_tap_wave;
        } else {
            zero_taps<ROUNDS>(ring + s * slot_f4, cs);
        }
    };
    int mark[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { feed(t_first + j, j); mark[j] = issued; }

    auto round = [&](int i, const float4* cur, float4* out, bool store) {
        const float4 qa0 = lds_b128(cur + cs.a0[i]), qa1 = lds_b128(cur + cs.a1[i]);
        const float4 qb0 = lds_b128(cur + cs.b0[i]), qb1 = lds_b128(cur + cs.b1[i]);
        float q[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            q[m] = uH * (tap<OFF>(qa0, qa1, m) * uW + tap<OFF>(qa0, qa1, m + 1) * rW) +
                   rH * (tap<OFF>(qb0, qb1, m) * uW + tap<OFF>(qb0, qb1, m + 1) * rW);
        if (store) {
            float4 o;
            o.x = uT * Bprev[i].x + rT * q[0];
            o.y = uT * Bprev[i].y + rT * q[1];
            o.z = uT * Bprev[i].z + rT * q[2];
            o.w = uT * Bprev[i].w + rT * q[3];
            stream_store(reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + cs.off0 + 4096 * i), o);
        }
        Bprev[i] = make_float4(q[0], q[1], q[2], q[3]);
    };

    int slot = 0;                                                  // slot of step k = k % R
#pragma nounroll
    for (int k = 0; k < steps; ++k) {
        wait_vmcnt(issued - mark[0]);                              // my pieces of plane k have landed
        if (BN && in_range(t_first + k)) bn_taps<ROUNDS>(ring + slot * slot_f4, cs, bn_a, bn_b);   // z -> relu(bn(z))
        __syncthreads();                                           // everyone's have; plane k-1 is retired
        {
            int sn = slot + D; if (sn >= R) sn -= R;               // = slot of plane k-1, free now
            feed(t_first + k + D, sn);
#pragma unroll
            for (int j = 0; j + 1 < D; ++j) mark[j] = mark[j + 1];
            mark[D - 1] = issued;
        }
        const float4* cur = ring + slot * slot_f4;
        const bool emit = k >= 1;                                  // output plane to = k - 1
        float4* out = reinterpret_cast<float4*>(out0 + (size_t)(emit ? k - 1 : 0) * tstride);
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, out, emit);
        if (cs.tail_on) round(ROUNDS - 1, cur, out, emit && cs.tail_live);
        if (emit) issued += cs.n_out_wave;
        if (++slot == R) slot = 0;
    }
}

template <bool NEGATE, int ROUNDS, int D, bool BN = false>
__global__ __launch_bounds__(kBlock) void k3d_dma_interp(const float* __restrict__ src,
                                                         const float* __restrict__ shift,
                                                         float* __restrict__ dst, BDims d,
                                                         const float4* __restrict__ abmi = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    const int band = blockIdx.x % d.nbands, col = blockIdx.x / d.nbands;
    const int c = col % d.C, n = col / d.C;
    float sT = shift[c], sH = shift[d.C + c], sW = shift[2 * d.C + c];
    if (NEGATE) { sT = -sT; sH = -sH; sW = -sW; }
    const Frac<float> fT = split_shift(sT), fH = split_shift(sH), fW = split_shift(sW);
    const int HW = d.H * d.W;
    const size_t tstride = (size_t)d.C * HW;
    const float* sp = src + ((size_t)n * d.T * d.C + c) * HW;
    float* dp = dst + ((size_t)n * d.T * d.C + c) * HW;
    const Band b = make_band(d, band, fH.fl);
    float bn_a = 0.f, bn_b = 0.f;
    if (BN) { const float4 pk = abmi[c]; bn_a = pk.x; bn_b = pk.y; }

    if (NEGATE && sT == 0 && sH == 0 && sW == 0) {                  // rubiks3d_kernels.cu:819-827: plain copy
        for (int t = 0; t < d.T; ++t)
            for (int cell = threadIdx.x; cell < b.cells_out; cell += kBlock)
                reinterpret_cast<float4*>(dp + (size_t)t * tstride)[b.out0 + cell] =
                    reinterpret_cast<const float4*>(sp + (size_t)t * tstride)[b.out0 + cell];
        return;
    }
    switch (((fW.fl % 4) + 4) % 4) {                                // wave-uniform
        case 0: dma_interp_loop<NEGATE, ROUNDS, D, 0, BN>(sp, dp, ring, d, b, fT, fH, fW, tstride, bn_a, bn_b); break;
        case 1: dma_interp_loop<NEGATE, ROUNDS, D, 1, BN>(sp, dp, ring, d, b, fT, fH, fW, tstride, bn_a, bn_b); break;
        case 2: dma_interp_loop<NEGATE, ROUNDS, D, 2, BN>(sp, dp, ring, d, b, fT, fH, fW, tstride, bn_a, bn_b); break;
        default: dma_interp_loop<NEGATE, ROUNDS, D, 3, BN>(sp, dp, ring, d, b, fT, fH, fW, tstride, bn_a, bn_b); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) + d(shift) partials (WRITE_GX) or the partials alone.  One partial per
// (n, c, band): part[c][3][P], P = N * nbands, p = n * nbands + band.
//   gy: ring of D+1 tap slots (plane tg needed whole for the taps while D more are in flight)
//   x : ring of D landing slots; a wave only reads back the cells it DMA'd itself (its own aligned
//       cells), into the 2-plane register window (x[to], x[to+1]) at the top of the step, which
//       frees the slot for the plane D steps ahead.
// Pairing of the fields of gy plane k with the x window (xa = x[k-1], xb = x[k]), as coefficients so that the SAME loop
// serves channels whose temporal shift is exactly integer:
//     gT += Q (cb xb + ca xa),   gH += QH (mb xb + ma xa),   gW += QW (mb xb + ma xa).
// Ordinary channel: (cb, ca, mb, ma) = (1, -1, 1-r'T, r'T) -- bit for bit the former x[k] - x[k-1] and fma(uT, xb, rT xa).
// Integer temporal shift (every channel of create_3d_from_2d(init_mode="tsm"), the reference's default,
// layer.py:137-141): the reference lowers the small temporal index by one (rubiks3d_kernels.cu:290-298) and uses that
// lowered plane, with weight 1 - rT = 1, in the H and W faces too (:362-431), which in this adjoint form reads
//     gT = sum_k <x[k], Q(k-1)> - <x[k-1], Q(k)>,   gH = sum_k <x[k-1], QH(k)>,   gW = sum_k <x[k-1], QW(k)>.
// Q(k-1) is the field the step before left in Qprev, so ONE walk does it: gT += Qprev (cp xb) + Q (ca xa) with
// (cb, ca, cp, mb, ma) = (0, -1, 1, 0, 1) against the ordinary channel's (1, -1, 0, 1-r'T, r'T).  Q(-1) -- gy plane
// fl'T - 1, paired with x[0] -- may be a real plane, so such a channel starts one plane earlier (T + 2 steps, x[-1] = 0):
// 1.1x the traffic of an ordinary channel.  (Round 2 walked the column twice, 2x: 221 us against 112 us at
// [32,8,64,56,56] with the tsm table.)  The walk loop sits INSIDE each tap-offset copy: around the switch over the
// copies the compiler hoists the set-up of all four out of it (256 VGPRs).

// QUANT (quantize = True): d(x) is the single nearest tap -- plane fl'T or fl'T + 1, row a or b, column m or m + 1 by
// "remainder >= 0.5" (rubiks3d_kernels.cu:819 ff. with the rounding of :76-93) -- which the walk already has at hand
// (the tap cells of the current plane, the previous plane's choice in Qprev); d(shift) is unchanged (K2 takes the
// fractional shift whatever quantize says).
// BN (training fusion, train_block.py): xp holds z = the input of relu(bn2(.)), not the activation.  The window keeps the
// raw z; the activation max(a z + b, 0) is recomputed where the d(shift) sums use it (a plane outside [0, T) is the ZERO
// activation, not max(b, 0): its (a, b) are 0 for that step), and the first half of bn2's backward happens on the way
// out: d(x) is masked with [a z + b > 0] and its sum(dz), sum(dz zhat), zhat = (z - mean) invstd, are reduced along with
// the d(shift) partials (accB1, accB2) -- k_bn_bwd_reduce's full pass over (d(a2), z) disappears.
template <int ROUNDS, bool WRITE_GX, int DG, int DX, int OFF, bool QUANT = false, bool BN = false>
__device__ __forceinline__ void dma_backward_loop(const float* __restrict__ xp, const float* __restrict__ gp,
                                                  float* __restrict__ op, float4* ring, const BDims& d,
                                                  const Band& b, const Frac<float>& fT, const Frac<float>& fH,
                                                  const Frac<float>& fW, size_t tstride, float& accT, float& accH,
                                                  float& accW, float4 bnp = make_float4(0.f, 0.f, 0.f, 0.f),
                                                  float* accB1 = nullptr, float* accB2 = nullptr) {
    static_assert(DG >= DX && DX >= 1, "gy runs at least as far ahead as x");
    constexpr int RG = DG + 1, RX = DX;
    const int gslot_f4 = b.cells_in + 1, xslot_f4 = b.cells_out + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 4);
    float4* const gring = ring;
    float4* const xring = ring + RG * gslot_f4;
    init_tap_slots<ROUNDS>(gring, RG, gslot_f4, b, cs);
    if (threadIdx.x < RX) xring[threadIdx.x * xslot_f4 + b.cells_out] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float rT = fT.r, rH = fH.r, rW = fW.r;
    const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const bool selT = !(rT < 0.5f), selH = !(rH < 0.5f), selW = !(rW < 0.5f);     // QUANT: wave-uniform
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(gring));
    const unsigned xaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(xring));
    const unsigned gslot_bytes = (unsigned)gslot_f4 * 16u, xslot_bytes = (unsigned)xslot_f4 * 16u;
    const float* gsrc0 = gp + (ptrdiff_t)b.src0 * 4;
    const float* xsrc0 = xp + (size_t)b.out0 * 4;
    float* out0 = WRITE_GX ? op + (size_t)b.out0 * 4 : nullptr;

    float sT = 0.f, sH = 0.f, sW = 0.f, sB1 = 0.f, sB2 = 0.f;
    int issued = 0;
    const bool t_integer = rT == 0;                               // wave-uniform
    const int early = t_integer ? 1 : 0;                         // integer temporal shift: one leading step for Q(-1)
    const float cb = t_integer ? 0.f : 1.f, ca = -1.f, cp = t_integer ? 1.f : 0.f;
    const float mb = t_integer ? 0.f : uT, ma = t_integer ? 1.f : rT;
    float4 xa[ROUNDS], xb[ROUNDS], Qprev[ROUNDS];
    float4 Qfield[(QUANT || !WRITE_GX) ? ROUNDS : 1];            // the previous plane's field when Qprev does not hold it
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) xa[i] = xb[i] = Qprev[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < ((QUANT || !WRITE_GX) ? ROUNDS : 1); ++i) Qfield[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float aA = 0.f, bA = 0.f, aB = 0.f, bB = 0.f;                   // BN: affine map of the planes in xa / xb (0: no plane)
    const bool bn_sums = BN;

    // step k: gy plane tg = t_first + k is in gy slot k % RG; to = k - 1 - early; x[k - early] (-> xb) is in x slot k % RX
    const int t_first = fT.fl - early, steps = d.T + 1 + early;
    auto in_range = [&](int t) { return t >= 0 && t < d.T; };
    auto feed = [&](int tg, int gs, int tx, int xs) {               // DMA when the plane exists, zeros when not
        if (in_range(tg)) {
            dma_taps<ROUNDS>(gsrc0 + (ptrdiff_t)tg * (ptrdiff_t)tstride, gaddr + gs * gslot_bytes, cs);
            issued += cs.n_tap_wave;
        } else {
            zero_taps<ROUNDS>(gring + gs * gslot_f4, cs);
        }
        if (in_range(tx)) {
            dma_own<ROUNDS>(xsrc0 + (ptrdiff_t)tx * (ptrdiff_t)tstride, xaddr + xs * xslot_bytes, cs);
            issued += cs.n_out_wave;
        } else {
            zero_own<ROUNDS>(xring + xs * xslot_f4, cs);
        }
    };
    // marks follow the stream that is issued last for a step (x, DX ahead); gy for the same step went out
    // earlier (DG >= DX) and VMEM retires in order
    int mark[DX];
#pragma unroll
    for (int j = 0; j < DG; ++j) {
        if (in_range(t_first + j)) {
            dma_taps<ROUNDS>(gsrc0 + (ptrdiff_t)(t_first + j) * (ptrdiff_t)tstride, gaddr + j * gslot_bytes, cs);
            issued += cs.n_tap_wave;
        } else {
            zero_taps<ROUNDS>(gring + j * gslot_f4, cs);
        }
        if (j < DX) {
            if (in_range(j - early)) {
                dma_own<ROUNDS>(xsrc0 + (ptrdiff_t)(j - early) * (ptrdiff_t)tstride, xaddr + j * xslot_bytes, cs);
                issued += cs.n_out_wave;
            } else {
                zero_own<ROUNDS>(xring + j * xslot_f4, cs);
            }
            mark[j] = issued;
        }
    }

    // one round of one step: fields of the gy plane at this thread's cell meet x[to], x[to+1]
    auto round = [&](int i, const float4* cur, float4* out, bool store) {
        const float4 qa0 = lds_b128(cur + cs.a0[i]), qa1 = lds_b128(cur + cs.a1[i]);
        const float4 qb0 = lds_b128(cur + cs.b0[i]), qb1 = lds_b128(cur + cs.b1[i]);
        float xav[4] = {xa[i].x, xa[i].y, xa[i].z, xa[i].w};
        float xbv[4] = {xb[i].x, xb[i].y, xb[i].z, xb[i].w};
        if (BN) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                xav[m] = fmaxf(fmaf(aA, xav[m], bA), 0.f);
                xbv[m] = fmaxf(fmaf(aB, xbv[m], bB), 0.f);
            }
        }
        float col[5], q[4];
#pragma unroll
        for (int m = 0; m < 5; ++m) col[m] = fmaf(uH, tap<OFF>(qa0, qa1, m), rH * tap<OFF>(qb0, qb1, m));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float la = tap<OFF>(qa0, qa1, m) * uW + tap<OFF>(qa0, qa1, m + 1) * rW;
            const float lb = tap<OFF>(qb0, qb1, m) * uW + tap<OFF>(qb0, qb1, m + 1) * rW;
            q[m] = uH * la + rH * lb;                             // the reference's tree, contraction off
            const float dx = fmaf(cb, xbv[m], ca * xav[m]);
            const float mx = fmaf(mb, xbv[m], ma * xav[m]);
            sT = fmaf(q[m], dx, sT);
            sT = fmaf(tap4((QUANT || !WRITE_GX) ? Qfield[i] : Qprev[i], m), cp * xbv[m], sT);   // integer temporal shift:
                                                                  // + <x[k], Q(k-1)> (cp = 0 for every other channel)
            sH = fmaf(la - lb, mx, sH);
            sW = fmaf(col[m] - col[m + 1], mx, sW);
        }
        if (WRITE_GX) {
            float nv[4];                                          // QUANT: the nearest tap of THIS plane
            if (QUANT) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float va = selW ? tap<OFF>(qa0, qa1, m + 1) : tap<OFF>(qa0, qa1, m);
                    const float vb = selW ? tap<OFF>(qb0, qb1, m + 1) : tap<OFF>(qb0, qb1, m);
                    nv[m] = selH ? vb : va;
                }
            }
            if (store) {
                float4 o;
                if (QUANT) {
                    o = selT ? make_float4(nv[0], nv[1], nv[2], nv[3]) : Qprev[i];
                } else {
                    o.x = uT * Qprev[i].x + rT * q[0];
                    o.y = uT * Qprev[i].y + rT * q[1];
                    o.z = uT * Qprev[i].z + rT * q[2];
                    o.w = uT * Qprev[i].w + rT * q[3];
                }
                if (BN) {                                         // the output plane is xa's: mask + bn2's reduction sums
                    const float zv[4] = {xa[i].x, xa[i].y, xa[i].z, xa[i].w};
                    float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        ov[m] = xav[m] > 0.f ? ov[m] : 0.f;
                        if (bn_sums) {
                            sB1 += ov[m];
                            sB2 = fmaf(ov[m], (zv[m] - bnp.z) * bnp.w, sB2);
                        }
                    }
                    o = make_float4(ov[0], ov[1], ov[2], ov[3]);
                }
                stream_store(reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + cs.off0 + 4096 * i), o);
            }
            Qprev[i] = QUANT ? make_float4(nv[0], nv[1], nv[2], nv[3]) : make_float4(q[0], q[1], q[2], q[3]);
        }
        if (QUANT || !WRITE_GX) Qfield[i] = make_float4(q[0], q[1], q[2], q[3]);   // (else Qprev is the field itself)
    };

    int gslot = 0, xslot = 0;
    // one step; EMIT is a compile-time flag (step 0 produces no output plane): a run-time flag here costs
    // ~55 VGPRs in hipcc's allocation of this loop
    auto step = [&](int k, auto emit_tag) {
        constexpr bool EMIT = decltype(emit_tag)::value;
        wait_vmcnt(issued - mark[0]);                             // my pieces of gy(tg) and x[k] have landed
        __syncthreads();                                          // everyone's gy pieces have; step k-1 retired
        const char* xs = reinterpret_cast<const char*>(xring + xslot * xslot_f4);
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) {                    // window: x[k-1], x[k]
            xa[i] = xb[i];
            xb[i] = *reinterpret_cast<const float4*>(xs + cs.off0 + 4096 * i);
        }
        xa[ROUNDS - 1] = xb[ROUNDS - 1];
        xb[ROUNDS - 1] = reinterpret_cast<const float4*>(xs)[cs.xown];
        if (BN) {                                                 // x[k] exists for k < T; the window slides
            aA = aB; bA = bB;
            aB = in_range(k - early) ? bnp.x : 0.f;
            bB = in_range(k - early) ? bnp.y : 0.f;
        }
        {
            int gs = gslot + DG; if (gs >= RG) gs -= RG;          // gy slot of plane k-1: free now
            feed(t_first + k + DG, gs, k + DX - early, xslot);    // (the DMA waits for the LDS reads above)
#pragma unroll
            for (int j = 0; j + 1 < DX; ++j) mark[j] = mark[j + 1];
            mark[DX - 1] = issued;
        }
        const float4* cur = gring + gslot * gslot_f4;
        constexpr bool emit = WRITE_GX && EMIT;                   // output plane to = k - 1 - early
        float4* out = reinterpret_cast<float4*>(out0 + (size_t)(emit ? k - 1 - early : 0) * tstride);
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, out, emit);
        if (cs.tail_on) round(ROUNDS - 1, cur, out, emit && cs.tail_live);
        if (emit) issued += cs.n_out_wave;
        if (++gslot == RG) gslot = 0;
        if (++xslot == RX) xslot = 0;
    };
    step(0, std::false_type{});
    if (early) step(1, std::false_type{});
#pragma nounroll
    for (int k = 1 + early; k < steps; ++k) step(k, std::true_type{});
    accT = sT; accH = sH; accW = sW;
    if (BN) { *accB1 = sB1; *accB2 = sB2; }
}

// Row-sum + K5 inside the backward launch (FUSED): rk_dma.hpp, "Row-sum of the d(shift) partials INSIDE the backward
// launch".  K5 = rubiks3d_kernels.cu:932-960.
struct Fin3 {
    Fin f;
    float* gshift;                // [3][C]
    int normalize;
    float t_factor;
};
// training fusion (BN above): what the shift backward needs of bn2 and where bn2's backward constants go
struct BnFuse {
    const float4* abmi;           // [C] (a, b, mean, invstd)
    float* k12;                   // [2][C]: sum(dz) / count, sum(dz zhat) / count
    float* dgamma; float* dbeta;  // [C]
    float inv_count;              // 1 / (N T H W)
};
template <int D>
__device__ __forceinline__ void finalizer_wave(const Fin3& fin, int c, int C, int P, const BnFuse& bn = BnFuse{}, int count = -1) {
    double s[D];
    const bool ok = fin_collect<D>(fin.f, c, P, s, count);
    if (threadIdx.x == 0) {
        if constexpr (D == 5) {
            const float nanv = __uint_as_float(0x7fc00000u);
            bn.dbeta[c] = ok ? (float)s[3] : nanv;
            bn.dgamma[c] = ok ? (float)s[4] : nanv;
            bn.k12[c] = ok ? (float)(s[3] * (double)bn.inv_count) : nanv;
            bn.k12[C + c] = ok ? (float)(s[4] * (double)bn.inv_count) : nanv;
        }
        float gT = (float)s[0], gH = (float)s[1], gW = (float)s[2];
        if (fin.normalize) {
            float a, b, w;
            if (fin.t_factor < 0) { a = gT; b = 0; w = 0; }
            else { a = gT * fin.t_factor; b = gH; w = gW; }
            const float mag = sqrtf(a * a + b * b + w * w);
            if (mag > 0) { gT = a / mag; gH = b / mag; gW = w / mag; }
        }
        if (!ok) gT = gH = gW = __uint_as_float(0x7fc00000u);
        fin.gshift[c] = gT;
        fin.gshift[C + c] = gH;
        fin.gshift[2 * C + c] = gW;
    }
}

// (forcing <= 128 VGPRs with __launch_bounds__(256, 4) on an earlier version spilled and ran 13% slower)
template <int ROUNDS, bool WRITE_GX, int DG, int DX, bool FUSED, bool QUANT = false, bool BN = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3))) void k3d_dma_backward(const float* __restrict__ x,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ gy,
                                                           float* __restrict__ gx,
                                                           float* __restrict__ part, BDims d, Dims3 gd, Fin3 fin,
                                                           BnFuse bn = BnFuse{}) {
    constexpr int ND = BN ? 5 : 3;                                  // partials per (channel, column-band)
    if (FUSED && (int)blockIdx.x >= fin.f.producers) {
        if (threadIdx.x < kWave) finalizer_wave<ND>(fin, (int)blockIdx.x - fin.f.producers, d.C, d.N * d.nbands, bn);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    __shared__ float red[ND][kBlock / kWave];
    // Two bands per plane: workgroups b and b + 8 (same XCD under the round-robin dispatch, launched together) take the two
    // bands of one plane, so the halo row of gy they share is an L2 hit for the second one instead of a second HBM read.
    int bid = (int)blockIdx.x;
    const int nprod = FUSED ? fin.f.producers : (int)gridDim.x;
    if (d.nbands == 2 && (bid | 15) < nprod) bid = (bid & ~15) + ((bid & 7) << 1) + ((bid >> 3) & 1);
    const int band = bid % d.nbands, col = bid / d.nbands;
    const int c = col % d.C, n = col / d.C;
    const float s0 = shift[c], s1 = shift[d.C + c], s2 = shift[2 * d.C + c];
    float accT = 0.f, accH = 0.f, accW = 0.f, accB1 = 0.f, accB2 = 0.f;
    float4 bnp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BN) bnp = bn.abmi[c];

    if (split_shift(s1).r == 0 || split_shift(s2).r == 0) {
        // exactly-integer H or W component (lowered-index quirk / zero-shift copy branch): rare, per element.
        // Band 0 does the whole column; the other bands contribute zero partials.
        if (band == 0) {
            if (WRITE_GX)
                for (int t = 0; t < d.T; ++t)
                    backward_input_plane<float, QUANT>(shift, gy, gx, gd, n, t, c, threadIdx.x, kBlock);
            if (BN) {
                // x holds z: d(shift) from relu(bn(z)) evaluated per tap, then mask d(x) (each thread re-reads the
                // elements it wrote itself) and bn2's reduction sums
                const BnAct act{bnp.x, bnp.y};
                for (int to = 0; to < d.T; ++to)
                    shift_grad_plane<float>(x, shift, gy, gd, n, to, c, threadIdx.x, kBlock, accT, accH, accW, act);
                const int HW = d.H * d.W;
                for (int t = 0; t < d.T; ++t) {
                    const size_t base = (((size_t)n * d.T + t) * d.C + c) * HW;
                    for (int e = threadIdx.x; e < HW; e += kBlock) {
                        const float zv = x[base + e];
                        const float dz = fmaf(bnp.x, zv, bnp.y) > 0.f ? gx[base + e] : 0.f;
                        gx[base + e] = dz;
                        accB1 += dz;
                        accB2 = fmaf(dz, (zv - bnp.z) * bnp.w, accB2);
                    }
                }
            } else {
            for (int to = 0; to < d.T; ++to)
                shift_grad_plane<float>(x, shift, gy, gd, n, to, c, threadIdx.x, kBlock, accT, accH, accW);
            }
        }
    } else {
        const Frac<float> fT = split_shift(-s0), fH = split_shift(-s1), fW = split_shift(-s2);   // fl', r'
        const int HW = d.H * d.W;
        const size_t tstride = (size_t)d.C * HW;
        const float* xp = x + ((size_t)n * d.T * d.C + c) * HW;
        const float* gp = gy + ((size_t)n * d.T * d.C + c) * HW;
        float* op = WRITE_GX ? gx + ((size_t)n * d.T * d.C + c) * HW : nullptr;
        const Band b = make_band(d, band, fH.fl);
        const int off = ((fW.fl % 4) + 4) % 4;
        switch (off) {   // wave-uniform; one specialised copy of the loop per tap offset
            case 0: dma_backward_loop<ROUNDS, WRITE_GX, DG, DX, 0, QUANT, BN>(xp, gp, op, ring, d, b, fT, fH, fW, tstride, accT, accH, accW, bnp, &accB1, &accB2); break;
            case 1: dma_backward_loop<ROUNDS, WRITE_GX, DG, DX, 1, QUANT, BN>(xp, gp, op, ring, d, b, fT, fH, fW, tstride, accT, accH, accW, bnp, &accB1, &accB2); break;
            case 2: dma_backward_loop<ROUNDS, WRITE_GX, DG, DX, 2, QUANT, BN>(xp, gp, op, ring, d, b, fT, fH, fW, tstride, accT, accH, accW, bnp, &accB1, &accB2); break;
            default: dma_backward_loop<ROUNDS, WRITE_GX, DG, DX, 3, QUANT, BN>(xp, gp, op, ring, d, b, fT, fH, fW, tstride, accT, accH, accW, bnp, &accB1, &accB2); break;
        }
    }

    accT = group_sum(accT, kBlock, red[0]);
    accH = group_sum(accH, kBlock, red[1]);
    accW = group_sum(accW, kBlock, red[2]);
    if (BN) {
        accB1 = group_sum(accB1, kBlock, red[ND - 2]);
        accB2 = group_sum(accB2, kBlock, red[ND - 1]);
    }
    if (threadIdx.x == 0) {
        const int P = d.N * d.nbands;
        const size_t at = (size_t)c * ND * P + (size_t)n * d.nbands + band;
        if (FUSED) {
            fin_publish(fin.f, at, accT);
            fin_publish(fin.f, at + P, accH);
            fin_publish(fin.f, at + 2 * P, accW);
            if (BN) {
                fin_publish(fin.f, at + 3 * P, accB1);
                fin_publish(fin.f, at + 4 * P, accB2);
            }
        } else {
            part[at] = accT;
            part[at + P] = accH;
            part[at + 2 * P] = accW;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host side: launchers.
// false = shape not handled by the DMA kernels
inline bool make_bdims(BDims& b, const Dims3& d, bool backward = false) {
    const bool s1p0 = d.sT == 1 && d.sH == 1 && d.sW == 1 && d.pT == 0 && d.pH == 0 && d.pW == 0;
    if (!s1p0 || d.W % 4 != 0 || d.W < 4 || !streaming_kernels_on()) return false;
    b.N = d.N; b.T = d.T; b.C = d.C; b.H = d.H; b.W = d.W; b.W4 = d.W / 4;
    return choose_bands(b, backward);
}

template <bool NEGATE, int D, bool BN = false>
inline void launch_interp_d(const float* src, const float* shift, float* dst, const BDims& b, hipStream_t stream,
                            const float4* abmi = nullptr) {
    const size_t lds = interp_ring_bytes(b, D);
    const dim3 grid((unsigned)(b.N * b.C * b.nbands)), block(kBlock);
    switch (rounds_of(b)) {
        case 1: hipLaunchKernelGGL((k3d_dma_interp<NEGATE, 1, D, BN>), grid, block, lds, stream, src, shift, dst, b, abmi); break;
        case 2: hipLaunchKernelGGL((k3d_dma_interp<NEGATE, 2, D, BN>), grid, block, lds, stream, src, shift, dst, b, abmi); break;
        case 3: hipLaunchKernelGGL((k3d_dma_interp<NEGATE, 3, D, BN>), grid, block, lds, stream, src, shift, dst, b, abmi); break;
        default: hipLaunchKernelGGL((k3d_dma_interp<NEGATE, 4, D, BN>), grid, block, lds, stream, src, shift, dst, b, abmi); break;
    }
}

// forward / d(x)-only; false = not handled here.  2 planes in flight per column (1 and 3 measured within 1 %).
template <bool NEGATE>
inline bool launch_interp(const float* src, const float* shift, float* dst, const Dims3& d, hipStream_t stream) {
    constexpr int kDepth = 2;
    BDims b;
    if (!make_bdims(b, d) || !aligned16(src) || !aligned16(dst)) return false;
    if (interp_ring_bytes(b, kDepth) > 64 * 1024) return false;
    launch_interp_d<NEGATE, kDepth>(src, shift, dst, b, stream);
    return true;
}
// forward of relu(bn(z)) (train_block.py): abmi [C] = (a, b, mean, invstd)
inline bool launch_forward_bn(const float* z, const float* shift, float* y, const float4* abmi, const Dims3& d,
                              hipStream_t stream) {
    constexpr int kDepth = 2;
    BDims b;
    if (!make_bdims(b, d) || !aligned16(z) || !aligned16(y) || !aligned16(abmi)) return false;
    if (interp_ring_bytes(b, kDepth) > 64 * 1024) return false;
    launch_interp_d<false, kDepth, true>(z, shift, y, b, stream, abmi);
    return true;
}

template <bool WRITE_GX, int DG, int DX, bool FUSED, bool QUANT = false, bool BN = false>
inline void launch_bwd_d(const float* x, const float* shift, const float* gy, float* gx, float* ws, const BDims& b,
                         const Dims3& d, const Fin3& fin, hipStream_t stream, const BnFuse& bn = BnFuse{}) {
    const size_t lds = bwd_ring_bytes(b, DG, DX);
    const dim3 grid((unsigned)(b.N * b.C * b.nbands + (FUSED ? b.C : 0))), block(kBlock);
    switch (rounds_of(b)) {
        case 1: hipLaunchKernelGGL((k3d_dma_backward<1, WRITE_GX, DG, DX, FUSED, QUANT, BN>), grid, block, lds, stream, x, shift, gy, gx, ws, b, d, fin, bn); break;
        case 2: hipLaunchKernelGGL((k3d_dma_backward<2, WRITE_GX, DG, DX, FUSED, QUANT, BN>), grid, block, lds, stream, x, shift, gy, gx, ws, b, d, fin, bn); break;
        case 3: hipLaunchKernelGGL((k3d_dma_backward<3, WRITE_GX, DG, DX, FUSED, QUANT, BN>), grid, block, lds, stream, x, shift, gy, gx, ws, b, d, fin, bn); break;
        default: hipLaunchKernelGGL((k3d_dma_backward<4, WRITE_GX, DG, DX, FUSED, QUANT, BN>), grid, block, lds, stream, x, shift, gy, gx, ws, b, d, fin, bn); break;
    }
}

// d(shift) (+ d(x) when gx != nullptr).  One gy plane and one x plane in flight (2 / 2 measured within 1 %: the
// memory system, not latency, bounds it).  gshift != nullptr: row-sum + K5 fused into the launch (ws holds 8-byte
// granules [C][3][P]); gshift == nullptr: plain float partials ws[C][3][P] for a separate finalize (two-phase API).
// Returns P (0 = not handled here).
// quant: quantize = True (only the fused one-call form with both gradients is built for it)
inline int launch_bwd(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws,
                      const Dims3& d, int normalize, float t_factor, hipStream_t stream, bool quant = false) {
    BDims b;
    if (!make_bdims(b, d, true) || !aligned16(x) || !aligned16(gy) || (gx && !aligned16(gx))) return 0;
    if (bwd_ring_bytes(b, 1, 1) > 64 * 1024) return 0;
    Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = b.N * b.C * b.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
    if (quant) {
        if (!(gshift && gx)) return 0;
        launch_bwd_d<true, 1, 1, true, true>(x, shift, gy, gx, ws, b, d, fin, stream);
    } else if (gshift) {
        if (gx) launch_bwd_d<true, 1, 1, true>(x, shift, gy, gx, ws, b, d, fin, stream);
        else launch_bwd_d<false, 1, 1, true>(x, shift, gy, gx, ws, b, d, fin, stream);
    } else {
        if (gx) launch_bwd_d<true, 1, 1, false>(x, shift, gy, gx, ws, b, d, fin, stream);
        else launch_bwd_d<false, 1, 1, false>(x, shift, gy, gx, ws, b, d, fin, stream);
    }
    return b.N * b.nbands;
}

// training fusion: x = z (bn2's input), gx <- masked d(z-activation), + bn2's backward constants.  ws: granule pairs
// [C][5][P].  false = shape not handled here.
inline bool launch_bwd_bn(const float* z, const float* shift, const float* gy, float* gx, float* gshift, float* ws,
                          const Dims3& d, int normalize, float t_factor, int quantize, const BnFuse& bn, hipStream_t stream) {
    BDims b;
    if (!make_bdims(b, d, true) || !aligned16(z) || !aligned16(gy) || !aligned16(gx) || !aligned16(bn.abmi)) return false;
    if (bwd_ring_bytes(b, 1, 1) > 64 * 1024) return false;
    Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = b.N * b.C * b.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
    if (quantize) launch_bwd_d<true, 1, 1, true, true, true>(z, shift, gy, gx, ws, b, d, fin, stream, bn);
    else launch_bwd_d<true, 1, 1, true, false, true>(z, shift, gy, gx, ws, b, d, fin, stream, bn);
    return true;
}
inline int bwd_bn_partials(const Dims3& d) {
    BDims b;
    return make_bdims(b, d, true) ? b.N * b.nbands : 0;
}

}  // namespace dma3d
}  // namespace rk
