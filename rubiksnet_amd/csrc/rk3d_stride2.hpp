// rk3d_stride2.hpp -- RubiksShift3D streaming kernels for the downsampling layers: stride (1,2,2), pad 0, fp32,
// W % 4 == 0 and Wo % 4 == 0 (112x112 -> 56x56 and 56x56 -> 28x28 in the networks: SURVEY Appendix B; the 28 -> 14
// and 14 -> 7 layers stay on rk3d_column.hpp).  Reference: K1 rubiks3d_kernels.cu:15-205 (forward), K2 :218-452
// (d(shift)), K4 :455-723 (d(x) through the stride un-mapping).
//
// The 2x2 tap windows of a stride-2 layer TILE the input plane (offset (flH, flW)): every x element is a tap of
// exactly one output per source plane.  So the walk of rk3d_dma.hpp carries over with a different cell: a workgroup
// owns (n, c, band of BHo output rows), DMAs the 2*BHo source rows of each plane into an LDS slot (no halo row),
// and a thread owns 4 consecutive outputs = a 2 x 8 window of x, read as 2 x 3 aligned b128 cells with the compile-
// time offset OFF = flW mod 4.  The blend over T is the register recurrence of the stride-1 kernels
// (y[to] = (1-rT) S(to+flT) + rT S(to+flT+1), S = the reference's bilinear tree on one plane), so x is read once.
// Expression trees are the reference's with contraction off => y is bit-identical to the oracle.
#pragma once
#include "rk3d_dma.hpp"

namespace rk {
namespace s2 {

using namespace dma;

struct SDims {
    int N, T, C, H, W, W4;       // input
    int Ho, Wo, Wo4;             // output
    int BHo, nbands;             // output rows per band (Ho % BHo == 0)
};

// per-workgroup band of 2*BHo source rows starting at r0 = 2*band*BHo + flH
struct SBand {
    int cells_in;                // 2*BHo*W4 slot cells (+1 zero cell behind them)
    int cells_out;               // BHo*Wo4 output cells
    int src0;                    // float4 index (may be negative) of slot cell 0 inside a source plane
    int s_lo, s_hi;              // slot cells [s_lo, s_hi) hold rows inside the plane
    int out0;                    // float4 index of the band's first output cell inside an output plane
};
__device__ __forceinline__ SBand make_sband(const SDims& d, int band, int flH) {
    SBand b;
    const int rows = 2 * d.BHo, r0 = 2 * band * d.BHo + flH;
    b.cells_in = rows * d.W4;
    b.cells_out = d.BHo * d.Wo4;
    b.src0 = r0 * d.W4;
    int j_lo = r0 < 0 ? -r0 : 0;
    j_lo = j_lo > rows ? rows : j_lo;
    int j_hi = d.H - r0;
    j_hi = j_hi < 0 ? 0 : (j_hi > rows ? rows : j_hi);
    b.s_lo = j_lo * d.W4;
    b.s_hi = j_hi > j_lo ? j_hi * d.W4 : b.s_lo;
    b.out0 = band * d.BHo * d.Wo4;
    return b;
}

// the DMA side of a thread (the fields of BCells that dma_taps / zero_taps use), up to 4 rounds of 256 cells
template <int DR> __device__ __forceinline__ void make_feed(BCells<DR>& cs, const SBand& b) {
    cs.off0 = (int)threadIdx.x * 16;
    int n_tap = 0;
#pragma unroll
    for (int i = 0; i < DR; ++i) {
        const int o = (int)threadIdx.x + kBlock * i;
        cs.in_act[i] = o >= b.s_lo && o < b.s_hi;
        n_tap += (__ballot(cs.in_act[i]) != 0ull) ? 1 : 0;
    }
    cs.n_tap_wave = __builtin_amdgcn_readfirstlane(n_tap);
}
template <int DR> __device__ __forceinline__ void init_slots(float4* ring, int nslots, int slot_f4, const SBand& b,
                                                             const BCells<DR>& cs) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nslots; ++s) {
        float4* slot = ring + s * slot_f4;
        if (threadIdx.x == 0) slot[b.cells_in] = z;
#pragma unroll
        for (int i = 0; i < DR; ++i) {
            const int o = (int)threadIdx.x + kBlock * i;
            if (o < b.cells_in && !cs.in_act[i]) slot[o] = z;
        }
    }
}

// element OFF + e (e = 0..7) of the aligned 12-float window (q0, q1, q2); constant index after unrolling
template <int OFF> __device__ __forceinline__ float tap12(const float4& q0, const float4& q1, const float4& q2, int e) {
    const int j = OFF + e;   // 0..10
    const float4& q = j < 4 ? q0 : (j < 8 ? q1 : q2);
    const int r = j & 3;
    return r == 0 ? q.x : r == 1 ? q.y : r == 2 ? q.z : q.w;
}

// the thread's window: slot cell indices of rows A (2j) and B (2j+1), 3 cells each (zero cell when outside)
struct Window { int a[3], b[3]; bool live; };
__device__ __forceinline__ Window make_window(const SDims& d, const SBand& b, int group_shift) {
    Window w;
    const int oc = (int)threadIdx.x;
    w.live = oc < b.cells_out;
    const int o = w.live ? oc : 0;
    const int j = o / d.Wo4, wo4 = o - j * d.Wo4;
    const int g0 = 2 * wo4 + group_shift;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int g = g0 + k;
        const bool ok = w.live && g >= 0 && g < d.W4;
        w.a[k] = ok ? (2 * j) * d.W4 + g : b.cells_in;
        w.b[k] = ok ? (2 * j + 1) * d.W4 + g : b.cells_in;
    }
    return w;
}

// ---------------------------------------------------------------------------------------------
// Forward.
// BN (training fusion, train_block.py): xp holds z, the shift applies to max(a z + b, 0); the landed pieces of a plane are
// normalised in the slot by the wave that DMA'd them (rk_dma.hpp bn_taps).
template <int DR, int D, int OFF, bool BN = false>
__device__ __forceinline__ void forward_loop(const float* __restrict__ xp, float* __restrict__ yp, float4* ring,
                                             const SDims& d, const SBand& b, const Frac<float>& fT,
                                             const Frac<float>& fH, const Frac<float>& fW, size_t tstride_in,
                                             size_t tstride_out, float bn_a = 0.f, float bn_b = 0.f) {
    constexpr int R = D + 1;
    const int slot_f4 = b.cells_in + 1;
    BCells<DR> cs;
    make_feed<DR>(cs, b);
    init_slots<DR>(ring, R, slot_f4, b, cs);
    const Window w = make_window(d, b, (fW.fl - OFF) / 4);

    const float rT = fT.r, rH = fH.r, rW = fW.r;
    const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const unsigned slot_bytes = (unsigned)slot_f4 * 16u;
    const float* src0 = xp + (ptrdiff_t)b.src0 * 4;
    float4* out0 = reinterpret_cast<float4*>(yp) + b.out0 + threadIdx.x;

    const int t_first = fT.fl, steps = d.T + 1;                   // plane of step k is t_first + k
    auto in_range = [&](int t) { return t >= 0 && t < d.T; };
    int issued = 0;
    auto feed = [&](int t, int s) {
        if (in_range(t)) {
            dma_taps<DR>(src0 + (ptrdiff_t)t * (ptrdiff_t)tstride_in, ring_addr + s * slot_bytes, cs);
            issued += cs.n_tap_wave;
        } else {
            zero_taps<DR>(ring + s * slot_f4, cs);
        }
    };
    int mark[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { feed(t_first + j, j); mark[j] = issued; }

    const bool wave_live = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~(kWave - 1))) < b.cells_out;
    float Sprev[4] = {0.f, 0.f, 0.f, 0.f};
    int slot = 0;
#pragma nounroll
    for (int k = 0; k < steps; ++k) {
        wait_vmcnt(issued - mark[0]);                              // my pieces of plane k have landed
        if (BN && in_range(t_first + k)) bn_taps<DR>(ring + slot * slot_f4, cs, bn_a, bn_b);
        __syncthreads();                                           // everyone's have; plane k-1 is retired
        {
            int sn = slot + D; if (sn >= R) sn -= R;
            feed(t_first + k + D, sn);
#pragma unroll
            for (int j = 0; j + 1 < D; ++j) mark[j] = mark[j + 1];
            mark[D - 1] = issued;
        }
        if (wave_live) {
            const float4* cur = ring + slot * slot_f4;
            const float4 a0 = lds_b128(cur + w.a[0]), a1 = lds_b128(cur + w.a[1]), a2 = lds_b128(cur + w.a[2]);
            const float4 b0 = lds_b128(cur + w.b[0]), b1 = lds_b128(cur + w.b[1]), b2 = lds_b128(cur + w.b[2]);
            float S[4];
#pragma unroll
            for (int m = 0; m < 4; ++m)                            // trilerp's per-plane part, :193-203
                S[m] = uH * (tap12<OFF>(a0, a1, a2, 2 * m) * uW + tap12<OFF>(a0, a1, a2, 2 * m + 1) * rW) +
                       rH * (tap12<OFF>(b0, b1, b2, 2 * m) * uW + tap12<OFF>(b0, b1, b2, 2 * m + 1) * rW);
            if (k >= 1) {                                          // output plane to = k - 1
                if (w.live) {
                    float4 o;
                    o.x = uT * Sprev[0] + rT * S[0];
                    o.y = uT * Sprev[1] + rT * S[1];
                    o.z = uT * Sprev[2] + rT * S[2];
                    o.w = uT * Sprev[3] + rT * S[3];
                    stream_store(out0 + (size_t)(k - 1) * (tstride_out / 4), o);
                }
                ++issued;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) Sprev[m] = S[m];
        }
        if (++slot == R) slot = 0;
    }
}

template <int DR, int D, bool BN = false>
__global__ __launch_bounds__(kBlock) void k3d_s2_forward(const float* __restrict__ x, const float* __restrict__ shift,
                                                         float* __restrict__ y, SDims d,
                                                         const float4* __restrict__ abmi = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    const int band = blockIdx.x % d.nbands, col = blockIdx.x / d.nbands;
    const int c = col % d.C, n = col / d.C;
    const Frac<float> fT = split_shift(shift[c]), fH = split_shift(shift[d.C + c]), fW = split_shift(shift[2 * d.C + c]);
    const size_t tin = (size_t)d.C * d.H * d.W, tout = (size_t)d.C * d.Ho * d.Wo;
    const float* xp = x + ((size_t)n * d.T * d.C + c) * d.H * d.W;
    float* yp = y + ((size_t)n * d.T * d.C + c) * d.Ho * d.Wo;
    const SBand b = make_sband(d, band, fH.fl);
    float bn_a = 0.f, bn_b = 0.f;
    if (BN) { const float4 pk = abmi[c]; bn_a = pk.x; bn_b = pk.y; }
    switch (((fW.fl % 4) + 4) % 4) {                                // wave-uniform
        case 0: forward_loop<DR, D, 0, BN>(xp, yp, ring, d, b, fT, fH, fW, tin, tout, bn_a, bn_b); break;
        case 1: forward_loop<DR, D, 1, BN>(xp, yp, ring, d, b, fT, fH, fW, tin, tout, bn_a, bn_b); break;
        case 2: forward_loop<DR, D, 2, BN>(xp, yp, ring, d, b, fT, fH, fW, tin, tout, bn_a, bn_b); break;
        default: forward_loop<DR, D, 3, BN>(xp, yp, ring, d, b, fT, fH, fW, tin, tout, bn_a, bn_b); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) (WRITE_GX) + d(shift) partials part[c][3][P], P = N * nbands, in ONE walk (adjoint form of
// rk3d_dma.hpp, with the NEGATED shift (fl', r')).  Through the stride un-mapping (K4, :586-589) an x element
// (hi, wi) has exactly ONE valid gy tap per plane: hi + fl'H + dh = 2 ho with dh = 1 (tap "row b", weight r'H) for
// the window's upper row hi = 2 ho - fl'H - 1 and dh = 0 (tap "row a", weight 1 - r'H) for the lower one, columns
// alike, so the reference's trilinear tree with its seven zero taps collapses to
//     A(tau)[hi, wi] = wH (g wW),   gx[to] = (1 - r'T) A(to + fl'T) + r'T A(to + fl'T + 1)     (same roundings),
// and the face differences of d(shift) (:432-441) become  -/+ (g wW) and -/+ (wH g)  times the x pairing of
// rk3d_dma.hpp.  A thread owns the gy cell (ho, 4 wo4 .. +3) -- DMA'd into a one-slot ring at its own position --
// and the 2 x 8 window of x (from the band's tap slot, as in the forward); d(x) leaves through an LDS tile with the
// slot's geometry (window values land at their unaligned columns, cells no window covers stay zero), copied out as
// aligned 16-byte nt stores by the lanes that DMA the same cells of x.  gx rows that no band covers (fl'H != -1)
// are zero-filled by the first / last band up front.  Any exactly-integer shift component: per-element path.
// BN (training fusion): xp holds z.  The window keeps the raw z; the activation max(a z + b, 0) is recomputed where the
// d(shift) sums use it -- zero for window cells outside the plane and for planes outside [0, T) -- and the d(x) values a
// thread writes into the tile are masked with the ReLU of its own window of plane k - 1 (xa) while bn2's sums
// sum(dz), sum(dz zhat) are reduced next to the d(shift) partials (rk3d_dma.hpp, dma_backward_loop).
template <int DR, bool WRITE_GX, int OFF, bool BN = false>
__device__ __forceinline__ void backward_loop(const float* __restrict__ xp, const float* __restrict__ gp,
                                              float* __restrict__ op, float4* ring, const SDims& d, const SBand& b,
                                              const Frac<float>& fT, const Frac<float>& fH, const Frac<float>& fW,
                                              int flW_eff, size_t tstride_in, size_t tstride_out, float& accT,
                                              float& accH, float& accW, float4 bnp = make_float4(0.f, 0.f, 0.f, 0.f),
                                              float* accB1 = nullptr, float* accB2 = nullptr) {
    const int slot_f4 = b.cells_in + 1;                           // x slots: + zero cell; out tile: + dump cell
    BCells<DR> cs;
    make_feed<DR>(cs, b);
    float4* const xring = ring;                                   // 2 slots
    float4* const otile = ring + 2 * slot_f4;
    float4* const gslot = otile + slot_f4;                        // kBlock own cells + zero cell
    init_slots<DR>(xring, 2, slot_f4, b, cs);
    if (WRITE_GX)
        for (int o = threadIdx.x; o < slot_f4; o += kBlock) otile[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x == 0) gslot[kBlock] = make_float4(0.f, 0.f, 0.f, 0.f);
    const Window w = make_window(d, b, (flW_eff - OFF) / 4);

    const float rT = fT.r, rH = fH.r, rW = fW.r;
    const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const unsigned xaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(xring));
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(gslot)) +
                           __builtin_amdgcn_readfirstlane((unsigned)threadIdx.x >> 6) * 1024u;
    const unsigned slot_bytes = (unsigned)slot_f4 * 16u;
    const float* xsrc0 = xp + (ptrdiff_t)b.src0 * 4;
    const float* gsrc0 = gp + (size_t)b.out0 * 4;                 // this band's first gy cell
    float* out0 = WRITE_GX ? op + (ptrdiff_t)b.src0 * 4 : nullptr;
    const int off0 = (int)threadIdx.x * 16;
    const bool wave_live = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~(kWave - 1))) < b.cells_out;
    const int gidx = w.live ? (int)threadIdx.x : kBlock;
    // byte addresses (relative to the tile) of the window's three cells per row; outside the plane -> the dump cell
    unsigned wa[3], wb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { wa[k] = (unsigned)w.a[k] * 16u; wb[k] = (unsigned)w.b[k] * 16u; }

    const int t_first = fT.fl, steps = d.T + 1;                   // gy plane of step k is t_first + k; x plane is k
    auto in_range = [&](int t) { return t >= 0 && t < d.T; };
    int issued = 0;
    auto feed = [&](int k) {                                      // gy plane t_first + k, x plane k
        if (wave_live) {
            if (in_range(t_first + k)) {
                if (w.live) dma16s(gsrc0 + (ptrdiff_t)(t_first + k) * (ptrdiff_t)tstride_out, off0, gaddr);
                ++issued;
            } else if (w.live) {
                gslot[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (in_range(k)) {
            dma_taps<DR>(xsrc0 + (size_t)k * tstride_in, xaddr + (k & 1) * slot_bytes, cs);
            issued += cs.n_tap_wave;
        } else {
            zero_taps<DR>(xring + (k & 1) * slot_f4, cs);
        }
    };
    feed(0);
    int mark = issued;

    float xa[16], xb[16], Qprev[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) xa[i] = xb[i] = Qprev[i] = 0.f;
    float sT = 0.f, sH = 0.f, sW = 0.f, sB1 = 0.f, sB2 = 0.f;
    // BN: which of the window's 3 cells per row lie inside the plane (a cell outside reads the zero cell: its
    // activation must be 0, not max(b, 0)), and the affine maps of the planes in xa / xb (0: no plane)
    bool cin_a[3], cin_b[3];
    {
        // (rows of the slot outside the plane are zero-FILLED cells, not the zero cell: test the row as well)
        const int jrow = (w.live ? (int)threadIdx.x : 0) / d.Wo4;
        const int row_a = b.src0 / d.W4 + 2 * jrow, row_b = row_a + 1;      // plane rows of the window (src0 = r0 * W4)
        const bool ra = row_a >= 0 && row_a < d.H, rb = row_b >= 0 && row_b < d.H;
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3) {
            cin_a[k3] = ra && w.a[k3] != b.cells_in;
            cin_b[k3] = rb && w.b[k3] != b.cells_in;
        }
    }
    float aA = 0.f, bA = 0.f, aB = 0.f, bB = 0.f;

#pragma nounroll
    for (int k = 0; k < steps; ++k) {
        wait_vmcnt(issued - mark);                                // my pieces of gy(t_first + k) and x[k] have landed
        __syncthreads();                                          // everyone's have; step k-1 (tile copy-out too) retired
        float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wave_live) {
            g4 = gslot[gidx];
            const float4* cur = xring + (k & 1) * slot_f4;
            const float4 a0 = lds_b128(cur + w.a[0]), a1 = lds_b128(cur + w.a[1]), a2 = lds_b128(cur + w.a[2]);
            const float4 b0 = lds_b128(cur + w.b[0]), b1 = lds_b128(cur + w.b[1]), b2 = lds_b128(cur + w.b[2]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xa[e] = xb[e]; xa[8 + e] = xb[8 + e];
                xb[e] = tap12<OFF>(a0, a1, a2, e);
                xb[8 + e] = tap12<OFF>(b0, b1, b2, e);
            }
        }
        if (k + 1 < steps) feed(k + 1);                           // (the DMA waits for the LDS reads above)
        mark = issued;
        if (BN) { aA = aB; bA = bB; aB = in_range(k) ? bnp.x : 0.f; bB = in_range(k) ? bnp.y : 0.f; }
        if (wave_live) {
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
            float q[16];
            float za[16];                                         // BN: raw z of plane k - 1 at the window (for zhat)
            if (BN) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { za[e] = xa[e]; za[8 + e] = xa[8 + e]; }
            }
            float xav[16], xbv[16];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = OFF + e;
                if (BN) {
                    xav[e] = cin_a[j >> 2] ? fmaxf(fmaf(aA, xa[e], bA), 0.f) : 0.f;
                    xbv[e] = cin_a[j >> 2] ? fmaxf(fmaf(aB, xb[e], bB), 0.f) : 0.f;
                    xav[8 + e] = cin_b[j >> 2] ? fmaxf(fmaf(aA, xa[8 + e], bA), 0.f) : 0.f;
                    xbv[8 + e] = cin_b[j >> 2] ? fmaxf(fmaf(aB, xb[8 + e], bB), 0.f) : 0.f;
                } else {
                    xav[e] = xa[e]; xbv[e] = xb[e]; xav[8 + e] = xa[8 + e]; xbv[8 + e] = xb[8 + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float wW = (e & 1) ? uW : rW;               // even e: column 2 wo - fl'W - 1 (tap w1), odd: tap w0
                const float gw = g[e >> 1] * wW;
                q[e] = rH * gw;                                   // upper row: tap "row b"
                q[8 + e] = uH * gw;                               // lower row: tap "row a"
                const float ca = rH * g[e >> 1], cb = uH * g[e >> 1];   // H-blend of the single tap, per row
                const float sgn = (e & 1) ? 1.f : -1.f;
                {
                    const float dx = xbv[e] - xav[e], mx = fmaf(uT, xbv[e], rT * xav[e]);
                    sT = fmaf(q[e], dx, sT);
                    sH = fmaf(-gw, mx, sH);
                    sW = fmaf(sgn * ca, mx, sW);
                }
                {
                    const float dx = xbv[8 + e] - xav[8 + e], mx = fmaf(uT, xbv[8 + e], rT * xav[8 + e]);
                    sT = fmaf(q[8 + e], dx, sT);
                    sH = fmaf(gw, mx, sH);
                    sW = fmaf(sgn * cb, mx, sW);
                }
            }
            if (WRITE_GX && k >= 1) {                             // window of output plane to = k - 1 -> tile
                char* tile = reinterpret_cast<char*>(otile);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = OFF + e;                        // constant
                    float oa = uT * Qprev[e] + rT * q[e], ob = uT * Qprev[8 + e] + rT * q[8 + e];
                    if (BN) {                                     // plane k - 1 = the plane in xa: ReLU mask + bn2's sums
                        oa = xav[e] > 0.f ? oa : 0.f;             // (xav is 0 outside the plane: nothing leaks into the sums)
                        ob = xav[8 + e] > 0.f ? ob : 0.f;
                        sB1 += oa + ob;
                        sB2 = fmaf(oa, (za[e] - bnp.z) * bnp.w, sB2);
                        sB2 = fmaf(ob, (za[8 + e] - bnp.z) * bnp.w, sB2);
                    }
                    *reinterpret_cast<float*>(tile + wa[j >> 2] + 4 * (j & 3)) = oa;
                    *reinterpret_cast<float*>(tile + wb[j >> 2] + 4 * (j & 3)) = ob;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) Qprev[i] = q[i];
        }
        if (WRITE_GX && k >= 1) {
            __syncthreads();                                      // the tile of plane k - 1 is complete
            char* out = reinterpret_cast<char*>(out0 + (size_t)(k - 1) * tstride_in) + off0;
            const char* tile = reinterpret_cast<const char*>(otile) + off0;
#pragma unroll
            for (int i = 0; i < DR; ++i)
                if (cs.in_act[i])
                    stream_store(reinterpret_cast<float4*>(out + 4096 * i), *reinterpret_cast<const float4*>(tile + 4096 * i));
            issued += cs.n_tap_wave;
        }
    }
    accT = sT; accH = sH; accW = sW;
    if (BN) { *accB1 = sB1; *accB2 = sB2; }
}

template <int DR, bool WRITE_GX, bool FUSED, bool BN = false>
__global__ __launch_bounds__(kBlock) void k3d_s2_backward(const float* __restrict__ x, const float* __restrict__ shift,
                                                          const float* __restrict__ gy, float* __restrict__ gx,
                                                          float* __restrict__ part, SDims d, Dims3 gd, dma3d::Fin3 fin,
                                                          dma3d::BnFuse bn = dma3d::BnFuse{}) {
    constexpr int ND = BN ? 5 : 3;
    if (FUSED && (int)blockIdx.x >= fin.f.producers) {
        if (threadIdx.x < kWave) dma3d::finalizer_wave<ND>(fin, (int)blockIdx.x - fin.f.producers, d.C, d.N * d.nbands, bn);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    __shared__ float red[ND][kBlock / kWave];
    const int band = blockIdx.x % d.nbands, col = blockIdx.x / d.nbands;
    const int c = col % d.C, n = col / d.C;
    const float s0 = shift[c], s1 = shift[d.C + c], s2 = shift[2 * d.C + c];
    float accT = 0.f, accH = 0.f, accW = 0.f, accB1 = 0.f, accB2 = 0.f;
    float4 bnp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BN) bnp = bn.abmi[c];
    const Frac<float> fT = split_shift(-s0), fH = split_shift(-s1), fW = split_shift(-s2);   // fl', r'

    if (fT.r == 0 || fH.r == 0 || fW.r == 0) {
        // exactly-integer component (lowered-index quirk, zero-shift branch): rare, per element; band 0 does the column
        if (band == 0) {
            if (WRITE_GX)
                for (int t = 0; t < d.T; ++t)
                    backward_input_plane<float, false>(shift, gy, gx, gd, n, t, c, threadIdx.x, kBlock);
            if (BN) {
                const BnAct act{bnp.x, bnp.y};
                for (int to = 0; to < d.T; ++to)
                    shift_grad_plane<float>(x, shift, gy, gd, n, to, c, threadIdx.x, kBlock, accT, accH, accW, act);
                const int HW = d.H * d.W;
                for (int t = 0; t < d.T; ++t) {                     // (each thread re-reads the elements it wrote itself)
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
        const size_t tin = (size_t)d.C * d.H * d.W, tout = (size_t)d.C * d.Ho * d.Wo;
        const float* xp = x + ((size_t)n * d.T * d.C + c) * d.H * d.W;
        const float* gp = gy + ((size_t)n * d.T * d.C + c) * d.Ho * d.Wo;
        float* op = WRITE_GX ? gx + ((size_t)n * d.T * d.C + c) * d.H * d.W : nullptr;
        const int flH_eff = -fH.fl - 1, flW_eff = -fW.fl - 1;     // first row / column of the window of output (0, 0)
        if (WRITE_GX) {
            // rows of gx no band's window reaches: [0, flH_eff) by the first band, [H + flH_eff, H) by the last
            auto zero_rows = [&](int lo, int hi) {
                lo = lo < 0 ? 0 : lo;
                hi = hi > d.H ? d.H : hi;
                for (int t = 0; t < d.T; ++t)
                    for (int cell = lo * d.W4 + (int)threadIdx.x; cell < hi * d.W4; cell += kBlock)
                        stream_store(reinterpret_cast<float4*>(op + (size_t)t * tin) + cell, make_float4(0.f, 0.f, 0.f, 0.f));
            };
            if (band == 0) zero_rows(0, flH_eff);
            if (band == d.nbands - 1) zero_rows(2 * d.Ho + flH_eff, d.H);
        }
        const SBand b = make_sband(d, band, flH_eff);
#define RK_S2_BWD(O) backward_loop<DR, WRITE_GX, O, BN>(xp, gp, op, ring, d, b, fT, fH, fW, flW_eff, tin, tout, accT, accH, accW, bnp, &accB1, &accB2)
        switch (((flW_eff % 4) + 4) % 4) {                          // wave-uniform
            case 0: RK_S2_BWD(0); break;
            case 1: RK_S2_BWD(1); break;
            case 2: RK_S2_BWD(2); break;
            default: RK_S2_BWD(3); break;
        }
#undef RK_S2_BWD
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
                fin_publish(fin.f, at + 3 * (size_t)P, accB1);
                fin_publish(fin.f, at + 4 * (size_t)P, accB2);
            }
        } else {
            part[at] = accT;
            part[at + P] = accH;
            part[at + 2 * P] = accW;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host side.  false = shape not handled here.
inline bool make_sdims(SDims& s, const Dims3& d) {
    const bool ok = d.sT == 1 && d.sH == 2 && d.sW == 2 && d.pT == 0 && d.pH == 0 && d.pW == 0;
    if (!ok || d.W % 4 != 0 || d.Wo % 4 != 0 || d.H % 2 != 0 || d.W % 2 != 0 || !streaming_kernels_on()) return false;
    s.N = d.N; s.T = d.T; s.C = d.C; s.H = d.H; s.W = d.W; s.W4 = d.W / 4;
    s.Ho = d.Ho; s.Wo = d.Wo; s.Wo4 = d.Wo / 4;
    for (int nb = 1; nb <= s.Ho; ++nb) {                           // fewest bands with <= 1024 slot cells, <= 256 outputs
        if (s.Ho % nb) continue;
        const int bh = s.Ho / nb;
        if (2 * bh * s.W4 > 4 * kBlock || bh * s.Wo4 > kBlock) continue;
        s.BHo = bh; s.nbands = nb;
        return true;
    }
    return false;
}
inline size_t ring_bytes(const SDims& s, int slots) { return (size_t)slots * (2 * s.BHo * s.W4 + 1) * 16; }

inline bool launch_forward(const float* x, const float* shift, float* y, const Dims3& d, hipStream_t stream) {
    constexpr int D = 2;
    SDims s;
    if (!make_sdims(s, d) || !aligned16(x) || !aligned16(y)) return false;
    const size_t lds = ring_bytes(s, D + 1);
    if (lds > 64 * 1024) return false;
    const dim3 grid((unsigned)(s.N * s.C * s.nbands)), block(kBlock);
    hipLaunchKernelGGL((k3d_s2_forward<4, D>), grid, block, lds, stream, x, shift, y, s);
    return true;
}

// training fusion (BN): forward of relu(bn(z)) and its backward; false = not handled here
inline bool launch_forward_bn(const float* z, const float* shift, float* y, const float4* abmi, const Dims3& d,
                              hipStream_t stream) {
    constexpr int D = 2;
    SDims s;
    if (!make_sdims(s, d) || !aligned16(z) || !aligned16(y) || !aligned16(abmi)) return false;
    const size_t lds = ring_bytes(s, D + 1);
    if (lds > 64 * 1024) return false;
    const dim3 grid((unsigned)(s.N * s.C * s.nbands)), block(kBlock);
    hipLaunchKernelGGL((k3d_s2_forward<4, D, true>), grid, block, lds, stream, z, shift, y, s, abmi);
    return true;
}
inline size_t bwd_lds_bytes(const SDims& s);
inline bool launch_backward_bn(const float* z, const float* shift, const float* gy, float* gx, float* gshift, float* ws,
                               const Dims3& d, int normalize, float t_factor, const dma3d::BnFuse& bn, hipStream_t stream) {
    SDims s;
    if (!make_sdims(s, d) || !aligned16(z) || !aligned16(gy) || !aligned16(gx) || !aligned16(bn.abmi)) return false;
    const size_t lds = bwd_lds_bytes(s);
    if (lds > 64 * 1024) return false;
    dma3d::Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = s.N * s.C * s.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
    hipLaunchKernelGGL((k3d_s2_backward<4, true, true, true>), dim3((unsigned)(fin.f.producers + s.C)), dim3(kBlock), lds, stream,
                       z, shift, gy, gx, ws, s, d, fin, bn);
    return true;
}

// d(shift) (+ d(x) when gx != nullptr); gshift != nullptr: row-sum + K5 inside the launch (ws = 8-byte granules
// [C][3][P]), else plain partials ws[C][3][P] for the two-phase ABI.  Returns P (0 = not handled here).
inline size_t bwd_lds_bytes(const SDims& s) { return ((size_t)3 * (2 * s.BHo * s.W4 + 1) + kBlock + 1) * 16; }
inline int launch_backward(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws,
                           const Dims3& d, int normalize, float t_factor, hipStream_t stream) {
    SDims s;
    if (!make_sdims(s, d) || !aligned16(x) || !aligned16(gy) || (gx && !aligned16(gx))) return 0;
    const size_t lds = bwd_lds_bytes(s);
    if (lds > 64 * 1024) return 0;
    dma3d::Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = s.N * s.C * s.nbands;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
    const dim3 block(kBlock);
#define RK_S2_LAUNCH(GX, FU) hipLaunchKernelGGL((k3d_s2_backward<4, GX, FU>), dim3((unsigned)(fin.f.producers + (FU ? s.C : 0))), \
                                                block, lds, stream, x, shift, gy, gx, ws, s, d, fin)
    if (gshift) { if (gx) RK_S2_LAUNCH(true, true); else RK_S2_LAUNCH(false, true); }
    else { if (gx) RK_S2_LAUNCH(true, false); else RK_S2_LAUNCH(false, false); }
#undef RK_S2_LAUNCH
    return s.N * s.nbands;
}

}  // namespace s2
}  // namespace rk
