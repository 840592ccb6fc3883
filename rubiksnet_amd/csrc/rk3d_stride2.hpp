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
template <int DR, int D, int OFF>
__device__ __forceinline__ void forward_loop(const float* __restrict__ xp, float* __restrict__ yp, float4* ring,
                                             const SDims& d, const SBand& b, const Frac<float>& fT,
                                             const Frac<float>& fH, const Frac<float>& fW, size_t tstride_in,
                                             size_t tstride_out) {
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

template <int DR, int D>
__global__ __launch_bounds__(kBlock) void k3d_s2_forward(const float* __restrict__ x, const float* __restrict__ shift,
                                                         float* __restrict__ y, SDims d) {
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    const int band = blockIdx.x % d.nbands, col = blockIdx.x / d.nbands;
    const int c = col % d.C, n = col / d.C;
    const Frac<float> fT = split_shift(shift[c]), fH = split_shift(shift[d.C + c]), fW = split_shift(shift[2 * d.C + c]);
    const size_t tin = (size_t)d.C * d.H * d.W, tout = (size_t)d.C * d.Ho * d.Wo;
    const float* xp = x + ((size_t)n * d.T * d.C + c) * d.H * d.W;
    float* yp = y + ((size_t)n * d.T * d.C + c) * d.Ho * d.Wo;
    const SBand b = make_sband(d, band, fH.fl);
    switch (((fW.fl % 4) + 4) % 4) {                                // wave-uniform
        case 0: forward_loop<DR, D, 0>(xp, yp, ring, d, b, fT, fH, fW, tin, tout); break;
        case 1: forward_loop<DR, D, 1>(xp, yp, ring, d, b, fT, fH, fW, tin, tout); break;
        case 2: forward_loop<DR, D, 2>(xp, yp, ring, d, b, fT, fH, fW, tin, tout); break;
        default: forward_loop<DR, D, 3>(xp, yp, ring, d, b, fT, fH, fW, tin, tout); break;
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

}  // namespace s2
}  // namespace rk
