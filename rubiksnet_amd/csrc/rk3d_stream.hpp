// rk3d_stream.hpp -- LDS-tiled streaming kernels for RubiksShift3D, fp32, stride 1 / pad 0
// (13 of 17 shift layers of Tiny, 47 of 51 of Large, and the BASELINE benchmark shape).
//
// Why this shape: the shift is constant per channel, so output plane (n, to, c) is a fixed 2-D
// translate-and-blend of exactly two input planes (n, to+flT, c) and (n, to+flT+1, c), and the
// H/W-interpolated field B(t) of an input plane is shared by the two outputs that touch it:
//     y[to] = (1-rT) * B(to+flT) + rT * B(to+flT+1),
//     B(t)[h,w] = (1-rH) * lerpW(x[t][h+flH]) + rH * lerpW(x[t][h+flH+1]).
// That is the reference's own expression tree (rubiks3d_kernels.cu:193-203) evaluated once per
// input element instead of twice, so results stay bit-identical to the oracle.
//
// One 256-thread workgroup owns one (n, c) column and walks t.  Per step it
//   - stores the prefetched plane (16 B per lane, coalesced global loads issued one step
//     earlier) into an LDS tile [H+1][W+4] whose last row and last 4 columns are zero, so every
//     out-of-range tap is an ordinary LDS read of 0 -- no per-tap bounds branches;
//   - issues the global loads of the next plane (latency hides under this step's work);
//   - reads its taps with aligned ds_read_b128 only: the 5 consecutive values a float4 of
//     outputs needs lie in 2 aligned 4-groups, selected by the wave-uniform (flW mod 4);
//   - blends with the previous step's B (registers) and stores y with 16 B per lane.
// Each HBM byte is read once and written once (8 B/elem); nothing is memset.
//
// d(x) is the same kernel on gy with the negated shift (rubiks3d_kernels.cu:726-929).  Channels
// whose shift is exactly zero take the copy branch (:819-827) inside the same kernel.
// d(shift) streams x through the same tile and gy through registers (16 B per lane), reduces
// wave-shuffle -> LDS -> one partial per (n, c) -> k3d_finalize (no atomics).  Channels with an
// exactly-integer shift component (the lowered-index quirk, :290-298) fall back, per workgroup,
// to the per-element formulation shared with the generic kernels.
#pragma once
#include "rk3d_generic.hpp"

namespace rk {
namespace stream3d {

constexpr int kMaxRounds = 4;          // float4 cells per thread: planes up to 1024 cells (56x56 = 784)

struct SDims {
    int N, T, C, H, W;
    int W4, Wp, cells;                 // W/4, W+4, H*W/4
};

// the 5 consecutive values starting `off` floats into the aligned pair (q0, q1); off is wave-uniform
__device__ __forceinline__ void pick5(const float4& q0, const float4& q1, int off, float (&v)[5]) {
    switch (off) {
        case 0: v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; break;
        case 1: v[0] = q0.y; v[1] = q0.z; v[2] = q0.w; v[3] = q1.x; v[4] = q1.y; break;
        case 2: v[0] = q0.z; v[1] = q0.w; v[2] = q1.x; v[3] = q1.y; v[4] = q1.z; break;
        default: v[0] = q0.w; v[1] = q1.x; v[2] = q1.y; v[3] = q1.z; v[4] = q1.w; break;
    }
}

// Per-thread geometry of its ROUNDS float4 cells: cell i covers elements [4*(tid + 256*i), +4) of the
// flat plane = row h, columns 4*w4 .. 4*w4+3.  All LDS indices are in float4 units (the tile pitch
// Wp/4 = W4+1 groups), so every LDS access is a 16 B-aligned b128.  Threads past the end of the
// plane (last round only) are clamped onto the last cell: they load and compute like it but
// never store (`live`).
template <int ROUNDS> struct Cells {
    int cell[ROUNDS];   // clamped flat float4 index within the plane
    bool live[ROUNDS];  // this thread really owns the cell
    int own[ROUNDS];    // tile slot of the cell itself
    int rowA[ROUNDS];   // tile slot of the start of tap row h+flH   (the zero row if outside)
    int rowB[ROUNDS];   // ... of tap row h+flH+1
    int g0[ROUNDS];     // slot offset of the first aligned tap group in a row (the zero group if outside)
    int g1[ROUNDS];     // ... of the second
};

template <int ROUNDS>
__device__ __forceinline__ void make_cells(Cells<ROUNDS>& cs, const SDims& d, int flH, int group_shift) {
    const int P4 = d.W4 + 1;   // tile pitch in float4 slots
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int raw = (int)threadIdx.x + kBlock * i;
        cs.live[i] = raw < d.cells;
        const int cell = cs.live[i] ? raw : d.cells - 1;
        cs.cell[i] = cell;
        const int h = cell / d.W4, w4 = cell - h * d.W4;
        cs.own[i] = h * P4 + w4;
        const int ra = h + flH, rb = h + flH + 1;
        cs.rowA[i] = ((ra >= 0 && ra < d.H) ? ra : d.H) * P4;
        cs.rowB[i] = ((rb >= 0 && rb < d.H) ? rb : d.H) * P4;
        const int ga = w4 + group_shift, gb = ga + 1;
        cs.g0[i] = (ga >= 0 && ga < d.W4) ? ga : d.W4;
        cs.g1[i] = (gb >= 0 && gb < d.W4) ? gb : d.W4;
    }
}

// does any lane of this wave own a cell in round i?  (wave-uniform, so whole rounds are skipped)
__device__ __forceinline__ bool wave_round_on(int i, int cells) {
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~(kWave - 1))) + kBlock * i < cells;
}

__device__ __forceinline__ void zero_halo(float4* tile, const SDims& d) {
    const int P4 = d.W4 + 1;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < P4; i += kBlock) tile[d.H * P4 + i] = z;          // zero row
    for (int i = threadIdx.x; i < d.H; i += kBlock) tile[i * P4 + d.W4] = z;        // zero group per row
}

// issue the 16 B-per-lane loads of plane t of one (n, c) column into registers (zeros outside [0,T))
template <int ROUNDS>
__device__ __forceinline__ void fetch_plane(float4 (&pre)[ROUNDS], const Cells<ROUNDS>& cs, const float* col,
                                            int t, int T, size_t tstride) {
    if (t >= 0 && t < T) {
        const float4* p = reinterpret_cast<const float4*>(col + (size_t)t * tstride);
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) pre[i] = p[cs.cell[i]];
    }
}

// same, for a plane index the caller guarantees to be in range: straight-line loads hipcc can count
template <int ROUNDS>
__device__ __forceinline__ void fetch_plane_always(float4 (&pre)[ROUNDS], const Cells<ROUNDS>& cs, const float* col,
                                                   int t, size_t tstride) {
    const float4* p = reinterpret_cast<const float4*>(col + (size_t)t * tstride);
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) pre[i] = p[cs.cell[i]];
}

// Pin: everything that produces `v` (including the wait for the loads behind it) is scheduled before
// this point, and no later load is hoisted above it.
template <int ROUNDS> __device__ __forceinline__ void pin(float4 (&v)[ROUNDS]) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
        asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w)::"memory");
}

template <int ROUNDS>
__device__ __forceinline__ void stash_plane(float4* tile, const float4 (&pre)[ROUNDS], const Cells<ROUNDS>& cs) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) tile[cs.own[i]] = pre[i];   // clamped duplicates write identical data
}

// ---------------------------------------------------------------------------------------------
// Forward (NEGATE = false, src = x, dst = y) and d(x) (NEGATE = true, src = gy, dst = gx).
template <bool NEGATE, int ROUNDS>
__global__ __launch_bounds__(kBlock) void k3d_stream_interp(const float* __restrict__ src,
                                                            const float* __restrict__ shift,
                                                            float* __restrict__ dst, SDims d) {
    extern __shared__ __attribute__((aligned(16))) float4 tile[];
    const int c = blockIdx.x % d.C, n = blockIdx.x / d.C;
    float sT = shift[c], sH = shift[d.C + c], sW = shift[2 * d.C + c];
    if (NEGATE) { sT = -sT; sH = -sH; sW = -sW; }
    const Frac<float> fT = split_shift(sT), fH = split_shift(sH), fW = split_shift(sW);
    const int HW = d.H * d.W;
    const size_t tstride = (size_t)d.C * HW;
    const float* sp = src + ((size_t)n * d.T * d.C + c) * HW;   // (n, t = 0, c)
    float* dp = dst + ((size_t)n * d.T * d.C + c) * HW;

    if (NEGATE && sT == 0 && sH == 0 && sW == 0) {              // rubiks3d_kernels.cu:819-827
        for (int t = 0; t < d.T; ++t)
            for (int cell = threadIdx.x; cell < d.cells; cell += kBlock)
                reinterpret_cast<float4*>(dp + (size_t)t * tstride)[cell] =
                    reinterpret_cast<const float4*>(sp + (size_t)t * tstride)[cell];
        return;
    }

    const int off = ((fW.fl % 4) + 4) % 4;                       // wave-uniform
    Cells<ROUNDS> cs;
    make_cells<ROUNDS>(cs, d, fH.fl, (fW.fl - off) / 4);
    zero_halo(tile, d);

    const float rT = fT.r, rH = fH.r, rW = fW.r;
    const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;

    float4 pre[ROUNDS];                 // prefetched plane (this thread's cells)
    float4 Bprev[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) pre[i] = Bprev[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int t_first = fT.fl, t_last = d.T + fT.fl;            // T+1 steps; planes outside [0,T) are zero
    fetch_plane<ROUNDS>(pre, cs, sp, t_first, d.T, tstride);
    for (int t = t_first; t <= t_last; ++t) {
        const bool valid = t >= 0 && t < d.T;
        if (valid) stash_plane<ROUNDS>(tile, pre, cs);
        __syncthreads();
        fetch_plane<ROUNDS>(pre, cs, sp, t + 1, d.T, tstride);
        const int to = t - fT.fl - 1;
        const bool emit = to >= 0 && to < d.T;
        float4* out = reinterpret_cast<float4*>(dp + (size_t)(emit ? to : 0) * tstride);
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            if (!wave_round_on(i, d.cells)) continue;
            float4 Bc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) {
                float a[5], b[5];
                pick5(lds_b128(tile + cs.rowA[i] + cs.g0[i]), lds_b128(tile + cs.rowA[i] + cs.g1[i]), off, a);
                pick5(lds_b128(tile + cs.rowB[i] + cs.g0[i]), lds_b128(tile + cs.rowB[i] + cs.g1[i]), off, b);
                Bc.x = uH * (a[0] * uW + a[1] * rW) + rH * (b[0] * uW + b[1] * rW);
                Bc.y = uH * (a[1] * uW + a[2] * rW) + rH * (b[1] * uW + b[2] * rW);
                Bc.z = uH * (a[2] * uW + a[3] * rW) + rH * (b[2] * uW + b[3] * rW);
                Bc.w = uH * (a[3] * uW + a[4] * rW) + rH * (b[3] * uW + b[4] * rW);
            }
            if (emit && cs.live[i]) {
                float4 o;
                o.x = uT * Bprev[i].x + rT * Bc.x;
                o.y = uT * Bprev[i].y + rT * Bc.y;
                o.z = uT * Bprev[i].z + rT * Bc.z;
                o.w = uT * Bprev[i].w + rT * Bc.w;
                out[cs.cell[i]] = o;
            }
            Bprev[i] = Bc;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) and the d(shift) partials in ONE pass over (gy, x) -- 12 B/elem -- or, with
// WRITE_GX = false, the d(shift) partials alone (8 B/elem).  One partial per (n, c):
// part[c][3][P = N], p = n, reduced wave-shuffle -> LDS (no atomics), summed by k3d_finalize.
//
// Adjoint form: with the negated shift (fl', r') the reference's d(x) is
//     gx[t] = (1-r'T) Q(t+fl'T) + r'T Q(t+fl'T+1),   Q(tg) = bilinear(gy[tg]; fl'H, fl'W, r'H, r'W)
// (rubiks3d_kernels.cu:914-924, evaluated in that exact tree => bit-identical), and because
// y is linear in x the shift gradients (the face differences of rubiks3d_kernels.cu:432-441) can
// be collected on the INPUT side from the same taps:
//     gT = sum_x x[t] * (Q(t+fl'T) - Q(t+fl'T+1))
//     gH = sum_x x[t] * ((1-r'T) QH(t+fl'T) + r'T QH(t+fl'T+1)),  QH = lerpW'(row A) - lerpW'(row B)
//     gW = likewise with QW = lerpH'(col k) - lerpH'(col k+1)
// so only gy needs the LDS tile; x is read at the thread's own (aligned) cells, 16 B per lane,
// through a register window (x[to], x[to+1]) with one more plane in flight.
// Regrouped per gy plane tg (to = tg - fl'T - 1):
//     sT += Q(tg) * (x[to+1] - x[to]);   sH += QH(tg) * ((1-r'T) x[to+1] + r'T x[to]);   sW alike.
//
// Pipeline note: prefetch registers are written ONLY by loads (plane index clamped into range,
// out-of-range planes are masked where they are used).  Merging a prefetch with a constant
// ("else = 0") makes hipcc copy the loaded registers at the merge point behind an
// s_waitcnt vmcnt(0) right after the loads are issued, which serialises every step.
template <int ROUNDS, bool WRITE_GX>
__global__ __launch_bounds__(kBlock) void k3d_stream_backward(const float* __restrict__ x,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ gy,
                                                              float* __restrict__ gx,
                                                              float* __restrict__ part, SDims d, Dims3 gd) {
    extern __shared__ __attribute__((aligned(16))) float4 tile[];
    __shared__ float red[3][kBlock / kWave];
    const int c = blockIdx.x % d.C, n = blockIdx.x / d.C;
    const float s0 = shift[c], s1 = shift[d.C + c], s2 = shift[2 * d.C + c];
    float accT = 0.f, accH = 0.f, accW = 0.f;

    if (split_shift(s0).r == 0 || split_shift(s1).r == 0 || split_shift(s2).r == 0) {
        // exactly-integer component (lowered-index quirk / zero-shift copy branch): rare, per element
        if (WRITE_GX)
            for (int t = 0; t < d.T; ++t)
                backward_input_plane<float, false>(shift, gy, gx, gd, n, t, c, threadIdx.x, kBlock);
        for (int to = 0; to < d.T; ++to)
            shift_grad_plane<float>(x, shift, gy, gd, n, to, c, threadIdx.x, kBlock, accT, accH, accW);
    } else {
        const Frac<float> fT = split_shift(-s0), fH = split_shift(-s1), fW = split_shift(-s2);   // fl', r'
        const int HW = d.H * d.W;
        const size_t tstride = (size_t)d.C * HW;
        const float* xp = x + ((size_t)n * d.T * d.C + c) * HW;
        const float* gp = gy + ((size_t)n * d.T * d.C + c) * HW;
        float* op = WRITE_GX ? gx + ((size_t)n * d.T * d.C + c) * HW : nullptr;

        const int off = ((fW.fl % 4) + 4) % 4;
        Cells<ROUNDS> cs;
        make_cells<ROUNDS>(cs, d, fH.fl, (fW.fl - off) / 4);
        zero_halo(tile, d);
        const float rT = fT.r, rH = fH.r, rW = fW.r;
        const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;

        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gpre[ROUNDS], xpre[ROUNDS], xa[ROUNDS], xb[ROUNDS], Qprev[ROUNDS];
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) gpre[i] = xpre[i] = xa[i] = xb[i] = Qprev[i] = z4;
        float sT = 0.f, sH = 0.f, sW = 0.f;

        auto clampT = [&](int t) { return t < 0 ? 0 : (t >= d.T ? d.T - 1 : t); };
        const int t_first = fT.fl, t_last = d.T + fT.fl;          // gy plane index tg; T+1 steps
        fetch_plane<ROUNDS>(gpre, cs, gp, t_first, d.T, tstride);
        fetch_plane<ROUNDS>(xpre, cs, xp, 0, d.T, tstride);        // becomes x[to+1] of the first step (to = -1)
        for (int tg = t_first; tg <= t_last; ++tg) {
            const bool valid = tg >= 0 && tg < d.T;
            const int to = tg - fT.fl - 1;                          // -1 .. T-1
            if (valid) stash_plane<ROUNDS>(tile, gpre, cs);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < ROUNDS; ++i) { xa[i] = xb[i]; xb[i] = xpre[i]; }   // window: x[to], x[to+1]
            pin<ROUNDS>(xb);                                        // the wait for xpre sits HERE, before new loads
            fetch_plane_always<ROUNDS>(xpre, cs, xp, clampT(to + 2), tstride);     // always a real load
            fetch_plane<ROUNDS>(gpre, cs, gp, tg + 1, d.T, tstride);
            const float ka = (to >= 0) ? 1.f : 0.f, kb = (to + 1 < d.T) ? 1.f : 0.f; // x[-1] = x[T] = 0
            const bool emit = WRITE_GX && to >= 0;                  // to <= T-1 always
            float4* out = reinterpret_cast<float4*>(op + (size_t)(emit ? to : 0) * tstride);
#pragma unroll
            for (int i = 0; i < ROUNDS; ++i) {
                if (!wave_round_on(i, d.cells)) continue;
                float4 Qc = z4;
                if (valid) {
                    float a[5], b[5], col[5], q[4];
                    pick5(lds_b128(tile + cs.rowA[i] + cs.g0[i]), lds_b128(tile + cs.rowA[i] + cs.g1[i]), off, a);
                    pick5(lds_b128(tile + cs.rowB[i] + cs.g0[i]), lds_b128(tile + cs.rowB[i] + cs.g1[i]), off, b);
                    const float xav[4] = {ka * xa[i].x, ka * xa[i].y, ka * xa[i].z, ka * xa[i].w};
                    const float xbv[4] = {kb * xb[i].x, kb * xb[i].y, kb * xb[i].z, kb * xb[i].w};
#pragma unroll
                    for (int k = 0; k < 5; ++k) col[k] = fmaf(uH, a[k], rH * b[k]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float la = a[k] * uW + a[k + 1] * rW, lb = b[k] * uW + b[k + 1] * rW;
                        q[k] = uH * la + rH * lb;                   // the reference's tree, contraction off
                        if (cs.live[i]) {
                            const float dx = xbv[k] - xav[k];
                            const float mx = fmaf(uT, xbv[k], rT * xav[k]);
                            sT = fmaf(q[k], dx, sT);
                            sH = fmaf(la - lb, mx, sH);
                            sW = fmaf(col[k] - col[k + 1], mx, sW);
                        }
                    }
                    Qc = make_float4(q[0], q[1], q[2], q[3]);
                }
                if (emit && cs.live[i]) {
                    float4 o;
                    o.x = uT * Qprev[i].x + rT * Qc.x;
                    o.y = uT * Qprev[i].y + rT * Qc.y;
                    o.z = uT * Qprev[i].z + rT * Qc.z;
                    o.w = uT * Qprev[i].w + rT * Qc.w;
                    out[cs.cell[i]] = o;
                }
                if (WRITE_GX) Qprev[i] = Qc;
            }
            __syncthreads();
        }
        accT = sT; accH = sH; accW = sW;
    }

    accT = group_sum(accT, kBlock, red[0]);
    accH = group_sum(accH, kBlock, red[1]);
    accW = group_sum(accW, kBlock, red[2]);
    if (threadIdx.x == 0) {
        float* o = part + (size_t)c * 3 * d.N + n;
        o[0] = accT;
        o[d.N] = accH;
        o[2 * d.N] = accW;
    }
}

// ---------------------------------------------------------------------------------------------
inline bool env_force_generic() {
    static const bool v = [] { const char* e = getenv("RK_FORCE_GENERIC"); return e && e[0] == '1'; }();
    return v;
}

inline bool shape_ok(const Dims3& d) {
    const bool s1p0 = d.sT == 1 && d.sH == 1 && d.sW == 1 && d.pT == 0 && d.pH == 0 && d.pW == 0;
    const int cells = d.H * d.W / 4;
    return s1p0 && d.W % 4 == 0 && d.W >= 4 && cells <= kMaxRounds * kBlock &&
           (size_t)(d.H + 1) * (d.W + 4) * sizeof(float) <= 64 * 1024 && !env_force_generic();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline SDims make_sdims(const Dims3& d) {
    SDims s;
    s.N = d.N; s.T = d.T; s.C = d.C; s.H = d.H; s.W = d.W;
    s.W4 = d.W / 4; s.Wp = d.W + 4; s.cells = d.H * d.W / 4;
    return s;
}

template <typename T> bool forward_supported(const Dims3&, int, const void*, const void*) { return false; }
template <> inline bool forward_supported<float>(const Dims3& d, int quantize, const void* x, const void* y) {
    return !quantize && shape_ok(d) && aligned16(x) && aligned16(y);
}

template <typename T> bool backward_supported(const Dims3&, int, const void*, const void*, const void*) {
    return false;
}
template <> inline bool backward_supported<float>(const Dims3& d, int quantize, const void* x, const void* gy,
                                                  const void* gx) {
    return !quantize && shape_ok(d) && aligned16(x) && aligned16(gy) && aligned16(gx);
}

template <bool NEGATE>
inline void launch_interp(const float* src, const float* shift, float* dst, const SDims& s, hipStream_t stream) {
    const size_t lds = (size_t)(s.H + 1) * s.Wp * sizeof(float);
    const dim3 grid((unsigned)(s.N * s.C)), block(kBlock);
    const int rounds = (s.cells + kBlock - 1) / kBlock;
    switch (rounds) {
        case 1: hipLaunchKernelGGL((k3d_stream_interp<NEGATE, 1>), grid, block, lds, stream, src, shift, dst, s); break;
        case 2: hipLaunchKernelGGL((k3d_stream_interp<NEGATE, 2>), grid, block, lds, stream, src, shift, dst, s); break;
        case 3: hipLaunchKernelGGL((k3d_stream_interp<NEGATE, 3>), grid, block, lds, stream, src, shift, dst, s); break;
        default: hipLaunchKernelGGL((k3d_stream_interp<NEGATE, 4>), grid, block, lds, stream, src, shift, dst, s); break;
    }
}

template <typename T>
int launch_forward(const T*, const T*, T*, const Dims3&, hipStream_t) { return RK_ERR_LAUNCH; }
template <>
inline int launch_forward<float>(const float* x, const float* shift, float* y, const Dims3& d, hipStream_t stream) {
    launch_interp<false>(x, shift, y, make_sdims(d), stream);
    return launch_status();
}

template <bool WRITE_GX>
inline void launch_bwd(const float* x, const float* shift, const float* gy, float* gx, float* ws, const SDims& s,
                       const Dims3& d, hipStream_t stream) {
    const size_t lds = (size_t)(s.H + 1) * s.Wp * sizeof(float);
    const dim3 grid((unsigned)(s.N * s.C)), block(kBlock);
    switch ((s.cells + kBlock - 1) / kBlock) {
        case 1: hipLaunchKernelGGL((k3d_stream_backward<1, WRITE_GX>), grid, block, lds, stream, x, shift, gy, gx, ws, s, d); break;
        case 2: hipLaunchKernelGGL((k3d_stream_backward<2, WRITE_GX>), grid, block, lds, stream, x, shift, gy, gx, ws, s, d); break;
        case 3: hipLaunchKernelGGL((k3d_stream_backward<3, WRITE_GX>), grid, block, lds, stream, x, shift, gy, gx, ws, s, d); break;
        default: hipLaunchKernelGGL((k3d_stream_backward<4, WRITE_GX>), grid, block, lds, stream, x, shift, gy, gx, ws, s, d); break;
    }
}

template <typename T>
int launch_backward(const T*, const T*, const T*, T*, T*, const Dims3&, int, T, T*, hipStream_t) {
    return RK_ERR_LAUNCH;
}
template <>
inline int launch_backward<float>(const float* x, const float* shift, const float* gy, float* gx, float* gshift,
                                  const Dims3& d, int normalize, float t_factor, float* ws, hipStream_t stream) {
    const SDims s = make_sdims(d);
    if (gshift) {
        if (gx) launch_bwd<true>(x, shift, gy, gx, ws, s, d, stream);
        else launch_bwd<false>(x, shift, gy, nullptr, ws, s, d, stream);
        hipLaunchKernelGGL((k3d_finalize<float>), dim3(s.C), dim3(finalize_block(s.N)), 0, stream, (const float*)ws, gshift, s.C,
                           s.N, normalize, t_factor);
    } else if (gx) {
        launch_interp<true>(gy, shift, gx, s, stream);
    }
    return launch_status();
}

}  // namespace stream3d
}  // namespace rk
