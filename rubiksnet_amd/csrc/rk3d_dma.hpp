// rk3d_dma.hpp -- RubiksShift3D streaming kernels fed by LDS-DMA (gfx950 global_load_lds_dwordx4).
//
// Same maths and cell mapping as rk3d_stream.hpp; what changes is how planes reach the LDS.
// The register-staged kernels keep ONE plane per column in flight and pay for it in VGPRs, which
// caps a CU at ~50 KB of outstanding reads -- they measure latency-bound (waves wait ~60% of
// their cycles, HBM traffic == algorithmic bytes).  Here each wave DMAs its own 1 KiB chunks of
// a plane straight into a ring of R = D+1 linear LDS slots, D planes ahead, and only waits -- with
// a COUNTED s_waitcnt vmcnt(N), never a drain -- for the plane it is about to read.  hipcc neither
// sees the asm DMA nor waits for it, so the loads stay in flight across barriers
// (cdna_hip_programming.md 5.7 / "Pipelining across barriers").
//
// vmcnt bookkeeping (per wave, all wave-uniform): `issued` counts every VMEM instruction the wave
// has issued (DMA pieces and 16 B output stores, both = rounds-with-a-live-lane per plane);
// mark[j] is `issued` right after the DMA of the j-th plane ahead.  VMEM ops retire in order, so
// "plane k has landed" <=> outstanding <= issued - mark(k).  Under-counting `issued` only makes
// the wait stricter; it is never over-counted (a skipped round or plane is not counted).
//
// Slot layout: the plane's float4 cells in flat order (so the DMA destination is lane-linear), then
// ONE zero float4; every out-of-range (row, group) tap is redirected to that zero cell.
#pragma once
#include "rk3d_stream.hpp"

namespace rk {
namespace dma3d {

using stream3d::SDims;
using stream3d::lds_b128;
using stream3d::pick5;
using stream3d::wave_round_on;

// Output planes are written once and never read back by these kernels: store them non-temporal
// (global_store_dwordx4 ... nt) unless RK_NT_STORES=0 was set when the library was built.
#ifndef RK_NT_FWD
#define RK_NT_FWD 1
#endif
#ifndef RK_NT_BWD
#define RK_NT_BWD 1
#endif
// Input planes are read exactly once, by one CU: stream them too (MI355X_MICROARCH "nt-weights")
#ifndef RK_NT_LOADS_FWD
#define RK_NT_LOADS_FWD 1
#endif
#ifndef RK_NT_LOADS_BWD
#define RK_NT_LOADS_BWD 1
#endif
template <bool NT> __device__ __forceinline__ void stream_store(float4* p, const float4& v) {
    if (NT) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
    } else {
        *p = v;
    }
}

__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// one wave-instruction: lane l copies 16 B from its own `gsrc` to LDS byte address lds_dst + 16*l.
// M0 holds the LDS base; it is compiler-reserved, so it is saved, set and restored in ONE statement.
template <bool NT> __device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    if (NT)
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst_uniform)
            : "memory");
    else
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst_uniform)
            : "memory");
}

#define RK_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
// wait until at most n VMEM ops of this wave are outstanding (n wave-uniform; clamped down = stricter)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n < 0 ? 0 : (n > 32 ? 32 : n)) {
        RK_VMCNT_CASE(0) RK_VMCNT_CASE(1) RK_VMCNT_CASE(2) RK_VMCNT_CASE(3) RK_VMCNT_CASE(4)
        RK_VMCNT_CASE(5) RK_VMCNT_CASE(6) RK_VMCNT_CASE(7) RK_VMCNT_CASE(8) RK_VMCNT_CASE(9)
        RK_VMCNT_CASE(10) RK_VMCNT_CASE(11) RK_VMCNT_CASE(12) RK_VMCNT_CASE(13) RK_VMCNT_CASE(14)
        RK_VMCNT_CASE(15) RK_VMCNT_CASE(16) RK_VMCNT_CASE(17) RK_VMCNT_CASE(18) RK_VMCNT_CASE(19)
        RK_VMCNT_CASE(20) RK_VMCNT_CASE(21) RK_VMCNT_CASE(22) RK_VMCNT_CASE(23) RK_VMCNT_CASE(24)
        RK_VMCNT_CASE(25) RK_VMCNT_CASE(26) RK_VMCNT_CASE(27) RK_VMCNT_CASE(28) RK_VMCNT_CASE(29)
        RK_VMCNT_CASE(30) RK_VMCNT_CASE(31) RK_VMCNT_CASE(32)
    }
}
#undef RK_VMCNT_CASE

// Per-thread geometry in float4 units relative to a slot's start.
template <int ROUNDS> struct DCells {
    int cell[ROUNDS];   // clamped flat float4 index of the thread's cell
    bool live[ROUNDS];
    int a0[ROUNDS], a1[ROUNDS], b0[ROUNDS], b1[ROUNDS];   // tap groups (row A/B x group 0/1), zero cell if outside
};

template <int ROUNDS>
__device__ __forceinline__ void make_dcells(DCells<ROUNDS>& cs, const SDims& d, int flH, int group_shift) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int raw = (int)threadIdx.x + kBlock * i;
        cs.live[i] = raw < d.cells;
        const int cell = cs.live[i] ? raw : d.cells - 1;
        cs.cell[i] = cell;
        const int h = cell / d.W4, w4 = cell - h * d.W4;
        const int ra = h + flH, rb = ra + 1, ga = w4 + group_shift, gb = ga + 1;
        const bool ra_ok = ra >= 0 && ra < d.H, rb_ok = rb >= 0 && rb < d.H;
        const bool ga_ok = ga >= 0 && ga < d.W4, gb_ok = gb >= 0 && gb < d.W4;
        const int zero = d.cells;
        cs.a0[i] = (ra_ok && ga_ok) ? ra * d.W4 + ga : zero;
        cs.a1[i] = (ra_ok && gb_ok) ? ra * d.W4 + gb : zero;
        cs.b0[i] = (rb_ok && ga_ok) ? rb * d.W4 + ga : zero;
        cs.b1[i] = (rb_ok && gb_ok) ? rb * d.W4 + gb : zero;
    }
}

// number of rounds in which this wave has at least one live lane (= VMEM instructions per plane)
template <int ROUNDS> __device__ __forceinline__ int wave_rounds(int cells) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) n += wave_round_on(i, cells) ? 1 : 0;
    return n;
}

// DMA plane `plane` (global float pointer) into the slot whose LDS byte address is `slot_addr`
template <int ROUNDS, bool NEGATE>
__device__ __forceinline__ void dma_plane(const float* plane, unsigned slot_addr, const DCells<ROUNDS>& cs, int cells) {
    const unsigned wave = __builtin_amdgcn_readfirstlane((unsigned)threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        if (!wave_round_on(i, cells)) continue;
        const unsigned dst = slot_addr + (wave + 4u * i) * 1024u;       // chunk (wave + 4 i) of 64 cells
        if (cs.live[i]) dma16<(NEGATE ? RK_NT_LOADS_BWD : RK_NT_LOADS_FWD) != 0>(reinterpret_cast<const float4*>(plane) + cs.cell[i], dst);
    }
}

// ---------------------------------------------------------------------------------------------
// Forward (NEGATE = false: src = x, dst = y) and d(x) (NEGATE = true: src = gy, dst = gx).
template <bool NEGATE, int ROUNDS, int D>
__global__ __launch_bounds__(kBlock) void k3d_dma_interp(const float* __restrict__ src,
                                                         const float* __restrict__ shift,
                                                         float* __restrict__ dst, SDims d) {
    constexpr int R = D + 1;
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
    const int slot_f4 = d.cells + 1;                                 // float4 per slot (plane + zero cell)
    const int c = blockIdx.x % d.C, n = blockIdx.x / d.C;
    float sT = shift[c], sH = shift[d.C + c], sW = shift[2 * d.C + c];
    if (NEGATE) { sT = -sT; sH = -sH; sW = -sW; }
    const Frac<float> fT = split_shift(sT), fH = split_shift(sH), fW = split_shift(sW);
    const int HW = d.H * d.W;
    const size_t tstride = (size_t)d.C * HW;
    const float* sp = src + ((size_t)n * d.T * d.C + c) * HW;
    float* dp = dst + ((size_t)n * d.T * d.C + c) * HW;

    if (NEGATE && sT == 0 && sH == 0 && sW == 0) {                   // rubiks3d_kernels.cu:819-827
        for (int t = 0; t < d.T; ++t)
            for (int cell = threadIdx.x; cell < d.cells; cell += kBlock)
                reinterpret_cast<float4*>(dp + (size_t)t * tstride)[cell] =
                    reinterpret_cast<const float4*>(sp + (size_t)t * tstride)[cell];
        return;
    }

    const int off = ((fW.fl % 4) + 4) % 4;
    DCells<ROUNDS> cs;
    make_dcells<ROUNDS>(cs, d, fH.fl, (fW.fl - off) / 4);
    if (threadIdx.x < R) ring[threadIdx.x * slot_f4 + d.cells] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float rT = fT.r, rH = fH.r, rW = fW.r;
    const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const unsigned slot_bytes = (unsigned)slot_f4 * 16u;
    const int nr = wave_rounds<ROUNDS>(d.cells);

    float4 Bprev[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) Bprev[i] = make_float4(0.f, 0.f, 0.f, 0.f);

    const int t_first = fT.fl, steps = d.T + 1;                      // plane of step k is t_first + k
    auto in_range = [&](int t) { return t >= 0 && t < d.T; };

    int issued = 0;
    int mark[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {                                    // prologue: planes 0..D-1 -> slots 0..D-1
        if (in_range(t_first + j)) {
            dma_plane<ROUNDS, NEGATE>(sp + (size_t)(t_first + j) * tstride, ring_addr + j * slot_bytes, cs, d.cells);
            issued += nr;
        }
        mark[j] = issued;
    }

    int slot = 0;                                                     // slot of step k = k % R
    for (int k = 0; k < steps; ++k) {
        const int t = t_first + k;
        const bool valid = in_range(t);
        if (valid) wait_vmcnt(issued - mark[0]);                      // my pieces of plane k have landed
        __syncthreads();                                              // everyone's have; plane k-1 is retired
        {
            const int tn = t + D;
            int sn = slot + D; if (sn >= R) sn -= R;                  // = slot of plane k-1, free now
            if (in_range(tn)) {
                dma_plane<ROUNDS, NEGATE>(sp + (size_t)tn * tstride, ring_addr + sn * slot_bytes, cs, d.cells);
                issued += nr;
            }
#pragma unroll
            for (int j = 0; j + 1 < D; ++j) mark[j] = mark[j + 1];
            mark[D - 1] = issued;
        }
        const float4* cur = ring + slot * slot_f4;
        const int to = k - 1;                                         // = t - flT - 1
        const bool emit = to >= 0;                                    // to <= T-1 always
        float4* out = reinterpret_cast<float4*>(dp + (size_t)(emit ? to : 0) * tstride);
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            if (!wave_round_on(i, d.cells)) continue;
            float4 Bc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) {
                float a[5], b[5];
                pick5(lds_b128(cur + cs.a0[i]), lds_b128(cur + cs.a1[i]), off, a);
                pick5(lds_b128(cur + cs.b0[i]), lds_b128(cur + cs.b1[i]), off, b);
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
                stream_store<(NEGATE ? RK_NT_BWD : RK_NT_FWD) != 0>(out + cs.cell[i], o);
            }
            Bprev[i] = Bc;
        }
        if (emit) issued += nr;
        if (++slot == R) slot = 0;
    }
}

// compile-time pick5: the 5 consecutive values starting OFF floats into the aligned pair (q0, q1)
template <int OFF> __device__ __forceinline__ float tap(const float4& q0, const float4& q1, int k) {
    const int j = OFF + k;   // 0..7, constant after unrolling
    return j == 0 ? q0.x : j == 1 ? q0.y : j == 2 ? q0.z : j == 3 ? q0.w : j == 4 ? q1.x : j == 5 ? q1.y
         : j == 6 ? q1.z : q1.w;
}

// Geometry with the invariant the launchers guarantee (ROUNDS = ceil(cells / 256)): rounds
// 0 .. ROUNDS-2 are FULL (every lane of every wave owns a cell), only the last round is ragged.
// So only the tail round carries a wave-uniform "on" test and a lane mask.
template <int ROUNDS> struct TCells {
    int off16[ROUNDS];                                   // byte offset of the own cell inside a plane / slot
    int a0[ROUNDS], a1[ROUNDS], b0[ROUNDS], b1[ROUNDS];  // tap float4 indices (zero cell if outside)
    int xown;                                            // tail round: own float4 index, or the zero cell if dead
    bool tail_live;                                      // tail round: this lane owns a cell
    bool tail_on;                                        // tail round: this WAVE owns at least one cell (uniform)
};

template <int ROUNDS>
__device__ __forceinline__ void make_tcells(TCells<ROUNDS>& cs, const SDims& d, int flH, int group_shift) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int raw = (int)threadIdx.x + kBlock * i;
        const bool live = raw < d.cells;
        const int cell = live ? raw : d.cells - 1;
        cs.off16[i] = cell * 16;
        const int h = cell / d.W4, w4 = cell - h * d.W4;
        const int ra = h + flH, rb = ra + 1, ga = w4 + group_shift, gb = ga + 1;
        const bool ra_ok = ra >= 0 && ra < d.H, rb_ok = rb >= 0 && rb < d.H;
        const bool ga_ok = ga >= 0 && ga < d.W4, gb_ok = gb >= 0 && gb < d.W4;
        const int zero = d.cells;
        cs.a0[i] = (ra_ok && ga_ok) ? ra * d.W4 + ga : zero;
        cs.a1[i] = (ra_ok && gb_ok) ? ra * d.W4 + gb : zero;
        cs.b0[i] = (rb_ok && ga_ok) ? rb * d.W4 + ga : zero;
        cs.b1[i] = (rb_ok && gb_ok) ? rb * d.W4 + gb : zero;
        if (i == ROUNDS - 1) { cs.tail_live = live; cs.xown = live ? cell : zero; }
    }
    cs.tail_on = wave_round_on(ROUNDS - 1, d.cells);
}

#if RK_NT_LOADS_BWD
#define RK_BWD_LD_NT " nt"
#else
#define RK_BWD_LD_NT ""
#endif
// One wave-instruction of LDS-DMA in the saddr form: lane l copies 16 B from (sbase + voff_l) to LDS byte
// address lds_dst + 16*l.  An s_waitcnt lgkmcnt(0) in front orders it behind this wave's earlier LDS reads
// of the slot being refilled.
__device__ __forceinline__ void dma16s(const void* sbase_uniform, int voff, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "global_load_lds_dwordx4 %1, %2" RK_BWD_LD_NT "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform)
        : "memory");
}

// DMA one plane (uniform global pointer `plane`) into the slot at LDS byte address `slot_addr`.
// Returns the number of VMEM instructions this wave issued.
template <int ROUNDS>
__device__ __forceinline__ int dma_plane_t(const float* plane, unsigned slot_addr, const TCells<ROUNDS>& cs) {
    const unsigned wave = __builtin_amdgcn_readfirstlane((unsigned)threadIdx.x >> 6);
    const unsigned dst = slot_addr + wave * 1024u;                   // chunk (wave + 4 i) of 64 cells
#pragma unroll
    for (int i = 0; i + 1 < ROUNDS; ++i) dma16s(plane, cs.off16[i], dst + 4096u * i);
    if (cs.tail_on) {
        if (cs.tail_live) dma16s(plane, cs.off16[ROUNDS - 1], dst + 4096u * (ROUNDS - 1));
        return ROUNDS;
    }
    return ROUNDS - 1;
}

template <int ROUNDS>
__device__ __forceinline__ void zero_plane_t(float4* slot, const TCells<ROUNDS>& cs) {
    char* base = reinterpret_cast<char*>(slot);
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
        *reinterpret_cast<float4*>(base + cs.off16[i]) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------
// Backward, all loads by LDS-DMA: d(x) + d(shift) partials (WRITE_GX) or the partials alone.
// Maths = stream3d::k3d_stream_backward (adjoint form; gx in the reference's tree, bit-identical).
//   gy: ring of D+1 tap slots (plane tg needed whole for the taps while D more are in flight)
//   x : ring of D landing slots; a wave only ever reads back the cells it DMA'd itself (its own
//       aligned cells), into the 2-plane register window (x[to], x[to+1]) at the top of the step,
//       which frees the slot for the plane D steps ahead.
// The first version of this loop was instruction-issue bound (a lone workgroup needs ~3.5 us per
// step whatever the prefetch depth; 308 of ~650 VALU instructions were v_mov from a runtime tap
// selection, from merging "plane out of range -> 0" with computed values, and from masking).
// Hence: the tap offset (flW mod 4) is a template parameter (the kernel switches once into one
// of four copies of the loop); an out-of-range plane is a slot of zeros, filled with plain LDS
// stores where its DMA would have been issued, so the step body has no validity branches or
// masks; the full rounds carry no liveness tests at all; lanes past the end of the plane (tail
// round) read x from a zero cell, so they add nothing.
template <int ROUNDS, bool WRITE_GX, int D, int OFF>
__device__ __forceinline__ void dma_backward_loop(const float* __restrict__ xp, const float* __restrict__ gp,
                                                  float* __restrict__ op, float4* ring, const SDims& d,
                                                  const Frac<float>& fT, const Frac<float>& fH,
                                                  const Frac<float>& fW, size_t tstride, float& accT, float& accH,
                                                  float& accW) {
    constexpr int RG = D + 1, RX = D;
    const int slot_f4 = d.cells + 1;
    TCells<ROUNDS> cs;
    make_tcells<ROUNDS>(cs, d, fH.fl, (fW.fl - OFF) / 4);
    float4* const gring = ring;
    float4* const xring = ring + RG * slot_f4;
    if (threadIdx.x < RG + RX) ring[threadIdx.x * slot_f4 + d.cells] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float rT = fT.r, rH = fH.r, rW = fW.r;
    const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(gring));
    const unsigned xaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(xring));
    const unsigned slot_bytes = (unsigned)slot_f4 * 16u;

    float4 xa[ROUNDS], xb[ROUNDS], Qprev[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) xa[i] = xb[i] = Qprev[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float sT = 0.f, sH = 0.f, sW = 0.f;

    // step k: gy plane tg = t_first + k is in gy slot k % RG; to = k - 1; x[k] (-> xb) is in x slot k % RX
    const int t_first = fT.fl, steps = d.T + 1;
    auto in_range = [&](int t) { return t >= 0 && t < d.T; };
    int issued = 0;
    // bring plane (gy: tg, x: tx) into its slot: DMA when it exists, zeros when it does not
    auto feed = [&](int tg, int gs, int tx, int xs) {
        if (in_range(tg)) issued += dma_plane_t<ROUNDS>(gp + (size_t)tg * tstride, gaddr + gs * slot_bytes, cs);
        else zero_plane_t<ROUNDS>(gring + gs * slot_f4, cs);
        if (in_range(tx)) issued += dma_plane_t<ROUNDS>(xp + (size_t)tx * tstride, xaddr + xs * slot_bytes, cs);
        else zero_plane_t<ROUNDS>(xring + xs * slot_f4, cs);
    };
    int mark[D];                       // `issued` after the DMAs that feed step k+j (j = 0..D-1)
#pragma unroll
    for (int j = 0; j < D; ++j) { feed(t_first + j, j, j, j); mark[j] = issued; }

    // one round of one step: fields of the gy plane at this thread's cell meet x[to], x[to+1]
    auto round = [&](int i, const float4* cur, float4* out, bool store) {
        const float4 qa0 = lds_b128(cur + cs.a0[i]), qa1 = lds_b128(cur + cs.a1[i]);
        const float4 qb0 = lds_b128(cur + cs.b0[i]), qb1 = lds_b128(cur + cs.b1[i]);
        const float xav[4] = {xa[i].x, xa[i].y, xa[i].z, xa[i].w};
        const float xbv[4] = {xb[i].x, xb[i].y, xb[i].z, xb[i].w};
        float col[5], q[4];
#pragma unroll
        for (int m = 0; m < 5; ++m) col[m] = fmaf(uH, tap<OFF>(qa0, qa1, m), rH * tap<OFF>(qb0, qb1, m));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float la = tap<OFF>(qa0, qa1, m) * uW + tap<OFF>(qa0, qa1, m + 1) * rW;
            const float lb = tap<OFF>(qb0, qb1, m) * uW + tap<OFF>(qb0, qb1, m + 1) * rW;
            q[m] = uH * la + rH * lb;                             // the reference's tree, contraction off
            const float dx = xbv[m] - xav[m];
            const float mx = fmaf(uT, xbv[m], rT * xav[m]);
            sT = fmaf(q[m], dx, sT);
            sH = fmaf(la - lb, mx, sH);
            sW = fmaf(col[m] - col[m + 1], mx, sW);
        }
        if (WRITE_GX) {
            if (store) {
                float4 o;
                o.x = uT * Qprev[i].x + rT * q[0];
                o.y = uT * Qprev[i].y + rT * q[1];
                o.z = uT * Qprev[i].z + rT * q[2];
                o.w = uT * Qprev[i].w + rT * q[3];
                stream_store<RK_NT_BWD != 0>(reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + cs.off16[i]), o);
            }
            Qprev[i] = make_float4(q[0], q[1], q[2], q[3]);
        }
    };

    int gslot = 0, xslot = 0;
    for (int k = 0; k < steps; ++k) {
        wait_vmcnt(issued - mark[0]);                             // my pieces of gy(tg) and x[k] have landed
        __syncthreads();                                          // everyone's gy pieces have; step k-1 retired
        const char* xs = reinterpret_cast<const char*>(xring + xslot * slot_f4);
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) {                    // window: x[k-1], x[k]
            xa[i] = xb[i];
            xb[i] = *reinterpret_cast<const float4*>(xs + cs.off16[i]);
        }
        xa[ROUNDS - 1] = xb[ROUNDS - 1];
        xb[ROUNDS - 1] = reinterpret_cast<const float4*>(xs)[cs.xown];
        {
            int gs = gslot + D; if (gs >= RG) gs -= RG;           // gy slot of plane k-1: free now
            feed(t_first + k + D, gs, k + D, xslot);              // (its DMA waits for the LDS reads above)
#pragma unroll
            for (int j = 0; j + 1 < D; ++j) mark[j] = mark[j + 1];
            mark[D - 1] = issued;
        }
        const float4* cur = gring + gslot * slot_f4;
        const bool emit = WRITE_GX && k >= 1;                     // output plane to = k - 1
        float4* out = reinterpret_cast<float4*>(op + (size_t)(emit ? k - 1 : 0) * tstride);
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, out, emit);
        if (cs.tail_on) round(ROUNDS - 1, cur, out, emit && cs.tail_live);
        if (emit) issued += cs.tail_on ? ROUNDS : ROUNDS - 1;
        if (++gslot == RG) gslot = 0;
        if (++xslot == RX) xslot = 0;
    }
    accT = sT; accH = sH; accW = sW;
}

// (forcing <= 128 VGPRs with __launch_bounds__(256, 4) spills 32 B/lane and measures 13% slower)
template <int ROUNDS, bool WRITE_GX, int D>
__global__ __launch_bounds__(kBlock) void k3d_dma_backward(const float* __restrict__ x,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ gy,
                                                           float* __restrict__ gx,
                                                           float* __restrict__ part, SDims d, Dims3 gd) {
    extern __shared__ __attribute__((aligned(16))) float4 ring[];
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
        switch (((fW.fl % 4) + 4) % 4) {   // wave-uniform; one specialised copy of the loop per tap offset
            case 0: dma_backward_loop<ROUNDS, WRITE_GX, D, 0>(xp, gp, op, ring, d, fT, fH, fW, tstride, accT, accH, accW); break;
            case 1: dma_backward_loop<ROUNDS, WRITE_GX, D, 1>(xp, gp, op, ring, d, fT, fH, fW, tstride, accT, accH, accW); break;
            case 2: dma_backward_loop<ROUNDS, WRITE_GX, D, 2>(xp, gp, op, ring, d, fT, fH, fW, tstride, accT, accH, accW); break;
            default: dma_backward_loop<ROUNDS, WRITE_GX, D, 3>(xp, gp, op, ring, d, fT, fH, fW, tstride, accT, accH, accW); break;
        }
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

inline size_t ring_bytes(const SDims& s, int D) { return (size_t)(D + 1) * (s.cells + 1) * 16; }

template <bool NEGATE, int D>
inline void launch_interp_d(const float* src, const float* shift, float* dst, const SDims& s, hipStream_t stream) {
    const size_t lds = ring_bytes(s, D);
    const dim3 grid((unsigned)(s.N * s.C)), block(kBlock);
    switch ((s.cells + kBlock - 1) / kBlock) {
        case 1: hipLaunchKernelGGL((k3d_dma_interp<NEGATE, 1, D>), grid, block, lds, stream, src, shift, dst, s); break;
        case 2: hipLaunchKernelGGL((k3d_dma_interp<NEGATE, 2, D>), grid, block, lds, stream, src, shift, dst, s); break;
        case 3: hipLaunchKernelGGL((k3d_dma_interp<NEGATE, 3, D>), grid, block, lds, stream, src, shift, dst, s); break;
        default: hipLaunchKernelGGL((k3d_dma_interp<NEGATE, 4, D>), grid, block, lds, stream, src, shift, dst, s); break;
    }
}

inline int env_depth() {   // RK_DMA = 0 (register-staged kernels), 1..3 = planes in flight per column
    static const int v = [] { const char* e = getenv("RK_DMA"); return e ? atoi(e) : 2; }();
    return v;
}

template <bool NEGATE>
inline bool launch_interp(const float* src, const float* shift, float* dst, const SDims& s, hipStream_t stream) {
    switch (env_depth()) {
        case 1: launch_interp_d<NEGATE, 1>(src, shift, dst, s, stream); return true;
        case 2: launch_interp_d<NEGATE, 2>(src, shift, dst, s, stream); return true;
        case 3: launch_interp_d<NEGATE, 3>(src, shift, dst, s, stream); return true;
        default: return false;
    }
}

inline size_t bwd_ring_bytes(const SDims& s, int D) { return (size_t)(2 * D + 1) * (s.cells + 1) * 16; }

template <bool WRITE_GX, int D>
inline void launch_bwd_d(const float* x, const float* shift, const float* gy, float* gx, float* ws, const SDims& s,
                         const Dims3& d, hipStream_t stream) {
    const size_t lds = bwd_ring_bytes(s, D);
    const dim3 grid((unsigned)(s.N * s.C)), block(kBlock);
    switch ((s.cells + kBlock - 1) / kBlock) {
        case 1: hipLaunchKernelGGL((k3d_dma_backward<1, WRITE_GX, D>), grid, block, lds, stream, x, shift, gy, gx, ws, s, d); break;
        case 2: hipLaunchKernelGGL((k3d_dma_backward<2, WRITE_GX, D>), grid, block, lds, stream, x, shift, gy, gx, ws, s, d); break;
        case 3: hipLaunchKernelGGL((k3d_dma_backward<3, WRITE_GX, D>), grid, block, lds, stream, x, shift, gy, gx, ws, s, d); break;
        default: hipLaunchKernelGGL((k3d_dma_backward<4, WRITE_GX, D>), grid, block, lds, stream, x, shift, gy, gx, ws, s, d); break;
    }
}

inline int env_bwd_depth() {   // RK_DMA_BWD = 0 (register-staged kernel), 1..3 = planes in flight per stream
    static const int v = [] { const char* e = getenv("RK_DMA_BWD"); return e ? atoi(e) : 1; }();
    return v;
}

// d(shift) partials (+ d(x) when gx != nullptr); false = not handled here
inline bool launch_bwd(const float* x, const float* shift, const float* gy, float* gx, float* ws, const SDims& s,
                       const Dims3& d, hipStream_t stream) {
    const int D = env_bwd_depth();
    if (D < 1 || D > 3 || bwd_ring_bytes(s, D) + 64 > 160 * 1024) return false;
    if (gx) {
        if (D == 1) launch_bwd_d<true, 1>(x, shift, gy, gx, ws, s, d, stream);
        else if (D == 2) launch_bwd_d<true, 2>(x, shift, gy, gx, ws, s, d, stream);
        else launch_bwd_d<true, 3>(x, shift, gy, gx, ws, s, d, stream);
    } else {
        if (D == 1) launch_bwd_d<false, 1>(x, shift, gy, gx, ws, s, d, stream);
        else if (D == 2) launch_bwd_d<false, 2>(x, shift, gy, gx, ws, s, d, stream);
        else launch_bwd_d<false, 3>(x, shift, gy, gx, ws, s, d, stream);
    }
    return true;
}

}  // namespace dma3d
}  // namespace rk
