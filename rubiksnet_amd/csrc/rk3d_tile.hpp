// rk3d_tile.hpp -- RubiksShift3D on 14x14 planes (fp32, stride 1 / pad 0): LDS-DMA streaming, one WAVE per (n, c).
//
// 14x14 is 35 of RubiksNet-Large's 51 shift layers.  A plane is 784 B: too small to be a
// workgroup's unit of streaming (tools/stream_pattern_probe.hip: row bands below ~6 KB collapse to 1.4-3.4 TB/s),
// W % 4 != 0 rules out the float4 cells of rk3d_dma.hpp, and the column kernels of rk3d_column.hpp (per-element
// global taps through L1/L2) sit at 2.1 TB/s.  Here the unit is one plane = 49 aligned 16-byte pieces, DMA'd straight
// into LDS (global_load_lds_dwordx4 nt, counted s_waitcnt vmcnt) by the wave that owns its (n, c) column; a
// 256-thread workgroup = 4 independent waves = 4 consecutive channels = 3.1 KB of contiguous traffic per plane.  A
// wave reads only what it DMA'd itself, so there is NO workgroup barrier anywhere.
//
// Work of a wave: walk t over its column (T + 1 steps).  A "round" is 64 lanes over 64 consecutive elements (4
// rounds per plane, the last one 4 lanes wide); the shift and the weights are wave-uniform and lane l's taps sit at
// consecutive LDS words.  Tap addresses are precomputed once; a tap outside the plane points at the slot's own zero
// word (no value masking).  The ring of 3 slots is UNROLLED INTO THE CODE: the slot of a step is a compile-time
// constant, so a tap is one ds_read_b32 with the slot in its immediate offset, the counted waits are literals and there
// is no ring bookkeeping -- the kernels turned out to be bound by instructions issued per step, not by bytes in flight
// (rings of 3..6 slots: identical times; the generic version with a runtime ring, a fifo of issue counts and a 33-way
// s_waitcnt switch: ~140 instructions per step and 16 % slower).  Rounds are processed as packed pairs (v_pk_mul_f32
// / v_pk_add_f32: IEEE per component, so the results are those of the scalar tree).  The H/W-interpolated field of a
// source plane stays in registers for the next step, as in rk3d_dma.hpp.
//
// Arithmetic: the reference's expression trees (rubiks3d_kernels.cu:193-203, :914-924), contraction off ->
// y and d(x) bit-identical to the oracle; d(shift) in the adjoint form of rk3d_dma.hpp, one partial per (n, c)
// -> finalizer blocks of the same launch (rk_dma.hpp) or k3d_finalize.  A channel with an exactly-integer shift
// component takes the per-element helpers of rk3d_generic.hpp in the backward (lowered-index quirk :290-298).
// (Two and four channels per wave were measured 3-18 % slower: twice the waves win.  7x7 planes on this scheme --
// 16 or 8 channels per wave -- came out level with the column kernels, 74.6 vs 74.8 us fwd+bwd at [32,8,576,7,7];
// they stay there.)
#pragma once
#include "rk3d_dma.hpp"

namespace rk {
namespace tile3d {

using namespace dma;

struct TDims { int N, T, C; };

namespace t14 {
constexpr int kHW = 196, kTileB = 784, kZ = 784, kZWords = 32, kStride = kTileB + 4 * kZWords, kRC = 4, kPieces = 49;
constexpr int kSlotsBwd = 7;          // LDS slots of a backward wave: 3 gy + 3 x + the staging plane of d(x)
template <int I> struct IC { static constexpr int value = I; };

__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <int OFFB> __device__ __forceinline__ float lds_at(unsigned a) {
    return *(__attribute__((address_space(3))) const float*)(size_t)(a + (unsigned)OFFB);
}
// A slot ends in kZWords = 32 zero words, one per LDS bank of a ds_read_b32 (bank = word mod 32 inside each half-wave): a lane
// whose tap lies outside the plane reads the zero word in the SAME BANK its in-plane address would have had.  The taps of
// the lanes of a half-wave are consecutive words (word = p + 14 flH + flW + const), i.e. 32 distinct banks, so a redirected
// lane never lands on a bank another lane of its instruction uses -- with ONE zero word per slot (rounds 2-5) every tap
// instruction that had an outside lane paid a 2-way conflict: SQ_LDS_BANK_CONFLICT was 18-50 % of the kernels' LDS cycles
// (profiles/r05_tile14_pmc.csv).  Slots are kStride = 912 bytes apart, the same bank shift for every lane.
__device__ __forceinline__ unsigned zero_word(unsigned slot0, int natural_word) {
    const int zw = (int)((slot0 + kZ) >> 2);
    return (unsigned)(zw + ((natural_word - zw) & (kZWords - 1))) << 2;
}
// absolute LDS byte addresses (slot 0) of the 4 taps of element p = lane + 64 * rc; outside the plane -> a zero word
__device__ __forceinline__ void taps(unsigned (&rel)[4], unsigned slot0, int flH, int flW, int lane, int rc) {
    const int p = lane + kWave * rc;
    const bool live = p < kHW;
    const int h = p / 14, w = p - h * 14;                             // (p >= kHW: rows 14..18 -- only the bank matters)
    const int h0 = h + flH, w0 = w + flW;
    const bool mh0 = (unsigned)h0 < 14u, mh1 = (unsigned)(h0 + 1) < 14u;
    const bool mw0 = (unsigned)w0 < 14u, mw1 = (unsigned)(w0 + 1) < 14u;
    const int wn = (int)(slot0 >> 2) + h0 * 14 + w0;                  // the word tap (0, 0) has, or would have
    const unsigned a = (unsigned)wn << 2;
    rel[0] = live && mh0 && mw0 ? a : zero_word(slot0, wn);
    rel[1] = live && mh0 && mw1 ? a + 4u : zero_word(slot0, wn + 1);
    rel[2] = live && mh1 && mw0 ? a + 56u : zero_word(slot0, wn + 14);
    rel[3] = live && mh1 && mw1 ? a + 60u : zero_word(slot0, wn + 15);
}
// every zero word of a wave's `slots` slots (8 float4 each), once per kernel
__device__ __forceinline__ void zero_regions(char* ring, int slots, int lane) {
    if (lane < 8 * slots) *reinterpret_cast<float4*>(ring + (lane >> 3) * kStride + kZ + (lane & 7) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// BN (training fusion, train_block.py): src holds z and the shift applies to max(a z + b, 0): the wave normalises the
// plane it DMA'd -- its own 49 pieces, in place, after its counted wait -- before it reads the taps.  A plane outside
// [0, T) is a slot of zeros and stays one.
template <bool NEGATE, bool BN = false>
__global__ __launch_bounds__(kBlock) void k3d_tile14_interp(const float* __restrict__ src, const float* __restrict__ shift,
                                                            float* __restrict__ dst, TDims d,
                                                            const float4* __restrict__ abmi = nullptr) {
    constexpr int R = 3;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (kWave - 1);
    const long long id = (long long)blockIdx.x * (kBlock / kWave) + wave;
    if (id >= (long long)d.N * d.C) return;                          // whole wave; no barriers in this kernel
    const int n = (int)(id / d.C), c = (int)(id - (long long)n * d.C);
    char* ring = lds_raw + wave * (R * kStride);
    zero_regions(ring, R, lane);
    float s0 = shift[c], s1 = shift[d.C + c], s2 = shift[2 * d.C + c];
    if (NEGATE) { s0 = -s0; s1 = -s1; s2 = -s2; }
    const Frac<float> fT = split_shift(s0), fH = split_shift(s1), fW = split_shift(s2);
    const float rT = uni(fT.r), rH = uni(fH.r), rW = uni(fW.r), uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const int f0 = __builtin_amdgcn_readfirstlane(fT.fl);
    const int flH = __builtin_amdgcn_readfirstlane(fH.fl), flW = __builtin_amdgcn_readfirstlane(fW.fl);
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const long long tstride = (long long)d.C * kHW;
    float bn_a = 0.f, bn_b = 0.f;
    if (BN) { const float4 pk = abmi[c]; bn_a = uni(pk.x); bn_b = uni(pk.y); }
    const float* col = src + ((size_t)n * d.T * d.C + c) * kHW;       // plane t = 0 of my column
    float* optr = dst + ((size_t)n * d.T * d.C + c) * kHW + lane;     // my element of the next output plane

    unsigned rel[kRC][4];
#pragma unroll
    for (int rc = 0; rc < kRC; ++rc) taps(rel[rc], ring_addr, flH, flW, lane, rc);

    int tf = f0;                                                     // next source plane to fetch ...
    const float* pf = col + (long long)f0 * tstride;                 // ... and where it lives (never dereferenced out of range)
    auto fetch = [&](int slot) {
        if ((unsigned)tf < (unsigned)d.T) {
            if (lane < kPieces) dma16s<true>(pf, lane * 16, ring_addr + slot * kStride);
        } else if (lane < kPieces) {
            *reinterpret_cast<float4*>(ring + slot * kStride + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        ++tf; pf += tstride;
    };
    const f32x2 uW2 = {uW, uW}, rW2 = {rW, rW}, uH2 = {uH, uH}, rH2 = {rH, rH}, uT2 = {uT, uT}, rT2 = {rT, rT};
    f32x2 Bprev[2] = {{0.f, 0.f}, {0.f, 0.f}};
    // VMEM order of a step j: [fetch of plane j+2][4 stores, j >= 1]; "plane k has landed" leaves outstanding the fetch of
    // plane k+1 (when it was in range: tf - 1 now) and the stores of steps k-2, k-1 (LATE: k >= 3).
    auto step = [&](auto SC, auto LATE, bool store) {
        constexpr int S = decltype(SC)::value;
        constexpr int W0 = decltype(LATE)::value ? 8 : 0;
        if ((unsigned)(tf - 1) < (unsigned)d.T) wait_vmcnt(W0 + 1); else wait_vmcnt(W0);
        if (BN && (unsigned)(tf - 2) < (unsigned)d.T && lane < kPieces) {      // the plane of this step: z -> relu(bn(z))
            float4* pc = reinterpret_cast<float4*>(ring + S * kStride + lane * 16);
            float4 v = *pc;
            v.x = fmaxf(fmaf(bn_a, v.x, bn_b), 0.f); v.y = fmaxf(fmaf(bn_a, v.y, bn_b), 0.f);
            v.z = fmaxf(fmaf(bn_a, v.z, bn_b), 0.f); v.w = fmaxf(fmaf(bn_a, v.w, bn_b), 0.f);
            *pc = v;
        }
        fetch((S + 2) % 3);                                          // into the slot of plane k-1
        f32x2 q[2][4];                                               // rounds (0,1) and (2,3) as packed pairs
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q[h][j].x = lds_at<S * kStride>(rel[2 * h][j]);
                q[h][j].y = lds_at<S * kStride>(rel[2 * h + 1][j]);
            }
        // all 16 reads in flight before the first multiply (the scheduler otherwise serialises them round by round)
        asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]), "+v"(q[1][0]), "+v"(q[1][1]),
                          "+v"(q[1][2]), "+v"(q[1][3]));
        f32x2 Bnew[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
            Bnew[h] = uH2 * (q[h][0] * uW2 + q[h][1] * rW2) + rH2 * (q[h][2] * uW2 + q[h][3] * rW2);
        if (store) {
            const f32x2 o0 = uT2 * Bprev[0] + rT2 * Bnew[0], o1 = uT2 * Bprev[1] + rT2 * Bnew[1];
            __builtin_nontemporal_store(o0.x, optr);
            __builtin_nontemporal_store(o0.y, optr + 64);
            __builtin_nontemporal_store(o1.x, optr + 128);
            if (lane < kHW - 192) __builtin_nontemporal_store(o1.y, optr + 192);
            optr += tstride;
        }
        Bprev[0] = Bnew[0]; Bprev[1] = Bnew[1];
    };
    fetch(0);
    fetch(1);
    int k = 0;
    do {                                                             // steps 0..2: nothing but fetches outstanding
        step(IC<0>{}, IC<0>{}, false); if (++k > d.T) break;
        step(IC<1>{}, IC<0>{}, true); if (++k > d.T) break;
        step(IC<2>{}, IC<0>{}, true); if (++k > d.T) break;
        for (;;) {
            step(IC<0>{}, IC<1>{}, true); if (++k > d.T) break;
            step(IC<1>{}, IC<1>{}, true); if (++k > d.T) break;
            step(IC<2>{}, IC<1>{}, true); if (++k > d.T) break;
        }
    } while (false);
}

// Backward of the same shape: d(x) (WRITE_GX) + the d(shift) partial of (n, c), adjoint form (rk3d_dma.hpp) -- gy taps
// with the negated shift from a ring of 3, my own x elements from a second ring of 3 (x[k-1] stays in registers).
// VMEM order of a step j: [gy plane j+2][x plane j+2][1 float4 store, j >= 1 with WRITE_GX].
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// BN (training fusion): x holds z; the activation is recomputed where the d(shift) sums use it, d(x) leaves masked by
// the ReLU and bn2's reduction sums ride along (rk3d_dma.hpp, dma_backward_loop).
// d(x) leaves as 49 float4 stores per plane: a lane computes elements l, l + 64, l + 128, l + 192 (consecutive LDS words per
// tap read), so the plane is turned through a wave-private LDS staging plane (4 ds_write_b32 + 1 ds_read_b128 per step,
// same wave, in-order LDS: no barrier) and lane l < 49 stores elements 4 l .. 4 l + 3 -- one VMEM store instruction per
// step instead of four 256-byte ones that each straddle three 128-byte lines (planes start at multiples of 784 bytes).
// Round 6, [32,8,288,14,14]: 36.8 -> 36.1 us; [32,8,216,14,14]: 29.0 -> 28.3 us (profiles/r06_tile14_variants.txt, with
// the variants that did NOT pay: an XCD-contiguous workgroup map, 4 waves per SIMD, the next fetch ahead of the wait).
template <bool WRITE_GX, bool FUSED, bool BN = false>
__global__ __launch_bounds__(kBlock) void k3d_tile14_backward(const float* __restrict__ x, const float* __restrict__ shift,
                                                              const float* __restrict__ gy, float* __restrict__ gx,
                                                              float* __restrict__ part, TDims d, Dims3 gd, dma3d::Fin3 fin,
                                                              dma3d::BnFuse bn = dma3d::BnFuse{}) {
    constexpr int ND = BN ? 5 : 3;
    if (FUSED && (int)blockIdx.x >= fin.f.producers) {                // row-sum + K5 inside the launch (rk_dma.hpp)
        if (threadIdx.x < kWave) dma3d::finalizer_wave<ND>(fin, (int)blockIdx.x - fin.f.producers, d.C, d.N, bn);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (kWave - 1);
    const long long id = (long long)blockIdx.x * (kBlock / kWave) + wave;
    if (id >= (long long)d.N * d.C) return;
    const int n = (int)(id / d.C), c = (int)(id - (long long)n * d.C);
    const size_t at = (size_t)c * ND * d.N + n;
    auto publish = [&](float a, float b, float w, float b1 = 0.f, float b2 = 0.f) {
        if (lane == 0) {
            if (FUSED) {
                fin_publish(fin.f, at, a); fin_publish(fin.f, at + d.N, b); fin_publish(fin.f, at + 2 * d.N, w);
                if (BN) { fin_publish(fin.f, at + 3 * (size_t)d.N, b1); fin_publish(fin.f, at + 4 * (size_t)d.N, b2); }
            } else { part[at] = a; part[at + d.N] = b; part[at + 2 * d.N] = w; }
        }
    };
    float4 bnp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BN) bnp = bn.abmi[c];
    const float s0 = shift[c], s1 = shift[d.C + c], s2 = shift[2 * d.C + c];
    const bool integer = split_shift(s0).r == 0 || split_shift(s1).r == 0 || split_shift(s2).r == 0;
    if (__builtin_amdgcn_readfirstlane((int)integer)) {
        // exactly-integer component: the reference's per-element formulation (lowered-index quirk :290-298)
        float aT = 0.f, aH = 0.f, aW = 0.f, aB1 = 0.f, aB2 = 0.f;
        const BnAct act{bnp.x, bnp.y};
        for (int t = 0; t < d.T; ++t) {
            if (WRITE_GX) backward_input_plane<float, false>(shift, gy, gx, gd, n, t, c, lane, kWave);
            if (BN) {
                shift_grad_plane<float>(x, shift, gy, gd, n, t, c, lane, kWave, aT, aH, aW, act);
                const size_t base = (((size_t)n * d.T + t) * d.C + c) * kHW;
                for (int e = lane; e < kHW; e += kWave) {             // (each lane re-reads the elements it wrote itself)
                    const float zv = x[base + e];
                    const float dz = fmaf(bnp.x, zv, bnp.y) > 0.f ? gx[base + e] : 0.f;
                    gx[base + e] = dz;
                    aB1 += dz;
                    aB2 = fmaf(dz, (zv - bnp.z) * bnp.w, aB2);
                }
            } else {
                shift_grad_plane<float>(x, shift, gy, gd, n, t, c, lane, kWave, aT, aH, aW);
            }
        }
        publish(wave_sum(aT), wave_sum(aH), wave_sum(aW), wave_sum(aB1), wave_sum(aB2));
        return;
    }
    char* ring = lds_raw + wave * (kSlotsBwd * kStride);              // slots 0..2: gy, 3..5: x, 6: the staging plane of d(x)
    zero_regions(ring, 6, lane);
    const Frac<float> fT = split_shift(-s0), fH = split_shift(-s1), fW = split_shift(-s2);   // fl', r' of the negated shift
    const float rT = uni(fT.r), rH = uni(fH.r), rW = uni(fW.r), uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const int f0 = __builtin_amdgcn_readfirstlane(fT.fl);
    const int flH = __builtin_amdgcn_readfirstlane(fH.fl), flW = __builtin_amdgcn_readfirstlane(fW.fl);
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring)), xaddr = gaddr + 3 * kStride;
    const long long tstride = (long long)d.C * kHW;
    const size_t col0 = ((size_t)n * d.T * d.C + c) * kHW;
    float* optr = WRITE_GX ? gx + col0 + 4 * lane : nullptr;           // my float4 of the next output plane
    float* stage = reinterpret_cast<float*>(ring + 6 * kStride);

    unsigned rel[kRC][4], xoff[kRC];
#pragma unroll
    for (int rc = 0; rc < kRC; ++rc) {
        taps(rel[rc], gaddr, flH, flW, lane, rc);
        const int p = lane + kWave * rc;
        xoff[rc] = p < kHW ? xaddr + (unsigned)(p * 4) : zero_word(xaddr, (int)(xaddr >> 2) + p);
    }
    int tg = f0, tx = 0;                                             // next gy / x planes to fetch
    const float* pg = gy + col0 + (long long)f0 * tstride;           // (never dereferenced out of range)
    const float* px = x + col0;
    auto fetch1 = [&](const float* p, int t, unsigned base, int slot) {
        if ((unsigned)t < (unsigned)d.T) {
            if (lane < kPieces) dma16s<true>(p, lane * 16, base + slot * kStride);
        } else if (lane < kPieces) {
            *reinterpret_cast<float4*>(ring + (base - gaddr) + slot * kStride + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto fetch = [&](int slot) {
        fetch1(pg, tg, gaddr, slot); ++tg; pg += tstride;
        fetch1(px, tx, xaddr, slot); ++tx; px += tstride;
    };
    const f32x2 uW2 = {uW, uW}, rW2 = {rW, rW}, uH2 = {uH, uH}, rH2 = {rH, rH}, uT2 = {uT, uT}, rT2 = {rT, rT};
    f32x2 Qprev[2] = {{0.f, 0.f}, {0.f, 0.f}}, xa[2] = {{0.f, 0.f}, {0.f, 0.f}};
    f32x2 aT = {0.f, 0.f}, aH = {0.f, 0.f}, aW = {0.f, 0.f};
    f32x2 za[2] = {{0.f, 0.f}, {0.f, 0.f}};                           // BN: raw z of the plane in xa
    float aB1 = 0.f, aB2 = 0.f;
    const float bn_a = uni(bnp.x), bn_b = uni(bnp.y), bn_m = uni(bnp.z), bn_i = uni(bnp.w);
    auto step = [&](auto SC, auto LATE, bool store) {
        constexpr int S = decltype(SC)::value;
        constexpr int W0 = (decltype(LATE)::value && WRITE_GX) ? 2 : 0;
        // outstanding behind "gy plane k and x plane k have landed": the fetches of step k-1 that were VMEM ops
        // (planes tg - 1, tx - 1 now) and the stores of steps k-2, k-1 (LATE: k >= 3; one store instruction per step)
        const int more = ((unsigned)(tg - 1) < (unsigned)d.T ? 1 : 0) + ((unsigned)(tx - 1) < (unsigned)d.T ? 1 : 0);
        if (more == 2) wait_vmcnt(W0 + 2); else if (more == 1) wait_vmcnt(W0 + 1); else wait_vmcnt(W0);
        const bool xb_real = (unsigned)(tx - 2) < (unsigned)d.T;     // BN: the x plane of this step exists
        f32x2 q[2][4], xb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            xb[h].x = lds_at<S * kStride>(xoff[2 * h]);
            xb[h].y = lds_at<S * kStride>(xoff[2 * h + 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                q[h][j].x = lds_at<S * kStride>(rel[2 * h][j]);
                q[h][j].y = lds_at<S * kStride>(rel[2 * h + 1][j]);
            }
        }
        fetch((S + 2) % 3);                                          // into the slots of planes k-1 (their reads are done)
        asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]), "+v"(q[1][0]), "+v"(q[1][1]),
                          "+v"(q[1][2]), "+v"(q[1][3]), "+v"(xb[0]), "+v"(xb[1]));
        f32x2 Qnew[2], xold[2], zold[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 zb = xb[h];
            if (BN) {                                                // activation of the plane (zero plane: zero activation)
                const float ea = xb_real ? bn_a : 0.f, eb = xb_real ? bn_b : 0.f;
                xb[h].x = fmaxf(fmaf(ea, zb.x, eb), 0.f);
                xb[h].y = fmaxf(fmaf(ea, zb.y, eb), 0.f);
            }
            const f32x2 la = q[h][0] * uW2 + q[h][1] * rW2, lb = q[h][2] * uW2 + q[h][3] * rW2;
            Qnew[h] = uH2 * la + rH2 * lb;                           // the reference's tree, contraction off
            const f32x2 c0 = fma2(uH2, q[h][0], rH2 * q[h][2]), c1 = fma2(uH2, q[h][1], rH2 * q[h][3]);
            const f32x2 dx = xb[h] - xa[h], mx = fma2(uT2, xb[h], rT2 * xa[h]);
            aT = fma2(Qnew[h], dx, aT);
            aH = fma2(la - lb, mx, aH);
            aW = fma2(c0 - c1, mx, aW);
            xold[h] = xa[h]; zold[h] = za[h];
            xa[h] = xb[h]; za[h] = zb;
        }
        if (WRITE_GX) {
            if (store) {
                f32x2 o0 = uT2 * Qprev[0] + rT2 * Qnew[0], o1 = uT2 * Qprev[1] + rT2 * Qnew[1];
                if (BN) {                                            // output plane k-1 = the plane xa held: mask + bn2's sums
                    const bool l3 = lane < kHW - 192;
                    o0.x = xold[0].x > 0.f ? o0.x : 0.f;  o0.y = xold[0].y > 0.f ? o0.y : 0.f;
                    o1.x = xold[1].x > 0.f ? o1.x : 0.f;  o1.y = (l3 && xold[1].y > 0.f) ? o1.y : 0.f;
                    aB1 += (o0.x + o0.y) + (o1.x + o1.y);
                    aB2 = fmaf(o0.x, (zold[0].x - bn_m) * bn_i, aB2);
                    aB2 = fmaf(o0.y, (zold[0].y - bn_m) * bn_i, aB2);
                    aB2 = fmaf(o1.x, (zold[1].x - bn_m) * bn_i, aB2);
                    aB2 = fmaf(o1.y, (zold[1].y - bn_m) * bn_i, aB2);
                }
                stage[lane] = o0.x; stage[lane + 64] = o0.y; stage[lane + 128] = o1.x;
                if (lane < kHW - 192) stage[lane + 192] = o1.y;
                if (lane < kPieces) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + 4 * lane);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(optr));
                }
                optr += tstride;
            }
            Qprev[0] = Qnew[0]; Qprev[1] = Qnew[1];
        }
    };
    fetch(0);
    fetch(1);
    int k = 0;
    do {
        step(IC<0>{}, IC<0>{}, false); if (++k > d.T) break;
        step(IC<1>{}, IC<0>{}, true); if (++k > d.T) break;
        step(IC<2>{}, IC<0>{}, true); if (++k > d.T) break;
        for (;;) {
            step(IC<0>{}, IC<1>{}, true); if (++k > d.T) break;
            step(IC<1>{}, IC<1>{}, true); if (++k > d.T) break;
            step(IC<2>{}, IC<1>{}, true); if (++k > d.T) break;
        }
    } while (false);
    publish(wave_sum(aT.x + aT.y), wave_sum(aH.x + aH.y), wave_sum(aW.x + aW.y), wave_sum(aB1), wave_sum(aB2));
}
}  // namespace t14

// ---------------------------------------------------------------------------------------------
// Host side.
inline bool s1p0(const Dims3& d) {
    return d.sT == 1 && d.sH == 1 && d.sW == 1 && d.pT == 0 && d.pH == 0 && d.pW == 0 && streaming_kernels_on();
}

// forward / d(x)-only; false = not handled here
template <bool NEGATE>
inline bool launch_interp(const float* src, const float* shift, float* dst, const Dims3& d, hipStream_t stream) {
    if (!s1p0(d) || !aligned16(src) || !aligned16(dst)) return false;
    if (d.H != 14 || d.W != 14) return false;
    TDims t{d.N, d.T, d.C};
    const unsigned grid = (unsigned)(((long long)d.N * d.C + 3) / 4);
    hipLaunchKernelGGL((t14::k3d_tile14_interp<NEGATE>), dim3(grid), dim3(kBlock), 4 * 3 * t14::kStride, stream, src, shift,
                       dst, t);
    return true;
}

// training fusion (BN): forward of relu(bn(z)) and its backward; false = not handled here
inline bool launch_forward_bn(const float* z, const float* shift, float* y, const float4* abmi, const Dims3& d,
                              hipStream_t stream) {
    if (!s1p0(d) || !aligned16(z) || !aligned16(y) || !aligned16(abmi)) return false;
    if (d.H != 14 || d.W != 14) return false;
    TDims t{d.N, d.T, d.C};
    const unsigned grid = (unsigned)(((long long)d.N * d.C + 3) / 4);
    hipLaunchKernelGGL((t14::k3d_tile14_interp<false, true>), dim3(grid), dim3(kBlock), 4 * 3 * t14::kStride, stream, z, shift,
                       y, t, abmi);
    return true;
}
inline bool launch_bwd_bn(const float* z, const float* shift, const float* gy, float* gx, float* gshift, float* ws,
                          const Dims3& d, int normalize, float t_factor, const dma3d::BnFuse& bn, hipStream_t stream) {
    if (!s1p0(d) || !aligned16(z) || !aligned16(gy) || !aligned16(gx) || !aligned16(bn.abmi)) return false;
    if (d.H != 14 || d.W != 14) return false;
    TDims t{d.N, d.T, d.C};
    const unsigned producers = (unsigned)(((long long)d.N * d.C + 3) / 4);
    const size_t lds = 4 * t14::kSlotsBwd * t14::kStride;
    dma3d::Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = (int)producers;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
    hipLaunchKernelGGL((t14::k3d_tile14_backward<true, true, true>), dim3(producers + d.C), dim3(kBlock), lds, stream, z, shift,
                       gy, gx, ws, t, d, fin, bn);
    return true;
}

// d(shift) (+ d(x) when gx != nullptr); gshift != nullptr: row-sum + K5 fused into the launch (ws = granules), else
// plain partials ws[C][3][P].  Returns P (0 = not handled here)
inline int launch_bwd(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws,
                      const Dims3& d, int normalize, float t_factor, hipStream_t stream) {
    if (!s1p0(d) || !aligned16(x) || !aligned16(gy) || (gx && !aligned16(gx))) return 0;
    if (d.H != 14 || d.W != 14) return 0;
    TDims t{d.N, d.T, d.C};
    const unsigned producers = (unsigned)(((long long)d.N * d.C + 3) / 4);
    const size_t lds = 4 * t14::kSlotsBwd * t14::kStride;
    dma3d::Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = (int)producers;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
#define RK_T14_BWD(GX, FU) hipLaunchKernelGGL((t14::k3d_tile14_backward<GX, FU>), dim3(producers + (FU ? d.C : 0)), \
                                              dim3(kBlock), lds, stream, x, shift, gy, gx, ws, t, d, fin)
    if (gshift) { if (gx) RK_T14_BWD(true, true); else RK_T14_BWD(false, true); }
    else { if (gx) RK_T14_BWD(true, false); else RK_T14_BWD(false, false); }
#undef RK_T14_BWD
    return d.N;
}

}  // namespace tile3d
}  // namespace rk
