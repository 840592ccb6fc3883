// rk3d_tile.hpp -- RubiksShift3D on SMALL planes (14x14, 7x7; fp32, stride 1 / pad 0): LDS-DMA streaming with
// one WAVE per group of consecutive channels.
//
// 14x14 is 35 of RubiksNet-Large's 51 shift layers.  Their planes are 784 B / 196 B: too
// small to be a workgroup's unit of streaming (tools/stream_pattern_probe.hip: row bands below ~6 KB collapse to
// 1.4-3.4 TB/s), W % 4 != 0 rules out the float4 cells of rk3d_dma.hpp, and the column kernels of rk3d_column.hpp
// (per-element global taps through L1/L2) sit at 2.1 TB/s.  But in [N,T,C,H,W] the planes of consecutive channels
// of one (n,t) are contiguous, so here the unit is a TILE = GC consecutive channels x one plane (14x14: GC = 1,
// 784 B = 49 aligned 16-byte pieces; GC = 2 measured 3-18 % slower: twice the waves win), DMA'd straight into LDS
// (global_load_lds_dwordx4 nt, counted s_waitcnt vmcnt) by the wave that owns it, and a 256-thread workgroup = 4
// independent waves = 4 consecutive channels = 3.1 KB of contiguous traffic per plane.  A wave reads only what it DMA'd itself, so there is NO workgroup
// barrier anywhere.
//
// Work of a wave: walk t over its tile column (T + 1 steps).  A "round" is 64 lanes over 64 consecutive elements
// of ONE channel (14x14: 4 rounds per channel, 8 per step), so the channel's shift and weights are wave-uniform
// inside a round and lane l's taps sit at consecutive LDS words.  Tap addresses are precomputed relative to a
// ring slot; a tap outside the plane points at the slot's own zero word (no value masking), so a step is "add the
// slot base, 4 ds_read_b32, the reference's tree", every round of the step in one straight-line block.  The
// H/W-interpolated field of a source plane is kept in registers for the next step (one float per round), as in
// rk3d_dma.hpp; channels of one tile may have different floor(shift_T), so the LDS ring is a sliding window of
// source planes [f0 + k, f0 + k + 1] (+1 in flight) and a tile whose temporal floors span more than two values
// is walked again for the remaining channels (never for U(-1,1)-initialised shifts).
//
// Arithmetic: the reference's expression trees (rubiks3d_kernels.cu:193-203, :914-924), contraction off ->
// y and d(x) bit-identical to the oracle; d(shift) in the adjoint form of rk3d_dma.hpp, one partial per (n, c)
// -> k3d_finalize.  Channels with an exactly-integer shift component use the per-element helpers of
// rk3d_generic.hpp for d(x) / d(shift) (lowered-index quirk :290-298), one wave per channel column.
#pragma once
#include "rk3d_dma.hpp"

namespace rk {
namespace tile3d {

using namespace dma;

struct TDims {
    int N, T, C;
    int NG;          // channel groups per clip: ceil(C / GC)
};

template <int H_, int W_, int GC_ = 0> struct Geo {
    static constexpr int H = H_, W = W_, HW = H_ * W_;
    static constexpr int RC = (HW + kWave - 1) / kWave;          // rounds per channel
    static constexpr int GC = GC_ ? GC_ : ((16 / RC) < 1 ? 1 : (16 / RC));   // channels per wave (default: 16 rounds per step)
    static constexpr int ROUNDS = GC * RC;
    static constexpr int TILE_F = GC * HW;                       // floats per tile
    static constexpr int TILE_B = TILE_F * 4;
    static constexpr int PIECE_ROUNDS = (TILE_B / 16 + kWave - 1) / kWave;
    static_assert(TILE_B % 16 == 0, "a tile is a whole number of 16-byte pieces");
};

// per-channel constants, one 32-byte record per channel of the tile, in the wave's LDS
struct alignas(16) Chan {
    float rT, rH, rW;
    int flT;
    int flH, flW;
    int state;        // 0 = not in this tensor (c >= C), 1 = streamed, 2 = exactly-integer component (per-element path)
    int pad;
};

// One wave: DMA `npieces` 16-byte pieces of a tile (uniform global pointer) to LDS byte address dst; returns the
// number of VMEM instructions issued.
template <typename G>
__device__ __forceinline__ int dma_tile(const float* tile, unsigned dst, int npieces, int lane) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < G::PIECE_ROUNDS; ++i) {
        if (kWave * i < npieces) {                                   // wave-uniform
            if (lane + kWave * i < npieces) dma16s<true>(tile, (lane + kWave * i) * 16, dst + 1024u * i);
            ++n;
        }
    }
    return n;
}
template <typename G> __device__ __forceinline__ void zero_tile(char* slot, int lane) {
#pragma unroll
    for (int i = 0; i < G::PIECE_ROUNDS; ++i)
        if (lane + kWave * i < G::TILE_B / 16)
            *reinterpret_cast<float4*>(slot + (lane + kWave * i) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ float lds_word(unsigned byte_addr) {
    return *(__attribute__((address_space(3))) const float*)(size_t)byte_addr;
}

// A slot of the wave's LDS ring: [tile: TILE_B bytes][16 zero bytes].  Tap addresses are kept RELATIVE to the slot:
// a tap outside its plane points at the slot's own zero word, so the step body is "add the slot base, read".
template <typename G> struct SlotGeo {
    static constexpr int STRIDE = G::TILE_B + 16;
    static constexpr unsigned ZOFF = G::TILE_B;
};

// relative byte offsets of the 4 taps of my element in round (g, rc)
template <typename G>
__device__ __forceinline__ void make_taps(unsigned (&rel)[4], const Chan& ch, bool active, int g, int lane, int rc) {
    const int p = lane + kWave * rc;
    const bool live = active && p < G::HW;
    const int pc = p < G::HW ? p : 0;
    const int h = pc / G::W, w = pc - h * G::W;
    const int h0 = h + ch.flH, w0 = w + ch.flW;
    const bool mh0 = (unsigned)h0 < (unsigned)G::H, mh1 = (unsigned)(h0 + 1) < (unsigned)G::H;
    const bool mw0 = (unsigned)w0 < (unsigned)G::W, mw1 = (unsigned)(w0 + 1) < (unsigned)G::W;
    const unsigned a = (unsigned)((g * G::HW + h0 * G::W + w0) * 4);
    constexpr unsigned Z = SlotGeo<G>::ZOFF;
    rel[0] = live && mh0 && mw0 ? a : Z;
    rel[1] = live && mh0 && mw1 ? a + 4u : Z;
    rel[2] = live && mh1 && mw0 ? a + 4u * G::W : Z;
    rel[3] = live && mh1 && mw1 ? a + 4u * G::W + 4u : Z;
}

// Fill the wave's channel table; returns {min, max} of floor(shift_T) over the streamed channels (min > max: none).
template <typename G, bool NEGATE, bool BACKWARD>
__device__ __forceinline__ void make_chans(Chan* tab, const float* __restrict__ shift, int C, int c0, int lane,
                                           int& fmin, int& fmax) {
    int lo = 0x7fffffff, hi = -0x7fffffff - 1;
    if (lane < G::GC) {
        Chan ch;
        const int c = c0 + lane;
        ch.state = 0; ch.pad = 0;
        ch.rT = ch.rH = ch.rW = 0.f; ch.flT = ch.flH = ch.flW = 0;
        if (c < C) {
            const float s0 = shift[c], s1 = shift[C + c], s2 = shift[2 * C + c];
            const bool integer = split_shift(s0).r == 0 || split_shift(s1).r == 0 || split_shift(s2).r == 0;
            const Frac<float> fT = split_shift(NEGATE ? -s0 : s0), fH = split_shift(NEGATE ? -s1 : s1),
                              fW = split_shift(NEGATE ? -s2 : s2);
            ch.rT = fT.r; ch.rH = fH.r; ch.rW = fW.r;
            ch.flT = fT.fl; ch.flH = fH.fl; ch.flW = fW.fl;
            ch.state = (BACKWARD && integer) ? 2 : 1;
            if (ch.state == 1) { lo = hi = fT.fl; }
        }
        tab[lane] = ch;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // table visible to every lane of the wave
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o, kWave));
        hi = max(hi, __shfl_xor(hi, o, kWave));
    }
    fmin = __builtin_amdgcn_readfirstlane(lo);
    fmax = __builtin_amdgcn_readfirstlane(hi);
}

struct Item { int n, c0, nch; bool live; };
template <typename G> __device__ __forceinline__ Item my_item(const TDims& d, int wave) {
    Item it;
    const long long id = (long long)blockIdx.x * (kBlock / kWave) + wave;
    it.live = id < (long long)d.N * d.NG;
    const long long q = it.live ? id : 0;
    const int cg = (int)(q % d.NG);
    it.n = (int)(q / d.NG);
    it.c0 = cg * G::GC;
    it.nch = d.C - it.c0 < G::GC ? d.C - it.c0 : G::GC;
    return it;
}

// LDS of one wave: [channel table][slots ...]
template <typename G> constexpr int wave_lds_bytes(int slots) { return G::GC * (int)sizeof(Chan) + slots * SlotGeo<G>::STRIDE; }

// per-channel uniforms of one pass (SGPRs)
struct ChanU { float rT, rH, rW, uT, uH, uW; int rel; bool on; };
template <typename G>
__device__ __forceinline__ void chan_uniforms(ChanU (&cu)[G::GC], const Chan* tab, int f0) {
#pragma unroll
    for (int g = 0; g < G::GC; ++g) {
        const Chan ch = tab[g];
        cu[g].rT = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ch.rT)));
        cu[g].rH = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ch.rH)));
        cu[g].rW = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ch.rW)));
        cu[g].uT = 1 - cu[g].rT; cu[g].uH = 1 - cu[g].rH; cu[g].uW = 1 - cu[g].rW;
        cu[g].rel = __builtin_amdgcn_readfirstlane(ch.flT) - f0;
        cu[g].on = __builtin_amdgcn_readfirstlane(ch.state) == 1 && (cu[g].rel == 0 || cu[g].rel == 1);
    }
}

// ---------------------------------------------------------------------------------------------
// Forward (NEGATE = false: src = x, dst = y) and d(x) alone (NEGATE = true: src = gy, dst = gx).
template <int H, int W, int GCO, int kRing, bool NEGATE>
__global__ __launch_bounds__(kBlock) void k3d_tile_interp(const float* __restrict__ src, const float* __restrict__ shift,
                                                          float* __restrict__ dst, TDims d) {
    using G = Geo<H, W, GCO>;
    using SG = SlotGeo<G>;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (kWave - 1);
    const Item it = my_item<G>(d, wave);
    if (!it.live) return;                                            // whole wave; no barriers in this kernel
    char* mine = lds_raw + wave * wave_lds_bytes<G>(kRing);
    Chan* tab = reinterpret_cast<Chan*>(mine);
    char* ring = mine + G::GC * sizeof(Chan);
    if (lane < kRing) *reinterpret_cast<float4*>(ring + lane * SG::STRIDE + SG::ZOFF) = make_float4(0.f, 0.f, 0.f, 0.f);
    int fmin, fmax;
    make_chans<G, NEGATE, false>(tab, shift, d.C, it.c0, lane, fmin, fmax);
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const size_t tstride = (size_t)d.C * G::HW;
    const float* col = src + ((size_t)it.n * d.T * d.C + it.c0) * G::HW;      // tile of plane t = 0
    char* ocol = reinterpret_cast<char*>(dst + ((size_t)it.n * d.T * d.C + it.c0) * G::HW);
    const int npieces = it.nch * G::HW / 4;

    unsigned rel[G::ROUNDS][4];                                      // tap offsets inside a slot
    int ooff[G::ROUNDS];                                             // byte offset of my output inside a tile, -1 = none
#pragma unroll
    for (int g = 0; g < G::GC; ++g)
#pragma unroll
        for (int rc = 0; rc < G::RC; ++rc) {
            const Chan ch = tab[g];
            make_taps<G>(rel[g * G::RC + rc], ch, ch.state == 1, g, lane, rc);
            const int p = lane + kWave * rc;
            ooff[g * G::RC + rc] = (ch.state == 1 && p < G::HW) ? (g * G::HW + p) * 4 : -1;
        }

    for (int f0 = fmin; f0 <= fmax; f0 += 2) {                        // one pass unless the temporal floors spread
        ChanU cu[G::GC];
        chan_uniforms<G>(cu, tab, f0);
        float Bprev[G::ROUNDS];
#pragma unroll
        for (int r = 0; r < G::ROUNDS; ++r) Bprev[r] = 0.f;
        int issued = 0;
        auto fetch = [&](int j) {                                    // source plane f0 + j -> slot j % kRing
            const int t = f0 + j, s = j % kRing;
            if (t >= 0 && t < d.T) issued += dma_tile<G>(col + (size_t)t * tstride, ring_addr + s * SG::STRIDE, npieces, lane);
            else zero_tile<G>(ring + s * SG::STRIDE, lane);
        };
        // Planes f0+k and f0+k+1 are read at step k while planes up to f0+k+kRing-1 are in flight.  fifo[i] is
        // `issued` right after the fetch of plane f0+k+1+i, so "plane f0+k+1 has landed" <=> outstanding VMEM ops
        // <= issued - fifo[0] (VMEM retires in order).
        constexpr int S = kRing - 2;
        int fifo[S];
        fetch(0);
#pragma unroll
        for (int i = 0; i < S; ++i) { fetch(1 + i); fifo[i] = issued; }
        for (int k = 0; k <= d.T; ++k) {
            wait_vmcnt(issued - fifo[0]);
            fetch(k + kRing - 1);                                    // into the slot of plane f0+k-1
#pragma unroll
            for (int i = 0; i + 1 < S; ++i) fifo[i] = fifo[i + 1];
            fifo[S - 1] = issued;
            const unsigned slot_lo = ring_addr + (k % kRing) * SG::STRIDE, slot_hi = ring_addr + ((k + 1) % kRing) * SG::STRIDE;
            // phase A: every round's field, straight-line (inactive channels compute on whatever their taps hit)
            float Bnew[G::ROUNDS];
#pragma unroll
            for (int g = 0; g < G::GC; ++g) {
                const unsigned sb = cu[g].rel == 1 ? slot_hi : slot_lo;
#pragma unroll
                for (int rc = 0; rc < G::RC; ++rc) {
                    const int r = g * G::RC + rc;
                    const float q00 = lds_word(rel[r][0] + sb), q01 = lds_word(rel[r][1] + sb);
                    const float q10 = lds_word(rel[r][2] + sb), q11 = lds_word(rel[r][3] + sb);
                    Bnew[r] = cu[g].uH * (q00 * cu[g].uW + q01 * cu[g].rW) + cu[g].rH * (q10 * cu[g].uW + q11 * cu[g].rW);
                }
            }
            // phase B: blend with the previous plane's field and store
            if (k >= 1) {
                char* out = ocol + (size_t)(k - 1) * tstride * 4;
#pragma unroll
                for (int g = 0; g < G::GC; ++g) {
                    if (!cu[g].on) continue;                         // wave-uniform
#pragma unroll
                    for (int rc = 0; rc < G::RC; ++rc) {
                        const int r = g * G::RC + rc;
                        if (ooff[r] >= 0)
                            __builtin_nontemporal_store(cu[g].uT * Bprev[r] + cu[g].rT * Bnew[r],
                                                        reinterpret_cast<float*>(out + ooff[r]));
                        ++issued;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < G::ROUNDS; ++r) Bprev[r] = Bnew[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) (WRITE_GX) + d(shift) partials part[c][3][P = N], p = n.
template <int H, int W, int GCO, int kRing, bool WRITE_GX, bool FUSED>
__global__ __launch_bounds__(kBlock) void k3d_tile_backward(const float* __restrict__ x, const float* __restrict__ shift,
                                                            const float* __restrict__ gy, float* __restrict__ gx,
                                                            float* __restrict__ part, TDims d, Dims3 gd, dma3d::Fin3 fin) {
    using G = Geo<H, W, GCO>;
    if (FUSED && (int)blockIdx.x >= fin.f.producers) {                // row-sum + K5 inside the launch (rk_dma.hpp)
        if (threadIdx.x < kWave) dma3d::finalizer_wave(fin, (int)blockIdx.x - fin.f.producers, d.C, d.N);
        return;
    }
    using SG = SlotGeo<G>;
    constexpr int S = kRing - 2;                                     // x runs S planes ahead, like gy beyond f0+k+1
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (kWave - 1);
    const Item it = my_item<G>(d, wave);
    if (!it.live) return;
    char* mine = lds_raw + wave * wave_lds_bytes<G>(kRing + S);
    Chan* tab = reinterpret_cast<Chan*>(mine);
    char* gring = mine + G::GC * sizeof(Chan);
    char* xring = gring + kRing * SG::STRIDE;
    if (lane < kRing + S) *reinterpret_cast<float4*>(gring + lane * SG::STRIDE + SG::ZOFF) = make_float4(0.f, 0.f, 0.f, 0.f);
    int fmin, fmax;
    make_chans<G, true, true>(tab, shift, d.C, it.c0, lane, fmin, fmax);      // fl', r' of the NEGATED shift
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(gring));
    const unsigned xaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(xring));
    const size_t tstride = (size_t)d.C * G::HW;
    const size_t col0 = ((size_t)it.n * d.T * d.C + it.c0) * G::HW;
    const float* gcol = gy + col0;
    const float* xcol = x + col0;
    char* ocol = WRITE_GX ? reinterpret_cast<char*>(gx + col0) : nullptr;
    const int npieces = it.nch * G::HW / 4;

    unsigned rel[G::ROUNDS][4];
    unsigned xoff[G::ROUNDS];                                        // my own element inside an x slot (or its zero word)
    int ooff[G::ROUNDS];
#pragma unroll
    for (int g = 0; g < G::GC; ++g)
#pragma unroll
        for (int rc = 0; rc < G::RC; ++rc) {
            const Chan ch = tab[g];
            const int r = g * G::RC + rc, p = lane + kWave * rc;
            make_taps<G>(rel[r], ch, ch.state == 1, g, lane, rc);
            const bool mine_ = ch.state == 1 && p < G::HW;
            ooff[r] = mine_ ? (g * G::HW + p) * 4 : -1;
            xoff[r] = mine_ ? (unsigned)((g * G::HW + p) * 4) : SG::ZOFF;
        }

    float sT[G::GC], sH[G::GC], sW[G::GC];
#pragma unroll
    for (int g = 0; g < G::GC; ++g) sT[g] = sH[g] = sW[g] = 0.f;

    for (int f0 = fmin; f0 <= fmax; f0 += 2) {
        ChanU cu[G::GC];
        chan_uniforms<G>(cu, tab, f0);
        float xa[G::ROUNDS], xb[G::ROUNDS], Qprev[G::ROUNDS];
#pragma unroll
        for (int r = 0; r < G::ROUNDS; ++r) xa[r] = xb[r] = Qprev[r] = 0.f;
        int issued = 0;
        auto fetch_g = [&](int j) {                                  // gy plane f0 + j -> slot j % kRing
            const int t = f0 + j, s = j % kRing;
            if (t >= 0 && t < d.T) issued += dma_tile<G>(gcol + (size_t)t * tstride, gaddr + s * SG::STRIDE, npieces, lane);
            else zero_tile<G>(gring + s * SG::STRIDE, lane);
        };
        auto fetch_x = [&](int k) {                                  // x plane k -> x slot k % S
            if (k >= 0 && k < d.T) issued += dma_tile<G>(xcol + (size_t)k * tstride, xaddr + (k % S) * SG::STRIDE, npieces, lane);
            else zero_tile<G>(xring + (k % S) * SG::STRIDE, lane);
        };
        // issue groups {gy plane f0+j+1, x plane j}: step k needs exactly group k; fifo[i] = `issued` after group k+i
        int fifo[S];
        fetch_g(0);
#pragma unroll
        for (int i = 0; i < S; ++i) { fetch_g(1 + i); fetch_x(i); fifo[i] = issued; }
        for (int k = 0; k <= d.T; ++k) {
            wait_vmcnt(issued - fifo[0]);                            // gy planes f0+k, f0+k+1 and x[k] have landed
            const unsigned xs = xaddr + (k % S) * SG::STRIDE;
#pragma unroll
            for (int r = 0; r < G::ROUNDS; ++r) {                    // window x[k-1], x[k] at my own elements
                xa[r] = xb[r];
                xb[r] = lds_word(xoff[r] + xs);
            }
            fetch_g(k + kRing - 1);
            fetch_x(k + S);                                          // (the DMA waits for the LDS reads above)
#pragma unroll
            for (int i = 0; i + 1 < S; ++i) fifo[i] = fifo[i + 1];
            fifo[S - 1] = issued;
            const unsigned slot_lo = gaddr + (k % kRing) * SG::STRIDE, slot_hi = gaddr + ((k + 1) % kRing) * SG::STRIDE;
            float Qnew[G::ROUNDS];
#pragma unroll
            for (int g = 0; g < G::GC; ++g) {
                const unsigned sb = cu[g].rel == 1 ? slot_hi : slot_lo;
                const float rT = cu[g].rT, rH = cu[g].rH, rW = cu[g].rW, uT = cu[g].uT, uH = cu[g].uH, uW = cu[g].uW;
                float aT = 0.f, aH = 0.f, aW = 0.f;
#pragma unroll
                for (int rc = 0; rc < G::RC; ++rc) {
                    const int r = g * G::RC + rc;
                    const float q00 = lds_word(rel[r][0] + sb), q01 = lds_word(rel[r][1] + sb);
                    const float q10 = lds_word(rel[r][2] + sb), q11 = lds_word(rel[r][3] + sb);
                    const float la = q00 * uW + q01 * rW, lb = q10 * uW + q11 * rW;
                    const float q = uH * la + rH * lb;                // the reference's tree, contraction off
                    const float c0 = fmaf(uH, q00, rH * q10), c1 = fmaf(uH, q01, rH * q11);
                    const float dx = xb[r] - xa[r];
                    const float mx = fmaf(uT, xb[r], rT * xa[r]);
                    aT = fmaf(q, dx, aT);
                    aH = fmaf(la - lb, mx, aH);
                    aW = fmaf(c0 - c1, mx, aW);
                    Qnew[r] = q;
                }
                if (cu[g].on) { sT[g] += aT; sH[g] += aH; sW[g] += aW; }   // wave-uniform
            }
            if (WRITE_GX && k >= 1) {
                char* out = ocol + (size_t)(k - 1) * tstride * 4;
#pragma unroll
                for (int g = 0; g < G::GC; ++g) {
                    if (!cu[g].on) continue;
#pragma unroll
                    for (int rc = 0; rc < G::RC; ++rc) {
                        const int r = g * G::RC + rc;
                        if (ooff[r] >= 0)
                            __builtin_nontemporal_store(cu[g].uT * Qprev[r] + cu[g].rT * Qnew[r],
                                                        reinterpret_cast<float*>(out + ooff[r]));
                        ++issued;
                    }
                }
            }
            if (WRITE_GX) {
#pragma unroll
                for (int r = 0; r < G::ROUNDS; ++r) Qprev[r] = Qnew[r];
            }
        }
    }

    // exactly-integer channels: the reference's per-element formulation (shared with rk3d_generic.hpp)
#pragma unroll 1
    for (int g = 0; g < it.nch; ++g) {
        if (__builtin_amdgcn_readfirstlane(tab[g].state) != 2) continue;
        const int c = it.c0 + g;
        float aT = 0.f, aH = 0.f, aW = 0.f;
        for (int t = 0; t < d.T; ++t) {
            if (WRITE_GX) backward_input_plane<float, false>(shift, gy, gx, gd, it.n, t, c, lane, kWave);
            shift_grad_plane<float>(x, shift, gy, gd, it.n, t, c, lane, kWave, aT, aH, aW);
        }
        aT = wave_sum(aT); aH = wave_sum(aH); aW = wave_sum(aW);
        if (lane == 0) {
            const size_t at = (size_t)c * 3 * d.N + it.n;
            if (FUSED) { fin_publish(fin.f, at, aT); fin_publish(fin.f, at + d.N, aH); fin_publish(fin.f, at + 2 * d.N, aW); }
            else { part[at] = aT; part[at + d.N] = aH; part[at + 2 * d.N] = aW; }
        }
    }
#pragma unroll
    for (int g = 0; g < G::GC; ++g) {
        const float a = wave_sum(sT[g]), b = wave_sum(sH[g]), w = wave_sum(sW[g]);
        if (lane == 0 && g < it.nch && tab[g].state == 1) {
            const size_t at = (size_t)(it.c0 + g) * 3 * d.N + it.n;
            if (FUSED) { fin_publish(fin.f, at, a); fin_publish(fin.f, at + d.N, b); fin_publish(fin.f, at + 2 * d.N, w); }
            else { part[at] = a; part[at + d.N] = b; part[at + 2 * d.N] = w; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 14x14, one channel per wave, ring of 3 with the step loop unrolled by the ring: the slot of a step is a compile-time
// constant, so a tap is ONE ds_read_b32 with the slot in its immediate offset (no per-tap address add), the counted
// waits are literals (VMEM order of a step: [fetch of plane k+2][4 stores]; "plane k has landed" leaves the 4 + 4
// stores of steps k-2, k-1 and the fetch of plane k+1 outstanding) and the ring bookkeeping (modulo, fifo of issue
// counts, the 33-way s_waitcnt switch) is gone: ~75 instead of ~140 instructions per step and wave.
namespace t14 {
constexpr int kHW = 196, kTileB = 784, kStride = 800, kZ = 784, kRC = 4, kPieces = 49;
template <int I> struct IC { static constexpr int value = I; };

__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
template <int OFFB> __device__ __forceinline__ float lds_at(unsigned a) {
    return *(__attribute__((address_space(3))) const float*)(size_t)(a + (unsigned)OFFB);
}
// absolute LDS byte addresses (slot 0) of the 4 taps of element p = lane + 64 * rc; outside the plane -> the zero word
__device__ __forceinline__ void taps(unsigned (&rel)[4], unsigned slot0, int flH, int flW, int lane, int rc) {
    const int p = lane + kWave * rc;
    const bool live = p < kHW;
    const int pc = live ? p : 0;
    const int h = pc / 14, w = pc - h * 14;
    const int h0 = h + flH, w0 = w + flW;
    const bool mh0 = (unsigned)h0 < 14u, mh1 = (unsigned)(h0 + 1) < 14u;
    const bool mw0 = (unsigned)w0 < 14u, mw1 = (unsigned)(w0 + 1) < 14u;
    const unsigned a = slot0 + (unsigned)((h0 * 14 + w0) * 4), Z = slot0 + kZ;
    rel[0] = live && mh0 && mw0 ? a : Z;
    rel[1] = live && mh0 && mw1 ? a + 4u : Z;
    rel[2] = live && mh1 && mw0 ? a + 56u : Z;
    rel[3] = live && mh1 && mw1 ? a + 60u : Z;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool NEGATE>
__global__ __launch_bounds__(kBlock) void k3d_tile14_interp(const float* __restrict__ src, const float* __restrict__ shift,
                                                            float* __restrict__ dst, TDims d) {
    constexpr int R = 3;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (kWave - 1);
    const long long id = (long long)blockIdx.x * (kBlock / kWave) + wave;
    if (id >= (long long)d.N * d.C) return;                          // whole wave; no barriers in this kernel
    const int n = (int)(id / d.C), c = (int)(id - (long long)n * d.C);
    char* ring = lds_raw + wave * (R * kStride);
    if (lane < R) *reinterpret_cast<float4*>(ring + lane * kStride + kZ) = make_float4(0.f, 0.f, 0.f, 0.f);
    float s0 = shift[c], s1 = shift[d.C + c], s2 = shift[2 * d.C + c];
    if (NEGATE) { s0 = -s0; s1 = -s1; s2 = -s2; }
    const Frac<float> fT = split_shift(s0), fH = split_shift(s1), fW = split_shift(s2);
    const float rT = uni(fT.r), rH = uni(fH.r), rW = uni(fW.r), uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const int f0 = __builtin_amdgcn_readfirstlane(fT.fl);
    const int flH = __builtin_amdgcn_readfirstlane(fH.fl), flW = __builtin_amdgcn_readfirstlane(fW.fl);
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));
    const long long tstride = (long long)d.C * kHW;
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
// VMEM order of a step j: [gy plane j+2][x plane j+2][4 stores, j >= 1 with WRITE_GX].
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

template <bool WRITE_GX, bool FUSED>
__global__ __launch_bounds__(kBlock) void k3d_tile14_backward(const float* __restrict__ x, const float* __restrict__ shift,
                                                              const float* __restrict__ gy, float* __restrict__ gx,
                                                              float* __restrict__ part, TDims d, Dims3 gd, dma3d::Fin3 fin) {
    if (FUSED && (int)blockIdx.x >= fin.f.producers) {                // row-sum + K5 inside the launch (rk_dma.hpp)
        if (threadIdx.x < kWave) dma3d::finalizer_wave(fin, (int)blockIdx.x - fin.f.producers, d.C, d.N);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (kWave - 1);
    const long long id = (long long)blockIdx.x * (kBlock / kWave) + wave;
    if (id >= (long long)d.N * d.C) return;
    const int n = (int)(id / d.C), c = (int)(id - (long long)n * d.C);
    const size_t at = (size_t)c * 3 * d.N + n;
    auto publish = [&](float a, float b, float w) {
        if (lane == 0) {
            if (FUSED) { fin_publish(fin.f, at, a); fin_publish(fin.f, at + d.N, b); fin_publish(fin.f, at + 2 * d.N, w); }
            else { part[at] = a; part[at + d.N] = b; part[at + 2 * d.N] = w; }
        }
    };
    const float s0 = shift[c], s1 = shift[d.C + c], s2 = shift[2 * d.C + c];
    const bool integer = split_shift(s0).r == 0 || split_shift(s1).r == 0 || split_shift(s2).r == 0;
    if (__builtin_amdgcn_readfirstlane((int)integer)) {
        // exactly-integer component: the reference's per-element formulation (lowered-index quirk :290-298)
        float aT = 0.f, aH = 0.f, aW = 0.f;
        for (int t = 0; t < d.T; ++t) {
            if (WRITE_GX) backward_input_plane<float, false>(shift, gy, gx, gd, n, t, c, lane, kWave);
            shift_grad_plane<float>(x, shift, gy, gd, n, t, c, lane, kWave, aT, aH, aW);
        }
        publish(wave_sum(aT), wave_sum(aH), wave_sum(aW));
        return;
    }
    char* ring = lds_raw + wave * (6 * kStride);                      // slots 0..2: gy, 3..5: x
    if (lane < 6) *reinterpret_cast<float4*>(ring + lane * kStride + kZ) = make_float4(0.f, 0.f, 0.f, 0.f);
    const Frac<float> fT = split_shift(-s0), fH = split_shift(-s1), fW = split_shift(-s2);   // fl', r' of the negated shift
    const float rT = uni(fT.r), rH = uni(fH.r), rW = uni(fW.r), uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const int f0 = __builtin_amdgcn_readfirstlane(fT.fl);
    const int flH = __builtin_amdgcn_readfirstlane(fH.fl), flW = __builtin_amdgcn_readfirstlane(fW.fl);
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring)), xaddr = gaddr + 3 * kStride;
    const long long tstride = (long long)d.C * kHW;
    const size_t col0 = ((size_t)n * d.T * d.C + c) * kHW;
    float* optr = WRITE_GX ? gx + col0 + lane : nullptr;

    unsigned rel[kRC][4], xoff[kRC];
#pragma unroll
    for (int rc = 0; rc < kRC; ++rc) {
        taps(rel[rc], gaddr, flH, flW, lane, rc);
        const int p = lane + kWave * rc;
        xoff[rc] = xaddr + (p < kHW ? (unsigned)(p * 4) : (unsigned)kZ);
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
    auto step = [&](auto SC, auto LATE, bool store) {
        constexpr int S = decltype(SC)::value;
        constexpr int W0 = (decltype(LATE)::value && WRITE_GX) ? 8 : 0;
        // outstanding behind "gy plane k and x plane k have landed": the fetches of step k-1 that were VMEM ops
        // (planes tg - 1, tx - 1 now) and the stores of steps k-2, k-1 (LATE: k >= 3)
        const int more = ((unsigned)(tg - 1) < (unsigned)d.T ? 1 : 0) + ((unsigned)(tx - 1) < (unsigned)d.T ? 1 : 0);
        if (more == 2) wait_vmcnt(W0 + 2); else if (more == 1) wait_vmcnt(W0 + 1); else wait_vmcnt(W0);
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
        f32x2 Qnew[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 la = q[h][0] * uW2 + q[h][1] * rW2, lb = q[h][2] * uW2 + q[h][3] * rW2;
            Qnew[h] = uH2 * la + rH2 * lb;                           // the reference's tree, contraction off
            const f32x2 c0 = fma2(uH2, q[h][0], rH2 * q[h][2]), c1 = fma2(uH2, q[h][1], rH2 * q[h][3]);
            const f32x2 dx = xb[h] - xa[h], mx = fma2(uT2, xb[h], rT2 * xa[h]);
            aT = fma2(Qnew[h], dx, aT);
            aH = fma2(la - lb, mx, aH);
            aW = fma2(c0 - c1, mx, aW);
            xa[h] = xb[h];
        }
        if (WRITE_GX) {
            if (store) {
                const f32x2 o0 = uT2 * Qprev[0] + rT2 * Qnew[0], o1 = uT2 * Qprev[1] + rT2 * Qnew[1];
                __builtin_nontemporal_store(o0.x, optr);
                __builtin_nontemporal_store(o0.y, optr + 64);
                __builtin_nontemporal_store(o1.x, optr + 128);
                if (lane < kHW - 192) __builtin_nontemporal_store(o1.y, optr + 192);
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
    publish(wave_sum(aT.x + aT.y), wave_sum(aH.x + aH.y), wave_sum(aW.x + aW.y));
}
}  // namespace t14

// ---------------------------------------------------------------------------------------------
// Host side.
template <typename G> inline bool tile_dims(TDims& t, const Dims3& d) {
    if (d.H != G::H || d.W != G::W) return false;
    if ((G::HW % 4) != 0 && (d.C % 4) != 0) return false;             // every tile must start on a 16-byte boundary
    t.N = d.N; t.T = d.T; t.C = d.C;
    t.NG = (d.C + G::GC - 1) / G::GC;
    return true;
}
inline bool s1p0(const Dims3& d) {
    return d.sT == 1 && d.sH == 1 && d.sW == 1 && d.pT == 0 && d.pH == 0 && d.pW == 0 && streaming_kernels_on();
}

template <int H, int W, int GCO, int RING, bool NEGATE>
inline bool launch_interp_hw(const float* src, const float* shift, float* dst, const Dims3& d, hipStream_t stream) {
    using G = Geo<H, W, GCO>;
    TDims t;
    if (!tile_dims<G>(t, d)) return false;
    const unsigned grid = (unsigned)(((long long)t.N * t.NG + 3) / 4);
    hipLaunchKernelGGL((k3d_tile_interp<H, W, GCO, RING, NEGATE>), dim3(grid), dim3(kBlock), 4 * wave_lds_bytes<G>(RING),
                       stream, src, shift, dst, t);
    return true;
}
// forward / d(x)-only; false = not handled here.  14x14: 1 channel per wave (4 rounds), ring of 3 (ring of 4: slower).
// (7x7 planes were measured on this scheme too -- 16 or 8 channels per wave -- and came out level with the column
// kernels, 74.6 vs 74.8 us fwd+bwd at [32,8,576,7,7]; they stay there.)
template <bool NEGATE>
inline bool launch_interp(const float* src, const float* shift, float* dst, const Dims3& d, hipStream_t stream) {
    if (!s1p0(d) || !aligned16(src) || !aligned16(dst)) return false;
    if (d.H != 14 || d.W != 14) return false;
    TDims t{d.N, d.T, d.C, d.C};
    const unsigned grid = (unsigned)(((long long)d.N * d.C + 3) / 4);
    hipLaunchKernelGGL((t14::k3d_tile14_interp<NEGATE>), dim3(grid), dim3(kBlock), 4 * 3 * t14::kStride, stream, src, shift,
                       dst, t);
    return true;
}

template <int H, int W, int GCO, int RING>
inline int launch_bwd_hw(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws,
                         const Dims3& d, int normalize, float t_factor, hipStream_t stream) {
    using G = Geo<H, W, GCO>;
    TDims t;
    if (!tile_dims<G>(t, d)) return 0;
    const unsigned producers = (unsigned)(((long long)t.N * t.NG + 3) / 4);
    const size_t lds = 4 * wave_lds_bytes<G>(RING + RING - 2);
    dma3d::Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin.f.tag = next_launch_tag();
    fin.f.producers = (int)producers;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
#define RK_TILE_BWD(GX, FU) hipLaunchKernelGGL((k3d_tile_backward<H, W, GCO, RING, GX, FU>), dim3(producers + (FU ? d.C : 0)), \
                                               dim3(kBlock), lds, stream, x, shift, gy, gx, ws, t, d, fin)
    if (gshift) { if (gx) RK_TILE_BWD(true, true); else RK_TILE_BWD(false, true); }
    else { if (gx) RK_TILE_BWD(true, false); else RK_TILE_BWD(false, false); }
#undef RK_TILE_BWD
    return d.N;
}
// d(shift) (+ d(x) when gx != nullptr); gshift != nullptr: row-sum + K5 fused into the launch (ws = granules), else
// plain partials ws[C][3][P].  Returns P (0 = not handled here)
inline int launch_bwd(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws,
                      const Dims3& d, int normalize, float t_factor, hipStream_t stream) {
    if (!s1p0(d) || !aligned16(x) || !aligned16(gy) || (gx && !aligned16(gx))) return 0;
    if (d.H != 14 || d.W != 14) return 0;
    TDims t{d.N, d.T, d.C, d.C};
    const unsigned producers = (unsigned)(((long long)d.N * d.C + 3) / 4);
    const size_t lds = 4 * 6 * t14::kStride;
    dma3d::Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin.f.tag = next_launch_tag();
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
