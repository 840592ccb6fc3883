// rk2d_tile.hpp -- RubiksShift2D on 14x14 planes (stride 1 / pad 0; fp32, f16, bf16): the 35 layer-3 blocks of the
// -aq networks ([256,288,14,14] per GPU for Large-AQ), which the column kernels of rk2d_column.hpp
// run at 0.8-1.8 TB/s.  Same idea as rk3d_tile.hpp: the planes of consecutive channels of one frame are contiguous,
// so the unit is a TILE = one plane of one channel in fp32 (784 B = 49 aligned 16-byte pieces) and of 2 consecutive
// channels for the 16-bit types (2 x 392 B = 49 pieces), LDS-DMA'd RAW (global_load_lds_dwordx4 nt, counted vmcnt) by the WAVE that owns it; a
// workgroup is 4 independent waves and there is no workgroup barrier.  A wave owns (channel group, group of FG
// frames) and walks its frames through a ring of R slots; the shift is per channel, so all tap addresses are
// computed ONCE per wave, relative to a slot, a tap outside the plane pointing at the slot's zero word.
//
// A lane owns PAIRS of horizontally adjacent elements (W is even, so a pair never straddles a row): six taps
// (rows A / B x three columns) feed two outputs, 16-bit taps are widened after the LDS read (ds_read_u16), and the
// pair leaves as one 8-byte (fp32) or 4-byte (16-bit) nt store.  Arithmetic: interp2d (rubiks2d_kernels.cu:60-66)
// for the forward (K6 :94-146) and d(x) (K8 :276-378) with contraction off, rounded once for the 16-bit types =>
// bit-identical to the other 2-D kernels and to the oracle; d(shift) (K7 :164-273) in the adjoint form of
// rk2d_dma.hpp, integer shifts by extra walks with lowered floors (IntegerPlan), row-sum + K9 inside the launch.
#pragma once
#include "rk2d_raw16.hpp"

namespace rk {
namespace tile2d {

using namespace dma;
using dma2d::Fin2;
using dma2d::IntegerPlan;
using dma2d::BnFuse2;
using dma2d::finalizer_wave2_bn;
using g2d::Dims2;

template <typename T, int H_, int W_> struct Geo {
    static constexpr int H = H_, W = W_, HW = H_ * W_;
    static constexpr int ES = (int)sizeof(T);
    static constexpr int GC = ES == 4 ? 1 : 2;                    // channels per tile (16-bit: 2 x 392 B is the smallest whole number of 16-byte pieces)
    static constexpr int PAIRS = HW / 2;
    static constexpr int RC = (PAIRS + kWave - 1) / kWave;        // rounds per channel
    static constexpr int ROUNDS = GC * RC;
    static constexpr int TILE_B = GC * HW * ES;
    static constexpr int PIECES = TILE_B / 16;
    static constexpr int PIECE_ROUNDS = (PIECES + kWave - 1) / kWave;
    static constexpr int STRIDE = TILE_B + 16;                    // slot: [tile][16 zero bytes]
    static constexpr unsigned ZOFF = TILE_B;
    static_assert(W_ % 2 == 0 && TILE_B % 16 == 0, "pairs inside a row; a tile is whole 16-byte pieces");
};

struct TDims2 {
    int F, C, NG;                // frames, channels, channel groups (C / GC)
    int FG, ngroups;             // frames per wave, frame groups
};

// LDS element -> fp32
template <typename T> struct LdsElem;
template <> struct LdsElem<float> {
    __device__ static __forceinline__ float get(unsigned a) { return *(__attribute__((address_space(3))) const float*)(size_t)a; }
};
template <> struct LdsElem<__hip_bfloat16> {
    __device__ static __forceinline__ float get(unsigned a) {
        return __uint_as_float((unsigned)(*(__attribute__((address_space(3))) const unsigned short*)(size_t)a) << 16);
    }
};
template <> struct LdsElem<__half> {
    __device__ static __forceinline__ float get(unsigned a) {
        return __half2float(__builtin_bit_cast(__half, *(__attribute__((address_space(3))) const unsigned short*)(size_t)a));
    }
};
// own pair (aligned) -> two fp32
template <typename T> __device__ __forceinline__ void lds_pair(unsigned a, float& v0, float& v1) {
    if constexpr (sizeof(T) == 4) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 v = *(__attribute__((address_space(3))) const f32x2*)(size_t)a;
        v0 = v.x; v1 = v.y;
    } else {
        const unsigned w = *(__attribute__((address_space(3))) const unsigned*)(size_t)a;
        v0 = raw16::Wide<T>::get(w, 0); v1 = raw16::Wide<T>::get(w, 1);
    }
}
template <typename T> __device__ __forceinline__ void store_pair(char* p, float v0, float v1) {
    if constexpr (sizeof(T) == 4) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 t = {v0, v1};
        __builtin_nontemporal_store(t, reinterpret_cast<f32x2*>(p));
    } else {
        using C4 = stage2d::Cell4<T>;
        __builtin_nontemporal_store(C4::bits(v0) | (C4::bits(v1) << 16), reinterpret_cast<unsigned*>(p));
    }
}

template <typename G> __device__ __forceinline__ int dma_tile(const void* tile, unsigned dst, int lane) {
#pragma unroll
    for (int i = 0; i < G::PIECE_ROUNDS; ++i)
        if (lane + kWave * i < G::PIECES) dma16s<true>(tile, (lane + kWave * i) * 16, dst + 1024u * i);
    return G::PIECE_ROUNDS;
}

// Training fusion (round 5, fused_bn.bn_relu_shift2d): the tile holds z = conv2's output and the shift applies to
// relu(bn2(z)) = max(a z + b, 0) rounded to the storage type -- the tensor the unfused path would have stored.  The wave
// transforms the tile it DMA'd in place once it has landed (its own counted wait; LDS operations of one wave execute in
// order, so the tap reads that follow see the transformed values); the slot's zero cell is not touched.
template <typename G, typename T>
__device__ __forceinline__ void bn_tile(char* slot, int lane, const float (&a)[G::GC], const float (&b)[G::GC]) {
#pragma unroll
    for (int i = 0; i < G::PIECE_ROUNDS; ++i) {
        const int p = lane + kWave * i;
        if (p >= G::PIECES) continue;
        if constexpr (sizeof(T) == 4) {
            float4* q = reinterpret_cast<float4*>(slot + 16 * p);
            float4 v = *q;
            v.x = fmaxf(fmaf(a[0], v.x, b[0]), 0.f); v.y = fmaxf(fmaf(a[0], v.y, b[0]), 0.f);
            v.z = fmaxf(fmaf(a[0], v.z, b[0]), 0.f); v.w = fmaxf(fmaf(a[0], v.w, b[0]), 0.f);
            *q = v;
        } else {
            using C4 = stage2d::Cell4<T>;
            uint4* q = reinterpret_cast<uint4*>(slot + 16 * p);
            const uint4 w = *q;
            const unsigned in[4] = {w.x, w.y, w.z, w.w};
            unsigned out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool g0 = 8 * p + 2 * j >= G::HW, g1 = 8 * p + 2 * j + 1 >= G::HW;      // second channel of the tile
                const float a0 = (G::GC == 2 && g0) ? a[G::GC - 1] : a[0], b0 = (G::GC == 2 && g0) ? b[G::GC - 1] : b[0];
                const float a1 = (G::GC == 2 && g1) ? a[G::GC - 1] : a[0], b1 = (G::GC == 2 && g1) ? b[G::GC - 1] : b[0];
                const float v0 = fmaxf(fmaf(a0, raw16::Wide<T>::get(in[j], 0), b0), 0.f);
                const float v1 = fmaxf(fmaf(a1, raw16::Wide<T>::get(in[j], 1), b1), 0.f);
                out[j] = C4::bits(v0) | (C4::bits(v1) << 16);
            }
            *q = make_uint4(out[0], out[1], out[2], out[3]);
        }
    }
}
// a value as the storage type holds it
template <typename T> __device__ __forceinline__ float round_to(float v) {
    if constexpr (sizeof(T) == 4) return v;
    else return raw16::Wide<T>::get(stage2d::Cell4<T>::bits(v), 0);
}

// one channel of the tile in one walk: fractions, what it does in this walk
struct ChanW { Frac<float> fH, fW; bool on, store; };

// relative byte offsets (inside a slot) of the six taps of my pair in round (g, rc), my pair's own offset inside a tile
// (or ZOFF / -1 when this lane has no pair in the round or the channel sits this walk out)
template <typename G>
__device__ __forceinline__ void make_round(unsigned (&rel)[6], unsigned& own, int& ooff, const ChanW& ch, int g, int lane,
                                           int rc) {
    const int p = lane + kWave * rc;
    const bool live = ch.on && p < G::PAIRS;
    const int e = 2 * (p < G::PAIRS ? p : 0);
    const int h = e / G::W, w = e - h * G::W;
    const int h0 = h + ch.fH.fl, w0 = w + ch.fW.fl;
    const bool mh0 = (unsigned)h0 < (unsigned)G::H, mh1 = (unsigned)(h0 + 1) < (unsigned)G::H;
    const unsigned a = (unsigned)((g * G::HW + h0 * G::W + w0) * G::ES);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool mw = (unsigned)(w0 + k) < (unsigned)G::W;
        rel[k] = live && mh0 && mw ? a + (unsigned)(k * G::ES) : G::ZOFF;
        rel[3 + k] = live && mh1 && mw ? a + (unsigned)((G::W + k) * G::ES) : G::ZOFF;
    }
    own = live ? (unsigned)((g * G::HW + e) * G::ES) : G::ZOFF;
    ooff = live && ch.store ? (g * G::HW + e) * G::ES : -1;
}

struct Item { int c0, f0, nf; bool live; };
template <typename G> __device__ __forceinline__ Item my_item(const TDims2& d, int wave) {
    Item it;
    const long long id = (long long)blockIdx.x * (kBlock / kWave) + wave;
    it.live = id < (long long)d.NG * d.ngroups;
    const long long q = it.live ? id : 0;
    it.c0 = (int)(q % d.NG) * G::GC;
    it.f0 = (int)(q / d.NG) * d.FG;
    it.nf = d.F - it.f0 < d.FG ? d.F - it.f0 : d.FG;
    return it;
}

// ---------------------------------------------------------------------------------------------
// Forward (NEGATE = false: src = x) and d(x) alone (NEGATE = true: src = gy, negated shift).
template <typename T, typename S, int H, int W, int R, bool NEGATE, bool BN = false>
__global__ __launch_bounds__(kBlock) void k2d_tile_interp(const T* __restrict__ src, const S* __restrict__ shift,
                                                          T* __restrict__ dst, TDims2 d, const float* __restrict__ ab = nullptr) {
    using G = Geo<T, H, W>;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (kWave - 1);
    const Item it = my_item<G>(d, wave);
    if (!it.live) return;                                            // whole wave; no barriers in this kernel
    char* ring = lds_raw + wave * (R * G::STRIDE);
    if (lane < R) *reinterpret_cast<float4*>(ring + lane * G::STRIDE + G::ZOFF) = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(ring));

    unsigned rel[G::ROUNDS][6], own;
    int ooff[G::ROUNDS];
    float uH[G::GC], rH[G::GC], uW[G::GC], rW[G::GC];
#pragma unroll
    for (int g = 0; g < G::GC; ++g) {
        float sH = ld(shift + it.c0 + g), sW = ld(shift + d.C + it.c0 + g);
        if (NEGATE) { sH = -sH; sW = -sW; }
        ChanW ch;
        ch.fH = split_shift(sH); ch.fW = split_shift(sW);
        ch.on = true; ch.store = true;
        rH[g] = ch.fH.r; rW[g] = ch.fW.r; uH[g] = 1 - rH[g]; uW[g] = 1 - rW[g];
#pragma unroll
        for (int rc = 0; rc < G::RC; ++rc) make_round<G>(rel[g * G::RC + rc], own, ooff[g * G::RC + rc], ch, g, lane, rc);
    }
    (void)own;
    float bn_a[G::GC], bn_b[G::GC];
#pragma unroll
    for (int g = 0; g < G::GC; ++g) { bn_a[g] = BN ? ab[it.c0 + g] : 1.f; bn_b[g] = BN ? ab[d.C + it.c0 + g] : 0.f; }
    const size_t fstride = (size_t)d.C * G::HW;                       // elements between frames
    const T* col = src + ((size_t)it.f0 * d.C + it.c0) * G::HW;
    char* ocol = reinterpret_cast<char*>(dst + ((size_t)it.f0 * d.C + it.c0) * G::HW);

    if (it.nf == R) {
        // A full group (the usual case): frame k lives in slot k, so the ring needs no bookkeeping -- all R frames are
        // requested up front, the slot is an immediate offset of the LDS reads and the counted waits are literals (VMEM
        // order: R fetches, then ROUNDS stores per step).  The kernels are bound by instructions issued per step
        // (rk3d_tile.hpp), and this path drops the per-tap address add and the runtime s_waitcnt switch.
#pragma unroll
        for (int r = 0; r < G::ROUNDS; ++r)
#pragma unroll
            for (int j = 0; j < 6; ++j) rel[r][j] += ring_addr;
#pragma unroll
        for (int k = 0; k < R; ++k) dma_tile<G>(col + (size_t)k * fstride, ring_addr + k * G::STRIDE, lane);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            wait_vmcnt((R - 1 - k) * G::PIECE_ROUNDS + k * G::ROUNDS);
            if (BN) bn_tile<G, T>(ring + k * G::STRIDE, lane, bn_a, bn_b);
            char* out = ocol + (size_t)k * fstride * G::ES;
            float o0[G::ROUNDS], o1[G::ROUNDS];
#pragma unroll
            for (int g = 0; g < G::GC; ++g)
#pragma unroll
                for (int rc = 0; rc < G::RC; ++rc) {
                    const int r = g * G::RC + rc;
                    const unsigned sb = (unsigned)(k * G::STRIDE);
                    const float a0 = LdsElem<T>::get(rel[r][0] + sb), a1 = LdsElem<T>::get(rel[r][1] + sb),
                                a2 = LdsElem<T>::get(rel[r][2] + sb);
                    const float b0 = LdsElem<T>::get(rel[r][3] + sb), b1 = LdsElem<T>::get(rel[r][4] + sb),
                                b2 = LdsElem<T>::get(rel[r][5] + sb);
                    o0[r] = a0 * uH[g] * uW[g] + a1 * uH[g] * rW[g] + b0 * rH[g] * uW[g] + b1 * rH[g] * rW[g];   // interp2d
                    o1[r] = a1 * uH[g] * uW[g] + a2 * uH[g] * rW[g] + b1 * rH[g] * uW[g] + b2 * rH[g] * rW[g];
                }
#pragma unroll
            for (int r = 0; r < G::ROUNDS; ++r)
                if (ooff[r] >= 0) store_pair<T>(out + ooff[r], o0[r], o1[r]);
        }
        return;
    }
    int issued = 0;
    auto fetch = [&](int k) {                                        // frame k -> slot k % R
        if (k < it.nf) issued += dma_tile<G>(col + (size_t)k * fstride, ring_addr + (k % R) * G::STRIDE, lane);
    };
    // fifo[i] = `issued` right after the fetch of frame k + i: "frame k has landed" <=> outstanding <= issued - fifo[0]
    int fifo[R - 1];
#pragma unroll
    for (int i = 0; i < R - 1; ++i) { fetch(i); fifo[i] = issued; }
    for (int k = 0; k < it.nf; ++k) {
        wait_vmcnt(issued - fifo[0]);
        fetch(k + R - 1);                                            // into the slot of frame k - 1
#pragma unroll
        for (int i = 0; i + 1 < R - 1; ++i) fifo[i] = fifo[i + 1];
        fifo[R - 2] = issued;
        const unsigned sb = ring_addr + (k % R) * G::STRIDE;
        if (BN) bn_tile<G, T>(ring + (k % R) * G::STRIDE, lane, bn_a, bn_b);
        char* out = ocol + (size_t)k * fstride * G::ES;
        float o0[G::ROUNDS], o1[G::ROUNDS];
#pragma unroll
        for (int g = 0; g < G::GC; ++g)
#pragma unroll
            for (int rc = 0; rc < G::RC; ++rc) {
                const int r = g * G::RC + rc;
                const float a0 = LdsElem<T>::get(rel[r][0] + sb), a1 = LdsElem<T>::get(rel[r][1] + sb),
                            a2 = LdsElem<T>::get(rel[r][2] + sb);
                const float b0 = LdsElem<T>::get(rel[r][3] + sb), b1 = LdsElem<T>::get(rel[r][4] + sb),
                            b2 = LdsElem<T>::get(rel[r][5] + sb);
                o0[r] = a0 * uH[g] * uW[g] + a1 * uH[g] * rW[g] + b0 * rH[g] * uW[g] + b1 * rH[g] * rW[g];   // interp2d
                o1[r] = a1 * uH[g] * uW[g] + a2 * uH[g] * rW[g] + b1 * rH[g] * uW[g] + b2 * rH[g] * rW[g];
            }
#pragma unroll
        for (int r = 0; r < G::ROUNDS; ++r) {
            if (ooff[r] >= 0) store_pair<T>(out + ooff[r], o0[r], o1[r]);
            ++issued;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) + d(shift); partials [C][2][P = ngroups] as granules, row-sum + K9 by the finalizer blocks.
// BN (training fusion): x holds z = bn2's input.  The activation relu(a z + b) (rounded to the storage type: what the forward
// shifted) is recomputed where the d(shift) sums use it -- a lane needs x only at its OWN pair --, d(x) leaves masked by the
// ReLU (= d(bn2's output), what k_bn_bwd_dx_pre expects), and bn2's two reduction sums (sum dz, sum dz zhat) ride along as
// partials 2 and 3: [C][4][P]; the finalizer also writes k12 / d(gamma) / d(beta).
// (the bn2-fused bf16 instance takes 161 VGPRs = 3 waves per SIMD; forced to 4 waves it spills 100 bytes per lane: 34.5 -> 43.9 us
// at [256,288,14,14])
#ifndef RK_T2_BN_WAVES
#define RK_T2_BN_WAVES 2
#endif
template <typename T, typename S, int H, int W, int R, bool BN = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(BN && sizeof(T) == 2 ? RK_T2_BN_WAVES : 2)))
void k2d_tile_backward(const T* __restrict__ gy, const T* __restrict__ x,
                                                            const S* __restrict__ shift, T* __restrict__ gx, TDims2 d,
                                                            Fin2<S> fin, BnFuse2 bn = BnFuse2{}) {
    using G = Geo<T, H, W>;
    constexpr int ND = BN ? 4 : 2;
    if ((int)blockIdx.x >= fin.f.producers) {                         // row-sum + K9 inside the launch (rk_dma.hpp)
        if (threadIdx.x < kWave) {
            if (BN) finalizer_wave2_bn(fin, (int)blockIdx.x - fin.f.producers, d.C, d.ngroups, bn);
            else dma2d::finalizer_wave2(fin, (int)blockIdx.x - fin.f.producers, d.C, d.ngroups);
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & (kWave - 1);
    const Item it = my_item<G>(d, wave);
    if (!it.live) return;
    char* gring = lds_raw + wave * (2 * R * G::STRIDE);
    char* xring = gring + R * G::STRIDE;
    if (lane < 2 * R) *reinterpret_cast<float4*>(gring + lane * G::STRIDE + G::ZOFF) = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned gaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(gring));
    const unsigned xaddr = __builtin_amdgcn_readfirstlane(lds_byte_addr(xring));
    const size_t fstride = (size_t)d.C * G::HW;
    const size_t base = ((size_t)it.f0 * d.C + it.c0) * G::HW;
    const T* gcol = gy + base;
    const T* xcol = x + base;
    char* ocol = reinterpret_cast<char*>(gx + base);

    IntegerPlan plan[G::GC];
    bool extra = false;                                              // does any channel of the tile need more than walk 0?
#pragma unroll
    for (int g = 0; g < G::GC; ++g) {
        plan[g] = dma2d::plan_walks(ld(shift + it.c0 + g), ld(shift + d.C + it.c0 + g));
        extra = extra || plan[g].separate_gx || plan[g].hint || plan[g].wint;
    }
    float sumH0[G::GC], sumW0[G::GC], sumH1[G::GC], sumW2[G::GC], sumB1[G::GC], sumB2[G::GC];
    float4 bnp[G::GC];
#pragma unroll
    for (int g = 0; g < G::GC; ++g) {
        sumH0[g] = sumW0[g] = sumH1[g] = sumW2[g] = sumB1[g] = sumB2[g] = 0.f;
        bnp[g] = BN ? bn.abmi[it.c0 + g] : make_float4(1.f, 0.f, 0.f, 1.f);
    }

    // walk -1: d(x) alone with the true remainder (channels whose |r| < 1e-7 but != 0: K8 has no tolerance);
    // walk 0: sums (+ d(x) for every ordinary channel); walk 1 / 2: sums with the H / W floor lowered by one.
#pragma nounroll
    for (int walk = extra ? -1 : 0; walk < (extra ? 3 : 1); ++walk) {
        unsigned rel[G::ROUNDS][6], own[G::ROUNDS];
        int ooff[G::ROUNDS];
        float uH[G::GC], rH[G::GC], uW[G::GC], rW[G::GC];
        bool any = false, st_on[G::GC];
#pragma unroll
        for (int g = 0; g < G::GC; ++g) {
            ChanW ch;
            if (walk < 0) { ch.fH = plan[g].gH; ch.fW = plan[g].gW; ch.on = plan[g].separate_gx; ch.store = true; }
            else {
                ch.fH = plan[g].sH; ch.fW = plan[g].sW;
                if (walk == 1) ch.fH.fl -= 1;
                if (walk == 2) ch.fW.fl -= 1;
                ch.on = plan[g].walk_on(walk);
                ch.store = walk == 0 && !plan[g].separate_gx;
            }
            any = any || ch.on;
            st_on[g] = ch.on && ch.store;                            // wave-uniform
            rH[g] = ch.fH.r; rW[g] = ch.fW.r; uH[g] = 1 - rH[g]; uW[g] = 1 - rW[g];
#pragma unroll
            for (int rc = 0; rc < G::RC; ++rc)
                make_round<G>(rel[g * G::RC + rc], own[g * G::RC + rc], ooff[g * G::RC + rc], ch, g, lane, rc);
        }
        if (!any) continue;                                          // wave-uniform
        float aH[G::GC], aW[G::GC], bB1[G::GC], bB2[G::GC];
#pragma unroll
        for (int g = 0; g < G::GC; ++g) aH[g] = aW[g] = bB1[g] = bB2[g] = 0.f;

        int issued = 0;
        auto fetch = [&](int k) {                                    // frame k: gy -> gy slot, x -> x slot k % R
            if (k < it.nf) {
                issued += dma_tile<G>(gcol + (size_t)k * fstride, gaddr + (k % R) * G::STRIDE, lane);
                issued += dma_tile<G>(xcol + (size_t)k * fstride, xaddr + (k % R) * G::STRIDE, lane);
            }
        };
        int fifo[R - 1];
#pragma unroll
        for (int i = 0; i < R - 1; ++i) { fetch(i); fifo[i] = issued; }
        for (int k = 0; k < it.nf; ++k) {
            wait_vmcnt(issued - fifo[0]);
            fetch(k + R - 1);
#pragma unroll
            for (int i = 0; i + 1 < R - 1; ++i) fifo[i] = fifo[i + 1];
            fifo[R - 2] = issued;
            const unsigned gb = gaddr + (k % R) * G::STRIDE, xb = xaddr + (k % R) * G::STRIDE;
            char* out = ocol + (size_t)k * fstride * G::ES;
            float o0[G::ROUNDS], o1[G::ROUNDS];
#pragma unroll
            for (int g = 0; g < G::GC; ++g)
#pragma unroll
                for (int rc = 0; rc < G::RC; ++rc) {
                    const int r = g * G::RC + rc;
                    const float a0 = LdsElem<T>::get(rel[r][0] + gb), a1 = LdsElem<T>::get(rel[r][1] + gb),
                                a2 = LdsElem<T>::get(rel[r][2] + gb);
                    const float b0 = LdsElem<T>::get(rel[r][3] + gb), b1 = LdsElem<T>::get(rel[r][4] + gb),
                                b2 = LdsElem<T>::get(rel[r][5] + gb);
                    float x0, x1;
                    lds_pair<T>(own[r] + xb, x0, x1);
                    o0[r] = a0 * uH[g] * uW[g] + a1 * uH[g] * rW[g] + b0 * rH[g] * uW[g] + b1 * rH[g] * rW[g];   // K8
                    o1[r] = a1 * uH[g] * uW[g] + a2 * uH[g] * rW[g] + b1 * rH[g] * uW[g] + b2 * rH[g] * rW[g];
                    if (BN) {
                        const bool lv = own[r] != G::ZOFF;            // (a lane without a pair reads the zero cell: relu(b) is not 0)
                        const float z0 = x0, z1 = x1;
                        x0 = lv ? round_to<T>(fmaxf(fmaf(bnp[g].x, z0, bnp[g].y), 0.f)) : 0.f;
                        x1 = lv ? round_to<T>(fmaxf(fmaf(bnp[g].x, z1, bnp[g].y), 0.f)) : 0.f;
                        o0[r] = x0 > 0.f ? round_to<T>(o0[r]) : 0.f;  // d(bn2's output): the value as it is stored, ReLU-masked
                        o1[r] = x1 > 0.f ? round_to<T>(o1[r]) : 0.f;
                        if (st_on[g]) {                              // (wave-uniform) the walk that stores this channel's d(x)
                            bB1[g] += o0[r] + o1[r];
                            bB2[g] = fmaf(o0[r], (z0 - bnp[g].z) * bnp[g].w, bB2[g]);
                            bB2[g] = fmaf(o1[r], (z1 - bnp[g].z) * bnp[g].w, bB2[g]);
                        }
                    }
                    const float c0 = fmaf(uH[g], a0, rH[g] * b0), c1 = fmaf(uH[g], a1, rH[g] * b1),
                                c2 = fmaf(uH[g], a2, rH[g] * b2);
                    const float la0 = fmaf(a0, uW[g], a1 * rW[g]), lb0 = fmaf(b0, uW[g], b1 * rW[g]);
                    const float la1 = fmaf(a1, uW[g], a2 * rW[g]), lb1 = fmaf(b1, uW[g], b2 * rW[g]);
                    aH[g] = fmaf(la0 - lb0, x0, aH[g]);
                    aH[g] = fmaf(la1 - lb1, x1, aH[g]);
                    aW[g] = fmaf(c0 - c1, x0, aW[g]);
                    aW[g] = fmaf(c1 - c2, x1, aW[g]);
                }
#pragma unroll
            for (int g = 0; g < G::GC; ++g) {
                if (!st_on[g]) continue;                             // wave-uniform: no store instruction, none counted
#pragma unroll
                for (int rc = 0; rc < G::RC; ++rc) {
                    const int r = g * G::RC + rc;
                    if (ooff[r] >= 0) store_pair<T>(out + ooff[r], o0[r], o1[r]);
                    ++issued;
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G::GC; ++g) {
            if (walk == 0) { sumH0[g] = aH[g]; sumW0[g] = aW[g]; }
            else if (walk == 1) sumH1[g] = aH[g];
            else if (walk == 2) sumW2[g] = aW[g];
            if (BN && st_on[g]) { sumB1[g] = bB1[g]; sumB2[g] = bB2[g]; }      // (exactly one walk stores a channel)
        }
    }

#pragma unroll
    for (int g = 0; g < G::GC; ++g) {
        float accH = plan[g].hint ? 0.5f * (sumH0[g] + sumH1[g]) : sumH0[g];
        float accW = plan[g].wint ? 0.5f * (sumW0[g] + sumW2[g]) : sumW0[g];
        accH = wave_sum(accH);
        accW = wave_sum(accW);
        float accB1 = 0.f, accB2 = 0.f;
        if (BN) { accB1 = wave_sum(sumB1[g]); accB2 = wave_sum(sumB2[g]); }
        if (lane == 0) {
            const int P = d.ngroups;
            const size_t at = (size_t)(it.c0 + g) * ND * P + (size_t)(it.f0 / d.FG);
            fin_publish(fin.f, at, accH);
            fin_publish(fin.f, at + P, accW);
            if (BN) { fin_publish(fin.f, at + 2 * (size_t)P, accB1); fin_publish(fin.f, at + 3 * (size_t)P, accB2); }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Host side.  false / 0 = shape not handled here.
// frames per wave, [256,288,14,14] fwd / bwd us (steady state): bf16 1: 22 / 44, 2: 19 / 31, 4: 17 / 26, 8: 17 / 26,
// 16: 19 / 28, 32: 26 / 40, 64: 45 / 62; fp32 1: 22 / 50, 2: 21 / 40, 4: 22.5 / 35, 8: 24 / 36, 16: 27 / 42, 32: 30 / 46
// The backward of the 16-bit types (3 waves per SIMD) in ONE round of resident waves: with 4 frames per wave [256,288,14,14] is
// 9 216 waves = 3 rounds, each wave paying its address plan, DMA ramp and finalizer hand-off for 4 frames; frames per wave
// so that the waves just fit the machine (12 at 256 CUs) measured 25.8 -> 24.3 us plain, 34.4 -> 30.5 us bn2-fused (sweep,
// us plain / fused: 4: 25.8 / 34.4, 6: 25.1 / 33.7, 8: 26.2 / 34.0, 10: 25.0 / 35.1, 12: 24.3 / 30.5, 13: 24.9 / 31.8, 14: 26.0 /
// 33.6, 16: 28.2 / 36.9 -- the optimum is the round boundary; the forward is best at 4: 14.4 / 15.7 us against 15.5 / 17.6 at 12).
constexpr int kTileFrames = 4;
template <typename T, int H, int W> inline bool make_tdims(TDims2& t, const Dims2& d, bool backward = false) {
    using G = Geo<T, H, W>;
    const bool s1p0 = d.sH == 1 && d.sW == 1 && d.pH == 0 && d.pW == 0;
    if (!s1p0 || d.H != H || d.W != W || d.C % G::GC != 0 || !streaming_kernels_on()) return false;
    t.F = d.N; t.C = d.C; t.NG = d.C / G::GC;
    int fg = kTileFrames;
    if (backward && sizeof(T) == 2) {
        const long long slots = (long long)device_cus() * 12;       // 3 waves per SIMD
        const long long want = ((long long)d.N * t.NG + slots - 1) / slots;
        fg = (int)(want < kTileFrames ? kTileFrames : (want > 16 ? 16 : want));
    }
    t.FG = fg < d.N ? fg : d.N;
    t.ngroups = (d.N + t.FG - 1) / t.FG;
    return true;
}
template <typename T> inline int backward2_partials(const Dims2& d) {
    TDims2 t;
    return make_tdims<T, 14, 14>(t, d, true) ? t.ngroups : 0;
}

template <typename T, bool NEGATE, typename S>
inline bool launch_interp2(const T* src, const S* shift, T* dst, const Dims2& d, hipStream_t stream) {
    constexpr int R = 4;
    using G = Geo<T, 14, 14>;
    TDims2 t;
    if (!make_tdims<T, 14, 14>(t, d) || !aligned16(src) || !aligned16(dst)) return false;
    const long long waves = (long long)t.NG * t.ngroups;
    const size_t lds = (size_t)4 * R * G::STRIDE;
    hipLaunchKernelGGL((k2d_tile_interp<T, S, 14, 14, R, NEGATE>), dim3((unsigned)((waves + 3) / 4)), dim3(kBlock), lds, stream,
                       src, shift, dst, t);
    return true;
}

template <typename T, typename S>
inline bool launch_backward2(const T* gy, const T* x, const S* shift, T* gx, S* gshift, void* ws, int normalize,
                             const Dims2& d, hipStream_t stream) {
    constexpr int R = 3;       // (the forward's static full-group schedule was tried here too: 245 VGPRs, 26 -> 31 us)
    using G = Geo<T, 14, 14>;
    TDims2 t;
    if (!make_tdims<T, 14, 14>(t, d, true) || !aligned16(gy) || !aligned16(x) || !aligned16(gx)) return false;
    const long long waves = (long long)t.NG * t.ngroups;
    const size_t lds = (size_t)4 * 2 * R * G::STRIDE;
    Fin2<S> fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = (int)((waves + 3) / 4);
    fin.gshift = gshift;
    fin.normalize = normalize;
    hipLaunchKernelGGL((k2d_tile_backward<T, S, 14, 14, R>), dim3((unsigned)(fin.f.producers + t.C)), dim3(kBlock), lds, stream, gy,
                       x, shift, gx, t, fin);
    return true;
}

// training fusion: forward of relu(bn2(z)) (ab [2][C]) and its backward; false = shape not handled here
template <typename T, typename S>
inline bool launch_forward2_bn(const T* z, const float* ab, const S* shift, T* y, const Dims2& d, hipStream_t stream) {
    constexpr int R = 4;
    using G = Geo<T, 14, 14>;
    TDims2 t;
    if (!make_tdims<T, 14, 14>(t, d) || !aligned16(z) || !aligned16(y)) return false;
    const long long waves = (long long)t.NG * t.ngroups;
    const size_t lds = (size_t)4 * R * G::STRIDE;
    hipLaunchKernelGGL((k2d_tile_interp<T, S, 14, 14, R, false, true>), dim3((unsigned)((waves + 3) / 4)), dim3(kBlock), lds, stream,
                       z, shift, y, t, ab);
    return true;
}
template <typename T, typename S>
inline bool launch_backward2_bn(const T* gy, const T* z, const S* shift, T* dz, S* gshift, void* ws, int normalize,
                                const BnFuse2& bn, const Dims2& d, hipStream_t stream) {
    constexpr int R = 3;
    using G = Geo<T, 14, 14>;
    TDims2 t;
    if (!make_tdims<T, 14, 14>(t, d, true) || !aligned16(gy) || !aligned16(z) || !aligned16(dz) || !aligned16(bn.abmi)) return false;
    const long long waves = (long long)t.NG * t.ngroups;
    const size_t lds = (size_t)4 * 2 * R * G::STRIDE;
    Fin2<S> fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = (int)((waves + 3) / 4);
    fin.gshift = gshift;
    fin.normalize = normalize;
    hipLaunchKernelGGL((k2d_tile_backward<T, S, 14, 14, R, true>), dim3((unsigned)(fin.f.producers + t.C)), dim3(kBlock), lds, stream,
                       gy, z, shift, dz, t, fin, bn);
    return true;
}

}  // namespace tile2d
}  // namespace rk
