// rk_tshift.hip -- per-channel 3-tap temporal filter, the device half of AttentionShift
// (rubiksnet/attention_shift.py:32-39) for gfx950.
//
// The reference realises y[n,t] = s0*x[n,t-1] + s1*x[n,t] + s2*x[n,t+1] as
// view -> transpose -> conv1d(groups = C*H*W, inflated [C*H*W,1,3] weight) -> transpose ->
// contiguous: >= 3 full passes over the activation plus a pathological grouped conv.
// Here each thread owns VEC contiguous elements of one (n, c) column and walks t with a
// 3-plane register window, so x is read once and y written once (8 B/elem fp32), 16 B per
// lane per access when H*W allows.  Backward does the same walk over (gy, x) producing gx
// and the [C,3] tap gradients through wave-shuffle -> LDS -> per-(n,c) partial -> fixed-order
// fp64 finalize (no atomics).
#include <type_traits>
#include "rk_common.hpp"
#include "rk_dma.hpp"

using namespace rk;

namespace {

struct DimsT {
    int NB, S, C, HW;   // n_batch, n_segment, channels, H*W
    int E, logE;        // threads per (n, c) column
};

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };

template <typename T, int VEC>
__device__ __forceinline__ void load_pack(const T* p, typename Compute<T>::type (&o)[VEC]) {
    const Pack<T, VEC> q = *reinterpret_cast<const Pack<T, VEC>*>(p);
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = ld(&q.v[k]);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_pack(T* p, const typename Compute<T>::type (&o)[VEC]) {
    Pack<T, VEC> q;
#pragma unroll
    for (int k = 0; k < VEC; ++k) st(&q.v[k], o[k]);
    *reinterpret_cast<Pack<T, VEC>*>(p) = q;
}

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_raw(const T* p) { return *reinterpret_cast<const Pack<T, VEC>*>(p); }
template <typename T, int VEC>
__device__ __forceinline__ void unpack(const Pack<T, VEC>& q, typename Compute<T>::type (&o)[VEC]) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = ld(&q.v[k]);
}
// Streaming access for the n_segment = 8 path (NT bit 0: loads, bit 1: stores non-temporal: every element is touched
// once by one thread) and a fence for the register allocator: the packs are converted to fp32 where they are USED.  Left to
// itself hipcc widens every pack as soon as it lands -- 8 time steps x VEC fp32 values per tensor live at once: the bf16
// backward took 212 VGPRs (2 waves per SIMD), its bn1-fused form 256 (ONE wave per SIMD, half the plain kernel's rate).
// Measured (bf16, [256,C,H,H], us, loads+stores / stores only / loads only / neither): forward 112x112 173.7 / 174.6 / 180.4 /
// 176.6, 28x28 21.9 / 19.0 / 21.4 / 22.7, 14x14 12.9 / 11.2 / 12.6 / 11.9; fused backward 112x112 272 / 278 / 460 / 534, 56x56 73 /
// 79 / 146 / 165, 28x28 43 / 48 / 93 / 103, 14x14 25.1 / 24.4 / 40.7 / 42.5 -- the forward streams its stores only, the backward both.
#ifndef RK_TS_NT_FWD
#define RK_TS_NT_FWD 2
#endif
#ifndef RK_TS_NT_BWD
#define RK_TS_NT_BWD 3
#endif
template <int BYTES> struct RawVec { typedef unsigned type __attribute__((ext_vector_type(BYTES / 4))); };
template <> struct RawVec<4> { typedef unsigned type; };
template <typename T, int VEC, int NT>
__device__ __forceinline__ Pack<T, VEC> load_stream(const T* p) {
    constexpr int B = (int)sizeof(T) * VEC;
    if constexpr (B % 4 == 0 && B <= 16 && (NT & 1)) {
        using V = typename RawVec<B>::type;
        const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
        return __builtin_bit_cast(Pack<T, VEC>, v);
    } else return *reinterpret_cast<const Pack<T, VEC>*>(p);
}
template <typename T, int VEC, int NT>
__device__ __forceinline__ void store_stream(T* p, const typename Compute<T>::type (&o)[VEC]) {
    constexpr int B = (int)sizeof(T) * VEC;
    Pack<T, VEC> q;
#pragma unroll
    for (int k = 0; k < VEC; ++k) st(&q.v[k], o[k]);
    if constexpr (B % 4 == 0 && B <= 16 && (NT & 2)) {
        using V = typename RawVec<B>::type;
        __builtin_nontemporal_store(__builtin_bit_cast(V, q), reinterpret_cast<V*>(p));
    } else *reinterpret_cast<Pack<T, VEC>*>(p) = q;
}
// the pack stays packed up to here
template <typename T, int VEC> __device__ __forceinline__ void pin(Pack<T, VEC>& q) {
    constexpr int B = (int)sizeof(T) * VEC;
    if constexpr (B % 4 == 0) {
        unsigned w[B / 4];
        __builtin_memcpy(w, &q, B);
#pragma unroll
        for (int k = 0; k < B / 4; ++k) asm volatile("" : "+v"(w[k]));
        __builtin_memcpy(&q, w, B);
    }
}
// n_segment = 8 (every network of the reference): the whole column of a thread -- 8 packs per tensor -- is requested
// before the first one is used.  With one plane of look-ahead a wave had 0.5-1 KB in flight and the kernels sat at
// 3-4 TB/s, latency-bound; the arithmetic (and its order) is that of the generic walk below.
constexpr int kSeg = 8;

__device__ __forceinline__ bool my_column(const DimsT& d, int& n, int& c, int& e) {
    const int sub = threadIdx.x >> d.logE;
    e = threadIdx.x & (d.E - 1);
    const long long col = (long long)blockIdx.x * (kBlock >> d.logE) + sub;
    const bool valid = col < (long long)d.NB * d.C;
    const long long q = valid ? col : 0;
    c = (int)(q % d.C);
    n = (int)(q / d.C);
    return valid;
}

// BN: the input is relu(a[c] x + b[c]) of what is stored -- the block's training-mode bn1 + ReLU (rubiksnet/backbone.py:
// 129: out = relu(bn1(x)) feeds the AttentionShift in front of conv2), applied to each loaded value (the zero padding in
// t is that of the activation, not transformed); `ab` = [2][C]: a = gamma invstd, b = beta - mean a (rk_bn_stats_finish).
template <typename T, int VEC, bool BN = false>
__global__ __launch_bounds__(kBlock) void k_tshift3_forward(const T* __restrict__ x,
                                                            const typename Compute<T>::type* __restrict__ taps,
                                                            T* __restrict__ y, DimsT d, const float* __restrict__ ab = nullptr) {
    using CT = typename Compute<T>::type;
    int n, c, e;
    if (!my_column(d, n, c, e)) return;
    const CT s0 = taps[c * 3 + 0], s1 = taps[c * 3 + 1], s2 = taps[c * 3 + 2];
    CT pa = 1, pb = 0;
    if constexpr (BN) { pa = (CT)ab[c]; pb = (CT)ab[d.C + c]; }
    auto act = [&](CT (&v)[VEC]) {
        if constexpr (BN) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { const CT t = fmaf(pa, v[k], pb); v[k] = t > 0 ? t : 0; }
        }
    };
    const size_t tstride = (size_t)d.C * d.HW;
    const size_t base = ((size_t)n * d.S * d.C + c) * d.HW;
    for (int i = e * VEC; i < d.HW; i += d.E * VEC) {
        const T* xp = x + base + i;
        T* yp = y + base + i;
        CT prev[VEC], cur[VEC], nxt[VEC], out[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) prev[k] = 0;
        if (d.S == kSeg) {
            Pack<T, VEC> raw[kSeg];
#pragma unroll
            for (int t = 0; t < kSeg; ++t) raw[t] = load_stream<T, VEC, RK_TS_NT_FWD>(xp + (size_t)t * tstride);
            pin(raw[0]);
            unpack<T, VEC>(raw[0], cur);
            act(cur);
#pragma unroll
            for (int t = 0; t < kSeg; ++t) {
                if (t + 1 < kSeg) { pin(raw[t + 1 < kSeg ? t + 1 : 0]); unpack<T, VEC>(raw[t + 1 < kSeg ? t + 1 : 0], nxt); act(nxt); }
                else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) nxt[k] = 0;
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    out[k] = s0 * prev[k] + s1 * cur[k] + s2 * nxt[k];
                    prev[k] = cur[k];
                    cur[k] = nxt[k];
                }
                store_stream<T, VEC, RK_TS_NT_FWD>(yp + (size_t)t * tstride, out);
            }
            continue;
        }
        load_pack<T, VEC>(xp, cur);
        act(cur);
        for (int t = 0; t < d.S; ++t) {
            if (t + 1 < d.S) { load_pack<T, VEC>(xp + (size_t)(t + 1) * tstride, nxt); act(nxt); }
            else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) nxt[k] = 0;
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                out[k] = s0 * prev[k] + s1 * cur[k] + s2 * nxt[k];
                prev[k] = cur[k];
                cur[k] = nxt[k];
            }
            store_pack<T, VEC>(yp + (size_t)t * tstride, out);
        }
    }
}

// partials part[c][3][P], P = n_batch.  FUSED (fp32 partials): they are published as {value, tag} granules and the C
// extra blocks at the end of the grid sum them into gtaps (rk_dma.hpp: the row-sum inside the launch).
// BN (see the forward): x is the block's input BEFORE bn1; the activation relu(a x + b) is recomputed for the tap
// gradients, and what is written is dz = d(activation) masked by the ReLU -- together with this column's partial sums
// bred[c][n] = (sum dz, sum dz xhat) of BatchNorm's backward reduction (xhat = (x - mean) invstd), so that the BatchNorm
// backward is k_bn_bwd_finish_tiles + k_bn_bwd_dx_pre (rk_bn.hip): 3 tensor passes instead of 5.
struct BnBwdT {
    const float* ab;            // [2][C]
    const float* mean;          // [C]
    const float* invstd;        // [C]
    float2* bred;               // [C][n_batch] (nullptr with k12: the sums travel as granules 3 and 4 and are finished in the launch)
    float* k12;                 // [2][C] or nullptr: BatchNorm's backward constants (sum dz, sum dz xhat) / count ...
    float* dgamma; float* dbeta;    // ... and d(gamma), d(beta) [C], written by the finalizer waves (k_bn_bwd_finish_tiles' job)
    float inv_count;            // 1 / (NT H W)
    const void* gsmall;         // FORK: [NT, C, H/2, W/2] -- a second gradient of the activation, given at its even pixels only
    int W;                      //       (the projecting shortcut's, pointwise.fork_shortcut); plane width (W % VEC == 0, W even)
};
// FORK (with BN; downsampling blocks): the activation also feeds the stride-2 projecting shortcut, whose gradient -- a quarter-
// size tensor, the even pixels -- joins d(activation) here, BEFORE the ReLU mask and BatchNorm's sums (unfused: zeros + scatter
// + add + k_bn_bwd_reduce + k_bn_bwd_dx over the full-size tensor).
template <typename T, int VEC, bool FUSED, bool BN = false, bool FORK = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(sizeof(T) * VEC >= 16 || FORK ? 3 : 4)))
void k_tshift3_backward(const T* __restrict__ gy, const T* __restrict__ x,
                                                             const typename Compute<T>::type* __restrict__ taps,
                                                             T* __restrict__ gx,
                                                             typename Compute<T>::type* __restrict__ part, DimsT d,
                                                             dma::Fin fin, typename Compute<T>::type* __restrict__ gtaps,
                                                             BnBwdT bn = BnBwdT{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                                                nullptr, 0.f, nullptr, 0}) {
    using CT = typename Compute<T>::type;
    __shared__ CT red[BN ? 5 : 3][kBlock / kWave];
    if constexpr (FUSED) {
        if ((int)blockIdx.x >= fin.producers) {
            if (threadIdx.x < kWave) {
                const int cf = (int)blockIdx.x - fin.producers;
                const CT nanv = (CT)__uint_as_float(0x7fc00000u);
                if (BN && bn.k12) {                                  // + BatchNorm's two sums: k_bn_bwd_finish_tiles inside the launch
                    double s[5];
                    const bool ok = dma::fin_collect<5>(fin, cf, d.NB, s);
                    if (threadIdx.x == 0) {
                        for (int k = 0; k < 3; ++k) gtaps[cf * 3 + k] = ok ? (CT)s[k] : nanv;
                        bn.dbeta[cf] = ok ? (float)s[3] : (float)nanv;
                        bn.dgamma[cf] = ok ? (float)s[4] : (float)nanv;
                        bn.k12[cf] = ok ? (float)(s[3] * (double)bn.inv_count) : (float)nanv;
                        bn.k12[d.C + cf] = ok ? (float)(s[4] * (double)bn.inv_count) : (float)nanv;
                    }
                } else {
                    double s[3];
                    const bool ok = dma::fin_collect<3>(fin, cf, d.NB, s);
                    if (threadIdx.x == 0)
                        for (int k = 0; k < 3; ++k) gtaps[cf * 3 + k] = ok ? (CT)s[k] : nanv;
                }
            }
            return;
        }
    }
    int n, c, e;
    const bool valid = my_column(d, n, c, e);
    CT a0 = 0, a1 = 0, a2 = 0;
    CT z1 = 0, z2 = 0;                                    // BN: sum dz, sum dz xhat
    if (valid) {
        const CT s0 = taps[c * 3 + 0], s1 = taps[c * 3 + 1], s2 = taps[c * 3 + 2];
        CT pa = 1, pb = 0, mu = 0, iv = 1;
        if constexpr (BN) { pa = (CT)bn.ab[c]; pb = (CT)bn.ab[d.C + c]; mu = (CT)bn.mean[c]; iv = (CT)bn.invstd[c]; }
        // BN: v -> activation relu(pa v + pb) in place, hv <- xhat of the stored value
        auto act = [&](CT (&v)[VEC], CT (&hv)[VEC]) {
            if constexpr (BN) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    hv[k] = (v[k] - mu) * iv;
                    const CT t = fmaf(pa, v[k], pb);
                    v[k] = t > 0 ? t : 0;
                }
            }
        };
        // BN: out (= d activation at the current t) -> dz, and the two sums
        auto mask = [&](CT (&out)[VEC], const CT (&xc)[VEC], const CT (&hc)[VEC]) {
            if constexpr (BN) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const CT dz = xc[k] > 0 ? out[k] : 0;
                    out[k] = dz;
                    z1 += dz;
                    z2 = fmaf(dz, hc[k], z2);
                }
            }
        };
        const size_t tstride = (size_t)d.C * d.HW;
        const size_t base = ((size_t)n * d.S * d.C + c) * d.HW;
        for (int i = e * VEC; i < d.HW; i += d.E * VEC) {
            const T* xp = x + base + i;
            const T* gp = gy + base + i;
            T* op = gx + base + i;
            CT xprev[VEC], xcur[VEC], xnxt[VEC], gprev[VEC], gcur[VEC], gnxt[VEC], out[VEC];
            CT hcur[VEC], hnxt[VEC];                      // BN: xhat of the current / next time step
            // FORK: this pack's even elements have a partner in the small gradient when its row is even
            constexpr int HV = VEC >= 2 ? VEC / 2 : 1;
            bool fork_row = false;
            const T* gsp = nullptr;
            if constexpr (FORK && VEC >= 2) {
                const int h = i / bn.W, w = i - h * bn.W;
                fork_row = (h & 1) == 0;
                gsp = (const T*)bn.gsmall + ((size_t)n * d.S * d.C + c) * (size_t)(d.HW / 4) + (size_t)(h >> 1) * (bn.W >> 1) + (w >> 1);
            }
            const size_t sstride = (size_t)d.C * (d.HW / 4);
            auto join = [&](CT (&o)[VEC], const Pack<T, HV>& sp) {
                if constexpr (FORK && VEC >= 2) {
                    if (fork_row) {
                        CT sv[HV];
                        unpack<T, HV>(sp, sv);
#pragma unroll
                        for (int k = 0; k < HV; ++k) o[2 * k] += sv[k];
                    }
                }
            };
#pragma unroll
            for (int k = 0; k < VEC; ++k) { hcur[k] = 0; hnxt[k] = 0; }
#pragma unroll
            for (int k = 0; k < VEC; ++k) { xprev[k] = 0; gprev[k] = 0; }
            if (d.S == kSeg) {
                // The tap sums regrouped by the x they use: a0 += gy[t+1] x[t], a1 += gy[t] x[t], a2 += gy[t-1] x[t] -- the same
                // products in the same order as the walk below (whose t = 0 / t = S - 1 terms against the zero padding are
                // +-0), but only ONE time step of x is live (the walk carries three, and the fused form their xhat as well:
                // 256 VGPRs, one wave per SIMD).
                Pack<T, VEC> xr[kSeg], gr[kSeg];
#pragma unroll
                for (int t = 0; t < kSeg; ++t) {
                    gr[t] = load_stream<T, VEC, RK_TS_NT_BWD>(gp + (size_t)t * tstride);
                    xr[t] = load_stream<T, VEC, RK_TS_NT_BWD>(xp + (size_t)t * tstride);
                }
                // (FORK: the small gradient's packs are requested two steps ahead, not all up front: with 8 more packs live the
                // 16-byte form spilled 650 bytes per lane)
                Pack<T, HV> sr[3];
                if constexpr (FORK && VEC >= 2) {
                    sr[0] = load_raw<T, HV>(fork_row ? gsp : (const T*)bn.gsmall);
                    sr[1] = load_raw<T, HV>(fork_row ? gsp + sstride : (const T*)bn.gsmall);
                }
                pin(gr[0]);
                unpack<T, VEC>(gr[0], gcur);
#pragma unroll
                for (int t = 0; t < kSeg; ++t) {
                    pin(xr[t]);
                    unpack<T, VEC>(xr[t], xcur);
                    if constexpr (BN) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            hcur[k] = (xcur[k] - mu) * iv;
                            const CT v = fmaf(pa, xcur[k], pb);
                            xcur[k] = v > 0 ? v : 0;
                        }
                    }
                    if (t + 1 < kSeg) {
                        pin(gr[t + 1 < kSeg ? t + 1 : 0]);
                        unpack<T, VEC>(gr[t + 1 < kSeg ? t + 1 : 0], gnxt);
                    } else {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) gnxt[k] = 0;
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        out[k] = s0 * gnxt[k] + s1 * gcur[k] + s2 * gprev[k];
                        a0 += gnxt[k] * xcur[k];
                        a1 += gcur[k] * xcur[k];
                        a2 += gprev[k] * xcur[k];
                    }
                    if constexpr (FORK && VEC >= 2) {
                        if (t + 2 < kSeg) sr[(t + 2) % 3] = load_raw<T, HV>(fork_row ? gsp + (size_t)(t + 2) * sstride : (const T*)bn.gsmall);
                        join(out, sr[t % 3]);
                    }
                    mask(out, xcur, hcur);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { gprev[k] = gcur[k]; gcur[k] = gnxt[k]; }
                    store_stream<T, VEC, RK_TS_NT_BWD>(op + (size_t)t * tstride, out);
                }
                continue;
            }
            load_pack<T, VEC>(xp, xcur);
            act(xcur, hcur);
            load_pack<T, VEC>(gp, gcur);
            for (int t = 0; t < d.S; ++t) {
                if (t + 1 < d.S) {
                    load_pack<T, VEC>(xp + (size_t)(t + 1) * tstride, xnxt);
                    act(xnxt, hnxt);
                    load_pack<T, VEC>(gp + (size_t)(t + 1) * tstride, gnxt);
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { xnxt[k] = 0; gnxt[k] = 0; }
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    // y[t'] uses x[t'-1+j] with tap j  =>  gx[t] = s0*gy[t+1] + s1*gy[t] + s2*gy[t-1]
                    out[k] = s0 * gnxt[k] + s1 * gcur[k] + s2 * gprev[k];
                    a0 += gcur[k] * xprev[k];
                    a1 += gcur[k] * xcur[k];
                    a2 += gcur[k] * xnxt[k];
                }
                if constexpr (FORK && VEC >= 2) {
                    if (fork_row) join(out, load_raw<T, HV>(gsp + (size_t)t * sstride));
                }
                mask(out, xcur, hcur);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    xprev[k] = xcur[k]; xcur[k] = xnxt[k]; hcur[k] = hnxt[k];
                    gprev[k] = gcur[k]; gcur[k] = gnxt[k];
                }
                store_pack<T, VEC>(op + (size_t)t * tstride, out);
            }
        }
    }
    const bool bnfin = BN && bn.k12 != nullptr;                      // (uniform)
    if constexpr (BN) {
        z1 = group_sum(z1, d.E, red[3]);
        z2 = group_sum(z2, d.E, red[4]);
        if (valid && e == 0 && !bnfin) bn.bred[(size_t)c * d.NB + n] = make_float2((float)z1, (float)z2);
    }
    a0 = group_sum(a0, d.E, red[0]);
    a1 = group_sum(a1, d.E, red[1]);
    a2 = group_sum(a2, d.E, red[2]);
    if (valid && e == 0) {
        const size_t at = (size_t)c * (bnfin ? 5 : 3) * d.NB + n;
        if constexpr (FUSED) {
            dma::fin_publish(fin, at, (float)a0);
            dma::fin_publish(fin, at + d.NB, (float)a1);
            dma::fin_publish(fin, at + 2 * d.NB, (float)a2);
            if (bnfin) {
                dma::fin_publish(fin, at + 3 * (size_t)d.NB, (float)z1);
                dma::fin_publish(fin, at + 4 * (size_t)d.NB, (float)z2);
            }
        } else {
            part[at] = a0;
            part[at + d.NB] = a1;
            part[at + 2 * d.NB] = a2;
        }
    }
}

template <typename CT>
__global__ __launch_bounds__(kBlock) void k_tshift3_finalize(const CT* __restrict__ part, CT* __restrict__ gtaps,
                                                             int P) {
    __shared__ double red[3][kBlock / kWave];
    const int c = blockIdx.x;
    const CT* p = part + (size_t)c * 3 * P;
    double s[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
        for (int i = threadIdx.x; i < P; i += kBlock) s[k] += (double)p[(size_t)k * P + i];
    for (int k = 0; k < 3; ++k) s[k] = group_sum(s[k], kBlock, red[k]);
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) gtaps[c * 3 + k] = (CT)s[k];
}


// ---- the [C,3] half of AttentionShift (attention_shift.py:29-30): taps = softmax((w / (std(w) + 1e-6)) / T) over the
// three taps of a channel, std unbiased (torch.std default).  One thread per channel; in PyTorch this is ~15 tiny
// kernels forward and ~25 backward per layer (2 000 launches per Large-AQ train step).
__device__ __forceinline__ void soft_taps_one(const float* __restrict__ w, float T, float* __restrict__ taps, int c);
__device__ __forceinline__ void soft_taps_grad_one(const float* __restrict__ w, float T, const float* __restrict__ taps,
                                                   const float* __restrict__ gtaps, float* __restrict__ gw, int c);
__global__ __launch_bounds__(kBlock) void k_soft_taps_forward(const float* __restrict__ w, const float* __restrict__ Tp,
                                                              float* __restrict__ taps, int C) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c < C) soft_taps_one(w, Tp[0], taps, c);
}

// d(w) from d(taps): softmax backward, then z_i = w_i a(w) with a = 1 / (T (std + eps)), d std / d w_k = (w_k - m) / (2 std)
// (0 / 0 = NaN when the three weights are equal, as torch.std's backward)
__global__ __launch_bounds__(kBlock) void k_soft_taps_backward(const float* __restrict__ w, const float* __restrict__ Tp,
                                                               const float* __restrict__ taps,
                                                               const float* __restrict__ gtaps, float* __restrict__ gw,
                                                               int C) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c < C) soft_taps_grad_one(w, Tp[0], taps, gtaps, gw, c);
}

// Every AttentionShift layer of a network in ONE launch each way (attention_shift.presoftened: 51 layers in Large-AQ, 102
// launches of ~4 us between dependent kernels of a train step): blockIdx.y = the layer; a layer's taps / gradients live at
// channel offset `off` of concatenated [sum C][3] buffers, its weight and temperature stay where the module keeps them.
struct SoftJob { const float* w; const float* T; long long off; int C; int pad; };
__device__ __forceinline__ void soft_taps_one(const float* __restrict__ w, float T, float* __restrict__ taps, int c) {
    const float w0 = w[3 * c], w1 = w[3 * c + 1], w2 = w[3 * c + 2];
    const float m = (w0 + w1 + w2) / 3.0f;
    const float d0 = w0 - m, d1 = w1 - m, d2 = w2 - m;
    const float sd = sqrtf((d0 * d0 + d1 * d1 + d2 * d2) / 2.0f);
    const float den = sd + 1e-6f;
    const float z0 = w0 / den / T, z1 = w1 / den / T, z2 = w2 / den / T;
    const float zm = fmaxf(z0, fmaxf(z1, z2));
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm), e2 = expf(z2 - zm);
    const float es = e0 + e1 + e2;
    taps[3 * c] = e0 / es;
    taps[3 * c + 1] = e1 / es;
    taps[3 * c + 2] = e2 / es;
}
__device__ __forceinline__ void soft_taps_grad_one(const float* __restrict__ w, float T, const float* __restrict__ taps,
                                                   const float* __restrict__ gtaps, float* __restrict__ gw, int c) {
    const float w0 = w[3 * c], w1 = w[3 * c + 1], w2 = w[3 * c + 2];
    const float p0 = taps[3 * c], p1 = taps[3 * c + 1], p2 = taps[3 * c + 2];
    const float g0 = gtaps[3 * c], g1 = gtaps[3 * c + 1], g2 = gtaps[3 * c + 2];
    const float dot = g0 * p0 + g1 * p1 + g2 * p2;
    const float z0 = p0 * (g0 - dot), z1 = p1 * (g1 - dot), z2 = p2 * (g2 - dot);   // dL/dz
    const float m = (w0 + w1 + w2) / 3.0f;
    const float d0 = w0 - m, d1 = w1 - m, d2 = w2 - m;
    const float sd = sqrtf((d0 * d0 + d1 * d1 + d2 * d2) / 2.0f);
    const float den = sd + 1e-6f;
    const float a = 1.0f / (T * den);
    const float zw = z0 * w0 + z1 * w1 + z2 * w2;
    const float k = -zw / (T * den * den) / (2.0f * sd);
    gw[3 * c] = a * z0 + k * d0;
    gw[3 * c + 1] = a * z1 + k * d1;
    gw[3 * c + 2] = a * z2 + k * d2;
}
__global__ __launch_bounds__(kBlock) void k_soft_taps_many_forward(const SoftJob* __restrict__ jobs, float* __restrict__ taps) {
    const SoftJob j = jobs[blockIdx.y];
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c < j.C) soft_taps_one(j.w, j.T[0], taps + 3 * j.off, c);
}
__global__ __launch_bounds__(kBlock) void k_soft_taps_many_backward(const SoftJob* __restrict__ jobs, const float* __restrict__ taps,
                                                                    const float* __restrict__ gtaps, float* __restrict__ gw) {
    const SoftJob j = jobs[blockIdx.y];
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c < j.C) soft_taps_grad_one(j.w, j.T[0], taps + 3 * j.off, gtaps + 3 * j.off, gw + 3 * j.off, c);
}

int make_dimsT(DimsT& d, int NT, int S, int C, int HW, int vec) {
    if (NT <= 0 || S <= 0 || C <= 0 || HW <= 0 || NT % S != 0) return RK_ERR_BAD_DIMS;
    if ((long long)NT * C * HW > 0x7fffffffLL) return RK_ERR_BAD_DIMS;
    d.NB = NT / S; d.S = S; d.C = C; d.HW = HW;
    d.E = pow2_at_least((HW + vec - 1) / vec, kWave, kBlock);
    d.logE = (d.E == 64) ? 6 : (d.E == 128 ? 7 : 8);
    return RK_OK;
}

unsigned gridT(const DimsT& d) {
    const int per_block = kBlock / d.E;
    return (unsigned)(((long long)d.NB * d.C + per_block - 1) / per_block);
}

template <typename T> constexpr int max_vec() { return 16 / (int)sizeof(T); }   // 16 B per lane

// widest pack that divides H*W and that every pointer is aligned for
template <typename T>
int pick_vec(int HW, const void* a, const void* b, const void* c) {
    int v = max_vec<T>();
    const uintptr_t bits = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    while (v > 1 && (HW % v != 0 || bits % (v * sizeof(T)) != 0)) v >>= 1;
    return v;
}

template <typename T, int VEC>
int fwd_launch(const T* x, const typename Compute<T>::type* taps, T* y, int NT, int S, int C, int HW, hipStream_t stream) {
    DimsT d;
    if (int rc = make_dimsT(d, NT, S, C, HW, VEC)) return rc;
    hipLaunchKernelGGL((k_tshift3_forward<T, VEC>), dim3(gridT(d)), dim3(kBlock), 0, stream, x, taps, y, d);
    return launch_status();
}

template <typename T, int VEC>
int bwd_launch(const T* gy, const T* x, const typename Compute<T>::type* taps, T* gx,
               typename Compute<T>::type* gtaps, int NT, int S, int C, int HW,
               void* ws, hipStream_t stream) {
    using CT = typename Compute<T>::type;
    DimsT d;
    if (int rc = make_dimsT(d, NT, S, C, HW, VEC)) return rc;
    if constexpr (std::is_same<CT, float>::value) {
        dma::Fin fin;
        fin.gran = reinterpret_cast<unsigned long long*>(ws);
        dma::fin_arm(fin);
        fin.producers = (int)gridT(d);
        hipLaunchKernelGGL((k_tshift3_backward<T, VEC, true>), dim3(gridT(d) + C), dim3(kBlock), 0, stream, gy, x, taps, gx,
                           (CT*)ws, d, fin, gtaps);
    } else {
        dma::Fin fin{nullptr, 0u, 0, 0};
        hipLaunchKernelGGL((k_tshift3_backward<T, VEC, false>), dim3(gridT(d)), dim3(kBlock), 0, stream, gy, x, taps, gx,
                           (CT*)ws, d, fin, gtaps);
        hipLaunchKernelGGL((k_tshift3_finalize<CT>), dim3(C), dim3(kBlock), 0, stream, (const CT*)ws, gtaps, d.NB);
    }
    return launch_status();
}

template <typename T>
int forwardT(const void* x_, const typename Compute<T>::type* taps, void* y_, int NT, int S, int C, int HW, rk_stream_t stream_) {
    const T* x = (const T*)x_; T* y = (T*)y_;
    if (!x || !taps || !y) return RK_ERR_NULL_POINTER;
    hipStream_t stream = (hipStream_t)stream_;
    if (HW <= 0) return RK_ERR_BAD_DIMS;
    switch (pick_vec<T>(HW, x, y, nullptr)) {
        case 8: if constexpr (max_vec<T>() >= 8) return fwd_launch<T, 8>(x, taps, y, NT, S, C, HW, stream);
        case 4: if constexpr (max_vec<T>() >= 4) return fwd_launch<T, 4>(x, taps, y, NT, S, C, HW, stream);
        case 2: return fwd_launch<T, 2>(x, taps, y, NT, S, C, HW, stream);
        default: return fwd_launch<T, 1>(x, taps, y, NT, S, C, HW, stream);
    }
}

template <typename T>
int backwardT(const void* gy_, const void* x_, const typename Compute<T>::type* taps, void* gx_,
              typename Compute<T>::type* gtaps, int NT, int S, int C,
              int HW, void* ws, size_t ws_bytes, rk_stream_t stream_) {
    const T* gy = (const T*)gy_; const T* x = (const T*)x_; T* gx = (T*)gx_;
    if (!gy || !x || !taps || !gx || !gtaps) return RK_ERR_NULL_POINTER;
    if (HW <= 0 || S <= 0) return RK_ERR_BAD_DIMS;
    if (!ws || ws_bytes < rk_tshift3_backward_workspace_bytes(NT, S, C, HW)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    switch (pick_vec<T>(HW, gy, x, gx)) {
        case 8: if constexpr (max_vec<T>() >= 8) return bwd_launch<T, 8>(gy, x, taps, gx, gtaps, NT, S, C, HW, ws, stream);
        case 4: if constexpr (max_vec<T>() >= 4) return bwd_launch<T, 4>(gy, x, taps, gx, gtaps, NT, S, C, HW, ws, stream);
        case 2: return bwd_launch<T, 2>(gy, x, taps, gx, gtaps, NT, S, C, HW, ws, stream);
        default: return bwd_launch<T, 1>(gy, x, taps, gx, gtaps, NT, S, C, HW, ws, stream);
    }
}

// ---- the same with the block's bn1 + ReLU folded in (k_tshift3_forward / _backward <.., BN = true>) ----
template <typename T, int VEC>
int fwd_bn_launch(const T* x, const float* taps, const float* ab, T* y, int NT, int S, int C, int HW, hipStream_t stream) {
    DimsT d;
    if (int rc = make_dimsT(d, NT, S, C, HW, VEC)) return rc;
    hipLaunchKernelGGL((k_tshift3_forward<T, VEC, true>), dim3(gridT(d)), dim3(kBlock), 0, stream, x, taps, y, d, ab);
    return launch_status();
}
template <typename T, int VEC>
int bwd_bn_launch(const T* gy, const T* x, const float* taps, const BnBwdT& bn, T* dz, float* gtaps, int NT, int S, int C,
                  int HW, void* ws, hipStream_t stream) {
    DimsT d;
    if (int rc = make_dimsT(d, NT, S, C, HW, VEC)) return rc;
    dma::Fin fin;
    fin.gran = reinterpret_cast<unsigned long long*>(ws);
    dma::fin_arm(fin);
    fin.producers = (int)gridT(d);
    if (bn.gsmall) {
        if constexpr (VEC >= 2)
            hipLaunchKernelGGL((k_tshift3_backward<T, VEC, true, true, true>), dim3(gridT(d) + C), dim3(kBlock), 0, stream, gy, x,
                               taps, dz, (float*)ws, d, fin, gtaps, bn);
        else return RK_ERR_UNSUPPORTED;
    } else {
        hipLaunchKernelGGL((k_tshift3_backward<T, VEC, true, true>), dim3(gridT(d) + C), dim3(kBlock), 0, stream, gy, x, taps, dz,
                           (float*)ws, d, fin, gtaps, bn);
    }
    return launch_status();
}
template <typename T>
int forward_bnT(const void* x_, const float* taps, const float* ab, void* y_, int NT, int S, int C, int HW,
                rk_stream_t stream_) {
    const T* x = (const T*)x_; T* y = (T*)y_;
    if (!x || !taps || !ab || !y) return RK_ERR_NULL_POINTER;
    if (HW <= 0) return RK_ERR_BAD_DIMS;
    hipStream_t stream = (hipStream_t)stream_;
    switch (pick_vec<T>(HW, x, y, nullptr)) {
        case 8: if constexpr (max_vec<T>() >= 8) return fwd_bn_launch<T, 8>(x, taps, ab, y, NT, S, C, HW, stream);
        case 4: return fwd_bn_launch<T, 4>(x, taps, ab, y, NT, S, C, HW, stream);
        case 2: return fwd_bn_launch<T, 2>(x, taps, ab, y, NT, S, C, HW, stream);
        default: return fwd_bn_launch<T, 1>(x, taps, ab, y, NT, S, C, HW, stream);
    }
}
template <typename T>
int backward_bnT(const void* gy_, const void* x_, const float* taps, const float* ab, const float* mean, const float* invstd,
                 void* dz_, float* gtaps, void* bred, int NT, int S, int C, int HW, void* ws, size_t ws_bytes,
                 rk_stream_t stream_, float* k12 = nullptr, float* dgamma = nullptr, float* dbeta = nullptr,
                 const void* gsmall = nullptr, int W = 0) {
    const T* gy = (const T*)gy_; const T* x = (const T*)x_; T* dz = (T*)dz_;
    if (!gy || !x || !taps || !ab || !mean || !invstd || !dz || !gtaps) return RK_ERR_NULL_POINTER;
    if (k12 ? (!dgamma || !dbeta) : !bred) return RK_ERR_NULL_POINTER;
    if (HW <= 0 || S <= 0) return RK_ERR_BAD_DIMS;
    const size_t need = rk_tshift3_backward_workspace_bytes(NT, S, C, HW);
    if (!ws || ws_bytes < (k12 ? need / 3 * 5 : need)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const BnBwdT bn{ab, mean, invstd, (float2*)bred, k12, dgamma, dbeta, (float)(1.0 / ((double)NT * HW)), gsmall, W};
    int vec = pick_vec<T>(HW, gy, x, dz);
    if (gsmall) {                                               // a pack inside one row, its even elements at even columns
        if (W <= 0 || W % 2 || HW % W || (HW / W) % 2) return RK_ERR_BAD_DIMS;
        // (16-bit storage: 4 elements per pack at most -- the 8-element FORK instance compiles to 570 bytes of scratch per lane
        // whatever the register budget)
        if (sizeof(T) == 2 && vec > 4) vec = 4;
        while (vec > 1 && (W % vec != 0 || ((uintptr_t)gsmall % ((vec / 2) * sizeof(T))) != 0)) vec >>= 1;
        if (vec < 2) return RK_ERR_UNSUPPORTED;
    }
    switch (vec) {
        case 8: if constexpr (max_vec<T>() >= 8) return bwd_bn_launch<T, 8>(gy, x, taps, bn, dz, gtaps, NT, S, C, HW, ws, stream);
        case 4: return bwd_bn_launch<T, 4>(gy, x, taps, bn, dz, gtaps, NT, S, C, HW, ws, stream);
        case 2: return bwd_bn_launch<T, 2>(gy, x, taps, bn, dz, gtaps, NT, S, C, HW, ws, stream);
        default: return bwd_bn_launch<T, 1>(gy, x, taps, bn, dz, gtaps, NT, S, C, HW, ws, stream);
    }
}

}  // namespace

extern "C" {

size_t rk_tshift3_backward_workspace_bytes(int NT, int S, int C, int HW) {
    (void)HW;
    if (NT <= 0 || S <= 0 || C <= 0) return 0;
    return (size_t)C * 3 * (size_t)(NT / S) * 16;  // fp32 partials are 16-byte granule pairs (rk_dma.hpp); fp64 ones use half
}

#define RK_DEF_TAP(SFX, TYPE, CTYPE, TAPT)                                                                     \
    int rk_tshift3_forward_##SFX(const CTYPE* x, const TAPT* taps, CTYPE* y, int NT, int S, int C, int HW,     \
                                 rk_stream_t stream) {                                                         \
        return forwardT<TYPE>(x, taps, y, NT, S, C, HW, stream);                                               \
    }                                                                                                          \
    int rk_tshift3_backward_##SFX(const CTYPE* gy, const CTYPE* x, const TAPT* taps, CTYPE* gx, TAPT* gtaps,   \
                                  int NT, int S, int C, int HW, void* ws, size_t ws_bytes, rk_stream_t stream) { \
        return backwardT<TYPE>(gy, x, taps, gx, gtaps, NT, S, C, HW, ws, ws_bytes, stream);                    \
    }
RK_DEF_TAP(f32, float, float, float)
RK_DEF_TAP(f64, double, double, double)
RK_DEF_TAP(f16, __half, void, float)
RK_DEF_TAP(bf16, __hip_bfloat16, void, float)
#undef RK_DEF_TAP

// y = tshift3(relu(a[c] x + b[c])): the training-mode bn1 + ReLU of a block folded into its temporal 3-tap filter (ab = [2][C]
// from rk_bn_stats_finish_*); backward: gy -> dz = d(relu(a x + b)) masked by the ReLU (x = the block input before bn1),
// gtaps, and bred[C][NT / S] = (sum dz, sum dz xhat) per (channel, clip) for rk_bn_bwd_finish_tiles_f32 (tiles = NT / S)
int rk_tshift3_bn_forward_f32(const float* x, const float* taps, const float* ab, float* y, int NT, int S, int C, int HW,
                              rk_stream_t stream) {
    return forward_bnT<float>(x, taps, ab, y, NT, S, C, HW, stream);
}
int rk_tshift3_bn_forward_bf16(const void* x, const float* taps, const float* ab, void* y, int NT, int S, int C, int HW,
                               rk_stream_t stream) {
    return forward_bnT<__hip_bfloat16>(x, taps, ab, y, NT, S, C, HW, stream);
}
int rk_tshift3_bn_backward_f32(const float* gy, const float* x, const float* taps, const float* ab, const float* save_mean,
                               const float* save_invstd, float* dz, float* gtaps, void* bred, int NT, int S, int C, int HW,
                               void* ws, size_t ws_bytes, rk_stream_t stream) {
    return backward_bnT<float>(gy, x, taps, ab, save_mean, save_invstd, dz, gtaps, bred, NT, S, C, HW, ws, ws_bytes, stream);
}
int rk_tshift3_bn_backward_bf16(const void* gy, const void* x, const float* taps, const float* ab, const float* save_mean,
                                const float* save_invstd, void* dz, float* gtaps, void* bred, int NT, int S, int C, int HW,
                                void* ws, size_t ws_bytes, rk_stream_t stream) {
    return backward_bnT<__hip_bfloat16>(gy, x, taps, ab, save_mean, save_invstd, dz, gtaps, bred, NT, S, C, HW, ws, ws_bytes,
                                        stream);
}

// The same with BatchNorm's backward constants finished inside the launch (no bred, no rk_bn_bwd_finish_tiles_f32): k12 [2][C] =
// (sum dz, sum dz xhat) / (NT HW), dgamma / dbeta [C]; workspace of rk_tshift3_bn_backward_fin_workspace_bytes() bytes.
size_t rk_tshift3_bn_backward_fin_workspace_bytes(int NT, int S, int C, int HW) {
    return rk_tshift3_backward_workspace_bytes(NT, S, C, HW) / 3 * 5;
}
int rk_tshift3_bn_backward_fin_f32(const float* gy, const float* x, const float* taps, const float* ab, const float* save_mean,
                                   const float* save_invstd, float* dz, float* gtaps, float* k12, float* dgamma, float* dbeta,
                                   int NT, int S, int C, int HW, void* ws, size_t ws_bytes, rk_stream_t stream) {
    if (!k12) return RK_ERR_NULL_POINTER;
    return backward_bnT<float>(gy, x, taps, ab, save_mean, save_invstd, dz, gtaps, nullptr, NT, S, C, HW, ws, ws_bytes, stream,
                               k12, dgamma, dbeta);
}
int rk_tshift3_bn_backward_fin_bf16(const void* gy, const void* x, const float* taps, const float* ab, const float* save_mean,
                                    const float* save_invstd, void* dz, float* gtaps, float* k12, float* dgamma, float* dbeta,
                                    int NT, int S, int C, int HW, void* ws, size_t ws_bytes, rk_stream_t stream) {
    if (!k12) return RK_ERR_NULL_POINTER;
    return backward_bnT<__hip_bfloat16>(gy, x, taps, ab, save_mean, save_invstd, dz, gtaps, nullptr, NT, S, C, HW, ws, ws_bytes,
                                        stream, k12, dgamma, dbeta);
}

// ... and with a second gradient of the activation joined before the ReLU mask: gsmall [NT, C, H/2, W/2] = the gradient of
// the activation's even pixels (the stride-2 projecting shortcut of a downsampling block); W = the plane's width (even, H even)
int rk_tshift3_bn_backward_fork_f32(const float* gy, const float* x, const float* taps, const float* ab, const float* save_mean,
                                    const float* save_invstd, const float* gsmall, float* dz, float* gtaps, float* k12,
                                    float* dgamma, float* dbeta, int NT, int S, int C, int H, int W, void* ws, size_t ws_bytes,
                                    rk_stream_t stream) {
    if (!k12 || !gsmall) return RK_ERR_NULL_POINTER;
    return backward_bnT<float>(gy, x, taps, ab, save_mean, save_invstd, dz, gtaps, nullptr, NT, S, C, H * W, ws, ws_bytes, stream,
                               k12, dgamma, dbeta, gsmall, W);
}
int rk_tshift3_bn_backward_fork_bf16(const void* gy, const void* x, const float* taps, const float* ab, const float* save_mean,
                                     const float* save_invstd, const void* gsmall, void* dz, float* gtaps, float* k12,
                                     float* dgamma, float* dbeta, int NT, int S, int C, int H, int W, void* ws, size_t ws_bytes,
                                     rk_stream_t stream) {
    if (!k12 || !gsmall) return RK_ERR_NULL_POINTER;
    return backward_bnT<__hip_bfloat16>(gy, x, taps, ab, save_mean, save_invstd, dz, gtaps, nullptr, NT, S, C, H * W, ws, ws_bytes,
                                        stream, k12, dgamma, dbeta, gsmall, W);
}

int rk_soft_taps_forward_f32(const float* weight, const float* T, float* taps, int C, rk_stream_t stream) {
    if (!weight || !T || !taps) return RK_ERR_NULL_POINTER;
    if (C <= 0) return RK_ERR_BAD_DIMS;
    hipLaunchKernelGGL(k_soft_taps_forward, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, weight, T,
                       taps, C);
    return launch_status();
}
int rk_soft_taps_backward_f32(const float* weight, const float* T, const float* taps, const float* gtaps, float* gweight,
                              int C, rk_stream_t stream) {
    if (!weight || !T || !taps || !gtaps || !gweight) return RK_ERR_NULL_POINTER;
    if (C <= 0) return RK_ERR_BAD_DIMS;
    hipLaunchKernelGGL(k_soft_taps_backward, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, weight, T,
                       taps, gtaps, gweight, C);
    return launch_status();
}
// jobs: device array of n records {const float* weight; const float* T; int64 off; int C; int pad} (32 bytes): layer i's taps /
// d(taps) / d(weight) are rows off .. off + C of the concatenated [sum C][3] buffers; max_c = the largest C
int rk_soft_taps_many_forward_f32(const void* jobs, int n, float* taps, int max_c, rk_stream_t stream) {
    if (!jobs || !taps) return RK_ERR_NULL_POINTER;
    if (n <= 0 || n > 65535 || max_c <= 0 || ((uintptr_t)jobs & 7)) return RK_ERR_BAD_DIMS;
    static_assert(sizeof(SoftJob) == 32, "the record layout attention_shift.py writes");
    hipLaunchKernelGGL(k_soft_taps_many_forward, dim3((max_c + kBlock - 1) / kBlock, n), dim3(kBlock), 0, (hipStream_t)stream,
                       (const SoftJob*)jobs, taps);
    return launch_status();
}
int rk_soft_taps_many_backward_f32(const void* jobs, int n, const float* taps, const float* gtaps, float* gweight, int max_c,
                                   rk_stream_t stream) {
    if (!jobs || !taps || !gtaps || !gweight) return RK_ERR_NULL_POINTER;
    if (n <= 0 || n > 65535 || max_c <= 0 || ((uintptr_t)jobs & 7)) return RK_ERR_BAD_DIMS;
    hipLaunchKernelGGL(k_soft_taps_many_backward, dim3((max_c + kBlock - 1) / kBlock, n), dim3(kBlock), 0, (hipStream_t)stream,
                       (const SoftJob*)jobs, taps, gtaps, gweight);
    return launch_status();
}

}  // extern "C"
