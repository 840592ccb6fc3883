// rk2d_column.hpp -- RubiksShift2D "column" kernels: any stride / padding, any H x W, every storage type,
// quantize off.  They take what the streaming kernels (rk2d_dma.hpp / rk2d_stage.hpp) do not: the stride-2
// layers and the 14x14 / 7x7 planes (W % 4 != 0) of the -aq networks, which on the per-plane kernels of
// rk2d_generic.hpp were 14 % of the Tiny-AQ bf16 train step (two passes for the backward, 4 predicated taps
// with % and / per element, geometry recomputed for every plane).
//
// Same idea as rk3d_column.hpp without the T coupling: a group of E threads owns (channel, chunk of the plane)
// and walks a group of frames, so the channel's shift, the per-element tap offsets and weights are computed
// once; taps are per-element global loads served by L1/L2.
//   forward : interp2d of the 4 taps (rubiks2d_kernels.cu:60-66, :94-146) -- bit-identical to the oracle.
//   backward: d(x) + d(shift) in ONE pass, adjoint form on the input side with the negated shift (fl', r'):
//             gx = interp2d of the gy taps that exist ((h+pH+fl'H+j) % sH == 0, rubiks2d_kernels.cu:298-300)
//             -- the tree of K8, bit-identical -- and gH = sum x (la - lb), gW = sum x (colA - colB) from the same
//             taps (la / lb: W-lerps of the two tap rows, colA / colB: H-lerps of the two tap columns).
//             SINGLE (both strides >= 2): an input element has at most one tap and the tree collapses exactly
//             to (v wj) wk.
// Channels within 1e-7 of an integer shift (central-difference branch, :189-253) run the per-element reference
// formulation of rk2d_generic.hpp for their frames.  Partials part[c][2][P], P = groups * chunks.
#pragma once
#include "rk2d_generic.hpp"

namespace rk {
namespace col2d {

using namespace g2d;

struct C2Dims {
    Dims2 d;
    int E, logE;          // threads per group (64 / 128 / 256)
    int nchunks;          // chunks per plane (E * M elements each)
    int M;                // elements per thread (1 or 4)
    int FG, ngroups;      // frames per group, groups
};

struct Col2Id { int c, chunk, g; bool valid; };

__device__ __forceinline__ Col2Id my_column2(const C2Dims& cd, int& e) {
    const int sub = threadIdx.x >> cd.logE;
    e = threadIdx.x & (cd.E - 1);
    const long long id = (long long)blockIdx.x * (kBlock >> cd.logE) + sub;     // (g, c, chunk) flattened
    Col2Id r;
    r.valid = id < (long long)cd.ngroups * cd.d.C * cd.nchunks;
    const long long q = r.valid ? id : 0;
    r.chunk = (int)(q % cd.nchunks);
    const long long col = q / cd.nchunks;
    r.c = (int)(col % cd.d.C);
    r.g = (int)(col / cd.d.C);
    return r;
}

template <typename T> struct alignas(sizeof(T) * 4) Quad { T v[4]; };
// a quad leaves as one streaming store (the result is written once and is far larger than L2; RK_C2_NT=0 at build time: plain)
#ifndef RK_C2_NT
#define RK_C2_NT 1
#endif
#ifndef RK_C2_BWD_WAVES
#define RK_C2_BWD_WAVES 3
#endif
#ifndef RK_C2_WIDE_BWD
#define RK_C2_WIDE_BWD 0
#endif
template <typename T> __device__ __forceinline__ void store_quad(T* p, const Quad<T>& q) {
    if constexpr (RK_C2_NT && sizeof(T) == 2) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(__builtin_bit_cast(u32x2, q), reinterpret_cast<u32x2*>(p));
    } else if constexpr (RK_C2_NT && sizeof(T) == 4) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, q), reinterpret_cast<u32x4*>(p));
    } else *reinterpret_cast<Quad<T>*>(p) = q;
}

// training fusion (bn2 + ReLU inside the shift, fused_bn.bn_relu_shift2d): the value the unfused path would have stored
// -- rounded to the storage type -- so that both paths see the same activation
// a 16-bit storage value given as bits
template <typename T> __device__ __forceinline__ float from_bits16(unsigned b) {
    if constexpr (std::is_same<T, __hip_bfloat16>::value) return __uint_as_float(b << 16);
    else return __half2float(__builtin_bit_cast(__half, (unsigned short)b));
}
template <typename T> __device__ __forceinline__ float as_stored(float v) { T t; st(&t, v); return ld(&t); }
template <typename T> __device__ __forceinline__ float bn_relu_of(float z, float a, float b) {
    return as_stored<T>(fmaxf(fmaf(a, z, b), 0.f));
}

// ------------------------------------------------------------------------------------ forward
// VEC (kM == 4, output planes of a multiple of 4 elements, aligned y): a thread owns 4 CONSECUTIVE outputs and stores
// them as one 8- / 16-byte access; frames are taken in batches, every tap of a batch requested before the first is used.
// BN: x holds z (bn2's input) and every tap is relu(a z + b) (ab = [2][C]), the activation that is never stored.
template <typename T, typename S, int kM, bool VEC = false, bool BN = false>
__global__ __launch_bounds__(kBlock) void k2d_forward_column(const T* __restrict__ x, const S* __restrict__ shift,
                                                             T* __restrict__ y, C2Dims cd,
                                                             const float* __restrict__ ab = nullptr) {
    using CT = typename Compute<T>::type;
    const Dims2& d = cd.d;
    int e;
    const Col2Id id = my_column2(cd, e);
    if (!id.valid) return;
    float bnA = 1.f, bnB = 0.f;
    if constexpr (BN) { bnA = ab[id.c]; bnB = ab[d.C + id.c]; }
    const CT offH = ld(shift + id.c), offW = ld(shift + d.C + id.c);
    const int iH = floor_fast(offH), iW = floor_fast(offW);
    const CT rH = offH - (CT)iH, rW = offW - (CT)iW;
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
    const size_t fsi = (size_t)d.C * HW, fso = (size_t)d.C * HWo;
    const int f0 = id.g * cd.FG, nf = min(cd.FG, d.N - f0);
    const T* xc = x + ((size_t)f0 * d.C + id.c) * HW;
    T* yc = y + ((size_t)f0 * d.C + id.c) * HWo;

    int o00[kM], oidx[kM];
    unsigned mask[kM];
#pragma unroll
    for (int m = 0; m < kM; ++m) {
        const int i = VEC ? id.chunk * cd.E * kM + e * kM + m : id.chunk * cd.E * kM + m * cd.E + e;
        oidx[m] = i < HWo ? i : -1;
        const int ii = i < HWo ? i : 0;
        const int ho = ii / d.Wo, wo = ii - ho * d.Wo;
        const int h0 = ho * d.sH - d.pH + iH, w0 = wo * d.sW - d.pW + iW;
        const bool mh0 = h0 >= 0 && h0 < d.H, mh1 = h0 + 1 >= 0 && h0 + 1 < d.H;
        const bool mw0 = w0 >= 0 && w0 < d.W, mw1 = w0 + 1 >= 0 && w0 + 1 < d.W;
        o00[m] = h0 * d.W + w0;
        mask[m] = (i < HWo) ? ((mh0 && mw0 ? 1u : 0u) | (mh0 && mw1 ? 2u : 0u) | (mh1 && mw0 ? 4u : 0u) |
                               (mh1 && mw1 ? 8u : 0u)) : 0u;
    }
    // WIDE (16-bit storage, stride (2,2) / pad 0, even planes, Wo % 4 == 0, shift floors in {-1, 0}: every downsampling layer of
    // the -aq networks): the 8 taps of a thread's 4 consecutive outputs are 8 CONSECUTIVE input elements in each of two rows, so
    // they arrive as two 16-byte loads (2-byte aligned: the hardware takes them) instead of 16 two-byte gathers; a window that
    // would start at column -1 starts at 0 and is moved up one element in registers (its first tap is masked anyway).
    constexpr bool S2W = VEC && kM == 4 && sizeof(T) == 2;
    const bool wide = S2W && d.sH == 2 && d.sW == 2 && d.pH == 0 && d.pW == 0 && (d.Wo & 3) == 0 && !(d.H & 1) && !(d.W & 1) &&
                      d.W >= 8 && (iH == 0 || iH == -1) && (iW == 0 || iW == -1);      // (uniform per thread group)
    if (wide) {
        const int i0 = id.chunk * cd.E * kM + e * kM;
        const int ii0 = i0 < HWo ? i0 : 0;
        const int ho = ii0 / d.Wo, wo0 = ii0 - ho * d.Wo;
        const int h0 = 2 * ho + iH, c0 = 2 * wo0 + iW;
        const bool up1 = c0 < 0;
        const int colL = up1 ? 0 : c0;
        const int wa = (h0 < 0 ? 0 : h0) * d.W + colL, wb = (h0 + 1) * d.W + colL;
        struct Win { unsigned a[4], b[4]; };
        auto load_win = [&](int k, Win& f) {
            const T* p = xc + (size_t)k * fsi;
            __builtin_memcpy(f.a, p + wa, 16);
            __builtin_memcpy(f.b, p + wb, 16);
        };
        auto use_win = [&](int k, const Win& f) {
            T* out = yc + (size_t)k * fso;
            Quad<T> oq;
#pragma unroll
            for (int m = 0; m < kM; ++m) {
                const unsigned ra = up1 ? ((f.a[m] << 16) | (m > 0 ? f.a[m > 0 ? m - 1 : 0] >> 16 : 0u)) : f.a[m];
                const unsigned rb = up1 ? ((f.b[m] << 16) | (m > 0 ? f.b[m > 0 ? m - 1 : 0] >> 16 : 0u)) : f.b[m];
                const unsigned mk = mask[m];
                CT t0 = (CT)from_bits16<T>(ra & 0xffffu), t1 = (CT)from_bits16<T>(ra >> 16);
                CT t2 = (CT)from_bits16<T>(rb & 0xffffu), t3 = (CT)from_bits16<T>(rb >> 16);
                if constexpr (BN) {
                    t0 = bn_relu_of<T>(t0, bnA, bnB); t1 = bn_relu_of<T>(t1, bnA, bnB);
                    t2 = bn_relu_of<T>(t2, bnA, bnB); t3 = bn_relu_of<T>(t3, bnA, bnB);
                }
                const CT p00 = (mk & 1u) ? t0 : (CT)0, p01 = (mk & 2u) ? t1 : (CT)0;
                const CT p10 = (mk & 4u) ? t2 : (CT)0, p11 = (mk & 8u) ? t3 : (CT)0;
                st(&oq.v[m & 3], interp2d(p00, p01, p10, p11, rH, rW));
            }
            if (oidx[0] >= 0) store_quad<T>(out + oidx[0], oq);
        };
        int k = 0;
        for (; k + 3 < nf; k += 4) {
            Win fr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) load_win(k + j, fr[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) use_win(k + j, fr[j]);
        }
        for (; k < nf; ++k) {
            Win f;
            load_win(k, f);
            use_win(k, f);
        }
        return;
    }
    struct Frame { CT q[kM][4]; };
    auto load_frame = [&](int k, Frame& f) {                              // addresses clamped into the plane, masked at use
        const T* p = xc + (size_t)k * fsi;
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            const unsigned mk = mask[m];
            f.q[m][0] = ld(p + ((mk & 1u) ? o00[m] : 0));
            f.q[m][1] = ld(p + ((mk & 2u) ? o00[m] + 1 : 0));
            f.q[m][2] = ld(p + ((mk & 4u) ? o00[m] + d.W : 0));
            f.q[m][3] = ld(p + ((mk & 8u) ? o00[m] + d.W + 1 : 0));
        }
    };
    auto use_frame = [&](int k, const Frame& f) {
        T* out = yc + (size_t)k * fso;
        Quad<T> oq;
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            const unsigned mk = mask[m];
            CT t0 = f.q[m][0], t1 = f.q[m][1], t2 = f.q[m][2], t3 = f.q[m][3];
            if constexpr (BN) {
                t0 = bn_relu_of<T>(t0, bnA, bnB); t1 = bn_relu_of<T>(t1, bnA, bnB);
                t2 = bn_relu_of<T>(t2, bnA, bnB); t3 = bn_relu_of<T>(t3, bnA, bnB);
            }
            const CT p00 = (mk & 1u) ? t0 : (CT)0, p01 = (mk & 2u) ? t1 : (CT)0;
            const CT p10 = (mk & 4u) ? t2 : (CT)0, p11 = (mk & 8u) ? t3 : (CT)0;
            const CT v = interp2d(p00, p01, p10, p11, rH, rW);
            if constexpr (VEC) st(&oq.v[m & 3], v);
            else if (oidx[m] >= 0) st(out + oidx[m], v);
        }
        if constexpr (VEC) {
            if (oidx[0] >= 0) store_quad<T>(out + oidx[0], oq);
        }
    };
    constexpr int kRegsPerFrame = kM * 4 * (int)(sizeof(CT) / 4);
    constexpr int kBatch = kRegsPerFrame * 4 <= 40 ? 4 : (kRegsPerFrame * 2 <= 32 ? 2 : 1);
    int k = 0;
    if constexpr (kBatch > 1) {
        for (; k + kBatch - 1 < nf; k += kBatch) {
            Frame fr[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) load_frame(k + j, fr[j]);
#pragma unroll
            for (int j = 0; j < kBatch; ++j) use_frame(k + j, fr[j]);
        }
    }
    for (; k < nf; ++k) {
        Frame f;
        load_frame(k, f);
        use_frame(k, f);
    }
}

// ----------------------------------------------------------------------------------- backward
// VEC (kM == 4, input planes of a multiple of 4 elements, 16-byte aligned tensors): a thread owns 4 CONSECUTIVE input
// elements, so x and gx -- the two big streams of a strided layer's backward -- move as one 8-byte (16-bit types) or
// 16-byte (fp32) access per thread and frame instead of four 2- / 4-byte ones; the gy taps stay scalar (L1-served).
// Frames are taken four at a time, every load of the four requested before the first is used.  Same arithmetic per
// element, bit-identical.
// BN (training fusion): x holds z = bn2's input.  d(shift) uses the recomputed activation relu(a z + b); what is stored is
// d(bn2's output) = d(activation) masked by the ReLU (rounded to the storage type first, as the unfused path stores it), and
// BatchNorm's two reduction sums over it (sum dz, sum dz (z - mean) invstd) leave as partials 2 and 3: part[c][4][P].
template <typename T, typename S, int kM, bool SINGLE, bool VEC = false, bool BN = false>
// (16-bit storage: 3 waves per SIMD -- left to itself the fused stride-2 instance takes 180 VGPRs, 2 waves: 420 -> 365 us at
// [256,72,112,112]; fp32 is faster left alone)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? RK_C2_BWD_WAVES : 2)))
void k2d_backward_column(const T* __restrict__ gy, const T* __restrict__ x,
                                                              const S* __restrict__ shift, T* gx,
                                                              typename Compute<T>::type* __restrict__ part,
                                                              C2Dims cd, const float4* __restrict__ abmi = nullptr) {
    using CT = typename Compute<T>::type;
    constexpr int ND = BN ? 4 : 2;
    __shared__ CT red[ND][kBlock / kWave];
    const Dims2& d = cd.d;
    int e;
    const Col2Id id = my_column2(cd, e);
    CT accH = 0, accW = 0, accB1 = 0, accB2 = 0;
    if (id.valid) {
        const CT s0 = ld(shift + id.c), s1 = ld(shift + d.C + id.c);
        const int f0 = id.g * cd.FG, nf = min(cd.FG, d.N - f0);
        const CT u0 = s0 - (CT)floor_fast(s0), u1 = s1 - (CT)floor_fast(s1);
        float4 bnp = make_float4(1.f, 0.f, 0.f, 1.f);
        if constexpr (BN) bnp = abmi[id.c];
        if (u0 < (CT)1e-7f || u1 < (CT)1e-7f) {
            // central-difference branch: the reference's per-element formulation; chunk 0 does the whole planes
            if (id.chunk == 0)
                for (int k = 0; k < nf; ++k) {
                    backward_input_plane2<T, false>(gy, shift, gx, d, f0 + k, id.c, e, cd.E);
                    if constexpr (BN) {
                        shift_grad_plane2<T>(gy, x, shift, d, f0 + k, id.c, e, cd.E, accH, accW,
                                             [&](CT z) { return (CT)bn_relu_of<T>((float)z, bnp.x, bnp.y); });
                        // mask what this thread just stored (the same elements: e, e + E, ...) and take bn2's sums
                        const size_t pl = ((size_t)(f0 + k) * d.C + id.c) * (d.H * d.W);
                        for (int i = e; i < d.H * d.W; i += cd.E) {
                            const float z = (float)ld(x + pl + i);
                            const float g = bn_relu_of<T>(z, bnp.x, bnp.y) > 0.f ? (float)ld(gx + pl + i) : 0.f;
                            st(gx + pl + i, (CT)g);
                            accB1 += g;
                            accB2 = fmaf(g, (z - bnp.z) * bnp.w, accB2);
                        }
                    } else {
                        shift_grad_plane2<T>(gy, x, shift, d, f0 + k, id.c, e, cd.E, accH, accW);
                    }
                }
        } else {
            const CT nH = -s0, nW = -s1;
            const int flH = floor_fast(nH), flW = floor_fast(nW);
            const CT rH = nH - (CT)flH, rW = nW - (CT)flW;                     // fl', r'
            const int HW = d.H * d.W, HWo = d.Ho * d.Wo;
            const size_t fsi = (size_t)d.C * HW, fso = (size_t)d.C * HWo;
            const T* xc = x + ((size_t)f0 * d.C + id.c) * HW;
            const T* gc = gy + ((size_t)f0 * d.C + id.c) * HWo;
            T* oc = gx + ((size_t)f0 * d.C + id.c) * HW;

            int iidx[kM], tap[kM][SINGLE ? 1 : 4];
            CT wj[kM], wk[kM], sj[kM], sk[kM];                                 // SINGLE: the tap's weights and signs
#pragma unroll
            for (int m = 0; m < kM; ++m) {
                const int i = VEC ? id.chunk * cd.E * kM + e * kM + m : id.chunk * cd.E * kM + m * cd.E + e;
                const bool live = i < HW;
                iidx[m] = live ? i : -1;
                const int ii = live ? i : 0;
                const int h = ii / d.W, w = ii - h * d.W;
                const int r0 = unmap2(h + d.pH + flH, d.sH, d.Ho), r1 = unmap2(h + d.pH + flH + 1, d.sH, d.Ho);
                const int c0 = unmap2(w + d.pW + flW, d.sW, d.Wo), c1 = unmap2(w + d.pW + flW + 1, d.sW, d.Wo);
                if (SINGLE) {
                    const int r = r0 >= 0 ? r0 : r1, c = c0 >= 0 ? c0 : c1;   // at most one of each exists
                    tap[m][0] = (live && r >= 0 && c >= 0) ? r * d.Wo + c : -1;
                    wj[m] = r0 >= 0 ? 1 - rH : rH;  sj[m] = r0 >= 0 ? (CT)1 : (CT)-1;
                    wk[m] = c0 >= 0 ? 1 - rW : rW;  sk[m] = c0 >= 0 ? (CT)1 : (CT)-1;
                } else {
                    tap[m][0] = (live && r0 >= 0 && c0 >= 0) ? r0 * d.Wo + c0 : -1;
                    tap[m][SINGLE ? 0 : 1] = (live && r0 >= 0 && c1 >= 0) ? r0 * d.Wo + c1 : -1;
                    tap[m][SINGLE ? 0 : 2] = (live && r1 >= 0 && c0 >= 0) ? r1 * d.Wo + c0 : -1;
                    tap[m][SINGLE ? 0 : 3] = (live && r1 >= 0 && c1 >= 0) ? r1 * d.Wo + c1 : -1;
                }
            }
            CT sH = 0, sW = 0, sB1 = 0, sB2 = 0;
            constexpr int NTAP = SINGLE ? 1 : 4;
            // WIDE (16-bit storage, both strides >= 2, a thread's 4 consecutive inputs in one row): the 4 taps are 2-3 CONSECUTIVE
            // gy elements of one row, so they arrive as ONE 8-byte load (2-byte aligned: the hardware takes it) at `gbase` and are
            // picked out by shifts -- 3 memory instructions per thread and frame instead of 6 (the kernel ran at 0.30-0.36 of 8 TB/s
            // with four two-byte gathers per thread)
            constexpr bool WIDE = SINGLE && VEC && kM == 4 && sizeof(T) == 2;
            // (measured: 420 -> 436 us at [256,72,112,112] -- the kernel is bound by its ~40 VALU operations per element, not by the
            // gathers; the path stays compiled for RK_C2_WIDE_BWD=1 builds)
            const bool wide = WIDE && RK_C2_WIDE_BWD && (d.W & 3) == 0 && HWo >= 4;     // (uniform; W % 4 != 0: a thread's inputs straddle rows)
            int gbase = 0;
            unsigned gsh[kM];
            if (wide) {
                int vmin = 0x7fffffff;
#pragma unroll
                for (int m = 0; m < kM; ++m) if (tap[m][0] >= 0 && tap[m][0] < vmin) vmin = tap[m][0];
                gbase = vmin == 0x7fffffff ? 0 : vmin;
                gbase = gbase < HWo - 4 ? gbase : HWo - 4;            // (the load stays inside the plane; host: HWo >= 4)
#pragma unroll
                for (int m = 0; m < kM; ++m) {
                    int ix = tap[m][0] >= 0 ? tap[m][0] - gbase : 0;
                    if (ix > 3) { ix = 0; tap[m][0] = -1; }           // (cannot happen with both strides 2: the taps span <= 3 elements)
                    gsh[m] = 16u * (unsigned)ix;
                }
            }
            struct Frame { CT xv[kM]; CT q[kM][WIDE ? 1 : NTAP]; unsigned long long gq; };
            auto load_frame = [&](int k, Frame& f) {                          // addresses clamped into the plane, values masked at use
                const T* p = gc + (size_t)k * fso;
                const T* xp = xc + (size_t)k * fsi;
                if constexpr (VEC) {
                    const Quad<T> xq = *reinterpret_cast<const Quad<T>*>(xp + (iidx[0] >= 0 ? iidx[0] : 0));
#pragma unroll
                    for (int m = 0; m < kM; ++m) f.xv[m] = ld(&xq.v[m]);
                } else {
#pragma unroll
                    for (int m = 0; m < kM; ++m) f.xv[m] = ld(xp + (iidx[m] >= 0 ? iidx[m] : 0));
                }
                if (wide) {
                    __builtin_memcpy(&f.gq, p + gbase, 8);
                } else {
#pragma unroll
                    for (int m = 0; m < kM; ++m)
#pragma unroll
                        for (int j = 0; j < NTAP; ++j) f.q[m][j] = ld(p + (tap[m][j] >= 0 ? tap[m][j] : 0));
                }
            };
            auto use_frame = [&](int k, const Frame& f) {
                T* out = oc + (size_t)k * fsi;
                Quad<T> oq;
#pragma unroll
                for (int m = 0; m < kM; ++m) {
                    CT xv = iidx[m] >= 0 ? f.xv[m] : (CT)0;
                    if constexpr (BN) xv = iidx[m] >= 0 ? (CT)bn_relu_of<T>((float)f.xv[m], bnp.x, bnp.y) : (CT)0;
                    CT Q, QH, QW;
                    if (SINGLE) {
                        CT v;
                        if (wide) v = tap[m][0] >= 0 ? (CT)from_bits16<T>((unsigned)(f.gq >> gsh[m]) & 0xffffu) : (CT)0;
                        else v = tap[m][0] >= 0 ? f.q[m][0] : (CT)0;
                        const CT vj = v * wj[m];
                        Q = vj * wk[m];                                      // = K8's interp2d with three zero taps
                        QH = sj[m] * (v * wk[m]);
                        QW = sk[m] * vj;
                    } else {
                        const CT q00 = tap[m][0] >= 0 ? f.q[m][0] : (CT)0;
                        const CT q01 = tap[m][NTAP > 1 ? 1 : 0] >= 0 ? f.q[m][NTAP > 1 ? 1 : 0] : (CT)0;
                        const CT q10 = tap[m][NTAP > 1 ? 2 : 0] >= 0 ? f.q[m][NTAP > 1 ? 2 : 0] : (CT)0;
                        const CT q11 = tap[m][NTAP > 1 ? 3 : 0] >= 0 ? f.q[m][NTAP > 1 ? 3 : 0] : (CT)0;
                        Q = interp2d(q00, q01, q10, q11, rH, rW);            // K8's tree, contraction off
                        QH = (q00 * (1 - rW) + q01 * rW) - (q10 * (1 - rW) + q11 * rW);
                        QW = ((1 - rH) * q00 + rH * q10) - ((1 - rH) * q01 + rH * q11);
                    }
                    sH += QH * xv;
                    sW += QW * xv;
                    if constexpr (BN) {
                        Q = xv > 0 ? (CT)as_stored<T>((float)Q) : (CT)0;
                        sB1 += Q;
                        sB2 = fmaf((float)Q, ((float)f.xv[m] - bnp.z) * bnp.w, (float)sB2);
                    }
                    if constexpr (VEC) st(&oq.v[m & 3], Q);
                    else if (iidx[m] >= 0) st(out + iidx[m], Q);
                }
                if constexpr (VEC) {
                    if (iidx[0] >= 0) store_quad<T>(out + iidx[0], oq);
                }
            };
            // frames in flight at once: as many (<= 4) as fit ~40 registers of loaded values
            constexpr int kRegsPerFrame = kM * (1 + NTAP) * (int)(sizeof(CT) / 4);
            constexpr int kBatch = kRegsPerFrame * 4 <= 40 ? 4 : (kRegsPerFrame * 2 <= 32 ? 2 : 1);
            int k = 0;
            if constexpr (kBatch > 1) {
                for (; k + kBatch - 1 < nf; k += kBatch) {
                    Frame fr[kBatch];
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) load_frame(k + j, fr[j]);
#pragma unroll
                    for (int j = 0; j < kBatch; ++j) use_frame(k + j, fr[j]);
                }
            }
            for (; k < nf; ++k) {
                Frame f;
                load_frame(k, f);
                use_frame(k, f);
            }
            accH = sH; accW = sW; accB1 = sB1; accB2 = sB2;
        }
    }
    accH = group_sum(accH, cd.E, red[0]);
    accW = group_sum(accW, cd.E, red[1]);
    if constexpr (BN) {
        accB1 = group_sum(accB1, cd.E, red[ND - 2]);
        accB2 = group_sum(accB2, cd.E, red[ND - 1]);
    }
    if (id.valid && e == 0) {
        const int P = cd.ngroups * cd.nchunks;
        CT* o = part + (size_t)id.c * ND * P + (size_t)id.g * cd.nchunks + id.chunk;
        o[0] = accH;
        o[P] = accW;
        if constexpr (BN) { o[2 * (size_t)P] = accB1; o[3 * (size_t)P] = accB2; }
    }
}

// row-sum + K9 (as k2d_finalize) + bn2's constants from partials 2 and 3: d(beta), d(gamma) and the two means
// rk_bn_bwd_dx_pre_* takes (k12 = [2][C])
__global__ __launch_bounds__(kBlock) void k2d_finalize_bn(const float* __restrict__ part, float* __restrict__ gshift,
                                                          float* __restrict__ k12, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int C, int P, int normalize,
                                                          float inv_count) {
    __shared__ double red[4][kBlock / kWave];
    const int c = blockIdx.x;
    const float* p = part + (size_t)c * 4 * P;
    double s[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k)
        for (int i = threadIdx.x; i < P; i += blockDim.x) s[k] += (double)p[(size_t)k * P + i];
    for (int k = 0; k < 4; ++k) s[k] = group_sum(s[k], (int)blockDim.x, red[k]);
    if (threadIdx.x == 0) {
        float gH = (float)s[0], gW = (float)s[1];
        if (normalize) {
            const float mag = sqrtf(gH * gH + gW * gW);
            if (mag > 0) { gH = gH / mag; gW = gW / mag; }
        }
        gshift[c] = gH;
        gshift[C + c] = gW;
        dbeta[c] = (float)s[2];
        dgamma[c] = (float)s[3];
        k12[c] = (float)(s[2] * (double)inv_count);
        k12[C + c] = (float)(s[3] * (double)inv_count);
    }
}

// ----------------------------------------------------------------------------------- host side
inline bool supported(int quantize) {
    return column_kernels_on() && !quantize;
}

// plane_elems: the plane the threads index (output plane for forward, input plane for backward)
inline C2Dims make_c2dims(const Dims2& d, int plane_elems) {
    C2Dims cd;
    cd.d = d;
    if (plane_elems <= kWave) { cd.E = kWave; cd.M = 1; }
    else if (plane_elems <= kBlock) { cd.E = kBlock; cd.M = 1; }
    else { cd.E = kBlock; cd.M = 4; }
    cd.logE = (cd.E == 64) ? 6 : (cd.E == 128 ? 7 : 8);
    cd.nchunks = (plane_elems + cd.E * cd.M - 1) / (cd.E * cd.M);
    // frames per group: enough groups to fill the chip (>= ~4096 thread groups), at least 4 frames each
    int fg = 16;
    while (fg > 4 && (long long)((d.N + fg - 1) / fg) * d.C * cd.nchunks * cd.E < 4096LL * kBlock) fg /= 2;
    cd.FG = fg < d.N ? fg : d.N;
    cd.ngroups = (d.N + cd.FG - 1) / cd.FG;
    return cd;
}
inline unsigned grid_of(const C2Dims& cd) {
    const long long groups = (long long)cd.ngroups * cd.d.C * cd.nchunks;
    const int per_block = kBlock / cd.E;
    return (unsigned)((groups + per_block - 1) / per_block);
}

template <typename T, typename S>
inline void launch_forward(const T* x, const S* shift, T* y, const Dims2& d, hipStream_t stream) {
    const C2Dims cd = make_c2dims(d, d.Ho * d.Wo);
    const bool vec = cd.M == 4 && (d.Ho * d.Wo) % 4 == 0 && ((uintptr_t)y & 15) == 0;
    if (cd.M == 1)
        hipLaunchKernelGGL((k2d_forward_column<T, S, 1>), dim3(grid_of(cd)), dim3(kBlock), 0, stream, x, shift, y, cd);
    else if (vec)
        hipLaunchKernelGGL((k2d_forward_column<T, S, 4, true>), dim3(grid_of(cd)), dim3(kBlock), 0, stream, x, shift, y, cd);
    else
        hipLaunchKernelGGL((k2d_forward_column<T, S, 4>), dim3(grid_of(cd)), dim3(kBlock), 0, stream, x, shift, y, cd);
}

template <typename T>
inline void launch_forward_bn(const T* z, const float* ab, const float* shift, T* y, const Dims2& d, hipStream_t stream) {
    const C2Dims cd = make_c2dims(d, d.Ho * d.Wo);
    const bool vec = cd.M == 4 && (d.Ho * d.Wo) % 4 == 0 && ((uintptr_t)y & 15) == 0;
    if (cd.M == 1)
        hipLaunchKernelGGL((k2d_forward_column<T, float, 1, false, true>), dim3(grid_of(cd)), dim3(kBlock), 0, stream, z, shift,
                           y, cd, ab);
    else if (vec)
        hipLaunchKernelGGL((k2d_forward_column<T, float, 4, true, true>), dim3(grid_of(cd)), dim3(kBlock), 0, stream, z, shift,
                           y, cd, ab);
    else
        hipLaunchKernelGGL((k2d_forward_column<T, float, 4, false, true>), dim3(grid_of(cd)), dim3(kBlock), 0, stream, z, shift,
                           y, cd, ab);
}

inline int backward_partials(const Dims2& d) {
    const C2Dims cd = make_c2dims(d, d.H * d.W);
    return cd.ngroups * cd.nchunks;
}

// d(x) + d(shift) partials into ws[C][2][P]; returns P
template <typename T, typename S>
inline int launch_backward(const T* gy, const T* x, const S* shift, T* gx, typename Compute<T>::type* ws,
                           const Dims2& d, hipStream_t stream) {
    const C2Dims cd = make_c2dims(d, d.H * d.W);
    const bool single = d.sH >= 2 && d.sW >= 2;
    const bool vec = cd.M == 4 && (d.H * d.W) % 4 == 0 && (((uintptr_t)x | (uintptr_t)gx) & 15) == 0;
#define RK_C2_BWD(MM, SG, VC) hipLaunchKernelGGL((k2d_backward_column<T, S, MM, SG, VC>), dim3(grid_of(cd)), dim3(kBlock), 0, \
                                                 stream, gy, x, shift, gx, ws, cd)
    if (cd.M == 1) { if (single) RK_C2_BWD(1, true, false); else RK_C2_BWD(1, false, false); }
    else if (vec) { if (single) RK_C2_BWD(4, true, true); else RK_C2_BWD(4, false, true); }
    else { if (single) RK_C2_BWD(4, true, false); else RK_C2_BWD(4, false, false); }
#undef RK_C2_BWD
    return cd.ngroups * cd.nchunks;
}

// the training fusion: d(bn2's output) + d(shift) + bn2's sums into ws[C][4][P]; returns P
template <typename T>
inline int launch_backward_bn(const T* gy, const T* z, const float* shift, T* dz, float* ws, const float4* abmi,
                              const Dims2& d, hipStream_t stream) {
    const C2Dims cd = make_c2dims(d, d.H * d.W);
    const bool single = d.sH >= 2 && d.sW >= 2;
    const bool vec = cd.M == 4 && (d.H * d.W) % 4 == 0 && (((uintptr_t)z | (uintptr_t)dz) & 15) == 0;
#define RK_C2_BWD(MM, SG, VC) hipLaunchKernelGGL((k2d_backward_column<T, float, MM, SG, VC, true>), dim3(grid_of(cd)), \
                                                 dim3(kBlock), 0, stream, gy, z, shift, dz, ws, cd, abmi)
    if (cd.M == 1) { if (single) RK_C2_BWD(1, true, false); else RK_C2_BWD(1, false, false); }
    else if (vec) { if (single) RK_C2_BWD(4, true, true); else RK_C2_BWD(4, false, true); }
    else { if (single) RK_C2_BWD(4, true, false); else RK_C2_BWD(4, false, false); }
#undef RK_C2_BWD
    return cd.ngroups * cd.nchunks;
}

}  // namespace col2d
}  // namespace rk
