// rk_pw3.hip -- the fp32 1x1 convolution GEMM for the MFMA-bound layers of RubiksNet-Large (unfused half;
// rubiksnet/backbone.py:44-45, :123-135): [F, 288 -> 288, 14, 14] is 70 of its 102 convolutions, 8.3 GFLOP on 116 MB each.
//
// What the first two generations lose there: rk_pw.hip's 64 x 128 wave tiles quantise [288 x 50 176] into
// 1.5 workgroups per resident slot, rk_pw2.hip's barrier-free waves re-read the small operand from L2 once per 64-column
// tile (260 MB in 64-byte pieces) and saturate the L2 -> L1 path before the matrix pipe.  This kernel is the classic
// LDS-tiled GEMM sized so that NEITHER happens:
//
//   * one workgroup per CU and per launch round: the columns are cut into as many equal ranges as there are CUs (196
//     columns = one 14 x 14 frame at 256 frames), a range is ONE workgroup of 12 waves = 6 row groups (3 blocks of 16 rows)
//     x 2 column halves (up to 7 blocks of 16 columns): 21 accumulator blocks = 84 registers per wave, 3 waves per SIMD,
//     every SIMD with the same number of MFMAs;
//   * both operands stream global -> LDS by LDS-DMA (global_load_lds_dwordx4, counted vmcnt, a ring of 4 stages of 16
//     reduction indices): the small operand once per workgroup (85 MB per launch instead of 260), X once in all (58 MB);
//     fragments are ds_read_b128 (A, [M][K] layout: 4 k per lane) / ds_read_b32;
//   * one barrier per stage = per 84 MFMAs of a wave (2 688 matrix-pipe cycles).
//
// k of a stage: lane (j, kq), step s  <->  k0 + 4 kq + s (as rk_pw2.hip).  Arithmetic: exact f32 products, f32 accumulation
// in k order inside a wave (v_mfma_f32_16x16x4_f32).
#include <type_traits>
#include "rk_common.hpp"
#include "rk_dma.hpp"
#include "rk_pw2.hpp"
#include "rk_pw3.hpp"

namespace rk {
namespace pw3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
using pw2::GFuse;
using pw2::GTrain;

struct Dims {
    int F, K, M, P;
    long long ntot;
    int Cw;                  // columns per workgroup (multiple of 4)
    int G;                   // 16-byte groups per X row in LDS (odd, >= Cw / 4: rows 4 apart fall into different bank halves)
    int split;               // columns of the first column half (multiple of 16)
    int nstages;             // K / 16
    int a_is_mk;
};

__device__ __forceinline__ void dma16v(const void* p, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(p), "s"(lds_dst_uniform)
        : "memory");
}
// Non-temporal Y stores measured WORSE here (RubiksNet-Large train step 53.4 -> 57.8 ms): the 58 MB results of the 14 x 14
// layers are re-read by the next kernel out of L2 / Infinity Cache.  (The streaming kernel of the 56 x 56 layers, rk_pw4.hip,
// and the BatchNorm d(x) sweep gain from them.)
#ifndef RK_PW3_NT
#define RK_PW3_NT 0
#endif
__device__ __forceinline__ void st_y1(float* p, float v) {
#if RK_PW3_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_y4(float* p, float a, float b, float c, float d) {
#if RK_PW3_NT
    f32x4 t = {a, b, c, d};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#endif
}
__device__ __forceinline__ float row16_sum_to_lane15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}

constexpr int kNS = 4;                   // stages of the LDS ring
constexpr int kKC = 16;                  // reduction indices per stage

// RB row blocks per wave, NRG row groups, NCS = 2 column halves, CB column blocks per wave (of 16 columns)
template <int RB, int NRG, int CB, bool A_MK, bool PRO, int EPI>
__global__ __launch_bounds__(64 * NRG * 2) void k_pw3_gemm(const float* __restrict__ A, const float* __restrict__ X,
                                                           const float* R, float* Y, Dims d, GFuse fz, GTrain tr) {
    constexpr int NW = NRG * 2;
    constexpr int MR = 16 * RB * NRG;                       // rows of the A stage
    constexpr int ABYTES = MR * kKC * 4;                    // [MR][16] (A_MK) or [16][MR]
    constexpr int NPA = ABYTES / 1024;                      // A pieces per stage
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    const int P = d.P, K = d.K, M = d.M;
    const int xrow = d.G * 16;                              // bytes per X row in LDS
    const int XBYTES = kKC * xrow;
    const int NPX = (XBYTES + 1023) / 1024;
    const int stage_bytes = ABYTES + NPX * 1024;
    float* Ks = reinterpret_cast<float*>(lds + kNS * stage_bytes);     // PRO: ka[K], kb[K]
    if constexpr (PRO) {
        for (int e = threadIdx.x; e < K; e += blockDim.x) { Ks[e] = fz.ka[e]; Ks[K + e] = fz.kb[e]; }
    }
    const long long c0 = (long long)blockIdx.x * d.Cw;      // first column of the workgroup
    long long cend = c0 + d.Cw;
    cend = cend < d.ntot ? cend : d.ntot;
    const int ncols = (int)(cend - c0);                     // valid columns (multiple of 4)

    // ---- DMA role: pieces pi = wave + NW t; pi < NPA: A, else X ----
    const int npieces = NPA + NPX;
    constexpr int NPW = 4;                                  // pieces per wave and stage, at most (NPA + NPX <= 4 NW)
    const float* src[NPW];
    int step_f[NPW];                                        // floats to advance per stage
    unsigned dst[NPW];
    int npw = 0;
#pragma unroll
    for (int t = 0; t < NPW; ++t) {
        const int pi = wave + NW * t;
        src[t] = X; step_f[t] = 0; dst[t] = 0;
        if (pi < npieces) {
            ++npw;
            if (pi < NPA) {
                const int q = pi * 64 + lane;               // 16-byte unit of the A stage
                if (A_MK) {                                  // [MR rows][4 units of 4 k]
                    int row = q >> 2;
                    row = row < M ? row : M - 1;             // rows past M: a copy, never stored
                    src[t] = A + (size_t)row * K + 4 * (q & 3);
                    step_f[t] = kKC;
                } else {                                     // [16 k][MR / 4 units of 4 rows]
                    const int k = q / (MR / 4);
                    int m4 = 4 * (q - k * (MR / 4));
                    m4 = m4 + 4 <= M ? m4 : (M - 4);         // (M % 4 == 0)
                    src[t] = A + (size_t)k * M + m4;
                    step_f[t] = kKC * M;
                }
                dst[t] = (unsigned)(pi * 1024);
            } else {
                const int q = (pi - NPA) * 64 + lane;       // 16-byte unit of the X stage: [16 k][G groups]
                int k = q / d.G;
                const int g = q - k * d.G;
                k = k < kKC ? k : kKC - 1;                   // (units past the stage land in the padding of the last piece)
                long long c = c0 + 4 * g;
                c = c + 4 <= cend ? c : cend - 4;            // groups past the range: a copy
                const int f = (int)(c / P), p = (int)(c - (long long)f * P);
                src[t] = X + ((size_t)f * K + k) * P + p;
                step_f[t] = kKC * P;
                dst[t] = (unsigned)(ABYTES + (pi - NPA) * 1024);
            }
        }
    }
    const unsigned lds0 = dma::lds_byte_addr(lds);
    int issued = 0;
    auto issue = [&]() {                                    // the next stage, into slot issued % kNS
        const unsigned base = lds0 + (unsigned)((issued % kNS) * stage_bytes);
#pragma unroll
        for (int t = 0; t < NPW; ++t) {
            if (wave + NW * t < npieces) {
                dma16v(src[t], base + dst[t]);
                src[t] += step_f[t];
            }
        }
        ++issued;
    };

    // ---- compute role ----
    const int rg = wave % NRG, cs = wave / NRG;
    const int col_lo = cs == 0 ? 0 : d.split;               // first column of this wave inside the workgroup
    int wcols = (cs == 0 ? d.split : ncols - d.split);
    wcols = wcols < 0 ? 0 : (wcols > ncols ? ncols : wcols);          // valid columns of this wave
    const int mrow0 = 16 * RB * rg;
    f32x4 acc[RB][CB];
    // ---- output geometry: acc[rb][cb][r] = row mrow0 + 16 rb + 4 kq + r, column col_lo + (4 j + cb | 64 + 3 j + cb - 4) ----
    size_t yo[CB];                                          // offset of (frame, pixel) of this lane's column of block cb
    bool con[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int rel = cb < 4 ? 4 * j + cb : 64 + (CB - 4) * j + (cb - 4);
        con[cb] = rel < wcols;
        const long long c = c0 + col_lo + (con[cb] ? rel : 0);
        const long long cc = c < d.ntot ? c : 0;
        const int f = (int)(cc / P), p = (int)(cc - (long long)f * P);
        yo[cb] = ((size_t)f * M) * P + p;
    }
    // the residual is the initial accumulator value: its loads land under the prologue's DMA wait instead of in twelve
    // load -> use chains after the last MFMA (measured: + 22 us on a 84 us kernel)
    if (R) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + 16 * rb + 4 * kq + r;
                const size_t mo = (size_t)(m < M ? m : 0) * P;
                const float4 t = *reinterpret_cast<const float4*>(R + yo[0] + mo);
                acc[rb][0][r] = t.x; acc[rb][1][r] = t.y; acc[rb][2][r] = t.z; acc[rb][3][r] = t.w;
#pragma unroll
                for (int cb = 4; cb < CB; ++cb) acc[rb][cb][r] = R[yo[cb] + mo];
            }
    } else {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // LDS read offsets inside a stage
    const int a_off = A_MK ? ((mrow0 + j) * 64 + kq * 16) : ((4 * kq) * (MR * 4) + (mrow0 + j) * 4);
    // column of (block cb, lane j): cb < 4: 4 j + cb (the first 64 columns, 4 interleaved blocks: one ds_read_b128 per k, and
    // the results of a row are 4 consecutive columns: 16-byte stores); cb >= 4: 64 + (CB - 4) j + (cb - 4)
    const int x_off = ABYTES + (4 * kq) * xrow + (col_lo + 4 * j) * 4;
    const int x_off2 = ABYTES + (4 * kq) * xrow + (col_lo + 64 + (CB - 4) * j) * 4;

    const int ns = d.nstages;
#pragma unroll
    for (int s = 0; s < kNS - 1; ++s)
        if (s < ns) issue();
    if constexpr (PRO) __syncthreads();

#pragma nounroll
    for (int st = 0; st < ns; ++st) {
        const int left = ns - 1 - st;
        dma::wait_vmcnt(npw * (left < kNS - 2 ? left : kNS - 2));
        __syncthreads();                                    // stage st is in; stage st - 1 is consumed by every wave
        if (st + kNS - 1 < ns) issue();
        const char* sb = lds + (st % kNS) * stage_bytes;
        float a[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (A_MK) {
                const float4 t = *reinterpret_cast<const float4*>(sb + a_off + rb * (16 * 64));
                a[rb][0] = t.x; a[rb][1] = t.y; a[rb][2] = t.z; a[rb][3] = t.w;
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) a[rb][s] = *reinterpret_cast<const float*>(sb + a_off + s * (MR * 4) + rb * 64);
            }
        }
        float4 pa, pb;
        if constexpr (PRO) {
            pa = *reinterpret_cast<const float4*>(Ks + st * kKC + 4 * kq);
            pb = *reinterpret_cast<const float4*>(Ks + K + st * kKC + 4 * kq);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float b[CB];
            {
                const float4 t = *reinterpret_cast<const float4*>(sb + x_off + s * xrow);
                b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
#pragma unroll
                for (int cb = 4; cb < CB; ++cb) b[cb] = *reinterpret_cast<const float*>(sb + x_off2 + s * xrow + (cb - 4) * 4);
            }
            if constexpr (PRO) {
                const float ka = s == 0 ? pa.x : s == 1 ? pa.y : s == 2 ? pa.z : pa.w;
                const float kb = s == 0 ? pb.x : s == 1 ? pb.y : s == 2 ? pb.z : pb.w;
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const float t = fmaf(ka, b[cb], kb);
                    b[cb] = fz.relu_in ? fmaxf(t, 0.f) : t;
                }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][s], b[cb], acc[rb][cb], 0, 0, 0);
        }
    }

    const long long tile = (long long)blockIdx.x * 2 + cs;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        float xv[4][CB];
        float4 pk[4];
        if constexpr (EPI == 2) {                           // the 4 rows of a block requested together
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + 16 * rb + 4 * kq + r;
                const size_t mo = (size_t)(m < M ? m : 0) * P;
                const float4 t = *reinterpret_cast<const float4*>(tr.bx + yo[0] + mo);
                xv[r][0] = t.x; xv[r][1] = t.y; xv[r][2] = t.z; xv[r][3] = t.w;
#pragma unroll
                for (int cb = 4; cb < CB; ++cb) xv[r][cb] = tr.bx[yo[cb] + mo];
                pk[r] = tr.bpack[m < M ? m : 0];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mrow0 + 16 * rb + 4 * kq + r;
            const bool mok = m < M;
            const size_t mo = (size_t)(mok ? m : 0) * P;
            float o[CB];
            float s1 = 0.f, s2 = 0.f, piv = 0.f;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) o[cb] = acc[rb][cb][r];
            if constexpr (EPI == 1) piv = __shfl(o[0], lane & 48);            // the row's first column of this split
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const bool on = con[cb] && mok;
                if constexpr (EPI == 1) {
                    if (on) { const float t = o[cb] - piv; s1 += t; s2 = fmaf(t, t, s2); }
                } else if constexpr (EPI == 2) {
                    if (on) {
                        o[cb] = fmaf(pk[r].x, xv[r][cb], pk[r].y) <= 0.f ? 0.f : o[cb];
                        s1 += o[cb];
                        s2 = fmaf(o[cb], (xv[r][cb] - pk[r].z) * pk[r].w, s2);
                    }
                }
                if (on && cb >= 4) st_y1(Y + yo[cb] + mo, o[cb]);
            }
            if (con[0] && mok) st_y4(Y + yo[0] + mo, o[0], o[1], o[2], o[3]);
            if constexpr (EPI != 0) {
                s1 = row16_sum_to_lane15(s1);
                s2 = row16_sum_to_lane15(s2);
                if (j == 15 && mok) {
                    if constexpr (EPI == 1) tr.stats[(size_t)m * tr.J + tile] = make_float4(piv, s1, s2, (float)wcols);
                    else tr.bred[(size_t)m * tr.J + tile] = make_float2(s1, s2);
                }
            }
        }
    }
}

// ---- host ----
struct Plan { int Cw, G, split, nwg; };

// The kernel instance: 288 rows (17 or 18 blocks), 12 waves.  Column ranges of 193 .. 224 columns, i.e. 7 blocks for the
// first half and up to 7 for the second: anything else would leave the 7-block wave tiles partly empty.
bool plan(Plan& pl, int F, int K, int M, int P) {
    static const int mode = [] { const char* e = getenv("RK_PW3"); return e ? atoi(e) : 1; }();
    if (!mode || P % 4 || K % 16 || M % 4 || M <= 256 || M > 288 || K > 1024) return false;
    const long long ntot = (long long)F * P;
    const int cus = device_cus();                    // cached per device, thread-safe (rk_common.hpp)
    const long long rounds = (ntot + (long long)cus * 224 - 1) / ((long long)cus * 224);
    long long cw = (ntot + cus * rounds - 1) / (cus * rounds);
    cw = (cw + 3) / 4 * 4;
    if (cw <= 192 || cw > 224) return false;
    pl.Cw = (int)cw;
    pl.split = 112;
    pl.G = (pl.Cw / 4) | 1;
    pl.nwg = (int)((ntot + cw - 1) / cw);
    return true;
}

int tiles(int F, int K, int M, int P) {
    Plan pl;
    return plan(pl, F, K, M, P) ? 2 * pl.nwg : 0;
}

template <bool A_MK, bool PRO, int EPI>
int launch(const float* A, const float* X, const float* R, float* Y, const Dims& d, const Plan& pl, const GFuse& fz,
           const GTrain& tr, hipStream_t stream) {
    constexpr int RB = 3, NRG = 6, CB = 7;
    const int abytes = 16 * RB * NRG * kKC * 4;
    const int npx = (kKC * pl.G * 16 + 1023) / 1024;
    const size_t lds = (size_t)kNS * (abytes + npx * 1024) + (PRO ? 2 * d.K * sizeof(float) : 0);
    static DynLdsRaised raised;                      // per instance and device (rk_common.hpp)
    if (const int rc = raise_dynamic_lds(reinterpret_cast<const void*>(&k_pw3_gemm<RB, NRG, CB, A_MK, PRO, EPI>), lds, raised))
        return rc;
    hipLaunchKernelGGL((k_pw3_gemm<RB, NRG, CB, A_MK, PRO, EPI>), dim3((unsigned)pl.nwg), dim3(64 * NRG * 2), lds, stream, A, X, R,
                       Y, d, fz, tr);
    return launch_status();
}

int gemm(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk, const GFuse* fuse,
         const GTrain* train, int epi, hipStream_t stream) {
    if (!A || !X || !Y) return RK_ERR_NULL_POINTER;
    Plan pl;
    if (!plan(pl, F, K, M, P)) return RK_ERR_UNSUPPORTED;
    if (((uintptr_t)A & 15) || ((uintptr_t)X & 15)) return RK_ERR_UNSUPPORTED;
    GFuse fz = fuse ? *fuse : GFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
    GTrain tr = train ? *train : GTrain{nullptr, nullptr, nullptr, nullptr, 0};
    if (fz.ma) return RK_ERR_UNSUPPORTED;                   // (inference epilogue: the other generations)
    const bool pro = fz.ka != nullptr;
    if (epi == 1 && !tr.stats) return RK_ERR_NULL_POINTER;
    if (epi == 2 && !(tr.bred && tr.bx && tr.bpack)) return RK_ERR_NULL_POINTER;
    if (epi && tr.J != 2 * pl.nwg) return RK_ERR_BAD_DIMS;
    Dims d;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P; d.Cw = pl.Cw; d.G = pl.G; d.split = pl.split;
    d.nstages = K / kKC; d.a_is_mk = a_is_mk;
    if (a_is_mk) {
        if (epi == 1) return pro ? launch<true, true, 1>(A, X, R, Y, d, pl, fz, tr, stream) : launch<true, false, 1>(A, X, R, Y, d, pl, fz, tr, stream);
        if (epi == 2) return RK_ERR_UNSUPPORTED;
        return pro ? launch<true, true, 0>(A, X, R, Y, d, pl, fz, tr, stream) : launch<true, false, 0>(A, X, R, Y, d, pl, fz, tr, stream);
    }
    if (pro || epi == 1) return RK_ERR_UNSUPPORTED;
    if (epi == 2) return launch<false, false, 2>(A, X, R, Y, d, pl, fz, tr, stream);
    return launch<false, false, 0>(A, X, R, Y, d, pl, fz, tr, stream);
}

}  // namespace pw3
}  // namespace rk
