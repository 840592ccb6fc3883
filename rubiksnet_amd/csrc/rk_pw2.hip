// rk_pw2.hip -- second-generation fp32 kernels of the 1x1 convolutions (unfused half;
// rubiksnet/backbone.py:44-45, :123-135): forward / d(input) GEMM and d(weight) on v_mfma_f32_16x16x4_f32.
//
// Why a second generation.  f32-input MFMA runs at the f32 VECTOR rate (64 FLOP/clk/SIMD: one 16x16x4 instruction holds a
// SIMD's matrix pipe for 32 cycles), so on the deep layers ([256, 288 -> 288, 14 x 14]: 8.3 GFLOP against 116 MB) the only
// thing that matters is that every SIMD's pipe always has an instruction to issue; operand bandwidth is an order of magnitude
// below what LDS / L1 deliver.  The first-generation kernels (rk_pw.hip, 32x32x2) lose the pipe to structure, not bandwidth:
// 64 x 128 wave tiles quantise [288 x 50 176] into 1.7 waves per SIMD (86 % at best), 32-row blocks pad 72 / 144 channels by
// 33 / 11 %, and a workgroup barrier per 16 channels couples the four waves of a tile.  Here:
//
//   * GEMM (k_pw2_gemm): NO LDS, NO barrier.  A wave is an independent unit = RB row blocks of 16 x one 64-pixel column tile
//     (4 interleaved 16-column blocks: lane (j, kq) owns pixels 4 j .. 4 j + 3, so the streamed operand is read with one
//     16-byte load per lane and k-step straight into B fragments and the results leave as 16-byte stores).  The small
//     operand's fragments are 16-byte loads along K straight from global memory (L1 / L2 resident; [M][K] layout), or
//     dword loads ([K][M] layout: 64 contiguous bytes per k), or -- when K % 4 != 0 (54 channels) -- reads of an LDS image
//     made once per workgroup.  48-row units ([288 rows] = 6 units per column tile = one 6-wave workgroup, so the column
//     tile's X lines are shared through L1) give 4 704 units for 1 024 SIMDs: 4-5 resident waves per SIMD, 92 % balance.
//   * d(weight) (k_pw2_wgrad): both operands have the reduction index (pixels) contiguous, so a 16-byte piece of a row is 4
//     k-steps of a fragment lane.  Stages of 32 pixels x (tile rows of dY + tile rows of X) go global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4: 8 rows x 128 B per wave-instruction; the 16-byte slot of a row is XOR-swizzled through the
//     lane's GLOBAL address so that the ds_read_b128 fragment reads are bank-conflict free), NS stages deep, one barrier per
//     stage, counted vmcnt.  Wave tiles of up to RA x RX blocks of 16 x 16; optional second pixel phase (waves that share a
//     wave tile and split a stage's pixels, summed through LDS at the end).  Partials per split -> k_pw2_reduce.
//
// Arithmetic: exact f32 products and f32 accumulation in k order inside a wave (v_mfma_f32_16x16x4_f32 == an fmaf chain),
// i.e. the arithmetic class of rk_pw.hip; only the summation ORDER differs (tests: <= 2e-6 sqrt(K) against fp64).
#include <type_traits>
#include "rk_common.hpp"
#include "rk_dma.hpp"
#include "rk_pw2.hpp"
#include "rk_reduce.hpp"

namespace rk {
namespace pw2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// forward / d(input):  Y[f] = A X[f] (+ R[f])
// ---------------------------------------------------------------------------------------------------------------------
struct GDims {
    int F, K, M, P;
    long long ntot;          // F * P columns
    int nrg;                 // row groups of RB blocks (ceil(ceil(M / 16) / RB))
    int ct;                  // 64-pixel column tiles per workgroup; waves = nrg * ct
    int ngroups;             // ceil(K / 16)
    int a_is_mk;             // A given as [M][K] (else [K][M])
    int mpad;                // AMODE 2: row stride (floats) of the LDS image As[k][mpad]
};
// sum over the 16 lanes of a DPP row (lanes that share lane >> 4); the total lands in the row's lane 15.  Fixed order.
__device__ __forceinline__ float row16_sum_to_lane15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
    return v;
}

// (non-temporal Y stores: level in the train steps -- Large 53.4 / 53.1, Tiny 19.4 / 19.6 ms -- left off)
#ifndef RK_PW2_NT
#define RK_PW2_NT 0
#endif
__device__ __forceinline__ void st_y4(float* p, const float4& o) {
#if RK_PW2_NT
    f32x4 t = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = o;
#endif
}
// AMODE 0: A [M][K], K % 4 == 0, 16-byte aligned: fragments by float4 loads along K
// AMODE 1: A [K][M]: fragments by dword loads (64 contiguous bytes per k)
// AMODE 2: any layout / any K: an LDS image As[k][mpad] made once per workgroup
// k of a group of 16: lane (j, kq), step s  <->  k0 + 4 kq + s  (same map for A and X)
template <int RB, int AMODE, bool PRO, int EPI, bool OUTAFF>
__global__ __launch_bounds__(256, RB == 3 ? 5 : (RB == 4 ? 4 : 3)) void k_pw2_gemm(const float* __restrict__ A, const float* __restrict__ X, const float* R, float* Y, GDims d,
                           GFuse fz, GTrain tr) {
    // LDS: [PRO: (ka, kb) of every k, zero padded to whole groups: 2 x 16 ngroups floats][AMODE 2: the image of A]
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    float* Ks = lds_f;
    float* As = lds_f + (PRO ? 32 * d.ngroups : 0);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    if constexpr (PRO) {
        for (int e = threadIdx.x; e < 16 * d.ngroups; e += blockDim.x) {
            Ks[e] = e < d.K ? fz.ka[e] : 0.f;
            Ks[16 * d.ngroups + e] = e < d.K ? fz.kb[e] : 0.f;
        }
    }
    if constexpr (AMODE == 2) {
        const int rows = 16 * d.ngroups, total = rows * d.mpad;
        for (int e = threadIdx.x; e < total; e += blockDim.x) {
            const int k = e / d.mpad, m = e - k * d.mpad;
            const bool ok = k < d.K && m < d.M;
            As[e] = ok ? A[d.a_is_mk ? (size_t)m * d.K + k : (size_t)k * d.M + m] : 0.f;
        }
    }
    if constexpr (PRO || AMODE == 2) __syncthreads();
    long long tile;
    int rg;
    if (d.ct == 2) {                                        // 2 row groups x 2 column tiles per workgroup (nrg even)
        const int nrp = d.nrg >> 1;
        const long long b8 = blockIdx.x >> 3;               // workgroups of one tile pair: 8 ids apart (same XCD)
        const long long tp = (b8 / nrp) * 8 + (blockIdx.x & 7);
        const int rgp = (int)(b8 % nrp);
        tile = 2 * tp + (wave >> 1);
        rg = 2 * rgp + (wave & 1);
    } else {
        const long long unit = (long long)blockIdx.x * 4 + wave;      // units of one column tile are consecutive
        tile = unit / d.nrg;
        rg = (int)(unit - tile * d.nrg);
    }
    if (tile * 64 >= d.ntot) return;
    const long long c0 = tile * 64 + 4 * j;                 // this lane's 4 columns
    const bool valid = c0 < d.ntot;
    const long long cc = valid ? c0 : 0;
    const int f = (int)(cc / d.P), p = (int)(cc - (long long)f * d.P);
    const int P = d.P, K = d.K, M = d.M;
    const float* xf = X + ((size_t)f * K) * P + p;                    // row k: + k P
    const size_t yoff = ((size_t)f * M) * P + p;                      // row m: + m P
    const int mrow0 = 16 * rg * RB;                                   // first row of this unit
    int nv = (M + 15) / 16 - rg * RB;                                 // row blocks of this unit that hold rows (uniform)
    nv = nv > RB ? RB : nv;

    f32x4 acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[rb][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A fragment sources
    const float* ap[RB];
    bool mok[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int m = mrow0 + 16 * rb + j;
        mok[rb] = m < M;
        const int mc = mok[rb] ? m : 0;
        if (AMODE == 0) ap[rb] = A + (size_t)mc * K;
        else if (AMODE == 1) ap[rb] = A + mc;
        else ap[rb] = As + (mrow0 + 16 * rb + j);
    }

    const int ng = d.ngroups;
    struct Frag {
        float4 x[4];
        float a[RB][4];
        float4 pa, pb;
        int kb;                                             // first k of this lane's 4 (the selects on A happen at use:
    };                                                      //  a select next to the load would wait for the load)
    // Branch-free: every load is issued every time, from a clamped (valid) address -- a predicated load is a branch, and
    // at the merge hipcc's s_waitcnt pass falls back to vmcnt(0), which drains the prefetch in front of every MFMA group.
    // X rows past K re-read row K - 1 (a legitimate row of the same pixel; its A column is zero), rows of A past M re-read
    // row 0 (results never stored); only A's k-columns past K are zeroed by a select.
    auto load = [&](int g, Frag& fr) {
        const int k0 = 16 * g, kb = k0 + 4 * kq;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kr = kb + s < K ? kb + s : K - 1;
            fr.x[s] = *reinterpret_cast<const float4*>(xf + (size_t)kr * P);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (AMODE == 0) {
                const int kc = kb < K ? kb : K - 4;                  // K % 4 == 0: a float4 is inside or outside
                const float4 t = *reinterpret_cast<const float4*>(ap[rb] + kc);
                fr.a[rb][0] = t.x; fr.a[rb][1] = t.y; fr.a[rb][2] = t.z; fr.a[rb][3] = t.w;
            } else if (AMODE == 1) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kr = kb + s < K ? kb + s : K - 1;
                    fr.a[rb][s] = ap[rb][(size_t)kr * M];
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) fr.a[rb][s] = ap[rb][(kb + s) * d.mpad];     // (zero padded image)
            }
        }
        fr.kb = kb;
        if constexpr (PRO) {
            fr.pa = *reinterpret_cast<const float4*>(Ks + kb);
            fr.pb = *reinterpret_cast<const float4*>(Ks + 16 * ng + kb);
        }
    };
    auto compute = [&](const Frag& fr, auto NVc) {
        constexpr int NV = decltype(NVc)::value;            // row blocks that take part
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float b[4] = {fr.x[s].x, fr.x[s].y, fr.x[s].z, fr.x[s].w};
            if constexpr (PRO) {
                const float pa = s == 0 ? fr.pa.x : s == 1 ? fr.pa.y : s == 2 ? fr.pa.z : fr.pa.w;
                const float pb = s == 0 ? fr.pb.x : s == 1 ? fr.pb.y : s == 2 ? fr.pb.z : fr.pb.w;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = fmaf(pa, b[q], pb);
                    b[q] = fz.relu_in ? fmaxf(t, 0.f) : t;
                }
            }
#pragma unroll
            for (int rb = 0; rb < NV; ++rb) {
                const float av = (AMODE == 2 || fr.kb + s < K) ? fr.a[rb][s] : 0.f;      // A's k-columns past K are zero
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[rb][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[q], acc[rb][q], 0, 0, 0);
            }
        }
    };
    // two register sets: group g + 1 is requested before the MFMAs of group g issue.  The K loop exists once per block
    // count (chosen here, outside the loop: one clean loop body each)
    auto kloop = [&](auto NVc) {
        // One register set, no software pipelining inside a wave: hipcc's scheduler and s_waitcnt pass undo it (a rotated loop
        // with vmcnt(0) at its head), and 4-5 resident waves per SIMD cover the load latency instead.  The selects on A
        // stay in every group: the loop without them measured SLOWER (111 vs 87 us at [256,288->288,14x14]; its three waits
        // then sit in front of the first MFMA instead of between the row blocks).
        Frag f0;
#pragma nounroll
        for (int g = 0; g < ng; ++g) {
            load(g, f0);
            __builtin_amdgcn_sched_barrier(0);              // (the scheduler otherwise sinks loads below the MFMAs)
            compute(f0, NVc);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (nv == RB) kloop(std::integral_constant<int, RB>{});
    else kloop(std::integral_constant<int, RB - 1>{});      // (fewer still: rows past M are zero rows of A)

    // ---- epilogue: acc[rb][q][r] = row mrow0 + 16 rb + 4 kq + r, pixel p + q ----
    if constexpr (EPI == 0) {
        if (!valid) return;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float4 rv[4];
            if (R) {                                        // the 4 rows of a block requested together (clamped addresses)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + 16 * rb + 4 * kq + r;
                    rv[r] = *reinterpret_cast<const float4*>(R + yoff + (size_t)(m < M ? m : M - 1) * P);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + 16 * rb + 4 * kq + r;
                if (m >= M) continue;
                float4 o = make_float4(acc[rb][0][r], acc[rb][1][r], acc[rb][2][r], acc[rb][3][r]);
                if constexpr (OUTAFF) {
                    if (fz.ma) {
                        const float ea = fz.ma[m], eb = fz.mb[m];
                        o.x = fmaf(ea, o.x, eb); o.y = fmaf(ea, o.y, eb); o.z = fmaf(ea, o.z, eb); o.w = fmaf(ea, o.w, eb);
                        if (fz.relu_out) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    }
                }
                if (R) { o.x += rv[r].x; o.y += rv[r].y; o.z += rv[r].z; o.w += rv[r].w; }
                st_y4(Y + yoff + (size_t)m * P, o);
            }
        }
    } else {
        // training epilogues: every lane walks every row (the DPP sums need whole rows of 16 lanes)
        long long nleft = d.ntot - tile * 64;
        const float ntile = (float)(nleft < 64 ? nleft : 64);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float4 xv[4], pk[4], rv[4];
            if (R) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + 16 * rb + 4 * kq + r;
                    rv[r] = *reinterpret_cast<const float4*>(R + yoff + (size_t)(m < M ? m : M - 1) * P);
                }
            }
            if constexpr (EPI == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + 16 * rb + 4 * kq + r;
                    const int mc = m < M ? m : M - 1;                // (clamped: rows past M are not stored or summed)
                    xv[r] = *reinterpret_cast<const float4*>(tr.bx + yoff + (size_t)mc * P);
                    pk[r] = tr.bpack[mc];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mrow0 + 16 * rb + 4 * kq + r;
                const bool on = valid && m < M;
                float4 o = make_float4(acc[rb][0][r], acc[rb][1][r], acc[rb][2][r], acc[rb][3][r]);
                if (R) { o.x += rv[r].x; o.y += rv[r].y; o.z += rv[r].z; o.w += rv[r].w; }
                float s1 = 0.f, s2 = 0.f, piv = 0.f;
                if constexpr (EPI == 1) {
                    piv = __shfl(o.x, lane & 48);                     // the row's first column in this tile (always valid)
                    if (on) {
                        float t;
                        t = o.x - piv; s1 += t; s2 = fmaf(t, t, s2);
                        t = o.y - piv; s1 += t; s2 = fmaf(t, t, s2);
                        t = o.z - piv; s1 += t; s2 = fmaf(t, t, s2);
                        t = o.w - piv; s1 += t; s2 = fmaf(t, t, s2);
                    }
                } else if (on) {
                    const float pa = pk[r].x, pb = pk[r].y, mu = pk[r].z, iv = pk[r].w;
                    o.x = fmaf(pa, xv[r].x, pb) <= 0.f ? 0.f : o.x;  s1 += o.x;  s2 = fmaf(o.x, (xv[r].x - mu) * iv, s2);
                    o.y = fmaf(pa, xv[r].y, pb) <= 0.f ? 0.f : o.y;  s1 += o.y;  s2 = fmaf(o.y, (xv[r].y - mu) * iv, s2);
                    o.z = fmaf(pa, xv[r].z, pb) <= 0.f ? 0.f : o.z;  s1 += o.z;  s2 = fmaf(o.z, (xv[r].z - mu) * iv, s2);
                    o.w = fmaf(pa, xv[r].w, pb) <= 0.f ? 0.f : o.w;  s1 += o.w;  s2 = fmaf(o.w, (xv[r].w - mu) * iv, s2);
                }
                if (on) st_y4(Y + yoff + (size_t)m * P, o);
                s1 = row16_sum_to_lane15(s1);
                s2 = row16_sum_to_lane15(s2);
                if (j == 15 && m < M) {
                    if constexpr (EPI == 1) tr.stats[(size_t)m * tr.J + tile] = make_float4(piv, s1, s2, ntile);
                    else tr.bred[(size_t)m * tr.J + tile] = make_float2(s1, s2);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// d(weight):  dW[m][k] = sum over pixels n = (f, p) of dY[f][m][p] X[f][k][p]
// ---------------------------------------------------------------------------------------------------------------------
struct WDims {
    int F, K, M, P;
    long long ntot;
    int nbM, nbK;                // 16-row blocks of dY / X
    int tbM, tbK;                // blocks per workgroup tile
    int tilesM, tilesK;
    int S;                       // splits of the pixel range
    long long span;              // pixels per split (a multiple of 32)
    const float* ka; const float* kb; int relu_in;      // PRO: X operand = relu?(ka[k] x + kb[k])
};

// one wave-instruction of LDS-DMA with a full 64-bit address per lane: lane l copies 16 B from p_l to LDS lds_dst + 16 l
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

constexpr int kStagePx = 32;                 // pixels per stage: 128 B per row = 8 slots of 16 B

// the MFMA phase of one round (16 pixels: slot 4 h + kq of every row) for NA x NB blocks
template <int RA, int RX, int NA, int NB, bool PRO, bool GEN>
__device__ __forceinline__ void wg_round(const char* sa, const char* sx, bool masked, const float (&pa)[RX],
                                         const float (&pb)[RX], int relu, int na, int nb, f32x4 (&acc)[RA][RX]) {
    float4 fa[NA > 0 ? NA : 1], fb[NB > 0 ? NB : 1];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        fa[a] = *reinterpret_cast<const float4*>(sa + a * (16 * 128));
        if (masked) fa[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        fb[b] = *reinterpret_cast<const float4*>(sx + b * (16 * 128));
        if constexpr (PRO) {
            fb[b].x = fmaf(pa[b], fb[b].x, pb[b]); fb[b].y = fmaf(pa[b], fb[b].y, pb[b]);
            fb[b].z = fmaf(pa[b], fb[b].z, pb[b]); fb[b].w = fmaf(pa[b], fb[b].w, pb[b]);
            if (relu) { fb[b].x = fmaxf(fb[b].x, 0.f); fb[b].y = fmaxf(fb[b].y, 0.f); fb[b].z = fmaxf(fb[b].z, 0.f); fb[b].w = fmaxf(fb[b].w, 0.f); }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            if (GEN && a >= na) break;
            const float av = c == 0 ? fa[a].x : c == 1 ? fa[a].y : c == 2 ? fa[a].z : fa[a].w;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (GEN && b >= nb) break;
                const float bv = c == 0 ? fb[b].x : c == 1 ? fb[b].y : c == 2 ? fb[b].z : fb[b].w;
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[a][b], 0, 0, 0);
            }
        }
}

// RA x RX: blocks per wave tile; WR x WC wave tiles per workgroup tile; PH pixel phases (1: a wave does both rounds of a
// stage, 2: round ph); NS stages of the LDS ring.  Waves = WR WC PH (even).
template <int RA, int RX, int WR, int WC, int PH, int NS, bool PRO>
__global__ __launch_bounds__(64 * WR * WC * PH, (WR * WC * PH == 12 || RA * RX > 9) ? 3 : 4) void k_pw2_wgrad(const float* __restrict__ dY, const float* __restrict__ X,
                                                                 float* __restrict__ ws, WDims d) {
    constexpr int NW = WR * WC * PH;
    static_assert(NW % 2 == 0 && (PH == 1 || PH == 2), "waves");
    constexpr int TA = 16 * RA * WR, TX = 16 * RX * WC, ROWS = TA + TX;
    constexpr int NPIECE = ROWS / 8;                       // DMA instructions per stage (8 rows x 128 B each)
    constexpr int NPW = (NPIECE + NW - 1) / NW;
    constexpr int STAGE = ROWS * 128;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;

    // workgroup -> (split, tile): the tiles of one split are 8 ids apart (same XCD, dispatched together), so the second
    // reader of an operand row finds it in that XCD's L2
    const int T = d.tilesM * d.tilesK;
    const int blk8 = blockIdx.x / (8 * T), rem = blockIdx.x - blk8 * 8 * T;
    const int tile = rem >> 3, split = blk8 * 8 + (rem & 7);
    if (split >= d.S) return;
    const int tm = tile / d.tilesK, tk = tile - tm * d.tilesK;
    const long long n_lo = (long long)split * d.span;
    long long n_hi = n_lo + d.span;
    n_hi = n_hi < d.ntot ? n_hi : d.ntot;
    const int nst = (int)((n_hi - n_lo + kStagePx - 1) / kStagePx);
    const int P = d.P;

    // ---- DMA role: piece pi = wave + NW t covers stage rows 8 pi .. 8 pi + 7; lane -> row 8 pi + (lane >> 3), LDS slot
    // lane & 7 of that row, which holds pixel group (lane & 7) ^ key(row), key(row) = (row >> 1) & 7 = 4 (pi & 1) + (lane >> 4)
    // [pi has the parity of the wave: NW is even] ----
    const int lr = lane >> 3;
    const int G = (lane & 7) ^ (4 * (wave & 1) + (lr >> 1));
    long long dn = n_lo + 4 * G;                            // this lane's pixel group in the current DMA stage
    int df, dp;
    {
        const long long q = dn < d.ntot ? dn : d.ntot - 4;
        df = (int)(q / P); dp = (int)(q - (long long)df * P);
    }
    int chP[NPW];                                           // channel row offset (floats) of this lane's row in piece t
    bool isY[NPW];
    int npw = 0;                                            // pieces of this wave (uniform)
#pragma unroll
    for (int t = 0; t < NPW; ++t) {
        const int pi = wave + NW * t;
        const int r = 8 * (pi < NPIECE ? pi : NPIECE - 1) + lr;
        isY[t] = r < TA;
        int ch = isY[t] ? 16 * d.tbM * tm + r : 16 * d.tbK * tk + (r - TA);
        const int C = isY[t] ? d.M : d.K;
        ch = ch < C ? ch : C - 1;                           // rows past the operand: a copy, never stored
        chP[t] = ch * P;
        npw += pi < NPIECE ? 1 : 0;
    }
    const unsigned lds0 = dma::lds_byte_addr(lds);
    auto issue = [&](int s) {                               // (called with s = 0, 1, 2, ... in order)
        const unsigned dst = lds0 + (unsigned)((s % NS) * STAGE);
        const size_t fy = (size_t)df * d.M * P + dp, fx = (size_t)df * d.K * P + dp;
#pragma unroll
        for (int t = 0; t < NPW; ++t) {
            const int pi = wave + NW * t;
            if (pi < NPIECE) {
                const float* src = isY[t] ? dY + fy + chP[t] : X + fx + chP[t];
                dma16v(src, dst + (unsigned)(pi * 1024));
            }
        }
        dn += kStagePx; dp += kStagePx;
        while (dp >= P) { dp -= P; ++df; }
        if (dn >= d.ntot) { const long long q = d.ntot - 4; df = (int)(q / P); dp = (int)(q - (long long)df * P); }
    };

    // ---- compute role ----
    const int ph = wave / (WR * WC), wt = wave - ph * (WR * WC);
    const int wr = wt / WC, wc = wt - wr * WC;
    const int i = lane & 15, kq = lane >> 4;
    int tbm = d.nbM - tm * d.tbM; tbm = tbm < d.tbM ? tbm : d.tbM;      // blocks of this tile
    int tbk = d.nbK - tk * d.tbK; tbk = tbk < d.tbK ? tbk : d.tbK;
    int na = tbm - wr * RA; na = na < 0 ? 0 : (na > RA ? RA : na);      // blocks of this wave (uniform)
    int nb = tbk - wc * RX; nb = nb < 0 ? 0 : (nb > RX ? RX : nb);
    const int key = (i >> 1) & 7;
    const int rowA = (16 * wr * RA + i) * 128, rowX = (TA + 16 * wc * RX + i) * 128;
    float pa[RX], pb[RX];
#pragma unroll
    for (int b = 0; b < RX; ++b) {
        pa[b] = 0.f; pb[b] = 0.f;
        if constexpr (PRO) {
            const int k = 16 * (d.tbK * tk + wc * RX + b) + i;
            if (k < d.K) { pa[b] = d.ka[k]; pb[b] = d.kb[k]; }
        }
    }
    f32x4 acc[RA][RX];
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < RX; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nst) issue(s);

    // the stage loop exists once per (na, nb) of the wave (chosen here, outside the loop); GEN: run-time guards per block
    auto stages = [&](auto NAc, auto NBc, auto GENc) {
        constexpr int NA = decltype(NAc)::value, NB = decltype(NBc)::value;
        constexpr bool GEN = decltype(GENc)::value;
#pragma nounroll
        for (int s = 0; s < nst; ++s) {
            // stage s has landed when only the younger stages of this wave's stream (up to NS - 2) are outstanding
            const int left = nst - 1 - s;
            dma::wait_vmcnt(npw * (left < NS - 2 ? left : NS - 2));
            __syncthreads();                                // every wave's pieces of stage s are in; stage s - 1 is consumed
            if (s + NS - 1 < nst) issue(s + NS - 1);        // into the slot of stage s - 1
            const char* st = lds + (s % NS) * STAGE;
            const long long n_stage = n_lo + (long long)s * kStagePx;
            const bool last = s == nst - 1;
#pragma unroll
            for (int hh = 0; hh < (PH == 1 ? 2 : 1); ++hh) {
                const int h = PH == 1 ? hh : ph;
                const int off = ((4 * h + kq) ^ key) * 16;
                const bool masked = last && (n_stage + 4 * (4 * h + kq) >= n_hi);
                wg_round<RA, RX, NA, NB, PRO, GEN>(st + rowA + off, st + rowX + off, masked, pa, pb, d.relu_in, na, nb, acc);
            }
        }
    };
    using std::integral_constant;
    if (na == RA && nb == RX) stages(integral_constant<int, RA>{}, integral_constant<int, RX>{}, std::false_type{});
    else if (na == RA && nb == RX - 1) stages(integral_constant<int, RA>{}, integral_constant<int, RX - 1>{}, std::false_type{});
    else if (na == RA - 1 && nb == RX) stages(integral_constant<int, RA - 1>{}, integral_constant<int, RX>{}, std::false_type{});
    else if (na == RA - 1 && nb == RX - 1) stages(integral_constant<int, RA - 1>{}, integral_constant<int, RX - 1>{}, std::false_type{});
    else stages(integral_constant<int, RA>{}, integral_constant<int, RX>{}, std::true_type{});      // ragged edge tiles / idle waves

    // ---- the second pixel phase hands its accumulators to the first through LDS (the ring is free now) ----
    if constexpr (PH == 2) {
        __syncthreads();
        float* buf = reinterpret_cast<float*>(lds) + (size_t)wt * (RA * RX * 4 * 64);
        if (ph == 1) {
#pragma unroll
            for (int a = 0; a < RA; ++a)
#pragma unroll
                for (int b = 0; b < RX; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) buf[((a * RX + b) * 4 + r) * 64 + lane] = acc[a][b][r];
        }
        __syncthreads();
        if (ph == 0) {
#pragma unroll
            for (int a = 0; a < RA; ++a)
#pragma unroll
                for (int b = 0; b < RX; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][b][r] += buf[((a * RX + b) * 4 + r) * 64 + lane];
        }
    }
    if (ph != 0) return;
    // acc[a][b][r]: m = 16 (block a) + 4 kq + r, k = 16 (block b) + i
    float* out = ws + (size_t)split * d.M * d.K;
#pragma unroll
    for (int a = 0; a < RA; ++a)
#pragma unroll
        for (int b = 0; b < RX; ++b) {
            if (a >= na || b >= nb) continue;
            const int k = 16 * (d.tbK * tk + wc * RX + b) + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * (d.tbM * tm + wr * RA + a) + 4 * kq + r;
                if (m < d.M && k < d.K) out[(size_t)m * d.K + k] = acc[a][b][r];
            }
        }
}

// out[i] = sum over the S partial matrices, fixed order: 4 slices of the split range per output (one per wave), each summed
// front to back, then the 4 slice sums added in slice order
__global__ __launch_bounds__(kBlock) void k_pw2_reduce(const float* __restrict__ in, float* __restrict__ out, int MK, int S) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + lane;
    const int per = (S + 3) / 4, c0 = slice * per, c1 = (c0 + per) < S ? (c0 + per) : S;
    float acc = 0.f;
    if (idx < MK) {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) acc += in[(size_t)c * MK + idx];
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && idx < MK) out[idx] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// RB (row blocks per wave) for nrb row blocks: the one that pads least; ties -> the larger
inline int pick_rb(int nrb) {
    const int cand[3] = {5, 4, 3};
    int best = 3, waste = 1 << 30;
    for (int c : cand) {
        const int w = (nrb + c - 1) / c * c - nrb;
        if (w < waste) { waste = w; best = c; }
    }
    if (nrb % 3 == 0 && nrb >= 9) best = 3;                 // deep layers: many small units balance best
    return best;
}


inline bool gemm_cfg(GCfg& c, int K, int M, int a_is_mk, const float* A) {
    const int nrb = (M + 15) / 16;
    c.rb = pick_rb(nrb);
    const int nrg = (nrb + c.rb - 1) / c.rb;
    if (nrg > 16) return false;
    c.ct = (nrg % 2 == 0) ? 2 : 1;
    if (!a_is_mk) c.amode = 1;
    else if (K % 4 == 0 && ((uintptr_t)A & 15) == 0) c.amode = 0;
    else {
        const int ng = (K + 15) / 16;
        const size_t bytes = (size_t)16 * ng * (16 * nrb + 4) * sizeof(float);
        if (bytes > 64 * 1024) return false;
        c.amode = 2;
    }
    return true;
}

template <int RB, int AMODE, bool PRO, int EPI, bool OUTAFF>
int launch_gemm(const float* A, const float* X, const float* R, float* Y, const GDims& d, const GFuse& fz, const GTrain& tr,
                hipStream_t stream) {
    const long long tiles = (d.ntot + 63) / 64;
    const long long units = tiles * d.nrg;
    const size_t lds = ((PRO ? 32 * d.ngroups : 0) + (AMODE == 2 ? (size_t)16 * d.ngroups * d.mpad : 0)) * sizeof(float);
    unsigned grid = (unsigned)((units + 3) / 4);
    if (d.ct == 2) {
        const long long tps = (tiles + 1) / 2;
        grid = (unsigned)(((tps + 7) / 8) * (d.nrg / 2) * 8);
    }
    hipLaunchKernelGGL((k_pw2_gemm<RB, AMODE, PRO, EPI, OUTAFF>), dim3(grid), dim3(256),
                       lds, stream, A, X, R, Y, d, fz, tr);
    return launch_status();
}

template <int RB, int AMODE>
int launch_gemm_flags(int pro, int epi, int outaff, const float* A, const float* X, const float* R, float* Y, const GDims& d,
                      const GFuse& fz, const GTrain& tr, hipStream_t stream) {
    if (epi == 2) {
        if constexpr (AMODE != 0) return launch_gemm<RB, AMODE, false, 2, false>(A, X, R, Y, d, fz, tr, stream);
        else return RK_ERR_UNSUPPORTED;
    }
    if (epi == 1) {
        if constexpr (AMODE != 1) {
            return pro ? launch_gemm<RB, AMODE, true, 1, false>(A, X, R, Y, d, fz, tr, stream)
                       : launch_gemm<RB, AMODE, false, 1, false>(A, X, R, Y, d, fz, tr, stream);
        } else return RK_ERR_UNSUPPORTED;
    }
    if (outaff) {
        if constexpr (AMODE != 1) {
            return pro ? launch_gemm<RB, AMODE, true, 0, true>(A, X, R, Y, d, fz, tr, stream)
                       : launch_gemm<RB, AMODE, false, 0, true>(A, X, R, Y, d, fz, tr, stream);
        } else return RK_ERR_UNSUPPORTED;
    }
    if (pro) {
        if constexpr (AMODE != 1) return launch_gemm<RB, AMODE, true, 0, false>(A, X, R, Y, d, fz, tr, stream);
        else return RK_ERR_UNSUPPORTED;
    }
    return launch_gemm<RB, AMODE, false, 0, false>(A, X, R, Y, d, fz, tr, stream);
}

}  // namespace pw2
}  // namespace rk

using namespace rk;
using namespace rk::pw2;

namespace rk {
namespace pw2 {

// Should this generation run the call?  Measured against rk_pw.hip (tools/pw2_probe.py, [256, K -> M, P]): ahead by 1.2-1.4x
// up to 224 rows (54 ... 216 channels), level at 288 rows (where both sit at ~55 % of the f32 MFMA rate: DESIGN 3.5c), and
// behind when A needs the LDS image (K % 4 != 0 with the [M][K] layout: 54 input channels).
bool gemm_wanted(int K, int M, int P, int a_is_mk, const float* A) {
    static const int mode = [] { const char* e = getenv("RK_PW2"); return e ? atoi(e) : 1; }();   // 0: off, 2: wherever it can
    if (mode == 0 || P % 4 != 0) return false;
    GCfg c;
    if (!gemm_cfg(c, K, M, a_is_mk, A)) return false;
    if (mode == 2) return true;
    return c.amode != 2 && M <= 224;
}
bool wgrad_wanted(int P) {
    static const int mode = [] { const char* e = getenv("RK_PW2"); return e ? atoi(e) : 1; }();
    return mode != 0 && P % 4 == 0;
}

// Y[f] = epi(A pro(X[f])) (+ R[f]) on the second-generation kernel;
// RK_ERR_UNSUPPORTED when the shape / layout has no instance (the caller then runs rk_pw.hip's kernel).
// cfg_override: rb / amode / ct > 0 replace the planner's choice.
int gemm(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk,
         const GFuse* fuse, const GTrain* train, int epi, hipStream_t stream, const GCfg* cfg_override) {
    if (!A || !X || !Y) return RK_ERR_NULL_POINTER;
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P % 4 != 0) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)X & 15) || ((uintptr_t)Y & 15) || (R && ((uintptr_t)R & 15))) return RK_ERR_BAD_DIMS;
    GCfg c;
    if (!gemm_cfg(c, K, M, a_is_mk, A)) return RK_ERR_UNSUPPORTED;
    if (cfg_override) {
        if (cfg_override->rb > 0) c.rb = cfg_override->rb;
        if (cfg_override->amode >= 0) c.amode = cfg_override->amode;
        if (cfg_override->ct > 0) c.ct = cfg_override->ct;
    }
    GDims d;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P; d.a_is_mk = a_is_mk;
    const int nrb = (M + 15) / 16;
    d.nrg = (nrb + c.rb - 1) / c.rb; d.ct = c.ct; d.ngroups = (K + 15) / 16; d.mpad = 16 * nrb + 4;
    if (c.amode == 0 && (!a_is_mk || K % 4 != 0 || ((uintptr_t)A & 15))) return RK_ERR_UNSUPPORTED;
    if (c.amode == 1 && a_is_mk) return RK_ERR_UNSUPPORTED;
    if (c.amode == 2 && (size_t)16 * d.ngroups * d.mpad * sizeof(float) > 64 * 1024) return RK_ERR_UNSUPPORTED;
    GFuse fz = fuse ? *fuse : GFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
    GTrain tr = train ? *train : GTrain{nullptr, nullptr, nullptr, nullptr, 0};
    const int pro = fz.ka != nullptr, outaff = fz.ma != nullptr;
    if (epi == 1 && !tr.stats) return RK_ERR_NULL_POINTER;
    if (epi == 2 && !(tr.bred && tr.bx && tr.bpack)) return RK_ERR_NULL_POINTER;
    if (epi && (long long)tr.J * 64 < d.ntot) return RK_ERR_BAD_DIMS;
    if (epi && outaff) return RK_ERR_BAD_DIMS;
#define RK_G2(RBV) do { \
        if (c.amode == 0) return launch_gemm_flags<RBV, 0>(pro, epi, outaff, A, X, R, Y, d, fz, tr, stream); \
        if (c.amode == 1) return launch_gemm_flags<RBV, 1>(pro, epi, outaff, A, X, R, Y, d, fz, tr, stream); \
        return launch_gemm_flags<RBV, 2>(pro, epi, outaff, A, X, R, Y, d, fz, tr, stream); } while (0)
    switch (c.rb) {
        case 3: RK_G2(3);
        case 4: RK_G2(4);
        case 5: RK_G2(5);
        default: return RK_ERR_UNSUPPORTED;
    }
#undef RK_G2
}

// ---- d(weight) planner ----
// WCfg: id = index into the instance table below; ns = stages; splits: 0 = planner

struct WInst { int RA, RX, WR, WC, PH; };
constexpr WInst kWInst[] = {
    {3, 3, 2, 2, 1},     // 0:  96 x  96, 4 waves
    {3, 3, 2, 2, 2},     // 1:  96 x  96, 8 waves
    {3, 3, 3, 2, 2},     // 2: 144 x  96, 12 waves
    {3, 3, 1, 2, 2},     // 3:  48 x  96, 4 waves
    {2, 2, 2, 2, 2},     // 4:  64 x  64, 8 waves
    {4, 4, 1, 1, 2},     // 5:  64 x  64, 2 waves
    {4, 4, 2, 2, 1},     // 6: 128 x 128, 4 waves (7 + 7 blocks -> 4 + 3 / 4 + 3)
    {4, 4, 2, 2, 2},     // 7: 128 x 128, 8 waves
    {5, 3, 1, 2, 2},     // 8:  80 x  96, 4 waves (5 blocks x (3 + 2))
    {3, 3, 2, 3, 2},     // 9:  96 x 144, 12 waves
};
constexpr int kNWInst = sizeof(kWInst) / sizeof(kWInst[0]);

inline int make_wdims(WDims& d, const WInst& in, int ns, int F, int K, int M, int P, int splits) {
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P % 4 != 0) return RK_ERR_BAD_DIMS;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P;
    d.ka = d.kb = nullptr; d.relu_in = 0;
    d.nbM = (M + 15) / 16; d.nbK = (K + 15) / 16;
    const int capM = in.RA * in.WR, capK = in.RX * in.WC;
    d.tilesM = (d.nbM + capM - 1) / capM; d.tilesK = (d.nbK + capK - 1) / capK;
    d.tbM = (d.nbM + d.tilesM - 1) / d.tilesM; d.tbK = (d.nbK + d.tilesK - 1) / d.tilesK;
    const int T = d.tilesM * d.tilesK;
    long long S = splits;
    if (S <= 0) {
        // one full round of resident workgroups: per CU as many as the LDS ring (and 20 waves) allow; partial matrices
        // under a quarter of the operands
        const int waves = in.WR * in.WC * in.PH;
        const int rows = 16 * in.RA * in.WR + 16 * in.RX * in.WC;
        const long long lds = (long long)ns * rows * 128;
        long long wpc = (160 * 1024) / lds;
        if (wpc > 20 / waves) wpc = 20 / waves;
        if (wpc < 1) wpc = 1;
        S = ((long long)device_cus() * wpc) / T;             // (workspace size and launch read the same cached count)
        const long long cap = ((long long)(M + K) * d.ntot) / (4LL * M * K);
        if (S > cap) S = cap;
    }
    const long long by_steps = d.ntot / (4 * kStagePx);               // at least 4 stages per split
    if (S > by_steps) S = by_steps;
    if (S < 1) S = 1;
    long long span = (d.ntot + S - 1) / S;
    span = (span + kStagePx - 1) / kStagePx * kStagePx;
    d.span = span;
    d.S = (int)((d.ntot + span - 1) / span);
    return RK_OK;
}

template <int RA, int RX, int WR, int WC, int PH, int NS, bool PRO>
int launch_wgrad_i(const float* dY, const float* X, float* ws, const WDims& d, hipStream_t stream) {
    constexpr int ROWS = 16 * RA * WR + 16 * RX * WC;
    constexpr size_t ring = (size_t)NS * ROWS * 128;
    constexpr size_t fold = PH == 2 ? (size_t)WR * WC * RA * RX * 4 * 64 * sizeof(float) : 0;
    constexpr size_t lds = ring > fold ? ring : fold;
    static DynLdsRaised raised;                      // per instance and device (rk_common.hpp)
    if (const int rc = raise_dynamic_lds(reinterpret_cast<const void*>(&k_pw2_wgrad<RA, RX, WR, WC, PH, NS, PRO>), lds, raised))
        return rc;
    const int T = d.tilesM * d.tilesK;
    const unsigned grid = (unsigned)(((d.S + 7) / 8) * 8 * T);
    hipLaunchKernelGGL((k_pw2_wgrad<RA, RX, WR, WC, PH, NS, PRO>), dim3(grid), dim3(64 * WR * WC * PH), lds, stream, dY, X, ws, d);
    return launch_status();
}
template <int RA, int RX, int WR, int WC, int PH>
int launch_wgrad_ns(int ns, bool pro, const float* dY, const float* X, float* ws, const WDims& d, hipStream_t stream) {
    if (ns == 2) return pro ? launch_wgrad_i<RA, RX, WR, WC, PH, 2, true>(dY, X, ws, d, stream)
                            : launch_wgrad_i<RA, RX, WR, WC, PH, 2, false>(dY, X, ws, d, stream);
    return pro ? launch_wgrad_i<RA, RX, WR, WC, PH, 3, true>(dY, X, ws, d, stream)
               : launch_wgrad_i<RA, RX, WR, WC, PH, 3, false>(dY, X, ws, d, stream);
}
inline int launch_wgrad(int id, int ns, bool pro, const float* dY, const float* X, float* ws, const WDims& d, hipStream_t stream) {
    switch (id) {
        case 0: return launch_wgrad_ns<3, 3, 2, 2, 1>(ns, pro, dY, X, ws, d, stream);
        case 1: return launch_wgrad_ns<3, 3, 2, 2, 2>(ns, pro, dY, X, ws, d, stream);
        case 2: return launch_wgrad_ns<3, 3, 3, 2, 2>(ns, pro, dY, X, ws, d, stream);
        case 3: return launch_wgrad_ns<3, 3, 1, 2, 2>(ns, pro, dY, X, ws, d, stream);
        case 4: return launch_wgrad_ns<2, 2, 2, 2, 2>(ns, pro, dY, X, ws, d, stream);
        case 5: return launch_wgrad_ns<4, 4, 1, 1, 2>(ns, pro, dY, X, ws, d, stream);
        case 6: return launch_wgrad_ns<4, 4, 2, 2, 1>(ns, pro, dY, X, ws, d, stream);
        case 7: return launch_wgrad_ns<4, 4, 2, 2, 2>(ns, pro, dY, X, ws, d, stream);
        case 8: return launch_wgrad_ns<5, 3, 1, 2, 2>(ns, pro, dY, X, ws, d, stream);
        case 9: return launch_wgrad_ns<3, 3, 2, 3, 2>(ns, pro, dY, X, ws, d, stream);
        default: return RK_ERR_UNSUPPORTED;
    }
}

// planner.  A workgroup's waves meet at a barrier per stage, so a tile costs (its busiest wave's blocks) x (wave tiles);
// efficiency of an instance = real blocks / that cost summed over the tiles.  Candidates in order of preference (ties):
// the measured winners of tools/pw2_probe.py -- 96 x 96 / 8 waves (288, 576), 128 x 128 / 8 waves (108, 216: 7 = 4 + 3
// blocks), 144 x 96 / 12 waves (144), 80 x 96 / 4 waves (72: 5 x (3 + 2) blocks), 64 x 64 / 8 waves (54).
inline double inst_efficiency(const WInst& in, int nbM, int nbK) {
    const int capM = in.RA * in.WR, capK = in.RX * in.WC;
    const int tilesM = (nbM + capM - 1) / capM, tilesK = (nbK + capK - 1) / capK;
    const int tbM = (nbM + tilesM - 1) / tilesM, tbK = (nbK + tilesK - 1) / tilesK;
    long long cost = 0;
    for (int tm = 0; tm < tilesM; ++tm)
        for (int tk = 0; tk < tilesK; ++tk) {
            int tbm = nbM - tm * tbM; tbm = tbm < tbM ? tbm : tbM;
            int tbk = nbK - tk * tbK; tbk = tbk < tbK ? tbk : tbK;
            int busiest = 0;
            for (int wr = 0; wr < in.WR; ++wr)
                for (int wc = 0; wc < in.WC; ++wc) {
                    int na = tbm - wr * in.RA; na = na < 0 ? 0 : (na > in.RA ? in.RA : na);
                    int nb = tbk - wc * in.RX; nb = nb < 0 ? 0 : (nb > in.RX ? in.RX : nb);
                    busiest = na * nb > busiest ? na * nb : busiest;
                }
            cost += (long long)busiest * in.WR * in.WC;
        }
    return cost > 0 ? (double)nbM * nbK / (double)cost : 0.0;
}
inline WCfg wgrad_cfg(int K, int M) {
    const int nbM = (M + 15) / 16, nbK = (K + 15) / 16;
    const int order[5] = {1, 7, 2, 8, 4};
    const int stages[5] = {2, 3, 2, 3, 2};
    WCfg c{1, 2, 0};
    double best = -1.0;
    for (int i = 0; i < 5; ++i) {
        const double e = inst_efficiency(kWInst[order[i]], nbM, nbK);
        if (e > best + 0.03) { best = e; c.id = order[i]; c.ns = stages[i]; }
    }
    return c;
}

size_t wgrad_workspace_bytes(int F, int K, int M, int P) {
    const WCfg c = wgrad_cfg(K, M);
    WDims d;
    if (make_wdims(d, kWInst[c.id], c.ns, F, K, M, P, 0)) return 0;
    return (size_t)d.S * M * K * sizeof(float);
}

int wgrad(const float* dY, const float* X, float* dW, int F, int K, int M, int P, void* ws, size_t ws_bytes,
          const float* ka, const float* kb, int relu_in, hipStream_t stream, const WCfg* cfg_override) {
    if (!dY || !X || !dW) return RK_ERR_NULL_POINTER;
    if (((uintptr_t)dY & 15) || ((uintptr_t)X & 15)) return RK_ERR_BAD_DIMS;
    WCfg c = wgrad_cfg(K, M);
    if (cfg_override) {
        if (cfg_override->id >= 0) c.id = cfg_override->id;
        if (cfg_override->ns > 0) c.ns = cfg_override->ns;
        c.splits = cfg_override->splits;
    }
    if (c.id < 0 || c.id >= kNWInst) return RK_ERR_UNSUPPORTED;
    WDims d;
    if (int rc = make_wdims(d, kWInst[c.id], c.ns, F, K, M, P, c.splits)) return rc;
    if (!ws || ws_bytes < (size_t)d.S * M * K * sizeof(float)) return RK_ERR_WORKSPACE;
    d.ka = ka; d.kb = kb; d.relu_in = relu_in;
    if (int rc = launch_wgrad(c.id, c.ns, ka && kb, dY, X, (float*)ws, d, stream)) return rc;
    const int MK = M * K;
    if (!launch_reduce_partials4((const float*)ws, dW, MK, d.S, stream))
        hipLaunchKernelGGL(k_pw2_reduce, dim3((MK + 63) / 64), dim3(kBlock), 0, stream, (const float*)ws, dW, MK, d.S);
    return launch_status();
}

}  // namespace pw2
}  // namespace rk

extern "C" {

// Tuning / test hooks of the second-generation fp32 kernels: the same operations as rk_pw_gemm_f32 / rk_pw_wgrad_f32 with
// the kernel configuration given explicitly (negative / zero = the planner's choice).  RK_ERR_UNSUPPORTED when the
// configuration has no instance.
int rk_pw2_gemm_cfg_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk,
                        const float* ka, const float* kb, int relu_in, int rb, int amode, int ct, rk_stream_t stream) {
    const GFuse fz{ka, kb, nullptr, nullptr, relu_in, 0};
    const GCfg c{rb, amode, ct};
    return pw2::gemm(A, X, R, Y, F, K, M, P, a_is_mk, &fz, nullptr, 0, (hipStream_t)stream, &c);
}
size_t rk_pw2_wgrad_workspace_bytes(int F, int K, int M, int P) { return pw2::wgrad_workspace_bytes(F, K, M, P); }
int rk_pw2_wgrad_cfg_f32(const float* dY, const float* X, float* dW, int F, int K, int M, int P, void* ws, size_t ws_bytes,
                         const float* ka, const float* kb, int relu_in, int inst, int stages, int splits, rk_stream_t stream) {
    const WCfg c{inst, stages, splits};
    return pw2::wgrad(dY, X, dW, F, K, M, P, ws, ws_bytes, ka, kb, relu_in, (hipStream_t)stream, &c);
}

}  // extern "C"
