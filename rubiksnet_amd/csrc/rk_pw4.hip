// rk_pw4.hip -- streaming fp32 GEMM for the SHALLOW 1x1 convolutions (unfused half;
// rubiksnet/backbone.py:44-45, :123-135): 54 -> 54 and 72 -> 72 channels on 56 x 56 and 112 x 112 planes.
//
// Why another kernel.  On these layers the matrix pipe and HBM are equally loaded ([256, 72 -> 72, 56 x 56]: 59 us of
// v_mfma_f32_16x16x4_f32 issue, padded to 80 rows, against 58 us of HBM traffic at 8 TB/s), so a kernel has to keep BOTH
// busy all the time.  rk_pw2.hip's barrier-free GEMM does neither: a wave starts every 64-pixel unit with cold loads
// (4 KB in flight per wave and round trip, 4-5 round trips per unit, plus one per row block in the training epilogues) and
// re-reads the small operand from L2 once per unit -- 103 / 124 us on the two shapes, 198 / 389 us with the BatchNorm-
// backward epilogue.  Here:
//
//   * the small operand lives in REGISTERS for the whole kernel (lane (j, kq) holds A[16 rb + j][4 st + kq] for every row
//     block rb and k-step st: 56 VGPRs at 54 x 54, 90 at 72 x 72);
//   * a wave is persistent over a contiguous range of 64-pixel column tiles and everything it reads from HBM arrives as a
//     FIFO of 1 KB RECORDS in its private LDS ring, filled by LDS-DMA (global_load_lds_dwordx4) NREC - 1 records ahead,
//     across tile boundaries, with one counted s_waitcnt vmcnt per record.  A record = one wave-instruction = 4 rows x 256 B
//     (lane l -> row l >> 4, 16-byte piece l & 15), written lane-linear and read back lane-linear with ONE ds_read_b128:
//       - X record st: rows 4 st .. 4 st + 3 of the tile = exactly the B operands of k-step st for the four interleaved
//         column blocks (lane (j, kq): k = 4 st + kq, pixels 4 j .. 4 j + 3);
//       - epilogue record (rb, r) (residual R, or the BatchNorm input x of the bn-backward epilogue): rows
//         16 rb + 4 kq + r = exactly the C-layout rows of accumulator component r of row block rb.
//     No barrier, no workgroup coupling; 12-16 KB in flight per wave all the time;
//   * results leave as 16-byte stores straight from the accumulators; statistics / BatchNorm-backward sums by DPP row sums
//     as in rk_pw2.hip (same 64-column tile records, so `rk_pw_gemm_tiles` is unchanged).
//
// Arithmetic: the fmaf chain of v_mfma_f32_16x16x4_f32 in k order, as rk_pw2.hip.
#include <type_traits>
#include "rk_common.hpp"
#include "rk_dma.hpp"
#include "rk_pw4.hpp"

namespace rk {
namespace pw4 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Dims {
    int F, K, M, P;
    long long ntot;          // F * P columns
    long long ntiles;        // ceil(ntot / 64)
    int a_is_mk;
};

#ifndef RK_PW4_NT
#define RK_PW4_NT 2      // 1: non-temporal DMA loads, 2: non-temporal Y stores (measured: stores help, loads within noise)
#endif
#if RK_PW4_NT & 1
#define RK_PW4_LDNT " nt"
#else
#define RK_PW4_LDNT ""
#endif
__device__ __forceinline__ void dma16v(const void* p, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "global_load_lds_dwordx4 %1, off" RK_PW4_LDNT "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(p), "s"(lds_dst_uniform)
        : "memory");
}
__device__ __forceinline__ float row16_sum_to_lane15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    return v;
}
__device__ __forceinline__ void store_y(float* p, const float4& o) {
#if RK_PW4_NT & 2
    f32x4 t = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = o;
#endif
}
template <int N, typename Fn> __device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (N > 0) {
        static_for<N - 1>(fn);
        fn(std::integral_constant<int, N - 1>{});
    }
}

constexpr int waves_of(bool ALDS) { return ALDS ? 8 : 4; }     // waves per workgroup (independent of each other but for the A table)

// Stores CERTAINLY issued (as wave instructions) in the slots a .. b of the record stream, slot 0 = the current tile's first
// (negative: the previous tile's; `first`: those do not exist).  A slot's stores follow its DMA.  Row blocks below the
// last one always hold rows (< M) and every tile in range has a valid column, so each of their rows issues `ns` stores;
// the last row block is not counted (fewer = a stricter wait).  NE == 0: all of a tile's stores follow slot NRT - 1.
constexpr int certain_stores(int a, int b, bool first, int NRT, int NST, int NE, int RB, int ns) {
    const bool ne0 = NE == 0;
    int n = 0;
    for (int u = a; u <= b; ++u) {
        int uu = u;
        if (u < 0) {
            if (first) continue;
            uu = u + NRT;
        }
        if (ne0) n += uu == NRT - 1 ? 4 * (RB - 1) * ns : 0;
        else if (uu >= NST && uu < NST + NE && (uu - NST) / 4 < RB - 1) n += ns;
    }
    return n;
}
// Ring geometry: NREC slots with NREC | NRT, so that record s of a tile always sits in slot s % NREC (immediate LDS
// offsets, no slot counter).  maxrec bounds the ring by the LDS budget; when NRT has no divisor in [8, maxrec] the tile is
// padded with dummy records (a 1 KB re-read of the head of X, ignored) until it has one.
constexpr int best_div(int n, int maxrec) {
    for (int dd = maxrec; dd >= 2; --dd)
        if (n % dd == 0) return dd;
    return 1;
}
constexpr int pad_of(int n, int maxrec) {
    for (int p = 0; p < 8; ++p)
        if (best_div(n + p, maxrec) >= 8) return p;
    return 0;
}
constexpr int maxrec_of(bool ALDS) { return ALDS ? 17 : 19; }       // 8 x 17 KB + 23 KB table / 2 x 4 x 19 KB

// RB row blocks of 16, NST k-steps of 4 (A zero-padded to 16 RB x 4 NST in registers).
// PRO: B operands pass through relu?(ka[k] x + kb[k]).  EPI 0 / 1 (statistics tiles) / 2 (BatchNorm-backward mask + sums,
// x by records).  RES: + R (by records).
// ALDS: the small operand in an LDS table shared by the workgroup's 8 waves instead of registers, fragment-major
// ([rb][st][lane]: one conflict-free ds_read_b32 per row block and k-step, requested one slot ahead) -- for the instances
// whose operand + accumulators + epilogue do not fit 256 registers (72 channels with a training epilogue: 90 + 80 + ...).
template <int RB, int NST, bool PRO, int EPI, bool RES, bool ALDS>
__global__ __launch_bounds__(64 * waves_of(ALDS), ALDS ? 1 : 2) void k_pw4_gemm(const float* __restrict__ A, const float* __restrict__ X,
                                                             const float* __restrict__ R, float* __restrict__ Y, Dims d,
                                                             pw2::GFuse fz, pw2::GTrain tr) {
    static_assert(!(EPI == 2 && (RES || PRO)), "no such instance");
    constexpr int NE = (EPI == 2 || RES) ? 4 * RB : 0;           // epilogue records per tile
    constexpr int NPAD = pad_of(NST + NE, maxrec_of(ALDS));      // dummy records per tile
    constexpr int NRT = NST + NE + NPAD;                         // records per tile
    constexpr int NREC = best_div(NRT, maxrec_of(ALDS));         // ring slots; NREC | NRT: record s of a tile sits in slot s % NREC
    constexpr int LOOK = NREC - 1;                               // DMAs in flight behind the record being consumed
    constexpr int NS = EPI == 0 ? 1 : 2;                         // stores per output row
    static_assert(LOOK >= 2 && LOOK <= NRT, "ring");
    constexpr int kWaves = waves_of(ALDS);
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int j = lane & 15, kq = lane >> 4;
    float* ring = lds_f + wave * (NREC * 256);
    float2* Ks = reinterpret_cast<float2*>(lds_f + kWaves * NREC * 256);              // [4 NST] (ka, kb)
    float4* Bp = reinterpret_cast<float4*>(lds_f + kWaves * NREC * 256 + 8 * NST);    // [16 RB] (a, b, mean, invstd)
    float* At = lds_f + kWaves * NREC * 256 + 8 * NST + 64 * RB;                      // ALDS: [RB][NST][64] fragments
    const int K = d.K, M = d.M, P = d.P;
    if constexpr (ALDS) {
        for (int e = threadIdx.x; e < RB * NST * 64; e += 64 * kWaves) {
            const int l = e & 63, fs = e >> 6, st = fs % NST, rb = fs / NST;
            const int m = 16 * rb + (l & 15), k = 4 * st + (l >> 4);
            const bool ok = m < M && k < K;
            const size_t at = d.a_is_mk ? (size_t)m * K + k : (size_t)k * M + m;
            const float v = A[ok ? at : 0];
            At[e] = ok ? v : 0.f;
        }
    }
    if constexpr (PRO) {
        for (int e = threadIdx.x; e < 4 * NST; e += 64 * kWaves)
            Ks[e] = e < K ? make_float2(fz.ka[e], fz.kb[e]) : make_float2(0.f, 0.f);
    }
    if constexpr (EPI == 2) {
        for (int e = threadIdx.x; e < 16 * RB; e += 64 * kWaves) Bp[e] = tr.bpack[e < M ? e : M - 1];
    }
    const bool outaff = EPI == 0 && fz.ma != nullptr;       // inference: Y = relu?(ma[m] (A X) + mb[m]) (+ R) (wave-uniform)
    if (outaff) {
        for (int e = threadIdx.x; e < 16 * RB; e += 64 * kWaves) {
            const int m = e < M ? e : M - 1;
            Bp[e] = make_float4(fz.ma[m], fz.mb[m], 0.f, 0.f);
        }
    }
    if (PRO || EPI == 2 || ALDS || outaff) __syncthreads();

    // this wave's tiles: gw, gw + nwv, gw + 2 nwv, ... -- at any moment the chip works on one contiguous span of nwv tiles
    // (256 B x nwv of every row: DRAM pages stay open); contiguous per-wave ranges scatter 256-byte pieces over the whole
    // tensor at every instant
    const int gw = blockIdx.x * kWaves + wave, nwv = gridDim.x * kWaves;
    if (gw >= d.ntiles) return;
    const int ntw = (int)((d.ntiles - gw + nwv - 1) / nwv);     // tiles of this wave: tile i = gw + i nwv

    const float* E = EPI == 2 ? tr.bx : R;
    // Per-lane geometry.  All per-record addresses are RUNNING pointers bumped by wave-uniform strides (P, 4 P, 13 P floats):
    // a table of per-record offsets (64-bit, per lane, loop invariant) is what the compiler builds otherwise -- 60 VGPRs.
    // The lane's column (f, p) of the producer's tile advances by 64 nwv columns per tile: (df, dp) with one carry.
    const unsigned cstep = 64u * (unsigned)nwv, df = cstep / (unsigned)P, dp = cstep - df * (unsigned)P;
    unsigned pf_, pp_;                                 // (frame, pixel) of this lane's 4 columns in the producer's NEXT tile
    {
        unsigned c = (unsigned)gw * 64u + 4u * (unsigned)j;     // (host: F P < 2^31)
        c = c < (unsigned)d.ntot ? c : 0u;
        pf_ = c / (unsigned)P; pp_ = c - pf_ * (unsigned)P;
    }
    int ptile = 0;                                    // index of the producer's next tile (clamped to the wave's last)
    const unsigned ring_b = dma::lds_byte_addr(ring);
    const size_t kqP = (size_t)kq * P;
    int xfix;                                         // rows past K (last X record) re-read row K - 1: correction in floats (<= 0)
    {
        const int k = 4 * (NST - 1) + kq;
        xfix = ((k < K ? k : K - 1) - k) * P;
    }
    const float* xp = nullptr;                        // producer: next X record of this lane (row 4 st + kq)
    const float* ep = nullptr;                        // producer: next epilogue record (row 16 rb + 4 kq + r)
    size_t yo_next = 0;                               // Y offset of the tile the producer entered last
    auto enter_tile = [&]() {                         // the producer crosses into its next tile
        const unsigned f = (pf_ * (unsigned)P + pp_) < (unsigned)d.ntot ? pf_ : 0u, p = f == pf_ ? pp_ : 0u;   // (columns past the end: column 0)
        xp = X + ((size_t)f * K) * P + p + kqP;
        yo_next = ((size_t)f * M) * P + p;
        if constexpr (NE > 0) ep = E + yo_next + 4 * kqP;
        if (ptile + 1 < ntw) {                        // (past the wave's last tile: the last one again, never consumed)
            ++ptile;
            pf_ += df; pp_ += dp;
            if (pp_ >= (unsigned)P) { pp_ -= (unsigned)P; ++pf_; }
        }
    };
    // record ri of the producer's tile -> ring slot ri % NREC
    auto issue = [&](auto RIc) {
        constexpr int ri = decltype(RIc)::value;
        if constexpr (ri == 0) enter_tile();
        const float* src;
        if constexpr (ri >= NST + NE) {
            src = X + lane * 4;                       // padding record (host: X holds >= 256 floats)
        } else if constexpr (ri < NST) {
            src = ri == NST - 1 ? xp + xfix : xp;
            xp += 4 * (size_t)P;
        } else {
            constexpr int e = ri - NST, rb = e / 4, r = e % 4;
            src = ep;
            if constexpr (rb == RB - 1) {             // rows past M: a valid row, never stored or summed
                const int m = 16 * rb + 4 * kq + r;
                src += ((m < M ? m : M - 1) - m) * P;
            }
            ep += (r == 3 ? 13 : 1) * (size_t)P;
        }
        dma16v(src, ring_b + (unsigned)((ri % NREC) * 1024));
    };
    auto read_rec = [&](auto RIc) {                   // record ri of the consumer's tile (or ri - NRT of the next)
        constexpr int ri = decltype(RIc)::value;
        return *reinterpret_cast<const f32x4*>(ring + (ri % NREC) * 256 + lane * 4);
    };

    f32x4 acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[rb][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    static_for<LOOK>([&](auto I) { issue(I); });
    size_t yo_c = yo_next, yo_next_c = yo_next;       // consumer's tile, and its successor's

    // the small operand, once (behind the first records' DMAs): a[rb][st] = A[16 rb + j][4 st + kq] (zero outside)
    float a[ALDS ? 1 : RB][ALDS ? 1 : NST];
    if constexpr (!ALDS) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int m = 16 * rb + j, k = 4 * st + kq;
                const bool ok = m < M && k < K;
                const size_t at = d.a_is_mk ? (size_t)m * K + k : (size_t)k * M + m;
                const float v = A[ok ? at : 0];
                a[rb][st] = ok ? v : 0.f;
            }
    }
    float acur[RB];                                   // ALDS: the fragments of the k-step consumed next
    auto read_a = [&](auto STc, float (&dst)[RB]) {
        constexpr int st = decltype(STc)::value;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) dst[rb] = At[(rb * NST + st) * 64 + lane];
    };
    if constexpr (ALDS) read_a(std::integral_constant<int, 0>{}, acur);

    // Record s is READ one slot early (under the MFMAs of record s - 1): two waves of a SIMD run the same instruction stream
    // in phase, so an LDS round trip at the head of every 20-MFMA batch is paid by both at once (measured: the MFMA-only
    // skeleton at 1.25x the issue time).
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOOK - 1) : "memory");
    f32x4 vcur = read_rec(std::integral_constant<int, 0>{});

    int t = 0;
    auto tile_body = [&](auto FIRSTc) {
        constexpr bool FIRST = decltype(FIRSTc)::value;
        const long long tg = (long long)gw + (long long)t * nwv;      // the tile's global index
        const bool valid = tg * 64 + 4 * j < d.ntot;
        const long long nleft = d.ntot - tg * 64;
        const float ntile = (float)(nleft < 64 ? nleft : 64);
        float* yp = Y + yo_c + 4 * kqP;               // consumer: next output row of this lane (row 16 rb + 4 kq + r)
        size_t rec = (size_t)(4 * kq) * tr.J + (size_t)tg;      // ... and its tile record (EPI 1 / 2)
        // one output row group: acc[rb][q][r] = row 16 rb + 4 kq + r, pixel 4 j + q; ev = the epilogue record's 4 values
        auto finish_row = [&](auto RBc, auto Rc, const f32x4& ev) {
            constexpr int rb = decltype(RBc)::value, r = decltype(Rc)::value;
            const int m = 16 * rb + 4 * kq + r;
            // (only the last row block can hold rows past M -- the instance has exactly RB row blocks; testing every row
            // makes 2 x 4 RB loop-invariant lane masks = 80 SGPRs)
            const bool live = rb < RB - 1 || m < M;
            const bool on = valid && live;
            float4 o = make_float4(acc[rb][0][r], acc[rb][1][r], acc[rb][2][r], acc[rb][3][r]);
            if constexpr (EPI == 0) {
                if (outaff) {
                    const float4 pk = Bp[16 * rb + 4 * kq + r];
                    o.x = fmaf(pk.x, o.x, pk.y); o.y = fmaf(pk.x, o.y, pk.y); o.z = fmaf(pk.x, o.z, pk.y); o.w = fmaf(pk.x, o.w, pk.y);
                    if (fz.relu_out) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                }
            }
            if constexpr (RES) { o.x += ev[0]; o.y += ev[1]; o.z += ev[2]; o.w += ev[3]; }
            if constexpr (EPI == 0) {
                if (on) store_y(yp, o);
            } else {
                float s1 = 0.f, s2 = 0.f, piv = 0.f;
                if constexpr (EPI == 1) {
                    piv = __shfl(o.x, lane & 48);                     // the row's first column in this tile (always valid)
                    if (on) {
                        float tq;
                        tq = o.x - piv; s1 += tq; s2 = fmaf(tq, tq, s2);
                        tq = o.y - piv; s1 += tq; s2 = fmaf(tq, tq, s2);
                        tq = o.z - piv; s1 += tq; s2 = fmaf(tq, tq, s2);
                        tq = o.w - piv; s1 += tq; s2 = fmaf(tq, tq, s2);
                    }
                } else {
                    const float4 pk = Bp[16 * rb + 4 * kq + r];
                    if (on) {
                        const float pa = pk.x, pb = pk.y, mu = pk.z, iv = pk.w;
                        o.x = fmaf(pa, ev[0], pb) <= 0.f ? 0.f : o.x;  s1 += o.x;  s2 = fmaf(o.x, (ev[0] - mu) * iv, s2);
                        o.y = fmaf(pa, ev[1], pb) <= 0.f ? 0.f : o.y;  s1 += o.y;  s2 = fmaf(o.y, (ev[1] - mu) * iv, s2);
                        o.z = fmaf(pa, ev[2], pb) <= 0.f ? 0.f : o.z;  s1 += o.z;  s2 = fmaf(o.z, (ev[2] - mu) * iv, s2);
                        o.w = fmaf(pa, ev[3], pb) <= 0.f ? 0.f : o.w;  s1 += o.w;  s2 = fmaf(o.w, (ev[3] - mu) * iv, s2);
                    }
                }
                if (on) store_y(yp, o);
                s1 = row16_sum_to_lane15(s1);
                s2 = row16_sum_to_lane15(s2);
                if (j == 15 && live) {
                    if constexpr (EPI == 1) tr.stats[rec] = make_float4(piv, s1, s2, ntile);
                    else tr.bred[rec] = make_float2(s1, s2);
                }
                rec += (r == 3 ? 13 : 1) * (size_t)tr.J;
            }
            yp += (r == 3 ? 13 : 1) * (size_t)P;
        };

        static_for<NRT>([&](auto S) {
            constexpr int s = decltype(S)::value;
            // slot s: DMA of record s + LOOK (this tile or the next), then the READ of record s + 1 -- which needs that
            // record landed: VMEM retires in order, so "at most the operations issued after its DMA are outstanding":
            // the DMAs of slots s + 2 - LOOK .. s and the stores of slots s + 1 - LOOK .. s - 1, counted (a wait that also
            // covers the stores exposes the loaded write latency once per tile while the reads drain).
            issue(std::integral_constant<int, (s + LOOK) % NRT>{});
            if constexpr ((s + LOOK) % NRT == 0) yo_next_c = yo_next;         // (the producer entered the consumer's next tile)
            constexpr int W = (LOOK - 1) + certain_stores(s + 1 - LOOK, s - 1, FIRST, NRT, NST, NE, RB, NS);
            static_assert(W <= 63, "vmcnt is 6 bits");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
            const f32x4 vnext = read_rec(std::integral_constant<int, (s + 1) % NRT>{});
            const f32x4 v = vcur;
            float anext[RB];
            if constexpr (ALDS && ((s + 1) % NRT) < NST) read_a(std::integral_constant<int, (s + 1) % NRT>{}, anext);
            if constexpr (s < NST) {
                float b[4] = {v[0], v[1], v[2], v[3]};
                if constexpr (PRO) {
                    const float2 kk = Ks[4 * s + kq];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float tq = fmaf(kk.x, b[q], kk.y);
                        b[q] = fz.relu_in ? fmaxf(tq, 0.f) : tq;
                    }
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[rb][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ALDS ? acur[rb] : a[ALDS ? 0 : rb][ALDS ? 0 : s], b[q], acc[rb][q], 0, 0, 0);
            } else if constexpr (s < NST + NE) {
                constexpr int e = s - NST;
                finish_row(std::integral_constant<int, e / 4>{}, std::integral_constant<int, e % 4>{}, v);
            }
            vcur = vnext;
            if constexpr (ALDS && ((s + 1) % NRT) < NST) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acur[rb] = anext[rb];
            }
        });
        if constexpr (NE == 0) {
            static_for<4 * RB>([&](auto Ec) {
                constexpr int e = decltype(Ec)::value;
                finish_row(std::integral_constant<int, e / 4>{}, std::integral_constant<int, e % 4>{}, f32x4{0.f, 0.f, 0.f, 0.f});
            });
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[rb][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        yo_c = yo_next_c;
    };
    tile_body(std::true_type{});
    for (t = 1; t < ntw; ++t) tile_body(std::false_type{});
}

// ---- host ----
struct Inst { int rb, nst; };
inline bool pick(Inst& in, int K, int M) {
    const int nrb = (M + 15) / 16, nst = (K + 3) / 4;
    if (nrb == 4 && nst >= 12 && nst <= 14) { in = Inst{4, 14}; return true; }
    if (nrb == 5 && nst >= 16 && nst <= 18) { in = Inst{5, 18}; return true; }
    return false;
}
inline int mode() {
    static const int m = [] { const char* e = getenv("RK_PW4"); return e ? atoi(e) : 1; }();    // 0: off, 2: any size
    return m;
}
inline int num_cus() { return device_cus(); }        // cached per device, thread-safe (rk_common.hpp)
constexpr long long kMinTiles = 4096;                // below: rk_pw2.hip (more, shorter-lived waves)

template <int RB, int NST, bool PRO, int EPI, bool RES>
int launch(const float* A, const float* X, const float* R, float* Y, const Dims& d, const pw2::GFuse& fz, const pw2::GTrain& tr,
           hipStream_t stream) {
    constexpr bool ALDS = RB == 5 && EPI != 0;        // (the 72-channel instance with a training epilogue: see the kernel)
    constexpr int NE = (EPI == 2 || RES) ? 4 * RB : 0;
    constexpr int NREC = best_div(NST + NE + pad_of(NST + NE, maxrec_of(ALDS)), maxrec_of(ALDS));
    constexpr int kWaves = waves_of(ALDS);
    if ((long long)d.F * d.K * d.P < 256) return RK_ERR_UNSUPPORTED;
    constexpr int wgs_per_cu = ALDS ? 1 : 2;
    const size_t lds = (size_t)kWaves * NREC * 1024 + 8 * NST * sizeof(float) + 16 * RB * sizeof(float4) +
                       (ALDS ? (size_t)RB * NST * 256 : 0);
    if (lds > 160 * 1024) return RK_ERR_UNSUPPORTED;
    static DynLdsRaised raised;                      // per instance and device (rk_common.hpp)
    if (const int rc = raise_dynamic_lds(reinterpret_cast<const void*>(&k_pw4_gemm<RB, NST, PRO, EPI, RES, ALDS>), lds, raised))
        return rc;
    long long wgs = (long long)num_cus() * wgs_per_cu;
    const long long need = (d.ntiles + kWaves - 1) / kWaves;
    wgs = wgs < need ? wgs : need;
    hipLaunchKernelGGL((k_pw4_gemm<RB, NST, PRO, EPI, RES, ALDS>), dim3((unsigned)wgs), dim3(64 * kWaves), lds, stream, A, X, R, Y, d,
                       fz, tr);
    return launch_status();
}
template <int RB, int NST>
int launch_flags(int pro, int epi, int res, const float* A, const float* X, const float* R, float* Y, const Dims& d,
                 const pw2::GFuse& fz, const pw2::GTrain& tr, hipStream_t stream) {
    if (epi == 2) {
        if (pro || res) return RK_ERR_UNSUPPORTED;
        return launch<RB, NST, false, 2, false>(A, X, R, Y, d, fz, tr, stream);
    }
#define RK_P4(E) do { \
        if (pro) return res ? launch<RB, NST, true, E, true>(A, X, R, Y, d, fz, tr, stream) \
                            : launch<RB, NST, true, E, false>(A, X, R, Y, d, fz, tr, stream); \
        return res ? launch<RB, NST, false, E, true>(A, X, R, Y, d, fz, tr, stream) \
                   : launch<RB, NST, false, E, false>(A, X, R, Y, d, fz, tr, stream); } while (0)
    if (epi == 1) RK_P4(1);
    RK_P4(0);
#undef RK_P4
}

// epi / res: the epilogue the call will carry.  The 72-channel instance keeps its operand in registers (90 + 80 registers of
// operand + accumulators: the 256-VGPR limit of two waves per SIMD) for plain / + R calls and in an LDS table for the
// training epilogues; statistics + residual (38 records per tile) measured level with rk_pw2.hip (161 / 601 us against
// 160 / 591 at [256, 72 -> 72, 56 x 56 / 112 x 112]) and stays there.
long long tiles(int F, int K, int M, int P, int epi, int res, bool force) {
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P % 4 != 0 || (long long)F * P >= (1ll << 31)) return 0;
    if (mode() == 0) return 0;
    Inst in;
    if (!pick(in, K, M)) return 0;
    const long long nt = ((long long)F * P + 63) / 64;
    if (!force && mode() != 2) {
        if (nt < kMinTiles) return 0;
        if (in.rb == 5 && epi == 1 && res) return 0;
    }
    return nt;
}

int gemm(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk, const pw2::GFuse* fuse,
         const pw2::GTrain* train, int epi, hipStream_t stream, bool force) {
    if (!A || !X || !Y) return RK_ERR_NULL_POINTER;
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P % 4 != 0) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)X & 15) || ((uintptr_t)Y & 15) || (R && ((uintptr_t)R & 15))) return RK_ERR_BAD_DIMS;
    Inst in;
    if (!pick(in, K, M)) return RK_ERR_UNSUPPORTED;
    if (tiles(F, K, M, P, epi, R != nullptr, force) <= 0) return RK_ERR_UNSUPPORTED;
    pw2::GFuse fz = fuse ? *fuse : pw2::GFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
    pw2::GTrain tr = train ? *train : pw2::GTrain{nullptr, nullptr, nullptr, nullptr, 0};
    if (fz.ma && (epi != 0 || !fz.mb)) return RK_ERR_UNSUPPORTED;
    const int pro = fz.ka != nullptr, res = R != nullptr;
    Dims d;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P; d.ntiles = (d.ntot + 63) / 64; d.a_is_mk = a_is_mk;
    if (epi == 1 && !tr.stats) return RK_ERR_NULL_POINTER;
    if (epi == 2 && !(tr.bred && tr.bx && tr.bpack)) return RK_ERR_NULL_POINTER;
    if (epi == 2 && ((uintptr_t)tr.bx & 15)) return RK_ERR_BAD_DIMS;
    if (epi && (long long)tr.J != d.ntiles) return RK_ERR_BAD_DIMS;
    if (in.rb == 4) return launch_flags<4, 14>(pro, epi, res, A, X, R, Y, d, fz, tr, stream);
    return launch_flags<5, 18>(pro, epi, res, A, X, R, Y, d, fz, tr, stream);
}

}  // namespace pw4
}  // namespace rk

extern "C" {
using namespace rk;
// test / probe hook: the streaming kernel regardless of the size threshold (RK_ERR_UNSUPPORTED when no instance fits).
// epi 0: plain (+ prologue ka / kb, + R); 1: + statistics tiles (stats [M][tiles] float4); 2: BatchNorm-backward epilogue
// (bx, bpack [M][4], bred [M][tiles] float2); tiles = ceil(F P / 64).
int rk_pw4_gemm_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk,
                    const float* ka, const float* kb, int relu_in, int epi, void* stats, const float* bx, const float* bpack,
                    void* bred, int tiles, rk_stream_t stream) {
    const pw2::GFuse fz{ka, kb, nullptr, nullptr, relu_in, 0};
    const pw2::GTrain tr{(float4*)stats, (float2*)bred, bx, (const float4*)bpack, tiles};
    return pw4::gemm(A, X, R, Y, F, K, M, P, a_is_mk, &fz, &tr, epi, (hipStream_t)stream, true);
}
}  // extern "C"
