// rk_pw16.hip -- the 1x1 convolutions of the bf16 (autocast) train step, second generation (the unfused
// half; rubiksnet/backbone.py:44-45, :87-104 under torch.autocast(bfloat16)).  At 2 bytes per element these GEMMs are
// HBM-bound by a wide margin (Large-AQ's [256, 288, 14, 14] layer: 58 MB of operands against 8.3 GFLOP = 3.3 us of
// bf16 MFMA), so the kernels are built around "every activation byte crosses HBM once":
//
//   * forward / d(input) (k_pw16_gemm): a workgroup owns ALL rows of a 128-pixel column tile (up to 288 rows per
//     workgroup: 144 accumulator registers per lane, 2 workgroups per CU), so X is read once (the first-generation
//     kernel of rk_pw.hip re-read it once per 64-row tile, 5x at 288 rows).  X is streamed by LDS-DMA in its memory
//     layout ([32 channels][128 pixels] per chunk, 4-byte pieces so that a tile may straddle frames at any P % 4 ==
//     0), 3 chunks deep; the channel-major -> k-major transposition a bf16 MFMA operand needs happens in REGISTERS
//     (8 ds_read_b64 + 16 v_perm_b32 per wave and chunk give the 4 B-fragments of a 64-pixel column group; no 2-byte
//     LDS writes).  The small operand is pre-packed once per weight version (k_pw16_pack) into bf16 MFMA A-fragment
//     order and DMA'd 16 B per lane, one chunk ahead.  v_mfma_f32_16x16x32_bf16: 16-row blocks fit 72 / 144 / 288
//     rows with little padding.
//   * d(weight) (k_pw16_wgrad): see below.
//
// Arithmetic: bf16 operands (the weight rounded to bf16 as autocast would), fp32 accumulation, one rounding of the
// result (+ residual) to bf16.
#include <type_traits>
#include "rk_common.hpp"
#include "rk_dma.hpp"

namespace rk {
namespace pw16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCh = 32;                    // channels per chunk = K of one MFMA
constexpr int kTilePx = 128;               // pixels per workgroup tile (2 column groups of 64)
constexpr int kXGroup = 4 * 2 * kTilePx + 64;   // bytes per 4 channel rows of an X stage (one DMA instruction) + pad: rows 8
                                               // apart sit 128 B apart mod 256 (conflict-free 8-byte fragment reads)
constexpr int kXStage = (kCh / 4) * kXGroup;    // 8 704 B

struct Dims {
    int F, K, M, P;
    long long ntot;
    int nrb, nch;                          // 16-row blocks of the packed operand, 32-channel chunks
    int U;                                 // 16-byte units (8 pixels) per frame row: ceil(P / 8)
    long long nunits;                      // F * U
    int dbg;
};

__device__ __forceinline__ unsigned bf16_bits(float f) {
    return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(f));
}

// ---- packing: fp32 [M][K] (mk != 0) or [K][M] -> bf16 fragments [chunk][row block][lane][8]; lane (m = l & 15,
// g = l >> 4) of block rb, chunk c holds rows 16 rb + m, reduction indices 32 c + 8 g .. + 7 (zeros outside) ----
__global__ __launch_bounds__(kBlock) void k_pw16_pack(const float* __restrict__ A, int M, int K, int mk, int nrb, int nch,
                                                      uint4* __restrict__ out) {
    const int u = blockIdx.x * kBlock + threadIdx.x;
    if (u >= nrb * nch * 64) return;
    const int lane = u & 63, rb = (u >> 6) % nrb, c = (u >> 6) / nrb;
    const int m = 16 * rb + (lane & 15), k0 = kCh * c + 8 * (lane >> 4);
    unsigned h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const bool ok = m < M && k < K;
        const float v = ok ? A[mk ? (size_t)m * K + k : (size_t)k * M + m] : 0.f;
        h[j] = bf16_bits(v);
    }
    out[u] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}

__device__ __forceinline__ const char* uniform_bytes(const char* p) {
    return reinterpret_cast<const char*>(dma::uniform_ptr(reinterpret_cast<const float*>(p)));
}

// Y[f] = A X[f] (+ R[f]).  RB: 16-row blocks per wave; the workgroup's 4 waves are 2 column groups x 2 row halves, so
// a workgroup covers 32 RB rows (blockIdx.y selects the row range when M is larger).  DX / DA: stages of the X / A
// rings.  Waves 0-1 issue the X DMAs and waves 2-3 the A DMAs: vmcnt is per wave and in order, so each stream is
// counted on its own (a wave issuing both could not wait for the one-ahead A chunk without draining the X prefetch).
template <int RB, bool RES, int DX, int DA>
__global__ __launch_bounds__(kBlock, 2) void k_pw16_gemm(const char* __restrict__ Apk, const __hip_bfloat16* __restrict__ X,
                                                         const __hip_bfloat16* R, __hip_bfloat16* Y, Dims d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int kAStage = 2 * RB * 1024;
    char* Xs = lds;                                  // [DX][kXStage]
    char* As = lds + DX * kXStage;                   // [DA][kAStage]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cgp = wave & 1, rh = wave >> 1;
    const int n = lane & 15, g = lane >> 4;
    const int rb0 = blockIdx.y * 2 * RB;             // first row block of this workgroup
    const bool x_wave = wave < 2;

    // The tile's 128 columns are 16 "units" of 8 pixels = one 16-byte piece of a frame row each (units never span
    // frames; when P % 8 == 4 the last unit of a frame is the row's last 8 pixels, i.e. its first half repeats pixels
    // of the unit before and is not stored).  X DMA: one instruction = 4 channel rows x 16 units, lane l -> row l >> 4,
    // unit l & 15; wave w (0, 1) issues the chunk's row groups 4 w .. 4 w + 3.
    const bool odd_tail = (d.P & 7) != 0;
    const long long ug_dma = (long long)blockIdx.x * 16 + (lane & 15);
    const bool dma_ok = ug_dma < d.nunits;
    long long xoff0;                                  // byte offset of this lane's unit in channel row 0 of its frame
    {
        const long long q = dma_ok ? ug_dma : 0;
        const int f = (int)(q / d.U), j = (int)(q - (long long)f * d.U);
        const int px = (odd_tail && j == d.U - 1) ? d.P - 8 : 8 * j;
        xoff0 = (((long long)f * d.K) * d.P + px) * 2;
    }
    const char* xbase = uniform_bytes(reinterpret_cast<const char*>(X));
    const unsigned xs0 = dma::lds_byte_addr(Xs), as0 = dma::lds_byte_addr(As);
    auto issue_x = [&](int c, int stage) {
        const unsigned dst = xs0 + (unsigned)(stage * kXStage + 4 * wave * kXGroup);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int k = kCh * c + 16 * wave + 4 * i + (lane >> 4);
            k = k < d.K ? k : d.K - 1;                               // rows past K: a valid row, masked at the read
            const int voff = (int)(xoff0 + (long long)k * d.P * 2);
            if (dma_ok) dma::dma16s<false>(xbase, voff, dst + (unsigned)(i * kXGroup));
        }
    };
    // A DMA: the chunk's 2 RB blocks are contiguous in the packed operand; wave w (2, 3) copies blocks w - 2, w, ...
    int nA = 0;                                      // DMAs per chunk of this wave
#pragma unroll
    for (int b = 0; b < RB; ++b)
        if (rb0 + (wave & 1) + 2 * b < d.nrb) ++nA;
    auto issue_a = [&](int c, int stage) {
        const char* src0 = Apk + ((size_t)c * d.nrb + rb0) * 1024;
        const unsigned dst0 = as0 + (unsigned)(stage * kAStage);
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int blk = (wave & 1) + 2 * b;
            if (rb0 + blk < d.nrb && !(d.dbg & 4))
                dma::dma16s<false>(uniform_bytes(src0 + (size_t)blk * 1024), lane * 16, dst0 + (unsigned)(blk * 1024));
        }
    };

    // output geometry of this lane, and its residual values: loaded up front (they land under the K loop; being older
    // than every DMA of the wave they do not disturb the counted waits below)
    const long long ug = (long long)blockIdx.x * 16 + 8 * cgp + (n >> 1);
    bool out_ok = ug < d.nunits;
    size_t at0 = 0;
    {
        const long long q = out_ok ? ug : 0;
        const int f = (int)(q / d.U), j = (int)(q - (long long)f * d.U);
        const bool tail = odd_tail && j == d.U - 1;
        if (tail && !(n & 1)) out_ok = false;                        // the repeated half of a frame's last unit
        const int p = tail ? d.P - 4 : 8 * j + 4 * (n & 1);
        at0 = ((size_t)f * d.M) * d.P + p;
    }
    const int rowb = 16 * (rb0 + rh * RB) + 4 * g;
    uint2 rr[RES ? RB : 1][4];
    if (RES) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rowb + 16 * r + i;
                rr[r][i] = (out_ok && row < d.M) ? *reinterpret_cast<const uint2*>(R + at0 + (size_t)row * d.P) : make_uint2(0u, 0u);
            }
    }

    f32x4 acc[RB][4];                                // initialised in the first iteration (zeros, or the residual)

    const int nch = d.nch;
    if (x_wave) {
#pragma unroll
        for (int j = 0; j < DX - 1; ++j)
            if (j < nch) issue_x(j, j);
    } else {
#pragma unroll
        for (int j = 0; j < DA - 1; ++j)
            if (j < nch) issue_a(j, j);
    }
    int sx = 0, sa = 0;                              // stages of chunk c: c % DX, c % DA
    const bool ragged = (d.K & (kCh - 1)) != 0;
    const char* xrd = Xs + (2 * g) * kXGroup + cgp * 128 + n * 8;          // row 8 g + i: group 2 g + (i >> 2), row i & 3
    const char* ard = As + (rh * RB) * 1024 + lane * 16;

    auto step = [&](int c, auto first) {
        // chunk c has landed when only the younger chunks of this wave's stream (up to D - 2 of them) are outstanding
        const int left = nch - 1 - c;
        if (x_wave) dma::wait_vmcnt(4 * (left < DX - 2 ? left : DX - 2));
        else dma::wait_vmcnt(nA * (left < DA - 2 ? left : DA - 2));
        __syncthreads();
        if (x_wave) {
            if (c + DX - 1 < nch) issue_x(c + DX - 1, sx == 0 ? DX - 1 : sx - 1);      // into the stage of chunk c - 1
        } else {
            if (c + DA - 1 < nch) issue_a(c + DA - 1, sa == 0 ? DA - 1 : sa - 1);
        }

        if (decltype(first)::value) {
            // the accumulators start from the residual: no register is spent on it past this point (the compiler's wait
            // for these loads also drains the DMAs issued so far -- once per workgroup)
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (RES) {
                        acc[r][0][i] = __uint_as_float(rr[r][i].x << 16);
                        acc[r][1][i] = __uint_as_float(rr[r][i].x & 0xffff0000u);
                        acc[r][2][i] = __uint_as_float(rr[r][i].y << 16);
                        acc[r][3][i] = __uint_as_float(rr[r][i].y & 0xffff0000u);
                    } else {
                        acc[r][0][i] = acc[r][1][i] = acc[r][2][i] = acc[r][3][i] = 0.f;
                    }
                }
        }
        const char* xs = xrd + sx * kXStage;
        const char* as = ard + sa * kAStage;
        sx = sx == DX - 1 ? 0 : sx + 1;
        sa = sa == DA - 1 ? 0 : sa + 1;
        uint2 raw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = *reinterpret_cast<const uint2*>(xs + (i >> 2) * kXGroup + (i & 3) * 256);
        if (ragged && c == nch - 1) {
            const int kb = kCh * c + 8 * g;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (kb + i >= d.K) raw[i] = make_uint2(0u, 0u);
        }
        bf16x8 bq[4];
        {
            u32x4 t0, t1, t2, t3;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = __builtin_amdgcn_perm(raw[2 * j + 1].x, raw[2 * j].x, 0x05040100u);
                t1[j] = __builtin_amdgcn_perm(raw[2 * j + 1].x, raw[2 * j].x, 0x07060302u);
                t2[j] = __builtin_amdgcn_perm(raw[2 * j + 1].y, raw[2 * j].y, 0x05040100u);
                t3[j] = __builtin_amdgcn_perm(raw[2 * j + 1].y, raw[2 * j].y, 0x07060302u);
            }
            bq[0] = __builtin_bit_cast(bf16x8, t0);
            bq[1] = __builtin_bit_cast(bf16x8, t1);
            bq[2] = __builtin_bit_cast(bf16x8, t2);
            bq[3] = __builtin_bit_cast(bf16x8, t3);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (rb0 + rh * RB + r < d.nrb && !(d.dbg & 1)) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(as + r * 1024);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[q], acc[r][q], 0, 0, 0);
            }
        }
    };
    step(0, std::true_type{});
#pragma nounroll
    for (int c = 1; c < nch; ++c) step(c, std::false_type{});

    // results: lane (n, g) holds rows 16 rb + 4 g + i, columns 4 n + q of its column group = half a unit: 4 consecutive
    // pixels per row
    if (!out_ok) return;
    if ((d.dbg & 2) && acc[0][0][0] != 12345.f) return;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rowb + 16 * r + i;
            if (row >= d.M) continue;
            const float o0 = acc[r][0][i], o1 = acc[r][1][i], o2 = acc[r][2][i], o3 = acc[r][3][i];
            *reinterpret_cast<uint2*>(Y + at0 + (size_t)row * d.P) =
                make_uint2(bf16_bits(o0) | (bf16_bits(o1) << 16), bf16_bits(o2) | (bf16_bits(o3) << 16));
        }
    }
}

inline int rows_per_wave(int nrb) { return nrb > 10 ? 9 : (nrb > 6 ? 5 : 3); }

template <int RB, bool RES, int DX, int DA>
int launch_gemm(const char* Apk, const __hip_bfloat16* X, const __hip_bfloat16* R, __hip_bfloat16* Y, const Dims& d,
                hipStream_t stream) {
    constexpr size_t lds = (size_t)DX * kXStage + (size_t)DA * 2 * RB * 1024;
    static bool raised = false;                      // > 64 KB of dynamic LDS needs the attribute, once per instance
    if (lds > 65536 && !raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pw16_gemm<RB, RES, DX, DA>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return RK_ERR_LAUNCH;
        raised = true;
    }
    const dim3 grid((unsigned)((d.nunits + 15) / 16), (unsigned)((d.nrb + 2 * RB - 1) / (2 * RB)));
    hipLaunchKernelGGL((k_pw16_gemm<RB, RES, DX, DA>), grid, dim3(kBlock), lds, stream, Apk, X, R, Y, d);
    return launch_status();
}

}  // namespace pw16
}  // namespace rk

using namespace rk;
using namespace rk::pw16;

extern "C" {

// bytes of the packed (bf16, MFMA fragment order) form of an operand with `rows` GEMM rows and reduction depth `depth`
size_t rk_pw_packed_bytes(int rows, int depth) {
    if (rows <= 0 || depth <= 0) return 0;
    return (size_t)((rows + 15) / 16) * ((depth + kCh - 1) / kCh) * 1024;
}

// Packs the fp32 weight W [Cout][Cin] of a 1x1 convolution for rk_pw_gemm_packed_bf16: `fwd` (rows = Cout, depth = Cin:
// the forward operand) and / or `bwd` (rows = Cin, depth = Cout: W^T, the d(input) operand); either may be NULL.
int rk_pw_pack_bf16(const float* W, int Cout, int Cin, void* fwd, void* bwd, rk_stream_t stream_) {
    if (!W || (!fwd && !bwd)) return RK_ERR_NULL_POINTER;
    if (Cout <= 0 || Cin <= 0) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)fwd & 15) || ((uintptr_t)bwd & 15)) return RK_ERR_BAD_DIMS;
    hipStream_t stream = (hipStream_t)stream_;
    if (fwd) {
        const int nrb = (Cout + 15) / 16, nch = (Cin + kCh - 1) / kCh, units = nrb * nch * 64;
        hipLaunchKernelGGL(k_pw16_pack, dim3((units + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, W, Cout, Cin, 1, nrb, nch,
                           (uint4*)fwd);
    }
    if (bwd) {                                                     // rows = Cin, depth = Cout, element (m, k) = W[k][m]
        const int nrb = (Cin + 15) / 16, nch = (Cout + kCh - 1) / kCh, units = nrb * nch * 64;
        hipLaunchKernelGGL(k_pw16_pack, dim3((units + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, W, Cin, Cout, 0, nrb, nch,
                           (uint4*)bwd);
    }
    return launch_status();
}

// Y[f] = A X[f] (+ R[f]) with A packed by rk_pw_pack_bf16 (M rows, depth K).  X [F, K, P], Y / R [F, M, P] bf16,
// P % 4 == 0; R may be NULL and may be Y itself.
int rk_pw_gemm_packed_bf16(const void* Apk, const void* X_, const void* R_, void* Y_, int F, int K, int M, int P,
                           rk_stream_t stream_) {
    const __hip_bfloat16* X = (const __hip_bfloat16*)X_;
    const __hip_bfloat16* R = (const __hip_bfloat16*)R_;
    __hip_bfloat16* Y = (__hip_bfloat16*)Y_;
    if (!Apk || !X || !Y) return RK_ERR_NULL_POINTER;
    if (F <= 0 || K <= 0 || M <= 0 || P < 8 || P % 4 != 0) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)Apk & 15) || ((uintptr_t)X & 7) || ((uintptr_t)Y & 7) || (R && ((uintptr_t)R & 7))) return RK_ERR_BAD_DIMS;
    if ((long long)F * K * P * 2 >= (1ll << 31)) return RK_ERR_BAD_DIMS;             // 32-bit byte offsets in the DMA
    Dims d;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P;
    d.nrb = (M + 15) / 16; d.nch = (K + kCh - 1) / kCh;
    d.U = (P + 7) / 8; d.nunits = (long long)F * d.U;
    { const char* e = getenv("RK_PW16_DBG"); d.dbg = e ? atoi(e) : 0; }
    hipStream_t stream = (hipStream_t)stream_;
    const char* A = (const char*)Apk;
    const int rb = rows_per_wave(d.nrb);
#define RK_GO(RBV, DXV, DAV) (R ? launch_gemm<RBV, true, DXV, DAV>(A, X, R, Y, d, stream) : launch_gemm<RBV, false, DXV, DAV>(A, X, R, Y, d, stream))
    const bool deep = (d.dbg & 8) != 0;
    if (rb == 9) return deep ? RK_GO(9, 4, 2) : RK_GO(9, 3, 2);
    if (rb == 5) return deep ? RK_GO(5, 4, 3) : RK_GO(5, 3, 2);
    return deep ? RK_GO(3, 4, 3) : RK_GO(3, 3, 2);
#undef RK_GO
}

}  // extern "C"
