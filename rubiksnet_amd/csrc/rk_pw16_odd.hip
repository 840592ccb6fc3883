// rk_pw16_odd.hip -- 1x1 convolutions of bf16 activations on planes whose rows have no 16-byte unit: P = H * W with
// P % 4 != 0, P <= 64 -- the 7x7 planes of layer4 (rubiksnet/backbone.py:44-45 on [NT, 576, 7, 7]), which rk_pw16.hip cannot
// take (its columns are 8-pixel pieces of a frame row) and which ran on MIOpen between NCHW <-> NHWC transposes of both tensors
// (58-118 us per call at [256, 576 -> 576, 7, 7]; a batched rocBLAS GEMM on the NCHW tensor measured the same,
// tools/odd_gemm_probe.py) although the layer is 8.3 GFLOP and 29 MB.
//
// What IS contiguous and aligned here is a whole FRAME: X[f] = [K][P] bf16 is K * P * 2 bytes (K % 32 == 0: whole 16-byte
// pieces).  So a workgroup owns one frame: its 64 columns are the frame's P pixels (+ padding that is computed and dropped),
// and X[f] streams through LDS in its memory layout, a 32-channel chunk = 64 P contiguous bytes at a time.
//   * forward / d(input) (k_pw16_odd_gemm): Y[f] = A X[f] (+ R[f]) with A packed by rk_pw_pack_bf16 (the same image the
//     other bf16 GEMM takes).  The workgroup's (up to 40) row blocks are dealt out evenly to its 8 waves; a wave owns its
//     <= RB row blocks x all 4 column blocks.  B fragments come from LDS as 8 two-byte reads per fragment (k-major gathers from the
//     channel-major chunk: 32 ds_read_u16 per chunk against 4 RB MFMAs -- the operand with 576 rows pays for them); A
//     fragments are 16-byte loads straight from the packed image (L2), three chunks ahead in registers.  Results are staged
//     in LDS as fp32 [rows][P] -- the frame's output block in memory order -- and leave as 16-byte pieces, the residual added
//     in fp32 before the one rounding to bf16.
//   * d(weight) (k_pw16_odd_wgrad): dW[m][k] = sum_f sum_p dY[f][m][p] X[f][k][p]; the reduction index p is contiguous in both
//     operands, so both fragments are 8 consecutive bf16 of a row, read from LDS two bytes at a time (rows are 2-byte
//     aligned) with the pixels past P forced to zero.  A workgroup owns an output tile of up to 160 x 160 (2 x 2 waves of up to
//     5 x 5 blocks) and a range of frames; partials ws[split][M][K] are summed by rk_pw16.hip's k_pw16_reduce order
//     (k_pw16_odd_reduce: the same fixed order).
// Arithmetic as rk_pw16.hip: bf16 operands, fp32 accumulation (v_mfma_f32_16x16x32_bf16), one rounding of the result.
#include <type_traits>
#include "rk_common.hpp"
#include "rk_reduce.hpp"

namespace rk {
namespace pw16odd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCh = 32;                    // channels per chunk = K of one MFMA
constexpr int kPad = 128;                  // bytes past a staged block that the padded columns' reads may touch

struct ODims {
    int F, K, M, P;
    int nrb, nch;                          // 16-row blocks of the packed operand, 32-channel chunks
    int rbw;                               // row blocks per workgroup (blockIdx.y selects the range)
};

__device__ __forceinline__ unsigned bf16_bits(float f) {
    return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(f));
}
__device__ __forceinline__ unsigned lds_u16(const char* p) { return (unsigned)*reinterpret_cast<const unsigned short*>(p); }

// 8 bf16 at p, p + stride, ..., p + 7 stride (bytes) -> one MFMA operand
__device__ __forceinline__ bf16x8 gather8(const char* p, int stride) {
    u32x4 t;
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = lds_u16(p + (2 * j) * stride) | (lds_u16(p + (2 * j + 1) * stride) << 16);
    return __builtin_bit_cast(bf16x8, t);
}

// ---------------------------------------------------------------------------------------------
template <int RB, bool RES, int NW>
__global__ __launch_bounds__(64 * NW) void k_pw16_odd_gemm(const char* __restrict__ Apk, const __hip_bfloat16* __restrict__ X,
                                                          const __hip_bfloat16* R, __hip_bfloat16* Y, ODims d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = lane & 15, g = lane >> 4;
    const int f = blockIdx.x;
    // the workgroup's row blocks [rb0, rb0 + nb) dealt out evenly: wave w owns lb0 .. lb0 + cnt - 1 (cnt <= RB; slots past cnt
    // compute copies of the wave's last block and are not stored)
    const int rb0 = blockIdx.y * d.rbw;
    const int nb = d.nrb - rb0 < d.rbw ? d.nrb - rb0 : d.rbw;
    const int lb0 = (wave * nb) / NW, cnt = ((wave + 1) * nb) / NW - lb0;
    const int xstage = kCh * d.P * 2 + kPad;                          // a chunk in its memory layout + the padded columns' slack
    char* xs = lds;                                                  // [3][xstage]
    float* ys = reinterpret_cast<float*>(lds + 3 * xstage);          // [16 rbw][P] fp32

    // X chunk c of the frame: 4 P pieces of 16 bytes, piece t by thread t
    const char* xf = reinterpret_cast<const char*>(X) + (size_t)f * d.K * d.P * 2;
    const int chunk_bytes = kCh * d.P * 2;
    const bool x_on = (int)threadIdx.x < 4 * d.P;                  // (4 P <= 256 threads)
    auto load_x = [&](int c) -> u32x4 {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (x_on && c < d.nch) v = *reinterpret_cast<const u32x4*>(xf + (size_t)c * chunk_bytes + 16 * threadIdx.x);
        return v;
    };
    auto put_x = [&](int c, const u32x4& v) {
        if (x_on) *reinterpret_cast<u32x4*>(xs + (c % 3) * xstage + 16 * threadIdx.x) = v;
    };
    // A: row blocks past the operand read its last block (copies; not stored)
    int arb[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) arb[r] = rb0 + lb0 + (r < cnt ? r : (cnt > 0 ? cnt - 1 : 0));
    auto load_a = [&](int c, bf16x8 (&a)[RB]) {
        if (c >= d.nch) return;
        const char* base = Apk + ((size_t)c * d.nrb * 64 + lane) * 16;
#pragma unroll
        for (int r = 0; r < RB; ++r) a[r] = *reinterpret_cast<const bf16x8*>(base + (size_t)arb[r] * 1024);
    };

    f32x4 acc[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 a0[RB], a1[RB], a2[RB];
    u32x4 x0 = load_x(0), x1 = load_x(1), x2 = load_x(2);
    load_a(0, a0); load_a(1, a1); load_a(2, a2);
    put_x(0, x0);
    __syncthreads();

    // step c: stage chunk c + 1, request chunk c + 3 (X into the registers chunk c left, A after the MFMAs that used it)
    auto step = [&](int c, bf16x8 (&a)[RB], u32x4& xnext, u32x4& xfree) {
        if (c + 1 < d.nch) put_x(c + 1, xnext);
        xfree = load_x(c + 3);
        const char* st = xs + (c % 3) * xstage + ((8 * g) * d.P + n) * 2;
        bf16x8 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = gather8(st + 32 * q, 2 * d.P);
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r], b[q], acc[r][q], 0, 0, 0);
        load_a(c + 3, a);
        __syncthreads();
    };
#pragma nounroll
    for (int c = 0; c < d.nch; c += 3) {
        step(c, a0, x1, x0);                                         // (x0 held chunk c: staged one step earlier)
        if (c + 1 < d.nch) step(c + 1, a1, x2, x1);
        if (c + 2 < d.nch) step(c + 2, a2, x0, x2);
    }

    // results -> ys[row][p] (lane (n, g) of tile (r, q): rows 16 (lb0 + r) + 4 g + i, column 16 q + n)
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (r >= cnt) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 16 * q + n;
            if (col < d.P) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ys[(16 * (lb0 + r) + 4 * g + i) * d.P + col] = acc[r][q][i];
            }
        }
    }
    __syncthreads();
    // the frame's output rows [row0, row0 + rows) are one contiguous block: 16-byte pieces of 8 values
    const int row0 = 16 * rb0;
    int rows = d.M - row0;
    rows = rows < 16 * d.rbw ? rows : 16 * d.rbw;
    const int pieces = rows * d.P / 8;                               // (M % 8 == 0: whole pieces)
    const size_t ybase = ((size_t)f * d.M + row0) * d.P;
    for (int j = threadIdx.x; j < pieces; j += 64 * NW) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(ys + 8 * j), hi = *reinterpret_cast<const f32x4*>(ys + 8 * j + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (RES) {
            const u32x4 rr = *reinterpret_cast<const u32x4*>(R + ybase + 8 * j);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[2 * k] += __uint_as_float(rr[k] << 16); v[2 * k + 1] += __uint_as_float(rr[k] & 0xffff0000u); }
        }
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = bf16_bits(v[2 * k]) | (bf16_bits(v[2 * k + 1]) << 16);
        *reinterpret_cast<u32x4*>(Y + ybase + 8 * j) = o;
    }
}

// row blocks per workgroup: all of them up to 40 (8 waves x 5), else equal ranges of at most 40
inline int blocks_per_group(int nrb) { const int groups = (nrb + 39) / 40; return (nrb + groups - 1) / groups; }
inline size_t gemm_lds(int rbw, int P) { return 3 * (size_t)(kCh * P * 2 + kPad) + (size_t)16 * rbw * P * 4; }

template <int RB, bool RES, int NW>
int launch_gemm(const char* Apk, const __hip_bfloat16* X, const __hip_bfloat16* R, __hip_bfloat16* Y, const ODims& d,
                hipStream_t stream) {
    const size_t lds = gemm_lds(d.rbw, d.P);
    if (lds > 160 * 1024) return RK_ERR_UNSUPPORTED;
    static DynLdsRaised raised;                      // > 64 KB of dynamic LDS needs the attribute: per instance and device
    if (const int rc = raise_dynamic_lds(reinterpret_cast<const void*>(&k_pw16_odd_gemm<RB, RES, NW>), lds, raised)) return rc;
    const dim3 grid((unsigned)d.F, (unsigned)((d.nrb + d.rbw - 1) / d.rbw));
    hipLaunchKernelGGL((k_pw16_odd_gemm<RB, RES, NW>), grid, dim3(64 * NW), lds, stream, Apk, X, R, Y, d);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------
struct OWDims {
    int F, K, M, P;
    int S, fps;                  // splits of the frame range, frames per split
    int tilesM, tilesK;          // output tiles
    int tbM, tbK;                // 16-row blocks per tile
    int mbT, kbT;                // total blocks: ceil(M / 16), ceil(K / 16)
};

// 8 consecutive bf16 of a row at p (2-byte aligned), elements at or past `valid` zero
__device__ __forceinline__ bf16x8 row8(const char* p, int valid) {
    u32x4 t;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned lo = 2 * j < valid ? lds_u16(p + 4 * j) : 0u, hi = 2 * j + 1 < valid ? lds_u16(p + 4 * j + 2) : 0u;
        t[j] = lo | (hi << 16);
    }
    return __builtin_bit_cast(bf16x8, t);
}

template <int BM, int BK, int NL>
__global__ __launch_bounds__(kBlock, 2) void k_pw16_odd_wgrad(const __hip_bfloat16* __restrict__ dY,
                                                              const __hip_bfloat16* __restrict__ X, float* __restrict__ ws,
                                                              OWDims d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm = wave & 1, wk = wave >> 1;
    const int m16 = lane & 15, g = lane >> 4;
    const int T = d.tilesM * d.tilesK;
    const int tile = blockIdx.x % T, split = blockIdx.x / T;
    const int tm = tile / d.tilesK, tk = tile - tm * d.tilesK;
    const int f_lo = split * d.fps;
    int f_hi = f_lo + d.fps;
    f_hi = f_hi < d.F ? f_hi : d.F;
    // the tile's rows of dY[f] / X[f]: one contiguous, 16-byte aligned block each (row0 % 16 == 0; the last tile may be short)
    const int rowM0 = 16 * d.tbM * tm, rowK0 = 16 * d.tbK * tk;
    int rowsM = d.M - rowM0, rowsK = d.K - rowK0;
    rowsM = rowsM < 16 * d.tbM ? rowsM : 16 * d.tbM;
    rowsK = rowsK < 16 * d.tbK ? rowsK : 16 * d.tbK;
    const int bytesM = (rowsM * d.P * 2 + 15) & ~15, bytesK = (rowsK * d.P * 2 + 15) & ~15;   // (whole pieces: M, K % 8 == 0)
    const int offK = 16 * d.tbM * d.P * 2 + kPad;                    // X block behind the dY block
    const int stage = offK + 16 * d.tbK * d.P * 2 + kPad;
    const int piecesM = bytesM / 16, pieces = piecesM + bytesK / 16;
    // NL: pieces per thread and frame (at most 2 x 160 rows x 64 px x 2 B / 16 / 256 = 10; 7 at 2 x 144 rows x 49 px)
    auto load_frame = [&](int f, u32x4 (&v)[NL]) {
        const char* gm = reinterpret_cast<const char*>(dY) + ((size_t)f * d.M + rowM0) * d.P * 2;
        const char* gk = reinterpret_cast<const char*>(X) + ((size_t)f * d.K + rowK0) * d.P * 2;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int t = (int)threadIdx.x + kBlock * j;
            v[j] = u32x4{0u, 0u, 0u, 0u};
            if (f < f_hi && t < pieces) v[j] = *reinterpret_cast<const u32x4*>(t < piecesM ? gm + 16 * t : gk + 16 * (t - piecesM));
        }
    };
    auto put_frame = [&](int s, const u32x4 (&v)[NL]) {
        char* base = lds + s * stage;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int t = (int)threadIdx.x + kBlock * j;
            if (t < pieces) *reinterpret_cast<u32x4*>(t < piecesM ? base + 16 * t : base + offK + 16 * (t - piecesM)) = v[j];
        }
    };

    f32x4 acc[BM][BK];
#pragma unroll
    for (int a = 0; a < BM; ++a)
#pragma unroll
        for (int b = 0; b < BK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's blocks of the tile (clamped: blocks past the tile compute copies that are not stored)
    int ra[BM], rbk[BK];
#pragma unroll
    for (int a = 0; a < BM; ++a) { const int blk = BM * wm + a; ra[a] = (blk < d.tbM ? blk : d.tbM - 1) * 16 + m16; }
#pragma unroll
    for (int b = 0; b < BK; ++b) { const int blk = BK * wk + b; rbk[b] = (blk < d.tbK ? blk : d.tbK - 1) * 16 + m16; }
    const int ksteps = (d.P + 31) / 32;

    u32x4 v[NL];
    load_frame(f_lo, v);
    put_frame(0, v);
    load_frame(f_lo + 1, v);
    __syncthreads();
    for (int f = f_lo; f < f_hi; ++f) {
        const int s = (f - f_lo) & 1;
        if (f + 1 < f_hi) put_frame(s ^ 1, v);                       // (its last readers passed the barrier of frame f - 1)
        load_frame(f + 2, v);
        const char* bm = lds + s * stage;
        const char* bk = bm + offK;
        for (int h = 0; h < ksteps; ++h) {
            const int p0 = 32 * h + 8 * g, valid = d.P - p0;         // this lane's 8 pixels; those at or past P are zero
            bf16x8 fa[BM];
#pragma unroll
            for (int a = 0; a < BM; ++a) fa[a] = row8(bm + (ra[a] * d.P + p0) * 2, valid);
#pragma unroll
            for (int b = 0; b < BK; ++b) {                           // (one X fragment live at a time: the kernel sits at its register cap)
                const bf16x8 fb = row8(bk + (rbk[b] * d.P + p0) * 2, valid);
#pragma unroll
                for (int a = 0; a < BM; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // partial dW tile -> ws[split][M][K]: lane (n = m16, g) of block (a, b) holds rows 4 g + i, column n
    float* out = ws + (size_t)split * d.M * d.K;
#pragma unroll
    for (int a = 0; a < BM; ++a) {
        if (BM * wm + a >= d.tbM) continue;
#pragma unroll
        for (int b = 0; b < BK; ++b) {
            if (BK * wk + b >= d.tbK) continue;
            const int col = rowK0 + 16 * (BK * wk + b) + m16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rowM0 + 16 * (BM * wm + a) + 4 * g + i;
                if (row < d.M && col < d.K) out[(size_t)row * d.K + col] = acc[a][b][i];
            }
        }
    }
}

// out[i] = sum over the S partial matrices in a fixed order (rk_pw16.hip's k_pw16_reduce: 4 slices of the split range per
// output, each front to back, then the slice sums in slice order)
__global__ __launch_bounds__(kBlock) void k_pw16_odd_reduce(const float* __restrict__ in, float* __restrict__ out, int MK, int S) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int per = (S + 3) / 4, c0 = slice * per, c1 = (c0 + per) < S ? (c0 + per) : S;
    float acc = 0.f;
    if (i < MK) {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) acc += in[(size_t)c * MK + i];
    }
    part[slice][lane] = acc;
    __syncthreads();
    if (slice == 0 && i < MK) out[i] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

inline int make_owdims(OWDims& d, int F, int K, int M, int P) {
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P > 64 || K % 8 || M % 8) return RK_ERR_BAD_DIMS;
    d.F = F; d.K = K; d.M = M; d.P = P;
    d.mbT = (M + 15) / 16; d.kbT = (K + 15) / 16;
    d.tilesM = (d.mbT + 9) / 10; d.tilesK = (d.kbT + 9) / 10;
    d.tbM = (d.mbT + d.tilesM - 1) / d.tilesM; d.tbK = (d.kbT + d.tilesK - 1) / d.tilesK;
    const int T = d.tilesM * d.tilesK;
    // about two workgroups per CU, at least 4 frames each, partial matrices of at most 48 MB in all
    long long S = 512 / T;
    const long long by_frames = F / 4, by_bytes = (48ll << 20) / ((long long)M * K * 4);
    S = S < by_frames ? S : by_frames;
    S = S < by_bytes ? S : by_bytes;
    S = S < 1 ? 1 : S;
    d.fps = (int)((F + S - 1) / S);
    d.S = (F + d.fps - 1) / d.fps;
    return RK_OK;
}
inline size_t wgrad_lds(const OWDims& d) { return 2 * (size_t)(16 * (d.tbM + d.tbK) * d.P * 2 + 2 * kPad); }

template <int BM, int BK, int NL>
int launch_wgrad_nl(const __hip_bfloat16* dY, const __hip_bfloat16* X, float* ws, const OWDims& d, hipStream_t stream) {
    const size_t lds = wgrad_lds(d);
    if (lds > 80 * 1024) return RK_ERR_UNSUPPORTED;
    static DynLdsRaised raised;
    if (const int rc = raise_dynamic_lds(reinterpret_cast<const void*>(&k_pw16_odd_wgrad<BM, BK, NL>), lds, raised)) return rc;
    hipLaunchKernelGGL((k_pw16_odd_wgrad<BM, BK, NL>), dim3((unsigned)(d.tilesM * d.tilesK * d.S)), dim3(kBlock), lds, stream, dY,
                       X, ws, d);
    return launch_status();
}
template <int BM, int BK>
int launch_wgrad(const __hip_bfloat16* dY, const __hip_bfloat16* X, float* ws, const OWDims& d, hipStream_t stream) {
    const int pieces = 16 * (d.tbM + d.tbK) * d.P * 2 / 16;           // per frame, at most (full tiles)
    if (pieces <= 7 * kBlock) return launch_wgrad_nl<BM, BK, 7>(dY, X, ws, d, stream);
    return launch_wgrad_nl<BM, BK, 10>(dY, X, ws, d, stream);
}

}  // namespace pw16odd
}  // namespace rk

using namespace rk;
using namespace rk::pw16odd;

extern "C" {

// 1 when the odd-plane kernels take [F, K -> M, P] (P <= 64 with no 16-byte unit in a row is their reason to exist; any P <= 64
// with K % 32 == 0, M % 8 == 0 works)
int rk_pw_odd16_supported(int F, int K, int M, int P) {
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P > 64 || K % kCh || M % 8) return 0;
    if ((long long)F * (K > M ? K : M) * P * 2 >= (1ll << 31)) return 0;
    return gemm_lds(blocks_per_group((M + 15) / 16), P) <= 160 * 1024 ? 1 : 0;
}

// Y[f] = A X[f] (+ R[f]) with A packed by rk_pw_pack_bf16 (M rows, depth K).  X [F, K, P], Y / R [F, M, P] bf16, 16-byte aligned;
// R may be NULL and may be Y itself.
int rk_pw_gemm_packed_odd_bf16(const void* Apk, const void* X_, const void* R_, void* Y_, int F, int K, int M, int P,
                               rk_stream_t stream_) {
    const __hip_bfloat16* X = (const __hip_bfloat16*)X_;
    const __hip_bfloat16* R = (const __hip_bfloat16*)R_;
    __hip_bfloat16* Y = (__hip_bfloat16*)Y_;
    if (!Apk || !X || !Y) return RK_ERR_NULL_POINTER;
    if (!rk_pw_odd16_supported(F, K, M, P)) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)Apk & 15) || ((uintptr_t)X & 15) || ((uintptr_t)Y & 15) || ((uintptr_t)R & 15)) return RK_ERR_BAD_DIMS;
    ODims d;
    d.F = F; d.K = K; d.M = M; d.P = P;
    d.nrb = (M + 15) / 16; d.nch = K / kCh;
    d.rbw = blocks_per_group(d.nrb);
    hipStream_t stream = (hipStream_t)stream_;
    const char* A = (const char*)Apk;
    // 8 waves: two per SIMD, one's LDS gathers and barrier waits under the other's MFMAs (4 waves x 9 blocks measured 23.7 us at
    // [256, 576 -> 576, 49])
    if (d.rbw > 24) return R ? launch_gemm<5, true, 8>(A, X, R, Y, d, stream) : launch_gemm<5, false, 8>(A, X, R, Y, d, stream);
    return R ? launch_gemm<3, true, 8>(A, X, R, Y, d, stream) : launch_gemm<3, false, 8>(A, X, R, Y, d, stream);
}

// d(weight) [M][K] (fp32) = sum_f dY[f] X[f]^T, dY [F, M, P], X [F, K, P] bf16 (16-byte aligned, K % 8 == M % 8 == 0, P <= 64);
// ws of rk_pw_wgrad_odd16_workspace_bytes() bytes holds the per-split partial matrices.
size_t rk_pw_wgrad_odd16_workspace_bytes(int F, int K, int M, int P) {
    OWDims d;
    if (make_owdims(d, F, K, M, P)) return 0;
    return (size_t)d.S * M * K * sizeof(float);
}
int rk_pw_wgrad_odd16_bf16(const void* dY_, const void* X_, float* dW, int F, int K, int M, int P, void* ws, size_t ws_bytes,
                           rk_stream_t stream_) {
    const __hip_bfloat16* dY = (const __hip_bfloat16*)dY_;
    const __hip_bfloat16* X = (const __hip_bfloat16*)X_;
    if (!dY || !X || !dW) return RK_ERR_NULL_POINTER;
    OWDims d;
    if (int rc = make_owdims(d, F, K, M, P)) return rc;
    if ((long long)F * (K > M ? K : M) * P * 2 >= (1ll << 31)) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)dY & 15) || ((uintptr_t)X & 15)) return RK_ERR_BAD_DIMS;
    if (!ws || ws_bytes < (size_t)d.S * M * K * sizeof(float)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int bm = (d.tbM + 1) / 2 > 3 ? 5 : 3, bk = (d.tbK + 1) / 2 > 3 ? 5 : 3;
    int rc;
    if (bm == 5 && bk == 5) rc = launch_wgrad<5, 5>(dY, X, (float*)ws, d, stream);
    else if (bm == 5) rc = launch_wgrad<5, 3>(dY, X, (float*)ws, d, stream);
    else if (bk == 5) rc = launch_wgrad<3, 5>(dY, X, (float*)ws, d, stream);
    else rc = launch_wgrad<3, 3>(dY, X, (float*)ws, d, stream);
    if (rc) return rc;
    const int MK = M * K;
    if (!launch_reduce_partials4((const float*)ws, dW, MK, d.S, stream))
        hipLaunchKernelGGL(k_pw16_odd_reduce, dim3((MK + 63) / 64), dim3(kBlock), 0, stream, (const float*)ws, dW, MK, d.S);
    return launch_status();
}

}  // extern "C"
