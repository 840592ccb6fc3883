// rk_pw16.hip -- the 1x1 convolutions of the bf16 (autocast) train step, second generation (the unfused
// half; rubiksnet/backbone.py:44-45, :87-104 under torch.autocast(bfloat16)).  At 2 bytes per element these GEMMs are
// HBM-bound by a wide margin (Large-AQ's [256, 288, 14, 14] layer: 58 MB of operands against 8.3 GFLOP = 3.3 us of
// bf16 MFMA), so the kernels are built around "every activation byte crosses HBM once":
//
//   * forward / d(input) (k_pw16_gemm): a workgroup owns ALL rows of a 128-pixel column tile (up to 288 rows per
//     workgroup: 144 accumulator registers per lane, 2 workgroups per CU), so X is read once (the first-generation
//     kernel of rk_pw.hip re-read it once per 64-row tile, 5x at 288 rows).  X is streamed by LDS-DMA in its memory
//     layout ([32 channels][128 pixels] per chunk, 16-byte units that never span frames), 3 chunks deep; the
//     channel-major -> k-major transposition a bf16 MFMA operand needs happens in REGISTERS
//     (8 ds_read_b64 + 16 v_perm_b32 per wave and chunk give the 4 B-fragments of a 64-pixel column group; no 2-byte
//     LDS writes).  The small operand is pre-packed (k_pw16_pack, one launch per forward) into bf16 MFMA A-fragment
//     order and moved 16 B per lane, one chunk ahead.  v_mfma_f32_16x16x32_bf16: 16-row blocks fit 72 / 144 / 288
//     rows with little padding.
//   * d(weight) (k_pw16_wgrad): see below.
//
// Arithmetic: bf16 operands (the weight rounded to bf16 as autocast would), fp32 accumulation, one rounding of the
// result (+ residual) to bf16.
#include <type_traits>
#include "rk_common.hpp"
#include "rk_dma.hpp"
#include "rk_reduce.hpp"

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
    int nrb, nch;                          // 16-row blocks of the packed operand, 32-channel chunks
    int U;                                 // 16-byte units (8 pixels) per frame row: ceil(P / 8)
    long long nunits;                      // F * U
};

__device__ __forceinline__ unsigned bf16_bits(float f) {
    return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(f));
}

// ---- packing: fp32 [M][K] (mk != 0) or [K][M] -> bf16 fragments [chunk][row block][lane][8]; lane (m = l & 15,
// g = l >> 4) of block rb, chunk c holds rows 16 rb + m, reduction indices 32 c + 8 g .. + 7 (zeros outside) ----
__device__ __forceinline__ void pack_unit(const float* __restrict__ A, int M, int K, int mk, int nrb, int u,
                                          uint4* __restrict__ out) {
    const int lane = u & 63, rb = (u >> 6) % nrb, c = (u >> 6) / nrb;
    const int m = 16 * rb + (lane & 15), k0 = kCh * c + 8 * (lane >> 4);
    unsigned h[8];
    if (mk && (K & 3) == 0 && m < M && k0 + 8 <= K) {              // 8 consecutive weights of a row: two 16-byte loads
        const float4 q0 = *reinterpret_cast<const float4*>(A + (size_t)m * K + k0);
        const float4 q1 = *reinterpret_cast<const float4*>(A + (size_t)m * K + k0 + 4);
        h[0] = bf16_bits(q0.x); h[1] = bf16_bits(q0.y); h[2] = bf16_bits(q0.z); h[3] = bf16_bits(q0.w);
        h[4] = bf16_bits(q1.x); h[5] = bf16_bits(q1.y); h[6] = bf16_bits(q1.z); h[7] = bf16_bits(q1.w);
    } else {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                              // all 8 requested before the first conversion
            const int k = k0 + j;
            const bool ok = m < M && k < K;
            v[j] = ok ? A[mk ? (size_t)m * K + k : (size_t)k * M + m] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = bf16_bits(v[j]);
    }
    out[u] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
}
// W [Cout][Cin] -> fwd (rows Cout, depth Cin) in the first nf 16-byte units of the launch, bwd (rows Cin, depth Cout: W^T)
// in the rest; either output may be absent (its unit count 0)
__global__ __launch_bounds__(kBlock) void k_pw16_pack(const float* __restrict__ W, int Cout, int Cin, int nf, int nb,
                                                      uint4* __restrict__ fwd, uint4* __restrict__ bwd) {
    const int u = blockIdx.x * kBlock + threadIdx.x;
    if (u < nf) pack_unit(W, Cout, Cin, 1, (Cout + 15) / 16, u, fwd);
    else if (u < nf + nb) pack_unit(W, Cin, Cout, 0, (Cin + 15) / 16, u - nf, bwd);
}

// every 1x1 weight of a network in ONE launch (pointwise.prepacked: once per train step instead of once per forward of every
// layer -- k_pw16_pack x 100 was 0.5 ms of the Large-AQ step): blockIdx.y = the weight, outputs at offsets of one buffer
struct PackJob { const float* W; long long fwd_off, bwd_off; int Cout, Cin, nf, nb; };
__global__ __launch_bounds__(kBlock) void k_pw16_pack_many(const PackJob* __restrict__ jobs, char* __restrict__ base) {
    const PackJob j = jobs[blockIdx.y];
    const int u = blockIdx.x * kBlock + threadIdx.x;
    if (u < j.nf) pack_unit(j.W, j.Cout, j.Cin, 1, (j.Cout + 15) / 16, u, reinterpret_cast<uint4*>(base + j.fwd_off));
    else if (u < j.nf + j.nb) pack_unit(j.W, j.Cin, j.Cout, 0, (j.Cin + 15) / 16, u - j.nf, reinterpret_cast<uint4*>(base + j.bwd_off));
}

__device__ __forceinline__ const char* uniform_bytes(const char* p) {
    return reinterpret_cast<const char*>(dma::uniform_ptr(reinterpret_cast<const float*>(p)));
}

// Y[f] = A X[f] (+ R[f]).  RB: 16-row blocks per wave; the workgroup's 4 waves are 2 column groups x 2 row halves, so
// a workgroup covers 32 RB rows (blockIdx.y selects the row range when M is larger).  DX: stages of the X ring.
// Waves 0-1 issue the X DMAs (counted vmcnt), waves 2-3 move the A chunks (ordinary loads, compiler-counted): vmcnt is
// per wave and in order, so a wave doing both could not wait for its A chunk without draining the X prefetch.
// STATS (training, round 5): the tile statistics of Y for the BatchNorm that consumes it -- stats[row][2 blockIdx.x + column
// group] = (pivot = 0, sum y, sum y^2, columns) over the wave's 64 columns, of the values AS STORED (rounded
// to bf16), the record rk_bn_finish_tiles_f32 finishes (rk_bn.hip) -- so that the consumer's statistics pass over Y
// (k_bn_stats: 98 launches, 1.5 ms of a Large-AQ train step) never runs.
template <int RB, bool RES, int DX, bool STATS = false>
__global__ __launch_bounds__(kBlock, 2) void k_pw16_gemm(const char* __restrict__ Apk, const __hip_bfloat16* __restrict__ X,
                                                         const __hip_bfloat16* R, __hip_bfloat16* Y, Dims d,
                                                         float4* __restrict__ stats = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int kAStage = 2 * RB * 1024;
    char* Xs = lds;                                  // [DX][kXStage]
    char* As = lds + DX * kXStage;                   // [2][kAStage]
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
    const unsigned xs0 = dma::lds_byte_addr(Xs);
    // (per chunk the 4 byte offsets of a lane advance by 32 rows: kept in registers and bumped, the row clamp only where
    // the last chunk is ragged -- recomputing them, with the 64-bit products, was 370 cycles per chunk on the critical path)
    int xvo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int k = 16 * wave + 4 * i + (lane >> 4);
        k = k < d.K ? k : d.K - 1;
        xvo[i] = (int)(xoff0 + (long long)k * d.P * 2);
    }
    const int xstep = kCh * d.P * 2;                                // bytes per 32 channel rows
    const bool k_ragged = (d.K & (kCh - 1)) != 0;
    // RES: the residual rides through the SAME ring as nchR more chunks after X's -- R [F, M, P] has X's layout, and
    // Y = [A | I] [X ; R]: chunk nch + j holds R's rows 16 rb0 + 32 j .. + 31, and its A stage is two identity blocks among
    // zeros, written to LDS by the A waves from registers instead of from the packed operand -- the step below does not
    // know the difference (its MFMAs on zero blocks are free: the pipe is idle 80 % of the time).  1.0 x r and the fp32 add
    // are exact, so this is "accumulate, add the residual in fp32, round once".  (Before: 4 RB 8-byte loads per lane ahead of the K loop as the
    // accumulators' initial values -- a 72 KB round trip per workgroup that nothing overlapped: 30.8 us against 17.9 plain
    // at [256, 288 -> 288, 14, 14]; adding them in the epilogue from loads one row block ahead measured the same.)
    const int nchR = RES ? ((d.nrb - rb0 < 2 * RB ? d.nrb - rb0 : 2 * RB) + 1) / 2 : 0;
    const char* rbase = RES ? uniform_bytes(reinterpret_cast<const char*>(R)) : xbase;
    long long roff0 = 0;
    if (RES) {
        const long long q = dma_ok ? ug_dma : 0;
        const int f = (int)(q / d.U), j = (int)(q - (long long)f * d.U);
        const int px = (odd_tail && j == d.U - 1) ? d.P - 8 : 8 * j;
        roff0 = (((long long)f * d.M) * d.P + px) * 2;
    }
    auto issue_r = [&](int j, int stage) {                          // chunk j of the residual's rows of this workgroup
        const unsigned dst = xs0 + (unsigned)(stage * kXStage + 4 * wave * kXGroup);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int k = 16 * rb0 + kCh * j + 16 * wave + 4 * i + (lane >> 4);
            k = k < d.M ? k : d.M - 1;                              // rows past M: a valid row, added to rows that are not stored
            if (dma_ok) dma::dma16s<false>(rbase, (int)(roff0 + (long long)k * d.P * 2), dst + (unsigned)(i * kXGroup));
        }
    };
    auto issue_x = [&](int c, int stage) {                          // (called with c = 0, 1, 2, ... in order)
        if (RES && c >= d.nch) { issue_r(c - d.nch, stage); return; }
        const unsigned dst = xs0 + (unsigned)(stage * kXStage + 4 * wave * kXGroup);
        const bool clamp = k_ragged && c == d.nch - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int voff = xvo[i];
            if (clamp) {                                            // rows past K: a valid row, masked at the read
                int k = kCh * c + 16 * wave + 4 * i + (lane >> 4);
                k = k < d.K ? k : d.K - 1;
                voff = (int)(xoff0 + (long long)k * d.P * 2);
            }
            if (dma_ok) dma::dma16s<false>(xbase, voff, dst + (unsigned)(i * kXGroup));
            xvo[i] += xstep;
        }
    };
    // A: ordinary 16-byte loads of the packed fragments by waves 2-3 (block w - 2, w, ... of the chunk's 2 RB), written to
    // LDS a chunk ahead.  NOT by LDS-DMA: a CU's DMA engine lands about 25 GB/s (MI355X_MICROARCH.md, ldsdma-fill), and
    // every workgroup re-reads the whole operand from L2 -- 2.2x the bytes of X at 288 rows; measured 19.7 -> see DESIGN.
    u32x4 areg[RB];
    int ablk[RB];                                                   // 16-byte index of this lane's piece of block b
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        int blk = rb0 + (wave & 1) + 2 * b;
        blk = blk < d.nrb ? blk : d.nrb - 1;                        // (blocks past the operand: a copy, never multiplied)
        ablk[b] = blk * 64 + lane;
    }
    const u32x4* asrc = reinterpret_cast<const u32x4*>(Apk);
    const int astep = d.nrb * 64;                                   // 16-byte pieces per chunk of the packed operand
    // identity fragments for the residual's chunks (see issue_r below), made in registers: lane (m = lane & 15, kg = lane >> 4)
    // of the block with parity h holds a 1.0 at k = 16 h + m, i.e. in element m & 7 of k-group kg = 2 h + (m >> 3)
    const int id_kg = (lane >> 4) - ((lane & 15) >> 3);             // = 2 h where this lane holds the 1.0
    const int id_dw = (lane & 7) >> 1;
    const unsigned id_one = (lane & 1) ? 0x3f800000u : 0x00003f80u; // bf16 1.0 in the element's half of its dword
    auto fetch_a = [&](int c) {
        if (RES && c >= d.nch) {
            const int j = c - d.nch;
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int blk = (wave & 1) + 2 * b;                 // (workgroup-local block: the chunk holds blocks 2 j, 2 j + 1)
                const unsigned v = ((blk >> 1) == j && id_kg == 2 * (blk & 1)) ? id_one : 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) areg[b][e] = id_dw == e ? v : 0u;
            }
            return;
        }
        const u32x4* src0 = asrc + (size_t)c * astep;
#pragma unroll
        for (int b = 0; b < RB; ++b) areg[b] = src0[ablk[b]];
    };
    auto deposit_a = [&](int stage) {
        u32x4* dst0 = reinterpret_cast<u32x4*>(As + stage * kAStage) + lane;
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int blk = (wave & 1) + 2 * b;
            dst0[blk * 64] = areg[b];
        }
    };

    // output geometry of this lane
    const long long ug = (long long)blockIdx.x * 16 + 8 * cgp + (n >> 1);
    const bool out_ok = ug < d.nunits;
    size_t at0 = 0;                                  // element offset of this lane's UNIT in row 0 of its frame's output
    {
        const long long q = out_ok ? ug : 0;
        const int f = (int)(q / d.U), j = (int)(q - (long long)f * d.U);
        const int p = (odd_tail && j == d.U - 1) ? d.P - 8 : 8 * j;  // (a frame's last unit: its repeated half is stored again,
        at0 = ((size_t)f * d.M) * d.P + p;                           //  with the identical values)
    }
    const int half = n & 1;                          // this lane's 4 pixels within the unit
    const int rowb = 16 * (rb0 + rh * RB) + 4 * g;
    f32x4 acc[RB][4];                                // zeroed in the first iteration

    const int nch = d.nch, nchT = nch + nchR;        // X's chunks, then the residual's
    if (x_wave) {
#pragma unroll
        for (int j = 0; j < DX - 1; ++j)
            if (j < nchT) issue_x(j, j);
    } else {
        fetch_a(0);
        deposit_a(0);
    }
    int sx = 0, sa = 0;                              // stages of chunk c: c % DX, c & 1
    const bool ragged = (d.K & (kCh - 1)) != 0;
    const char* xrd = Xs + (2 * g) * kXGroup + cgp * 128 + n * 8;          // row 8 g + i: group 2 g + (i >> 2), row i & 3
    const char* ard = As + (rh * RB) * 1024 + lane * 16;

    auto step = [&](int c, auto first) {
        // chunk c has landed when only the younger chunks of this wave's stream (up to D - 2 of them) are outstanding
        const int left = nchT - 1 - c;
        if (x_wave) dma::wait_vmcnt(4 * (left < DX - 2 ? left : DX - 2));
        __syncthreads();
        if (decltype(first)::value) {
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[r][0][i] = acc[r][1][i] = acc[r][2][i] = acc[r][3][i] = 0.f;
        }
        if (x_wave) {
            if (c + DX - 1 < nchT) issue_x(c + DX - 1, sx == 0 ? DX - 1 : sx - 1);     // into the stage of chunk c - 1
        } else {
            if (c + 1 < nchT) fetch_a(c + 1);                                          // lands under this chunk's MFMAs
        }

        const char* xs = xrd + sx * kXStage;
        const char* as = ard + sa * kAStage;
        sx = sx == DX - 1 ? 0 : sx + 1;
        sa ^= 1;
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
            // (row blocks past the operand hold copies of its last block and are not stored: no branch per block)
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(as + r * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[q], acc[r][q], 0, 0, 0);
        }
        if (!x_wave && c + 1 < nchT) deposit_a(sa);          // sa is already the stage of chunk c + 1 (last read in step c - 1)
    };
    step(0, std::true_type{});
#pragma nounroll
    for (int c = 1; c < nchT; ++c) step(c, std::false_type{});

    // results: lane (n, g) holds rows 16 rb + 4 g + i, columns 4 n + q of its column group = half a unit.  Lane pairs
    // swap two rows each (DPP) so that every lane stores 2 rows x 16 bytes instead of 4 rows x 8: the epilogue is
    // store-ISSUE bound (MI355X_MICROARCH.md, "attention epilogue store tail").
    const unsigned hm = 0u - (unsigned)half;          // all ones in the odd lane of a pair
    // STATS: this lane's 4 columns exist (once) in Y -- not past the tensor, not the repeated half of a frame's last unit.
    // (recomputed here rather than kept from the prologue: carried across the K loop the flags cost 50 spilled SGPRs, the VGPRs
    // that held them 24 spilled VGPRs, on a kernel that sits at its 256-register cap)
    bool counted = false;
    float ncols = 0.f;
    if (STATS) {
        const long long q = out_ok ? ug : 0;
        const int j = (int)(q % d.U);
        counted = out_ok && !(odd_tail && j == d.U - 1 && (n & 1) == 0);
        ncols = counted ? 4.f : 0.f;
        ncols += dpp_or_zero<0x111, 0xf>(ncols); ncols += dpp_or_zero<0x112, 0xf>(ncols);
        ncols += dpp_or_zero<0x114, 0xf>(ncols); ncols += dpp_or_zero<0x118, 0xf>(ncols);     // lane n = 15: the wave's columns
    }
    const int J = 2 * (int)gridDim.x;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        unsigned w[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w[i][0] = bf16_bits(acc[r][0][i]) | (bf16_bits(acc[r][1][i]) << 16);
            w[i][1] = bf16_bits(acc[r][2][i]) | (bf16_bits(acc[r][3][i]) << 16);
        }
        if (STATS) {
            // (pivot 0: bf16 data carries 8 bits, the fp32 sums of 64 columns 24, and the finisher combines the records in fp64)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float y0 = __uint_as_float(w[i][0] << 16), y1 = __uint_as_float(w[i][0] & 0xffff0000u);
                const float y2 = __uint_as_float(w[i][1] << 16), y3 = __uint_as_float(w[i][1] & 0xffff0000u);
                float s1 = counted ? (y0 + y1) + (y2 + y3) : 0.f;
                float s2 = counted ? fmaf(y0, y0, y1 * y1) + fmaf(y2, y2, y3 * y3) : 0.f;
                s1 += dpp_or_zero<0x111, 0xf>(s1); s2 += dpp_or_zero<0x111, 0xf>(s2);
                s1 += dpp_or_zero<0x112, 0xf>(s1); s2 += dpp_or_zero<0x112, 0xf>(s2);
                s1 += dpp_or_zero<0x114, 0xf>(s1); s2 += dpp_or_zero<0x114, 0xf>(s2);
                s1 += dpp_or_zero<0x118, 0xf>(s1); s2 += dpp_or_zero<0x118, 0xf>(s2);
                int row = rowb + 16 * r + i;
                asm volatile("" : "+v"(row));          // (or hipcc precomputes the 36 row masks ahead of the K loop: 50 spilled SGPRs)
                if (n == 15 && row < d.M) stats[(size_t)row * J + 2 * blockIdx.x + cgp] = make_float4(0.f, s1, s2, ncols);
            }
        }
        // even lane keeps rows 0, 1 and receives the partner's halves of them; odd lane keeps rows 2, 3
        unsigned send[2][2], recv[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                send[e][h] = (w[e][h] & hm) | (w[2 + e][h] & ~hm);      // v_bfi_b32 (a ?: of array elements becomes a scratch index)
                recv[e][h] = (unsigned)__builtin_amdgcn_mov_dpp((int)send[e][h], 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            }
        if (!out_ok) continue;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = rowb + 16 * r + 2 * half + e;
            if (row >= d.M) continue;
            const unsigned own0 = (w[2 + e][0] & hm) | (w[e][0] & ~hm), own1 = (w[2 + e][1] & hm) | (w[e][1] & ~hm);
            u32x4 o;
            o[0] = (recv[e][0] & hm) | (own0 & ~hm); o[1] = (recv[e][1] & hm) | (own1 & ~hm);
            o[2] = (own0 & hm) | (recv[e][0] & ~hm); o[3] = (own1 & hm) | (recv[e][1] & ~hm);
            *reinterpret_cast<u32x4*>(Y + at0 + (size_t)row * d.P) = o;
        }
    }
}

inline int rows_per_wave(int nrb) { return nrb > 10 ? 9 : (nrb > 6 ? 5 : 3); }

template <int RB, bool RES, int DX, bool STATS = false>
int launch_gemm(const char* Apk, const __hip_bfloat16* X, const __hip_bfloat16* R, __hip_bfloat16* Y, const Dims& d,
                hipStream_t stream, float4* stats = nullptr) {
    constexpr size_t lds = (size_t)DX * kXStage + (size_t)2 * 2 * RB * 1024;
    static DynLdsRaised raised;                      // > 64 KB of dynamic LDS needs the attribute: once per instance and device
    if (const int rc = raise_dynamic_lds(reinterpret_cast<const void*>(&k_pw16_gemm<RB, RES, DX, STATS>), lds, raised)) return rc;
    const dim3 grid((unsigned)((d.nunits + 15) / 16), (unsigned)((d.nrb + 2 * RB - 1) / (2 * RB)));
    hipLaunchKernelGGL((k_pw16_gemm<RB, RES, DX, STATS>), grid, dim3(kBlock), lds, stream, Apk, X, R, Y, d, stats);
    return launch_status();
}

// ---------------------------------------------------------------------------------------------
// d(weight):  dW[m][k] = sum over pixels of dY[f][m][p] * X[f][k][p].  The reduction index is the contiguous one, so a
// 16-byte piece of a row (a "unit" = 8 pixels, as above) IS an MFMA fragment lane: both operands go into LDS unit by
// unit (16-byte loads, two stages ahead in registers; NOT LDS-DMA: the output tiles re-read both operands from L2 and a
// CU's DMA engine lands only ~25 GB/s, less on 64-byte pieces -- measured 53 us at 288 x 288) and come back with one
// ds_read_b128 per fragment.  A workgroup owns an output tile of up to 160 x 160
// (2 x 2 waves of up to 5 x 5 blocks of 16 x 16: at most 100 accumulator registers) and a range of units; a k-step is
// 4 units = 32 pixels of all the tile's rows of both operands.  LDS slot of (row, unit u of the step) = 4 row +
// (u ^ (-(row >> 2) & 3)): a row's 4 units stay 64 contiguous bytes for the loads, and the 16 rows of a fragment read fall into 16 distinct 16-byte bank groups.  When P % 8 == 4
// the last unit of a frame repeats 4 pixels of the unit before (see k_pw16_gemm): that half is zeroed in the dY
// fragment.  Partials go to ws[split][M][K]; k_pw16_reduce sums them in a fixed order.
struct WDims {
    int F, K, M, P;
    int U;
    long long nunits;
    int S;                       // splits of the unit sequence
    int ups;                     // units per split (a multiple of 4)
    int tilesM, tilesK;          // output tiles
    int tbM, tbK;                // 16-row blocks per tile
    int mbT, kbT;                // total blocks: ceil(M / 16), ceil(K / 16)
};

template <int BM, int BK>
__global__ __launch_bounds__(kBlock, 2) void k_pw16_wgrad(const __hip_bfloat16* __restrict__ dY,
                                                          const __hip_bfloat16* __restrict__ X, float* __restrict__ ws,
                                                          WDims d) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NL = 5;                            // 16-byte pieces per thread and stage (20 row groups of 16 / 4 waves)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm = wave & 1, wk = wave >> 1;
    const int m16 = lane & 15, g = lane >> 4;
    // workgroup -> (split, tile): the tiles of one split are 8 ids apart, i.e. on the same XCD and dispatched together, so
    // that the second reader of an operand row finds it in that XCD's L2
    const int T = d.tilesM * d.tilesK;
    const int blk8 = blockIdx.x / (8 * T), rem = blockIdx.x - blk8 * 8 * T;
    const int tile = rem >> 3, split = blk8 * 8 + (rem & 7);
    if (split >= d.S) return;
    const int tm = tile / d.tilesK, tk = tile - tm * d.tilesK;
    const int rowsM = 16 * d.tbM, rowsK = 16 * d.tbK;            // staged rows of dY / X
    const int nq = (rowsM + rowsK) / 16;                         // row groups of 16 per stage (<= 4 NL)
    const int stage_bytes = (rowsM + rowsK) * 64;
    const long long u_lo = (long long)split * d.ups;
    long long u_hi = u_lo + d.ups;
    u_hi = u_hi < d.nunits ? u_hi : d.nunits;
    const int nsteps = (int)((u_hi - u_lo + 3) / 4);
    const bool odd_tail = (d.P & 7) != 0;

    // load role: piece j of this thread = row group q = wave + 4 j, row 16 q + (lane >> 2), LDS slot lane of the group, i.e.
    // unit (lane & 3) ^ (-(lane >> 4) & 3) of the step (the same unit for every piece)
    const int du = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
    long long dug = u_lo + du;
    int df, dj;
    {
        const long long q = dug < d.nunits ? dug : d.nunits - 1;
        df = (int)(q / d.U); dj = (int)(q - (long long)df * d.U);
    }
    const char* ybase = reinterpret_cast<const char*>(dY);
    const char* xbase = reinterpret_cast<const char*>(X);
    long long choff[NL];                                          // byte offset of the piece's channel row in its frame
    bool pieceY[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        int q = wave + 4 * j;
        q = q < nq ? q : nq - 1;                                  // (pieces past the stage: a copy, not deposited)
        const int r = 16 * q + (lane >> 2);
        pieceY[j] = 16 * q < rowsM;
        int ch = pieceY[j] ? 16 * d.tbM * tm + r : 16 * d.tbK * tk + (r - rowsM);
        const int C = pieceY[j] ? d.M : d.K;
        ch = ch < C ? ch : C - 1;                                 // rows past the operand: a copy, never stored
        choff[j] = (long long)ch * d.P * 2;
    }
    auto fetch = [&](u32x4 (&v)[NL]) {
        // (units past the end of the tensor re-read its last unit: finite values, zeroed in the dY fragment)
        const int px = (odd_tail && dj == d.U - 1) ? d.P - 8 : 8 * dj;
        const long long offY = (((long long)df * d.M) * d.P + px) * 2, offX = (((long long)df * d.K) * d.P + px) * 2;
#pragma unroll
        for (int j = 0; j < NL; ++j)
            v[j] = *reinterpret_cast<const u32x4*>((pieceY[j] ? ybase + offY : xbase + offX) + choff[j]);
        dug += 4; dj += 4;
        while (dj >= d.U) { dj -= d.U; df += 1; }
        if (dug >= d.nunits) { df = (int)((d.nunits - 1) / d.U); dj = (int)((d.nunits - 1) - (long long)df * d.U); }
    };
    auto deposit = [&](const u32x4 (&v)[NL], int stage) {
        char* dst = lds + stage * stage_bytes + lane * 16;
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (wave + 4 * j < nq) *reinterpret_cast<u32x4*>(dst + (wave + 4 * j) * 1024) = v[j];
    };

    // compute role: lane (m16, g) reads unit g of rows 16 b + m16; its own running unit index for the masks
    long long cug = u_lo + g;
    int cj = (int)(cug % d.U);
    const int nbm = (d.tbM - wm * BM) < BM ? (d.tbM - wm * BM) : BM;       // blocks of this wave (may be <= 0)
    const int nbk = (d.tbK - wk * BK) < BK ? (d.tbK - wk * BK) : BK;
    f32x4 acc[BM][BK];
#pragma unroll
    for (int a = 0; a < BM; ++a)
#pragma unroll
        for (int b = 0; b < BK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // slot(row, u) = 4 row + (u ^ (-(row >> 2) & 3)); row = 16 b + m16 -> (row >> 2) & 3 = (m16 >> 2) & 3.  The key
    // (0, 3, 2, 1) per 4-row group -- not (0, 1, 2, 3) -- because a ds_read_b128 is served in the lane groups {0-3, 12-15,
    // 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): with the plain key two lanes of a group shared a
    // 16-byte bank group (SQ_LDS_BANK_CONFLICT = 34 % of the LDS cycles)
    const int rdoff = 16 * (4 * m16 + (g ^ ((0 - (m16 >> 2)) & 3)));

    auto step = [&](int s, u32x4 (&v)[NL]) {
        const int st = s & 1;
        deposit(v, st);                                           // stage s (loaded two steps ago) -> LDS[s & 1]
        __syncthreads();
        if (s + 2 < nsteps) fetch(v);                             // stage s + 2 into the registers just freed
        const char* base = lds + st * stage_bytes + rdoff;
        // mask of this lane's unit: 0 = all of it counts, 1 = only its second half (a frame's last unit), 2 = none
        const int mask = cug >= u_hi ? 2 : ((odd_tail && cj == d.U - 1) ? 1 : 0);
        cug += 4; cj += 4;
        while (cj >= d.U) cj -= d.U;
        bf16x8 fa[BM], fb[BK];
#pragma unroll
        for (int a = 0; a < BM; ++a) {
            const int ac = a < nbm ? a : 0;
            u32x4 t = *reinterpret_cast<const u32x4*>(base + (wm * BM + ac) * 1024);
            if (mask >= 1) { t[0] = 0u; t[1] = 0u; }
            if (mask == 2) { t[2] = 0u; t[3] = 0u; }
            fa[a] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
        for (int b = 0; b < BK; ++b) {
            const int bc = b < nbk ? b : 0;
            fb[b] = *reinterpret_cast<const bf16x8*>(base + rowsM * 64 + (wk * BK + bc) * 1024);
        }
#pragma unroll
        for (int a = 0; a < BM; ++a)
#pragma unroll
            for (int b = 0; b < BK; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);   // (blocks past the tile: copies, not
                                                                                                        //  stored; a guard here is a branch per MFMA)
    };

    u32x4 va[NL], vb[NL];
    fetch(va);
    if (1 < nsteps) fetch(vb);
    else {
#pragma unroll
        for (int j = 0; j < NL; ++j) vb[j] = va[j];
    }
#pragma nounroll
    for (int s = 0; s < nsteps; s += 2) {
        step(s, va);
        if (s + 1 < nsteps) step(s + 1, vb);
    }

    // acc[a][b][i]: m = 16 (block a) + 4 g + i, k = 16 (block b) + m16
    float* out = ws + (size_t)split * d.M * d.K;
#pragma unroll
    for (int a = 0; a < BM; ++a)
#pragma unroll
        for (int b = 0; b < BK; ++b) {
            if (a >= nbm || b >= nbk) continue;
            const int k = 16 * (d.tbK * tk + wk * BK + b) + m16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * (d.tbM * tm + wm * BM + a) + 4 * g + i;
                if (m < d.M && k < d.K) out[(size_t)m * d.K + k] = acc[a][b][i];
            }
        }
}

// A second form was measured in round 5 and not kept (profiles/r05_pw16_wgrad_{pmc,sweep}.txt): 3 x 3 waves of
// 3 x 3 blocks, one workgroup per CU, half the splits, 4 / 6 / 8 register stages of loads in flight, transposed blocks
// for 16-byte stores -- 35.3 -> 33.5 us stand-alone at 288 x 288, level at any prefetch depth, and level in the train step.
// The counters say why: L2 serves the second reader of every operand row (TCC misses x 128 B = operands once + partials),
// and the L1s sit in TCP_PENDING_STALL for more than half of the kernel -- 2.7 M 64-byte requests for 118 MB of loads (rows
// are 392 B apart at 14 x 14: a row's 64-byte piece straddles sectors), ~196 MB of sectors in all between L2 and the CUs
// at the ~6.5 TB/s every streaming kernel here tops out at.  The bound is the formulation's traffic (operands x tiles per
// side + 2 x splits x M K x 4), not latency.
// out[i] = sum over the S partial matrices, fixed order: 4 slices of the split range per output (one per wave, 64
// outputs per workgroup), each summed front to back, then the 4 slice sums added in slice order
__global__ __launch_bounds__(kBlock) void k_pw16_reduce(const float* __restrict__ in, float* __restrict__ out, int MK, int S) {
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

inline int make_wdims(WDims& d, int F, int K, int M, int P) {
    if (F <= 0 || K <= 0 || M <= 0 || P < 8 || P % 4 != 0) return RK_ERR_BAD_DIMS;
    if ((long long)F * (K > M ? K : M) * P * 2 >= (1ll << 31)) return RK_ERR_BAD_DIMS;
    d.F = F; d.K = K; d.M = M; d.P = P;
    d.U = (P + 7) / 8; d.nunits = (long long)F * d.U;
    d.mbT = (M + 15) / 16; d.kbT = (K + 15) / 16;
    d.tilesM = (d.mbT + 9) / 10; d.tilesK = (d.kbT + 9) / 10;
    d.tbM = (d.mbT + d.tilesM - 1) / d.tilesM; d.tbK = (d.kbT + d.tilesK - 1) / d.tilesK;
    const int T = d.tilesM * d.tilesK;
    // splits: about 512 workgroups, at least 12 k-steps each, partial matrices of at most 48 MB in all (24 MB: 292 of 512 workgroup slots at 288 x 288, 36.9 us; 48 MB and above: 33.7)
    long long S = 512 / T;
    const long long by_steps = d.nunits / 48, by_bytes = (48ll << 20) / ((long long)M * K * 4);
    S = S < by_steps ? S : by_steps;
    S = S < by_bytes ? S : by_bytes;
    S = S < 1 ? 1 : S;
    long long ups = (d.nunits + S - 1) / S;
    ups = (ups + 3) / 4 * 4;
    d.ups = (int)ups;
    d.S = (int)((d.nunits + ups - 1) / ups);
    return RK_OK;
}

template <int BM, int BK>
int launch_wgrad(const __hip_bfloat16* dY, const __hip_bfloat16* X, float* ws, const WDims& d, hipStream_t stream) {
    const size_t lds = (size_t)2 * (16 * d.tbM + 16 * d.tbK) * 64;          // <= 40 KB
    const int T = d.tilesM * d.tilesK;
    const unsigned grid = (unsigned)(((d.S + 7) / 8) * 8 * T);
    hipLaunchKernelGGL((k_pw16_wgrad<BM, BK>), dim3(grid), dim3(kBlock), lds, stream, dY, X, ws, d);
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
    const int nf = fwd ? ((Cout + 15) / 16) * ((Cin + kCh - 1) / kCh) * 64 : 0;
    const int nb = bwd ? ((Cin + 15) / 16) * ((Cout + kCh - 1) / kCh) * 64 : 0;
    hipLaunchKernelGGL(k_pw16_pack, dim3((nf + nb + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, W, Cout, Cin, nf, nb,
                       (uint4*)fwd, (uint4*)bwd);                  // one launch for both operands
    return launch_status();
}

// jobs: device array of n records {const float* W; int64 fwd_off, bwd_off; int Cout, Cin, nf, nb} (40 bytes, nf / nb = 16-byte
// units of the two images = rk_pw_packed_bytes / 16; offsets into `base`, multiples of 16); max_units = the largest nf + nb
int rk_pw_pack_many_bf16(const void* jobs, int n, void* base, int max_units, rk_stream_t stream_) {
    if (!jobs || !base) return RK_ERR_NULL_POINTER;
    if (n <= 0 || max_units <= 0 || n > 65535) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)base & 15) || ((uintptr_t)jobs & 7)) return RK_ERR_BAD_DIMS;
    static_assert(sizeof(PackJob) == 40, "the record layout pointwise.py writes");
    hipLaunchKernelGGL(k_pw16_pack_many, dim3((max_units + kBlock - 1) / kBlock, n), dim3(kBlock), 0, (hipStream_t)stream_,
                       (const PackJob*)jobs, (char*)base);
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
    if (R && (long long)F * M * P * 2 >= (1ll << 31)) return RK_ERR_BAD_DIMS;        // (the residual goes through the same DMA)
    Dims d;
    d.F = F; d.K = K; d.M = M; d.P = P;
    d.nrb = (M + 15) / 16; d.nch = (K + kCh - 1) / kCh;
    d.U = (P + 7) / 8; d.nunits = (long long)F * d.U;
    hipStream_t stream = (hipStream_t)stream_;
    const char* A = (const char*)Apk;
    const int rb = rows_per_wave(d.nrb);
#define RK_GO(RBV, DXV) (R ? launch_gemm<RBV, true, DXV>(A, X, R, Y, d, stream) : launch_gemm<RBV, false, DXV>(A, X, R, Y, d, stream))
    if (rb == 9) return RK_GO(9, 3);
    if (rb == 5) return RK_GO(5, 3);
    return RK_GO(3, 3);
#undef RK_GO
}

// The same GEMM in training: also the tile statistics of Y (float4 [M][tiles], tiles = rk_pw16_stat_tiles(F, P): one record
// per 64 columns) for the BatchNorm that consumes Y -- finished by rk_bn_finish_tiles_f32.
int rk_pw16_stat_tiles(int F, int P) {
    if (F <= 0 || P < 8) return 0;
    const long long nunits = (long long)F * ((P + 7) / 8);
    return (int)(2 * ((nunits + 15) / 16));
}
int rk_pw_gemm_packed_stats_bf16(const void* Apk, const void* X_, const void* R_, void* Y_, int F, int K, int M, int P,
                                 void* stats, int tiles, rk_stream_t stream_) {
    const __hip_bfloat16* X = (const __hip_bfloat16*)X_;
    const __hip_bfloat16* R = (const __hip_bfloat16*)R_;
    __hip_bfloat16* Y = (__hip_bfloat16*)Y_;
    if (!Apk || !X || !Y || !stats) return RK_ERR_NULL_POINTER;
    if (F <= 0 || K <= 0 || M <= 0 || P < 8 || P % 4 != 0) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)Apk & 15) || ((uintptr_t)X & 7) || ((uintptr_t)Y & 7) || (R && ((uintptr_t)R & 7)) || ((uintptr_t)stats & 15))
        return RK_ERR_BAD_DIMS;
    if ((long long)F * K * P * 2 >= (1ll << 31) || (R && (long long)F * M * P * 2 >= (1ll << 31))) return RK_ERR_BAD_DIMS;
    if (tiles != rk_pw16_stat_tiles(F, P)) return RK_ERR_BAD_DIMS;
    Dims d;
    d.F = F; d.K = K; d.M = M; d.P = P;
    d.nrb = (M + 15) / 16; d.nch = (K + kCh - 1) / kCh;
    d.U = (P + 7) / 8; d.nunits = (long long)F * d.U;
    hipStream_t stream = (hipStream_t)stream_;
    const char* A = (const char*)Apk;
    const int rb = rows_per_wave(d.nrb);
    float4* st = (float4*)stats;
#define RK_GO(RBV, DXV) (R ? launch_gemm<RBV, true, DXV, true>(A, X, R, Y, d, stream, st) : launch_gemm<RBV, false, DXV, true>(A, X, R, Y, d, stream, st))
    if (rb == 9) return RK_GO(9, 3);
    if (rb == 5) return RK_GO(5, 3);
    return RK_GO(3, 3);
#undef RK_GO
}

// d(weight) [M][K] (fp32) = sum_f dY[f] X[f]^T for bf16 activations (dY [F, M, P], X [F, K, P], P % 4 == 0, P >= 8); ws of
// rk_pw_wgrad16_workspace_bytes() bytes holds the per-split partial matrices.
size_t rk_pw_wgrad16_workspace_bytes(int F, int K, int M, int P) {
    WDims d;
    if (make_wdims(d, F, K, M, P)) return 0;
    return (size_t)d.S * M * K * sizeof(float);
}
int rk_pw_wgrad16_bf16(const void* dY_, const void* X_, float* dW, int F, int K, int M, int P, void* ws, size_t ws_bytes,
                       rk_stream_t stream_) {
    const __hip_bfloat16* dY = (const __hip_bfloat16*)dY_;
    const __hip_bfloat16* X = (const __hip_bfloat16*)X_;
    if (!dY || !X || !dW) return RK_ERR_NULL_POINTER;
    WDims d;
    if (int rc = make_wdims(d, F, K, M, P)) return rc;
    if (((uintptr_t)dY & 7) || ((uintptr_t)X & 7)) return RK_ERR_BAD_DIMS;
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
        hipLaunchKernelGGL(k_pw16_reduce, dim3((MK + 63) / 64), dim3(kBlock), 0, stream, (const float*)ws, dW, MK, d.S);
    return launch_status();
}

}  // extern "C"
