// rk_pw.hip -- 1x1 ("pointwise") convolution on NCHW activations as an MFMA GEMM (the
// unfused half): the Conv1x1 layers around every shift (rubiksnet/backbone.py:44-45, :87-104: conv2, conv3,
// stride-1 shortcuts).  After the shift and BN+ReLU kernels these are 61 % of the Tiny train step; MIOpen runs
// them as NHWC implicit GEMMs between NCHW<->NHWC transposes, 1.8x (forward) to 2.7x (forward + backward) off
// a one-pass-per-tensor bound.
//
//   forward / d(input):  Y[f] = A X[f]            A: [M x K] (the weight, or its transpose for d(input)),
//                                                 X: [F, K, P], Y: [F, M, P]   (P = H*W pixels, contiguous)
//   d(weight):           dW = sum_f dY[f] X[f]^T  -> [M x K]
//
// No layout change: a pixel is a GEMM column, and the columns of a frame are contiguous, so the streamed
// operand is read with 16-byte loads straight into MFMA B-fragments -- lane l of a wave owns the 4 consecutive
// columns 4 (l & 31) .. +3 and feeds them to 4 interleaved 32x32 column blocks (block q = columns 4 i + q), which
// also makes the lane's 4 results of a row consecutive: outputs leave as 16-byte stores.  The small operand A
// is staged through LDS in [k][m] order (conflict-free fragment reads), K in chunks of 16, double buffered, one
// barrier per chunk.  fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32: exact f32, k-ordered fmaf chain
// (cdna_hip_programming.md, "FP32-input MFMA") -- same arithmetic class as the MIOpen / rocBLAS fp32 kernels.
#include <type_traits>
#include "rk_common.hpp"
#include "rk3d_generic.hpp"
#include <cstdio>
#include "rk_pw2.hpp"
#include "rk_pw3.hpp"
#include "rk_pw4.hpp"

namespace rk {
namespace pw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 4 consecutive pixels of the streamed tensors: fp32 or bf16 storage, fp32 arithmetic either way (activations
// under bf16 autocast; the weights and d(weight) stay fp32, so the autocast casts of the weight disappear)
template <typename T> struct Px4;
template <> struct Px4<float> {
    using Raw = float4;
    __device__ static __forceinline__ Raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ static __forceinline__ Raw load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    __device__ static __forceinline__ float4 widen(const Raw& r) { return r; }
    __device__ static __forceinline__ void store(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
};
template <> struct Px4<__hip_bfloat16> {
    using Raw = uint2;
    __device__ static __forceinline__ Raw zero() { return make_uint2(0u, 0u); }
    __device__ static __forceinline__ Raw load(const __hip_bfloat16* p) { return *reinterpret_cast<const uint2*>(p); }
    __device__ static __forceinline__ float4 widen(const Raw& r) {
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
                           __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
    }
    __device__ static __forceinline__ unsigned bits(float f) {
        return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(f));
    }
    __device__ static __forceinline__ void store(__hip_bfloat16* p, const float4& v) {
        *reinterpret_cast<uint2*>(p) = make_uint2(bits(v.x) | (bits(v.y) << 16), bits(v.z) | (bits(v.w) << 16));
    }
};

struct PwDims {
    int F, K, M, P;
    long long ntot;         // F * P columns
    int a_is_mk;            // A given as [M][K] row-major (the weight itself), else [K][M]
    int Cin, Hin, Win, Wo;  // STEM mode only: 3x3 / stride 2 / pad 1 convolution, K = 9 Cin, P = Ho * Wo
    int WM, WN;             // waves along M / along N (WM * WN = 4); workgroup tile = 64 WM rows x 128 WN columns
};


// Inference-time fusion of the block's BatchNorms into the GEMM (eval mode: running statistics are constants):
//   prologue on the streamed operand, per input channel k:   x' = relu?(ka[k] x + kb[k])   (relu(bn1(x)) -> conv2)
//   epilogue on the result, per output channel m:             y  = relu?(ma[m] y + mb[m])   (relu(bn2(conv2(.))))
// NULL pointers switch a stage off.
// eval-mode BatchNorm as a per-channel affine map: a = gamma / sqrt(var + eps), b = beta - mean a (one launch instead of
// the five elementwise PyTorch kernels per BatchNorm and forward)
__global__ __launch_bounds__(kBlock) void k_bn_fold(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ mean, const float* __restrict__ var,
                                                    float eps, float* __restrict__ a, float* __restrict__ b, int C) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] * (1.0f / sqrtf(var[c] + eps));       // = rk_bn.hip's affine(): the unfused eval path
    a[c] = s;
    b[c] = fmaf(-mean[c], s, beta[c]);
}

// every eval-mode BatchNorm of a network in ONE launch (pointwise.prefolded: 28 folds of ~5 us per RubiksNet-Tiny forward):
// blockIdx.y = the layer; a[off .. off + C) / b[off .. off + C) of one [2][total] buffer
struct FoldJob { const float* gamma; const float* beta; const float* mean; const float* var; long long off; int C; float eps; };
__global__ __launch_bounds__(kBlock) void k_bn_fold_many(const FoldJob* __restrict__ jobs, float* __restrict__ ab, long long total) {
    const FoldJob j = jobs[blockIdx.y];
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= j.C) return;
    const float s = j.gamma[c] * (1.0f / sqrtf(j.var[c] + j.eps));  // (k_bn_fold's expression: bit-identical)
    ab[j.off + c] = s;
    ab[total + j.off + c] = fmaf(-j.mean[c], s, j.beta[c]);
}

struct PwFuse {
    const float* ka; const float* kb; const float* ma; const float* mb;
    int relu_in, relu_out;
};
constexpr int kProK = 336;               // input channels a prologue supports (320, padded to whole chunks of 12 / 16)

// Training-mode fusions of the block's BatchNorms into the GEMM epilogue (rubiksnet/backbone.py:
// 123-135).  A wave tile is 64 rows x 128 columns; tile j = the wave's index along the columns.
//   EPI = 1 (forward): the statistics pass of the BatchNorm that CONSUMES Y (bn2 after conv2, the next block's bn1 after
//     conv3 + shortcut) is done on the result tile while it is still in registers: per (row m, tile j) one float4
//     (pivot, sum(y - pivot), sum((y - pivot)^2), -) with pivot = the row's first value in the tile -- the shifted-data
//     form k_bn_stats uses, so E[y^2] - mean^2 never cancels -- written to stats[m][j]; k_bn_finish_tiles (rk_bn.hip)
//     combines the J tiles of a channel in fp64.  k_bn_stats (one full read of Y) disappears.
//   EPI = 2 (d(input) of conv2 = gradient of relu(bn1(x))): the result tile is the gradient da of the activation; with x
//     read at the same positions the ReLU mask and xhat are recomputed, Y receives dz = da * [a x + b > 0] and the tile's
//     (sum dz, sum dz * xhat) go to bred[m][j]: k_bn_bwd_reduce (one full read of da and x) becomes one read of x here.
struct PwTrain {
    float4* stats;                                        // EPI 1: [M][J]
    float2* bred;                                         // EPI 2: [M][J]
    const float* bx;                                      // EPI 2: the BatchNorm's input x, [F, M, P]
    const float4* bpack;                                  // EPI 2: [M] (a, b, mean, invstd) of the BatchNorm
    int J;
};

// Sum v[idx] over the 32 lanes that share (lane >> 5), for 32 values at once, in 31 shuffles instead of 32 x 5: at each
// of the 5 butterfly levels a lane keeps one half of its values and hands the other half to its partner (lane distance
// 16, 8, 4, 2, 1), so that lane l ends up with the total of idx l & 31.  Fixed order: deterministic.
template <int HALF, int N>
__device__ __forceinline__ void halfwave_fold(float (&v)[N], int l31) {
    const bool up = (l31 & (16 * 2 * HALF / N)) != 0;   // level with HALF = N/2 pairs lanes 16 apart, N/4: 8 apart, ...
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float keep = up ? v[i + HALF] : v[i];
        const float send = up ? v[i] : v[i + HALF];
        v[i] = keep + __shfl_xor(send, 16 * 2 * HALF / N);
    }
}
// 32 values per lane, summed over the 32 lanes of a half wave: afterwards lane l holds the total of idx (l & 31) in v[0]
__device__ __forceinline__ void halfwave_transpose_sum32(float (&v)[32], int l31) {
    halfwave_fold<16>(v, l31);
    halfwave_fold<8>(v, l31);
    halfwave_fold<4>(v, l31);
    halfwave_fold<2>(v, l31);
    halfwave_fold<1>(v, l31);
}

// A chunk -> registers (global, L2-resident) -> LDS image As[kk][m], m < MT, zero padded
template <int MT, int kKC>
struct AStage {
    static constexpr int kPer = (kKC * MT + kBlock - 1) / kBlock;
    float v[kPer];
    __device__ __forceinline__ void fetch(const float* __restrict__ A, const PwDims& d, int m0, int k0) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int e = threadIdx.x + kBlock * i;       // kk-major: e = kk * MT + m
            const int kk = e / MT, m = e - kk * MT;
            const int gk = k0 + kk, gm = m0 + m;
            const bool ok = kk < kKC && gk < d.K && gm < d.M;
            const size_t idx = d.a_is_mk ? (size_t)gm * d.K + gk : (size_t)gk * d.M + gm;
            v[i] = ok ? A[ok ? idx : 0] : 0.f;
        }
    }
    __device__ __forceinline__ void deposit(float* As) const {
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (kKC * MT % kBlock == 0 || threadIdx.x + kBlock * i < kKC * MT) As[threadIdx.x + kBlock * i] = v[i];
    }
};

// kKC: K chunk (12 or 16: the launcher picks the one that pads K less)
// STEM: the streamed operand is gathered as the im2col of a 3x3 / stride-2 / pad-1 convolution instead of read as
// a [K, P] matrix (the backbone's first layer, backbone.py:154 Conv3x3(3, width, stride=2)): column = output
// pixel, k = (ci, kh, kw); a lane's 4 output pixels read 4 input pixels 2 apart (the only out-of-range ones
// are row -1 / Hin and column -1).
// S2 = 1: the streamed operand is a [K, Hin, Win] frame read at stride 2 (1x1 / stride-2 convolution, the projecting
// shortcuts, backbone.py:98-104): column = output pixel (ho, wo), a lane's 4 pixels read 4 input pixels 2 apart.
// S2 = 2: d(input) of that convolution: the result block of output pixel (ho, wo..wo+3) is scattered to the input-
// sized tensor at (2 ho, 2 wo ..) with zeros in between and in row 2 ho + 1 (four 16-byte stores, every element of
// d(input) written once: no memset).
template <typename T, int WM, int kKC, bool FUSE, bool STEM = false, int S2 = 0, int EPI = 0>
__global__ __launch_bounds__(kBlock, ((WM == 2 && kKC == 16 && sizeof(T) == 4) ? 1 : 2)) void k_pw_gemm(const float* __restrict__ A, const T* __restrict__ X,
                                                    const T* __restrict__ R, T* __restrict__ Y, PwDims d, PwFuse fz,
                                                    PwTrain tr) {
    using Raw = typename Px4<T>::Raw;
    constexpr int MT = 64 * WM, WN = 4 / WM;
    __shared__ float As[2][kKC * MT];
    // prologue coefficients (ka, kb) of every input channel, padded to whole chunks: read per k-step from LDS (two
    // broadcast reads) -- as global loads inside the MFMA loop they cost 43 us of a 92 us GEMM at [256,54->54,56x56]
    __shared__ float2 Ks[FUSE ? kProK : 1];
    if constexpr (FUSE) {
        if (fz.ka) {
            for (int k = threadIdx.x; k < kProK; k += kBlock)
                Ks[k] = k < d.K ? make_float2(fz.ka[k], fz.kb[k]) : make_float2(0.f, 0.f);
        }                                                   // (visible after the first __syncthreads() of the chunk loop)
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * MT;
    const long long cg = ((long long)blockIdx.x * WN + wn) * 128 + 4 * l31;      // this lane's 4 columns
    const bool valid = cg < d.ntot;
    const long long cgc = valid ? cg : 0;
    const int f = (int)(cgc / d.P), p = (int)(cgc - (long long)f * d.P);
    const T* xp = X + ((size_t)f * d.K) * d.P + p;
    T* yp = Y + ((size_t)f * d.M) * d.P + p;

    f32x16 acc[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][q][r] = 0.f;

    const int s_ho = (STEM || S2) ? p / d.Wo : 0, s_wo = (STEM || S2) ? p - s_ho * d.Wo : 0;   // this lane's first output pixel
    // S2: offsets of the lane's 4 output pixels inside an input plane.  With Wo % 4 == 0 they sit in one row, 2 apart
    // (0, 2, 4, 6); otherwise (the 28 -> 14 shortcut: Wo = 14) a group of 4 may wrap into the next row.
    int s2o[4] = {0, 2, 4, 6};
    if constexpr (S2 != 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pq = p + q, hq = pq / d.Wo, wq = pq - hq * d.Wo;
            s2o[q] = (2 * hq - 2 * s_ho) * d.Win + 2 * wq - 2 * s_wo;
        }
    }
    auto load_b = [&](int k) -> Raw {
        if constexpr (STEM && std::is_same<T, float>::value) {
            const int ci = k / 9, r9 = k - 9 * ci, kh3 = r9 / 3, kw3 = r9 - 3 * kh3;
            const int hi = 2 * s_ho - 1 + kh3, wi = 2 * s_wo - 1 + kw3;
            const bool ok = valid && k < d.K && hi >= 0 && hi < d.Hin;
            const float* row = X + (((size_t)f * d.Cin + (ok ? ci : 0)) * d.Hin + (ok ? hi : 0)) * d.Win;
            float4 v;
            v.x = (ok && wi >= 0) ? row[wi >= 0 ? wi : 0] : 0.f;
            v.y = ok ? row[wi + 2] : 0.f;
            v.z = ok ? row[wi + 4] : 0.f;
            v.w = ok ? row[wi + 6] : 0.f;
            return v;
        } else if constexpr (S2 == 1 && std::is_same<T, float>::value) {
            const bool ok = valid && k < d.K;
            const float* row = X + (((size_t)f * d.K + (ok ? k : 0)) * d.Hin + 2 * s_ho) * d.Win + 2 * s_wo;
            float4 v;
            v.x = ok ? row[s2o[0]] : 0.f;
            v.y = ok ? row[s2o[1]] : 0.f;
            v.z = ok ? row[s2o[2]] : 0.f;
            v.w = ok ? row[s2o[3]] : 0.f;
            return v;
        } else {
            return (valid && k < d.K) ? Px4<T>::load(xp + (size_t)k * d.P) : Px4<T>::zero();
        }
    };

    const bool blk1 = m0 + wm * 64 + 32 < d.M;            // wave-uniform: does the second 32-row block hold any row?
    AStage<MT, kKC> ast;
    Raw bq[kKC / 2];                                      // B fragments of the current chunk, refilled in place:
    ast.fetch(A, d, m0, 0);                               // step s of chunk c+1 is requested right after step s of
#pragma unroll                                            // chunk c has consumed its registers (one chunk of MFMAs ahead)
    for (int s = 0; s < kKC / 2; ++s) bq[s] = load_b(2 * s + kh);
    ast.deposit(As[0]);
    const int nchunks = (d.K + kKC - 1) / kKC;
#pragma nounroll
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                                   // chunk c is in As[c & 1]; As[(c+1) & 1] is free
        const bool more = c + 1 < nchunks;
        if (more) ast.fetch(A, d, m0, (c + 1) * kKC);
        const float* as = As[c & 1] + wm * 64 + l31;
#pragma unroll
        for (int s = 0; s < kKC / 2; ++s) {
            const float a0 = as[(2 * s + kh) * MT], a1 = as[(2 * s + kh) * MT + 32];
            const float4 bw = Px4<T>::widen(bq[s]);
            float bv[4] = {bw.x, bw.y, bw.z, bw.w};
            if (FUSE && fz.ka) {                            // BN (+ReLU) of the input channel, applied on the fly
                const float2 pk = Ks[c * kKC + 2 * s + kh];    // (padded k: zeros; A is zero there as well)
                const float pa = pk.x, pb = pk.y;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = fmaf(pa, bv[q], pb);
                    bv[q] = fz.relu_in ? fmaxf(t, 0.f) : t;
                }
            }
            bq[s] = load_b((c + 1) * kKC + 2 * s + kh);    // (all zeros past K)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[q], acc[0][q], 0, 0, 0);
            if (blk1)                                       // (72 or 144 channels: the last 32-row block is all padding)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[q], acc[1][q], 0, 0, 0);
        }
        if (more) ast.deposit(As[(c + 1) & 1]);
    }

    if constexpr (EPI != 0 && std::is_same<T, float>::value && S2 != 2) {
        // training epilogues (PwTrain above).  Every lane walks every row: the shuffles need the whole wave.  One 32-row
        // block at a time (its 16 rows per lane -> 32 partial sums -> folded over the 32 lanes of the half wave), the
        // residual / x rows of 8 rows requested together before the first is used.
        const long long tj = (long long)blockIdx.x * WN + wn;
        const bool tile_on = tj * 128 < d.ntot;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float sv[32];
            float mypiv = 0.f;
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                float4 rv[8], xv[8];
                bool on[8];
                size_t at[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j;
                    const int gm = m0 + wm * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    on[j] = valid && gm < d.M;
                    at[j] = (size_t)(yp - Y) + (size_t)(on[j] ? gm : 0) * d.P;
                    rv[j] = (R && on[j]) ? *reinterpret_cast<const float4*>(R + at[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (EPI == 2)
                        xv[j] = on[j] ? *reinterpret_cast<const float4*>(tr.bx + at[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j;
                    const int gm = m0 + wm * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    float4 o = make_float4(acc[b][0][r] + rv[j].x, acc[b][1][r] + rv[j].y, acc[b][2][r] + rv[j].z,
                                           acc[b][3][r] + rv[j].w);
                    float s1 = 0.f, s2 = 0.f;
                    if constexpr (EPI == 1) {
                        const float piv = __shfl(o.x, kh << 5);       // the row's first column in this tile (always valid)
                        if ((l31 >> 1) == r) mypiv = piv;
                        if (on[j]) {
                            float t;
                            t = o.x - piv; s1 += t; s2 = fmaf(t, t, s2);
                            t = o.y - piv; s1 += t; s2 = fmaf(t, t, s2);
                            t = o.z - piv; s1 += t; s2 = fmaf(t, t, s2);
                            t = o.w - piv; s1 += t; s2 = fmaf(t, t, s2);
                        }
                    } else if (on[j]) {
                        const float4 pk = tr.bpack[gm];
                        const float pa = pk.x, pb = pk.y, mu = pk.z, iv = pk.w;
                        o.x = fmaf(pa, xv[j].x, pb) <= 0.f ? 0.f : o.x;  s1 += o.x;  s2 = fmaf(o.x, (xv[j].x - mu) * iv, s2);
                        o.y = fmaf(pa, xv[j].y, pb) <= 0.f ? 0.f : o.y;  s1 += o.y;  s2 = fmaf(o.y, (xv[j].y - mu) * iv, s2);
                        o.z = fmaf(pa, xv[j].z, pb) <= 0.f ? 0.f : o.z;  s1 += o.z;  s2 = fmaf(o.z, (xv[j].z - mu) * iv, s2);
                        o.w = fmaf(pa, xv[j].w, pb) <= 0.f ? 0.f : o.w;  s1 += o.w;  s2 = fmaf(o.w, (xv[j].w - mu) * iv, s2);
                    }
                    sv[2 * r] = s1; sv[2 * r + 1] = s2;
                    if (on[j]) *reinterpret_cast<float4*>(Y + at[j]) = o;
                }
            }
            halfwave_transpose_sum32(sv, l31);
            // lane l31 holds sum idx l31 = 2 row + stat of this 32-row block; pair the two statistics of a row in the even lane
            const float other = __shfl_xor(sv[0], 1);
            const int rr = l31 >> 1;
            const int gmi = m0 + wm * 64 + 32 * b + (rr & 3) + 8 * (rr >> 2) + 4 * kh;
            if ((l31 & 1) == 0 && gmi < d.M && tile_on) {
                if constexpr (EPI == 1) {
                    const long long nl = d.ntot - tj * 128;          // columns of this tile (the finisher's n_j)
                    tr.stats[(size_t)gmi * tr.J + tj] = make_float4(mypiv, sv[0], other, (float)(nl < 128 ? nl : 128));
                }
                else tr.bred[(size_t)gmi * tr.J + tj] = make_float2(sv[0], other);
            }
        }
    } else if (valid) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += 8) {
                // the residual rows of 8 result rows are requested together, before the first is used: one load -> use
                // -> store chain per row left the read of R exposed (gemm + R 150 us against 92 us without R at
                // [256,54->54,56x56]; batched: 120 us)
                float4 rv[8];
                if (R) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = r0 + j;
                        const int gm = m0 + wm * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        rv[j] = gm < d.M ? Px4<T>::widen(Px4<T>::load(R + (yp - Y) + (size_t)gm * d.P))
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = r0 + j;
                    const int gm = m0 + wm * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * kh;   // C/D map of 32x32 MFMA
                    if (gm < d.M) {
                        float4 o = make_float4(acc[b][0][r], acc[b][1][r], acc[b][2][r], acc[b][3][r]);
                        if (FUSE && fz.ma) {                          // BN (+ReLU) of the output channel
                            const float ea = fz.ma[gm], eb = fz.mb[gm];
                            o.x = fmaf(ea, o.x, eb); o.y = fmaf(ea, o.y, eb); o.z = fmaf(ea, o.z, eb); o.w = fmaf(ea, o.w, eb);
                            if (fz.relu_out) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                        }
                        if (R) { o.x += rv[j].x; o.y += rv[j].y; o.z += rv[j].z; o.w += rv[j].w; }   // Y = A X + R (the shortcut)
                        if constexpr (S2 == 2 && std::is_same<T, float>::value) {
                            float* q = Y + (((size_t)f * d.M + gm) * d.Hin + 2 * s_ho) * d.Win + 2 * s_wo;
                            if (d.Wo % 4 == 0) {
                                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                                *reinterpret_cast<float4*>(q) = make_float4(o.x, 0.f, o.y, 0.f);
                                *reinterpret_cast<float4*>(q + 4) = make_float4(o.z, 0.f, o.w, 0.f);
                                *reinterpret_cast<float4*>(q + d.Win) = z;
                                *reinterpret_cast<float4*>(q + d.Win + 4) = z;
                            } else {                                  // per pixel: its 2 x 2 cell (8-byte aligned: Win even)
                                const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    *reinterpret_cast<float2*>(q + s2o[e]) = make_float2(ov[e], 0.f);
                                    *reinterpret_cast<float2*>(q + s2o[e] + d.Win) = make_float2(0.f, 0.f);
                                }
                            }
                        } else {
                            Px4<T>::store(yp + (size_t)gm * d.P, o);
                        }
                    }
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------
// Planes whose pixel count is not a multiple of 4 (the 7x7 planes of layer4: 49-float rows, so a lane's "4 consecutive
// pixels" straddle rows and frames and nothing is 16-byte aligned per row).  What IS aligned and contiguous is the run
// of KC consecutive channels of one frame: KC * P floats starting at (f K + k0) P, 16-byte aligned whenever K, k0 are
// multiples of 4.  So the streamed operand goes through LDS as whole frame chunks, copied flat with 16-byte loads
// (LDS image = memory image: [frame][k][P]), and the B fragments are single-word LDS reads at lane-consecutive addresses;
// the 128-bit trick of k_pw_gemm (a lane owns 4 consecutive columns) is not needed because nothing streams past the CU
// fast enough to matter: these tensors are 1/4 .. 1/64 of the others and the layers are MFMA-bound (432 -> 432).
//   wave tile 64 rows x 32 NCB columns (2 x NCB MFMA blocks), workgroup = 4 waves along the columns.  NCB = 2 (256
//   columns) gives [256 frames, 432 channels, 7x7] only 49 x 7 = 343 workgroups for 256 CUs -- a second, third-full round:
//   87 us; NCB = 1 (128 columns, 686 workgroups of half the LDS) balances better and is what the launcher uses.
//   GATHER = 1: the streamed operand is read at stride 2 from [F, K, 2 Ho, 2 Wo] planes (the 14 -> 7 projecting shortcut,
//   backbone.py:98-104): the frame chunk is gathered element by element into the same LDS image.
//   SCATTER = 1: d(input) of that shortcut: the result of output pixel (ho, wo) goes to (2 ho, 2 wo) of a [2 Ho, 2 Wo]
//   plane, zeros to the other three positions of its 2 x 2 cell (every element written, no memset).
constexpr int kOddKC = 16;               // channels per chunk
constexpr int kOddFr = 8;                // frames a 256-column tile can touch (P >= 37: 256 / P + 2 <= 8)

template <int GATHER, int SCATTER, int NCB>
__global__ __launch_bounds__(kBlock) void k_pw_gemm_odd(const float* __restrict__ A, const float* __restrict__ X,
                                                        const float* __restrict__ R, float* __restrict__ Y, PwDims d) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = d.P;
    const int fchunk = kOddKC * P;                          // floats of one frame chunk
    float* As = smem;                                       // [2][kOddKC * 64]
    constexpr int kFr = NCB == 2 ? kOddFr : 6;             // frames a tile of 128 NCB columns can touch (P >= 37)
    constexpr int kCols = 128 * NCB;                        // columns per workgroup: 4 waves x NCB blocks of 32
    float* Xs = smem + 2 * kOddKC * 64;                     // [2][kFr * fchunk]
    const int xbuf = kFr * fchunk;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * 64;
    const long long n0 = (long long)blockIdx.x * kCols;     // first column of the workgroup
    const int f0 = (int)(n0 / P);                           // first frame it touches
    long long nlast = n0 + kCols - 1;
    nlast = nlast < d.ntot - 1 ? nlast : d.ntot - 1;
    const int nfr = (int)(nlast / P) - f0 + 1;              // frames touched (<= kOddFr)

    // this lane's two columns (one per 32-column block of the wave's 64) and their LDS word offsets inside a buffer
    long long col[NCB];
    int xoff[NCB], cf[NCB], cp[NCB];
    bool con[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        col[cb] = n0 + wave * (32 * NCB) + cb * 32 + l31;
        con[cb] = col[cb] < d.ntot;
        const long long cc = con[cb] ? col[cb] : n0;
        cf[cb] = (int)(cc / P);
        cp[cb] = (int)(cc - (long long)cf[cb] * P);
        xoff[cb] = (cf[cb] - f0) * fchunk + cp[cb];
    }

    f32x16 acc[2][NCB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NCB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Staging is split into fetch (global -> registers, issued BEFORE the MFMAs of the current chunk) and deposit
    // (registers -> LDS, after them), so that the load latency hides under the 32 MFMAs of a chunk; a load -> LDS store
    // -> MFMA sequence per chunk left it exposed 27 times per tile (152 us against MIOpen's 47 us at [256,432->432,7x7]).
    constexpr int kXV = (kFr * kOddKC * 64 / 4 + kBlock - 1) / kBlock;           // float4 per thread, P <= 64: 8 / 6
    constexpr int kAV = kOddKC * 64 / kBlock;                                    // 4 floats per thread
    const int nv = fchunk / 4;                               // (kOddKC * P) % 4 == 0
    // Everything about a thread's pieces that does not depend on the chunk is computed once: float4 e = tid + 256 i of
    // the LDS image is (frame e / nv, float4 v = e % nv of its chunk); its LDS word offset is simply 4 e.  The chunk loop
    // itself is branch-free: pieces past the touched frames are clamped onto a valid address and zeroed by a select;
    // in the last, partial chunk (K % 16 != 0) the float4s past the real channels are read from the chunk's last valid
    // float4 instead -- finite data in k-slots whose A column is zero.
    int xfo[GATHER ? 1 : kXV], xv4[GATHER ? 1 : kXV];
    bool xon[GATHER ? 1 : kXV];
    if constexpr (GATHER == 0) {
#pragma unroll
        for (int i = 0; i < kXV; ++i) {
            const int e = threadIdx.x + kBlock * i;
            xon[i] = e < nfr * nv;
            const int ec = xon[i] ? e : 0;
            const int fr = ec / nv;
            xv4[i] = 4 * (ec - fr * nv);
            xfo[i] = fr * d.K * P;                            // (< 2^31: make_dims bounds the tensor)
        }
    }
    const float* Xf0 = X + (size_t)f0 * d.K * P;
    float4 xv[GATHER ? 1 : kXV];
    float av[kAV];
    auto fetch_x = [&](int k0) {
        if constexpr (GATHER == 0) {
            const int live4 = min(kOddKC, d.K - k0) * P - 4;  // last valid float4 start inside a frame chunk (K % 4 == 0)
            const float* src = Xf0 + (size_t)k0 * P;
#pragma unroll
            for (int i = 0; i < kXV; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(src + xfo[i] + min(xv4[i], live4));
                xv[i] = xon[i] ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto deposit_x = [&](int k0, float* dst) {
        if constexpr (GATHER == 0) {
#pragma unroll
            for (int i = 0; i < kXV; ++i)
                if (xon[i]) *reinterpret_cast<float4*>(dst + 4 * (threadIdx.x + kBlock * i)) = xv[i];
        } else {
            // stride-2 gather (one layer per network, small): LDS element (fr, kk, ho, wo) <- X[f0 + fr][k0 + kk][2 ho][2 wo];
            // 8 loads in flight per thread
            const int kc = min(kOddKC, d.K - k0);
            const int HW = d.Hin * d.Win;
            const int total = nfr * fchunk;
            for (int e0 = threadIdx.x; e0 < total; e0 += 8 * kBlock) {
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = e0 + j * kBlock;
                    const int ec = e < total ? e : 0;
                    const int fr = ec / fchunk, r = ec - fr * fchunk;
                    const int kk = r / P, p = r - kk * P;
                    const int ho = p / d.Wo, wo = p - ho * d.Wo;
                    const bool ok = e < total && kk < kc;
                    t[j] = ok ? X[((size_t)(f0 + fr) * d.K + k0 + (ok ? kk : 0)) * HW + 2 * ho * d.Win + 2 * wo] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (e0 + j * kBlock < total) dst[e0 + j * kBlock] = t[j];
            }
        }
    };
    // A: element e = tid + 256 i <-> (kk = e / 64, m = e % 64); address = base + k0 * kstep
    int abase[kAV];
    bool am_ok[kAV];
    const int akstep = d.a_is_mk ? 1 : d.M;
#pragma unroll
    for (int i = 0; i < kAV; ++i) {
        const int e = threadIdx.x + kBlock * i;
        const int kk = e >> 6, gm = m0 + (e & 63);
        am_ok[i] = gm < d.M;
        const int gmc = am_ok[i] ? gm : 0;
        abase[i] = d.a_is_mk ? gmc * d.K + kk : kk * d.M + gmc;
    }
    auto fetch_a = [&](int k0) {                            // As[kk][m], m < 64, zero padded (as AStage)
#pragma unroll
        for (int i = 0; i < kAV; ++i) {
            const int kk = (threadIdx.x + kBlock * i) >> 6;
            const bool ok = am_ok[i] && k0 + kk < d.K;
            av[i] = ok ? A[ok ? abase[i] + k0 * akstep : 0] : 0.f;
        }
    };
    auto deposit_a = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < kAV; ++i) dst[threadIdx.x + kBlock * i] = av[i];
    };

    const int nchunks = (d.K + kOddKC - 1) / kOddKC;
    fetch_a(0);
    fetch_x(0);
    deposit_a(As);
    deposit_x(0, Xs);
#pragma nounroll
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                                    // chunk c is complete; the other buffers are free
        const bool more = c + 1 < nchunks;
        if (more) {
            fetch_a((c + 1) * kOddKC);
            fetch_x((c + 1) * kOddKC);
        }
        const float* as = As + (c & 1) * kOddKC * 64 + l31;
        const float* xs = Xs + (c & 1) * xbuf;
        // all 32 fragment words of the chunk are requested before the first MFMA (one LDS round trip per chunk instead of
        // two per k-step: with a read -> wait -> MFMA sequence per step the loop ran at half the MFMA rate); rows past M
        // are zero columns of A, so the second 32-row block needs no test
        float fa0[kOddKC / 2], fa1[kOddKC / 2], fb[NCB][kOddKC / 2];
#pragma unroll
        for (int s = 0; s < kOddKC / 2; ++s) {
            const int kk = 2 * s + kh;
            fa0[s] = as[kk * 64]; fa1[s] = as[kk * 64 + 32];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) fb[cb][s] = xs[xoff[cb] + kk * P];
        }
#pragma unroll
        for (int s = 0; s < kOddKC / 2; ++s) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s], fb[cb][s], acc[0][cb], 0, 0, 0);
                acc[1][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s], fb[cb][s], acc[1][cb], 0, 0, 0);
            }
        }
        if (more) {
            deposit_a(As + ((c + 1) & 1) * kOddKC * 64);
            deposit_x((c + 1) * kOddKC, Xs + ((c + 1) & 1) * xbuf);
        }
    }

#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (!con[cb]) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (gm >= d.M) continue;
                float o = acc[a][cb][r];
                if (SCATTER == 0) {
                    const size_t at = ((size_t)cf[cb] * d.M + gm) * P + cp[cb];
                    if (R) o += R[at];
                    Y[at] = o;
                } else {
                    const int ho = cp[cb] / d.Wo, wo = cp[cb] - ho * d.Wo;
                    float* q = Y + ((size_t)cf[cb] * d.M + gm) * (d.Hin * d.Win) + 2 * ho * d.Win + 2 * wo;
                    q[0] = o; q[1] = 0.f; q[d.Win] = 0.f; q[d.Win + 1] = 0.f;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Forward / d(input) for bf16 activations on the bf16 MFMA.  Here the reduction index (channels) is the
// STRIDED one, and a bf16 fragment wants 8 consecutive k per lane, so the streamed operand is transposed on
// its way through LDS: each wave loads its 16-channel x 128-pixel chunk as 8-byte cells along the pixels
// (coalesced) and writes it pixel-major -- row rho(4 j + e) = 32 e + j, 16 channels = 32 B + 16 B pad -- so that
// the fragment of column block q, lane (j, g) is ONE aligned 16-byte read of row 32 q + j (conflict-free at
// the 48-byte row stride), while the lane's 4 results of a row are still 4 consecutive pixels (8-byte
// store).  The small operand is rounded to bf16 as it is staged ([m][16 k], same row format): under autocast
// that is the cast torch would have made.  8 MFMAs per 16-channel chunk per wave instead of 64 at fp32.
constexpr int kBC = 16;                   // channels per chunk = K of one bf16 MFMA
constexpr int kRowT = 48;                 // bytes per pixel row / per A row (32 data + 16 pad)

template <int WM, bool A_MK>
__global__ __launch_bounds__(kBlock, (WM == 4 ? 1 : 2)) void k_pw_gemm_bf16(const float* __restrict__ A,
                                                            const __hip_bfloat16* __restrict__ X,
                                                            const __hip_bfloat16* __restrict__ R,
                                                            __hip_bfloat16* __restrict__ Y, PwDims d) {
    constexpr int MT = 64 * WM, WN = 4 / WM;
    __shared__ __attribute__((aligned(16))) char As[2][MT * kRowT];
    __shared__ __attribute__((aligned(16))) char Xs[WN][2][128 * kRowT];   // one tile per column group, shared by its WM waves
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * MT;
    const long long cg = ((long long)blockIdx.x * WN + wn) * 128 + 4 * l31;
    const bool valid = cg < d.ntot;
    const long long cgc = valid ? cg : 0;
    const int f = (int)(cgc / d.P), p = (int)(cgc - (long long)f * d.P);
    const __hip_bfloat16* xp = X + ((size_t)f * d.K) * d.P + p;
    __hip_bfloat16* yp = Y + ((size_t)f * d.M) * d.P + p;

    f32x16 acc[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][q][r] = 0.f;

    // this lane's share of a chunk of X: channels k0 + 2 s + kh for the s = wm, wm + WM, ... of this wave (the WM
    // waves of a column group split the 8 channel pairs), pixels 4 l31 .. + 3
    constexpr int kXS = kBC / 2 / WM;
    auto fetch_x = [&](int k0, uint2 (&v)[kXS]) {
#pragma unroll
        for (int t = 0; t < kXS; ++t) {
            const int k = k0 + 2 * (wm + WM * t) + kh;
            v[t] = (valid && k < d.K) ? *reinterpret_cast<const uint2*>(xp + (size_t)k * d.P) : make_uint2(0u, 0u);
        }
    };
    auto deposit_x = [&](char* tile, const uint2 (&v)[kXS]) {
#pragma unroll
        for (int t = 0; t < kXS; ++t) {
            const int s = wm + WM * t;
            char* q = tile + l31 * kRowT + 2 * (2 * s + kh);                     // row 32 e + j, channel 2 s + kh
            *reinterpret_cast<unsigned short*>(q) = (unsigned short)(v[t].x & 0xffffu);
            *reinterpret_cast<unsigned short*>(q + 32 * kRowT) = (unsigned short)(v[t].x >> 16);
            *reinterpret_cast<unsigned short*>(q + 64 * kRowT) = (unsigned short)(v[t].y & 0xffffu);
            *reinterpret_cast<unsigned short*>(q + 96 * kRowT) = (unsigned short)(v[t].y >> 16);
        }
    };
    // the A chunk: MT x 16, one element per (thread, i)
    constexpr int kPerA = MT * kBC / kBlock;
    // element e of a chunk <-> (m, kk): along the operand's contiguous index, so that the (L2-resident) reads coalesce
    auto a_elem = [&](int e, int& m, int& kk) {
        if (A_MK) { m = e / kBC; kk = e - m * kBC; }
        else { kk = e / MT; m = e - kk * MT; }
    };
    auto fetch_a = [&](int k0, float (&v)[kPerA]) {
#pragma unroll
        for (int i = 0; i < kPerA; ++i) {
            int m, kk;
            a_elem(threadIdx.x + kBlock * i, m, kk);
            const int gk = k0 + kk, gm = m0 + m;
            const bool ok = gk < d.K && gm < d.M;
            const size_t idx = A_MK ? (size_t)gm * d.K + gk : (size_t)gk * d.M + gm;
            v[i] = ok ? A[ok ? idx : 0] : 0.f;
        }
    };
    auto deposit_a = [&](char* as, const float (&v)[kPerA]) {
#pragma unroll
        for (int i = 0; i < kPerA; ++i) {
            int m, kk;
            a_elem(threadIdx.x + kBlock * i, m, kk);
            *reinterpret_cast<unsigned short*>(as + m * kRowT + 2 * kk) =
                __builtin_bit_cast(unsigned short, __float2bfloat16(v[i]));
        }
    };

    uint2 xv[kXS];
    float av[kPerA];
    fetch_x(0, xv);
    fetch_a(0, av);
    deposit_x(Xs[wn][0], xv);
    deposit_a(As[0], av);
    const int nchunks = (d.K + kBC - 1) / kBC;
#pragma nounroll
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                                   // chunk c is in buffers c & 1; the others are free
        const bool more = c + 1 < nchunks;
        if (more) {
            fetch_x((c + 1) * kBC, xv);
            fetch_a((c + 1) * kBC, av);
        }
        const char* xt = Xs[wn][c & 1] + l31 * kRowT + 16 * kh;
        const char* at = As[c & 1] + (wm * 64 + l31) * kRowT + 16 * kh;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(at);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(at + 32 * kRowT);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bf16x8 bq = *reinterpret_cast<const bf16x8*>(xt + 32 * q * kRowT);
            acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq, acc[0][q], 0, 0, 0);
            if (m0 + wm * 64 + 32 < d.M) acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq, acc[1][q], 0, 0, 0);
        }
        if (more) {
            deposit_x(Xs[wn][(c + 1) & 1], xv);
            deposit_a(As[(c + 1) & 1], av);
        }
    }

    if (valid) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (gm < d.M) {
                    float4 o = make_float4(acc[b][0][r], acc[b][1][r], acc[b][2][r], acc[b][3][r]);
                    if (R) {
                        const float4 t = Px4<__hip_bfloat16>::widen(
                            Px4<__hip_bfloat16>::load(R + (yp - Y) + (size_t)gm * d.P));
                        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
                    }
                    Px4<__hip_bfloat16>::store(yp + (size_t)gm * d.P, o);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// d(weight):  dW[m][k] = sum over pixels n = (f, p) of dY[f][m][p] * X[f][k][p].
// The reduction index is the contiguous one, so both MFMA operands are "transposed" with respect to memory
// (lane <-> channel): tiles of 64 channels x 32 pixels are loaded with 16-byte loads along the pixels and
// passed through LDS (column index XOR-swizzled with the row: fragment reads are bank-conflict free).  A wave is an independent
// task -- (64-row block of dY, 64-row block of X, range of pixels) -> a 64 x 64 partial product in 64
// accumulator registers; the 4 waves of a workgroup take 4 consecutive tasks.  Partials go to
// ws[chunk][M][K] (no atomics); k_pw_wgrad_reduce sums the chunks in a fixed order.
constexpr int kNB = 32;                  // pixels per staged tile
__device__ __forceinline__ int tile_at(int row, int col) { return row * kNB + (col ^ (row & (kNB - 1))); }

struct WgDims {
    int F, K, M, P;
    long long ntot;
    int MB, KB;                          // 64-row blocks of dY / X
    int S;                               // pixel chunks
    int chunk;                           // pixels per chunk (multiple of kNB)
    int Cin, Hin, Win, Wo;               // STEM mode only: X is the image [F, Cin, Hin, Win], K = 9 Cin, P = Ho * Wo
    // PRO (training): the X operand is relu?(ka[k] x + kb[k]) of the stored tensor -- the block's relu(bn1(x)) recomputed
    // on the fly, so that the activation conv2 / the shortcut saw in the forward never has to be stored
    const float* ka; const float* kb; int relu_in;
};

// PRO: in the MFMA loop a lane reads X-tile rows l31 and 32 + l31 of its 64-row block only -- two fixed channels -- so
// the prologue is applied to the two fragment values right after the LDS read: 2 fma + 2 max per 4 MFMAs, four
// registers.  Pixels past the range need no care (the dY tile is zero there); rows past K get a = b = 0.
struct RowAffine2 {
    float a0, b0, a1, b1;
    int relu;
    __device__ __forceinline__ void load(const WgDims& d, int row0) {
        const int l31 = threadIdx.x & 31;
        const int r0 = row0 + l31, r1 = row0 + 32 + l31;
        a0 = r0 < d.K ? d.ka[r0] : 0.f;  b0 = r0 < d.K ? d.kb[r0] : 0.f;
        a1 = r1 < d.K ? d.ka[r1] : 0.f;  b1 = r1 < d.K ? d.kb[r1] : 0.f;
        relu = d.relu_in;
    }
    __device__ __forceinline__ void apply(float& v0, float& v1) const {
        v0 = fmaf(a0, v0, b0);
        v1 = fmaf(a1, v1, b1);
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
    }
};

// this lane's position in the pixel stream: n = (f, p); advanced 32 pixels per tile without dividing
struct PixCursor {
    long long n;
    int f, p;
    __device__ __forceinline__ void init(long long n0, int P) {
        n = n0 + 4 * (int)(threadIdx.x & 7);
        f = (int)(n / P);
        p = (int)(n - (long long)f * P);
    }
    __device__ __forceinline__ void advance(int P) {
        n += kNB;
        p += kNB;
        while (p >= P) { p -= P; ++f; }
    }
};

// one lane's share of a 64 x 32 tile: 8 float4 (row = 8 j + lane / 8, pixels 4 (lane % 8) ..+3)
template <typename TT>
__device__ __forceinline__ void wg_fetch(const TT* __restrict__ T, int rows, int r0, int P, const PixCursor& c,
                                         long long nend, float4 (&v)[8], int Cdim) {
    const int lane = threadIdx.x & 63;
    const bool nok = c.n < nend;
    const TT* base = T + (nok ? ((size_t)c.f * Cdim) * P + c.p : 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = r0 + 8 * j + (lane >> 3);
        v[j] = (nok && r < rows) ? Px4<TT>::widen(Px4<TT>::load(base + (size_t)r * P))
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// STEM: the X tile is the im2col of a 3x3 / stride-2 / pad-1 convolution gathered on the fly (row k = (ci, kh, kw),
// column = output pixel): the lane's 4 consecutive output pixels read 4 input pixels 2 apart (cf. k_pw_gemm's STEM)
__device__ __forceinline__ void wg_fetch_stem(const float* __restrict__ X, const WgDims& d, const PixCursor& c,
                                              long long nend, float4 (&v)[8]) {
    const int lane = threadIdx.x & 63;
    const bool nok = c.n < nend;
    const int ho = c.p / d.Wo, wo = c.p - ho * d.Wo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * j + (lane >> 3);
        const int ci = k / 9, r9 = k - 9 * ci, kh3 = r9 / 3, kw3 = r9 - 3 * kh3;
        const int hi = 2 * ho - 1 + kh3, wi = 2 * wo - 1 + kw3;
        const bool ok = nok && k < d.K && hi >= 0 && hi < d.Hin;
        const float* row = X + (((size_t)(nok ? c.f : 0) * d.Cin + (ok ? ci : 0)) * d.Hin + (ok ? hi : 0)) * d.Win;
        float4 t;
        t.x = (ok && wi >= 0) ? row[wi >= 0 ? wi : 0] : 0.f;
        t.y = ok ? row[wi + 2] : 0.f;
        t.z = ok ? row[wi + 4] : 0.f;
        t.w = ok ? row[wi + 6] : 0.f;
        v[j] = t;
    }
}
// stride-2 1x1 convolution: X is [F, K, Hin, Win], column = output pixel (ho, wo) <- input pixel (2 ho, 2 wo)
__device__ __forceinline__ void wg_fetch_s2(const float* __restrict__ X, const WgDims& d, int r0, const PixCursor& c,
                                            long long nend, float4 (&v)[8]) {
    const int lane = threadIdx.x & 63;
    const bool nok = c.n < nend;
    const int ho = c.p / d.Wo, wo = c.p - ho * d.Wo;
    int o1 = 2, o2 = 4, o3 = 6;
    if (d.Wo % 4 != 0) {                                   // a group of 4 output pixels may wrap into the next row (Wo = 14)
        const int p1 = c.p + 1, p2 = c.p + 2, p3 = c.p + 3;
        const int h1 = p1 / d.Wo, h2 = p2 / d.Wo, h3 = p3 / d.Wo;
        o1 = 2 * (h1 - ho) * d.Win + 2 * (p1 - h1 * d.Wo) - 2 * wo;
        o2 = 2 * (h2 - ho) * d.Win + 2 * (p2 - h2 * d.Wo) - 2 * wo;
        o3 = 2 * (h3 - ho) * d.Win + 2 * (p3 - h3 * d.Wo) - 2 * wo;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = r0 + 8 * j + (lane >> 3);
        const bool ok = nok && r < d.K;
        const float* row = X + (((size_t)(nok ? c.f : 0) * d.K + (ok ? r : 0)) * d.Hin + 2 * ho) * d.Win + 2 * wo;
        v[j] = ok ? make_float4(row[0], row[o1], row[o2], row[o3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// planes with P % 4 != 0 (7x7): the lane's 4 consecutive pixels n .. n+3 of the flattened (f, p) index are 4 separate
// words (they may straddle rows and frames).  S2: X is [F, rows, 2 Ho, 2 Wo] read at stride 2 (the 14 -> 7 shortcut).
template <bool S2>
__device__ __forceinline__ void wg_fetch_odd(const float* __restrict__ T, int rows, int r0, const WgDims& d, long long n,
                                             long long nend, float4 (&v)[8]) {
    const int lane = threadIdx.x & 63;
    size_t off[4];
    bool ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long ne = n + e;
        ok[e] = ne < nend;
        const long long nc = ok[e] ? ne : 0;
        const int f = (int)(nc / d.P), p = (int)(nc - (long long)f * d.P);
        if (S2) {
            const int ho = p / d.Wo, wo = p - ho * d.Wo;
            off[e] = (size_t)f * rows * (d.Hin * d.Win) + 2 * ho * d.Win + 2 * wo;
        } else {
            off[e] = (size_t)f * rows * d.P + p;
        }
    }
    const size_t rstride = S2 ? (size_t)d.Hin * d.Win : (size_t)d.P;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = r0 + 8 * j + (lane >> 3);
        const bool rok = r < rows;
        const float* base = T + (rok ? (size_t)r * rstride : 0);
        v[j].x = (rok && ok[0]) ? base[off[0]] : 0.f;
        v[j].y = (rok && ok[1]) ? base[off[1]] : 0.f;
        v[j].z = (rok && ok[2]) ? base[off[2]] : 0.f;
        v[j].w = (rok && ok[3]) ? base[off[3]] : 0.f;
    }
}
__device__ __forceinline__ void wg_deposit(float* tile, const float4 (&v)[8]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = 8 * j + (lane >> 3), col = 4 * (lane & 7);
        tile[tile_at(row, col)] = v[j].x; tile[tile_at(row, col + 1)] = v[j].y;
        tile[tile_at(row, col + 2)] = v[j].z; tile[tile_at(row, col + 3)] = v[j].w;
    }
}

// The MFMA phase of one staged 64 x 32 tile pair: fragments of 4 k-steps are read from LDS together, then their (up to) 16
// MFMAs issue back to back.  A1 / B1: does the second 32-row block of the dY / X tile hold any row -- compile time here,
// so that no branch sits between the reads and the MFMAs (with run-time tests and a read -> wait -> MFMA sequence per
// k-step the loop ran at about a third of the MFMA rate: one LDS round trip per 4 MFMAs, exposed).
template <bool A1, bool B1, bool PRO, typename ProT>
__device__ __forceinline__ void wg_mfma_tile(const float* ta, const float* tb, int rowa, int rowb, int kh, const ProT& pro,
                                             f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int s0 = 0; s0 < kNB / 2; s0 += 4) {
        float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 2 * (s0 + j) + kh;
            a0[j] = ta[tile_at(rowa, col)];
            if (A1) a1[j] = ta[tile_at(rowa + 32, col)];
            b0[j] = tb[tile_at(rowb, col)];
            if (B1) b1[j] = tb[tile_at(rowb + 32, col)]; else b1[j] = 0.f;
            if constexpr (PRO) pro.apply(b0[j], b1[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
            if (B1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[0][1], 0, 0, 0);
            if (A1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
            if (A1 && B1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[1][1], 0, 0, 0);
        }
    }
}
template <bool PRO, typename ProT>
__device__ __forceinline__ void wg_mfma_tile_dyn(bool a1_on, bool b1_on, const float* ta, const float* tb, int rowa, int rowb,
                                                 int kh, const ProT& pro, f32x16 (&acc)[2][2]) {
    if (a1_on && b1_on) wg_mfma_tile<true, true, PRO>(ta, tb, rowa, rowb, kh, pro, acc);
    else if (a1_on) wg_mfma_tile<true, false, PRO>(ta, tb, rowa, rowb, kh, pro, acc);
    else if (b1_on) wg_mfma_tile<false, true, PRO>(ta, tb, rowa, rowb, kh, pro, acc);
    else wg_mfma_tile<false, false, PRO>(ta, tb, rowa, rowb, kh, pro, acc);
}

template <typename T, bool STEM = false, bool S2 = false, bool PRO = false, int ODD = 0>
__global__ __launch_bounds__(kBlock) void k_pw_wgrad(const T* __restrict__ dY, const T* __restrict__ X,
                                                     float* __restrict__ ws, WgDims d) {
    __shared__ float tiles[4][2][64 * kNB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    // workgroup -> (pixel chunk, group of up to 4 (mb, kb) blocks); with fewer than 4 blocks in the group the
    // spare waves split the chunk's pixels and their accumulators are summed through LDS at the end
    const int nmk = d.MB * d.KB;
    const int bpw = nmk < 4 ? nmk : 4;                          // blocks per workgroup: 1, 2 (or 3), 4
    const int sub = 4 / bpw;                                    // pixel sub-chunks per workgroup: 4, 2, 1, 1
    const int groups = (nmk + bpw - 1) / bpw;
    const int chunk = blockIdx.x / groups, grp = blockIdx.x - chunk * groups;
    const int blk = wave % bpw, subc = wave / bpw;
    const int mk = grp * bpw + blk;
    const bool live = mk < nmk && subc < sub;
    const int mkc = live ? mk : 0;
    const int mb = mkc / d.KB, kb = mkc - mb * d.KB;
    const int per = d.chunk / sub;                              // multiple of kNB (make_wg)
    const long long n0 = (long long)chunk * d.chunk + (long long)subc * per;
    long long nend = n0 + per;
    nend = nend < d.ntot ? nend : d.ntot;
    if (!live) nend = n0;                                       // dead wave: zeros
    float* ta = tiles[wave][0];
    float* tb = tiles[wave][1];

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // 32-row blocks that are all padding (72, 144, 288 channels) are skipped; wave-uniform
    const bool a1_on = 64 * mb + 32 < d.M, b1_on = 64 * kb + 32 < d.K;
    float4 va[8], vb[8];
    PixCursor cur;
    cur.init(n0, d.P);
    RowAffine2 pro;
    if constexpr (PRO) pro.load(d, 64 * kb);
    if constexpr (ODD != 0 && std::is_same<T, float>::value) {
        wg_fetch_odd<false>(dY, d.M, 64 * mb, d, cur.n, nend, va);
        wg_fetch_odd<ODD == 2>(X, d.K, 64 * kb, d, cur.n, nend, vb);
    } else {
    wg_fetch(dY, d.M, 64 * mb, d.P, cur, nend, va, d.M);
    if constexpr (STEM && std::is_same<T, float>::value) wg_fetch_stem(X, d, cur, nend, vb);
    else if constexpr (S2 && std::is_same<T, float>::value) wg_fetch_s2(X, d, 64 * kb, cur, nend, vb);
    else wg_fetch(X, d.K, 64 * kb, d.P, cur, nend, vb, d.K);
    }
    const int steps = per / kNB;                                // same for every wave: barriers stay uniform
#pragma nounroll
    for (int it = 0; it < steps; ++it) {
        // (no workgroup barrier: the two tiles are this wave's own, and LDS executes a wave's accesses in order --
        // with barriers here the four waves marched in lock step, MFMA phases and load phases all at once)
        wg_deposit(ta, va);
        wg_deposit(tb, vb);
        if constexpr (ODD != 0 && std::is_same<T, float>::value) {
            cur.n += kNB;
            wg_fetch_odd<false>(dY, d.M, 64 * mb, d, cur.n, nend, va);
            wg_fetch_odd<ODD == 2>(X, d.K, 64 * kb, d, cur.n, nend, vb);
        } else {
        cur.advance(d.P);
        wg_fetch(dY, d.M, 64 * mb, d.P, cur, nend, va, d.M);           // next tile, in flight during the MFMAs
        if constexpr (STEM && std::is_same<T, float>::value) wg_fetch_stem(X, d, cur, nend, vb);
        else if constexpr (S2 && std::is_same<T, float>::value) wg_fetch_s2(X, d, 64 * kb, cur, nend, vb);
        else wg_fetch(X, d.K, 64 * kb, d.P, cur, nend, vb, d.K);
        }
        wg_mfma_tile_dyn<PRO>(a1_on, b1_on, ta, tb, l31, l31, kh, pro, acc);
    }
    // sum the sub-chunk waves of each block into its subc == 0 wave (fixed order), through the tile memory
    __syncthreads();
    if (sub > 1) {
        float* mine = &tiles[wave][0][0];                       // 64 x 64 floats = this wave's two tiles
        if (subc > 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[((a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
        }
        __syncthreads();
        if (subc == 0) {
            for (int o = 1; o < sub; ++o) {
                const float* other = &tiles[blk + o * bpw][0][0];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[a][b][r] += other[((a * 2 + b) * 16 + r) * 64 + lane];
            }
        }
    }
    if (live && subc == 0) {
        float* out = ws + (size_t)chunk * d.M * d.K;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = 64 * mb + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int gk = 64 * kb + 32 * b + l31;
                    if (gm < d.M && gk < d.K) out[(size_t)gm * d.K + gk] = acc[a][b][r];
                }
    }
}


// ---------------------------------------------------------------------------------------------
// d(weight) for wide layers (more than 64 channels on either side): one workgroup = a 128 x 128 block of dW,
// its 4 waves = the four 64 x 64 quadrants, sharing ONE pair of staged tiles (128 channels x 32 pixels each).
// Against four independent wave tasks this halves the loads and LDS writes per MFMA and -- what matters for
// 144 / 288 / 576 channels -- halves how many times each operand row is re-read across the grid (a 64-row
// block of dY is needed by every 64-column block of X: [256,288,14,14] re-read 2 x 5 x 58 MB through L2).
template <typename T, bool PRO = false>
__global__ __launch_bounds__(kBlock) void k_pw_wgrad_wide(const T* __restrict__ dY, const T* __restrict__ X,
                                                          float* __restrict__ ws, WgDims d) {
    __shared__ float tiles[2][128 * kNB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave & 1, wk = wave >> 1;
    const int MT2 = (d.M + 127) / 128, KT2 = (d.K + 127) / 128;
    const int pair = blockIdx.x % (MT2 * KT2), chunk = blockIdx.x / (MT2 * KT2);
    const int mt = pair / KT2, kt = pair - mt * KT2;
    const long long n0 = (long long)chunk * d.chunk;
    long long nend = n0 + d.chunk;
    nend = nend < d.ntot ? nend : d.ntot;
    float* ta = tiles[0];
    float* tb = tiles[1];
    const int rowA = 128 * mt + 64 * wm, rowB = 128 * kt + 64 * wk;       // this wave's quadrant
    const bool a0_on = rowA < d.M, a1_on = rowA + 32 < d.M, b0_on = rowB < d.K, b1_on = rowB + 32 < d.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // a thread's share of a 128-row tile: rows (tid >> 3) + 32 j, pixels 4 (tid & 7) ..+3
    PixCursor cur;
    cur.init(n0, d.P);
    auto fetch = [&](const T* __restrict__ src, int rows, int r0, int Cdim, float4 (&v)[4]) {
        const bool nok = cur.n < nend;
        const T* base = src + (nok ? ((size_t)cur.f * Cdim) * d.P + cur.p : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + (int)(threadIdx.x >> 3) + 32 * j;
            v[j] = (nok && r < rows) ? Px4<T>::widen(Px4<T>::load(base + (size_t)r * d.P)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto deposit = [&](float* tile, const float4 (&v)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (int)(threadIdx.x >> 3) + 32 * j, col = 4 * (int)(threadIdx.x & 7);
            tile[tile_at(row, col)] = v[j].x; tile[tile_at(row, col + 1)] = v[j].y;
            tile[tile_at(row, col + 2)] = v[j].z; tile[tile_at(row, col + 3)] = v[j].w;
        }
    };
    float4 va[4], vb[4];
    RowAffine2 pro;
    if constexpr (PRO) pro.load(d, rowB);
    fetch(dY, d.M, 128 * mt, d.M, va);
    fetch(X, d.K, 128 * kt, d.K, vb);
    const int steps = d.chunk / kNB;
#pragma nounroll
    for (int it = 0; it < steps; ++it) {
        __syncthreads();
        deposit(ta, va);
        deposit(tb, vb);
        __syncthreads();
        cur.advance(d.P);
        fetch(dY, d.M, 128 * mt, d.M, va);
        fetch(X, d.K, 128 * kt, d.K, vb);
        if (a0_on && b0_on) wg_mfma_tile_dyn<PRO>(a1_on, b1_on, ta, tb, 64 * wm + l31, 64 * wk + l31, kh, pro, acc);
    }
    if (a0_on && b0_on) {
        float* out = ws + (size_t)chunk * d.M * d.K;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = rowA + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int gk = rowB + 32 * b + l31;
                    if (gm < d.M && gk < d.K) out[(size_t)gm * d.K + gk] = acc[a][b][r];
                }
    }
}

// planner of the wide kernel: S pixel chunks so that ~768 workgroups run, partials under a quarter of the operands
// Worth it when the 128 x 128 tiles are mostly full: quadrants in use / quadrants launched >= 3/4 (72, 108, 216, 432,
// 576 channels yes; 144 and 288 -- 2.25 tiles per side -- and 54 -> 108 no: measured 216 -> 261 us, 176 -> 177 us,
// 216 -> 270 us against 292 -> 226 us at 72, 135 -> 100 us at 108, 131 -> 98 us at 216)
inline bool use_wide(int M, int K) {
    if (M <= 64 && K <= 64) return false;
    const int q = ((M + 63) / 64) * ((K + 63) / 64), slots = 4 * ((M + 127) / 128) * ((K + 127) / 128);
    return 4 * q >= 3 * slots;
}
inline int make_wg_wide(WgDims& d, int F, int K, int M, int P) {
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P % 4 != 0) return RK_ERR_BAD_DIMS;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P;
    d.ka = d.kb = nullptr; d.relu_in = 0;
    d.MB = (M + 63) / 64; d.KB = (K + 63) / 64;
    const int pairs = ((M + 127) / 128) * ((K + 127) / 128);
    constexpr int want = 768;                                         // workgroups (3 per CU)
    long long S = want / pairs;
    const long long cap = ((long long)(M + K) * d.ntot) / (4LL * M * K);
    if (S > cap) S = cap;
    if (S < 1) S = 1;
    long long chunk = (d.ntot + S - 1) / S;
    chunk = (chunk + kNB - 1) / kNB * kNB;
    d.chunk = (int)chunk;
    d.S = (int)((d.ntot + chunk - 1) / chunk);
    return RK_OK;
}

// ---------------------------------------------------------------------------------------------
// d(weight) for bf16 activations on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  The reduction
// index (pixels) is the contiguous one in NCHW, which is exactly what the bf16 fragments want: lane (i, g) holds
// 8 consecutive k = pixels of row i, a 16-byte read.  Same decomposition, staging and reductions as k_pw_wgrad;
// tiles stay bf16 in LDS (64 rows x 32 pixels, row stride 80 B: the ds_read_b128 fragment reads are
// conflict-free), 8 MFMAs per 32-pixel tile instead of 64.
constexpr int kRowB = 80;                 // bytes per tile row (64 data + 16 pad)

__device__ __forceinline__ void wgb_fetch(const __hip_bfloat16* __restrict__ T, int rows, int r0, int P,
                                          const PixCursor& c, long long nend, uint2 (&v)[8], int Cdim) {
    const int lane = threadIdx.x & 63;
    const bool nok = c.n < nend;
    const __hip_bfloat16* base = T + (nok ? ((size_t)c.f * Cdim) * P + c.p : 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = r0 + 8 * j + (lane >> 3);
        v[j] = (nok && r < rows) ? *reinterpret_cast<const uint2*>(base + (size_t)r * P) : make_uint2(0u, 0u);
    }
}
__device__ __forceinline__ void wgb_deposit(char* tile, const uint2 (&v)[8]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        *reinterpret_cast<uint2*>(tile + (8 * j + (lane >> 3)) * kRowB + 8 * (lane & 7)) = v[j];
}

__global__ __launch_bounds__(kBlock) void k_pw_wgrad_bf16(const __hip_bfloat16* __restrict__ dY,
                                                          const __hip_bfloat16* __restrict__ X,
                                                          float* __restrict__ ws, WgDims d) {
    __shared__ __attribute__((aligned(16))) char tiles[4][2][64 * kRowB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int l31 = lane & 31, kh = lane >> 5;
    const int nmk = d.MB * d.KB;
    const int bpw = nmk < 4 ? nmk : 4;
    const int sub = 4 / bpw;
    const int groups = (nmk + bpw - 1) / bpw;
    const int chunk = blockIdx.x / groups, grp = blockIdx.x - chunk * groups;
    const int blk = wave % bpw, subc = wave / bpw;
    const int mk = grp * bpw + blk;
    const bool live = mk < nmk && subc < sub;
    const int mkc = live ? mk : 0;
    const int mb = mkc / d.KB, kb = mkc - mb * d.KB;
    const int per = d.chunk / sub;
    const long long n0 = (long long)chunk * d.chunk + (long long)subc * per;
    long long nend = n0 + per;
    nend = nend < d.ntot ? nend : d.ntot;
    if (!live) nend = n0;
    char* ta = tiles[wave][0];
    char* tb = tiles[wave][1];

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // two tiles in flight (registers): with 8 MFMAs per tile the compute phase is far too short to cover a
    // global-load round trip, so the loop is unrolled by two over two register sets
    const bool a1_on = 64 * mb + 32 < d.M, b1_on = 64 * kb + 32 < d.K;   // skip all-padding 32-row blocks
    uint2 va0[8], vb0[8], va1[8], vb1[8];
    PixCursor cur;
    cur.init(n0, d.P);
    wgb_fetch(dY, d.M, 64 * mb, d.P, cur, nend, va0, d.M);
    wgb_fetch(X, d.K, 64 * kb, d.P, cur, nend, vb0, d.K);
    cur.advance(d.P);
    wgb_fetch(dY, d.M, 64 * mb, d.P, cur, nend, va1, d.M);
    wgb_fetch(X, d.K, 64 * kb, d.P, cur, nend, vb1, d.K);
    const int steps = per / kNB;
    auto consume = [&]() {
#pragma unroll
        for (int s = 0; s < kNB / 16; ++s) {                        // 16 pixels per MFMA
            const int off = 32 * s + 16 * kh;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ta + l31 * kRowB + off);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ta + (32 + l31) * kRowB + off);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(tb + l31 * kRowB + off);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(tb + (32 + l31) * kRowB + off);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            if (b1_on) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            if (a1_on) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            if (a1_on && b1_on) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
#pragma nounroll
    for (int it = 0; it < steps; it += 2) {                        // (tiles past the range are zeros)
        __syncthreads();
        wgb_deposit(ta, va0);
        wgb_deposit(tb, vb0);
        __syncthreads();
        cur.advance(d.P);
        wgb_fetch(dY, d.M, 64 * mb, d.P, cur, nend, va0, d.M);      // tile it + 2
        wgb_fetch(X, d.K, 64 * kb, d.P, cur, nend, vb0, d.K);
        consume();
        __syncthreads();
        wgb_deposit(ta, va1);
        wgb_deposit(tb, vb1);
        __syncthreads();
        cur.advance(d.P);
        wgb_fetch(dY, d.M, 64 * mb, d.P, cur, nend, va1, d.M);      // tile it + 3
        wgb_fetch(X, d.K, 64 * kb, d.P, cur, nend, vb1, d.K);
        consume();
    }
    // sum the sub-chunk waves of each block into its subc == 0 wave, one sub-chunk at a time through the tile
    // memory (bpw blocks x 16 KB fit the 40 KB of tiles only for one sub-chunk index at once)
    __syncthreads();
    for (int o = 1; o < sub; ++o) {
        float* buf = reinterpret_cast<float*>(&tiles[0][0][0]) + blk * (64 * 64);
        if (subc == o) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) buf[((a * 2 + b) * 16 + r) * 64 + lane] = acc[a][b][r];
        }
        __syncthreads();
        if (subc == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += buf[((a * 2 + b) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (live && subc == 0) {
        float* out = ws + (size_t)chunk * d.M * d.K;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = 64 * mb + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int gk = 64 * kb + 32 * b + l31;
                    if (gm < d.M && gk < d.K) out[(size_t)gm * d.K + gk] = acc[a][b][r];
                }
    }
}

// out[part][i] = sum of in[c][i] over the chunks c of this part (c = part, part + parts, ...): coalesced in i,
// fixed order.  One launch (parts = 1) up to kRedDirect chunks; beyond, twice: S chunks -> kRed parts -> 1.
constexpr int kRed = 32;
constexpr int kRedDirect = 128;           // up to this many chunks one launch sums them (two launches: +8 us per d(weight))
__global__ __launch_bounds__(kBlock) void k_pw_wgrad_reduce(const float* __restrict__ in, float* __restrict__ out,
                                                            int MK, int S, int parts) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int part = blockIdx.y;
    if (i >= MK) return;
    float s = 0.f;
#pragma unroll 8
    for (int c = part; c < S; c += parts) s += in[(size_t)c * MK + i];
    out[(size_t)part * MK + i] = s;
}

inline int make_wg(WgDims& d, int F, int K, int M, int P, bool odd = false) {
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || (P % 4 != 0 && !odd)) return RK_ERR_BAD_DIMS;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P;
    d.ka = d.kb = nullptr; d.relu_in = 0;
    d.MB = (M + 63) / 64; d.KB = (K + 63) / 64;
    // ~1536 wave tasks (6 per CU), but keep the partials (S * M * K floats, written and read once) under a
    // quarter of the operands' bytes
    const int nmk = d.MB * d.KB;
    // wave tasks: all 2 048 wave slots for layers of up to 4 (mb, kb) blocks (54 -> 54 at 56x56: 93 -> 80 us); with many
    // blocks per chunk (288 -> 288: 25) more chunks only add partials (142 -> 155 us)
    const int want_tasks = nmk <= 4 ? 2048 : 1536;
    long long S = want_tasks / (nmk < 4 ? 4 : nmk);                   // one partial per workgroup-chunk
    const long long cap = ((long long)(M + K) * d.ntot) / (4LL * M * K);
    if (S > cap) S = cap;
    if (S < 1) S = 1;
    long long chunk = (d.ntot + S - 1) / S;
    chunk = (chunk + 4 * kNB - 1) / (4 * kNB) * (4 * kNB);          // splits into 1, 2 or 4 sub-chunks of whole tiles
    d.chunk = (int)chunk;
    d.S = (int)((d.ntot + chunk - 1) / chunk);
    return RK_OK;
}

}  // namespace pw
}  // namespace rk

using namespace rk;
using namespace rk::pw;

namespace {

template <typename T>
int pw_gemm(const float* A, const void* X_, const void* R_, void* Y_, int F, int K, int M, int P, int a_is_mk,
            rk_stream_t stream_, const PwFuse* fuse = nullptr, const PwTrain* train = nullptr, int epi = 0) {
    const T* X = (const T*)X_; const T* R = (const T*)R_; T* Y = (T*)Y_;
    if (!A || !X || !Y) return RK_ERR_NULL_POINTER;
    const uintptr_t am = 4 * sizeof(T) - 1;
    if (R && ((uintptr_t)R & am)) return RK_ERR_BAD_DIMS;
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P % 4 != 0 || K % 2 != 0) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)X & am) || ((uintptr_t)Y & am)) return RK_ERR_BAD_DIMS;
    if constexpr (std::is_same<T, float>::value) {
        static const bool trace = getenv("RK_PW_TRACE") != nullptr;      // debugging aid: one line per call
        if (trace) fprintf(stderr, "pw_gemm F=%d K=%d M=%d P=%d mk=%d epi=%d R=%d pro=%d outaff=%d p3=%d p4=%d p2=%d\n", F, K, M, P, a_is_mk, epi,
                           R != nullptr, fuse && fuse->ka, fuse && fuse->ma, pw3::tiles(F, K, M, P), (int)pw4::tiles(F, K, M, P, epi, R != nullptr, false),
                           (int)pw2::gemm_wanted(K, M, P, a_is_mk, A));
        // second generation (rk_pw2.hip) where it is ahead; its training epilogues use 64-column tiles
        // (rk_pw_gemm_tiles() tells the caller which count to allocate)
        // third kernel (rk_pw3.hip): the LDS-tiled GEMM of the 288-row layers, one workgroup per CU
        if (pw3::tiles(F, K, M, P) > 0 && !(fuse && fuse->ma) && ((uintptr_t)A & 15) == 0) {
            const pw2::GFuse f3 = fuse ? pw2::GFuse{fuse->ka, fuse->kb, nullptr, nullptr, fuse->relu_in, 0}
                                       : pw2::GFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
            const pw2::GTrain t3 = train ? pw2::GTrain{train->stats, train->bred, train->bx, train->bpack, train->J}
                                         : pw2::GTrain{nullptr, nullptr, nullptr, nullptr, 0};
            const int rc = pw3::gemm(A, X, R, Y, F, K, M, P, a_is_mk, &f3, &t3, epi, (hipStream_t)stream_);
            if (rc != RK_ERR_UNSUPPORTED || epi) return rc;
        }
        // streaming kernel of the shallow layers (rk_pw4.hip): operand in registers, records through a per-wave LDS ring;
        // its tile records are the 64-column ones of rk_pw2.hip
        if (pw4::tiles(F, K, M, P, epi, R != nullptr, false) > 0 && !(epi && fuse && fuse->ma) && !(epi == 2 && R)) {
            const pw2::GFuse f4 = fuse ? pw2::GFuse{fuse->ka, fuse->kb, fuse->ma, fuse->mb, fuse->relu_in, fuse->relu_out}
                                       : pw2::GFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
            const pw2::GTrain t4 = train ? pw2::GTrain{train->stats, train->bred, train->bx, train->bpack, train->J}
                                         : pw2::GTrain{nullptr, nullptr, nullptr, nullptr, 0};
            const int rc = pw4::gemm(A, X, R, Y, F, K, M, P, a_is_mk, &f4, &t4, epi, (hipStream_t)stream_, false);
            if (rc != RK_ERR_UNSUPPORTED || epi) return rc;
        }
        if (pw2::gemm_wanted(K, M, P, a_is_mk, A)) {
            const pw2::GFuse f2 = fuse ? pw2::GFuse{fuse->ka, fuse->kb, fuse->ma, fuse->mb, fuse->relu_in, fuse->relu_out}
                                       : pw2::GFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
            const pw2::GTrain t2 = train ? pw2::GTrain{train->stats, train->bred, train->bx, train->bpack, train->J}
                                         : pw2::GTrain{nullptr, nullptr, nullptr, nullptr, 0};
            if (epi && (long long)t2.J != ((long long)F * P + pw2::kTileCols - 1) / pw2::kTileCols) return RK_ERR_BAD_DIMS;
            const int rc = pw2::gemm(A, X, R, Y, F, K, M, P, a_is_mk, &f2, &t2, epi, (hipStream_t)stream_, nullptr);
            if (rc != RK_ERR_UNSUPPORTED) return rc;
            if (epi) return rc;                              // (the tile count was promised for this generation)
        }
        // A training epilogue whose caller allocated 64-column tile records (what rk_pw_gemm_tiles() promises whenever
        // rk_pw4.hip takes the shape) but which no second-generation kernel WANTED -- rk_pw4.hip declines some (epilogue,
        // residual) pairs and rk_pw2.hip was counted on for them; with RK_PW2=0, or a shape rk_pw2.hip serves only in its
        // slow operand mode, nobody was left and the first-generation kernel below (128-column records) answered
        // RK_ERR_BAD_DIMS in the middle of a train step (round-4 advisor finding).  The promise is kept: such a call goes to
        // whichever 64-column kernel CAN take it, wanted or not.
        const long long fp = (long long)F * P;
        if (epi && train && fp > 64 && (long long)train->J == (fp + 63) / 64) {
            const pw2::GFuse f2 = fuse ? pw2::GFuse{fuse->ka, fuse->kb, fuse->ma, fuse->mb, fuse->relu_in, fuse->relu_out}
                                       : pw2::GFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
            const pw2::GTrain t2{train->stats, train->bred, train->bx, train->bpack, train->J};
            if (!(fuse && fuse->ma) && !(epi == 2 && R)) {
                const int rc = pw4::gemm(A, X, R, Y, F, K, M, P, a_is_mk, &f2, &t2, epi, (hipStream_t)stream_, true);
                if (rc != RK_ERR_UNSUPPORTED) return rc;
            }
            return pw2::gemm(A, X, R, Y, F, K, M, P, a_is_mk, &f2, &t2, epi, (hipStream_t)stream_, nullptr);
        }
    }
    PwDims d;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P; d.a_is_mk = a_is_mk;
    d.Cin = d.Hin = d.Win = d.Wo = 0;
    // rows per workgroup tile = 64 wm.  Above 128 rows 64-row tiles win although the streamed operand is then
    // re-read once per tile (L2 / Infinity Cache absorb it; 288 rows: 101 us against 141 / 179 us with 128 / 256-row
    // tiles, which also pad 288 to 384 / 512)
    // (bf16, 288 rows, Large-AQ train step: 64-row tiles 45.1 ms, 128-row 47.1, 256-row 57.7: workgroup count and the
    // per-chunk latency chain, not the re-read of the streamed operand, bound the bf16 kernel)
    const int wm = M <= 64 ? 1 : (M <= 128 ? 2 : 1);
    d.WM = wm; d.WN = 4 / wm;
    const int mt = 64 * wm;
    const dim3 grid((unsigned)((d.ntot + 128 * d.WN - 1) / (128 * d.WN)), (unsigned)((M + mt - 1) / mt)), block(kBlock);
    hipStream_t stream = (hipStream_t)stream_;
    // chunk of 12 or 16 (2 waves per SIMD; 18 needs too many registers): the one that pads K less
    const int kc = ((K + 11) / 12 * 12 <= (K + 15) / 16 * 16) ? 12 : 16;
    if constexpr (std::is_same<T, __hip_bfloat16>::value) {
        if (!(fuse && (fuse->ka || fuse->ma))) {                    // bf16 activations: the bf16-MFMA kernel
#define RK_PW_B(WMV) do { if (a_is_mk) hipLaunchKernelGGL((k_pw_gemm_bf16<WMV, true>), grid, block, 0, stream, A, X, R, Y, d); \
                          else hipLaunchKernelGGL((k_pw_gemm_bf16<WMV, false>), grid, block, 0, stream, A, X, R, Y, d); } while (0)
            if (wm == 1) RK_PW_B(1);
            else if (wm == 2) RK_PW_B(2);
            else RK_PW_B(4);
#undef RK_PW_B
            return launch_status();
        }
    }
    const bool fused = fuse && (fuse->ka || fuse->ma);
    if (fused && fuse->ka && (K + 15) / 16 * 16 > kProK) return RK_ERR_BAD_DIMS;      // prologue table in LDS
    PwFuse fz = fused ? *fuse : PwFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
    PwTrain tr = train ? *train : PwTrain{nullptr, nullptr, nullptr, nullptr, 0};
    if constexpr (std::is_same<T, float>::value) {
        if (epi) {                                           // training epilogues (fp32): statistics of Y / BN-backward sums
            if (epi == 1 ? !tr.stats : !(tr.bred && tr.bx && tr.bpack)) return RK_ERR_NULL_POINTER;
            if (fz.ma || (long long)tr.J != (d.ntot + 127) / 128) return RK_ERR_BAD_DIMS;
            if (epi == 2 && (((uintptr_t)tr.bx & am) || ((uintptr_t)tr.bpack & 15))) return RK_ERR_BAD_DIMS;
#define RK_PW_EP(WMV, KCV, FU, EP) hipLaunchKernelGGL((k_pw_gemm<float, WMV, KCV, FU, false, 0, EP>), grid, block, 0, stream, A, X, R, Y, d, fz, tr)
#define RK_PW_EK(WMV, FU, EP) do { if (kc == 12) RK_PW_EP(WMV, 12, FU, EP); else RK_PW_EP(WMV, 16, FU, EP); } while (0)
#define RK_PW_EW(FU, EP) do { if (wm == 1) RK_PW_EK(1, FU, EP); else RK_PW_EK(2, FU, EP); } while (0)
            if (epi == 1) { if (fused) RK_PW_EW(true, 1); else RK_PW_EW(false, 1); }
            else RK_PW_EW(false, 2);
#undef RK_PW_EW
#undef RK_PW_EK
#undef RK_PW_EP
            return launch_status();
        }
    }
#define RK_PW_GO(WMV, KCV) do { if (fused) hipLaunchKernelGGL((k_pw_gemm<T, WMV, KCV, true>), grid, block, 0, stream, A, X, R, Y, d, fz, tr); \
                                else hipLaunchKernelGGL((k_pw_gemm<T, WMV, KCV, false>), grid, block, 0, stream, A, X, R, Y, d, fz, tr); } while (0)
#define RK_PW_KC(WMV) do { if (kc == 12) RK_PW_GO(WMV, 12); else RK_PW_GO(WMV, 16); } while (0)
    if (wm == 1) RK_PW_KC(1);
    else if (wm == 2) RK_PW_KC(2);
    else RK_PW_KC(4);
#undef RK_PW_KC
#undef RK_PW_GO
    return launch_status();
}

template <typename T>
int pw_wgrad(const void* dY_, const void* X_, float* dW, int F, int K, int M, int P, void* ws, size_t ws_bytes,
             rk_stream_t stream_, const float* ka = nullptr, const float* kb = nullptr, int relu_in = 0) {
    const T* dY = (const T*)dY_; const T* X = (const T*)X_;
    if (!dY || !X || !dW) return RK_ERR_NULL_POINTER;
    if constexpr (std::is_same<T, float>::value) {
        if (pw2::wgrad_wanted(P) && F > 0 && K > 0 && M > 0) {         // second generation: LDS-DMA stages, 16 x 16 x 4 MFMA
            const int rc = pw2::wgrad((const float*)dY_, (const float*)X_, dW, F, K, M, P, ws, ws_bytes, ka, kb, relu_in,
                                      (hipStream_t)stream_, nullptr);
            if (rc != RK_ERR_UNSUPPORTED) return rc;
        }
    }
    WgDims d;
    // the wide (shared-tile) kernel for > 64 channels; bf16 activations keep the bf16-MFMA kernel
    const bool wide = use_wide(M, K) && !std::is_same<T, __hip_bfloat16>::value;
    if (int rc = wide ? make_wg_wide(d, F, K, M, P) : make_wg(d, F, K, M, P)) return rc;
    const uintptr_t am = 4 * sizeof(T) - 1;
    if (((uintptr_t)X & am) || ((uintptr_t)dY & am)) return RK_ERR_BAD_DIMS;
    if (!ws || ws_bytes < (size_t)(d.S + kRed) * M * K * sizeof(float)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int nmk = d.MB * d.KB, bpw = nmk < 4 ? nmk : 4, groups = (nmk + bpw - 1) / bpw;
    float* part = (float*)ws;
    float* part2 = part + (size_t)d.S * M * K;
    const int MK = M * K;
    const unsigned gi = (unsigned)((MK + kBlock - 1) / kBlock);
    const bool pro = ka && kb;
    d.ka = ka; d.kb = kb; d.relu_in = relu_in;
    if (pro && std::is_same<T, __hip_bfloat16>::value) return RK_ERR_BAD_DIMS;      // prologue: fp32 activations only
    if (wide) {
        const int pairs = ((M + 127) / 128) * ((K + 127) / 128);
        if (pro) hipLaunchKernelGGL((k_pw_wgrad_wide<T, true>), dim3((unsigned)(d.S * pairs)), dim3(kBlock), 0, stream, dY, X, part, d);
        else hipLaunchKernelGGL((k_pw_wgrad_wide<T>), dim3((unsigned)(d.S * pairs)), dim3(kBlock), 0, stream, dY, X, part, d);
    } else if (pro) {
        hipLaunchKernelGGL((k_pw_wgrad<T, false, false, true>), dim3((unsigned)(d.S * groups)), dim3(kBlock), 0, stream, dY, X, part, d);
    } else if constexpr (std::is_same<T, __hip_bfloat16>::value) {
        hipLaunchKernelGGL(k_pw_wgrad_bf16, dim3((unsigned)(d.S * groups)), dim3(kBlock), 0, stream, dY, X, part, d);
    } else {
        hipLaunchKernelGGL((k_pw_wgrad<T>), dim3((unsigned)(d.S * groups)), dim3(kBlock), 0, stream, dY, X, part, d);
    }
    if (d.S > kRedDirect) {
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, kRed), dim3(kBlock), 0, stream, (const float*)part, part2, MK, d.S, kRed);
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, 1), dim3(kBlock), 0, stream, (const float*)part2, dW, MK, kRed, 1);
    } else {
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, 1), dim3(kBlock), 0, stream, (const float*)part, dW, MK, d.S, 1);
    }
    return launch_status();
}

}  // namespace

extern "C" {

size_t rk_pw_wgrad_workspace_bytes(int F, int K, int M, int P) {
    WgDims d, w;
    if (make_wg(d, F, K, M, P) || make_wg_wide(w, F, K, M, P)) return 0;
    const size_t v1 = (size_t)((d.S > w.S ? d.S : w.S) + kRed) * M * K * sizeof(float);      // whichever kernel is chosen
    const size_t v2 = pw2::wgrad_workspace_bytes(F, K, M, P);
    return v1 > v2 ? v1 : v2;
}

// Y[f] = A X[f] (+ R[f]).  a_is_mk != 0: A is [M][K] row-major; else [K][M] (always fp32).  X [F,K,P], Y / R [F,M,P]
// fp32 or bf16, P % 4 == 0.  R may be NULL (no residual) and may alias Y.
int rk_pw_gemm_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P,
                   int a_is_mk, rk_stream_t stream) {
    return pw_gemm<float>(A, X, R, Y, F, K, M, P, a_is_mk, stream);
}
int rk_pw_gemm_bf16(const float* A, const void* X, const void* R, void* Y, int F, int K, int M, int P, int a_is_mk,
                    rk_stream_t stream) {
    return pw_gemm<__hip_bfloat16>(A, X, R, Y, F, K, M, P, a_is_mk, stream);
}
// 3x3 / stride 2 / pad 1 convolution, no bias (the stem): W [Cout][Cin][3][3], X [F, Cin, Hin, Win], Y [F, Cout, Ho, Wo],
// Ho = Hin / 2, Wo = Win / 2 (Hin even, Win % 8 == 0, 9 Cin <= 64).  Same GEMM, im2col gathered on the fly.
static int stem_conv(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                     rk_stream_t stream_, float4* stats, int J) {
    if (!W || !X || !Y) return RK_ERR_NULL_POINTER;
    if (F <= 0 || Cin <= 0 || Cout <= 0 || Hin <= 0 || Win <= 0 || Hin % 2 || Win % 8 || 9 * Cin > 64) return RK_ERR_BAD_DIMS;
    if ((uintptr_t)Y & 15) return RK_ERR_BAD_DIMS;
    PwDims d;
    d.F = F; d.K = 9 * Cin; d.M = Cout; d.Cin = Cin; d.Hin = Hin; d.Win = Win; d.Wo = Win / 2;
    d.P = (Hin / 2) * d.Wo; d.ntot = (long long)F * d.P; d.a_is_mk = 1;
    const int wm = Cout <= 64 ? 1 : (Cout <= 128 ? 2 : 4);
    d.WM = wm; d.WN = 4 / wm;
    const int mt = 64 * wm;
    const dim3 grid((unsigned)((d.ntot + 128 * d.WN - 1) / (128 * d.WN)), (unsigned)((Cout + mt - 1) / mt)), block(kBlock);
    hipStream_t stream = (hipStream_t)stream_;
    const PwFuse fz{nullptr, nullptr, nullptr, nullptr, 0, 0};
    const float* R = nullptr;
    if (stats) {                                          // + the statistics of Y for the first block's bn1 (PwTrain)
        if ((long long)J * 128 < d.ntot) return RK_ERR_BAD_DIMS;
        const PwTrain tr{stats, nullptr, nullptr, nullptr, J};
        if (wm == 1) hipLaunchKernelGGL((k_pw_gemm<float, 1, 16, false, true, 0, 1>), grid, block, 0, stream, W, X, R, Y, d, fz, tr);
        else if (wm == 2) hipLaunchKernelGGL((k_pw_gemm<float, 2, 12, false, true, 0, 1>), grid, block, 0, stream, W, X, R, Y, d, fz, tr);
        else return RK_ERR_BAD_DIMS;
        return launch_status();
    }
    const PwTrain tr{nullptr, nullptr, nullptr, nullptr, 0};
    if (wm == 1) hipLaunchKernelGGL((k_pw_gemm<float, 1, 16, false, true>), grid, block, 0, stream, W, X, R, Y, d, fz, tr);
    else if (wm == 2) hipLaunchKernelGGL((k_pw_gemm<float, 2, 12, false, true>), grid, block, 0, stream, W, X, R, Y, d, fz, tr);
    else hipLaunchKernelGGL((k_pw_gemm<float, 4, 16, false, true>), grid, block, 0, stream, W, X, R, Y, d, fz, tr);
    return launch_status();
}
int rk_stem_conv3x3s2_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                          rk_stream_t stream) {
    return stem_conv(W, X, Y, F, Cin, Cout, Hin, Win, stream, nullptr, 0);
}
// the same convolution + the tile statistics of Y for the BatchNorm that follows (EPI = 1 of k_pw_gemm; `stats` holds
// Cout x rk_pw_tiles(F, Ho * Wo) float4; Cout <= 128)
int rk_stem_conv3x3s2_stats_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                                void* stats, int tiles, rk_stream_t stream) {
    if (!stats) return RK_ERR_NULL_POINTER;
    return stem_conv(W, X, Y, F, Cin, Cout, Hin, Win, stream, (float4*)stats, tiles);
}
// d(weight) of the same stem convolution: dW [Cout][Cin][3][3] = sum over frames and output pixels of dY x im2col(X);
// dY [F, Cout, Ho, Wo], X [F, Cin, Hin, Win]; ws of rk_pw_wgrad_workspace_bytes(F, 9 Cin, Cout, Ho * Wo) bytes.
int rk_stem_wgrad3x3s2_f32(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win,
                           void* ws, size_t ws_bytes, rk_stream_t stream_) {
    if (!dY || !X || !dW) return RK_ERR_NULL_POINTER;
    if (F <= 0 || Cin <= 0 || Cout <= 0 || Hin <= 0 || Win <= 0 || Hin % 2 || Win % 8 || 9 * Cin > 64) return RK_ERR_BAD_DIMS;
    if ((uintptr_t)dY & 15) return RK_ERR_BAD_DIMS;
    const int K = 9 * Cin, M = Cout, P = (Hin / 2) * (Win / 2);
    WgDims d;
    if (int rc = make_wg(d, F, K, M, P)) return rc;
    d.Cin = Cin; d.Hin = Hin; d.Win = Win; d.Wo = Win / 2;
    if (!ws || ws_bytes < (size_t)(d.S + kRed) * M * K * sizeof(float)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int nmk = d.MB * d.KB, bpw = nmk < 4 ? nmk : 4, groups = (nmk + bpw - 1) / bpw;
    float* part = (float*)ws;
    float* part2 = part + (size_t)d.S * M * K;
    const int MK = M * K;
    const unsigned gi = (unsigned)((MK + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((k_pw_wgrad<float, true>), dim3((unsigned)(d.S * groups)), dim3(kBlock), 0, stream, dY, X, part, d);
    if (d.S > kRedDirect) {
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, kRed), dim3(kBlock), 0, stream, (const float*)part, part2, MK, d.S, kRed);
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, 1), dim3(kBlock), 0, stream, (const float*)part2, dW, MK, kRed, 1);
    } else {
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, 1), dim3(kBlock), 0, stream, (const float*)part, dW, MK, d.S, 1);
    }
    return launch_status();
}
// 1x1 / stride-2 convolution, no bias (the projecting shortcuts): forward (mode 1: X [F,K,Hin,Win] -> Y [F,M,Ho,Wo]) and
// d(input) (mode 2: X = dY [F,K=Cout,Ho,Wo] -> Y = dX [F,M=Cin,Hin,Win], every element written).  Hin, Win even,
// Wo % 4 == 0, K even.
static int pw_s2(const float* A, const float* X, float* Y, int F, int K, int M, int Hin, int Win, int mode, int a_is_mk,
                 rk_stream_t stream_, const PwFuse* fuse = nullptr) {
    if (!A || !X || !Y) return RK_ERR_NULL_POINTER;
    if (F <= 0 || K <= 0 || M <= 0 || Hin <= 0 || Win <= 0 || Hin % 2 || Win % 2 || K % 2) return RK_ERR_BAD_DIMS;
    if (((Hin / 2) * (Win / 2)) % 4) return RK_ERR_BAD_DIMS;          // output planes of whole 4-pixel groups (Win % 8 == 0: one row each)
    if (((uintptr_t)X & 15) || ((uintptr_t)Y & 15)) return RK_ERR_BAD_DIMS;
    PwDims d;
    d.F = F; d.K = K; d.M = M; d.Cin = 0; d.Hin = Hin; d.Win = Win; d.Wo = Win / 2;
    d.P = (Hin / 2) * d.Wo; d.ntot = (long long)F * d.P; d.a_is_mk = a_is_mk;
    const int wm = M <= 64 ? 1 : (M <= 128 ? 2 : 1);
    d.WM = wm; d.WN = 4 / wm;
    const int mt = 64 * wm;
    const dim3 grid((unsigned)((d.ntot + 128 * d.WN - 1) / (128 * d.WN)), (unsigned)((M + mt - 1) / mt)), block(kBlock);
    hipStream_t stream = (hipStream_t)stream_;
    const PwFuse fz = fuse ? *fuse : PwFuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (fuse && fuse->ka && (K + 15) / 16 * 16 > kProK) return RK_ERR_BAD_DIMS;
    const float* R = nullptr;
    const PwTrain tr{nullptr, nullptr, nullptr, nullptr, 0};
#define RK_S2_GO(WMV, MODE, FU) hipLaunchKernelGGL((k_pw_gemm<float, WMV, 12, FU, false, MODE>), grid, block, 0, stream, A, X, R, Y, d, fz, tr)
    if (mode == 1 && fuse) { if (wm == 1) RK_S2_GO(1, 1, true); else RK_S2_GO(2, 1, true); }
    else if (mode == 1) { if (wm == 1) RK_S2_GO(1, 1, false); else RK_S2_GO(2, 1, false); }
    else { if (wm == 1) RK_S2_GO(1, 2, false); else RK_S2_GO(2, 2, false); }
#undef RK_S2_GO
    return launch_status();
}
// inference: the same forward with relu?(ka[k] x + kb[k]) (the block's eval-mode bn1 + ReLU) applied to the streamed
// operand on the fly
int rk_pw_s2_forward_fused_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                               const float* ka, const float* kb, int relu_in, rk_stream_t stream) {
    if (!ka || !kb) return RK_ERR_NULL_POINTER;
    const PwFuse fz{ka, kb, nullptr, nullptr, relu_in, 0};
    return pw_s2(W, X, Y, F, Cin, Cout, Hin, Win, 1, 1, stream, &fz);
}
int rk_pw_s2_forward_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                         rk_stream_t stream) {
    return pw_s2(W, X, Y, F, Cin, Cout, Hin, Win, 1, 1, stream);
}
int rk_pw_s2_dgrad_f32(const float* W, const float* dY, float* dX, int F, int Cin, int Cout, int Hin, int Win,
                       rk_stream_t stream) {
    return pw_s2(W, dY, dX, F, Cout, Cin, Hin, Win, 2, 0, stream);     // W read as [K=Cout][M=Cin]
}
static int pw_s2_wgrad(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win, void* ws,
                       size_t ws_bytes, rk_stream_t stream_, const float* ka, const float* kb, int relu_in) {
    if (!dY || !X || !dW) return RK_ERR_NULL_POINTER;
    if (F <= 0 || Cin <= 0 || Cout <= 0 || Hin <= 0 || Win <= 0 || Hin % 2 || Win % 2) return RK_ERR_BAD_DIMS;
    if ((uintptr_t)dY & 15) return RK_ERR_BAD_DIMS;
    const int K = Cin, M = Cout, P = (Hin / 2) * (Win / 2);
    WgDims d;
    if (int rc = make_wg(d, F, K, M, P)) return rc;
    d.Cin = Cin; d.Hin = Hin; d.Win = Win; d.Wo = Win / 2;
    if (!ws || ws_bytes < (size_t)(d.S + kRed) * M * K * sizeof(float)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int nmk = d.MB * d.KB, bpw = nmk < 4 ? nmk : 4, groups = (nmk + bpw - 1) / bpw;
    float* part = (float*)ws;
    float* part2 = part + (size_t)d.S * M * K;
    const int MK = M * K;
    const unsigned gi = (unsigned)((MK + kBlock - 1) / kBlock);
    d.ka = ka; d.kb = kb; d.relu_in = relu_in;
    if (ka && kb) hipLaunchKernelGGL((k_pw_wgrad<float, false, true, true>), dim3((unsigned)(d.S * groups)), dim3(kBlock), 0, stream, dY, X, part, d);
    else hipLaunchKernelGGL((k_pw_wgrad<float, false, true>), dim3((unsigned)(d.S * groups)), dim3(kBlock), 0, stream, dY, X, part, d);
    if (d.S > kRedDirect) {
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, kRed), dim3(kBlock), 0, stream, (const float*)part, part2, MK, d.S, kRed);
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, 1), dim3(kBlock), 0, stream, (const float*)part2, dW, MK, kRed, 1);
    } else {
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, 1), dim3(kBlock), 0, stream, (const float*)part, dW, MK, d.S, 1);
    }
    return launch_status();
}
int rk_pw_s2_wgrad_f32(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win, void* ws,
                       size_t ws_bytes, rk_stream_t stream) {
    return pw_s2_wgrad(dY, X, dW, F, Cin, Cout, Hin, Win, ws, ws_bytes, stream, nullptr, nullptr, 0);
}
// training: the same d(weight) with X = relu?(ka x + kb) recomputed on the fly (the strided shortcut reads relu(bn1(x)))
int rk_pw_s2_wgrad_pro_f32(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win,
                           const float* ka, const float* kb, int relu_in, void* ws, size_t ws_bytes, rk_stream_t stream) {
    if (!ka || !kb) return RK_ERR_NULL_POINTER;
    return pw_s2_wgrad(dY, X, dW, F, Cin, Cout, Hin, Win, ws, ws_bytes, stream, ka, kb, relu_in);
}
int rk_bn_fold_f32(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                   float* a, float* b, int C, rk_stream_t stream) {
    if (!gamma || !beta || !running_mean || !running_var || !a || !b) return RK_ERR_NULL_POINTER;
    if (C <= 0) return RK_ERR_BAD_DIMS;
    hipLaunchKernelGGL(k_bn_fold, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, a, b, C);
    return launch_status();
}
// jobs: device array of n records {gamma*, beta*, running_mean*, running_var*, int64 off, int32 C, float eps} (48 bytes);
// ab = [2][total]: layer i's a / b at [off, off + C) of each half; max_c = the largest C
int rk_bn_fold_many_f32(const void* jobs, int n, float* ab, long long total, int max_c, rk_stream_t stream) {
    if (!jobs || !ab) return RK_ERR_NULL_POINTER;
    if (n <= 0 || n > 65535 || total <= 0 || max_c <= 0 || ((uintptr_t)jobs & 7)) return RK_ERR_BAD_DIMS;
    static_assert(sizeof(FoldJob) == 48, "the record layout pointwise.py writes");
    hipLaunchKernelGGL(k_bn_fold_many, dim3((max_c + kBlock - 1) / kBlock, n), dim3(kBlock), 0, (hipStream_t)stream,
                       (const FoldJob*)jobs, ab, total);
    return launch_status();
}
// Inference: Y[f] = epi(A pro(X[f])) (+ R[f]) with the per-channel affine (+ReLU) stages of PwFuse above; ka / kb
// have K entries, ma / mb have M; a NULL pair switches its stage off.
int rk_pw_gemm_fused_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P,
                         int a_is_mk, const float* ka, const float* kb, int relu_in, const float* ma,
                         const float* mb, int relu_out, rk_stream_t stream) {
    if ((ka == nullptr) != (kb == nullptr) || (ma == nullptr) != (mb == nullptr)) return RK_ERR_NULL_POINTER;
    const PwFuse fz{ka, kb, ma, mb, relu_in, relu_out};
    return pw_gemm<float>(A, X, R, Y, F, K, M, P, a_is_mk, stream, &fz);
}
// ---- planes with H * W % 4 != 0 (7x7): k_pw_gemm_odd ----
static int pw_gemm_odd(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk,
                       int mode, int Hin, int Win, rk_stream_t stream_) {
    if (!A || !X || !Y) return RK_ERR_NULL_POINTER;
    if (F <= 0 || K <= 0 || M <= 0 || P < 37 || P > 64) return RK_ERR_BAD_DIMS;           // 256 / P + 2 <= kOddFr; registers
    if (mode == 0 && (K % 4 || ((uintptr_t)X & 15))) return RK_ERR_BAD_DIMS;            // aligned frame chunks
    PwDims d;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P; d.a_is_mk = a_is_mk;
    d.Cin = 0; d.Hin = Hin; d.Win = Win; d.Wo = Win / 2; d.WM = 1; d.WN = 4;
    constexpr int NCB = 1, kFr = 6;                          // 128 columns per workgroup (k_pw_gemm_odd)
    const size_t lds = (size_t)(2 * kOddKC * 64 + 2 * kFr * kOddKC * P) * sizeof(float);      // <= 57 KB at P = 64
    const dim3 grid((unsigned)((d.ntot + 128 * NCB - 1) / (128 * NCB)), (unsigned)((M + 63) / 64)), block(kBlock);
    hipStream_t stream = (hipStream_t)stream_;
    if (mode == 0) hipLaunchKernelGGL((k_pw_gemm_odd<0, 0, NCB>), grid, block, lds, stream, A, X, R, Y, d);
    else if (mode == 1) hipLaunchKernelGGL((k_pw_gemm_odd<1, 0, NCB>), grid, block, lds, stream, A, X, R, Y, d);
    else hipLaunchKernelGGL((k_pw_gemm_odd<0, 1, NCB>), grid, block, lds, stream, A, X, R, Y, d);
    return launch_status();
}
// Y[f] = A X[f] (+ R[f]) for planes of P = H * W pixels with P % 4 != 0 (37 <= P <= 200; the 7x7 planes of layer4):
// forward (a_is_mk = 1) and d(input) (a_is_mk = 0) of a 1x1 / stride-1 convolution.  K % 4 == 0, X 16-byte aligned.
int rk_pw_gemm_odd_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P,
                       int a_is_mk, rk_stream_t stream) {
    return pw_gemm_odd(A, X, R, Y, F, K, M, P, a_is_mk, 0, 0, 0, stream);
}
// the 1x1 / stride-2 projecting shortcut onto such planes (14x14 -> 7x7): forward X [F, Cin, Hin, Win] -> Y [F, Cout,
// Hin/2, Win/2] and d(input) dY -> dX (zeros between the scattered results); Hin, Win even.
int rk_pw_s2_forward_odd_f32(const float* W, const float* X, float* Y, int F, int Cin, int Cout, int Hin, int Win,
                             rk_stream_t stream) {
    if (Hin <= 0 || Win <= 0 || Hin % 2 || Win % 2) return RK_ERR_BAD_DIMS;
    return pw_gemm_odd(W, X, nullptr, Y, F, Cin, Cout, (Hin / 2) * (Win / 2), 1, 1, Hin, Win, stream);
}
int rk_pw_s2_dgrad_odd_f32(const float* W, const float* dY, float* dX, int F, int Cin, int Cout, int Hin, int Win,
                           rk_stream_t stream) {
    if (Hin <= 0 || Win <= 0 || Hin % 2 || Win % 2) return RK_ERR_BAD_DIMS;
    const int P = (Hin / 2) * (Win / 2);
    if (Cout % 4 || ((uintptr_t)dY & 15)) return RK_ERR_BAD_DIMS;
    return pw_gemm_odd(W, dY, nullptr, dX, F, Cout, Cin, P, 0, 2, Hin, Win, stream);      // W read as [K=Cout][M=Cin]
}

// d(weight) for planes with P % 4 != 0: dW[M][K] = sum_f dY[f] X[f]^T, scalar pixel loads (k_pw_wgrad<.., ODD>).
// s2 != 0: X is [F, K, Hin, Win] read at stride 2, dY [F, M, Hin/2, Win/2] (the 14 -> 7 projecting shortcut).
static int pw_wgrad_odd(const float* dY, const float* X, float* dW, int F, int K, int M, int P, int s2, int Hin, int Win,
                        void* ws, size_t ws_bytes, rk_stream_t stream_) {
    if (!dY || !X || !dW) return RK_ERR_NULL_POINTER;
    WgDims d;
    if (int rc = make_wg(d, F, K, M, P, true)) return rc;
    d.Cin = 0; d.Hin = Hin; d.Win = Win; d.Wo = Win / 2;
    if (!ws || ws_bytes < (size_t)(d.S + kRed) * M * K * sizeof(float)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int nmk = d.MB * d.KB, bpw = nmk < 4 ? nmk : 4, groups = (nmk + bpw - 1) / bpw;
    float* part = (float*)ws;
    float* part2 = part + (size_t)d.S * M * K;
    const int MK = M * K;
    const unsigned gi = (unsigned)((MK + kBlock - 1) / kBlock);
    if (s2) hipLaunchKernelGGL((k_pw_wgrad<float, false, false, false, 2>), dim3((unsigned)(d.S * groups)), dim3(kBlock), 0, stream, dY, X, part, d);
    else hipLaunchKernelGGL((k_pw_wgrad<float, false, false, false, 1>), dim3((unsigned)(d.S * groups)), dim3(kBlock), 0, stream, dY, X, part, d);
    if (d.S > kRedDirect) {
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, kRed), dim3(kBlock), 0, stream, (const float*)part, part2, MK, d.S, kRed);
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, 1), dim3(kBlock), 0, stream, (const float*)part2, dW, MK, kRed, 1);
    } else {
        hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(gi, 1), dim3(kBlock), 0, stream, (const float*)part, dW, MK, d.S, 1);
    }
    return launch_status();
}
// workspace: rk_pw_wgrad_odd_workspace_bytes(F, K, M, P)
size_t rk_pw_wgrad_odd_workspace_bytes(int F, int K, int M, int P) {
    WgDims d;
    if (make_wg(d, F, K, M, P, true)) return 0;
    return (size_t)(d.S + kRed) * M * K * sizeof(float);
}
int rk_pw_wgrad_odd_f32(const float* dY, const float* X, float* dW, int F, int K, int M, int P, void* ws,
                        size_t ws_bytes, rk_stream_t stream) {
    return pw_wgrad_odd(dY, X, dW, F, K, M, P, 0, 0, 0, ws, ws_bytes, stream);
}
int rk_pw_s2_wgrad_odd_f32(const float* dY, const float* X, float* dW, int F, int Cin, int Cout, int Hin, int Win,
                           void* ws, size_t ws_bytes, rk_stream_t stream) {
    if (Hin <= 0 || Win <= 0 || Hin % 2 || Win % 2) return RK_ERR_BAD_DIMS;
    return pw_wgrad_odd(dY, X, dW, F, Cin, Cout, (Hin / 2) * (Win / 2), 1, Hin, Win, ws, ws_bytes, stream);
}

// ---- training-mode fusions (PwTrain) ----
// number of 128-column wave tiles of an [F, *, P] tensor = the J of the tile-partial arrays below
int rk_pw_tiles(int F, int P) {
    if (F <= 0 || P <= 0) return 0;
    return (int)(((long long)F * P + 127) / 128);
}
// tile count of the training epilogues of ONE GEMM call (rk_pw_gemm_stats_f32 / rk_pw_gemm_bnbwd_f32 with these
// arguments): the second-generation kernels (rk_pw2.hip) write one partial per 64 columns, the first per 128.
int rk_pw_gemm_tiles(const float* A, int F, int K, int M, int P, int a_is_mk) {
    if (F <= 0 || P <= 0 || K <= 0 || M <= 0) return 0;
    if (((uintptr_t)A & 15) == 0) {
        const int t3 = pw3::tiles(F, K, M, P);
        if (t3 > 0) return t3;
    }
    // (the streaming kernel's records are 64-column ones like rk_pw2.hip's; where it takes only some of a shape's epilogues,
    // rk_pw2.hip takes the others with the same count -- except the [M][K] 54-channel layers, which it takes entirely)
    const long long t4 = pw4::tiles(F, K, M, P, 1, 0, false);
    if (t4 > 0) return (int)t4;
    const int w = pw2::gemm_wanted(K, M, P, a_is_mk, A) ? pw2::kTileCols : 128;
    return (int)(((long long)F * P + w - 1) / w);
}
// forward of a block's conv2 / conv3 in training: Y[f] = A relu?(ka x + kb)(X[f]) (+ R[f]), and the tile statistics of Y
// (float4 [M][tiles]) for the BatchNorm that consumes Y.  ka / kb NULL: no prologue.
int rk_pw_gemm_stats_f32(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P,
                         int a_is_mk, const float* ka, const float* kb, int relu_in, void* stats, int tiles,
                         rk_stream_t stream) {
    if ((ka == nullptr) != (kb == nullptr)) return RK_ERR_NULL_POINTER;
    const PwFuse fz{ka, kb, nullptr, nullptr, relu_in, 0};
    const PwTrain tr{(float4*)stats, nullptr, nullptr, nullptr, tiles};
    return pw_gemm<float>(A, X, R, Y, F, K, M, P, a_is_mk, stream, &fz, &tr, 1);
}
// d(input) of conv2 in training, fused with the first half of bn1's backward: da = A dY (+ R) is masked with
// [ba x + bb > 0] (x = bn1's input) on its way out -> dZ, and the tile sums (sum dz, sum dz xhat) go to bred (float2
// [M][tiles]); xhat = (x - mean) invstd.  abmi: [M][4] = (a, b, mean, invstd) per channel, as rk_bn_finish_tiles_f32 packs it.
int rk_pw_gemm_bnbwd_f32(const float* A, const float* dY, const float* R, float* dZ, int F, int K, int M, int P,
                         int a_is_mk, const float* x, const float* abmi, void* bred, int tiles, rk_stream_t stream) {
    const PwTrain tr{nullptr, (float2*)bred, x, (const float4*)abmi, tiles};
    return pw_gemm<float>(A, dY, R, dZ, F, K, M, P, a_is_mk, stream, nullptr, &tr, 2);
}

// dW[M][K] (fp32) = sum_f dY[f] X[f]^T.  dY [F,M,P], X [F,K,P] fp32 or bf16, P % 4 == 0.
int rk_pw_wgrad_f32(const float* dY, const float* X, float* dW, int F, int K, int M, int P, void* ws,
                    size_t ws_bytes, rk_stream_t stream) {
    return pw_wgrad<float>(dY, X, dW, F, K, M, P, ws, ws_bytes, stream);
}
// training: d(weight) of conv2 with X = relu?(ka x + kb) recomputed from the block's input (fp32)
int rk_pw_wgrad_pro_f32(const float* dY, const float* X, float* dW, int F, int K, int M, int P, const float* ka,
                        const float* kb, int relu_in, void* ws, size_t ws_bytes, rk_stream_t stream) {
    if (!ka || !kb) return RK_ERR_NULL_POINTER;
    return pw_wgrad<float>(dY, X, dW, F, K, M, P, ws, ws_bytes, stream, ka, kb, relu_in);
}
int rk_pw_wgrad_bf16(const void* dY, const void* X, float* dW, int F, int K, int M, int P, void* ws,
                     size_t ws_bytes, rk_stream_t stream) {
    return pw_wgrad<__hip_bfloat16>(dY, X, dW, F, K, M, P, ws, ws_bytes, stream);
}

}  // extern "C"
