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
#include "rk_common.hpp"

namespace rk {
namespace pw {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PwDims {
    int F, K, M, P;
    long long ntot;         // F * P columns
    int a_is_mk;            // A given as [M][K] row-major (the weight itself), else [K][M]
    int WM, WN;             // waves along M / along N (WM * WN = 4); workgroup tile = 64 WM rows x 128 WN columns
};


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

// kKC: K chunk (even); the launcher picks one that divides K when it can, so that no MFMA runs on padding
template <int WM, int kKC>
__global__ __launch_bounds__(kBlock, (kKC > 16 ? 1 : 2)) void k_pw_gemm(const float* __restrict__ A, const float* __restrict__ X,
                                                    float* __restrict__ Y, PwDims d) {
    constexpr int MT = 64 * WM, WN = 4 / WM;
    __shared__ float As[2][kKC * MT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * MT;
    const long long cg = ((long long)blockIdx.x * WN + wn) * 128 + 4 * l31;      // this lane's 4 columns
    const bool valid = cg < d.ntot;
    const long long cgc = valid ? cg : 0;
    const int f = (int)(cgc / d.P), p = (int)(cgc - (long long)f * d.P);
    const float* xp = X + ((size_t)f * d.K) * d.P + p;
    float* yp = Y + ((size_t)f * d.M) * d.P + p;

    f32x16 acc[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][q][r] = 0.f;

    auto load_b = [&](int k) -> float4 {
        return (valid && k < d.K) ? *reinterpret_cast<const float4*>(xp + (size_t)k * d.P)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    AStage<MT, kKC> ast;
    float4 bq[kKC / 2];                                   // B fragments of the current chunk, refilled in place:
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
            const float bv[4] = {bq[s].x, bq[s].y, bq[s].z, bq[s].w};
            bq[s] = load_b((c + 1) * kKC + 2 * s + kh);    // (all zeros past K)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[0][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[q], acc[0][q], 0, 0, 0);
                acc[1][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[q], acc[1][q], 0, 0, 0);
            }
        }
        if (more) ast.deposit(As[(c + 1) & 1]);
    }

    if (valid) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + wm * 64 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * kh;   // C/D map of 32x32 MFMA
                if (gm < d.M)
                    *reinterpret_cast<float4*>(yp + (size_t)gm * d.P) =
                        make_float4(acc[b][0][r], acc[b][1][r], acc[b][2][r], acc[b][3][r]);
            }
    }
}

}  // namespace pw
}  // namespace rk

using namespace rk;
using namespace rk::pw;

extern "C" {

// Y[f] = A X[f].  a_is_mk != 0: A is [M][K] row-major; else [K][M].  X [F,K,P], Y [F,M,P] fp32, P % 4 == 0.
int rk_pw_gemm_f32(const float* A, const float* X, float* Y, int F, int K, int M, int P, int a_is_mk,
                   rk_stream_t stream_) {
    if (!A || !X || !Y) return RK_ERR_NULL_POINTER;
    if (F <= 0 || K <= 0 || M <= 0 || P <= 0 || P % 4 != 0 || K % 2 != 0) return RK_ERR_BAD_DIMS;
    if (((uintptr_t)X & 15) || ((uintptr_t)Y & 15)) return RK_ERR_BAD_DIMS;
    PwDims d;
    d.F = F; d.K = K; d.M = M; d.P = P; d.ntot = (long long)F * P; d.a_is_mk = a_is_mk;
    static const int wm_env = [] { const char* e = getenv("RK_PW_WM"); return e ? atoi(e) : 0; }();
    int wm = M <= 64 ? 1 : (M <= 128 ? 2 : 4);
    if (wm_env == 1 || wm_env == 2 || wm_env == 4) wm = wm_env;
    d.WM = wm; d.WN = 4 / wm;
    const int mt = 64 * wm;
    const dim3 grid((unsigned)((d.ntot + 128 * d.WN - 1) / (128 * d.WN)), (unsigned)((M + mt - 1) / mt)), block(kBlock);
    hipStream_t stream = (hipStream_t)stream_;
    static const int kc_env = [] { const char* e = getenv("RK_PW_KC"); return e ? atoi(e) : 0; }();
    int kc = 16;
    for (int cand : {18, 16, 12, 6}) if (K % cand == 0) { kc = cand; break; }
    if (kc_env == 6 || kc_env == 12 || kc_env == 16 || kc_env == 18) kc = kc_env;
#define RK_PW_GO(WMV, KCV) hipLaunchKernelGGL((k_pw_gemm<WMV, KCV>), grid, block, 0, stream, A, X, Y, d)
#define RK_PW_KC(WMV) do { if (kc == 18) RK_PW_GO(WMV, 18); else if (kc == 12) RK_PW_GO(WMV, 12); else if (kc == 6) RK_PW_GO(WMV, 6); else RK_PW_GO(WMV, 16); } while (0)
    if (wm == 1) RK_PW_KC(1);
    else if (wm == 2) RK_PW_KC(2);
    else RK_PW_KC(4);
#undef RK_PW_KC
#undef RK_PW_GO
    return launch_status();
}

}  // extern "C"
