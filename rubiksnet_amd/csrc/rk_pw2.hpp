// rk_pw2.hpp -- entry points of the second-generation fp32 1x1 kernels (rk_pw2.hip) for rk_pw.hip's dispatch.
#pragma once
#include "rk_common.hpp"

namespace rk {
namespace pw2 {

struct GFuse {               // per-channel affine (+ReLU) stages: prologue on X rows (ka, kb), epilogue on Y rows (ma, mb)
    const float* ka; const float* kb; const float* ma; const float* mb;
    int relu_in, relu_out;
};
struct GTrain {              // training epilogues (tile = 64 columns)
    float4* stats;           // EPI 1: [M][J] (pivot, sum(y - pivot), sum((y - pivot)^2), n)
    float2* bred;            // EPI 2: [M][J] (sum dz, sum dz xhat)
    const float* bx;         // EPI 2: the BatchNorm's input, [F, M, P]
    const float4* bpack;     // EPI 2: [M] (a, b, mean, invstd)
    int J;
};
struct GCfg { int rb, amode, ct; };
struct WCfg { int id; int ns; int splits; };

constexpr int kTileCols = 64;            // columns per statistics tile of this generation (rk_pw.hip: 128)

bool gemm_wanted(int K, int M, int P, int a_is_mk, const float* A);
bool wgrad_wanted(int P);
int gemm(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk, const GFuse* fuse,
         const GTrain* train, int epi, hipStream_t stream, const GCfg* cfg_override);
size_t wgrad_workspace_bytes(int F, int K, int M, int P);
int wgrad(const float* dY, const float* X, float* dW, int F, int K, int M, int P, void* ws, size_t ws_bytes, const float* ka,
          const float* kb, int relu_in, hipStream_t stream, const WCfg* cfg_override);

}  // namespace pw2
}  // namespace rk
