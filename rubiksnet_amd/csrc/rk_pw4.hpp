// rk_pw4.hpp -- entry points of the streaming fp32 GEMM for the shallow layers (rk_pw4.hip) for rk_pw.hip's dispatch.
#pragma once
#include "rk_common.hpp"
#include "rk_pw2.hpp"

namespace rk {
namespace pw4 {

// > 0: the number of 64-column statistics tiles the kernel writes for this shape (= it will take a call with epilogue
// `epi`, with a residual or not); 0: not taken.  force: ignore the size / epilogue policy (tests, probes)
long long tiles(int F, int K, int M, int P, int epi, int res, bool force);
// Y[f] = epi(A pro(X[f])) (+ R[f]); RK_ERR_UNSUPPORTED when the shape has no instance (the caller falls back)
int gemm(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk, const pw2::GFuse* fuse,
         const pw2::GTrain* train, int epi, hipStream_t stream, bool force);

}  // namespace pw4
}  // namespace rk
