// rk_pw3.hpp -- entry points of the LDS-tiled fp32 GEMM for the 288-row layers (rk_pw3.hip) for rk_pw.hip's dispatch.
#pragma once
#include "rk_common.hpp"
#include "rk_pw2.hpp"

namespace rk {
namespace pw3 {

// statistics / reduction tiles of one call (2 per workgroup: its column halves), 0 when this kernel does not take the shape
int tiles(int F, int K, int M, int P);
// Y[f] = A pro(X[f]) (+ R[f]) with the training epilogues of rk_pw2.hpp's GTrain; RK_ERR_UNSUPPORTED: not this kernel's shape
int gemm(const float* A, const float* X, const float* R, float* Y, int F, int K, int M, int P, int a_is_mk,
         const pw2::GFuse* fuse, const pw2::GTrain* train, int epi, hipStream_t stream);

}  // namespace pw3
}  // namespace rk
