// rk_reduce.hpp -- out[i] = sum over S partial matrices in[c][i] (the d(weight) kernels' split partials), fixed order.
// A thread owns 4 consecutive outputs (16-byte loads) and one of 16 slices of the split range, summed front to back; the 16
// slice sums are then added in slice order.  (The first form -- one output per thread, 4 slices -- kept S / 4 dependent
// 4-byte loads per thread: 42 us per call on average in RubiksNet-Large's fp32 step, 4.1 ms of it.)
#pragma once
#include "rk_common.hpp"

namespace rk {

template <int TAG = 0>
__global__ __launch_bounds__(kBlock) void k_reduce_partials4(const float* __restrict__ in, float* __restrict__ out, int MK, int S) {
    __shared__ float4 part[16][16];
    const int q = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int i = (blockIdx.x * 16 + q) * 4;
    const int per = (S + 15) / 16, c0 = slice * per, c1 = (c0 + per) < S ? (c0 + per) : S;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < MK) {
#pragma unroll 8
        for (int c = c0; c < c1; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(in + (size_t)c * MK + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    part[slice][q] = acc;
    __syncthreads();
    if (slice == 0 && i < MK) {
        float4 t = part[0][q];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = part[k][q]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(out + i) = t;
    }
}

// MK % 4 == 0 and 16-byte aligned buffers: the kernel above; false = the caller's scalar form
inline bool launch_reduce_partials4(const float* in, float* out, int MK, int S, hipStream_t stream) {
    if (MK % 4 || (((uintptr_t)in | (uintptr_t)out) & 15)) return false;
    hipLaunchKernelGGL(k_reduce_partials4<0>, dim3((MK / 4 + 15) / 16), dim3(kBlock), 0, stream, in, out, MK, S);
    return true;
}

}  // namespace rk
