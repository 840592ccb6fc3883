// rk_misc.hip -- version / error strings / shape helper of the C ABI (include/rubiks_hip.h).
#include "rk_common.hpp"
#include "rk_dma.hpp"

namespace {
using namespace rk;

// out[p][h][w] = main[p][h][w] + (h, w even ? small[p][h / 2][w / 2] : 0) for bf16 planes (main == nullptr: + 0): the gradient
// of a stride-2 gather joined to the gradient of the other consumer of the same tensor in ONE pass (pointwise.fork_shortcut;
// zeros + strided copy + add were three).  ELEMS elements per thread (8: 16-byte cells, W % 8 == 0; 2: any even W).
typedef unsigned m_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned m_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned add_lo(unsigned pair, unsigned small16) {      // low bf16 of `pair` += small16 (fp32 add, one rounding)
    const float v = __uint_as_float(pair << 16) + __uint_as_float(small16 << 16);
    return (pair & 0xffff0000u) | (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(v));
}
template <int ELEMS>
__global__ __launch_bounds__(kBlock) void k_scatter2_add(const __hip_bfloat16* __restrict__ main, const __hip_bfloat16* __restrict__ small,
                                                         __hip_bfloat16* __restrict__ out, long long cells, int H, int W) {
    const long long c = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (c >= cells) return;
    const long long e = c * ELEMS;
    const int cpr = W / ELEMS;                                      // cells per row
    const long long row = c / cpr;                                   // (plane, h) flattened
    const int w = (int)(c - row * cpr) * ELEMS;
    const int h = (int)(row % H);
    const long long p = row / H;
    const bool even = (h & 1) == 0;
    const long long so = (p * (H / 2) + h / 2) * (W / 2) + w / 2;
    if constexpr (ELEMS == 8) {
        m_u32x4 v = {0u, 0u, 0u, 0u};
        if (main) v = __builtin_nontemporal_load(reinterpret_cast<const m_u32x4*>(main + e));
        if (even) {
            const m_u32x2 s = *reinterpret_cast<const m_u32x2*>(small + so);
            v[0] = add_lo(v[0], s[0] & 0xffffu); v[1] = add_lo(v[1], s[0] >> 16);
            v[2] = add_lo(v[2], s[1] & 0xffffu); v[3] = add_lo(v[3], s[1] >> 16);
        }
        __builtin_nontemporal_store(v, reinterpret_cast<m_u32x4*>(out + e));
    } else {
        unsigned v = 0u;
        if (main) v = *reinterpret_cast<const unsigned*>(main + e);
        if (even) v = add_lo(v, (unsigned)*reinterpret_cast<const unsigned short*>(small + so));
        *reinterpret_cast<unsigned*>(out + e) = v;
    }
}
}  // namespace

extern "C" {

// out [planes][H][W] bf16 = main (NULL: zeros) + small [planes][H/2][W/2] scattered to the even (h, w); H, W even
int rk_scatter2x2_add_bf16(const void* main_, const void* small_, void* out_, long long planes, int H, int W, rk_stream_t stream) {
    if (!small_ || !out_) return RK_ERR_NULL_POINTER;
    if (planes <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2) return RK_ERR_BAD_DIMS;
    const __hip_bfloat16* main = (const __hip_bfloat16*)main_;
    const __hip_bfloat16* small = (const __hip_bfloat16*)small_;
    __hip_bfloat16* out = (__hip_bfloat16*)out_;
    const long long total = planes * H * W;
    const bool wide = W % 8 == 0 && !(((uintptr_t)main_ | (uintptr_t)out_) & 15) && !((uintptr_t)small_ & 7);
    if (((uintptr_t)main_ | (uintptr_t)out_) & 3) return RK_ERR_BAD_DIMS;
    const long long cells = total / (wide ? 8 : 2);
    const long long blocks = (cells + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return RK_ERR_BAD_DIMS;
    if (wide) hipLaunchKernelGGL(k_scatter2_add<8>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, main, small, out, cells, H, W);
    else hipLaunchKernelGGL(k_scatter2_add<2>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, main, small, out, cells, H, W);
    return launch_status();
}

int rk_version(void) { return 1000 * 0 + 1; }

const char* rk_error_string(int code) {
    switch (code) {
        case RK_OK: return "ok";
        case RK_ERR_NULL_POINTER: return "a required device pointer is NULL";
        case RK_ERR_BAD_DIMS: return "bad dimensions (non-positive, or element count exceeds int32)";
        case RK_ERR_BAD_STRIDE: return "stride must be > 0 and padding >= 0";
        case RK_ERR_WORKSPACE: return "workspace is NULL or smaller than *_workspace_bytes()";
        case RK_ERR_LAUNCH: return "HIP kernel launch failed (hipGetLastError != hipSuccess)";
        case RK_ERR_NO_DEVICE: return "no usable HIP device";
        case RK_ERR_UNSUPPORTED: return "no fused kernel for this configuration (use the unfused entry points)";
        default: return "unknown rubiks_hip error code";
    }
}

// cuda_src/rubiks.cpp:14-30 / :161-178
int rk_out_len(int in, int stride, int pad) {
    if (stride <= 0) return RK_ERR_BAD_STRIDE;
    return rk::out_len(in, stride, pad);
}

int rk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Test hook: the tag the NEXT fused-finalize launch will stamp its d(shift) granules with (rk_dma.hpp), not consumed.
// Lets tests/ pre-fill a workspace with near-miss granules; no product code calls it.
unsigned rk_debug_peek_launch_tag(void) {
    const unsigned t = rk::dma::launch_tag_counter().load(std::memory_order_relaxed);
    return t ? t : 1u;
}

// Test hook: the number of polls after which an in-launch d(shift) finalizer gives up and its outputs become NaN
// (rk_dma.hpp, kFinSpins ~ 2 s).  spins <= 0 restores the default.  Returns the previous value.  Takes effect for launches
// armed afterwards; no product code calls it.
int rk_debug_set_finalize_spins(int spins) {
    return rk::dma::fin_spin_budget().exchange(spins > 0 ? spins : rk::dma::kFinSpins, std::memory_order_relaxed);
}

}  // extern "C"
