// rk_common.hpp -- shared device/host helpers for librubiks_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <atomic>
#include <stdlib.h>
#include <string.h>

#include "rubiks_hip.h"

namespace rk {

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kBlock = 256;    // 4 waves = one per SIMD

// ---- storage <-> compute type mapping (f16/bf16 compute in fp32) -----------------
template <typename T> struct Compute { using type = T; };
template <> struct Compute<__half> { using type = float; };
template <> struct Compute<__hip_bfloat16> { using type = float; };

template <typename T> __device__ __forceinline__ typename Compute<T>::type ld(const T* p) { return *p; }
template <> __device__ __forceinline__ float ld<__half>(const __half* p) { return __half2float(*p); }
template <> __device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16* p) {
    return __bfloat162float(*p);
}
template <typename T> __device__ __forceinline__ void st(T* p, typename Compute<T>::type v) { *p = v; }
// (the empty asm keeps the fp32 value materialised: without it hipcc folds `(half)(a * b)` into v_fma_mixlo_f16, ONE
// rounding from the exact product, which differs from "fp32 result rounded once" on double-rounding ties and made
// the fused strided 2-D backward disagree with the d(x)-only kernel on 6 of 128 800 elements)
__device__ __forceinline__ float materialise(float v) { asm volatile("" : "+v"(v)); return v; }
template <> __device__ __forceinline__ void st<__half>(__half* p, float v) { *p = __float2half(materialise(v)); }
template <> __device__ __forceinline__ void st<__hip_bfloat16>(__hip_bfloat16* p, float v) {
    *p = __float2bfloat16(v);
}

// ---- wave / block reductions (deterministic: fixed tree, no float atomics) -------
// Sum over the 64 lanes, returned in EVERY lane.  float / double: DPP row shifts + row broadcasts (the gfx9 reduction
// idiom: 4 steps inside each row of 16 lanes, row_bcast:15, row_bcast:31, then lane 63 read back) -- a handful of VALU
// instructions; the __shfl_down tree it replaces is one ds_bpermute round trip per step (two for a double): ~1.8 us for
// the three fp64 sums of a d(shift) finalizer wave, on the tail of every fused backward launch.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_or_zero(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);       // 0 where there is no source lane / row masked
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_or_zero(float v) {
    return __builtin_bit_cast(float, dpp_or_zero<CTRL, ROW_MASK>(__builtin_bit_cast(int, v)));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_or_zero(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)dpp_or_zero<CTRL, ROW_MASK>((int)(unsigned)b);
    const unsigned hi = (unsigned)dpp_or_zero<CTRL, ROW_MASK>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename A> __device__ __forceinline__ A wave_sum_dpp(A v) {
    v += dpp_or_zero<0x111, 0xf>(v);      // row_shr:1
    v += dpp_or_zero<0x112, 0xf>(v);      // row_shr:2
    v += dpp_or_zero<0x114, 0xf>(v);      // row_shr:4
    v += dpp_or_zero<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row holds the row's sum
    v += dpp_or_zero<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}
// CONTRACT (every type): all 64 lanes of the wave are active and converged at the call (whole-wave callers only: a lane
// that has nothing to add passes 0), and the sum comes back in EVERY lane.  float / double take the DPP scan below, whose
// row_bcast controls exist on gfx9 only -- this library is gfx950 code and says so at compile time; any other type takes
// the shuffle tree and broadcasts lane 0's total, so that the contract does not depend on the type.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "librubiks_hip is written for gfx950 (CDNA4): wave64, DPP row_bcast, global_load_lds_dwordx4"
#endif
template <typename A> __device__ __forceinline__ A wave_sum(A v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return __shfl(v, 0, kWave);
}
template <> __device__ __forceinline__ float wave_sum<float>(float v) {
    v = wave_sum_dpp(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <> __device__ __forceinline__ double wave_sum<double>(double v) {
    v = wave_sum_dpp(v);
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Sum `v` over groups of `group` consecutive threads (group in {64,128,256}, a divisor of
// blockDim.x = 256).  Result valid in the first thread of every group.  `smem` must hold
// kBlock / kWave values of A per reduced quantity; caller passes a distinct slice per call
// or syncs in between.
template <typename A> __device__ __forceinline__ A group_sum(A v, int group, A* smem) {
    v = wave_sum(v);
    if (group == kWave) return v;              // uniform across the block
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    const int waves_per_group = group / kWave;
    A r = 0;
    if (lane == 0 && (wave % waves_per_group) == 0)
        for (int k = 0; k < waves_per_group; ++k) r += smem[wave + k];
    __syncthreads();
    return r;
}

// ---- the library's one switch -------------------------------------------------------
// RK_SHIFT_KERNELS = auto (default) | column | generic, read once per process:
//   auto    LDS-DMA streaming kernels (plane-group / column-walk / tile, 2-D twins) where a shape qualifies,
//           then the column kernels, then the per-plane generic kernels
//   column  no LDS-DMA streaming kernels (what a buffer that is not 16-byte aligned gets anyway)
//   generic per-plane generic kernels only (RK_FORCE_GENERIC=1 is the older spelling)
// Every family is bit-identical to the oracle for y and d(x); tests/test_fallback_paths_gpu.py runs the parity
// suite on each setting.  Tile shapes, prefetch depths and channel limits are constants next to the kernels.
enum class ShiftKernels { Auto, Column, Generic };
inline ShiftKernels shift_kernels() {
    static const ShiftKernels v = [] {
        const char* e = getenv("RK_SHIFT_KERNELS");
        if (e && !strcmp(e, "generic")) return ShiftKernels::Generic;
        if (e && !strcmp(e, "column")) return ShiftKernels::Column;
        const char* f = getenv("RK_FORCE_GENERIC");
        return (f && f[0] == '1') ? ShiftKernels::Generic : ShiftKernels::Auto;
    }();
    return v;
}
inline bool streaming_kernels_on() { return shift_kernels() == ShiftKernels::Auto; }
inline bool column_kernels_on() { return shift_kernels() != ShiftKernels::Generic; }

// ---- host helpers ----------------------------------------------------------------
inline int out_len(int in, int stride, int pad) { return (in + 2 * pad - 1) / stride + 1; }

// A full, aligned ds_read_b128.  The empty asm makes all four components "used", so the compiler
// cannot narrow the access to the components one switch-case of pick5 needs: narrowed b32/b64 reads
// at a 16 B lane stride are 4-way / 2-way bank conflicts (66% of LDS cycles in the first profile),
// b128 at a 16 B stride is conflict-free.
__device__ __forceinline__ float4 lds_b128(const float4* p) {
    float4 v = *p;
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    return v;
}

constexpr int kMaxDevices = 64;           // devices whose per-device host state is cached (beyond: queried every call)

// compute units of the current device, cached per device (256 on a full MI355X; fewer in partitioned modes).  The cache is
// a table of relaxed atomics: any thread of any device may call this (nn.DataParallel: one autograd thread per device).
inline int device_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    static std::atomic<int> cached[kMaxDevices];
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    int v = tracked ? cached[dev].load(std::memory_order_relaxed) : 0;
    if (!v) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        if (tracked) cached[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// Kernels that take more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised -- per DEVICE
// (every device has its own copy of the function), so the "already raised" record of a kernel instantiation is one bit
// per device, atomic, and the attribute is always raised to the hardware's 160 KB: every call that races sets the same
// value, so the order in which two threads get there cannot leave a lower ceiling behind.  One DynLdsRaised lives as a
// function-local static of each launcher template instantiation.
struct DynLdsRaised {
    std::atomic<unsigned long long> devices{0};
};
constexpr size_t kMaxDynLds = 160 * 1024;
inline int raise_dynamic_lds(const void* kernel, size_t lds, DynLdsRaised& st) {
    if (lds > kMaxDynLds) return RK_ERR_UNSUPPORTED;
    if (lds <= 65536) return RK_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RK_ERR_LAUNCH;
    const bool tracked = dev >= 0 && dev < kMaxDevices;
    const unsigned long long bit = tracked ? 1ull << dev : 0;
    if (tracked && (st.devices.load(std::memory_order_acquire) & bit)) return RK_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynLds) != hipSuccess) return RK_ERR_LAUNCH;
    if (tracked) st.devices.fetch_or(bit, std::memory_order_release);
    return RK_OK;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? RK_OK : RK_ERR_LAUNCH; }

// threads of a finalize workgroup for P partials per channel
inline int finalize_block(int P) { return P <= kWave ? kWave : kBlock; }

inline int pow2_at_least(int v, int lo, int hi) {
    int p = lo;
    while (p < v && p < hi) p <<= 1;
    return p;
}

}  // namespace rk
