// How does the f32 MFMA rate of a SIMD depend on the number of resident waves that issue MFMAs and on what sits between
// their MFMAs?  Pure register kernels (no memory): each wave issues ITER x 48 MFMAs on 12 independent accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_contention_probe.hip -o tools/bin/mfma_contention_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: 16x16x4 back to back; 1: 16x16x4 with a VALU op + s_nop after every 4; 2: 32x32x2 back to back (3 acc)
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x, b = b0;
    if constexpr (MODE == 2) {
        f32x16 acc[3];
        for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 3; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        if (s == 123.456f) out[threadIdx.x] = s;
    } else {
        f32x4 acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                    if (MODE == 1 && (i & 3) == 3) { asm volatile("v_add_f32 %0, %0, %1\n\ts_nop 1" : "+v"(a) : "v"(b)); }
                }
            }
        }
        float s = 0.f;
        for (int i = 0; i < 12; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        if (s == 123.456f) out[threadIdx.x] = s;
    }
}

template <int MODE>
void run(const char* name, int wgs_per_cu, float flop_per_mfma, int mfma_per_iter) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)grid * 4 * iters * mfma_per_iter;          // MFMAs in all
    const double cyc_per_mfma_simd = ms * 1e-3 * 2.4e9 / (n / 1024.0);
    printf("%-34s waves/SIMD %d: %8.3f ms  %6.1f TFLOP/s  %5.1f cycles per MFMA per SIMD (at 2.4 GHz)\n", name, wgs_per_cu, ms,
           n * flop_per_mfma / ms * 1e-9, cyc_per_mfma_simd);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 6; ++w) run<0>("16x16x4 f32 back to back", w, 2048.f, 48);
    for (int w = 1; w <= 6; ++w) run<1>("16x16x4 f32, VALU + nop every 4", w, 2048.f, 48);
    for (int w = 1; w <= 6; ++w) run<2>("32x32x2 f32 back to back", w, 4096.f, 24);
    return 0;
}
