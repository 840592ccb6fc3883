// rk_clip.hip -- input side of the network on the device and the squeeze-and-excitation
// gate of RubiksNet-Small (row f3).
//
// rk_clip_u8_to_chw_*: the reference's per-sample CPU transforms Stack -> ToTorchFormatTensor(div=True) ->
// GroupNormalize (rubiksnet/transforms.py:329-363, :66-79) as ONE pass on the GPU: a clip arrives as the stacked
// HWC uint8 image [H, W, 3T] (what Stack produces; what a JPEG decoder produces per frame, frames interleaved along
// the channel axis), leaves as [3T, H, W] floats, ((v / 255) - mean[c % 3]) / std[c % 3], each operation rounded in
// fp32 exactly as `img.float().div(255)` followed by `t.sub_(m).div_(s)` rounds it.  The reference notes "this
// transpose takes 80% of the loading time/CPU" (transforms.py:357); here it is 1 B read + 4 B written per element.
// A workgroup owns 256 consecutive pixels of one row band: their 256 * CS bytes are one contiguous run (16-byte
// loads), staged in LDS, then written channel by channel (consecutive lanes -> consecutive floats).
//
// rk_se_*: SELayer (rubiksnet/backbone.py:56-71): squeeze = per-(frame, channel) mean over H*W, excite = two tiny
// Linear layers + sigmoid (left to PyTorch: [F, C] x [C, C/r]), scale = x * gate.  The reference runs avg-pool,
// view, 2 x Linear, ReLU, sigmoid, expand_as and a broadcast multiply as separate kernels and autograd saves x and
// the expanded gate; here squeeze and scale are one pass each, and the backward of the scale produces d(x) and
// d(gate) (a per-plane dot product) in one pass over (dy, x).
#include <type_traits>
#include "rk_common.hpp"

using namespace rk;

namespace {

template <typename T> __device__ __forceinline__ void store_out(T* p, float v) { st(p, v); }

// ------------------------------------------------------------------------------------------ clip transform
template <typename T>
__global__ __launch_bounds__(kBlock) void k_clip_u8_to_chw(const unsigned char* __restrict__ src,
                                                           const float* __restrict__ mean3,
                                                           const float* __restrict__ std3, T* __restrict__ dst,
                                                           int HW, int CS) {
    extern __shared__ unsigned char tile[];                       // [256 pixels][CS]
    const int clip = blockIdx.y;
    const long long p0 = (long long)blockIdx.x * kBlock;          // first pixel of this workgroup
    const int npix = (HW - p0) < kBlock ? (int)(HW - p0) : kBlock;
    const unsigned char* in = src + ((long long)clip * HW + p0) * CS;
    const int nbytes = npix * CS;
    // the run starts 16-byte aligned when p0 * CS is (256 * CS always is) and the clip base is (HW * CS % 16 == 0,
    // checked by the launcher); the tail is copied bytewise
    const int n16 = nbytes / 16;
    for (int i = threadIdx.x; i < n16; i += kBlock)
        reinterpret_cast<uint4*>(tile)[i] = reinterpret_cast<const uint4*>(in)[i];
    for (int i = n16 * 16 + threadIdx.x; i < nbytes; i += kBlock) tile[i] = in[i];
    __syncthreads();
    const int px = threadIdx.x;
    if (px >= npix) return;
    T* out = dst + (long long)clip * CS * HW + p0 + px;
    for (int c = 0; c < CS; ++c) {
        const int k = c % 3;
        const float v = (float)tile[px * CS + c] / 255.0f;        // ToTorchFormatTensor: .float().div(255)
        store_out(out + (long long)c * HW, (v - mean3[k]) / std3[k]);   // GroupNormalize: sub_(m).div_(s)
    }
}

// ------------------------------------------------------------------------------------------ SE squeeze / scale
// a plane of P fp32 elements can be walked in 16-byte cells (wave-uniform: planes are P apart from an aligned base)
__device__ __forceinline__ bool se_vec4(const void* plane_ptr, int P) {
    return (P & 3) == 0 && (reinterpret_cast<uintptr_t>(plane_ptr) & 15) == 0;
}
// one wave per (frame, channel) plane
template <typename T>
__global__ __launch_bounds__(kBlock) void k_se_squeeze(const T* __restrict__ x, float* __restrict__ mean, long long planes,
                                                       int P) {
    const long long plane = (long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & (kWave - 1);
    const T* p = x + plane * P;
    float s = 0.f;
    if constexpr (std::is_same<T, float>::value) {
        if (se_vec4(p, P)) {                              // 16-byte accesses, 4 in flight per lane (the scalar walk below is one
            const int n4 = P >> 2;                        //  4-byte load per lane and round trip: latency-bound at 56 x 56)
#pragma unroll 4
            for (int i = lane; i < n4; i += kWave) {
                const float4 v = reinterpret_cast<const float4*>(p)[i];
                s += (v.x + v.y) + (v.z + v.w);
            }
            s = wave_sum(s);
            if (lane == 0) mean[plane] = s / (float)P;
            return;
        }
    }
    for (int i = lane; i < P; i += kWave) s += ld(p + i);
    s = wave_sum(s);
    if (lane == 0) mean[plane] = s / (float)P;
}

// y = x * gate[plane]
template <typename T>
__global__ __launch_bounds__(kBlock) void k_se_scale(const T* __restrict__ x, const float* __restrict__ gate, T* __restrict__ y,
                                                     long long planes, int P) {
    const long long plane = (long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & (kWave - 1);
    const float g = gate[plane];
    const T* p = x + plane * P;
    T* o = y + plane * P;
    if constexpr (std::is_same<T, float>::value) {
        if (se_vec4(p, P) && se_vec4(o, P)) {
            const int n4 = P >> 2;
#pragma unroll 4
            for (int i = lane; i < n4; i += kWave) {
                float4 v = reinterpret_cast<const float4*>(p)[i];
                v.x *= g; v.y *= g; v.z *= g; v.w *= g;
                reinterpret_cast<float4*>(o)[i] = v;
            }
            return;
        }
    }
    for (int i = lane; i < P; i += kWave) st(o + i, ld(p + i) * g);
}

// dx = dy * gate[plane] (+ dmean[plane] / P: the squeeze's share of d(x)), dgate[plane] = sum(dy * x)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_se_scale_backward(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const float* __restrict__ gate, T* __restrict__ dx,
                                                              float* __restrict__ dgate, long long planes, int P) {
    const long long plane = (long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & (kWave - 1);
    const float g = gate[plane];
    const T* pd = dy + plane * P;
    const T* px = x + plane * P;
    T* o = dx + plane * P;
    float s = 0.f;
    if constexpr (std::is_same<T, float>::value) {
        if (se_vec4(pd, P) && se_vec4(px, P) && se_vec4(o, P)) {
            const int n4 = P >> 2;
#pragma unroll 4
            for (int i = lane; i < n4; i += kWave) {
                const float4 d = reinterpret_cast<const float4*>(pd)[i];
                const float4 v = reinterpret_cast<const float4*>(px)[i];
                s = fmaf(d.x, v.x, s); s = fmaf(d.y, v.y, s); s = fmaf(d.z, v.z, s); s = fmaf(d.w, v.w, s);
                reinterpret_cast<float4*>(o)[i] = make_float4(d.x * g, d.y * g, d.z * g, d.w * g);
            }
            s = wave_sum(s);
            if (lane == 0) dgate[plane] = s;
            return;
        }
    }
    for (int i = lane; i < P; i += kWave) {
        const float d = ld(pd + i);
        s = fmaf(d, ld(px + i), s);
        st(o + i, d * g);
    }
    s = wave_sum(s);
    if (lane == 0) dgate[plane] = s;
}

// The fused training block's SE backward in 4 tensor passes instead of 5 (scale_backward = 3, then dx += dmean / P = 2):
// dgate[plane] = sum(dy * x) alone (2 reads), then -- after the two Linear layers gave d(mean) -- dx = dy * gate[plane] +
// add[plane] (1 read, 1 write).
__global__ __launch_bounds__(kBlock) void k_se_dgate(const float* __restrict__ dy, const float* __restrict__ x,
                                                     float* __restrict__ dgate, long long planes, int P) {
    const long long plane = (long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & (kWave - 1);
    const float* pd = dy + plane * P;
    const float* px = x + plane * P;
    float s = 0.f;
    if (se_vec4(pd, P) && se_vec4(px, P)) {
        const int n4 = P >> 2;
#pragma unroll 4
        for (int i = lane; i < n4; i += kWave) {
            const float4 d = reinterpret_cast<const float4*>(pd)[i];
            const float4 v = reinterpret_cast<const float4*>(px)[i];
            s = fmaf(d.x, v.x, s); s = fmaf(d.y, v.y, s); s = fmaf(d.z, v.z, s); s = fmaf(d.w, v.w, s);
        }
    } else {
        for (int i = lane; i < P; i += kWave) s = fmaf(pd[i], px[i], s);
    }
    s = wave_sum(s);
    if (lane == 0) dgate[plane] = s;
}
__global__ __launch_bounds__(kBlock) void k_se_scale_add(const float* __restrict__ x, const float* __restrict__ gate,
                                                         const float* __restrict__ add, float add_scale, float* __restrict__ y,
                                                         long long planes, int P) {
    const long long plane = (long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & (kWave - 1);
    const float g = gate[plane], a = add[plane] * add_scale;
    const float* p = x + plane * P;
    float* o = y + plane * P;
    if (se_vec4(p, P) && se_vec4(o, P)) {
        const int n4 = P >> 2;
#pragma unroll 4
        for (int i = lane; i < n4; i += kWave) {
            const float4 v = reinterpret_cast<const float4*>(p)[i];
            reinterpret_cast<float4*>(o)[i] = make_float4(fmaf(v.x, g, a), fmaf(v.y, g, a), fmaf(v.z, g, a), fmaf(v.w, g, a));
        }
    } else {
        for (int i = lane; i < P; i += kWave) o[i] = fmaf(p[i], g, a);
    }
}

// ---- the two bias-free Linear layers of the gate (backbone.py:60-64: Linear(C, C / r) -> ReLU -> Linear(C / r, C) -> Sigmoid)
// on the squeezed [F, C] vector, one workgroup per frame: as PyTorch ops this is 4 launches forward and 9 backward per block
// of ~5-15 us each (2 ms of the RubiksNet-Small train step for 40 MFLOP).  W1 [Cr][C], W2 [C][Cr] (nn.Linear layout).
constexpr int kSeMaxC = 2048, kSeMaxCr = 128;
__global__ __launch_bounds__(kBlock) void k_se_mlp_forward(const float* __restrict__ q, const float* __restrict__ W1,
                                                           const float* __restrict__ W2, float* __restrict__ h,
                                                           float* __restrict__ g, int C, int Cr) {
    __shared__ float qs[kSeMaxC], hs[kSeMaxCr];
    const int f = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    for (int c = threadIdx.x; c < C; c += kBlock) qs[c] = q[(size_t)f * C + c];
    __syncthreads();
    for (int j = wave; j < Cr; j += kBlock / kWave) {
        float s = 0.f;
        for (int c = lane; c < C; c += kWave) s = fmaf(W1[(size_t)j * C + c], qs[c], s);
        s = wave_sum(s);
        if (lane == 0) {
            s = s > 0.f ? s : 0.f;
            hs[j] = s;
            h[(size_t)f * Cr + j] = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += kBlock) {
        float s = 0.f;
        for (int j = 0; j < Cr; ++j) s = fmaf(W2[(size_t)c * Cr + j], hs[j], s);
        g[(size_t)f * C + c] = 1.f / (1.f + expf(-s));
    }
}
// dpre2 = dgate g (1 - g); dpre1 = (dpre2 W2) [h > 0]; dq = dpre1 W1
__global__ __launch_bounds__(kBlock) void k_se_mlp_backward(const float* __restrict__ dgate, const float* __restrict__ g,
                                                            const float* __restrict__ h, const float* __restrict__ W1,
                                                            const float* __restrict__ W2, float* __restrict__ dpre2,
                                                            float* __restrict__ dpre1, float* __restrict__ dq, int C, int Cr) {
    __shared__ float d2[kSeMaxC], p1[kSeMaxCr];
    const int f = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    for (int c = threadIdx.x; c < C; c += kBlock) {
        const float gv = g[(size_t)f * C + c];
        const float v = dgate[(size_t)f * C + c] * gv * (1.f - gv);
        d2[c] = v;
        dpre2[(size_t)f * C + c] = v;
    }
    __syncthreads();
    for (int j = wave; j < Cr; j += kBlock / kWave) {
        float s = 0.f;
        for (int c = lane; c < C; c += kWave) s = fmaf(d2[c], W2[(size_t)c * Cr + j], s);
        s = wave_sum(s);
        if (lane == 0) {
            s = h[(size_t)f * Cr + j] > 0.f ? s : 0.f;
            p1[j] = s;
            dpre1[(size_t)f * Cr + j] = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += kBlock) {
        float s = 0.f;
        for (int j = 0; j < Cr; ++j) s = fmaf(p1[j], W1[(size_t)j * C + c], s);
        dq[(size_t)f * C + c] = s;
    }
}
// dW2[c][j] = sum_f dpre2[f][c] h[f][j]; dW1[j][c] = sum_f dpre1[f][j] q[f][c]  (frames in order: deterministic)
__global__ __launch_bounds__(kBlock) void k_se_mlp_wgrad(const float* __restrict__ dpre2, const float* __restrict__ h,
                                                         const float* __restrict__ dpre1, const float* __restrict__ q,
                                                         float* __restrict__ dW1, float* __restrict__ dW2, int F, int C, int Cr) {
    const int idx = blockIdx.x * kBlock + threadIdx.x, n = C * Cr;
    if (idx < n) {
        const int c = idx / Cr, j = idx - c * Cr;
        float s = 0.f;
        for (int f = 0; f < F; ++f) s = fmaf(dpre2[(size_t)f * C + c], h[(size_t)f * Cr + j], s);
        dW2[idx] = s;
    } else if (idx < 2 * n) {
        const int i2 = idx - n, j = i2 / C, c = i2 - j * C;
        float s = 0.f;
        for (int f = 0; f < F; ++f) s = fmaf(dpre1[(size_t)f * Cr + j], q[(size_t)f * C + c], s);
        dW1[i2] = s;
    }
}

unsigned plane_grid(long long planes) { return (unsigned)((planes + kBlock / kWave - 1) / (kBlock / kWave)); }

template <typename T>
int clip_impl(const unsigned char* src, const float* mean3, const float* std3, void* dst, int nclips, int H, int W, int CS,
              rk_stream_t stream) {
    if (!src || !mean3 || !std3 || !dst) return RK_ERR_NULL_POINTER;
    if (nclips <= 0 || H <= 0 || W <= 0 || CS <= 0 || CS % 3 != 0 || CS > 192) return RK_ERR_BAD_DIMS;
    const long long HW = (long long)H * W;
    if (HW * CS * nclips > 0x7fffffffLL || ((HW * CS) % 16) != 0 || ((uintptr_t)src & 15)) return RK_ERR_BAD_DIMS;
    const dim3 grid((unsigned)((HW + kBlock - 1) / kBlock), (unsigned)nclips);
    hipLaunchKernelGGL((k_clip_u8_to_chw<T>), grid, dim3(kBlock), (size_t)kBlock * CS, (hipStream_t)stream, src, mean3, std3,
                       (T*)dst, (int)HW, CS);
    return launch_status();
}

}  // namespace

extern "C" {

int rk_clip_u8_to_chw_f32(const unsigned char* hwc, const float* mean3, const float* std3, float* chw, int nclips, int H,
                          int W, int CS, rk_stream_t stream) {
    return clip_impl<float>(hwc, mean3, std3, chw, nclips, H, W, CS, stream);
}
int rk_clip_u8_to_chw_bf16(const unsigned char* hwc, const float* mean3, const float* std3, void* chw, int nclips, int H,
                           int W, int CS, rk_stream_t stream) {
    return clip_impl<__hip_bfloat16>(hwc, mean3, std3, chw, nclips, H, W, CS, stream);
}

#define RK_SE_IMPL(SFX, T)                                                                                          \
    int rk_se_squeeze_##SFX(const void* x, float* mean, int F, int C, int P, rk_stream_t stream) {                  \
        if (!x || !mean) return RK_ERR_NULL_POINTER;                                                                \
        if (F <= 0 || C <= 0 || P <= 0) return RK_ERR_BAD_DIMS;                                                     \
        const long long planes = (long long)F * C;                                                                  \
        hipLaunchKernelGGL((k_se_squeeze<T>), dim3(plane_grid(planes)), dim3(kBlock), 0, (hipStream_t)stream,       \
                           (const T*)x, mean, planes, P);                                                           \
        return launch_status();                                                                                     \
    }                                                                                                               \
    int rk_se_scale_##SFX(const void* x, const float* gate, void* y, int F, int C, int P, rk_stream_t stream) {     \
        if (!x || !gate || !y) return RK_ERR_NULL_POINTER;                                                          \
        if (F <= 0 || C <= 0 || P <= 0) return RK_ERR_BAD_DIMS;                                                     \
        const long long planes = (long long)F * C;                                                                  \
        hipLaunchKernelGGL((k_se_scale<T>), dim3(plane_grid(planes)), dim3(kBlock), 0, (hipStream_t)stream,         \
                           (const T*)x, gate, (T*)y, planes, P);                                                    \
        return launch_status();                                                                                     \
    }                                                                                                               \
    int rk_se_scale_backward_##SFX(const void* dy, const void* x, const float* gate, void* dx, float* dgate, int F, \
                                   int C, int P, rk_stream_t stream) {                                              \
        if (!dy || !x || !gate || !dx || !dgate) return RK_ERR_NULL_POINTER;                                        \
        if (F <= 0 || C <= 0 || P <= 0) return RK_ERR_BAD_DIMS;                                                     \
        const long long planes = (long long)F * C;                                                                  \
        hipLaunchKernelGGL((k_se_scale_backward<T>), dim3(plane_grid(planes)), dim3(kBlock), 0,                     \
                           (hipStream_t)stream, (const T*)dy, (const T*)x, gate, (T*)dx, dgate, planes, P);         \
        return launch_status();                                                                                     \
    }
int rk_se_mlp_forward_f32(const float* q, const float* W1, const float* W2, float* h, float* g, int F, int C, int Cr,
                          rk_stream_t stream) {
    if (!q || !W1 || !W2 || !h || !g) return RK_ERR_NULL_POINTER;
    if (F <= 0 || C <= 0 || Cr <= 0) return RK_ERR_BAD_DIMS;
    if (C > kSeMaxC || Cr > kSeMaxCr) return RK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_se_mlp_forward, dim3((unsigned)F), dim3(kBlock), 0, (hipStream_t)stream, q, W1, W2, h, g, C, Cr);
    return launch_status();
}
int rk_se_mlp_backward_f32(const float* dgate, const float* g, const float* h, const float* q, const float* W1, const float* W2,
                           float* dpre2, float* dpre1, float* dq, float* dW1, float* dW2, int F, int C, int Cr,
                           rk_stream_t stream) {
    if (!dgate || !g || !h || !q || !W1 || !W2 || !dpre2 || !dpre1 || !dq || !dW1 || !dW2) return RK_ERR_NULL_POINTER;
    if (F <= 0 || C <= 0 || Cr <= 0) return RK_ERR_BAD_DIMS;
    if (C > kSeMaxC || Cr > kSeMaxCr) return RK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_se_mlp_backward, dim3((unsigned)F), dim3(kBlock), 0, (hipStream_t)stream, dgate, g, h, W1, W2, dpre2,
                       dpre1, dq, C, Cr);
    hipLaunchKernelGGL(k_se_mlp_wgrad, dim3((unsigned)((2 * C * Cr + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       dpre2, h, dpre1, q, dW1, dW2, F, C, Cr);
    return launch_status();
}
int rk_se_dgate_f32(const float* dy, const float* x, float* dgate, int F, int C, int P, rk_stream_t stream) {
    if (!dy || !x || !dgate) return RK_ERR_NULL_POINTER;
    if (F <= 0 || C <= 0 || P <= 0) return RK_ERR_BAD_DIMS;
    const long long planes = (long long)F * C;
    hipLaunchKernelGGL(k_se_dgate, dim3(plane_grid(planes)), dim3(kBlock), 0, (hipStream_t)stream, dy, x, dgate, planes, P);
    return launch_status();
}
int rk_se_scale_add_f32(const float* x, const float* gate, const float* add, float add_scale, float* y, int F, int C, int P,
                        rk_stream_t stream) {
    if (!x || !gate || !add || !y) return RK_ERR_NULL_POINTER;
    if (F <= 0 || C <= 0 || P <= 0) return RK_ERR_BAD_DIMS;
    const long long planes = (long long)F * C;
    hipLaunchKernelGGL(k_se_scale_add, dim3(plane_grid(planes)), dim3(kBlock), 0, (hipStream_t)stream, x, gate, add, add_scale,
                       y, planes, P);
    return launch_status();
}
RK_SE_IMPL(f32, float)
RK_SE_IMPL(bf16, __hip_bfloat16)
#undef RK_SE_IMPL

}  // extern "C"
