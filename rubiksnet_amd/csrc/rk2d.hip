// rk2d.hip -- RubiksShift2D for gfx950: kernels + C-ABI entry points (include/rubiks_hip.h).
//
// Semantics follow the reference's K6-K9 (cuda_src/rubiks2d_kernels.cu:94-397) and its
// host glue (cuda_src/rubiks.cpp:44-155).  Layout [N,C,H,W], shift [2,C] = (H,W).
// Same mapping idea as rk3d_generic.hpp: one (n, c) plane per group of E = 64/128/256
// threads, per-channel quantities hoisted, no per-element division, no float atomics:
// d(shift) goes wave-shuffle -> LDS -> one partial per plane -> fixed-order fp64 finalize.
// f16 / bf16 tensors are computed in fp32 and rounded once on store; only the quantize
// position arithmetic is done in the storage type, because it decides WHICH element is
// gathered (the reference instantiates the whole kernel at c10::Half).
#include <type_traits>
#include "rk2d_generic.hpp"
#include "rk2d_dma.hpp"
#include "rk2d_stage.hpp"
#include "rk2d_raw16.hpp"
#include "rk2d_tile.hpp"
#include "rk2d_column.hpp"

using namespace rk;
using namespace rk::g2d;

namespace {

int make_dims2(Dims2& d, int N, int C, int H, int W, int sH, int sW, int pH, int pW) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return RK_ERR_BAD_DIMS;
    if (sH <= 0 || sW <= 0 || pH < 0 || pW < 0) return RK_ERR_BAD_STRIDE;
    d.N = N; d.C = C; d.H = H; d.W = W; d.sH = sH; d.sW = sW; d.pH = pH; d.pW = pW;
    d.Ho = out_len(H, sH, pH); d.Wo = out_len(W, sW, pW);
    if (d.Ho <= 0 || d.Wo <= 0) return RK_ERR_BAD_DIMS;
    // the reference's accessors index with uint32 (utils_cuda.h:12-13)
    if ((long long)N * C * H * W > 0x7fffffffLL || (long long)N * C * d.Ho * d.Wo > 0x7fffffffLL)
        return RK_ERR_BAD_DIMS;
    return RK_OK;
}

void set_group2(Dims2& d, int plane_elems) {
    d.E = pow2_at_least(plane_elems, kWave, kBlock);
    d.logE = (d.E == 64) ? 6 : (d.E == 128 ? 7 : 8);
}

// bytes of d(shift) partials: [C][2][P], P = N for the generic kernels
size_t workspace2(const Dims2& d, size_t elem) {
    // the smaller group size gives the larger partial count: an upper bound for every storage type
    int P = dma2d::backward2_partials(d, dma2d::kFramesF32 < dma2d::kFrames16 ? dma2d::kFramesF32 : dma2d::kFrames16);
    const int Pc = col2d::backward_partials(d), Pr = raw16::backward2_partials(d);
    P = P > Pc ? P : Pc;
    P = P > Pr ? P : Pr;
    const int Pt = tile2d::backward2_partials<float>(d);             // (the same count for every storage type)
    P = P > Pt ? P : Pt;
    // the streaming backwards keep their fp32 partials as 16-byte granule pairs (rk_dma.hpp: fin_publish)
    return (size_t)d.C * 2 * (size_t)(P > d.N ? P : d.N) * 16;
}

unsigned grid2(const Dims2& d) {
    const int per_block = kBlock / d.E;
    return (unsigned)(((long long)d.N * d.C + per_block - 1) / per_block);
}

// T: storage type of the activations; S: storage type of the shift table and of d(shift).  S == T is the
// reference's instantiation (the whole kernel at one scalar type, rubiks2d_kernels.cu:422); S = float next to
// 16-bit activations keeps the fp32 parameter of an autocast network un-rounded (and returns d(shift) in fp32).
template <typename T, typename S>
int forward2(const void* x_, const void* shift_, void* y_, int N, int C, int H, int W, int sH, int sW, int pH,
             int pW, int quantize, rk_stream_t stream_) {
    const T* x = (const T*)x_; const S* shift = (const S*)shift_; T* y = (T*)y_;
    if (!x || !shift || !y) return RK_ERR_NULL_POINTER;
    Dims2 d;
    if (int rc = make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if constexpr (!std::is_same<T, double>::value) {
        if (!quantize && tile2d::launch_interp2<T, false>(x, shift, y, d, stream)) return launch_status();   // 14x14
    }
    if constexpr (std::is_same<T, float>::value) {
        if (!quantize && dma2d::launch_interp2<false>(x, shift, y, d, stream)) return launch_status();
    } else if constexpr (!std::is_same<T, double>::value) {
        if (!quantize && raw16::launch_interp2<T, false>(x, shift, y, d, stream)) return launch_status();   // W % 8 == 0
        if (!quantize && stage2d::launch_interp2<T, false>(x, shift, y, d, stream)) return launch_status();
    }
    // small planes only: on larger ones the per-plane kernel below is the faster forward (stride-2 56x56 ->
    // 28x28: 85 vs 102 us); the column BACKWARD wins everywhere (fused, single-tap)
    // (16-bit strided layers: the column kernel with 8-byte stores also wins on the large planes, 108 -> 84 us at
    // [256,108,56,56] bf16; in fp32 it loses there, 84 -> 97 us)
    if (col2d::supported(quantize) && (d.Ho * d.Wo <= kBlock || (sizeof(T) == 2 && (d.sH > 1 || d.sW > 1)))) {
        col2d::launch_forward<T>(x, shift, y, d, stream);
        return launch_status();
    }
    set_group2(d, d.Ho * d.Wo);
    if (quantize)
        hipLaunchKernelGGL((k2d_forward<T, S, true>), dim3(grid2(d)), dim3(kBlock), 0, stream, x, shift, y, d);
    else
        hipLaunchKernelGGL((k2d_forward<T, S, false>), dim3(grid2(d)), dim3(kBlock), 0, stream, x, shift, y, d);
    return launch_status();
}

template <typename T, typename S>
int backward2(const void* gy_, const void* x_, const void* shift_, void* gx_, void* gshift_, int N, int C, int H,
              int W, int sH, int sW, int pH, int pW, int normalize_grad, int enable_shift_grad, int quantize,
              void* ws, size_t ws_bytes, rk_stream_t stream_) {
    using CT = typename Compute<T>::type;
    const T* gy = (const T*)gy_; const T* x = (const T*)x_; const S* shift = (const S*)shift_;
    T* gx = (T*)gx_; S* gshift = (S*)gshift_;
    if (!gy || !shift || !gx) return RK_ERR_NULL_POINTER;
    if (enable_shift_grad && (!x || !gshift)) return RK_ERR_NULL_POINTER;
    Dims2 d;
    if (int rc = make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (enable_shift_grad) {
        if (!ws || ws_bytes < workspace2(d, sizeof(CT))) return RK_ERR_WORKSPACE;
    }
    if constexpr (!std::is_same<T, double>::value) {
        if (!quantize) {                                                                       // 14x14 planes
            if (!enable_shift_grad) {
                if (tile2d::launch_interp2<T, true>(gy, shift, gx, d, stream)) return launch_status();
            } else if (tile2d::launch_backward2<T>(gy, x, shift, gx, gshift, ws, normalize_grad, d, stream)) {
                return launch_status();
            }
        }
    }
    if constexpr (std::is_same<T, float>::value) {
        if (!quantize) {
            if (!enable_shift_grad) {
                if (dma2d::launch_interp2<true>(gy, shift, gx, d, stream)) return launch_status();
            } else if (dma2d::launch_backward2(gy, x, shift, gx, gshift, ws, normalize_grad, d, stream)) {
                return launch_status();                               // row-sum + K9 happened inside the launch
            }
        }
    } else if constexpr (!std::is_same<T, double>::value) {
        if (!quantize) {
            if (!enable_shift_grad) {
                if (raw16::launch_interp2<T, true>(gy, shift, gx, d, stream)) return launch_status();
                if (stage2d::launch_interp2<T, true>(gy, shift, gx, d, stream)) return launch_status();
            } else if (raw16::launch_backward2<T>(gy, x, shift, gx, gshift, ws, normalize_grad, d, stream) ||
                       stage2d::launch_backward2<T>(gy, x, shift, gx, gshift, ws, normalize_grad, d, stream)) {
                return launch_status();
            }
        }
    }
    if (enable_shift_grad && col2d::supported(quantize)) {                // fused d(x) + d(shift), any stride / H x W
        const int P = col2d::launch_backward<T>(gy, x, shift, gx, (CT*)ws, d, stream);
        hipLaunchKernelGGL((k2d_finalize<T, S>), dim3(C), dim3(finalize_block(P)), 0, stream, (const CT*)ws, gshift, C, P,
                           normalize_grad);
        return launch_status();
    }
    if (enable_shift_grad) {                                              // rubiks.cpp:126-149
        CT* part = (CT*)ws;
        set_group2(d, d.Ho * d.Wo);
        hipLaunchKernelGGL((k2d_backward_shift<T, S>), dim3(grid2(d)), dim3(kBlock), 0, stream, gy, x, shift, part, d);
        hipLaunchKernelGGL((k2d_finalize<T, S>), dim3(C), dim3(finalize_block(N)), 0, stream, (const CT*)part, gshift, C, N,
                           normalize_grad);
    }
    set_group2(d, d.H * d.W);                                             // rubiks.cpp:151-153
    if (quantize)
        hipLaunchKernelGGL((k2d_backward_input<T, S, true>), dim3(grid2(d)), dim3(kBlock), 0, stream, gy, shift, gx, d);
    else
        hipLaunchKernelGGL((k2d_backward_input<T, S, false>), dim3(grid2(d)), dim3(kBlock), 0, stream, gy, shift, gx, d);
    return launch_status();
}

// ---- training fusion (round 5): the shift applied to relu(bn2(z)) without the activation ever being stored
// (fused_bn.bn_relu_shift2d; backbone.py:129-131 under models.py:71-79's 2-D variant).  RK_ERR_UNSUPPORTED (nothing launched) when
// no fused kernel covers the configuration (quantize; column kernels switched off and a shape the streaming kernels do not take); the caller then normalises
// with rk_bn_apply_affine_* and calls the plain entry points.
// the column kernels' fused variants take what no streaming kernel does (fp32 planes the LDS-DMA kernels stream keep the
// normalise pass + those kernels: they are the faster pair there)
static bool col_bn_takes(const Dims2& d, int elem_size) {
    if (!col2d::supported(0)) return false;
    if (elem_size == 4) {
        dma2d::FDims f;
        if (dma2d::make_fdims(f, d, dma2d::kFramesF32, true) && bwd_ring_bytes(f.b, 1, 1) <= 64 * 1024) return false;
    }
    return true;
}
template <typename T>
static int forward2_bn(const void* z, const float* ab, const float* shift, void* y, int N, int C, int H, int W, int sH, int sW,
                       int pH, int pW, int quantize, rk_stream_t stream) {
    if (!z || !ab || !shift || !y) return RK_ERR_NULL_POINTER;
    Dims2 d;
    if (int rc = make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return rc;
    if (quantize) return RK_ERR_UNSUPPORTED;
    if (tile2d::launch_forward2_bn<T, float>((const T*)z, ab, shift, (T*)y, d, (hipStream_t)stream)) return launch_status();
    if constexpr (sizeof(T) == 2) {                                  // raw 16-bit planes: 56 x 56, 112 x 112
        if (raw16::launch_forward2_bn<T, float>((const T*)z, ab, shift, (T*)y, d, (hipStream_t)stream)) return launch_status();
        if (stage2d::launch_forward2_bn<T, float>((const T*)z, ab, shift, (T*)y, d, (hipStream_t)stream)) return launch_status();   // 28 x 28
    }
    if (col_bn_takes(d, (int)sizeof(T))) {                           // the strided layers and the 7 x 7 planes
        col2d::launch_forward_bn<T>((const T*)z, ab, shift, (T*)y, d, (hipStream_t)stream);
        return launch_status();
    }
    return RK_ERR_UNSUPPORTED;
}
template <typename T>
static int backward2_bn(const void* gy, const void* z, const float* abmi, const float* shift, void* dz, float* gshift, float* k12,
                        float* dgamma, float* dbeta, int N, int C, int H, int W, int sH, int sW, int pH, int pW,
                        int normalize_grad, int quantize, void* ws, size_t ws_bytes, rk_stream_t stream) {
    if (!gy || !z || !abmi || !shift || !dz || !gshift || !k12 || !dgamma || !dbeta) return RK_ERR_NULL_POINTER;
    Dims2 d;
    if (int rc = make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return rc;
    if (quantize) return RK_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < 2 * workspace2(d, 4)) return RK_ERR_WORKSPACE;
    dma2d::BnFuse2 bn;
    bn.abmi = reinterpret_cast<const float4*>(abmi);
    bn.k12 = k12; bn.dgamma = dgamma; bn.dbeta = dbeta;
    bn.inv_count = (float)(1.0 / ((double)N * H * W));
    if (tile2d::launch_backward2_bn<T, float>((const T*)gy, (const T*)z, shift, (T*)dz, gshift, ws, normalize_grad, bn, d,
                                              (hipStream_t)stream))
        return launch_status();
    if constexpr (sizeof(T) == 2) {
        if (raw16::launch_backward2_bn<T, float>((const T*)gy, (const T*)z, shift, (T*)dz, gshift, ws, normalize_grad, bn, d,
                                                 (hipStream_t)stream))
            return launch_status();
        if (stage2d::launch_backward2_bn<T, float>((const T*)gy, (const T*)z, shift, (T*)dz, gshift, ws, normalize_grad, bn, d,
                                                   (hipStream_t)stream))
            return launch_status();
    }
    if (col_bn_takes(d, (int)sizeof(T))) {
        const int P = col2d::launch_backward_bn<T>((const T*)gy, (const T*)z, shift, (T*)dz, (float*)ws, bn.abmi, d,
                                                   (hipStream_t)stream);
        hipLaunchKernelGGL(col2d::k2d_finalize_bn, dim3(C), dim3(finalize_block(P)), 0, (hipStream_t)stream, (const float*)ws,
                           gshift, k12, dgamma, dbeta, C, P, normalize_grad, bn.inv_count);
        return launch_status();
    }
    return RK_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

size_t rk2d_backward_workspace_bytes(int N, int C, int H, int W, int sH, int sW, int pH, int pW, int elem_size) {
    Dims2 d;
    if (make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return 0;
    return workspace2(d, elem_size == 8 ? 8 : 4);                  // partials are fp32 (fp64 for f64)
}

#define RK_DEF_2D(SFX, TYPE, CTYPE)                                                                              \
    int rk2d_forward_##SFX(const CTYPE* x, const CTYPE* shift, CTYPE* y, int N, int C, int H, int W, int sH,     \
                           int sW, int pH, int pW, int quantize, rk_stream_t stream) {                           \
        return forward2<TYPE, TYPE>(x, shift, y, N, C, H, W, sH, sW, pH, pW, quantize, stream);                      \
    }                                                                                                            \
    int rk2d_backward_##SFX(const CTYPE* gy, const CTYPE* x, const CTYPE* shift, CTYPE* gx, CTYPE* gshift,       \
                            int N, int C, int H, int W, int sH, int sW, int pH, int pW, int normalize_grad,     \
                            int enable_shift_grad, int quantize, void* ws, size_t ws_bytes, rk_stream_t stream) { \
        return backward2<TYPE, TYPE>(gy, x, shift, gx, gshift, N, C, H, W, sH, sW, pH, pW, normalize_grad,             \
                               enable_shift_grad, quantize, ws, ws_bytes, stream);                               \
    }
RK_DEF_2D(f32, float, float)
RK_DEF_2D(f64, double, double)
RK_DEF_2D(f16, __half, void)
RK_DEF_2D(bf16, __hip_bfloat16, void)
#undef RK_DEF_2D

// 16-bit activations with the shift table / d(shift) in fp32 (what autocast hands the operator: bf16 activations next
// to an fp32 nn.Parameter).  Everything that depends on the shift -- floor / remainder, the 1e-7 integer test, the
// quantize position arithmetic -- is then evaluated in fp32 exactly as by rk2d_*_f32.
#define RK_DEF_2D_MIXED(SFX, TYPE)                                                                               \
    int rk2d_forward_##SFX##_sf32(const void* x, const float* shift, void* y, int N, int C, int H, int W, int sH, \
                                  int sW, int pH, int pW, int quantize, rk_stream_t stream) {                    \
        return forward2<TYPE, float>(x, shift, y, N, C, H, W, sH, sW, pH, pW, quantize, stream);                 \
    }                                                                                                            \
    int rk2d_backward_##SFX##_sf32(const void* gy, const void* x, const float* shift, void* gx, float* gshift,   \
                                   int N, int C, int H, int W, int sH, int sW, int pH, int pW,                   \
                                   int normalize_grad, int enable_shift_grad, int quantize, void* ws,            \
                                   size_t ws_bytes, rk_stream_t stream) {                                        \
        return backward2<TYPE, float>(gy, x, shift, gx, gshift, N, C, H, W, sH, sW, pH, pW, normalize_grad,      \
                                      enable_shift_grad, quantize, ws, ws_bytes, stream);                        \
    }
RK_DEF_2D_MIXED(f16, __half)
RK_DEF_2D_MIXED(bf16, __hip_bfloat16)
#undef RK_DEF_2D_MIXED

// 1 when rk2d_forward_bn_* / rk2d_backward_bn_* have a fused kernel for this shape and storage size (4: fp32, 2: bf16), else 0:
// lets a caller decide BEFORE it runs bn2's statistics (whose side effects -- running statistics -- must happen once)
int rk2d_bn_fused_shape(int N, int C, int H, int W, int sH, int sW, int pH, int pW, int elem_size) {
    Dims2 d;
    if (make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return 0;
    if (elem_size != 4 && elem_size != 2) return 0;
    if (col_bn_takes(d, elem_size)) return 1;                       // the column kernels take what the others below do not
    if (elem_size == 4) { tile2d::TDims2 t; return tile2d::make_tdims<float, 14, 14>(t, d) ? 1 : 0; }
    tile2d::TDims2 t;
    if (tile2d::make_tdims<__hip_bfloat16, 14, 14>(t, d)) return 1;
    dma2d::FDims f;
    if (raw16::make_fdims8(f, d, raw16::kFramesRaw16) && bwd_ring_bytes(f.b, 1, 1) <= 64 * 1024 &&
        raw16::make_fdims8(f, d, raw16::kFramesRaw16Fwd) && interp_ring_bytes(f.b, 2) <= 64 * 1024)
        return 1;
    if (dma2d::make_fdims(f, d, dma2d::kFrames16)) return 1;         // register-staged 16-bit planes: other W % 4 == 0 (28 x 28)
    return 0;
}
size_t rk2d_backward_bn_workspace_bytes(int N, int C, int H, int W, int sH, int sW, int pH, int pW) {
    Dims2 d;
    if (make_dims2(d, N, C, H, W, sH, sW, pH, pW)) return 0;
    return 2 * workspace2(d, 4);                                   // four partials per (channel, group) instead of two
}
int rk2d_forward_bn_f32(const float* z, const float* ab, const float* shift, float* y, int N, int C, int H, int W, int sH,
                        int sW, int pH, int pW, int quantize, rk_stream_t stream) {
    return forward2_bn<float>(z, ab, shift, y, N, C, H, W, sH, sW, pH, pW, quantize, stream);
}
int rk2d_forward_bn_bf16_sf32(const void* z, const float* ab, const float* shift, void* y, int N, int C, int H, int W, int sH,
                              int sW, int pH, int pW, int quantize, rk_stream_t stream) {
    return forward2_bn<__hip_bfloat16>(z, ab, shift, y, N, C, H, W, sH, sW, pH, pW, quantize, stream);
}
int rk2d_backward_bn_f32(const float* gy, const float* z, const float* abmi, const float* shift, float* dz, float* gshift,
                         float* k12, float* dgamma, float* dbeta, int N, int C, int H, int W, int sH, int sW, int pH, int pW,
                         int normalize_grad, int quantize, void* ws, size_t ws_bytes, rk_stream_t stream) {
    return backward2_bn<float>(gy, z, abmi, shift, dz, gshift, k12, dgamma, dbeta, N, C, H, W, sH, sW, pH, pW, normalize_grad,
                               quantize, ws, ws_bytes, stream);
}
int rk2d_backward_bn_bf16_sf32(const void* gy, const void* z, const float* abmi, const float* shift, void* dz, float* gshift,
                               float* k12, float* dgamma, float* dbeta, int N, int C, int H, int W, int sH, int sW, int pH,
                               int pW, int normalize_grad, int quantize, void* ws, size_t ws_bytes, rk_stream_t stream) {
    return backward2_bn<__hip_bfloat16>(gy, z, abmi, shift, dz, gshift, k12, dgamma, dbeta, N, C, H, W, sH, sW, pH, pW,
                                        normalize_grad, quantize, ws, ws_bytes, stream);
}

}  // extern "C"
