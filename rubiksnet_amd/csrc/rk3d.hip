// rk3d.hip -- C-ABI entry points of the RubiksShift3D operator (include/rubiks_hip.h).
// Host glue restating cuda_src/rubiks.cpp:161-379 (shape math, dispatch) without ATen:
// caller-owned buffers and workspace, explicit stream, error codes instead of exit().
#include "rk3d_generic.hpp"
#include "rk3d_dma.hpp"
#include "rk3d_plane.hpp"
#include "rk3d_tile.hpp"
#include "rk3d_translate.hpp"
#include "rk3d_stride2.hpp"
#include "rk3d_column.hpp"
#include "rk3d_slab.hpp"

#include <type_traits>

using namespace rk;

namespace {

__global__ __launch_bounds__(kWave) void k3d_debug_finalize_only(dma3d::Fin3 fin, int C, int P) {
    dma3d::finalizer_wave<3>(fin, (int)blockIdx.x, C, P);
}

int make_dims(Dims3& d, int N, int T, int C, int H, int W, int sT, int sH, int sW, int pT, int pH, int pW) {
    if (N <= 0 || T <= 0 || C <= 0 || H <= 0 || W <= 0) return RK_ERR_BAD_DIMS;
    if (sT <= 0 || sH <= 0 || sW <= 0 || pT < 0 || pH < 0 || pW < 0) return RK_ERR_BAD_STRIDE;
    d.N = N; d.T = T; d.C = C; d.H = H; d.W = W;
    d.sT = sT; d.sH = sH; d.sW = sW; d.pT = pT; d.pH = pH; d.pW = pW;
    d.To = out_len(T, sT, pT); d.Ho = out_len(H, sH, pH); d.Wo = out_len(W, sW, pW);
    if (d.To <= 0 || d.Ho <= 0 || d.Wo <= 0) return RK_ERR_BAD_DIMS;
    // the reference indexes with int (rubiks3d_kernels.cu:34-36); keep the same limit
    const long long nin = (long long)N * T * C * H * W, nout = (long long)N * d.To * C * d.Ho * d.Wo;
    if (nin > 0x7fffffffLL || nout > 0x7fffffffLL) return RK_ERR_BAD_DIMS;
    return RK_OK;
}

void set_group(Dims3& d, int plane_elems) {
    d.E = pow2_at_least(plane_elems, kWave, kBlock);
    d.logE = (d.E == 64) ? 6 : (d.E == 128 ? 7 : 8);
}

unsigned grid_for(const Dims3& d, long long planes) {
    const int per_block = kBlock / d.E;
    return (unsigned)((planes + per_block - 1) / per_block);
}

bool is_s1p0(const Dims3& d) {
    return d.sT == 1 && d.sH == 1 && d.sW == 1 && d.pT == 0 && d.pH == 0 && d.pW == 0;
}

template <typename T>
int forward_impl(const T* x, const T* shift, T* y, int N, int Tn, int C, int H, int W, int sT, int sH, int sW,
                 int pT, int pH, int pW, int quantize, rk_stream_t stream_) {
    if (!x || !shift || !y) return RK_ERR_NULL_POINTER;
    Dims3 d;
    if (int rc = make_dims(d, N, Tn, C, H, W, sT, sH, sW, pT, pH, pW)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if constexpr (std::is_same<T, float>::value) {
        if (!quantize && plane3d::launch_interp<false>(x, shift, y, d, stream)) return launch_status();
        if (!quantize && dma3d::launch_interp<false>(x, shift, y, d, stream)) return launch_status();
        const bool p14 = d.H == 14 && d.W == 14;
        if (!quantize && p14 && slab3d::slab14_on(false) && slab3d::launch_interp(false, x, shift, y, d, stream)) return launch_status();
        if (!quantize && tile3d::launch_interp<false>(x, shift, y, d, stream)) return launch_status();
        if (!quantize && !p14 && slab3d::launch_interp(false, x, shift, y, d, stream)) return launch_status();   // small planes
        if (quantize && xlate3d::launch<false>(x, shift, y, d, stream)) return launch_status();   // plane translation
        if (!quantize && s2::launch_forward(x, shift, y, d, stream)) return launch_status();       // stride (1,2,2)
        if (!quantize && slab3d::launch_fwd_s2(x, shift, y, d, stream)) return launch_status();    // stride (1,2,2), 28 -> 14 and 14 -> 7
    }
    if (col3d::supported(d, quantize)) return col3d::launch_forward<T>(x, shift, y, d, stream);
    set_group(d, d.Ho * d.Wo);
    const unsigned grid = grid_for(d, (long long)d.N * d.To * d.C);
    if (quantize)
        hipLaunchKernelGGL((k3d_forward_generic<T, true>), dim3(grid), dim3(kBlock), 0, stream, x, shift, y, d);
    else
        hipLaunchKernelGGL((k3d_forward_generic<T, false>), dim3(grid), dim3(kBlock), 0, stream, x, shift, y, d);
    return launch_status();
}

// `P_out` != nullptr: two-phase use -- stop after the partials (ws[C][3][*P_out]) and report their count instead of
// running the row-sum + K5 (rk3d_backward_finalize_* does that).
template <typename T>
int backward_impl(const T* x, const T* shift, const T* gy, T* gx, T* gshift, int N, int Tn, int C, int H, int W,
                  int sT, int sH, int sW, int pT, int pH, int pW, int normalize_grad, T t_factor, int quantize,
                  void* ws, size_t ws_bytes, rk_stream_t stream_, int* P_out = nullptr) {
    if (!shift || !gy || (!gx && !gshift)) return RK_ERR_NULL_POINTER;
    if (gshift && !x) return RK_ERR_NULL_POINTER;
    Dims3 d;
    if (int rc = make_dims(d, N, Tn, C, H, W, sT, sH, sW, pT, pH, pW)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (gshift) {
        const size_t need = rk3d_backward_workspace_bytes(N, Tn, C, H, W, sT, sH, sW, pT, pH, pW, (int)sizeof(T));
        if (!ws || ws_bytes < need) return RK_ERR_WORKSPACE;
    }
    // row-sum of the P partials per channel + K5 -- or, in two-phase use, just report P
    auto finish = [&](int P) {
        if (P_out) { *P_out = P; return launch_status(); }
        hipLaunchKernelGGL((k3d_finalize<T>), dim3(d.C), dim3(finalize_block(P)), 0, stream, (const T*)ws, gshift, d.C, P,
                           normalize_grad, t_factor);
        return launch_status();
    };
    if constexpr (std::is_same<T, float>::value) {
        // quantize: d(x) is a plane translation; d(shift) does not depend on quantize (K2 takes the fractional
        // shift, rubiks.cpp:324-358), so it comes from the streaming backward without its d(x) half
        if (quantize && gx && gshift && !P_out) {                       // both gradients, one launch (rk3d_dma.hpp QUANT)
            if (dma3d::launch_bwd(x, shift, gy, gx, gshift, (float*)ws, d, normalize_grad, t_factor, stream, true))
                return launch_status();
        }
        if (quantize && gx && xlate3d::launch<true>(gy, shift, gx, d, stream)) {
            if (!gshift) return launch_status();
            gx = nullptr;
            int P = dma3d::launch_bwd(x, shift, gy, nullptr, P_out ? nullptr : gshift, (float*)ws, d, normalize_grad,
                                      t_factor, stream);
            if (!P) P = tile3d::launch_bwd(x, shift, gy, nullptr, P_out ? nullptr : gshift, (float*)ws, d, normalize_grad,
                                           t_factor, stream);
            if (P) {
                if (P_out) *P_out = P;
                return launch_status();
            }
        }
        if (!quantize && gshift) {
            // one-call form: row-sum + K5 happen inside the launch; two-phase form: plain partials
            if (const int P = dma3d::launch_bwd(x, shift, gy, gx, P_out ? nullptr : gshift, (float*)ws, d, normalize_grad,
                                                t_factor, stream)) {
                if (P_out) *P_out = P;
                return launch_status();
            }
            {   // 14x14: the tile kernels (or, RK_SLAB14=1, the slab kernels first); other small planes: the slab kernels
                const bool p14 = d.H == 14 && d.W == 14;
                float* gs = P_out ? nullptr : gshift;
                int P = 0;
                if (p14 && slab3d::slab14_on(true)) P = slab3d::launch_bwd(x, shift, gy, gx, gs, (float*)ws, d, normalize_grad, t_factor, stream);
                if (!P) P = tile3d::launch_bwd(x, shift, gy, gx, gs, (float*)ws, d, normalize_grad, t_factor, stream);
                if (!P && !p14) P = slab3d::launch_bwd(x, shift, gy, gx, gs, (float*)ws, d, normalize_grad, t_factor, stream);
                if (P) {
                    if (P_out) *P_out = P;
                    return launch_status();
                }
            }
            if (const int P = s2::launch_backward(x, shift, gy, gx, P_out ? nullptr : gshift, (float*)ws, d, normalize_grad,
                                                  t_factor, stream)) {      // stride (1,2,2)
                if (P_out) *P_out = P;
                return launch_status();
            }
            if (const int P = slab3d::launch_bwd_s2(x, shift, gy, gx, P_out ? nullptr : gshift, (float*)ws, d, normalize_grad,
                                                    t_factor, stream)) {    // stride (1,2,2), 28 -> 14 and 14 -> 7
                if (P_out) *P_out = P;
                return launch_status();
            }
        } else if (!quantize && gx) {
            if (plane3d::launch_interp<true>(gy, shift, gx, d, stream)) return launch_status();
            if (dma3d::launch_interp<true>(gy, shift, gx, d, stream)) return launch_status();
            const bool p14 = d.H == 14 && d.W == 14;
            if (p14 && slab3d::slab14_on(false) && slab3d::launch_interp(true, gy, shift, gx, d, stream)) return launch_status();
            if (tile3d::launch_interp<true>(gy, shift, gx, d, stream)) return launch_status();
            if (!p14 && slab3d::launch_interp(true, gy, shift, gx, d, stream)) return launch_status();
        }
    }
    if (gshift && col3d::supported(d, quantize)) {
        // one-call form (fp32): row-sum + K5 inside the launch; two-phase form / fp64: partials, then k3d_finalize
        const int P = col3d::launch_backward<T>(x, shift, gy, gx, (T*)ws, d, stream, P_out ? nullptr : gshift, normalize_grad,
                                                t_factor);
        return P < 0 ? launch_status() : finish(P);
    }

    if (gx) {       // rubiks.cpp:363-376
        set_group(d, d.H * d.W);
        const unsigned grid = grid_for(d, (long long)d.N * d.T * d.C);
        if (quantize)
            hipLaunchKernelGGL((k3d_backward_input_generic<T, true>), dim3(grid), dim3(kBlock), 0, stream, shift, gy,
                               gx, d);
        else
            hipLaunchKernelGGL((k3d_backward_input_generic<T, false>), dim3(grid), dim3(kBlock), 0, stream, shift,
                               gy, gx, d);
    }
    if (gshift) {   // rubiks.cpp:324-358
        set_group(d, d.Ho * d.Wo);
        hipLaunchKernelGGL((k3d_backward_shift_generic<T>), dim3(grid_for(d, (long long)d.N * d.To * d.C)),
                           dim3(kBlock), 0, stream, x, shift, gy, (T*)ws, d);
        return finish(d.N * d.To);
    }
    return launch_status();
}

}  // namespace

extern "C" {

size_t rk3d_backward_workspace_bytes(int N, int T, int C, int H, int W, int sT, int sH, int sW, int pT, int pH,
                                     int pW, int elem_size) {
    (void)sH; (void)sW; (void)pH; (void)pW;
    if (N <= 0 || T <= 0 || C <= 0 || sT <= 0 || pT < 0) return 0;
    // partials part[C][3][P]: P = N*To for the generic kernels, N*nbands (row bands) for the streaming ones
    size_t per_n = (size_t)out_len(T, sT, pT);
    if (H > 0 && (size_t)H > per_n) per_n = (size_t)H;    // row bands: nbands <= H
    if (H > 0 && W > 0) {                                 // column kernels: 256-element chunks at most
        const size_t chunks = ((size_t)H * (size_t)W + 255) / 256;
        if (chunks > per_n) per_n = chunks;
    }
    const size_t P = (size_t)N * per_n;
    // fp32: the streaming backward keeps its partials as 16-byte granule pairs (rk_dma.hpp: fin_publish)
    return (size_t)C * 3 * P * (size_t)(elem_size == 4 ? 16 : elem_size);
}

// ---- training fusion: the shift applied to relu(bn(z)) without the activation ever being stored (train_block.py) ----
// Both return RK_ERR_UNSUPPORTED (no launch, nothing touched) when no fused kernel covers the configuration; the caller
// then normalises with rk_bn_apply_affine_f32 and calls the plain entry points.
int rk3d_forward_bn_f32(const float* z, const float* abmi, const float* shift, float* y, int N, int T, int C, int H,
                        int W, int sT, int sH, int sW, int pT, int pH, int pW, int quantize, rk_stream_t stream_) {
    if (!z || !abmi || !shift || !y) return RK_ERR_NULL_POINTER;
    Dims3 d;
    if (int rc = make_dims(d, N, T, C, H, W, sT, sH, sW, pT, pH, pW)) return rc;
    if (quantize) return RK_ERR_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    const float4* pk = reinterpret_cast<const float4*>(abmi);
    if (plane3d::launch_forward_bn(z, shift, y, pk, d, stream)) return launch_status();
    if (dma3d::launch_forward_bn(z, shift, y, pk, d, stream)) return launch_status();
    if (tile3d::launch_forward_bn(z, shift, y, pk, d, stream)) return launch_status();
    if (s2::launch_forward_bn(z, shift, y, pk, d, stream)) return launch_status();
    return RK_ERR_UNSUPPORTED;
}
size_t rk3d_backward_bn_workspace_bytes(int N, int T, int C, int H, int W, int sT, int sH, int sW, int pT, int pH, int pW) {
    // five partials per (channel, column-band) instead of three, as 16-byte granule pairs
    return rk3d_backward_workspace_bytes(N, T, C, H, W, sT, sH, sW, pT, pH, pW, 4) / 3 * 5;
}
int rk3d_backward_bn_f32(const float* z, const float* abmi, const float* shift, const float* gy, float* dz, float* gshift,
                         float* k12, float* dgamma, float* dbeta, int N, int T, int C, int H, int W, int sT, int sH,
                         int sW, int pT, int pH, int pW, int normalize_grad, float t_factor, int quantize, void* ws,
                         size_t ws_bytes, rk_stream_t stream_) {
    if (!z || !abmi || !shift || !gy || !dz || !gshift || !k12 || !dgamma || !dbeta) return RK_ERR_NULL_POINTER;
    Dims3 d;
    if (int rc = make_dims(d, N, T, C, H, W, sT, sH, sW, pT, pH, pW)) return rc;
    if (!ws || ws_bytes < rk3d_backward_bn_workspace_bytes(N, T, C, H, W, sT, sH, sW, pT, pH, pW)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    dma3d::BnFuse bn;
    bn.abmi = reinterpret_cast<const float4*>(abmi);
    bn.k12 = k12; bn.dgamma = dgamma; bn.dbeta = dbeta;
    bn.inv_count = (float)(1.0 / ((double)N * T * H * W));
    if (dma3d::launch_bwd_bn(z, shift, gy, dz, gshift, (float*)ws, d, normalize_grad, t_factor, quantize, bn, stream))
        return launch_status();
    if (!quantize && tile3d::launch_bwd_bn(z, shift, gy, dz, gshift, (float*)ws, d, normalize_grad, t_factor, bn, stream))
        return launch_status();
    if (!quantize && s2::launch_backward_bn(z, shift, gy, dz, gshift, (float*)ws, d, normalize_grad, t_factor, bn, stream))
        return launch_status();
    return RK_ERR_UNSUPPORTED;
}

int rk3d_forward_f32(const float* x, const float* shift, float* y, int N, int T, int C, int H, int W, int sT,
                     int sH, int sW, int pT, int pH, int pW, int quantize, rk_stream_t stream) {
    return forward_impl<float>(x, shift, y, N, T, C, H, W, sT, sH, sW, pT, pH, pW, quantize, stream);
}
int rk3d_forward_f64(const double* x, const double* shift, double* y, int N, int T, int C, int H, int W, int sT,
                     int sH, int sW, int pT, int pH, int pW, int quantize, rk_stream_t stream) {
    return forward_impl<double>(x, shift, y, N, T, C, H, W, sT, sH, sW, pT, pH, pW, quantize, stream);
}
int rk3d_backward_f32(const float* x, const float* shift, const float* gy, float* gx, float* gshift, int N, int T,
                      int C, int H, int W, int sT, int sH, int sW, int pT, int pH, int pW, int normalize_grad,
                      float t_factor, int quantize, void* ws, size_t ws_bytes, rk_stream_t stream) {
    return backward_impl<float>(x, shift, gy, gx, gshift, N, T, C, H, W, sT, sH, sW, pT, pH, pW, normalize_grad,
                                t_factor, quantize, ws, ws_bytes, stream);
}
int rk3d_backward_f64(const double* x, const double* shift, const double* gy, double* gx, double* gshift, int N,
                      int T, int C, int H, int W, int sT, int sH, int sW, int pT, int pH, int pW,
                      int normalize_grad, double t_factor, int quantize, void* ws, size_t ws_bytes,
                      rk_stream_t stream) {
    return backward_impl<double>(x, shift, gy, gx, gshift, N, T, C, H, W, sT, sH, sW, pT, pH, pW, normalize_grad,
                                 t_factor, quantize, ws, ws_bytes, stream);
}

// Two-phase form of the fp32 backward (the reference's own host glue has these phases, rubiks.cpp:324-376: K2 + K3/K4,
// then addmv row-sum + K5): phase 1 writes d(x) and the per-channel partials ws[C][3][P] and returns P through
// *partials; phase 2 sums them and normalises.  rk3d_backward_f32 == phase 1 + phase 2.
int rk3d_backward_partials_f32(const float* x, const float* shift, const float* gy, float* gx, int N, int T, int C,
                               int H, int W, int sT, int sH, int sW, int pT, int pH, int pW, int quantize, void* ws,
                               size_t ws_bytes, int* partials, rk_stream_t stream) {
    if (!partials || !ws) return RK_ERR_NULL_POINTER;
    float* not_null = (float*)ws;        // "d(shift) wanted": the partials land in ws, nothing is written through this
    return backward_impl<float>(x, shift, gy, gx, not_null, N, T, C, H, W, sT, sH, sW, pT, pH, pW, 0, 1.0f, quantize, ws,
                                ws_bytes, stream, partials);
}
int rk3d_backward_finalize_f32(const void* ws, int C, int partials, float* gshift, int normalize_grad, float t_factor,
                               rk_stream_t stream) {
    if (!ws || !gshift) return RK_ERR_NULL_POINTER;
    if (C <= 0 || partials <= 0) return RK_ERR_BAD_DIMS;
    hipLaunchKernelGGL((k3d_finalize<float>), dim3(C), dim3(finalize_block(partials)), 0, (hipStream_t)stream,
                       (const float*)ws, gshift, C, partials, normalize_grad, t_factor);
    return launch_status();
}

// Test hook: ONLY the finalizer waves of a fused 3-D backward (rk3d_dma.hpp, finalizer_wave<3>) on a workspace no producer
// will ever publish to -- the give-up path: every channel must come back NaN once the poll budget
// (rk_debug_set_finalize_spins) is spent, and the launch must end.  No product code calls it.
int rk3d_debug_finalize_only_f32(void* ws, size_t ws_bytes, int C, int partials, float* gshift, int normalize_grad,
                                 float t_factor, rk_stream_t stream) {
    if (!ws || !gshift) return RK_ERR_NULL_POINTER;
    if (C <= 0 || partials <= 0) return RK_ERR_BAD_DIMS;
    if (ws_bytes < (size_t)C * 3 * partials * 16) return RK_ERR_WORKSPACE;
    dma3d::Fin3 fin{};
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    dma::fin_arm(fin.f);
    fin.f.producers = 0;
    fin.gshift = gshift;
    fin.normalize = normalize_grad;
    fin.t_factor = t_factor;
    hipLaunchKernelGGL(k3d_debug_finalize_only, dim3((unsigned)C), dim3(kWave), 0, (hipStream_t)stream, fin, C, partials);
    return launch_status();
}

}  // extern "C"
