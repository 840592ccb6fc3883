// rk_bn.hip -- fused BatchNorm2d (+ ReLU) for the pre-activation blocks of the backbone (SURVEY 8(f) f3:
// "remaining block glue as fused elementwise kernels").  Every BatchNorm2d of the network is followed by a
// ReLU (rubiksnet/backbone.py:129-131, :196); with stock kernels the pair costs BN (MIOpen: 1.1 TB/s
// effective forward, 2.4 TB/s backward on the Tiny train step) + an elementwise ReLU pass each way, 39 % +
// 7.5 % of the step.  Here:
//   training forward : k_bn_stats  (read x)              -> per-(channel, frame group) shifted sums
//                      k_bn_apply  (read x, write y)     -> each workgroup re-derives its channel's mean / var
//                                                            from the partials (fixed order, fp64), applies
//                                                            y = max(a x + b, 0); group 0 also writes the saved
//                                                            mean / invstd and updates the running statistics
//   backward         : k_bn_bwd_reduce (read dy, x)      -> partial sum(dz), sum(dz * xhat), dz = dy * [y > 0]
//                      k_bn_bwd_dx     (read dy, x, write dx; group 0 writes dgamma, dbeta)
//   eval forward     : k_bn_apply with the running statistics
// 12 B/elem forward, 20 B/elem backward (fp32), no atomics (deterministic), the ReLU mask is recomputed from
// x with the forward's own expression (nothing but x is saved for backward).
// Layout: x [F, C, P] (NCHW with F = N*T frames, P = H*W); a workgroup owns (channel c, group of FB frames)
// and sweeps its FB planes as 16-byte cells when P % 4 == 0 (every plane of the networks but 7x7).
// Variance: sums of (x - K) and (x - K)^2 with K = x[0, c, 0], combined in fp64 -- the shifted-data form, immune
// to the E[x^2] - mean^2 cancellation.  Parameters and statistics are fp32 whatever the storage type.
#include "rk_common.hpp"
#include "rk_dma.hpp"

namespace rk {
namespace bn {

struct BnDims {
    int F, C, P;        // frames, channels, plane elements
    int FB, G;          // frames per workgroup, groups per channel
};

template <typename T, int VEC> struct Pack;
template <> struct Pack<float, 4> {
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Pack<__hip_bfloat16, 4> {
    __device__ static __forceinline__ void load(const __hip_bfloat16* p, float (&v)[4]) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
    __device__ static __forceinline__ unsigned bits(float f) {
        return (unsigned)__builtin_bit_cast(unsigned short, __float2bfloat16(f));
    }
    __device__ static __forceinline__ void store(__hip_bfloat16* p, const float (&v)[4]) {
        *reinterpret_cast<uint2*>(p) = make_uint2(bits(v[0]) | (bits(v[1]) << 16), bits(v[2]) | (bits(v[3]) << 16));
    }
};
template <> struct Pack<__hip_bfloat16, 8> {                     // 16-byte cells (planes of a multiple of 8 elements)
    __device__ static __forceinline__ void load(const __hip_bfloat16* p, float (&v)[8]) {
        const uint4 r = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
    }
};
template <typename T> struct Pack<T, 1> {
    __device__ static __forceinline__ void load(const T* p, float (&v)[1]) { v[0] = ld(p); }
    __device__ static __forceinline__ void store(T* p, const float (&v)[1]) { st(p, v[0]); }
};

struct Where { int c, g, f0, nf; };
__device__ __forceinline__ Where where_am_i(const BnDims& d) {
    Where w;
    w.c = blockIdx.x % d.C;                   // channel fastest: consecutive workgroups sweep memory in order
    w.g = blockIdx.x / d.C;
    w.f0 = w.g * d.FB;
    w.nf = min(d.FB, d.F - w.f0);
    return w;
}

// visit the workgroup's elements VEC at a time: fn(element offset into the tensor)
template <int VEC, int UNROLL = 4, typename Fn>
__device__ __forceinline__ void sweep(const BnDims& d, const Where& w, Fn fn) {
    const int PV = d.P / VEC;
    const int total = w.nf * PV;
    const size_t base = ((size_t)w.f0 * d.C + w.c) * d.P;
    const size_t fstride = (size_t)d.C * d.P;
#pragma unroll UNROLL
    for (int j = threadIdx.x; j < total; j += kBlock) {
        const int fr = j / PV, i = j - fr * PV;
        fn(base + (size_t)fr * fstride + (size_t)i * VEC);
    }
}

// fixed-order fp64 sum of the channel's G partial pairs; every workgroup of the channel gets identical values
__device__ __forceinline__ void channel_sums(const float* __restrict__ part, int c, int G, double (&out)[2],
                                             double (*smem)[2]) {
    double a = 0, b = 0;
    if (threadIdx.x < kWave) {
        const float* p = part + (size_t)c * G * 2;
        for (int i = threadIdx.x; i < G; i += kWave) { a += (double)p[2 * i]; b += (double)p[2 * i + 1]; }
        a = wave_sum(a);
        b = wave_sum(b);
        if (threadIdx.x == 0) { smem[0][0] = a; smem[0][1] = b; }
    }
    __syncthreads();
    out[0] = smem[0][0];
    out[1] = smem[0][1];
}

__device__ __forceinline__ void block_pair_sum(float a, float b, float* out, float (*red)[kBlock / kWave]) {
    a = group_sum(a, kBlock, red[0]);
    b = group_sum(b, kBlock, red[1]);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_bn_stats(const T* __restrict__ x, float* __restrict__ part, BnDims d) {
    __shared__ float red[2][kBlock / kWave];
    const Where w = where_am_i(d);
    const float K = ld(x + (size_t)w.c * d.P);
    float s = 0.f, q = 0.f;
    sweep<VEC>(d, w, [&](size_t o) {
        float v[VEC];
        Pack<T, VEC>::load(x + o, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float t = v[e] - K; s += t; q = fmaf(t, t, q); }
    });
    block_pair_sum(s, q, part + ((size_t)w.c * d.G + w.g) * 2, red);
}

// The statistics pass + its finisher in ONE launch (rk_bn_stats_finish_*: the -aq blocks' bn1 / bn2, 98 calls per Large-AQ
// step): the producers publish their partial pair as granules (rk_dma.hpp) and C more blocks at the end of the grid -- one
// wave each -- sum a channel's G pairs in fp64 in a fixed order and do k_bn_finish_parts' arithmetic.  No second launch.
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_bn_stats_fused(const T* __restrict__ x, BnDims d, dma::Fin fin,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                           float* __restrict__ ab, float eps, float momentum,
                                                           long long* __restrict__ num_batches_tracked,
                                                           float4* __restrict__ abmi = nullptr) {
    if ((int)blockIdx.x >= fin.producers) {
        if (threadIdx.x >= kWave) return;
        const int c = (int)blockIdx.x - fin.producers;
        if (num_batches_tracked && c == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
        double S[2];
        const bool ok = dma::fin_collect<2>(fin, c, d.G, S);
        if (threadIdx.x != 0) return;
        const float nanv = __uint_as_float(0x7fc00000u);
        const double M = (double)d.F * d.P;
        const double ms = S[0] / M;
        double var = S[1] / M - ms * ms;
        var = var < 0 ? 0 : var;
        const float mean = ok ? (float)((double)ld(x + (size_t)c * d.P) + ms) : nanv;
        const float invstd = ok ? 1.0f / sqrtf((float)var + eps) : nanv;
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        const float a = gamma[c] * invstd;                              // (affine() below, declared after this kernel)
        const float bb = fmaf(-mean, a, beta[c]);
        ab[c] = a;
        ab[d.C + c] = bb;
        if (abmi) abmi[c] = make_float4(a, bb, mean, invstd);           // (the packed record the fused shift backwards read)
        if (running_mean) {
            const float unbiased = (float)(var * (M / (M > 1 ? M - 1 : 1)));
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
        return;
    }
    __shared__ float red[2][kBlock / kWave];
    const Where w = where_am_i(d);
    const float K = ld(x + (size_t)w.c * d.P);
    float s = 0.f, q = 0.f;
    sweep<VEC>(d, w, [&](size_t o) {                                 // (unroll 8 instead of 4 measured worse: 24 -> 28 us at 56x56)
        float v[VEC];
        Pack<T, VEC>::load(x + o, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float t = v[e] - K; s += t; q = fmaf(t, t, q); }
    });
    s = group_sum(s, kBlock, red[0]);
    q = group_sum(q, kBlock, red[1]);
    if (threadIdx.x == 0) {
        const size_t at = (size_t)w.c * 2 * d.G + w.g;
        dma::fin_publish(fin, at, s);
        dma::fin_publish(fin, at + d.G, q);
    }
}

// a, b of y = a x + b, identical in forward and backward (the ReLU mask depends on it)
__device__ __forceinline__ void affine(float gamma, float beta, float mean, float invstd, float& a, float& b) {
    a = gamma * invstd;
    b = fmaf(-mean, a, beta);
}

template <typename T, int VEC, bool RELU, bool TRAIN>
__global__ __launch_bounds__(kBlock) void k_bn_apply(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ part,
                                                     float* __restrict__ running_mean, float* __restrict__ running_var,
                                                     float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                     T* __restrict__ y, BnDims d, float eps, float momentum,
                                                     long long* __restrict__ num_batches_tracked) {
    __shared__ double sm[1][2];
    const Where w = where_am_i(d);
    // nn.BatchNorm2d's `num_batches_tracked += 1` rides on this launch (one thread), instead of a kernel of its own
    if (TRAIN && num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    float mean, invstd;
    if (TRAIN) {
        double S[2];
        channel_sums(part, w.c, d.G, S, sm);
        const double M = (double)d.F * d.P;
        const double ms = S[0] / M;
        double var = S[1] / M - ms * ms;
        var = var < 0 ? 0 : var;
        mean = (float)((double)ld(x + (size_t)w.c * d.P) + ms);
        invstd = 1.0f / sqrtf((float)var + eps);
        if (w.g == 0 && threadIdx.x == 0) {
            save_mean[w.c] = mean;
            save_invstd[w.c] = invstd;
            if (running_mean) {                                       // torch: unbiased variance in the running estimate
                const float unbiased = (float)(var * (M / (M > 1 ? M - 1 : 1)));
                running_mean[w.c] = (1.f - momentum) * running_mean[w.c] + momentum * mean;
                running_var[w.c] = (1.f - momentum) * running_var[w.c] + momentum * unbiased;
            }
        }
    } else {
        mean = running_mean[w.c];
        invstd = 1.0f / sqrtf(running_var[w.c] + eps);
    }
    float a, b;
    affine(gamma[w.c], beta[w.c], mean, invstd, a, b);
    sweep<VEC>(d, w, [&](size_t o) {
        float v[VEC];
        Pack<T, VEC>::load(x + o, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float t = fmaf(a, v[e], b);
            v[e] = RELU ? fmaxf(t, 0.f) : t;
        }
        Pack<T, VEC>::store(y + o, v);
    });
}

template <typename T, int VEC, bool RELU>
__global__ __launch_bounds__(kBlock) void k_bn_bwd_reduce(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd,
                                                          float* __restrict__ part, BnDims d) {
    __shared__ float red[2][kBlock / kWave];
    const Where w = where_am_i(d);
    const float mean = save_mean[w.c], invstd = save_invstd[w.c];
    float a, b;
    affine(gamma[w.c], beta[w.c], mean, invstd, a, b);
    float s1 = 0.f, s2 = 0.f;
    sweep<VEC>(d, w, [&](size_t o) {
        float xv[VEC], gv[VEC];
        Pack<T, VEC>::load(x + o, xv);
        Pack<T, VEC>::load(dy + o, gv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float dz = (RELU && fmaf(a, xv[e], b) <= 0.f) ? 0.f : gv[e];
            s1 += dz;
            s2 = fmaf(dz, (xv[e] - mean) * invstd, s2);
        }
    });
    block_pair_sum(s1, s2, part + ((size_t)w.c * d.G + w.g) * 2, red);
}

template <typename T, int VEC, bool RELU>
__global__ __launch_bounds__(kBlock) void k_bn_bwd_dx(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_invstd,
                                                      const float* __restrict__ part, const T* __restrict__ skip,
                                                      T* __restrict__ dx, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, BnDims d) {
    __shared__ double sm[1][2];
    const Where w = where_am_i(d);
    double S[2];
    channel_sums(part, w.c, d.G, S, sm);
    if (w.g == 0 && threadIdx.x == 0) {
        dbeta[w.c] = (float)S[0];
        dgamma[w.c] = (float)S[1];
    }
    const double M = (double)d.F * d.P;
    const float k1 = (float)(S[0] / M), k2 = (float)(S[1] / M);
    const float mean = save_mean[w.c], invstd = save_invstd[w.c];
    float a, b;
    affine(gamma[w.c], beta[w.c], mean, invstd, a, b);
    sweep<VEC>(d, w, [&](size_t o) {
        float xv[VEC], gv[VEC];
        Pack<T, VEC>::load(x + o, xv);
        Pack<T, VEC>::load(dy + o, gv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float dz = (RELU && fmaf(a, xv[e], b) <= 0.f) ? 0.f : gv[e];
            const float xh = (xv[e] - mean) * invstd;
            gv[e] = a * (dz - k1 - xh * k2);
        }
        if (skip) {                        // x also feeds the block's identity shortcut: add that branch's gradient here
            float sv[VEC];
            Pack<T, VEC>::load(skip + o, sv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) gv[e] += sv[e];
        }
        Pack<T, VEC>::store(dx + o, gv);
    });
}

// The statistics half of the training forward on its own (for a consumer that normalises in its operand load:
// rk_tshift3_bn_forward): k_bn_stats' partials -> save_mean / save_invstd / the affine map ab[2][C] and nn.BatchNorm2d's
// running-statistics bookkeeping, exactly k_bn_apply's arithmetic.  One wave per channel.
template <typename T>
__global__ __launch_bounds__(kWave) void k_bn_finish_parts(const T* __restrict__ x, const float* __restrict__ part,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                           float* __restrict__ ab, BnDims d, float eps, float momentum,
                                                           long long* __restrict__ num_batches_tracked,
                                                           float4* __restrict__ abmi = nullptr) {
    __shared__ double sm[1][2];
    const int c = blockIdx.x;
    if (num_batches_tracked && c == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    double S[2];
    channel_sums(part, c, d.G, S, sm);
    if (threadIdx.x != 0) return;
    const double M = (double)d.F * d.P;
    const double ms = S[0] / M;
    double var = S[1] / M - ms * ms;
    var = var < 0 ? 0 : var;
    const float mean = (float)((double)ld(x + (size_t)c * d.P) + ms);
    const float invstd = 1.0f / sqrtf((float)var + eps);
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    float a, b;
    affine(gamma[c], beta[c], mean, invstd, a, b);
    ab[c] = a;
    ab[d.C + c] = b;
    if (abmi) abmi[c] = make_float4(a, b, mean, invstd);
    if (running_mean) {
        const float unbiased = (float)(var * (M / (M > 1 ? M - 1 : 1)));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// ---------------------------------------------------------------------------------------------
// Training-mode fusion with the 1x1 GEMMs (rk_pw.hip, PwTrain): the statistics pass and the backward's reduction pass
// are done by the GEMM that produces the tensor, one partial per (channel, 128-column wave tile); the kernels below
// finish them.  One workgroup per channel, fp64 combination in a fixed order (deterministic).
//
// forward: stats[c][j] = (pivot_j, sum(y - pivot_j), sum((y - pivot_j)^2), -) over the n_j columns of tile j
// (n_j travels in the fourth component: 128 columns for rk_pw.hip's GEMM, 64 for rk_pw2.hip's, a column split of a
// workgroup for rk_pw3.hip's; the tiles of a channel add up to `count`).  With K = pivot_0:  sum(y - K) = s_j + n_j (p_j - K),
// sum((y - K)^2) = q_j + 2 (p_j - K) s_j + n_j (p_j - K)^2 -- then exactly k_bn_apply's arithmetic: mean, biased variance,
// invstd, the affine map (a, b) of y = a x + b, and nn.BatchNorm2d's running-statistics bookkeeping.
constexpr int kFin = 1024;               // threads of a finisher block: one block per channel, J tiles to sweep
__global__ __launch_bounds__(kFin) void k_bn_finish_tiles(const float4* __restrict__ stats, int J, long long count,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var,
                                                            float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                            float* __restrict__ a_out, float* __restrict__ b_out,
                                                            float4* __restrict__ pack, float eps,
                                                            float momentum, long long* __restrict__ num_batches_tracked) {
    __shared__ double red[3][kFin / kWave];
    const int c = blockIdx.x;
    const float4* p = stats + (size_t)c * J;
    const double K = (double)p[0].x;
    double s1 = 0, s2 = 0, sn = 0;
    for (int j = threadIdx.x; j < J; j += (int)blockDim.x) {
        const float4 t = p[j];
        const double n = (double)t.w;                            // columns of tile j: every producer writes it
        const double dp = (double)t.x - K;
        s1 += (double)t.y + n * dp;
        s2 += (double)t.z + 2.0 * dp * (double)t.y + n * dp * dp;
        sn += n;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    sn = wave_sum(sn);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s1; red[1][wave] = s2; red[2][wave] = sn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S1 = 0, S2 = 0, SN = 0;
        for (int w = 0; w < (int)blockDim.x / kWave; ++w) { S1 += red[0][w]; S2 += red[1][w]; SN += red[2][w]; }
        const double M = (double)count;
        // the tiles must add up to `count` columns: a record a producer left unwritten (torch.empty garbage in .w) or a
        // producer / consumer disagreement about the tile count yields NaN statistics -- loud -- instead of silently wrong ones
        if (SN != M) S1 = __longlong_as_double(0x7ff8000000000000ll);
        const double ms = S1 / M;
        double var = S2 / M - ms * ms;
        var = var < 0 ? 0 : var;
        const float mean = (float)(K + ms);
        const float invstd = 1.0f / sqrtf((float)var + eps);
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        float a, b;
        affine(gamma[c], beta[c], mean, invstd, a, b);
        a_out[c] = a;
        b_out[c] = b;
        if (pack) pack[c] = make_float4(a, b, mean, invstd);
        if (running_mean) {
            const float unbiased = (float)(var * (M / (M > 1 ? M - 1 : 1)));
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    }
}

// backward: bred[c][j] = (sum dz, sum dz xhat) of tile j -> the two per-channel constants of BatchNorm's d(x),
// k1 = sum(dz) / M, k2 = sum(dz xhat) / M, and d(beta) = sum(dz), d(gamma) = sum(dz xhat).
__global__ __launch_bounds__(kFin) void k_bn_bwd_finish_tiles(const float2* __restrict__ bred, int J, long long count,
                                                              float* __restrict__ k12, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int C) {
    __shared__ double red[2][kFin / kWave];
    const int c = blockIdx.x;
    const float2* p = bred + (size_t)c * J;
    double s1 = 0, s2 = 0;
    for (int j = threadIdx.x; j < J; j += (int)blockDim.x) {
        const float2 t = p[j];
        s1 += (double)t.x;
        s2 += (double)t.y;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s1; red[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double S1 = 0, S2 = 0;
        for (int w = 0; w < (int)blockDim.x / kWave; ++w) { S1 += red[0][w]; S2 += red[1][w]; }
        dbeta[c] = (float)S1;
        dgamma[c] = (float)S2;
        k12[c] = (float)(S1 / (double)count);
        k12[C + c] = (float)(S2 / (double)count);
    }
}

// y = relu?(a[c] x + b[c]) with a GIVEN affine map (the training forward after k_bn_finish_tiles)
template <typename T, int VEC, bool RELU>
__global__ __launch_bounds__(kBlock) void k_bn_apply_affine(const T* __restrict__ x, const float* __restrict__ av,
                                                            const float* __restrict__ bv, T* __restrict__ y, BnDims d) {
    const Where w = where_am_i(d);
    const float a = av[w.c], b = bv[w.c];
    sweep<VEC>(d, w, [&](size_t o) {
        float v[VEC];
        Pack<T, VEC>::load(x + o, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float t = fmaf(a, v[e], b);
            v[e] = RELU ? fmaxf(t, 0.f) : t;
        }
        Pack<T, VEC>::store(y + o, v);
    });
}

// dx = a (dz - k1 - xhat k2) (+ skip) with dz ALREADY masked by the ReLU and k1, k2 given (k_bn_bwd_finish_tiles)
template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_bn_bwd_dx_pre(const T* __restrict__ dz, const T* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, const float* __restrict__ k12,
                                                          const T* __restrict__ skip, T* __restrict__ dx, BnDims d) {
    const Where w = where_am_i(d);
    const float mean = save_mean[w.c], invstd = save_invstd[w.c];
    const float a = gamma[w.c] * invstd;
    const float k1 = k12[w.c], k2 = k12[d.C + w.c];
    sweep<VEC>(d, w, [&](size_t o) {
        float xv[VEC], gv[VEC];
        Pack<T, VEC>::load(x + o, xv);
        Pack<T, VEC>::load(dz + o, gv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xh = (xv[e] - mean) * invstd;
            gv[e] = a * (gv[e] - k1 - xh * k2);
        }
        if (skip) {
            float sv[VEC];
            Pack<T, VEC>::load(skip + o, sv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) gv[e] += sv[e];
        }
        Pack<T, VEC>::store(dx + o, gv);
    });
}

// the tile statistics of a tensor nobody's epilogue produced them for (the first block after an unfused stem, tests):
// same float4 format, one wave per (channel, group of tiles)
__global__ __launch_bounds__(kBlock) void k_bn_tile_stats(const float* __restrict__ x, float4* __restrict__ stats, int C,
                                                          int P, int J, long long count) {
    const long long wave = (long long)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= (long long)C * J) return;
    const int c = (int)(wave % C);
    const long long j = wave / C;
    // tile j = columns [128 j, 128 j + 128) of the flattened (f, p) index; a lane takes 2 consecutive columns
    const long long n0 = 128 * j + 2 * lane;
    float v0 = 0.f, v1 = 0.f;
    const bool on = n0 < count;                              // (count % 4 == 0: both columns or neither)
    if (on) {
        const long long f = n0 / P;
        const int p = (int)(n0 - f * P);
        const float2 t = *reinterpret_cast<const float2*>(x + ((size_t)f * C + c) * P + p);
        v0 = t.x; v1 = t.y;
    }
    const float piv = __shfl(v0, 0);
    float s1 = 0.f, s2 = 0.f;
    if (on) {
        float t = v0 - piv; s1 += t; s2 = fmaf(t, t, s2);
        t = v1 - piv; s1 += t; s2 = fmaf(t, t, s2);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const long long nl = count - 128 * j;
    if (lane == 0) stats[(size_t)c * J + j] = make_float4(piv, s1, s2, (float)(nl < 128 ? nl : 128));
}

// ---------------------------------------------------------------------------------------------
int make_bn(BnDims& d, int F, int C, int P) {
    if (F <= 0 || C <= 0 || P <= 0) return RK_ERR_BAD_DIMS;
    if ((long long)F * C * P > 0x7fffffffLL * 4) return RK_ERR_BAD_DIMS;
    constexpr int chunk = 8192;          // elements per workgroup (4096 .. 16384 measured within 2 %)
    d.F = F; d.C = C; d.P = P;
    int fb = (chunk + P - 1) / P;
    fb = fb < 1 ? 1 : (fb > F ? F : fb);
    d.FB = fb;
    d.G = (F + fb - 1) / fb;
    return RK_OK;
}
unsigned grid_bn(const BnDims& d) { return (unsigned)((long long)d.C * d.G); }
// (16 bytes per partial: the fused statistics kernel hands its pairs over as granules, rk_dma.hpp)
size_t ws_bn(const BnDims& d) { return (size_t)d.C * d.G * 2 * 16; }
template <typename T> bool vec4_ok(const BnDims& d, const void* a, const void* b, const void* c = nullptr) {
    const uintptr_t m = 4 * sizeof(T) - 1;
    return d.P % 4 == 0 && !((uintptr_t)a & m) && !((uintptr_t)b & m) && !((uintptr_t)c & m);
}

template <typename T>
int bn_forward(const void* x_, const float* gamma, const float* beta, float* running_mean, float* running_var,
               float* save_mean, float* save_invstd, void* y_, int F, int C, int P, float eps, float momentum,
               int relu, int training, void* ws, size_t ws_bytes, rk_stream_t stream_, long long* nbt) {
    const T* x = (const T*)x_; T* y = (T*)y_;
    if (!x || !y || !gamma || !beta) return RK_ERR_NULL_POINTER;
    if (training ? (!save_mean || !save_invstd) : (!running_mean || !running_var)) return RK_ERR_NULL_POINTER;
    if (training && ((running_mean == nullptr) != (running_var == nullptr))) return RK_ERR_NULL_POINTER;
    BnDims d;
    if (int rc = make_bn(d, F, C, P)) return rc;
    if (training && (!ws || ws_bytes < ws_bn(d))) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid(grid_bn(d)), block(kBlock);
    float* part = (float*)ws;
    const bool v4 = vec4_ok<T>(d, x, y);
#define RK_BN_APPLY(VEC, RELU, TRAIN)                                                                              \
    hipLaunchKernelGGL((k_bn_apply<T, VEC, RELU, TRAIN>), grid, block, 0, stream, x, gamma, beta, (const float*)part, \
                       running_mean, running_var, save_mean, save_invstd, y, d, eps, momentum, nbt)
    if (training) {
        if (v4) hipLaunchKernelGGL((k_bn_stats<T, 4>), grid, block, 0, stream, x, part, d);
        else hipLaunchKernelGGL((k_bn_stats<T, 1>), grid, block, 0, stream, x, part, d);
        if (v4) { if (relu) RK_BN_APPLY(4, true, true); else RK_BN_APPLY(4, false, true); }
        else { if (relu) RK_BN_APPLY(1, true, true); else RK_BN_APPLY(1, false, true); }
    } else {
        if (v4) { if (relu) RK_BN_APPLY(4, true, false); else RK_BN_APPLY(4, false, false); }
        else { if (relu) RK_BN_APPLY(1, true, false); else RK_BN_APPLY(1, false, false); }
    }
#undef RK_BN_APPLY
    return launch_status();
}

template <typename T>
int bn_backward(const void* dy_, const void* x_, const float* gamma, const float* beta, const float* save_mean,
                const float* save_invstd, const void* skip_, void* dx_, float* dgamma, float* dbeta, int F, int C, int P,
                int relu, void* ws, size_t ws_bytes, rk_stream_t stream_) {
    const T* dy = (const T*)dy_; const T* x = (const T*)x_; T* dx = (T*)dx_; const T* skip = (const T*)skip_;
    if (!dy || !x || !dx || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta)
        return RK_ERR_NULL_POINTER;
    BnDims d;
    if (int rc = make_bn(d, F, C, P)) return rc;
    if (!ws || ws_bytes < ws_bn(d)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid(grid_bn(d)), block(kBlock);
    float* part = (float*)ws;
    const bool v4 = vec4_ok<T>(d, x, dy, dx) && !((uintptr_t)skip & (4 * sizeof(T) - 1));
#define RK_BN_BWD(VEC, RELU)                                                                                       \
    do {                                                                                                           \
        hipLaunchKernelGGL((k_bn_bwd_reduce<T, VEC, RELU>), grid, block, 0, stream, dy, x, gamma, beta, save_mean,  \
                           save_invstd, part, d);                                                                  \
        hipLaunchKernelGGL((k_bn_bwd_dx<T, VEC, RELU>), grid, block, 0, stream, dy, x, gamma, beta, save_mean,      \
                           save_invstd, (const float*)part, skip, dx, dgamma, dbeta, d);                           \
    } while (0)
    if (v4) { if (relu) RK_BN_BWD(4, true); else RK_BN_BWD(4, false); }
    else { if (relu) RK_BN_BWD(1, true); else RK_BN_BWD(1, false); }
#undef RK_BN_BWD
    return launch_status();
}

inline bool stats_fused_on() {                              // RK_BN_STATS_FUSED=0: statistics kernel + finisher kernel
    static const bool on = [] { const char* e = getenv("RK_BN_STATS_FUSED"); return !(e && e[0] == '0'); }();
    return on;
}
template <typename T>
int bn_stats_finish(const T* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float* save_mean, float* save_invstd, float* ab, int F, int C, int P, float eps, float momentum,
                    long long* nbt, void* ws, size_t ws_bytes, rk_stream_t stream_, float4* abmi = nullptr) {
    if (!x || !gamma || !beta || !save_mean || !save_invstd || !ab) return RK_ERR_NULL_POINTER;
    if ((uintptr_t)abmi & 15) return RK_ERR_BAD_DIMS;
    if ((running_mean == nullptr) != (running_var == nullptr)) return RK_ERR_NULL_POINTER;
    BnDims d;
    if (int rc = make_bn(d, F, C, P)) return rc;
    if (!ws || ws_bytes < ws_bn(d)) return RK_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    if (stats_fused_on() && !((uintptr_t)ws & 15)) {
        dma::Fin fin;
        fin.gran = reinterpret_cast<unsigned long long*>(ws);
        dma::fin_arm(fin);
        fin.producers = (int)grid_bn(d);
        const dim3 grid(grid_bn(d) + C), block(kBlock);
#define RK_BN_SF(VEC) hipLaunchKernelGGL((k_bn_stats_fused<T, VEC>), grid, block, 0, stream, x, d, fin, gamma, beta, running_mean, \
                                         running_var, save_mean, save_invstd, ab, eps, momentum, nbt, abmi)
        if constexpr (std::is_same<T, __hip_bfloat16>::value) {
            if (P % 8 == 0 && !((uintptr_t)x & 15)) { RK_BN_SF(8); return launch_status(); }
        }
        if (vec4_ok<T>(d, x, x)) RK_BN_SF(4);
        else RK_BN_SF(1);
#undef RK_BN_SF
        return launch_status();
    }
    float* part = (float*)ws;
    if (vec4_ok<T>(d, x, x)) hipLaunchKernelGGL((k_bn_stats<T, 4>), dim3(grid_bn(d)), dim3(kBlock), 0, stream, x, part, d);
    else hipLaunchKernelGGL((k_bn_stats<T, 1>), dim3(grid_bn(d)), dim3(kBlock), 0, stream, x, part, d);
    hipLaunchKernelGGL((k_bn_finish_parts<T>), dim3(C), dim3(kWave), 0, stream, x, (const float*)part, gamma, beta, running_mean,
                       running_var, save_mean, save_invstd, ab, d, eps, momentum, nbt, abmi);
    return launch_status();
}
// The same map as a FLAT sweep of memory.  The per-channel form above gives a workgroup 8 192 elements of ONE channel, i.e.
// (at 14 x 14) 42 pieces of 784 bytes 225 KB apart -- tools/stream_pattern_probe.hip: pieces below ~6 KB collapse the
// achieved bandwidth -- although nothing here needs a channel to itself: the map is elementwise with per-channel
// constants.  Here a workgroup owns 8 192 CONTIGUOUS elements (32 KB of every tensor; consecutive lanes, consecutive
// 16-byte cells), and the constants of the <= 8 192 / P + 2 planes it touches sit in LDS, indexed by plane.
// Same expression per element: bit-identical to k_bn_bwd_dx_pre.
// Chunk size and cache policy, measured (round 4; [256,288,14,14] / [256,72,56,56] fp32, us, plain | + skip): 8192 elements,
// default policy 35.5 | 48.1 / 123.6 | 166.2; non-temporal store 33.9 | 46.1 / 119.1 | 160.1; + non-temporal dz / skip loads
// 30.0 | 40.7 / 114.1 | 155.6; + non-temporal x loads 32.0 | 44.2 / 113.0 | 154.2; 4096 elements with all three 28.7 | 39.0 /
// 112.1 | 156.2; 16384 elements 41.6 / 129.5.  Every operand is touched once by one workgroup and the result is far larger
// than L2; in the train steps all-non-temporal measured best (Large 54.0 -> 53.5 ms, Tiny 19.75 -> 19.6, Small 34.4 -> 34.0).
#ifndef RK_BNF_NT
#define RK_BNF_NT 7
#endif
constexpr int kFlatElems = 4096;
typedef float bn_f32x4 __attribute__((ext_vector_type(4)));
template <typename T, bool NT> __device__ __forceinline__ void flat_load(const T* p, float (&v)[4]) {
    if constexpr (NT && std::is_same<T, float>::value) {
        const bn_f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const bn_f32x4*>(p));
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else Pack<T, 4>::load(p, v);
}
template <typename T, bool NT> __device__ __forceinline__ void flat_store(T* p, const float (&v)[4]) {
    if constexpr (NT && std::is_same<T, float>::value) {
        bn_f32x4 q = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<bn_f32x4*>(p));
    } else Pack<T, 4>::store(p, v);
}
constexpr int kFlatPlanes = 2048;          // planes of P >= 4 elements a chunk can touch, + 1
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bn_bwd_dx_pre_flat(const T* __restrict__ dz, const T* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                               const float* __restrict__ save_invstd, const float* __restrict__ k12,
                                                               const T* __restrict__ skip, T* __restrict__ dx, int C, int P,
                                                               long long total) {
    __shared__ float4 coef[kFlatPlanes + 2];               // (mean, invstd, a, k1) ... and k2 in a second table
    __shared__ float coef2[kFlatPlanes + 2];
    const long long e0 = (long long)blockIdx.x * kFlatElems;
    const long long plane0 = e0 / P;                        // first (frame, channel) plane of the chunk
    long long eend = e0 + kFlatElems;
    eend = eend < total ? eend : total;
    const int nplanes = (int)((eend - 1) / P - plane0) + 1;
    for (int i = threadIdx.x; i < nplanes; i += kBlock) {
        const int c = (int)((plane0 + i) % C);
        const float invstd = save_invstd[c];
        coef[i] = make_float4(save_mean[c], invstd, gamma[c] * invstd, k12[c]);
        coef2[i] = k12[C + c];
    }
    __syncthreads();
    const int r0 = (int)(e0 - plane0 * P);                  // offset of the chunk inside its first plane
#pragma unroll 4
    for (int u = 0; u < kFlatElems / (4 * kBlock); ++u) {
        const int rel = 4 * ((int)threadIdx.x + kBlock * u);
        const long long e = e0 + rel;
        if (e >= eend) break;
        const int li = (r0 + rel) / P;                      // plane of this cell (P % 4 == 0: a cell never straddles planes)
        const float4 cf = coef[li];
        const float k2 = coef2[li];
        float xv[4], gv[4];
        flat_load<T, (RK_BNF_NT & 2) != 0>(x + e, xv);
        flat_load<T, (RK_BNF_NT & 1) != 0>(dz + e, gv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xh = (xv[q] - cf.x) * cf.y;
            gv[q] = cf.z * (gv[q] - cf.w - xh * k2);
        }
        if (skip) {
            float sv[4];
            flat_load<T, (RK_BNF_NT & 1) != 0>(skip + e, sv);
#pragma unroll
            for (int q = 0; q < 4; ++q) gv[q] += sv[q];
        }
        flat_store<T, (RK_BNF_NT & 4) != 0>(dx + e, gv);
    }
}

// 16-bit storage: the same sweep with 16-BYTE cells (8 elements per lane; the 4-element form above moves 8 bytes per lane and
// access, and measured ~2/3 of the per-channel k_bn_bwd_dx's rate on the same tensors -- it is the largest kernel of the bf16 -aq
// train step).  P % 4 == 0, so a cell's two halves may sit in two planes (14 x 14: 196 = 4 * 49): each half looks its plane up.
typedef unsigned bn_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kFlat16Elems = 8192;
inline bool flat16_on() {                                   // RK_BN_FLAT16=0: the 4-element sweep for 16-bit storage too
    static const bool on = [] { const char* e = getenv("RK_BN_FLAT16"); return !(e && e[0] == '0'); }();
    return on;
}
__device__ __forceinline__ void flat16_unpack(const bn_u32x4& r, float (&v)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(r[q] << 16); v[2 * q + 1] = __uint_as_float(r[q] & 0xffff0000u); }
}
template <bool NT> __device__ __forceinline__ bn_u32x4 flat16_load(const void* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const bn_u32x4*>(p));
    else return *reinterpret_cast<const bn_u32x4*>(p);
}
__global__ __launch_bounds__(kBlock) void k_bn_bwd_dx_pre_flat16(const __hip_bfloat16* __restrict__ dz,
                                                                 const __hip_bfloat16* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                                 const float* __restrict__ save_invstd, const float* __restrict__ k12,
                                                                 const __hip_bfloat16* __restrict__ skip, __hip_bfloat16* __restrict__ dx,
                                                                 int C, int P, long long total) {
    __shared__ float4 coef[kFlatPlanes + 2];               // (mean, invstd, a, k1) ... and k2 in a second table
    __shared__ float coef2[kFlatPlanes + 2];
    using B = Pack<__hip_bfloat16, 4>;
    const long long e0 = (long long)blockIdx.x * kFlat16Elems;
    const long long plane0 = e0 / P;
    long long eend = e0 + kFlat16Elems;
    eend = eend < total ? eend : total;
    const int nplanes = (int)((eend - 1) / P - plane0) + 1;
    for (int i = threadIdx.x; i < nplanes; i += kBlock) {
        const int c = (int)((plane0 + i) % C);
        const float invstd = save_invstd[c];
        coef[i] = make_float4(save_mean[c], invstd, gamma[c] * invstd, k12[c]);
        coef2[i] = k12[C + c];
    }
    __syncthreads();
    const int r0 = (int)(e0 - plane0 * P);
#pragma unroll 4
    for (int u = 0; u < kFlat16Elems / (8 * kBlock); ++u) {
        const int rel = 8 * ((int)threadIdx.x + kBlock * u);
        const long long e = e0 + rel;
        if (e >= eend) break;                               // (total % 8 == 0: a cell is whole)
        const int la = (r0 + rel) / P, lb = (r0 + rel + 4) / P;
        float xv[8], gv[8];
        flat16_unpack(flat16_load<(RK_BNF_NT & 2) != 0>(x + e), xv);
        flat16_unpack(flat16_load<(RK_BNF_NT & 1) != 0>(dz + e), gv);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 cf = coef[h ? lb : la];
            const float k2 = coef2[h ? lb : la];
#pragma unroll
            for (int q = 4 * h; q < 4 * h + 4; ++q) {
                const float xh = (xv[q] - cf.x) * cf.y;
                gv[q] = cf.z * (gv[q] - cf.w - xh * k2);
            }
        }
        if (skip) {
            float sv[8];
            flat16_unpack(flat16_load<(RK_BNF_NT & 1) != 0>(skip + e), sv);
#pragma unroll
            for (int q = 0; q < 8; ++q) gv[q] += sv[q];
        }
        bn_u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = B::bits(gv[2 * q]) | (B::bits(gv[2 * q + 1]) << 16);
        if constexpr ((RK_BNF_NT & 4) != 0) __builtin_nontemporal_store(o, reinterpret_cast<bn_u32x4*>(dx + e));
        else *reinterpret_cast<bn_u32x4*>(dx + e) = o;
    }
}

template <typename T>
int bn_bwd_dx_pre(const T* dz, const T* x, const float* gamma, const float* save_mean, const float* save_invstd,
                  const float* k12, const T* skip, T* dx, int F, int C, int P, rk_stream_t stream) {
    if (!dz || !x || !gamma || !save_mean || !save_invstd || !k12 || !dx) return RK_ERR_NULL_POINTER;
    BnDims d;
    if (int rc = make_bn(d, F, C, P)) return rc;
    const dim3 grid(grid_bn(d)), block(kBlock);
    const bool v4 = vec4_ok<T>(d, x, dz, dx) && !((uintptr_t)skip & (4 * sizeof(T) - 1));
    const long long total = (long long)F * C * P;
    if constexpr (std::is_same<T, __hip_bfloat16>::value) {
        if (v4 && total % 8 == 0 && kFlat16Elems / P + 2 <= kFlatPlanes &&
            !(((uintptr_t)dz | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)skip) & 15) && flat16_on()) {
            hipLaunchKernelGGL(k_bn_bwd_dx_pre_flat16, dim3((unsigned)((total + kFlat16Elems - 1) / kFlat16Elems)), block, 0,
                               (hipStream_t)stream, dz, x, gamma, save_mean, save_invstd, k12, skip, dx, C, P, total);
            return launch_status();
        }
    }
    if (v4 && kFlatElems / P + 2 <= kFlatPlanes) {
        hipLaunchKernelGGL((k_bn_bwd_dx_pre_flat<T>), dim3((unsigned)((total + kFlatElems - 1) / kFlatElems)), block, 0,
                           (hipStream_t)stream, dz, x, gamma, save_mean, save_invstd, k12, skip, dx, C, P, total);
        return launch_status();
    }
    if (v4) hipLaunchKernelGGL((k_bn_bwd_dx_pre<T, 4>), grid, block, 0, (hipStream_t)stream, dz, x, gamma, save_mean,
                               save_invstd, k12, skip, dx, d);
    else hipLaunchKernelGGL((k_bn_bwd_dx_pre<T, 1>), grid, block, 0, (hipStream_t)stream, dz, x, gamma, save_mean,
                            save_invstd, k12, skip, dx, d);
    return launch_status();
}

// xs[f][c][ho][wo] = relu(a[c] x[f][c][2 ho][2 wo] + b[c]) rounded to the storage type: the activation of a downsampling block
// at the pixels its stride-2 projecting shortcut reads (fused_bn.bn_relu_tshift_fork: the full-size activation is never stored)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bn_relu_gather2(const T* __restrict__ x, const float* __restrict__ ab, T* __restrict__ xs,
                                                            long long total, int C, int H, int W) {
    const long long o = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (o >= total) return;
    const int Wo = W / 2, Ho = H / 2;
    const int wo = (int)(o % Wo);
    const long long r = o / Wo;
    const int ho = (int)(r % Ho);
    const long long p = r / Ho;                                      // (f, c) plane
    const int c = (int)(p % C);
    const float v = fmaf(ab[c], ld(x + (p * H + 2 * ho) * W + 2 * wo), ab[C + c]);
    st(xs + o, fmaxf(v, 0.f));
}

// the same for 16-bit storage and W % 8 == 0: a thread reads 8 consecutive elements of an even row (16 bytes) and writes the 4
// even ones (8 bytes) -- the element-at-a-time form above read 2 bytes per lane: 58 us per call on the four layers that use it
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bn_relu_gather2_x4(const T* __restrict__ x, const float* __restrict__ ab, T* __restrict__ xs,
                                                               long long cells, int C, int H, int W) {
    const long long o = (long long)blockIdx.x * kBlock + threadIdx.x;       // cell = 4 consecutive outputs of one output row
    if (o >= cells) return;
    const int cpr = W / 8, Ho = H / 2;
    const int j = (int)(o % cpr);
    const long long r = o / cpr;
    const int ho = (int)(r % Ho);
    const long long p = r / Ho;
    const int c = (int)(p % C);
    const float a = ab[c], b = ab[C + c];
    const uint4 v = *reinterpret_cast<const uint4*>(x + (p * H + 2 * ho) * W + 8 * j);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    T out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T e;
        const unsigned short bits = (unsigned short)(w[k] & 0xffffu);       // the even element of the pair
        __builtin_memcpy(&e, &bits, 2);
        st(&out[k], fmaxf(fmaf(a, ld(&e), b), 0.f));
    }
    uint2 q;
    __builtin_memcpy(&q, out, 8);
    *reinterpret_cast<uint2*>(xs + (p * Ho + ho) * (W / 2) + 4 * j) = q;
}

}  // namespace bn
}  // namespace rk

using namespace rk;
using namespace rk::bn;

template <typename T>
static int bn_apply_affine(const T* x, const float* a, const float* b, T* y, int F, int C, int P, int relu, rk_stream_t stream) {
    if (!x || !a || !b || !y) return RK_ERR_NULL_POINTER;
    BnDims d;
    if (int rc = make_bn(d, F, C, P)) return rc;
    const dim3 grid(grid_bn(d)), block(kBlock);
    const bool v4 = vec4_ok<T>(d, x, y);
#define RK_AA(VEC, RELU) hipLaunchKernelGGL((k_bn_apply_affine<T, VEC, RELU>), grid, block, 0, (hipStream_t)stream, x, a, b, y, d)
    if (v4) { if (relu) RK_AA(4, true); else RK_AA(4, false); }
    else { if (relu) RK_AA(1, true); else RK_AA(1, false); }
#undef RK_AA
    return launch_status();
}

template <typename T>
static int bn_relu_gather2(const void* x, const float* ab, void* xs, int F, int C, int H, int W, rk_stream_t stream) {
    if (!x || !ab || !xs) return RK_ERR_NULL_POINTER;
    if (F <= 0 || C <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2) return RK_ERR_BAD_DIMS;
    const long long total = (long long)F * C * (H / 2) * (W / 2);
    if constexpr (sizeof(T) == 2) {
        if (W % 8 == 0 && !((uintptr_t)x & 15) && !((uintptr_t)xs & 7)) {
            const long long cells = total / 4, cblocks = (cells + kBlock - 1) / kBlock;
            if (cblocks > 0x7fffffffLL) return RK_ERR_BAD_DIMS;
            hipLaunchKernelGGL((k_bn_relu_gather2_x4<T>), dim3((unsigned)cblocks), dim3(kBlock), 0, (hipStream_t)stream, (const T*)x, ab,
                               (T*)xs, cells, C, H, W);
            return launch_status();
        }
    }
    const long long blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffLL) return RK_ERR_BAD_DIMS;
    hipLaunchKernelGGL((k_bn_relu_gather2<T>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, (const T*)x, ab, (T*)xs,
                       total, C, H, W);
    return launch_status();
}

extern "C" {

// xs [F, C, H/2, W/2] = relu(a x + b) at the even pixels of x [F, C, H, W]; ab = [2][C] (rk_bn_stats_finish_*)
int rk_bn_relu_gather2_f32(const float* x, const float* ab, float* xs, int F, int C, int H, int W, rk_stream_t stream) {
    return bn_relu_gather2<float>(x, ab, xs, F, C, H, W, stream);
}
int rk_bn_relu_gather2_bf16(const void* x, const float* ab, void* xs, int F, int C, int H, int W, rk_stream_t stream) {
    return bn_relu_gather2<__hip_bfloat16>(x, ab, xs, F, C, H, W, stream);
}

size_t rk_bn_workspace_bytes(int F, int C, int P) {
    BnDims d;
    return make_bn(d, F, C, P) ? 0 : ws_bn(d);
}

// ---- training-mode fusion with the GEMM epilogues (k_bn_finish_tiles & co. above) ----
// stats: float4 [C][tiles] from rk_pw_gemm_stats_f32 / rk_stem_conv3x3s2_stats_f32 / rk_bn_tile_stats_f32 over `count` =
// F * P elements per channel.  Writes save_mean / save_invstd / the affine map (a, b) [C] and updates the running
// statistics (NULL: not tracked) and *num_batches_tracked (NULL: not counted), as nn.BatchNorm2d's training forward.
int rk_bn_finish_tiles_f32(const void* stats, int tiles, long long count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* a,
                           float* b, float* abmi, int C, float eps, float momentum, long long* num_batches_tracked,
                           rk_stream_t stream) {
    if (!stats || !gamma || !beta || !save_mean || !save_invstd || !a || !b) return RK_ERR_NULL_POINTER;
    if ((running_mean == nullptr) != (running_var == nullptr)) return RK_ERR_NULL_POINTER;
    if (C <= 0 || tiles <= 0) return RK_ERR_BAD_DIMS;
    if (count <= 0) return RK_ERR_BAD_DIMS;
    if ((uintptr_t)abmi & 15) return RK_ERR_BAD_DIMS;
    hipLaunchKernelGGL(k_bn_finish_tiles, dim3(C), dim3(kFin), 0, (hipStream_t)stream, (const float4*)stats, tiles, count,
                       gamma, beta, running_mean, running_var, save_mean, save_invstd, a, b, (float4*)abmi, eps, momentum,
                       num_batches_tracked);
    return launch_status();
}
// x [F, C, P] fp32, P % 2 == 0 and F * P % 4 == 0 -> stats float4 [C][rk_pw_tiles(F, P)]
int rk_bn_tile_stats_f32(const float* x, void* stats, int F, int C, int P, rk_stream_t stream) {
    if (!x || !stats) return RK_ERR_NULL_POINTER;
    if (F <= 0 || C <= 0 || P <= 0 || P % 2 || ((long long)F * P) % 4 || ((uintptr_t)x & 7)) return RK_ERR_BAD_DIMS;
    const long long count = (long long)F * P;
    const int J = (int)((count + 127) / 128);
    const long long waves = (long long)C * J;
    hipLaunchKernelGGL(k_bn_tile_stats, dim3((unsigned)((waves + 3) / 4)), dim3(kBlock), 0, (hipStream_t)stream, x,
                       (float4*)stats, C, P, J, count);
    return launch_status();
}
// y = relu?(a[c] x + b[c])
int rk_bn_apply_affine_f32(const float* x, const float* a, const float* b, float* y, int F, int C, int P, int relu,
                           rk_stream_t stream) {
    return bn_apply_affine<float>(x, a, b, y, F, C, P, relu, stream);
}
int rk_bn_apply_affine_bf16(const void* x, const float* a, const float* b, void* y, int F, int C, int P, int relu,
                            rk_stream_t stream) {
    return bn_apply_affine<__hip_bfloat16>((const __hip_bfloat16*)x, a, b, (__hip_bfloat16*)y, F, C, P, relu, stream);
}
// bred: float2 [C][tiles] from rk_pw_gemm_bnbwd_f32 (or the shift backward) -> k12 [2][C] = (sum dz / count,
// sum dz xhat / count), d(gamma), d(beta)
int rk_bn_bwd_finish_tiles_f32(const void* bred, int tiles, long long count, float* k12, float* dgamma, float* dbeta,
                               int C, rk_stream_t stream) {
    if (!bred || !k12 || !dgamma || !dbeta) return RK_ERR_NULL_POINTER;
    if (C <= 0 || tiles <= 0 || count <= 0) return RK_ERR_BAD_DIMS;
    // (few tiles -- the per-clip partials of rk_tshift3_bn_backward: a single wave per channel)
    hipLaunchKernelGGL(k_bn_bwd_finish_tiles, dim3(C), dim3(tiles <= 256 ? kWave : kFin), 0, (hipStream_t)stream,
                       (const float2*)bred, tiles, count, k12, dgamma, dbeta, C);
    return launch_status();
}
// dx = gamma invstd (dz - k1 - xhat k2) (+ skip): dz already ReLU-masked, k12 from rk_bn_bwd_finish_tiles_f32
int rk_bn_bwd_dx_pre_f32(const float* dz, const float* x, const float* gamma, const float* save_mean,
                         const float* save_invstd, const float* k12, const float* skip, float* dx, int F, int C, int P,
                         rk_stream_t stream) {
    return bn_bwd_dx_pre<float>(dz, x, gamma, save_mean, save_invstd, k12, skip, dx, F, C, P, stream);
}
int rk_bn_bwd_dx_pre_bf16(const void* dz, const void* x, const float* gamma, const float* save_mean,
                          const float* save_invstd, const float* k12, const void* skip, void* dx, int F, int C, int P,
                          rk_stream_t stream) {
    return bn_bwd_dx_pre<__hip_bfloat16>((const __hip_bfloat16*)dz, (const __hip_bfloat16*)x, gamma, save_mean, save_invstd, k12,
                                         (const __hip_bfloat16*)skip, (__hip_bfloat16*)dx, F, C, P, stream);
}
// the statistics half of the training forward (k_bn_stats + k_bn_finish_parts): save_mean / save_invstd / ab [2][C] (y = a x +
// b) + the running statistics and *num_batches_tracked as nn.BatchNorm2d's forward; ws of rk_bn_workspace_bytes() bytes
int rk_bn_stats_finish_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float* save_mean, float* save_invstd, float* ab, int F, int C, int P, float eps, float momentum,
                           long long* num_batches_tracked, void* ws, size_t ws_bytes, rk_stream_t stream) {
    return bn_stats_finish<float>(x, gamma, beta, running_mean, running_var, save_mean, save_invstd, ab, F, C, P, eps, momentum,
                                  num_batches_tracked, ws, ws_bytes, stream);
}
int rk_bn_stats_finish_bf16(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float* save_mean, float* save_invstd, float* ab, int F, int C, int P, float eps, float momentum,
                            long long* num_batches_tracked, void* ws, size_t ws_bytes, rk_stream_t stream) {
    return bn_stats_finish<__hip_bfloat16>((const __hip_bfloat16*)x, gamma, beta, running_mean, running_var, save_mean,
                                           save_invstd, ab, F, C, P, eps, momentum, num_batches_tracked, ws, ws_bytes, stream);
}
// ... also leaving the packed record abmi [C][4] = (a, b, mean, invstd) that rk2d_backward_bn_* reads (16-byte aligned)
int rk_bn_stats_finish_abmi_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                float* save_mean, float* save_invstd, float* ab, float* abmi, int F, int C, int P, float eps,
                                float momentum, long long* num_batches_tracked, void* ws, size_t ws_bytes, rk_stream_t stream) {
    if (!abmi) return RK_ERR_NULL_POINTER;
    return bn_stats_finish<float>(x, gamma, beta, running_mean, running_var, save_mean, save_invstd, ab, F, C, P, eps, momentum,
                                  num_batches_tracked, ws, ws_bytes, stream, (float4*)abmi);
}
int rk_bn_stats_finish_abmi_bf16(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float* save_mean, float* save_invstd, float* ab, float* abmi, int F, int C, int P, float eps,
                                 float momentum, long long* num_batches_tracked, void* ws, size_t ws_bytes, rk_stream_t stream) {
    if (!abmi) return RK_ERR_NULL_POINTER;
    return bn_stats_finish<__hip_bfloat16>((const __hip_bfloat16*)x, gamma, beta, running_mean, running_var, save_mean,
                                           save_invstd, ab, F, C, P, eps, momentum, num_batches_tracked, ws, ws_bytes, stream,
                                           (float4*)abmi);
}

#define RK_DEF_BN(SFX, TYPE, CTYPE)                                                                               \
    int rk_bn_relu_forward_##SFX(const CTYPE* x, const float* gamma, const float* beta, float* running_mean,      \
                                 float* running_var, float* save_mean, float* save_invstd, CTYPE* y, int F,       \
                                 int C, int P, float eps, float momentum, int relu, int training, void* ws,      \
                                 size_t ws_bytes, rk_stream_t stream) {                                           \
        return bn_forward<TYPE>(x, gamma, beta, running_mean, running_var, save_mean, save_invstd, y, F, C, P,    \
                                eps, momentum, relu, training, ws, ws_bytes, stream, nullptr);                    \
    }                                                                                                             \
    int rk_bn_relu_forward_counted_##SFX(const CTYPE* x, const float* gamma, const float* beta,                   \
                                         float* running_mean, float* running_var, float* save_mean,               \
                                         float* save_invstd, CTYPE* y, int F, int C, int P, float eps,            \
                                         float momentum, int relu, long long* num_batches_tracked, void* ws,     \
                                         size_t ws_bytes, rk_stream_t stream) {                                   \
        return bn_forward<TYPE>(x, gamma, beta, running_mean, running_var, save_mean, save_invstd, y, F, C, P,    \
                                eps, momentum, relu, 1, ws, ws_bytes, stream, num_batches_tracked);               \
    }                                                                                                             \
    int rk_bn_relu_backward_##SFX(const CTYPE* dy, const CTYPE* x, const float* gamma, const float* beta,         \
                                  const float* save_mean, const float* save_invstd, const CTYPE* dskip,           \
                                  CTYPE* dx, float* dgamma, float* dbeta, int F, int C, int P, int relu,          \
                                  void* ws, size_t ws_bytes, rk_stream_t stream) {                                \
        return bn_backward<TYPE>(dy, x, gamma, beta, save_mean, save_invstd, dskip, dx, dgamma, dbeta, F, C, P,   \
                                 relu, ws, ws_bytes, stream);                                                     \
    }
RK_DEF_BN(f32, float, float)
RK_DEF_BN(bf16, __hip_bfloat16, void)
#undef RK_DEF_BN

}  // extern "C"
