// rk3d_plane.hpp -- RubiksShift3D forward / d(x)-only "plane group" kernels (fp32, stride 1 / pad 0, W % 4 == 0,
// row bands of >= 6 KB): a workgroup owns SP consecutive output planes of one (n, c, row band) instead of walking
// the whole (n, c) column as rk3d_dma.hpp does.
//
// Why (tools/stream_pattern_probe.hip, profiles/r02_pattern_probe.txt): a pure 1R+1W copy of the same 12.5 KB
// planes runs at 6.1 TB/s when every workgroup copies ONE plane and neighbouring workgroups own neighbouring
// planes, 6.0 TB/s with two planes per workgroup, 5.75 with four, and 5.5 TB/s when a workgroup walks the 8
// planes of a column (stride C*H*W) -- whether it keeps one plane in flight or requests all eight up front --
// which is exactly where the column-walking forward sat (5.45 TB/s); row bands below ~6 KB collapse to
// 2.3-3.4 TB/s.  The T blend couples planes t and t+1 of a column, so a group of SP output planes reads SP+1
// source planes; the shared plane is read again by the next group's workgroup, which sits R8 block ids later
// on the SAME XCD by construction (block id % 8 depends on (c, band) only), a few microseconds behind, so that
// read is an L2 hit and HBM sees every byte once -- but it is not free on the CU side (SP = 1, every plane read
// twice, is slower than the column walk).  Measured forward, [32,8,64,56,56]: column walk 79.6 us, SP=2 74.8,
// SP=4 with two 28-row bands 72.8 us; [32,8,54,112,112]: 245.6 -> 231.3 us (SP=2, four 28-row bands);
// 28x28 planes (3 KB) stay on the column walk (36.4 vs 41.2 us).  The same decomposition was built for the
// fused backward and measured 128-196 us against 117 for the column walk (the d(shift) sums need the fields of
// both neighbours of every x plane, so the arithmetic per element grows by 1.5-1.8x and the x planes add LDS);
// it was dropped, the backward stays on rk3d_dma.hpp.
//
// Maths, tap slots, LDS-DMA feed and the bit-exact expression trees are those of rk3d_dma.hpp / rk_dma.hpp.
#pragma once
#include "rk_dma.hpp"

namespace rk {
namespace plane3d {

using namespace dma;

struct PDims {
    BDims b;
    int G;        // plane groups per column: ceil(T / SP)
    int R, R8;    // (c, band) items per (n, group) row; R rounded up to a multiple of 8 (XCD count)
};

struct Item { int n, g, c, band; bool live; };

// block id -> item.  id % 8 (the XCD, observed round-robin) is a function of (c, band) only, so the two
// workgroups that read a source plane (same n, c, band; consecutive groups) share an L2.
__device__ __forceinline__ Item my_item(const PDims& p) {
    Item it;
    const int r = blockIdx.x % p.R8, q = blockIdx.x / p.R8;
    it.live = r < p.R;
    it.c = r / p.b.nbands;
    it.band = r - it.c * p.b.nbands;
    it.g = q % p.G;
    it.n = q / p.G;
    return it;
}

// Forward (NEGATE = false: src = x, dst = y) and d(x) alone (NEGATE = true: src = gy, dst = gx).
template <bool NEGATE, int ROUNDS, int SP, int OFF, bool BN = false>
__device__ __forceinline__ void plane_interp_body(const float* __restrict__ sp, float* __restrict__ dp, float4* slots,
                                                  const BDims& d, const Band& b, const Frac<float>& fT,
                                                  const Frac<float>& fH, const Frac<float>& fW, size_t tstride,
                                                  int to0, float bn_a = 0.f, float bn_b = 0.f) {
    constexpr int NS = SP + 1;
    const int slot_f4 = b.cells_in + 1;
    BCells<ROUNDS> cs;
    make_bcells<ROUNDS>(cs, d, b, (fW.fl - OFF) / 4);
    init_tap_slots<ROUNDS>(slots, NS, slot_f4, b, cs);

    const float rT = fT.r, rH = fH.r, rW = fW.r;
    const float uT = 1 - rT, uH = 1 - rH, uW = 1 - rW;
    const unsigned slots_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr(slots));
    const unsigned slot_bytes = (unsigned)slot_f4 * 16u;
    const float* src0 = sp + (ptrdiff_t)b.src0 * 4;
    float* out0 = dp + (size_t)b.out0 * 4;

    // slot i holds source plane to0 + flT + i; output to0 + i - 1 blends slots i-1 and i
    const int t0 = to0 + fT.fl;
    int issued = 0;
    int mark[NS];
    bool real[NS];                                                  // slot i holds a real plane (BN: to be normalised)
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int t = t0 + i;
        const bool wanted = i == 0 || to0 + i - 1 < d.T;           // ragged last group
        real[i] = wanted && t >= 0 && t < d.T;
        if (wanted && t >= 0 && t < d.T) {
            // first reader of the last slot's plane keeps it in L2 for the next group's workgroup; every other
            // plane is on its last use
            if (i == NS - 1) dma_taps<ROUNDS, false>(src0 + (ptrdiff_t)t * (ptrdiff_t)tstride, slots_addr + i * slot_bytes, cs);
            else dma_taps<ROUNDS, true>(src0 + (ptrdiff_t)t * (ptrdiff_t)tstride, slots_addr + i * slot_bytes, cs);
            issued += cs.n_tap_wave;
        } else {
            zero_taps<ROUNDS>(slots + i * slot_f4, cs);
        }
        mark[i] = issued;
    }

    float4 Bprev[ROUNDS];
    auto round = [&](int i, const float4* cur, float4* out, bool store) {
        const float4 qa0 = lds_b128(cur + cs.a0[i]), qa1 = lds_b128(cur + cs.a1[i]);
        const float4 qb0 = lds_b128(cur + cs.b0[i]), qb1 = lds_b128(cur + cs.b1[i]);
        float q[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            q[m] = uH * (tap<OFF>(qa0, qa1, m) * uW + tap<OFF>(qa0, qa1, m + 1) * rW) +
                   rH * (tap<OFF>(qb0, qb1, m) * uW + tap<OFF>(qb0, qb1, m + 1) * rW);
        if (store) {
            float4 o;
            o.x = uT * Bprev[i].x + rT * q[0];
            o.y = uT * Bprev[i].y + rT * q[1];
            o.z = uT * Bprev[i].z + rT * q[2];
            o.w = uT * Bprev[i].w + rT * q[3];
            stream_store(reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + cs.off0 + 4096 * i), o);
        }
        Bprev[i] = make_float4(q[0], q[1], q[2], q[3]);
    };

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool emit = s >= 1 && to0 + s - 1 < d.T;
        if (s >= 1 && !emit) break;                                 // wave-uniform
        wait_vmcnt(issued - mark[s]);                               // my pieces of slot s have landed
        if (BN && real[s]) bn_taps<ROUNDS>(slots + s * slot_f4, cs, bn_a, bn_b);      // z -> relu(bn(z)), my pieces
        __syncthreads();                                            // everyone's have (and the zero fills)
        const float4* cur = slots + s * slot_f4;
        float4* out = reinterpret_cast<float4*>(out0 + (size_t)(emit ? to0 + s - 1 : 0) * tstride);
#pragma unroll
        for (int i = 0; i + 1 < ROUNDS; ++i) round(i, cur, out, emit);
        if (cs.tail_on) round(ROUNDS - 1, cur, out, emit && cs.tail_live);
        if (emit) issued += cs.n_out_wave;
    }
}

template <bool NEGATE, int ROUNDS, int SP, bool BN = false>
__global__ __launch_bounds__(kBlock) void k3d_plane_interp(const float* __restrict__ src,
                                                           const float* __restrict__ shift,
                                                           float* __restrict__ dst, PDims p,
                                                           const float4* __restrict__ abmi = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float4 slots[];
    const Item it = my_item(p);
    if (!it.live) return;
    const BDims& d = p.b;
    const int c = it.c, n = it.n;
    float sT = shift[c], sH = shift[d.C + c], sW = shift[2 * d.C + c];
    if (NEGATE) { sT = -sT; sH = -sH; sW = -sW; }
    const Frac<float> fT = split_shift(sT), fH = split_shift(sH), fW = split_shift(sW);
    const int HW = d.H * d.W;
    const size_t tstride = (size_t)d.C * HW;
    const float* sp = src + ((size_t)n * d.T * d.C + c) * HW;
    float* dp = dst + ((size_t)n * d.T * d.C + c) * HW;
    const Band b = make_band(d, it.band, fH.fl);
    const int to0 = it.g * SP;
    float bn_a = 0.f, bn_b = 0.f;
    if (BN) { const float4 pk = abmi[c]; bn_a = pk.x; bn_b = pk.y; }

    if (NEGATE && sT == 0 && sH == 0 && sW == 0) {                  // rubiks3d_kernels.cu:819-827: plain copy
        for (int t = to0; t < to0 + SP && t < d.T; ++t)
            for (int cell = threadIdx.x; cell < b.cells_out; cell += kBlock)
                reinterpret_cast<float4*>(dp + (size_t)t * tstride)[b.out0 + cell] =
                    reinterpret_cast<const float4*>(sp + (size_t)t * tstride)[b.out0 + cell];
        return;
    }
    switch (((fW.fl % 4) + 4) % 4) {                                // wave-uniform
        case 0: plane_interp_body<NEGATE, ROUNDS, SP, 0, BN>(sp, dp, slots, d, b, fT, fH, fW, tstride, to0, bn_a, bn_b); break;
        case 1: plane_interp_body<NEGATE, ROUNDS, SP, 1, BN>(sp, dp, slots, d, b, fT, fH, fW, tstride, to0, bn_a, bn_b); break;
        case 2: plane_interp_body<NEGATE, ROUNDS, SP, 2, BN>(sp, dp, slots, d, b, fT, fH, fW, tstride, to0, bn_a, bn_b); break;
        default: plane_interp_body<NEGATE, ROUNDS, SP, 3, BN>(sp, dp, slots, d, b, fT, fH, fW, tstride, to0, bn_a, bn_b); break;
    }
}

// ---------------------------------------------------------------------------------------------
// Host side.
// Geometry: SP = 4 on half-size bands for planes up to 64 wide (56x56: two 28-row bands, 5 slots = 32.5 KB),
// SP = 2 on the default bands for wider ones (112x112: four 28-row bands, 3 slots = 39 KB); see the header.
inline bool make_pdims(PDims& p, int& SP, const Dims3& d) {
    const bool s1p0 = d.sT == 1 && d.sH == 1 && d.sW == 1 && d.pT == 0 && d.pH == 0 && d.pW == 0;
    if (!s1p0 || d.W % 4 != 0 || d.W < 4 || !streaming_kernels_on()) return false;
    BDims& b = p.b;
    b.N = d.N; b.T = d.T; b.C = d.C; b.H = d.H; b.W = d.W; b.W4 = d.W / 4;
    if (!choose_bands(b)) return false;
    SP = 2;
    if (b.W4 <= 16 && d.T >= 4 && b.H % (2 * b.nbands) == 0) {
        BDims h = b;
        h.nbands = 2 * b.nbands; h.BH = b.H / h.nbands;
        if (rounds_for(h.BH * h.W4) == rounds_for((h.BH + 1) * h.W4)) { b = h; SP = 4; }
    }
    if ((size_t)b.BH * b.W4 * 16 < 6 * 1024) return false;           // small bands: the column walk is faster
    p.G = (d.T + SP - 1) / SP;
    p.R = d.C * b.nbands;
    p.R8 = (p.R + 7) / 8 * 8;
    return (long long)d.N * p.G * p.R8 <= 0x7fffffffLL;
}
inline size_t slots_bytes(const BDims& b, int nslots) { return (size_t)nslots * ((b.BH + 1) * b.W4 + 1) * 16; }

template <bool NEGATE, int SP, bool BN = false>
inline void launch_interp_sp(const float* src, const float* shift, float* dst, const PDims& p, hipStream_t stream,
                             const float4* abmi = nullptr) {
    const size_t lds = slots_bytes(p.b, SP + 1);
    const dim3 grid((unsigned)(p.b.N * p.G * p.R8)), block(kBlock);
    switch (rounds_of(p.b)) {
        case 1: hipLaunchKernelGGL((k3d_plane_interp<NEGATE, 1, SP, BN>), grid, block, lds, stream, src, shift, dst, p, abmi); break;
        case 2: hipLaunchKernelGGL((k3d_plane_interp<NEGATE, 2, SP, BN>), grid, block, lds, stream, src, shift, dst, p, abmi); break;
        case 3: hipLaunchKernelGGL((k3d_plane_interp<NEGATE, 3, SP, BN>), grid, block, lds, stream, src, shift, dst, p, abmi); break;
        default: hipLaunchKernelGGL((k3d_plane_interp<NEGATE, 4, SP, BN>), grid, block, lds, stream, src, shift, dst, p, abmi); break;
    }
}

// forward / d(x)-only; false = not handled here (rk3d_dma.hpp takes it)
template <bool NEGATE>
inline bool launch_interp(const float* src, const float* shift, float* dst, const Dims3& d, hipStream_t stream) {
    PDims p;
    int SP;
    if (!make_pdims(p, SP, d) || !aligned16(src) || !aligned16(dst)) return false;
    if (slots_bytes(p.b, SP + 1) > 40 * 1024) return false;          // >= 4 workgroups per CU
    if (SP == 4) launch_interp_sp<NEGATE, 4>(src, shift, dst, p, stream);
    else launch_interp_sp<NEGATE, 2>(src, shift, dst, p, stream);
    return true;
}
// forward of relu(bn(z)) (train_block.py): abmi [C] = (a, b, mean, invstd)
inline bool launch_forward_bn(const float* z, const float* shift, float* y, const float4* abmi, const Dims3& d,
                              hipStream_t stream) {
    PDims p;
    int SP;
    if (!make_pdims(p, SP, d) || !aligned16(z) || !aligned16(y) || !aligned16(abmi)) return false;
    if (slots_bytes(p.b, SP + 1) > 40 * 1024) return false;
    if (SP == 4) launch_interp_sp<false, 4, true>(z, shift, y, p, stream, abmi);
    else launch_interp_sp<false, 2, true>(z, shift, y, p, stream, abmi);
    return true;
}

}  // namespace plane3d
}  // namespace rk
