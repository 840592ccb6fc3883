// rk3d_slab.hip -- RubiksShift3D on SMALL planes (fp32, stride 1 / pad 0, T <= 8, 48 <= H*W <= 256, W <= 15): the 7x7
// layers of every network (layer4: [N,8,432|576,7,7]) and, optionally, the 14x14 ones.  Round 5.
//
// Why another family.  The column kernels (rk3d_column.hpp) gave a 7x7 plane one wave -- 49 of 64 lanes -- and
// recomputed per-lane tap indices, bounds masks and 64-bit addresses at every plane step: 465 (forward) / 719
// (backward) VALU instructions per wave, i.e. the kernels were VALU-ISSUE-bound at algorithmic HBM traffic
// (profiles/r04_7x7_pmc.csv).  And a launch this small (29 MB tensors) is one round of waves: whatever a
// wave does in sequence -- fetch, wait, compute, store -- the whole chip does in lock step.
//
// Scheme.  x is [N,T,C,H,W], so for one (n, t) the C planes form ONE contiguous "slab" of C*H*W floats.  A WAVE owns
// 64 M consecutive elements of the slab (M = 4 forward, 2 backward: M elements per lane, all 64 lanes live whatever
// H*W is) across all T planes; a 256-thread workgroup is 4 such waves (a "chunk" of 256 M elements):
//   * the wave LDS-DMAs its piece of EVERY plane up front (global_load_lds_dwordx4 nt; piece = its 16 M cells of 16
//     bytes + a halo of W+1 floats each side: the taps of an element never leave its own plane, and a plane that
//     straddles the piece's edge continues in the halo) into T private slots, then consumes plane k as soon as IT has
//     landed -- counted s_waitcnt vmcnt(N) literals; VMEM retires in order -- and stores output plane k-1 at once:
//     all reads of the launch are queued from the first microsecond and the writes trickle out under them.  A wave
//     reads only what it fetched itself: NO barrier in the walk.  (First version: one barrier after all T planes, then
//     compute: 17.2 us forward / 41.5 us backward at [32,8,576,7,7]; second: ring of 3 slots, refilled in the walk:
//     16.7 us forward whatever the channel count -- three dependent memory round trips per wave.)
//   * every VMEM instruction of the walk is ours: the per-channel shifts come through SCALAR loads (lgkmcnt) -- a
//     compiler-issued vector load would make hipcc wait with a vmcnt that knows nothing of the DMAs before it and drain
//     them -- and the backward's x (the lane's own M elements of each plane) through inline-asm global loads into
//     registers, issued before the DMAs;
//   * per element the channel, its shift, the four tap addresses (a tap outside the plane points at the slot's zero
//     cell: validity costs nothing per step) and the weights are computed ONCE; the walk is unrolled (T <= 8), so the
//     slot of a step is a compile-time constant: a plane step is 4 M ds_read_b32 at immediate offsets, the
//     reference's expression tree, one store of M floats per lane;
//   * the temporal blend couples planes k+f and k+f+1 with f = floor(shift_T) PER CHANNEL.  Every lane evaluates the
//     field B(k) of the SAME plane k at step k and v(k) = (1-rT) B(k-1) + rT B(k); y[k-1] is v(k) for f = 0 and
//     v(k-1) for f = -1 -- one select -- so the whole piece of output plane k-1 leaves as one aligned store per lane
//     although its cells mix channels (49 % 4 != 0).
// A wave whose elements are all live and on covered channels and whose fetch instructions all have lanes (every wave
// of a launch but the ones at a slab's tail), at T = 8, runs the walk with literal wait counts ("regular"); any
// other wave runs the same code with vmcnt(0) waits.
// Arithmetic: the reference's trees (rubiks3d_kernels.cu:193-203, :914-924), contraction off -> y and d(x)
// bit-identical to the oracle; d(shift) in the adjoint form of rk3d_dma.hpp, per-channel sums through LDS in a fixed
// order, at most TWO partials per (n, channel) -- a plane may straddle two chunks -- published as granule pairs to the
// finalizer blocks of the same launch (rk_dma.hpp) or as plain partials (two-phase ABI): P = 2 N.
// Channels this does not cover -- floor(shift) outside {-1, 0} in any dimension, and (backward) an exactly-integer
// component (the lowered-index quirk :290-298) -- are "slow": their elements are masked out of the streaming pass and
// redone by the per-element helpers of rk3d_generic.hpp over exactly the chunk's share of the plane, so both
// workgroups that touch a straddling plane use the same formulation for it.
#include <type_traits>
#include <utility>

#include "rk3d_slab.hpp"
#include "rk3d_dma.hpp"

namespace rk {
namespace slab3d {

using namespace dma;

constexpr int kMaxT = 8;          // planes resident in LDS
constexpr int kMinHW = 16;        // (smaller planes: the column kernels)

struct SDims {
    int N, T, C, H, W, HW;
    int slab;                     // C * HW: elements of one (n, t)
    int nchunks;                  // ceil(slab / (256 M))
};

// M elements per lane; HALO 16-byte cells each side of a wave's piece
template <int M, int HALO> struct Geo {
    static constexpr int kOwn = kWave * M / 4;             // 16-byte cells a wave owns per plane
    static constexpr int kCells = kOwn + 2 * HALO;         // its piece of a plane
    static constexpr int kZ = kCells * 16;                 // byte offset of a slot's zero cell
    static constexpr int kStride = (kCells + 1) * 16;      // slot stride
    static constexpr int kNF = (kCells + kWave - 1) / kWave;      // DMA wave-instructions per plane
    static constexpr int kChunk = kBlock * M;              // elements per workgroup and plane
};

// (offb is a literal after unrolling: it lands in the instruction's offset field)
__device__ __forceinline__ float lds_at(unsigned a, int offb) {
    return *(__attribute__((address_space(3))) const float*)(size_t)(a + (unsigned)offb);
}
template <typename F, int... K> __device__ __forceinline__ void for_each_step(std::integer_sequence<int, K...>, F&& f) {
    (f(std::integral_constant<int, K>{}), ...);                      // steps 0..kMaxT with the step index a constant expression
}
template <int N> __device__ __forceinline__ void wait_lit() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int M> struct VecOf;
template <> struct VecOf<2> { using type = f32x2; };
template <> struct VecOf<4> { using type = f32x4; };

// what a lane knows about one of its M elements
struct Elem {
    unsigned rel[4];              // LDS byte addresses (slot 0) of the taps (h0,w0) (h0,w0+1) (h0+1,w0) (h0+1,w0+1)
    float rT, rH, rW;
    bool f0;                      // floor(shift_T) == 0 (else -1)
    bool fast;                    // live and covered by the streaming pass
    bool slow;                    // live and left to the helpers
};

// The shift of a lane's element: three inline-asm global loads (ours to wait for), issued AHEAD of the DMAs, so that
// "at most <the DMAs> outstanding" means they have landed.  (A compiler-issued load would make hipcc wait with a vmcnt
// that knows nothing of the DMAs and drain them; scalar loads + a select over the few channels a wave touches were
// turned by hipcc into a table in SCRATCH indexed per element -- VMEM again, same drain.)
__device__ __forceinline__ void load_f1(float& v, const float* sbase_uniform, int voff) {
    asm volatile("global_load_dword %0, %1, %2" : "+v"(v) : "v"(voff), "s"(sbase_uniform) : "memory");
}
template <int M> struct LaneShift { float s[M][3]; };
template <int M>
__device__ __forceinline__ void load_lane_shift(LaneShift<M>& ls, const SDims& d, const float* __restrict__ shift, int e0) {
    const float* sb = uniform_ptr(shift);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int c = min(e0 + m, d.slab - 1) / d.HW;
#pragma unroll
        for (int k = 0; k < 3; ++k) { ls.s[m][k] = 0.f; load_f1(ls.s[m][k], sb, (k * d.C + c) * 4); }
    }
}
// ... and the registers tied to the wait that covers them
template <int M> __device__ __forceinline__ void tie(LaneShift<M>& ls) {
#pragma unroll
    for (int m = 0; m < M; ++m) asm volatile("" : "+v"(ls.s[m][0]), "+v"(ls.s[m][1]), "+v"(ls.s[m][2]));
}

// g0: LDS byte address of the wave's slot 0; e: slab element index; own_off: byte offset of the element inside a slot
template <int M, int HALO, bool NEGATE, bool INT_IS_SLOW>
__device__ __forceinline__ void make_elem(Elem& el, const SDims& d, float s0, float s1, float s2, unsigned g0, int e, unsigned own_off) {
    using G = Geo<M, HALO>;
    const bool live = e < d.slab;
    const int ec = live ? e : 0;
    const int c = ec / d.HW, p = ec - c * d.HW;
    const int h = p / d.W, w = p - h * d.W;
    const Frac<float> fT = split_shift(NEGATE ? -s0 : s0), fH = split_shift(NEGATE ? -s1 : s1),
                      fW = split_shift(NEGATE ? -s2 : s2);
    bool near = (unsigned)(fT.fl + 1) < 2u && (unsigned)(fH.fl + 1) < 2u && (unsigned)(fW.fl + 1) < 2u;
    if (INT_IS_SLOW) near = near && fT.r != 0 && fH.r != 0 && fW.r != 0;
    el.fast = live && near;
    el.slow = live && !near;
    el.f0 = fT.fl == 0;
    el.rT = fT.r; el.rH = fH.r; el.rW = fW.r;
    const int h0 = h + fH.fl, w0 = w + fW.fl;
    const bool mh0 = (unsigned)h0 < (unsigned)d.H, mh1 = (unsigned)(h0 + 1) < (unsigned)d.H;
    const bool mw0 = (unsigned)w0 < (unsigned)d.W, mw1 = (unsigned)(w0 + 1) < (unsigned)d.W;
    const unsigned a = g0 + own_off + (unsigned)((fH.fl * d.W + fW.fl) * 4), Z = g0 + G::kZ;
    el.rel[0] = el.fast && mh0 && mw0 ? a : Z;
    el.rel[1] = el.fast && mh0 && mw1 ? a + 4u : Z;
    el.rel[2] = el.fast && mh1 && mw0 ? a + 4u * d.W : Z;
    el.rel[3] = el.fast && mh1 && mw1 ? a + 4u * d.W + 4u : Z;
}

// A wave's fetch geometry: DMA instruction A = piece cells 0..63, B = the rest (kNF == 2 only)
struct Fetch {
    const float* tap0;            // (uniform) source of piece cell 0 in plane 0 -- may lie before the tensor, never dereferenced there
    unsigned g0;                  // (uniform) LDS byte address of slot 0
    int slab;
    bool actA, actB;              // this lane takes part
    int nF;                       // (uniform) wave-instructions a plane fetch really issues
    int lane;
};
// cell0w: slab cell (16 bytes) of the wave's own cell 0
template <int M, int HALO>
__device__ __forceinline__ void make_fetch(Fetch& f, const SDims& d, int cell0w, int lane) {
    using G = Geo<M, HALO>;
    const int slab_cells = d.slab >> 2;
    const int pa = cell0w - HALO + lane, pb = pa + kWave;             // slab cells of my piece cells (A, B)
    f.actA = lane < G::kCells && pa >= 0 && pa < slab_cells;
    f.actB = kWave + lane < G::kCells && pb >= 0 && pb < slab_cells;
    f.nF = (__ballot(f.actA) != 0ull ? 1 : 0) + (__ballot(f.actB) != 0ull ? 1 : 0);
    f.slab = d.slab;
    f.lane = lane;
}
// plane p -> slot `slot`.  REG: every lane the geometry gives a cell is known to be active
template <bool REG, int M, int HALO>
__device__ __forceinline__ void fetch_plane(const Fetch& f, int p, int slot) {
    using G = Geo<M, HALO>;
    const float* src = uniform_ptr(f.tap0 + (ptrdiff_t)p * f.slab);
    const unsigned dst = f.g0 + (unsigned)(slot * G::kStride);
    if (REG ? (G::kCells >= kWave || f.lane < G::kCells) : f.actA) dma16s<true>(src, f.lane * 16, dst);
    if (G::kNF == 2 && (REG ? kWave + f.lane < G::kCells : f.actB)) dma16s<true>(src, (kWave + f.lane) * 16, dst + kWave * 16);
}

// VMEM instructions a regular wave (T = 8) may leave outstanding when it waits for plane k.  Issue order:
// [x loads, backward] F0 .. F7 | step j: [wait] reads, store(j >= 1)   ->   the fetches of planes k+1.. and the stores so far
// with a ring of RG slots: F0 .. F(RG-1) | step j: [wait] reads, F(j+RG), store(j >= 1)
constexpr int allowed(int k, int nF, int RG, bool stores) {
    int issued = 0, mark[kMaxT] = {};
    for (int p = 0; p < RG && p < kMaxT; ++p) { issued += nF; mark[p] = issued; }
    for (int j = 0; j <= kMaxT; ++j) {
        if (j == k) return k < kMaxT ? issued - mark[k] : 0;
        if (j + RG < kMaxT) { issued += nF; mark[j + RG] = issued; }
        if (stores && j >= 1) issued += 1;
    }
    return 0;
}

// forward of one output plane's sub-range through global memory (slow channels): the generic kernel's loop
__device__ __forceinline__ void forward_plane_range(const float* __restrict__ x, const float* __restrict__ shift,
                                                    float* __restrict__ y, const SDims& d, int n, int to, int c, int e,
                                                    int E, int lo, int hi) {
    const Frac<float> fT = split_shift(shift[c]), fH = split_shift(shift[d.C + c]), fW = split_shift(shift[2 * d.C + c]);
    const size_t tstride = (size_t)d.slab;
    const float* xc = x + ((size_t)n * d.T * d.C + c) * d.HW;
    float* yp = y + (((size_t)n * d.T + to) * d.C + c) * d.HW;
    const int t0 = to + fT.fl;
    const bool v0 = t0 >= 0 && t0 < d.T, v1 = t0 + 1 >= 0 && t0 + 1 < d.T;
    const float* p0 = xc + (v0 ? (size_t)t0 * tstride : 0);
    const float* p1 = xc + (v1 ? (size_t)(t0 + 1) * tstride : 0);
    for (int i = lo + e; i < hi; i += E) {
        const int ho = i / d.W, wo = i - ho * d.W;
        const int h0 = ho + fH.fl, w0 = wo + fW.fl;
        const bool mh0 = h0 >= 0 && h0 < d.H, mh1 = h0 + 1 >= 0 && h0 + 1 < d.H;
        const bool mw0 = w0 >= 0 && w0 < d.W, mw1 = w0 + 1 >= 0 && w0 + 1 < d.W;
        const int o00 = h0 * d.W + w0;
        float q000 = 0, q001 = 0, q010 = 0, q011 = 0, q100 = 0, q101 = 0, q110 = 0, q111 = 0;
        if (v0) {
            if (mh0 && mw0) q000 = p0[o00];
            if (mh0 && mw1) q001 = p0[o00 + 1];
            if (mh1 && mw0) q010 = p0[o00 + d.W];
            if (mh1 && mw1) q011 = p0[o00 + d.W + 1];
        }
        if (v1) {
            if (mh0 && mw0) q100 = p1[o00];
            if (mh0 && mw1) q101 = p1[o00 + 1];
            if (mh1 && mw0) q110 = p1[o00 + d.W];
            if (mh1 && mw1) q111 = p1[o00 + d.W + 1];
        }
        yp[i] = trilerp(q000, q001, q010, q011, q100, q101, q110, q111, fT.r, fH.r, fW.r);
    }
}

// the channels a chunk touches and, per channel, its share [lo, hi) of the plane (plane-relative element indices)
struct ChRange { int cA, cB; };
__device__ __forceinline__ ChRange chunk_channels(const SDims& d, int chunk, int CH) {
    const int e0 = chunk * CH, e1 = min(d.slab, e0 + CH) - 1;
    return ChRange{e0 / d.HW, e1 / d.HW};
}
__device__ __forceinline__ void channel_share(const SDims& d, int chunk, int CH, int c, int& lo, int& hi) {
    const int e0 = chunk * CH, e1 = min(d.slab, e0 + CH);
    lo = max(c * d.HW, e0) - c * d.HW;
    hi = min((c + 1) * d.HW, e1) - c * d.HW;
}
template <bool NEGATE, bool INT_IS_SLOW>
__device__ __forceinline__ bool channel_is_slow(const SDims& d, const float* __restrict__ shift, int c) {
    const float s0 = shift[c], s1 = shift[d.C + c], s2 = shift[2 * d.C + c];
    const Frac<float> fT = split_shift(NEGATE ? -s0 : s0), fH = split_shift(NEGATE ? -s1 : s1),
                      fW = split_shift(NEGATE ? -s2 : s2);
    bool near = (unsigned)(fT.fl + 1) < 2u && (unsigned)(fH.fl + 1) < 2u && (unsigned)(fW.fl + 1) < 2u;
    if (INT_IS_SLOW) near = near && fT.r != 0 && fH.r != 0 && fW.r != 0;
    return !near;
}


// ---------------------------------------------------------------------------------------------
// Forward (NEGATE = false: src = x, dst = y) and d(x) alone (NEGATE = true: src = gy, dst = gx).
// (dst is NOT __restrict__: a store through a noalias pointer may legally be moved across the counted-wait asm statements,
// whose "memory" clobber only orders accesses the asm could make -- hipcc did sink the stores of steps 4..7 below the
// waits of steps 5..7, which both breaks the literal counts and keeps 5 planes of results in registers)
template <bool NEGATE, int HALO, int RG>
__global__ __launch_bounds__(kBlock) void k3d_slab_interp(const float* __restrict__ src, const float* __restrict__ shift,
                                                          float* dst, SDims d, Dims3 gd) {
    constexpr int M = 4;
    using G = Geo<M, HALO>;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = (int)threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = (int)(blockIdx.x % (unsigned)d.nchunks), n = (int)(blockIdx.x / (unsigned)d.nchunks);
    const int cell0w = chunk * (G::kChunk / 4) + wave * G::kOwn;     // slab cell of the wave's cell 0
    const size_t nbase = (size_t)n * d.T * d.slab;
    const int e0 = 4 * cell0w + M * lane;                            // my first element
    LaneShift<M> ls;
    load_lane_shift<M>(ls, d, shift, e0);
    Fetch f;
    make_fetch<M, HALO>(f, d, cell0w, lane);
    f.g0 = __builtin_amdgcn_readfirstlane(lds_byte_addr(lds_raw)) + (unsigned)(wave * (RG * G::kStride));
    f.tap0 = src + nbase + (ptrdiff_t)(cell0w - HALO) * 4;
#pragma unroll
    for (int p = 0; p < RG; ++p)
        if (p < d.T) fetch_plane<false, M, HALO>(f, p, p);
    if (lane < RG) *reinterpret_cast<float4*>(lds_raw + wave * (RG * G::kStride) + lane * G::kStride + G::kZ) = make_float4(0.f, 0.f, 0.f, 0.f);

    wait_vmcnt(f.nF * min(RG, d.T));                                 // the shift loads are older than the fetches
    tie<M>(ls);
    Elem el[M];
#pragma unroll
    for (int m = 0; m < M; ++m)
        make_elem<M, HALO, NEGATE, false>(el[m], d, ls.s[m][0], ls.s[m][1], ls.s[m][2], f.g0, e0 + m, (unsigned)((HALO * 4 + M * lane + m) * 4));
    const bool cell_live = e0 < d.slab;
    bool all_fast = true, thread_slow = false;
#pragma unroll
    for (int m = 0; m < M; ++m) { all_fast = all_fast && el[m].fast; thread_slow = thread_slow || el[m].slow; }
    const bool regular = d.T == kMaxT && f.nF == G::kNF && __ballot(all_fast) == ~0ull;      // wave-uniform

    float* optr = dst + nbase + (size_t)e0;                          // my elements of output plane 0
    auto walk = [&](auto REGC) {
        constexpr bool REG = decltype(REGC)::value;
        float Bprev[M], vprev[M];
#pragma unroll
        for (int m = 0; m < M; ++m) { Bprev[m] = 0.f; vprev[m] = 0.f; }
        for_each_step(std::make_integer_sequence<int, kMaxT + 1>{}, [&](auto KC) {
            constexpr int k = decltype(KC)::value;
            if (!REG && k > d.T) return;
            float B[M];
#pragma unroll
            for (int m = 0; m < M; ++m) B[m] = 0.f;
            if (k < kMaxT && (REG || k < d.T)) {
                if (REG) wait_lit<allowed(k < kMaxT ? k : 0, G::kNF, RG, true)>(); else wait_vmcnt(0);
                float q[M][4];
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[m][j] = lds_at(el[m].rel[j], (k % RG) * G::kStride);
                if (k + RG < kMaxT && (REG || k + RG < d.T)) fetch_plane<REG, M, HALO>(f, k + RG, k % RG);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float uW = 1 - el[m].rW, uH = 1 - el[m].rH;
                    B[m] = uH * (q[m][0] * uW + q[m][1] * el[m].rW) + el[m].rH * (q[m][2] * uW + q[m][3] * el[m].rW);
                }
            }
            float o[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = (1 - el[m].rT) * Bprev[m] + el[m].rT * B[m];
                o[m] = el[m].f0 ? v : vprev[m];
                vprev[m] = v; Bprev[m] = B[m];
            }
            if (k >= 1) {
                float* out = optr + (size_t)(k - 1) * d.slab;
                const f32x4 t = {o[0], o[1], o[2], o[3]};
                if (REG) {
                    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(out));
                } else if (cell_live) {
                    if (all_fast) __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(out));
                    else {
#pragma unroll
                        for (int m = 0; m < M; ++m)
                            if (el[m].fast) out[m] = o[m];
                    }
                }
            }
        });
    };
    if (regular) walk(std::true_type{}); else walk(std::false_type{});

    if (__syncthreads_or((int)thread_slow)) {                        // workgroup-uniform from here on
        const ChRange cr = chunk_channels(d, chunk, G::kChunk);
        for (int c = cr.cA; c <= cr.cB; ++c) {
            if (!channel_is_slow<NEGATE, false>(d, shift, c)) continue;
            int lo, hi;
            channel_share(d, chunk, G::kChunk, c, lo, hi);
            for (int t = 0; t < d.T; ++t) {
                if (NEGATE) backward_input_plane<float, false>(shift, src, dst, gd, n, t, c, tid, kBlock, lo, hi);
                else forward_plane_range(src, shift, dst, d, n, t, c, tid, kBlock, lo, hi);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward: d(x) (WRITE_GX) + the d(shift) partials, one pass over (gy, x).  Partials: part[c][3][P], P = 2 N,
// index j N + n with j = 0 for the chunk holding the plane's first element, 1 for the next one (absent -- zero in the
// two-phase form -- when the plane does not straddle).  FUSED: granule pairs + finalizer blocks (row-sum + K5 inside the launch).
__device__ __forceinline__ void load_x2(f32x2& v, const float* sbase_uniform, int voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "+v"(v) : "v"(voff), "s"(sbase_uniform) : "memory");
}
template <bool WRITE_GX, bool FUSED, int HALO, int RG>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5))) void k3d_slab_backward(const float* __restrict__ x, const float* __restrict__ shift,
                                                            const float* __restrict__ gy, float* gx,
                                                            float* __restrict__ part, SDims d, Dims3 gd, dma3d::Fin3 fin) {
    constexpr int M = 2;
    using G = Geo<M, HALO>;
    const int P = 2 * d.N;
    if (FUSED && (int)blockIdx.x >= fin.f.producers) {
        if (threadIdx.x < kWave) {                                   // a plane inside one chunk has N partials, a straddling one 2 N
            const int c = (int)blockIdx.x - fin.f.producers;
            const bool two = (c * d.HW) / G::kChunk != ((c + 1) * d.HW - 1) / G::kChunk;
            dma3d::finalizer_wave<3>(fin, c, d.C, P, dma3d::BnFuse{}, two ? P : d.N);
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    __shared__ float red[3][kBlock / kWave];
    const int tid = (int)threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = (int)(blockIdx.x % (unsigned)d.nchunks), n = (int)(blockIdx.x / (unsigned)d.nchunks);
    const int cell0w = chunk * (G::kChunk / 4) + wave * G::kOwn;
    const size_t nbase = (size_t)n * d.T * d.slab;
    const int e0 = 4 * cell0w + M * lane;                            // my first element
    const bool cell_live = e0 < d.slab;

    LaneShift<M> ls;
    load_lane_shift<M>(ls, d, shift, e0);
    // my M elements of every x plane: asm loads (ours to wait for), ahead of the DMAs -- they are complete when plane 0 of gy is
    f32x2 xq[kMaxT];
    {
        const float* xw = x + nbase + (size_t)cell0w * 4;             // the wave's own cell 0 in plane 0
#pragma unroll
        for (int t = 0; t < kMaxT; ++t) {
            xq[t] = f32x2{0.f, 0.f};
            if (t < d.T && cell_live) load_x2(xq[t], uniform_ptr(xw + (size_t)t * d.slab), lane * (M * 4));
        }
    }
    Fetch f;
    make_fetch<M, HALO>(f, d, cell0w, lane);
    f.g0 = __builtin_amdgcn_readfirstlane(lds_byte_addr(lds_raw)) + (unsigned)(wave * (RG * G::kStride));
    f.tap0 = gy + nbase + (ptrdiff_t)(cell0w - HALO) * 4;
#pragma unroll
    for (int p = 0; p < RG; ++p)
        if (p < d.T) fetch_plane<false, M, HALO>(f, p, p);
    if (lane < RG) *reinterpret_cast<float4*>(lds_raw + wave * (RG * G::kStride) + lane * G::kStride + G::kZ) = make_float4(0.f, 0.f, 0.f, 0.f);

    wait_vmcnt(f.nF * min(RG, d.T));                                 // the shift (and x) loads are older than the fetches
    tie<M>(ls);
    Elem el[M];
#pragma unroll
    for (int m = 0; m < M; ++m)
        make_elem<M, HALO, true, true>(el[m], d, ls.s[m][0], ls.s[m][1], ls.s[m][2], f.g0, e0 + m, (unsigned)((HALO * 4 + M * lane + m) * 4));
    bool all_fast = true, thread_slow = false;
#pragma unroll
    for (int m = 0; m < M; ++m) { all_fast = all_fast && el[m].fast; thread_slow = thread_slow || el[m].slow; }
    const bool regular = d.T == kMaxT && f.nF == G::kNF && __ballot(all_fast) == ~0ull;      // wave-uniform

    float* optr = WRITE_GX ? gx + nbase + (size_t)e0 : nullptr;
    float aT[M], aH[M], aW[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { aT[m] = 0.f; aH[m] = 0.f; aW[m] = 0.f; }
    // step k: Q = field of gy plane k (zero at k = T).  A lane with f' = 0 pairs it with xb = x[k], xa = x[k-1]; one
    // with f' = -1 with xb = x[k+1], xa = x[k] (rk3d_column.hpp, step s = k - f').  gx[k-1] = v(k) resp. v(k-1).
    auto walk = [&](auto REGC) {
        constexpr bool REG = decltype(REGC)::value;
        float Qprev[M], vprev[M], xa[M];
#pragma unroll
        for (int m = 0; m < M; ++m) { Qprev[m] = 0.f; vprev[m] = 0.f; xa[m] = 0.f; }
        for_each_step(std::make_integer_sequence<int, kMaxT + 1>{}, [&](auto KC) {
            constexpr int k = decltype(KC)::value;
            if (!REG && k > d.T) return;
            if (k == 0) {
                // plane 0 of gy has landed, and with it (VMEM retires in order) every x load: the registers are tied to the wait
                if (REG) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(xq[4]),
                                      "+v"(xq[5]), "+v"(xq[6]), "+v"(xq[7]) : "n"(allowed(0, G::kNF, RG, WRITE_GX)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(xq[4]),
                                  "+v"(xq[5]), "+v"(xq[6]), "+v"(xq[7]) : : "memory");
#pragma unroll
                for (int m = 0; m < M; ++m) xa[m] = (el[m].fast && !el[m].f0) ? xq[0][m] : 0.f;   // f' = -1: step k = -1 (Q = 0) left x[0] here
            } else if (k < kMaxT && (REG || k < d.T)) {
                if (REG) wait_lit<allowed(k < kMaxT ? k : 0, G::kNF, RG, WRITE_GX)>(); else wait_vmcnt(0);
            }
            float q[M][4];
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) q[m][j] = 0.f;
            if (k < kMaxT && (REG || k < d.T)) {
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[m][j] = lds_at(el[m].rel[j], (k % RG) * G::kStride);
                if (k + RG < kMaxT && (REG || k + RG < d.T)) fetch_plane<REG, M, HALO>(f, k + RG, k % RG);
            }
            const f32x2 zero2 = {0.f, 0.f};
            const f32x2 xk = k < kMaxT ? xq[k < kMaxT ? k : 0] : zero2;              // x[k] (planes >= T were never loaded: zero)
            const f32x2 xk1 = k + 1 < kMaxT ? xq[k + 1 < kMaxT ? k + 1 : 0] : zero2;  // x[k+1]
            float o[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float uW = 1 - el[m].rW, uH = 1 - el[m].rH, uT = 1 - el[m].rT;
                const float la = q[m][0] * uW + q[m][1] * el[m].rW, lb = q[m][2] * uW + q[m][3] * el[m].rW;
                const float Q = uH * la + el[m].rH * lb;                             // the reference's tree, contraction off
                const float c0 = fmaf(uH, q[m][0], el[m].rH * q[m][2]), c1 = fmaf(uH, q[m][1], el[m].rH * q[m][3]);
                float xb = el[m].f0 ? xk[m] : xk1[m];
                if (!REG) xb = el[m].fast ? xb : 0.f;
                const float dx = xb - xa[m], mx = fmaf(uT, xb, el[m].rT * xa[m]);
                aT[m] = fmaf(Q, dx, aT[m]);
                aH[m] = fmaf(la - lb, mx, aH[m]);
                aW[m] = fmaf(c0 - c1, mx, aW[m]);
                xa[m] = xb;
                const float v = uT * Qprev[m] + el[m].rT * Q;
                o[m] = el[m].f0 ? v : vprev[m];
                vprev[m] = v; Qprev[m] = Q;
            }
            if (WRITE_GX && k >= 1) {
                float* out = optr + (size_t)(k - 1) * d.slab;
                const f32x2 t = {o[0], o[1]};
                if (REG) {
                    __builtin_nontemporal_store(t, reinterpret_cast<f32x2*>(out));
                } else if (cell_live) {
                    if (all_fast) __builtin_nontemporal_store(t, reinterpret_cast<f32x2*>(out));
                    else {
#pragma unroll
                        for (int m = 0; m < M; ++m)
                            if (el[m].fast) out[m] = o[m];
                    }
                }
            }
        });
    };
    if (regular) walk(std::true_type{}); else walk(std::false_type{});

    // per-channel sums of the chunk: my M x 3 sums -> LDS (behind the rings), one barrier, then ONE THREAD per (channel,
    // component) adds its share in index order and publishes it.  (First version: one wave per item, wave_sum + publish,
    // nine items in sequence per wave: the epilogue cost more than the walk -- 35.6 us with it, 16.8 us without, at
    // [32,8,576,7,7].)
    f32x2* acc = reinterpret_cast<f32x2*>(lds_raw + (kBlock / kWave) * RG * G::kStride);     // [3][256] pairs
    acc[tid] = f32x2{aT[0], aT[1]};
    acc[kBlock + tid] = f32x2{aH[0], aH[1]};
    acc[2 * kBlock + tid] = f32x2{aW[0], aW[1]};
    acc[3 * kBlock + tid] = f32x2{el[0].slow ? 1.f : 0.f, el[1].slow ? 1.f : 0.f};     // (spares the item threads three shift loads)
    const int any_slow = __syncthreads_or((int)thread_slow);
    const float* accf = reinterpret_cast<const float*>(acc);
    const ChRange cr = chunk_channels(d, chunk, G::kChunk);
    const int nch = cr.cB - cr.cA + 1;
    auto publish = [&](int c, int k, float v) {                      // this chunk's partial of (c, component k)
        const int a = (c * d.HW) / G::kChunk, b = ((c + 1) * d.HW - 1) / G::kChunk;
        const size_t at = ((size_t)c * 3 + k) * P + n;               // index j N + n
        const int j = chunk == a ? 0 : 1;
        if (FUSED) {
            fin_publish(fin.f, at + (size_t)j * d.N, v);             // (the finalizer knows which planes straddle)
        } else {
            part[at + (size_t)j * d.N] = v;
            if (a == b) part[at + d.N] = 0.f;
        }
    };
    // 16 lanes (one DPP row) per item: each lane takes every 16th element of the item's share -- up to 16 independent LDS
    // reads --, the row is summed by DPP shifts, its last lane publishes.  (One thread per item: a serial loop of up to 98
    // dependent LDS round trips, 5.7 us of the 14x14 kernel.)
    for (int base = 0; base < 3 * nch; base += kBlock / 16) {
        const int item = base + (tid >> 4), sub = tid & 15;
        const bool valid = item < 3 * nch;
        const int it = valid ? item : 0;
        const int k = it / nch, c = cr.cA + it - k * nch;
        int lo, hi;
        channel_share(d, chunk, G::kChunk, c, lo, hi);
        const float* a0 = accf + k * G::kChunk + (c * d.HW - chunk * G::kChunk);     // the plane's element 0
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int i = lo + sub + 16 * j; v[j] = (valid && i < hi) ? a0[i] : 0.f; }
        float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        s += ((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15]));
        s += dpp_or_zero<0x111, 0xf>(s);
        s += dpp_or_zero<0x112, 0xf>(s);
        s += dpp_or_zero<0x114, 0xf>(s);
        s += dpp_or_zero<0x118, 0xf>(s);                             // lane 15 of the row: the item's sum
        const bool slow_ch = a0[(3 - k) * G::kChunk + lo] != 0.f;    // the helpers below publish a slow channel
        if (valid && sub == 15 && !slow_ch) publish(c, k, s);
    }
    if (any_slow) {
        for (int c = cr.cA; c <= cr.cB; ++c) {
            if (!channel_is_slow<true, true>(d, shift, c)) continue;
            int lo, hi;
            channel_share(d, chunk, G::kChunk, c, lo, hi);
            float sT = 0.f, sH = 0.f, sW = 0.f;
            for (int t = 0; t < d.T; ++t) {
                if (WRITE_GX) backward_input_plane<float, false>(shift, gy, gx, gd, n, t, c, tid, kBlock, lo, hi);
                shift_grad_plane<float>(x, shift, gy, gd, n, t, c, tid, kBlock, sT, sH, sW, NoAct(), lo, hi);
            }
            sT = group_sum(sT, kBlock, red[0]);
            sH = group_sum(sH, kBlock, red[1]);
            sW = group_sum(sW, kBlock, red[2]);
            if (tid == 0) { publish(c, 0, sT); publish(c, 1, sH); publish(c, 2, sW); }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stride (1,2,2), pad 0, even H and W <= 56 with Wo % 4 != 0 -- 28 -> 14 and 14 -> 7, the third and fourth down-sampling
// layers of every network -- backward: d(x) + d(shift) in one pass.  (rk3d_stride2.hpp needs a 4-output cell; these
// two layers ran on the column kernels: 1 619 VALU instructions per wave, 143 us at [32,8,288,28,28], 3.6 TB/s.)
// Same scheme as above on the INPUT side: a wave owns 256 consecutive elements of the x slab (one float4 cell per lane:
// x in through inline-asm loads, d(x) out as float4 stores); with both spatial strides 2 an input element has at most
// ONE gy tap (rk3d_column.hpp, SINGLE: the tree collapses exactly to wj (v wk)), and the taps of 256 consecutive input
// elements lie in a short run of the gy slab -- (h/2 - 1) .. (h/2 + 1) rows around them, <= 58 cells of 16 bytes -- that ONE
// DMA instruction per plane brings into the wave's slot.  All T planes up front, counted waits, no barrier in the walk.
struct S2Dims {
    int N, T, C, H, W, HW, Wo, HWo;
    int slab_in, slab_out;        // C * HW, C * HWo
    int nchunks;                  // ceil(slab_in / 1024)
};
struct ElemS2 {
    unsigned rel;                 // LDS byte address (slot 0) of the element's one gy tap, or of the zero cell
    float wj, wk, sj, sk, rT;
    bool f0, fast, slow;
};
constexpr int kS2Chunk = 1024, kS2Cells = 64, kS2Z = kS2Cells * 16, kS2Stride = (kS2Cells + 1) * 16;

__device__ __forceinline__ void load_x4(f32x4& v, const float* sbase_uniform, int voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "+v"(v) : "v"(voff), "s"(sbase_uniform) : "memory");
}
// first output row's flat index in the gy slab for input element e: c HWo + (h / 2) Wo
__device__ __forceinline__ int s2_row0(const S2Dims& d, int e) {
    const int c = e / d.HW, p = e - c * d.HW;
    return c * d.HWo + ((p / d.W) >> 1) * d.Wo;
}
template <bool INT_IS_SLOW>
__device__ __forceinline__ bool s2_channel_is_slow(const S2Dims& d, const float* __restrict__ shift, int c) {
    const Frac<float> fT = split_shift(-shift[c]), fH = split_shift(-shift[d.C + c]), fW = split_shift(-shift[2 * d.C + c]);
    bool near = (unsigned)(fT.fl + 1) < 2u && (unsigned)(fH.fl + 1) < 2u && (unsigned)(fW.fl + 1) < 2u;
    if (INT_IS_SLOW) near = near && fT.r != 0 && fH.r != 0 && fW.r != 0;
    return !near;
}

template <bool WRITE_GX, bool FUSED>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3))) void k3d_slab_s2_backward(
    const float* __restrict__ x, const float* __restrict__ shift, const float* __restrict__ gy, float* gx,
    float* __restrict__ part, S2Dims d, Dims3 gd, dma3d::Fin3 fin) {
    constexpr int M = 4;
    const int P = 2 * d.N;
    if (FUSED && (int)blockIdx.x >= fin.f.producers) {
        if (threadIdx.x < kWave) {
            const int c = (int)blockIdx.x - fin.f.producers;
            const bool two = (c * d.HW) / kS2Chunk != ((c + 1) * d.HW - 1) / kS2Chunk;
            dma3d::finalizer_wave<3>(fin, c, d.C, P, dma3d::BnFuse{}, two ? P : d.N);
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    __shared__ float red[3][kBlock / kWave];
    const int tid = (int)threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = (int)(blockIdx.x % (unsigned)d.nchunks), n = (int)(blockIdx.x / (unsigned)d.nchunks);
    const int ew0 = chunk * kS2Chunk + wave * (kWave * M);            // the wave's first input element
    const int e0 = ew0 + M * lane;                                   // mine
    const bool cell_live = e0 < d.slab_in;
    const size_t nb_in = (size_t)n * d.T * d.slab_in, nb_out = (size_t)n * d.T * d.slab_out;

    // shifts of my 4 elements, then my cell of every x plane (asm loads: ours to wait for), then the gy pieces
    LaneShift<M> ls;
    {
        const float* sb = uniform_ptr(shift);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int c = min(e0 + m, d.slab_in - 1) / d.HW;
#pragma unroll
            for (int k = 0; k < 3; ++k) { ls.s[m][k] = 0.f; load_f1(ls.s[m][k], sb, (k * d.C + c) * 4); }
        }
    }
    f32x4 xq[kMaxT];
    {
        const float* xw = x + nb_in + (size_t)ew0;
#pragma unroll
        for (int t = 0; t < kMaxT; ++t) {
            xq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < d.T && cell_live) load_x4(xq[t], uniform_ptr(xw + (size_t)t * d.slab_in), lane * 16);
        }
    }
    // the wave's piece of a gy plane: 16-byte cells pc0 .. pc0 + 63 of the gy slab (those inside it)
    const int ew1 = min(ew0 + kWave * M, d.slab_in);
    const int g_lo = ew0 < d.slab_in ? s2_row0(d, ew0) - d.Wo : 0;
    const int pc0 = __builtin_amdgcn_readfirstlane(g_lo >= 0 ? g_lo >> 2 : -((3 - g_lo) >> 2));       // floor(g_lo / 4)
    const bool actA = ew0 < d.slab_in && pc0 + lane >= 0 && pc0 + lane < (d.slab_out >> 2);
    const int nF = __ballot(actA) != 0ull ? 1 : 0;
    const unsigned g0 = __builtin_amdgcn_readfirstlane(lds_byte_addr(lds_raw)) + (unsigned)(wave * (kMaxT * kS2Stride));
    {
        const float* gp = gy + nb_out + (ptrdiff_t)pc0 * 4;           // never dereferenced outside the slab
#pragma unroll
        for (int p = 0; p < kMaxT; ++p)
            if (p < d.T && actA) dma16s<true>(uniform_ptr(gp + (size_t)p * d.slab_out), lane * 16, g0 + (unsigned)(p * kS2Stride));
    }
    if (lane < kMaxT) *reinterpret_cast<float4*>(lds_raw + wave * (kMaxT * kS2Stride) + lane * kS2Stride + kS2Z) = make_float4(0.f, 0.f, 0.f, 0.f);
    (void)ew1;

    wait_vmcnt(nF * min(kMaxT, d.T));                                // the shift and x loads are older than the fetches
    tie<M>(ls);
    ElemS2 el[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int e = e0 + m;
        const bool live = e < d.slab_in;
        const int ec = live ? e : 0;
        const int c = ec / d.HW, p = ec - c * d.HW;
        const int h = p / d.W, w = p - h * d.W;
        const Frac<float> fT = split_shift(-ls.s[m][0]), fH = split_shift(-ls.s[m][1]), fW = split_shift(-ls.s[m][2]);
        const bool near = (unsigned)(fT.fl + 1) < 2u && (unsigned)(fH.fl + 1) < 2u && (unsigned)(fW.fl + 1) < 2u &&
                          fT.r != 0 && fH.r != 0 && fW.r != 0;
        el[m].fast = live && near;
        el[m].slow = live && !near;
        el[m].f0 = fT.fl == 0;
        el[m].rT = fT.r;
        // rubiks3d_kernels.cu:586-589 through rk3d_column.hpp (SINGLE): the one j in {0, 1} with (h + fl'H + j) even, k alike
        const int r0 = unmap(h + fH.fl, 2, d.H >> 1), r1 = unmap(h + fH.fl + 1, 2, d.H >> 1);
        const int c0 = unmap(w + fW.fl, 2, d.Wo), c1 = unmap(w + fW.fl + 1, 2, d.Wo);
        const int r = r0 >= 0 ? r0 : r1, q = c0 >= 0 ? c0 : c1;
        el[m].wj = r0 >= 0 ? 1 - fH.r : fH.r;  el[m].sj = r0 >= 0 ? 1.f : -1.f;
        el[m].wk = c0 >= 0 ? 1 - fW.r : fW.r;  el[m].sk = c0 >= 0 ? 1.f : -1.f;
        const int g = c * d.HWo + r * d.Wo + q - 4 * pc0;            // float index inside the piece
        const bool ok = el[m].fast && r >= 0 && q >= 0 && g >= 0 && g < 4 * kS2Cells;
        el[m].rel = g0 + (ok ? (unsigned)(g * 4) : (unsigned)kS2Z);
    }
    bool all_fast = true, thread_slow = false;
#pragma unroll
    for (int m = 0; m < M; ++m) { all_fast = all_fast && el[m].fast; thread_slow = thread_slow || el[m].slow; }
    const bool regular = d.T == kMaxT && nF == 1 && __ballot(all_fast) == ~0ull;      // wave-uniform

    float* optr = WRITE_GX ? gx + nb_in + (size_t)e0 : nullptr;
    float aT[M], aH[M], aW[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { aT[m] = 0.f; aH[m] = 0.f; aW[m] = 0.f; }
    auto walk = [&](auto REGC) {
        constexpr bool REG = decltype(REGC)::value;
        float Qprev[M], vprev[M], xa[M];
#pragma unroll
        for (int m = 0; m < M; ++m) { Qprev[m] = 0.f; vprev[m] = 0.f; xa[m] = 0.f; }
        for_each_step(std::make_integer_sequence<int, kMaxT + 1>{}, [&](auto KC) {
            constexpr int k = decltype(KC)::value;
            if (!REG && k > d.T) return;
            if (k == 0) {
                if (REG) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(xq[4]),
                                      "+v"(xq[5]), "+v"(xq[6]), "+v"(xq[7]) : "n"(allowed(0, 1, kMaxT, WRITE_GX)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(xq[4]),
                                  "+v"(xq[5]), "+v"(xq[6]), "+v"(xq[7]) : : "memory");
#pragma unroll
                for (int m = 0; m < M; ++m) xa[m] = (el[m].fast && !el[m].f0) ? xq[0][m] : 0.f;
            } else if (k < kMaxT && (REG || k < d.T)) {
                if (REG) wait_lit<allowed(k < kMaxT ? k : 0, 1, kMaxT, WRITE_GX)>(); else wait_vmcnt(0);
            }
            float v[M];
#pragma unroll
            for (int m = 0; m < M; ++m) v[m] = 0.f;
            if (k < kMaxT && (REG || k < d.T)) {
#pragma unroll
                for (int m = 0; m < M; ++m) v[m] = lds_at(el[m].rel, k * kS2Stride);
            }
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 xk = k < kMaxT ? xq[k < kMaxT ? k : 0] : zero4;
            const f32x4 xk1 = k + 1 < kMaxT ? xq[k + 1 < kMaxT ? k + 1 : 0] : zero4;
            float o[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float vk = v[m] * el[m].wk;
                const float Q = el[m].wj * vk;                        // = the reference's tree with three zero taps
                const float QH = el[m].sj * vk, QW = el[m].sk * (el[m].wj * v[m]);
                const float uT = 1 - el[m].rT;
                float xb = el[m].f0 ? xk[m] : xk1[m];
                if (!REG) xb = el[m].fast ? xb : 0.f;
                const float dx = xb - xa[m], mx = fmaf(uT, xb, el[m].rT * xa[m]);
                aT[m] = fmaf(Q, dx, aT[m]);
                aH[m] = fmaf(QH, mx, aH[m]);
                aW[m] = fmaf(QW, mx, aW[m]);
                xa[m] = xb;
                const float vv = uT * Qprev[m] + el[m].rT * Q;
                o[m] = el[m].f0 ? vv : vprev[m];
                vprev[m] = vv; Qprev[m] = Q;
            }
            if (WRITE_GX && k >= 1) {
                float* out = optr + (size_t)(k - 1) * d.slab_in;
                const f32x4 t = {o[0], o[1], o[2], o[3]};
                if (REG) {
                    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(out));
                } else if (cell_live) {
                    if (all_fast) __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(out));
                    else {
#pragma unroll
                        for (int m = 0; m < M; ++m)
                            if (el[m].fast) out[m] = o[m];
                    }
                }
            }
        });
    };
    if (regular) walk(std::true_type{}); else walk(std::false_type{});

    // per-channel sums of the chunk (see k3d_slab_backward): [4][1024] floats behind the slots
    f32x4* acc = reinterpret_cast<f32x4*>(lds_raw + (kBlock / kWave) * kMaxT * kS2Stride);
    acc[tid] = f32x4{aT[0], aT[1], aT[2], aT[3]};
    acc[kBlock + tid] = f32x4{aH[0], aH[1], aH[2], aH[3]};
    acc[2 * kBlock + tid] = f32x4{aW[0], aW[1], aW[2], aW[3]};
    acc[3 * kBlock + tid] = f32x4{el[0].slow ? 1.f : 0.f, el[1].slow ? 1.f : 0.f, el[2].slow ? 1.f : 0.f, el[3].slow ? 1.f : 0.f};
    const int any_slow = __syncthreads_or((int)thread_slow);
    const float* accf = reinterpret_cast<const float*>(acc);
    const int ce0 = chunk * kS2Chunk, ce1 = min(d.slab_in, ce0 + kS2Chunk);
    const int cA = ce0 / d.HW, cB = (ce1 - 1) / d.HW, nch = cB - cA + 1;
    auto share = [&](int c, int& lo, int& hi) {                      // the chunk's share of channel c's INPUT plane
        lo = max(c * d.HW, ce0) - c * d.HW;
        hi = min((c + 1) * d.HW, ce1) - c * d.HW;
    };
    auto publish = [&](int c, int k, float v) {
        const int a = (c * d.HW) / kS2Chunk, b = ((c + 1) * d.HW - 1) / kS2Chunk;
        const size_t at = ((size_t)c * 3 + k) * P + n;               // index j N + n
        const int j = chunk == a ? 0 : 1;
        if (FUSED) {
            fin_publish(fin.f, at + (size_t)j * d.N, v);
        } else {
            part[at + (size_t)j * d.N] = v;
            if (a == b) part[at + d.N] = 0.f;
        }
    };
    for (int base = 0; base < 3 * nch; base += kBlock / 16) {        // 16 lanes (one DPP row) per (channel, component)
        const int item = base + (tid >> 4), sub = tid & 15;
        const bool valid = item < 3 * nch;
        const int it = valid ? item : 0;
        const int k = it / nch, c = cA + it - k * nch;
        int lo, hi;
        share(c, lo, hi);
        const float* a0 = accf + k * kS2Chunk + (c * d.HW - ce0);    // the plane's element 0
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                // four chains, fixed order
        if (valid) {
            int i = lo + sub;
            for (; i + 48 < hi; i += 64) { s0 += a0[i]; s1 += a0[i + 16]; s2 += a0[i + 32]; s3 += a0[i + 48]; }
            for (; i < hi; i += 16) s0 += a0[i];
        }
        float sum = (s0 + s1) + (s2 + s3);
        sum += dpp_or_zero<0x111, 0xf>(sum);
        sum += dpp_or_zero<0x112, 0xf>(sum);
        sum += dpp_or_zero<0x114, 0xf>(sum);
        sum += dpp_or_zero<0x118, 0xf>(sum);                         // lane 15 of the row: the item's sum
        const bool slow_ch = a0[(3 - k) * kS2Chunk + lo] != 0.f;     // the helpers below publish a slow channel
        if (valid && sub == 15 && !slow_ch) publish(c, k, sum);
    }
    if (any_slow) {
        for (int c = cA; c <= cB; ++c) {
            if (!s2_channel_is_slow<true>(d, shift, c)) continue;
            int lo, hi;
            share(c, lo, hi);
            // the output-side helpers sum over OUTPUT elements: the chunk's share of the output plane, cut at the same fractions
            const int lo_o = (int)(((long long)lo * d.HWo + d.HW - 1) / d.HW), hi_o = (int)(((long long)hi * d.HWo + d.HW - 1) / d.HW);
            float sT = 0.f, sH = 0.f, sW = 0.f;
            for (int t = 0; t < d.T; ++t) {
                if (WRITE_GX) backward_input_plane<float, false>(shift, gy, gx, gd, n, t, c, tid, kBlock, lo, hi);
                shift_grad_plane<float>(x, shift, gy, gd, n, t, c, tid, kBlock, sT, sH, sW, NoAct(), lo_o, hi_o);
            }
            sT = group_sum(sT, kBlock, red[0]);
            sH = group_sum(sH, kBlock, red[1]);
            sW = group_sum(sW, kBlock, red[2]);
            if (tid == 0) { publish(c, 0, sT); publish(c, 1, sH); publish(c, 2, sW); }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The forward of the same layers (stride (1,2,2), 28 -> 14 and 14 -> 7), on the OUTPUT side: a wave owns 128 consecutive
// elements of the y slab (two per lane); the 2 x 2 tap windows of consecutive outputs tile consecutive input rows, so
// their taps lie in ONE run of the x slab -- 4 x 128 floats + a row and a half each side, <= 160 cells -- that three DMA
// instructions per plane bring into a ring of 3 private slots.  The rest is k3d_slab_interp's walk.
constexpr int kF2Cells = 160, kF2Z = kF2Cells * 16, kF2Stride = (kF2Cells + 1) * 16, kF2RG = 3, kF2NF = 3, kF2Chunk = 512;

__global__ __launch_bounds__(kBlock) void k3d_slab_s2_forward(const float* __restrict__ x, const float* __restrict__ shift,
                                                              float* y, S2Dims d, Dims3 gd) {
    constexpr int M = 2;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = (int)threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunks = (d.slab_out + kF2Chunk - 1) / kF2Chunk;
    const int chunk = (int)(blockIdx.x % (unsigned)nchunks), n = (int)(blockIdx.x / (unsigned)nchunks);
    const int ow0 = chunk * kF2Chunk + wave * (kWave * M);            // the wave's first OUTPUT element
    const int o0 = ow0 + M * lane;
    const bool cell_live = o0 < d.slab_out;
    const size_t nb_in = (size_t)n * d.T * d.slab_in, nb_out = (size_t)n * d.T * d.slab_out;

    LaneShift<M> ls;
    {
        const float* sb = uniform_ptr(shift);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int c = min(o0 + m, d.slab_out - 1) / d.HWo;
#pragma unroll
            for (int k = 0; k < 3; ++k) { ls.s[m][k] = 0.f; load_f1(ls.s[m][k], sb, (k * d.C + c) * 4); }
        }
    }
    // flat index in the x slab of the input element (2 ho, 2 wo) of output element o
    auto x0_of = [&](int o) {
        const int c = o / d.HWo, p = o - c * d.HWo;
        const int ho = p / d.Wo, wo = p - ho * d.Wo;
        return c * d.HW + 2 * ho * d.W + 2 * wo;
    };
    const int xl = ow0 < d.slab_out ? x0_of(ow0) - d.W - 1 : 0;
    const int pc0 = __builtin_amdgcn_readfirstlane(xl >= 0 ? xl >> 2 : -((3 - xl) >> 2));           // floor(xl / 4)
    const int slab_cells = d.slab_in >> 2;
    bool act[kF2NF];
    int nF = 0;
#pragma unroll
    for (int i = 0; i < kF2NF; ++i) {
        const int pc = kWave * i + lane;
        act[i] = ow0 < d.slab_out && pc < kF2Cells && pc0 + pc >= 0 && pc0 + pc < slab_cells;
        nF += __ballot(act[i]) != 0ull ? 1 : 0;
    }
    nF = __builtin_amdgcn_readfirstlane(nF);
    const unsigned g0 = __builtin_amdgcn_readfirstlane(lds_byte_addr(lds_raw)) + (unsigned)(wave * (kF2RG * kF2Stride));
    const float* xp = x + nb_in + (ptrdiff_t)pc0 * 4;                 // never dereferenced outside the slab
    auto fetch = [&](int p, int slot) {
        const float* src = uniform_ptr(xp + (size_t)p * d.slab_in);
#pragma unroll
        for (int i = 0; i < kF2NF; ++i)
            if (act[i]) dma16s<true>(src, (kWave * i + lane) * 16, g0 + (unsigned)(slot * kF2Stride + kWave * i * 16));
    };
#pragma unroll
    for (int p = 0; p < kF2RG; ++p)
        if (p < d.T) fetch(p, p);
    if (lane < kF2RG) *reinterpret_cast<float4*>(lds_raw + wave * (kF2RG * kF2Stride) + lane * kF2Stride + kF2Z) = make_float4(0.f, 0.f, 0.f, 0.f);

    wait_vmcnt(nF * min(kF2RG, d.T));                                // the shift loads are older than the fetches
    tie<M>(ls);
    Elem el[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int o = o0 + m;
        const bool live = o < d.slab_out;
        const int oc = live ? o : 0;
        const int c = oc / d.HWo, p = oc - c * d.HWo;
        const int ho = p / d.Wo, wo = p - ho * d.Wo;
        const Frac<float> fT = split_shift(ls.s[m][0]), fH = split_shift(ls.s[m][1]), fW = split_shift(ls.s[m][2]);
        const bool near = (unsigned)(fT.fl + 1) < 2u && (unsigned)(fH.fl + 1) < 2u && (unsigned)(fW.fl + 1) < 2u;
        el[m].fast = live && near;
        el[m].slow = live && !near;
        el[m].f0 = fT.fl == 0;
        el[m].rT = fT.r; el[m].rH = fH.r; el[m].rW = fW.r;
        const int h0 = 2 * ho + fH.fl, w0 = 2 * wo + fW.fl;
        const bool mh0 = (unsigned)h0 < (unsigned)d.H, mh1 = (unsigned)(h0 + 1) < (unsigned)d.H;
        const bool mw0 = (unsigned)w0 < (unsigned)d.W, mw1 = (unsigned)(w0 + 1) < (unsigned)d.W;
        const int g = c * d.HW + h0 * d.W + w0 - 4 * pc0;             // float index of tap (0,0) inside the piece
        const bool in = g >= 0 && g + d.W + 1 < 4 * kF2Cells;
        const unsigned a = g0 + (unsigned)(g * 4), Z = g0 + kF2Z;
        el[m].rel[0] = el[m].fast && in && mh0 && mw0 ? a : Z;
        el[m].rel[1] = el[m].fast && in && mh0 && mw1 ? a + 4u : Z;
        el[m].rel[2] = el[m].fast && in && mh1 && mw0 ? a + 4u * d.W : Z;
        el[m].rel[3] = el[m].fast && in && mh1 && mw1 ? a + 4u * d.W + 4u : Z;
    }
    bool all_fast = true, thread_slow = false;
#pragma unroll
    for (int m = 0; m < M; ++m) { all_fast = all_fast && el[m].fast; thread_slow = thread_slow || el[m].slow; }
    const bool regular = d.T == kMaxT && nF == kF2NF && __ballot(all_fast) == ~0ull;      // wave-uniform

    float* optr = y + nb_out + (size_t)o0;
    auto walk = [&](auto REGC) {
        constexpr bool REG = decltype(REGC)::value;
        float Bprev[M], vprev[M];
#pragma unroll
        for (int m = 0; m < M; ++m) { Bprev[m] = 0.f; vprev[m] = 0.f; }
        for_each_step(std::make_integer_sequence<int, kMaxT + 1>{}, [&](auto KC) {
            constexpr int k = decltype(KC)::value;
            if (!REG && k > d.T) return;
            float B[M];
#pragma unroll
            for (int m = 0; m < M; ++m) B[m] = 0.f;
            if (k < kMaxT && (REG || k < d.T)) {
                if (REG) wait_lit<allowed(k < kMaxT ? k : 0, kF2NF, kF2RG, true)>(); else wait_vmcnt(0);
                float q[M][4];
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) q[m][j] = lds_at(el[m].rel[j], (k % kF2RG) * kF2Stride);
                if (k + kF2RG < kMaxT && (REG || k + kF2RG < d.T)) fetch(k + kF2RG, k % kF2RG);
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float uW = 1 - el[m].rW, uH = 1 - el[m].rH;
                    B[m] = uH * (q[m][0] * uW + q[m][1] * el[m].rW) + el[m].rH * (q[m][2] * uW + q[m][3] * el[m].rW);
                }
            }
            float o[M];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float v = (1 - el[m].rT) * Bprev[m] + el[m].rT * B[m];
                o[m] = el[m].f0 ? v : vprev[m];
                vprev[m] = v; Bprev[m] = B[m];
            }
            if (k >= 1) {
                float* out = optr + (size_t)(k - 1) * d.slab_out;
                const f32x2 t = {o[0], o[1]};
                if (REG) {
                    __builtin_nontemporal_store(t, reinterpret_cast<f32x2*>(out));
                } else if (cell_live) {
                    if (all_fast) __builtin_nontemporal_store(t, reinterpret_cast<f32x2*>(out));
                    else {
#pragma unroll
                        for (int m = 0; m < M; ++m)
                            if (el[m].fast) out[m] = o[m];
                    }
                }
            }
        });
    };
    if (regular) walk(std::true_type{}); else walk(std::false_type{});

    if (__syncthreads_or((int)thread_slow)) {                        // workgroup-uniform from here on: the generic loop per slow channel
        const int ce0 = chunk * kF2Chunk, ce1 = min(d.slab_out, ce0 + kF2Chunk);
        for (int c = ce0 / d.HWo; c <= (ce1 - 1) / d.HWo; ++c) {
            const Frac<float> fT = split_shift(shift[c]), fH = split_shift(shift[d.C + c]), fW = split_shift(shift[2 * d.C + c]);
            if ((unsigned)(fT.fl + 1) < 2u && (unsigned)(fH.fl + 1) < 2u && (unsigned)(fW.fl + 1) < 2u) continue;
            const int lo = max(c * d.HWo, ce0) - c * d.HWo, hi = min((c + 1) * d.HWo, ce1) - c * d.HWo;
            const float* xc = x + nb_in + (size_t)c * d.HW;
            for (int to = 0; to < d.T; ++to) {
                float* yp = y + nb_out + (size_t)to * d.slab_out + (size_t)c * d.HWo;
                const int t0 = to + fT.fl;
                const bool v0 = t0 >= 0 && t0 < d.T, v1 = t0 + 1 >= 0 && t0 + 1 < d.T;
                const float* p0 = xc + (v0 ? (size_t)t0 * d.slab_in : 0);
                const float* p1 = xc + (v1 ? (size_t)(t0 + 1) * d.slab_in : 0);
                for (int i = lo + tid; i < hi; i += kBlock) {
                    const int ho = i / d.Wo, wo = i - ho * d.Wo;
                    const int h0 = 2 * ho + fH.fl, w0 = 2 * wo + fW.fl;
                    const bool mh0 = h0 >= 0 && h0 < d.H, mh1 = h0 + 1 >= 0 && h0 + 1 < d.H;
                    const bool mw0 = w0 >= 0 && w0 < d.W, mw1 = w0 + 1 >= 0 && w0 + 1 < d.W;
                    const int o00 = h0 * d.W + w0;
                    float q000 = 0, q001 = 0, q010 = 0, q011 = 0, q100 = 0, q101 = 0, q110 = 0, q111 = 0;
                    if (v0) {
                        if (mh0 && mw0) q000 = p0[o00];
                        if (mh0 && mw1) q001 = p0[o00 + 1];
                        if (mh1 && mw0) q010 = p0[o00 + d.W];
                        if (mh1 && mw1) q011 = p0[o00 + d.W + 1];
                    }
                    if (v1) {
                        if (mh0 && mw0) q100 = p1[o00];
                        if (mh0 && mw1) q101 = p1[o00 + 1];
                        if (mh1 && mw0) q110 = p1[o00 + d.W];
                        if (mh1 && mw1) q111 = p1[o00 + d.W + 1];
                    }
                    yp[i] = trilerp(q000, q001, q010, q011, q100, q101, q110, q111, fT.r, fH.r, fW.r);
                }
            }
        }
    }
    (void)gd;
}

// ---------------------------------------------------------------------------------------------
// Host side.
// 14x14 planes: the forward / d(x)-only kernels here are level with the tile kernels at C = 216 and 10 % faster at C = 288
// (22.0 vs 24.7 us), the fused backward is level (29.2 vs 29.1 us) and rk3d_tile.hpp has the BatchNorm-fused variants, so by
// default only the forward comes here.  RK_SLAB14 = 1: both, 0: neither.
static int slab14_mode() {
    static const int v = [] { const char* e = getenv("RK_SLAB14"); return e ? (e[0] == '1' ? 2 : 0) : 1; }();
    return v;
}
bool slab14_on(bool backward) { return slab14_mode() >= (backward ? 2 : 1); }
static bool make_sdims(SDims& s, const Dims3& d, int M) {
    const bool s1p0 = d.sT == 1 && d.sH == 1 && d.sW == 1 && d.pT == 0 && d.pH == 0 && d.pW == 0;
    if (!s1p0 || !streaming_kernels_on()) return false;
    if (d.T > kMaxT || d.W > 15 || d.H * d.W > 256 || d.H * d.W < kMinHW) return false;
    const long long slab = (long long)d.C * d.H * d.W;
    if (slab % 4 != 0 || slab > 0x1fffffff) return false;
    s.N = d.N; s.T = d.T; s.C = d.C; s.H = d.H; s.W = d.W; s.HW = d.H * d.W;
    s.slab = (int)slab;
    s.nchunks = (int)((slab + kBlock * M - 1) / (kBlock * M));
    return true;
}
static int halo_of(const SDims& s) { return s.W + 1 <= 8 ? 2 : 4; }
template <int M, int HL, int RG> constexpr size_t lds_of() { return (size_t)(kBlock / kWave) * RG * Geo<M, HL>::kStride; }
// planes a wave keeps in flight: 3 in the forward (16.6 / 18.2 / 21.8 us at 7x7x576 / 14x14x216 / 14x14x288 against 16.5 /
// 18.4 / 23.3 with all 8), all 8 in the backward (95 VGPRs, 5 waves per SIMD).  The backward with a ring of 3 needed 111 VGPRs
// and SPILLED -- among other things the x values load_x2 had just requested, one instruction behind the load (the compiler
// takes an asm load's result for ready): stale data, found by tools/asm_hazard_check.py in round 6.  That instantiation, only
// reachable through the RK_SLAB_RG tuning switch, is gone together with the switch.
constexpr int kRingFwd = 3, kRingBwd = kMaxT;

bool launch_interp(bool negate, const float* src, const float* shift, float* dst, const Dims3& d, hipStream_t stream) {
    SDims s;
    if (!make_sdims(s, d, 4) || !aligned16(src) || !aligned16(dst)) return false;
    const unsigned grid = (unsigned)((long long)s.N * s.nchunks);
#define RK_SLAB_FWD(NG, HL, RG) hipLaunchKernelGGL((k3d_slab_interp<NG, HL, RG>), dim3(grid), dim3(kBlock), \
                                                   (lds_of<4, HL, RG>()), stream, src, shift, dst, s, d)
#define RK_SLAB_FWD_R(NG, HL) RK_SLAB_FWD(NG, HL, kRingFwd)
    if (halo_of(s) == 2) { if (negate) RK_SLAB_FWD_R(true, 2); else RK_SLAB_FWD_R(false, 2); }
    else { if (negate) RK_SLAB_FWD_R(true, 4); else RK_SLAB_FWD_R(false, 4); }
#undef RK_SLAB_FWD_R
#undef RK_SLAB_FWD
    return true;
}

int launch_bwd(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws, const Dims3& d,
               int normalize, float t_factor, hipStream_t stream) {
    SDims s;
    if (!make_sdims(s, d, 2) || !aligned16(x) || !aligned16(gy) || (gx && !aligned16(gx))) return 0;
    const unsigned producers = (unsigned)((long long)s.N * s.nchunks);
    dma3d::Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = (int)producers;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
#define RK_SLAB_BWD(GX, FU, HL, RG) hipLaunchKernelGGL((k3d_slab_backward<GX, FU, HL, RG>), dim3(producers + (FU ? d.C : 0)), dim3(kBlock), \
                                                       (lds_of<2, HL, RG>() + 4 * kBlock * 2 * 4), stream, x, shift, gy, gx, ws, s, d, fin)
#define RK_SLAB_BWD_R(GX, FU, HL) RK_SLAB_BWD(GX, FU, HL, kRingBwd)
#define RK_SLAB_BWD_H(GX, FU) do { if (halo_of(s) == 2) RK_SLAB_BWD_R(GX, FU, 2); else RK_SLAB_BWD_R(GX, FU, 4); } while (0)
    if (gshift) { if (gx) RK_SLAB_BWD_H(true, true); else RK_SLAB_BWD_H(false, true); }
    else { if (gx) RK_SLAB_BWD_H(true, false); else RK_SLAB_BWD_H(false, false); }
#undef RK_SLAB_BWD_H
#undef RK_SLAB_BWD_R
#undef RK_SLAB_BWD
    return 2 * d.N;
}


// forward of the same layers; false = not handled here
bool launch_fwd_s2(const float* x, const float* shift, float* y, const Dims3& d, hipStream_t stream) {
    const bool s122 = d.sT == 1 && d.sH == 2 && d.sW == 2 && d.pT == 0 && d.pH == 0 && d.pW == 0;
    if (!s122 || !streaming_kernels_on()) return false;
    if (d.T > kMaxT || (d.H & 1) || (d.W & 1) || d.W > 56 || d.W < 4 || d.H < 2) return false;
    const long long slab_in = (long long)d.C * d.H * d.W, slab_out = (long long)d.C * d.Ho * d.Wo;
    if (slab_in % 4 != 0 || slab_out % 2 != 0 || slab_in > 0x1fffffff) return false;
    if (!aligned16(x) || !aligned16(y)) return false;
    // the piece of a wave: 4 x 128 floats of windows + a row and a column each side, in 16-byte cells
    if ((4 * kWave * 2 + 2 * d.W + 2 + 3) / 4 + 2 > kF2Cells) return false;
    S2Dims s;
    s.N = d.N; s.T = d.T; s.C = d.C; s.H = d.H; s.W = d.W; s.HW = d.H * d.W; s.Wo = d.Wo; s.HWo = d.Ho * d.Wo;
    s.slab_in = (int)slab_in; s.slab_out = (int)slab_out;
    s.nchunks = 0;
    const unsigned grid = (unsigned)((long long)s.N * ((slab_out + kF2Chunk - 1) / kF2Chunk));
    hipLaunchKernelGGL(k3d_slab_s2_forward, dim3(grid), dim3(kBlock), (size_t)(kBlock / kWave) * kF2RG * kF2Stride, stream, x, shift, y, s, d);
    return true;
}

// stride (1,2,2) backward of the small strided layers (28 -> 14, 14 -> 7); returns P (0 = not handled here)
int launch_bwd_s2(const float* x, const float* shift, const float* gy, float* gx, float* gshift, float* ws, const Dims3& d,
                  int normalize, float t_factor, hipStream_t stream) {
    const bool s122 = d.sT == 1 && d.sH == 2 && d.sW == 2 && d.pT == 0 && d.pH == 0 && d.pW == 0;
    if (!s122 || !streaming_kernels_on()) return 0;
    if (d.T > kMaxT || (d.H & 1) || (d.W & 1) || d.W > 56 || d.W < 4 || d.H < 2) return 0;
    const long long slab_in = (long long)d.C * d.H * d.W, slab_out = (long long)d.C * d.Ho * d.Wo;
    if (slab_in % 4 != 0 || slab_out % 4 != 0 || slab_in > 0x1fffffff) return 0;
    if (!aligned16(x) || !aligned16(gy) || (gx && !aligned16(gx))) return 0;
    S2Dims s;
    s.N = d.N; s.T = d.T; s.C = d.C; s.H = d.H; s.W = d.W; s.HW = d.H * d.W; s.Wo = d.Wo; s.HWo = d.Ho * d.Wo;
    s.slab_in = (int)slab_in; s.slab_out = (int)slab_out;
    s.nchunks = (int)((slab_in + kS2Chunk - 1) / kS2Chunk);
    const unsigned producers = (unsigned)((long long)s.N * s.nchunks);
    dma3d::Fin3 fin;
    fin.f.gran = reinterpret_cast<unsigned long long*>(ws);
    fin_arm(fin.f);
    fin.f.producers = (int)producers;
    fin.gshift = gshift;
    fin.normalize = normalize;
    fin.t_factor = t_factor;
    const size_t lds = (size_t)(kBlock / kWave) * kMaxT * kS2Stride + 4 * kS2Chunk * 4;
#define RK_SLAB_S2(GX, FU) hipLaunchKernelGGL((k3d_slab_s2_backward<GX, FU>), dim3(producers + (FU ? d.C : 0)), dim3(kBlock), lds, \
                                              stream, x, shift, gy, gx, ws, s, d, fin)
    if (gshift) { if (gx) RK_SLAB_S2(true, true); else RK_SLAB_S2(false, true); }
    else { if (gx) RK_SLAB_S2(true, false); else RK_SLAB_S2(false, false); }
#undef RK_SLAB_S2
    return 2 * d.N;
}

}  // namespace slab3d
}  // namespace rk
