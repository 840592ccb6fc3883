// rk_dma.hpp -- the LDS-DMA streaming machinery shared by the 3-D (rk3d_dma.hpp) and 2-D (rk2d_dma.hpp)
// operators: band / cell geometry, the global_load_lds_dwordx4 wrapper with counted s_waitcnt, tap slots
// with a shared zero cell, compile-time tap selection, and the host-side band choice.  See rk3d_dma.hpp
// for the description of the scheme.
#pragma once
#include <atomic>
#include <chrono>

#include "rk3d_generic.hpp"

namespace rk {
namespace dma {


struct BDims {
    int N, T, C, H, W, W4;
    int BH, nbands;          // output rows per band (H % BH == 0), bands per plane
};

__device__ __forceinline__ unsigned lds_byte_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

__device__ __forceinline__ void stream_store(float4* p, const float4& v) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
}

// A wave-uniform pointer, guaranteed to sit in SGPRs (the "s" operand of the DMA asm below prints whatever register the
// value lives in: in the BN variants of the forward kernels the compiler kept the plane pointer in VGPRs and the
// assembler rejected `global_load_lds_dwordx4 v1, v[28:29]`).  Folds away when the value is scalar already.
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}

// One wave-instruction of LDS-DMA, saddr form: lane l copies 16 B from (sbase + voff_l) to LDS byte
// address lds_dst + 16*l.  The s_waitcnt lgkmcnt(0) orders it behind this wave's earlier LDS reads of
// the slot being refilled (and covers the M0 write -> use hazard).
template <bool NT = true>
__device__ __forceinline__ void dma16s(const void* sbase_uniform, int voff, unsigned lds_dst_uniform) {
    unsigned keep;
    if (NT)
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform)
            : "memory");
    else   // default cache policy: the line stays in this XCD's L2 for a second reader (rk3d_plane.hpp)
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase_uniform), "s"(lds_dst_uniform)
            : "memory");
}

#define RK_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
// wait until at most n VMEM ops of this wave are outstanding (n wave-uniform; clamped down = stricter)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n < 0 ? 0 : (n > 32 ? 32 : n)) {
        RK_VMCNT_CASE(0) RK_VMCNT_CASE(1) RK_VMCNT_CASE(2) RK_VMCNT_CASE(3) RK_VMCNT_CASE(4)
        RK_VMCNT_CASE(5) RK_VMCNT_CASE(6) RK_VMCNT_CASE(7) RK_VMCNT_CASE(8) RK_VMCNT_CASE(9)
        RK_VMCNT_CASE(10) RK_VMCNT_CASE(11) RK_VMCNT_CASE(12) RK_VMCNT_CASE(13) RK_VMCNT_CASE(14)
        RK_VMCNT_CASE(15) RK_VMCNT_CASE(16) RK_VMCNT_CASE(17) RK_VMCNT_CASE(18) RK_VMCNT_CASE(19)
        RK_VMCNT_CASE(20) RK_VMCNT_CASE(21) RK_VMCNT_CASE(22) RK_VMCNT_CASE(23) RK_VMCNT_CASE(24)
        RK_VMCNT_CASE(25) RK_VMCNT_CASE(26) RK_VMCNT_CASE(27) RK_VMCNT_CASE(28) RK_VMCNT_CASE(29)
        RK_VMCNT_CASE(30) RK_VMCNT_CASE(31) RK_VMCNT_CASE(32)
    }
}
#undef RK_VMCNT_CASE

// compile-time tap selection: the 5 consecutive values starting OFF floats into the aligned pair (q0, q1)
template <int OFF> __device__ __forceinline__ float tap(const float4& q0, const float4& q1, int k) {
    const int j = OFF + k;   // 0..7, constant after unrolling
    return j == 0 ? q0.x : j == 1 ? q0.y : j == 2 ? q0.z : j == 3 ? q0.w : j == 4 ? q1.x : j == 5 ? q1.y
         : j == 6 ? q1.z : q1.w;
}

// ---------------------------------------------------------------------------------------------
// Per-workgroup band geometry (wave-uniform) and per-thread cell geometry.
struct Band {
    int cells_out;               // output float4 cells of this band (BH * W4)
    int out0;                    // float4 index of the band's first output cell inside a plane
    int cells_in;                // tap-slot cells: (BH + 1) * W4
    int src0;                    // float4 index (may be negative) of the slot's cell 0 inside a source plane
    int s_lo, s_hi;              // slot cells [s_lo, s_hi) hold real source rows; the rest stay zero
};

__device__ __forceinline__ Band make_band(const BDims& d, int band, int flH) {
    Band b;
    b.cells_out = d.BH * d.W4;
    b.out0 = band * d.BH * d.W4;
    b.cells_in = (d.BH + 1) * d.W4;
    const int r0 = band * d.BH + flH;                     // source row held by slot row 0
    b.src0 = r0 * d.W4;
    int j_lo = r0 < 0 ? -r0 : 0;                          // first slot row inside the plane
    j_lo = j_lo > d.BH + 1 ? d.BH + 1 : j_lo;
    int j_hi = d.H - r0;                                  // first slot row past the plane
    j_hi = j_hi < 0 ? 0 : (j_hi > d.BH + 1 ? d.BH + 1 : j_hi);
    b.s_lo = j_lo * d.W4;
    b.s_hi = j_hi > j_lo ? j_hi * d.W4 : b.s_lo;
    return b;
}

template <int ROUNDS> struct BCells {
    int off0;                                            // tid * 16: byte offset of cell `tid`; round i adds 4096 i
    int a0[ROUNDS], a1[ROUNDS], b0[ROUNDS], b1[ROUNDS];  // tap float4 indices into a tap slot (zero cell if outside)
    bool in_act[ROUNDS];                                 // this lane DMAs tap-slot cell tid + 256 i
    bool tail_live;                                      // this lane owns an output cell in the last round
    bool tail_on;                                        // ... and this WAVE has at least one such lane (uniform)
    int xown;                                            // last round: own float4 index in an x slot, or its zero cell
    int n_tap_wave;                                      // tap-plane DMA instructions per plane for this wave (uniform)
    int n_out_wave;                                      // output stores / x DMA instructions per plane for this wave
};

template <int ROUNDS>
__device__ __forceinline__ void make_bcells(BCells<ROUNDS>& cs, const BDims& d, const Band& b, int group_shift) {
    cs.off0 = (int)threadIdx.x * 16;
    int n_tap = 0;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int o = (int)threadIdx.x + kBlock * i;
        const bool live = o < b.cells_out;
        const int oc = live ? o : 0;
        const int j = oc / d.W4, w4 = oc - j * d.W4;     // band-local output row, column group
        const int ga = w4 + group_shift, gb = ga + 1;
        const bool ga_ok = ga >= 0 && ga < d.W4, gb_ok = gb >= 0 && gb < d.W4;
        const int zero = b.cells_in;                     // dead lanes read zeros everywhere
        cs.a0[i] = (live && ga_ok) ? j * d.W4 + ga : zero;
        cs.a1[i] = (live && gb_ok) ? j * d.W4 + gb : zero;
        cs.b0[i] = (live && ga_ok) ? (j + 1) * d.W4 + ga : zero;
        cs.b1[i] = (live && gb_ok) ? (j + 1) * d.W4 + gb : zero;
        cs.in_act[i] = o >= b.s_lo && o < b.s_hi;
        n_tap += (__ballot(cs.in_act[i]) != 0ull) ? 1 : 0;
        if (i == ROUNDS - 1) { cs.tail_live = live; cs.xown = live ? o : b.cells_out; }
    }
    cs.tail_on = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~(kWave - 1))) + kBlock * (ROUNDS - 1) < b.cells_out;
    cs.n_tap_wave = __builtin_amdgcn_readfirstlane(n_tap);
    cs.n_out_wave = cs.tail_on ? ROUNDS : ROUNDS - 1;
}

// DMA the band of a tap plane (uniform pointer to slot cell 0's source, may lie before the plane) into a slot
template <int ROUNDS, bool NT = true>
__device__ __forceinline__ void dma_taps(const float* src0, unsigned slot_addr, const BCells<ROUNDS>& cs) {
    src0 = uniform_ptr(src0);
    const unsigned dst = slot_addr + __builtin_amdgcn_readfirstlane((unsigned)threadIdx.x >> 6) * 1024u;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
        if (cs.in_act[i]) dma16s<NT>(src0, cs.off0 + 4096 * i, dst + 4096u * i);
}
// zero the same cells instead (the plane lies outside [0, T))
template <int ROUNDS>
__device__ __forceinline__ void zero_taps(float4* slot, const BCells<ROUNDS>& cs) {
    char* base = reinterpret_cast<char*>(slot) + cs.off0;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
        if (cs.in_act[i]) *reinterpret_cast<float4*>(base + 4096 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
}
// Training fusion (train_block.py): the planes a shift kernel reads are z = conv2's output and the shift applies to
// relu(bn2(z)) = max(a z + b, 0).  The wave that DMA'd a piece transforms it in place once it has landed (its own
// counted vmcnt wait) and before the step's barrier -- the slot, not the taps, so that rows outside the plane and the
// shared zero cell, which were never DMA'd, stay zero.  Same expression as k_bn_apply_affine: bit-identical to
// "normalise, then shift".
template <int ROUNDS>
__device__ __forceinline__ void bn_taps(float4* slot, const BCells<ROUNDS>& cs, float a, float b) {
    char* base = reinterpret_cast<char*>(slot) + cs.off0;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i)
        if (cs.in_act[i]) {
            float4* p = reinterpret_cast<float4*>(base + 4096 * i);
            float4 v = *p;
            v.x = fmaxf(fmaf(a, v.x, b), 0.f); v.y = fmaxf(fmaf(a, v.y, b), 0.f);
            v.z = fmaxf(fmaf(a, v.z, b), 0.f); v.w = fmaxf(fmaf(a, v.w, b), 0.f);
            *p = v;
        }
}
// DMA the thread's own output-aligned cells of a plane (uniform pointer to the band's first cell)
template <int ROUNDS>
__device__ __forceinline__ void dma_own(const float* band0, unsigned slot_addr, const BCells<ROUNDS>& cs) {
    band0 = uniform_ptr(band0);
    const unsigned dst = slot_addr + __builtin_amdgcn_readfirstlane((unsigned)threadIdx.x >> 6) * 1024u;
#pragma unroll
    for (int i = 0; i + 1 < ROUNDS; ++i) dma16s(band0, cs.off0 + 4096 * i, dst + 4096u * i);
    if (cs.tail_on && cs.tail_live) dma16s(band0, cs.off0 + 4096 * (ROUNDS - 1), dst + 4096u * (ROUNDS - 1));
}
template <int ROUNDS>
__device__ __forceinline__ void zero_own(float4* slot, const BCells<ROUNDS>& cs) {
    char* base = reinterpret_cast<char*>(slot) + cs.off0;
#pragma unroll
    for (int i = 0; i + 1 < ROUNDS; ++i) *reinterpret_cast<float4*>(base + 4096 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cs.tail_live) *reinterpret_cast<float4*>(base + 4096 * (ROUNDS - 1)) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// once per kernel: zero cell of every slot + the slot rows that lie outside the plane
template <int ROUNDS>
__device__ __forceinline__ void init_tap_slots(float4* ring, int nslots, int slot_f4, const Band& b,
                                               const BCells<ROUNDS>& cs) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nslots; ++s) {
        float4* slot = ring + s * slot_f4;
        if (threadIdx.x == 0) slot[b.cells_in] = z;
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int o = (int)threadIdx.x + kBlock * i;
            if (o < b.cells_in && !cs.in_act[i]) slot[o] = z;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row-sum of the d(shift) partials INSIDE the backward launch (instead of a separate finalize launch, ~4.5 us + a
// kernel boundary per call).  Producers do not wait for anything: a workgroup (or wave) writes each of its D
// partials as a pair of 8-byte granules ({fp32 value, launch tag} + a check granule, below) with device-scope stores
// -- the "data-tagged granule" hand-off of MI355X_MICROARCH.md (price-list rows handoff-1to1 / R2) -- and exits.  The
// LAST C blocks of the grid are finalizers, one wave per channel: they poll that channel's D*P granules with
// device-scope loads (s_sleep between sweeps) until every one carries this launch's tag, then sum them in index
// order in fp64 -- independent of arrival order, so results stay deterministic -- and the caller applies K5 / K9.
// Finalizers are dispatched behind the producers (and would be harmless ahead of them: C one-wave blocks against
// hundreds of producer slots; nothing waits on a finalizer).  The workspace is uninitialised memory: a stale
// granule passes for this launch's only if its upper 32 bits equal the tag (per-process counter seeded from the
// clock; p = 2^-32 per granule on memory these kernels never wrote); a finalizer retires every granule it has
// consumed (tag 0), so replaying a captured launch -- same tag -- is safe too.  A finalizer that has polled ~2 s gives
// up and the caller writes NaN: loud, never a hang.
// (The first fused version -- the last-ARRIVING producer finalizes, ticket by CAS -- cost +10 us: the store ->
// vmcnt(0) -> CAS round trips sat on every producer's exit while it held its LDS slot.)
struct Fin {
    unsigned long long* gran;     // [C][D][P] granule PAIRS (16 bytes per partial)
    unsigned tag;                 // != 0, unique per launch
    int producers;                // producer blocks; blocks beyond are finalizers
    int spins;                    // polls of one lane before its finalizer gives up (~0.25 us each)
};
// The library's only process-wide mutable state, both std::atomic (any thread, any device): the launch-tag counter and
// the finalizers' poll budget (a test hook: rk_debug_set_finalize_spins, rk_misc.hip).
inline std::atomic<unsigned>& launch_tag_counter() {
    static std::atomic<unsigned> tag{(unsigned)std::chrono::steady_clock::now().time_since_epoch().count() | 1u};
    return tag;
}
constexpr int kFinSpins = 8000000;                                   // ~2 s
inline std::atomic<int>& fin_spin_budget() {
    static std::atomic<int> spins{kFinSpins};
    return spins;
}
inline unsigned next_launch_tag() {
    std::atomic<unsigned>& tag = launch_tag_counter();
    unsigned t = tag.fetch_add(1, std::memory_order_relaxed);
    return t ? t : tag.fetch_add(1, std::memory_order_relaxed);
}
// every launch that hands partials to in-launch finalizers arms its Fin here: a fresh tag + the current poll budget
inline void fin_arm(Fin& f) {
    f.tag = next_launch_tag();
    f.spins = fin_spin_budget().load(std::memory_order_relaxed);
}
// A partial is handed over as a PAIR of 8-byte granules at gran[2*at], gran[2*at + 1]:
//   value granule {fp32 value, tag}   and   check granule {~value bits, tag2},  tag2 = a second word derived from tag.
// Each 8-byte store is single-copy atomic; a consumer accepts the pair only if both tags match AND the two payloads
// are complements, i.e. 128 consistent bits.  The workspace is uninitialised memory (torch.empty: stale fp32
// activations, whose bit patterns are anything but uniform): with the one-granule form of round 2 a stale word equal
// to the 32-bit tag would have passed for a finished partial (round-2 advisor finding); a stale 16-byte pattern that
// satisfies all three conditions by accident is out of reach (< 2^-90 per granule even for adversarial float data,
// tests/test_parity_3d.py::test_backward_ignores_adversarial_workspace_contents fills the workspace with near misses).
__device__ __forceinline__ unsigned fin_tag2(unsigned tag) { return tag * 2654435761u ^ 0x9e3779b9u; }
// A pair that is 16-byte aligned (any workspace that came from an allocator) moves as ONE 16-byte agent-scope access
// instead of two 8-byte ones: half the memory transactions of the hand-off (233 k scattered 8-byte stores + as many
// polls + as many retiring stores at [32,8,576,7,7]).  A 16-byte access is not single-copy atomic -- it does not have to
// be: a torn pair fails fin_ready (tags + complement) and is polled again, exactly like a half-written pair of 8-byte stores.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool fin_wide(const Fin& fin) { return ((size_t)fin.gran & 15) == 0; }     // wave-uniform
__device__ __forceinline__ void fin_store16(unsigned long long* p, unsigned long long v, unsigned long long w) {
    const u32x4 q = {(unsigned)v, (unsigned)(v >> 32), (unsigned)w, (unsigned)(w >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void fin_publish(const Fin& fin, size_t at, float v) {
    const unsigned bits = __float_as_uint(v);
    const unsigned long long a = ((unsigned long long)fin.tag << 32) | bits;
    const unsigned long long b = ((unsigned long long)fin_tag2(fin.tag) << 32) | (unsigned)~bits;
    if (fin_wide(fin)) { fin_store16(fin.gran + 2 * at, a, b); return; }
    __hip_atomic_store(fin.gran + 2 * at, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(fin.gran + 2 * at + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// both granules of a pair carry this launch's tags and agree on the payload
__device__ __forceinline__ bool fin_ready(const Fin& fin, unsigned long long v, unsigned long long w) {
    return (unsigned)(v >> 32) == fin.tag && (unsigned)(w >> 32) == fin_tag2(fin.tag) && (unsigned)v == ~(unsigned)w;
}
// one wave (lanes 0..63 of the block): s[k] = sum_i granule[c][k][i] over this launch's P partials; false = timed out.
// The D pairs of a lane (row stride P pairs, lane index i) are fetched in ONE asm statement -- D 16-byte device-scope
// loads and the s_waitcnt that covers them: one round trip per poll, not D dependent ones (the finalizers' sweep is the
// tail of the launch).  The wait sits INSIDE the statement on purpose: to the compiler an asm load is an ordinary
// instruction whose result is there when the statement ends, so it may copy the destination registers right behind it.
// Round 5 issued the loads and the wait as separate statements ("+v" ties in between); that held only as long as the
// register allocator happened to insert no copy between them -- a loop bound that became a kernel argument (the poll
// budget, round 6) moved the allocation, the copies appeared (v_mov of the destination one instruction after the load) and
// every finalizer read stale registers, timed out and wrote NaN.  A lane that is not `act` keeps what v / w hold.
template <int D>
__device__ __forceinline__ void fin_load(const Fin& fin, unsigned long long* g, int P, int i, unsigned long long (&v)[D],
                                         unsigned long long (&w)[D], bool act) {
    if (!act) return;
    if (fin_wide(fin)) {
        u32x4 q[D];
        const unsigned long long* a[D];
#pragma unroll
        for (int k = 0; k < D; ++k) a[k] = g + 2 * ((size_t)k * P + i);
        if constexpr (D == 2)
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q[0]), "=&v"(q[1]) : "v"(a[0]), "v"(a[1]) : "memory");
        else if constexpr (D == 3)
            asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                         "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]) : "v"(a[0]), "v"(a[1]), "v"(a[2]) : "memory");
        else if constexpr (D == 4)
            asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                         "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
        else if constexpr (D == 5)
            asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\t"
                         "global_load_dwordx4 %2, %7, off sc1\n\tglobal_load_dwordx4 %3, %8, off sc1\n\t"
                         "global_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]) : "memory");
        else {
#pragma unroll
            for (int k = 0; k < D; ++k)
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(q[k]) : "v"(a[k]) : "memory");
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            v[k] = ((unsigned long long)q[k].y << 32) | q[k].x;
            w[k] = ((unsigned long long)q[k].w << 32) | q[k].z;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const size_t at = 2 * ((size_t)k * P + i);
        v[k] = __hip_atomic_load(g + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w[k] = __hip_atomic_load(g + at + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// `count` <= P: the partials that exist for this channel (rows keep the stride P)
template <int D>
__device__ __forceinline__ bool fin_collect(const Fin& fin, int c, int P, double (&s)[D], int count = -1) {
    const int lane = threadIdx.x;
    if (count < 0) count = P;
    unsigned long long* g = fin.gran + 2 * (size_t)c * D * P;
    const unsigned long long done = (unsigned long long)fin.tag << 32;          // a ready pair holding 0.0f
    const unsigned long long done2 = ((unsigned long long)fin_tag2(fin.tag) << 32) | 0xffffffffull;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < D; ++k) s[k] = 0;
    for (int i0 = 0; i0 < count; i0 += kWave) {
        const int i = i0 + lane;
        const bool act = i < count;
        unsigned long long v[D], w[D];
#pragma unroll
        for (int k = 0; k < D; ++k) { v[k] = done; w[k] = done2; }
        fin_load<D>(fin, g, P, i, v, w, act);
        for (int spin = 0; spin < fin.spins; ++spin) {               // ~0.25 us per poll: gives up after ~2 s (kFinSpins)
            bool ready = true;
#pragma unroll
            for (int k = 0; k < D; ++k) ready = ready && fin_ready(fin, v[k], w[k]);
            if (ready) break;
            __builtin_amdgcn_s_sleep(8);
            fin_load<D>(fin, g, P, i, v, w, act);        // (all D pairs again: a pair that was ready reads back the same)
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            ok = ok && fin_ready(fin, v[k], w[k]);
            s[k] += (double)__uint_as_float((unsigned)v[k]);
            // consumed: retire the pair (tag 0 is never issued), so that a REPLAY of this launch with the same
            // tag -- a captured hipGraph -- cannot take the previous replay's partials for its own
            if (act) {
                const size_t at = 2 * ((size_t)k * P + i);
                if (fin_wide(fin)) fin_store16(g + at, 0ull, 0ull);
                else {
                    __hip_atomic_store(g + at, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(g + at + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) s[k] = wave_sum(s[k]);
    return __all(ok) != 0;
}

// ---------------------------------------------------------------------------------------------
// Host side: band choice.
inline int rounds_for(int cells) { return (cells + kBlock - 1) / kBlock; }

// Equal bands with (BH + 1) * W4 <= 1024 cells and the same number of rounds on the tap side and on the
// output side (so rounds 0..ROUNDS-2 are full).  Needs b.H and b.W4; false = no such banding.
// two_rounds: prefer the fewest bands of at most 2 rounds (512 cells) that are still >= 6 KB of a plane -- for the
// fused backward kernels, whose 4-round form holds 168 VGPRs (3 workgroups per CU): 56x56 in two 28-row bands is
// 116 VGPRs, 4 workgroups per CU and twice the workgroups, 115.0 -> 112.3 us on [32,8,64,56,56] although each band
// re-reads one halo row of gy (+1.2 % bytes); the forward kernels are faster on whole planes (68 vs 76 us).
inline bool choose_bands(BDims& b, bool two_rounds = false) {
    if (two_rounds) {
        for (int nb = 1; nb <= b.H; ++nb) {
            if (b.H % nb) continue;
            const int bh = b.H / nb, co = bh * b.W4, ci = (bh + 1) * b.W4;
            if (co * 16 < 6144) break;                               // bands below ~6 KB collapse (pattern probe)
            if (ci > 2 * kBlock || rounds_for(co) != rounds_for(ci)) continue;
            b.nbands = nb; b.BH = bh;
            return true;
        }
    }
    for (int nb = 1; nb <= b.H; ++nb) {
        if (b.H % nb) continue;
        const int bh = b.H / nb, co = bh * b.W4, ci = (bh + 1) * b.W4;
        if (ci > 4 * kBlock) continue;
        if (rounds_for(co) != rounds_for(ci)) continue;
        b.nbands = nb; b.BH = bh;
        return true;
    }
    return false;
}

inline int rounds_of(const BDims& b) { return rounds_for((b.BH + 1) * b.W4); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline size_t interp_ring_bytes(const BDims& b, int D) { return (size_t)(D + 1) * ((b.BH + 1) * b.W4 + 1) * 16; }
inline size_t bwd_ring_bytes(const BDims& b, int DG, int DX) {
    return ((size_t)(DG + 1) * ((b.BH + 1) * b.W4 + 1) + (size_t)DX * (b.BH * b.W4 + 1)) * 16;
}

}  // namespace dma
}  // namespace rk
