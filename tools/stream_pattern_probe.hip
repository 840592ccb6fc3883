// stream_pattern_probe.hip -- standalone probe (no torch): how fast does the memory system run a
// 1R+1W stream of 12.5 KB planes as a function of WHICH planes the concurrently resident
// workgroups touch?  Motivation: DESIGN 3.1 -- the 3-D kernels (a workgroup walks the T planes
// of one (n,c) column, stride C*H*W) sit at 5.45 TB/s where the 2-D twin (2 frames per
// workgroup) reaches 6.0-6.3 TB/s on the same bytes.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/stream_pattern_probe tools/stream_pattern_probe.hip
//
// x is [N,T,C,PL] float4 planes (PL = 784 float4 = one 56x56 fp32 plane).  A workgroup owns
// (n, c, chunk of S consecutive t) and copies its planes one after the other (register double
// buffer, nt loads / nt stores).  Block order: c fastest, then t-chunk, then n ("ctn") or
// c, n, t-chunk ("cnt").  `lds` pads dynamic LDS to cap workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PL = 784;       // float4 per plane
constexpr int RND = 4;        // ceil(784 / 256)

struct P { int N, T, C, S, order; };

__device__ __forceinline__ void ld_plane(f32x4 (&r)[RND], const f32x4* p) {
#pragma unroll
    for (int i = 0; i < RND; ++i) {
        const int o = threadIdx.x + 256 * i;
        if (o < PL) r[i] = __builtin_nontemporal_load(p + o);
    }
}
__device__ __forceinline__ void st_plane(const f32x4 (&r)[RND], f32x4* p) {
#pragma unroll
    for (int i = 0; i < RND; ++i) {
        const int o = threadIdx.x + 256 * i;
        if (o < PL) __builtin_nontemporal_store(r[i], p + o);
    }
}

__global__ __launch_bounds__(256) void k_walk(const f32x4* __restrict__ src, f32x4* __restrict__ dst, P p) {
    extern __shared__ float pad[];
    const int chunks = p.T / p.S;
    int b = blockIdx.x;
    const int c = b % p.C; b /= p.C;
    int n, ch;
    if (p.order == 0) { ch = b % chunks; n = b / chunks; } else { n = b % p.N; ch = b / p.N; }
    const size_t tstride = (size_t)p.C * PL;
    const size_t base = (((size_t)n * p.T + (size_t)ch * p.S) * p.C + c) * PL;
    f32x4 cur[RND], nxt[RND];
    ld_plane(nxt, src + base);
    for (int k = 0; k < p.S; ++k) {
#pragma unroll
        for (int i = 0; i < RND; ++i) cur[i] = nxt[i];
        if (k + 1 < p.S) ld_plane(nxt, src + base + (size_t)(k + 1) * tstride);
        st_plane(cur, dst + base + (size_t)k * tstride);
    }
    if (threadIdx.x == 1023) pad[0] = 0.f;
}


// ---- variant B: a workgroup owns (n, c, band of rows) for ALL T planes: every load is issued up front (registers),
// then every store; lifetime = one memory round trip.  cells_out float4 per plane are written, cells_in
// (>= cells_out: + halo rows) are read.  xcd != 0: the bands of a column sit on one XCD (consecutive ids there).
struct PB { int N, T, C, nb, cells_out, cells_in, xcd; };
template <int RB, int TT>
__global__ __launch_bounds__(256) void k_band(const f32x4* __restrict__ src, f32x4* __restrict__ dst, PB p) {
    int b = blockIdx.x, band, col;
    if (p.xcd) { const int x = b % 8, j = b / 8; band = j % p.nb; col = (j / p.nb) * 8 + x; }
    else { band = b % p.nb; col = b / p.nb; }
    const int c = col % p.C, n = col / p.C;
    if (n >= p.N) return;
    const size_t tstride = (size_t)p.C * PL;
    const size_t base = ((size_t)n * p.T * p.C + c) * PL + (size_t)band * p.cells_out;
    f32x4 r[TT][RB];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int o = threadIdx.x + 256 * i;
            if (o < p.cells_in && band * p.cells_out + o < PL) r[t][i] = __builtin_nontemporal_load(src + base + t * tstride + o);
        }
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int o = threadIdx.x + 256 * i;
            if (o < p.cells_out) __builtin_nontemporal_store(r[t][i], dst + base + t * tstride + o);
        }
}

// ---- variant C: 2R + 1W (the backward's mix): dst = a + b over the same walk as k_walk.
__global__ __launch_bounds__(256) void k_walk2(const f32x4* __restrict__ a, const f32x4* __restrict__ bsrc,
                                               f32x4* __restrict__ dst, P p) {
    extern __shared__ float pad[];
    const int chunks = p.T / p.S;
    int b = blockIdx.x;
    const int c = b % p.C; b /= p.C;
    int n, ch;
    if (p.order == 0) { ch = b % chunks; n = b / chunks; } else { n = b % p.N; ch = b / p.N; }
    const size_t tstride = (size_t)p.C * PL;
    const size_t base = (((size_t)n * p.T + (size_t)ch * p.S) * p.C + c) * PL;
    f32x4 ca[RND], cb[RND], na[RND], nb[RND];
    ld_plane(na, a + base);
    ld_plane(nb, bsrc + base);
    for (int k = 0; k < p.S; ++k) {
#pragma unroll
        for (int i = 0; i < RND; ++i) { ca[i] = na[i]; cb[i] = nb[i]; }
        if (k + 1 < p.S) { ld_plane(na, a + base + (size_t)(k + 1) * tstride); ld_plane(nb, bsrc + base + (size_t)(k + 1) * tstride); }
#pragma unroll
        for (int i = 0; i < RND; ++i) ca[i] += cb[i];
        st_plane(ca, dst + base + (size_t)k * tstride);
    }
    if (threadIdx.x == 1023) pad[0] = 0.f;
}

int main(int argc, char** argv) {
    const size_t planes = 32 * 8 * 64;
    const size_t bytes = planes * PL * 16;
    const int SETS = 3;
    f32x4 *src[SETS], *dst[SETS];
    for (int s = 0; s < SETS; ++s) {
        CK(hipMalloc(&src[s], bytes));
        CK(hipMalloc(&dst[s], bytes));
        CK(hipMemset(src[s], 1, bytes));
        CK(hipMemset(dst[s], 0, bytes));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Cfg { int N, T, C, S, order, lds; };
    std::vector<Cfg> cfgs;
    const int ldss[] = {0, 40 * 1024, 52 * 1024};   // uncapped (8/CU by waves), 4/CU, 3/CU
    for (int lds : ldss) {
        cfgs.push_back({32, 8, 64, 8, 0, lds});     // the 3-D walk
        cfgs.push_back({32, 8, 64, 4, 0, lds});     // T split in 2
        cfgs.push_back({32, 8, 64, 2, 0, lds});     // T split in 4
        cfgs.push_back({32, 8, 64, 1, 0, lds});     // one plane per workgroup
        cfgs.push_back({32, 8, 64, 4, 1, lds});     // T split, halves far apart in time
        cfgs.push_back({4, 8, 512, 8, 0, lds});     // long walk, few n resident
        cfgs.push_back({2, 8, 1024, 8, 0, lds});
        cfgs.push_back({1, 8, 2048, 8, 0, lds});
        cfgs.push_back({256, 1, 64, 1, 0, lds});
        cfgs.push_back({16, 16, 64, 16, 0, lds});   // longer walk
        cfgs.push_back({8, 32, 64, 32, 0, lds});
        cfgs.push_back({128, 8, 16, 8, 0, lds});    // many n resident (64 regions)
    }
    printf("%-28s %8s %8s %8s\n", "N,T,C,S,order,lds", "us(med)", "us(min)", "TB/s");
    for (const Cfg& c : cfgs) {
        P p{c.N, c.T, c.C, c.S, c.order};
        const unsigned grid = (unsigned)(c.N * c.C * (c.T / c.S));
        std::vector<float> ts;
        for (int it = 0; it < 14; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_walk, dim3(grid), dim3(256), (size_t)c.lds, 0, src[it % SETS], dst[(it + 1) % SETS], p);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        const float med = ts[ts.size() / 2];
        char name[64];
        snprintf(name, sizeof name, "%d,%d,%d,%d,%d,%d", c.N, c.T, c.C, c.S, c.order, c.lds / 1024);
        printf("%-28s %8.1f %8.1f %8.2f\n", name, med, ts[0], 2.0 * bytes / med / 1e6);
    }
    printf("\n%-34s %8s %8s %8s %8s\n", "band: nb,halo_rows,xcd,lds", "us(med)", "us(min)", "TB/s alg", "TB/s req");
    struct BC { int nb, halo, xcd, lds; };
    std::vector<BC> bcs;
    for (int lds : {0, 30 * 1024, 56 * 1024})
        for (int xcd : {0, 1})
            for (int halo : {0, 1})
                for (int nb : {1, 2, 4, 7, 14}) bcs.push_back({nb, halo, xcd, lds});
    for (const BC& c : bcs) {
        const int rows = 56 / c.nb;
        PB p{32, 8, 64, c.nb, rows * 14, (rows + c.halo) * 14, c.xcd};
        const int rb = (p.cells_in + 255) / 256;
        const unsigned grid = (unsigned)(32 * 64 * c.nb);
        std::vector<float> ts;
        for (int it = 0; it < 14; ++it) {
            CK(hipEventRecord(e0));
            const f32x4* s_ = src[it % SETS]; f32x4* d_ = dst[(it + 1) % SETS];
            switch (rb) {
                case 1: hipLaunchKernelGGL((k_band<1, 8>), dim3(grid), dim3(256), (size_t)c.lds, 0, s_, d_, p); break;
                case 2: hipLaunchKernelGGL((k_band<2, 8>), dim3(grid), dim3(256), (size_t)c.lds, 0, s_, d_, p); break;
                default: hipLaunchKernelGGL((k_band<4, 8>), dim3(grid), dim3(256), (size_t)c.lds, 0, s_, d_, p); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 2) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        const float med = ts[ts.size() / 2];
        char name[64];
        snprintf(name, sizeof name, "%d,%d,%d,%d", c.nb, c.halo, c.xcd, c.lds / 1024);
        const double req = bytes * (1.0 + (double)p.cells_in / p.cells_out);
        printf("%-34s %8.1f %8.1f %8.2f %8.2f\n", name, med, ts[0], 2.0 * bytes / med / 1e6, req / med / 1e6);
    }
    printf("\n# part 3: 2R + 1W (dst = a + b), same walk as part 1\n%-28s %8s %8s %8s\n", "N,T,C,S,order,lds", "us(med)", "us(min)", "TB/s");
    for (int lds : {0, 40 * 1024, 52 * 1024})
        for (int S : {8, 4, 2, 1}) {
            P p{32, 8, 64, S, 0};
            const unsigned grid = (unsigned)(32 * 64 * (8 / S));
            std::vector<float> ts;
            for (int it = 0; it < 14; ++it) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_walk2, dim3(grid), dim3(256), (size_t)lds, 0, src[it % SETS], dst[it % SETS], dst[(it + 1) % SETS], p);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 2) ts.push_back(ms * 1e3f);
            }
            std::sort(ts.begin(), ts.end());
            const float med = ts[ts.size() / 2];
            char name[64];
            snprintf(name, sizeof name, "32,8,64,%d,0,%d", S, lds / 1024);
            printf("%-28s %8.1f %8.1f %8.2f\n", name, med, ts[0], 3.0 * bytes / med / 1e6);
        }
    return 0;
}
