// line_probe.hip -- dev-only probe (not product, not a test): what does the MI355X memory system deliver for the
// access shapes of the walk kernel?  Stand-alone program, no torch:
//
//     hipcc -O3 --offload-arch=gfx950 tools/line_probe.hip -o tools/build/line_probe && tools/build/line_probe
//
// Shapes (all from 2048 workgroups x 256 lanes = 8 workgroups per CU, the walk kernel's residency):
//   gather4    K independent random 4-byte reads per lane and iteration (the neighbour read, no dependency)
//   gather8    the same with 8-byte reads (an indptr pair)
//   halves     lanes 2j / 2j+1 read the two 64-byte halves of ONE random 128-byte line: if an L2 miss moves a whole
//              128-byte line this costs the same as gather4 at half the lines; if it moves 64 bytes, the same per read
//   chain4     dependent chase pos = table[pos] (one load in flight per lane, like one walk)
//   walk2      the walk's two dependent reads per hop: 8-byte pair from a small table (row pointer), then 4 bytes
//              from the big one at a position derived from it
//   rec8       ONE dependent 8-byte read per hop from a table twice the size (packed hop record)
// Table sizes span L2 (4 MiB per XCD), Infinity Cache (256 MiB) and HBM.  Every kernel reports reads/s; run under
// `rocprofv3 --pmc FETCH_SIZE`, `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`, `--pmc TCC_MISS_sum TCC_REQ_sum`
// (PROBE_REPS=1) for the bytes per miss; dispatch order = print order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e__ = (x);                                                      \
        if (e__ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint64_t pos64(uint32_t a, uint32_t b, uint64_t n) {
    uint64_t h = ((uint64_t)a << 32 | b) + 0x9e3779b97f4a7c15ull;     // splitmix64 finaliser: every (a, b) its own position
    h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ull;
    h = (h ^ (h >> 27)) * 0x94d049bb133111ebull;
    h ^= h >> 31;
    return (uint64_t)(((unsigned __int128)h * n) >> 64);
}

constexpr int kThreads = 256;
constexpr int kLdsPad = 16 * 1024;   // 16 KB per workgroup -> 8 workgroups per CU, like the walk kernel

__global__ void fill_random(uint32_t *t, uint64_t n, uint64_t modulo, uint32_t salt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        t[i] = (uint32_t)pos64((uint32_t)i, (uint32_t)(i >> 32) + salt, modulo);
}

template <int K>
__global__ __launch_bounds__(kThreads) void gather4(const uint32_t *__restrict__ t, uint64_t n, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = t[pos64(gid, it * K + k, n)];
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= v[k];
    }
    if (acc == 0x12345678u) out[gid] = acc + pad[threadIdx.x];
}

template <int K>
__global__ __launch_bounds__(kThreads) void gather8(const uint2 *__restrict__ t, uint64_t n8, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint2 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = t[pos64(gid, it * K + k, n8)];
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= v[k].x ^ v[k].y;
    }
    if (acc == 0x12345678u) out[gid] = acc + pad[threadIdx.x];
}

template <int K>
__global__ __launch_bounds__(kThreads) void halves(const uint32_t *__restrict__ t, uint64_t n, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    const uint64_t lines = n / 32;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint64_t line = pos64(gid >> 1, it * K + k, lines);
            v[k] = t[line * 32 + (gid & 1) * 16 + (mix(gid + it) & 15)];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc ^= v[k];
    }
    if (acc == 0x12345678u) out[gid] = acc + pad[threadIdx.x];
}

// the same chase with other load flavours: does any of them move less than a 128-byte line per miss?
//   FL = 1 non-temporal (nt), 2 agent-scope atomic load (sc1), 3 system-scope atomic load (sc0 sc1)
template <int FL>
__device__ __forceinline__ uint32_t ld(const uint32_t *p) {
    if (FL == 1) return __builtin_nontemporal_load(p);
    if (FL == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (FL == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return *p;
}
template <int FL>
__global__ __launch_bounds__(kThreads) void chainfl(const uint32_t *__restrict__ t, uint64_t n, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    uint32_t p = (uint32_t)pos64(gid, 99, n);
    for (int it = 0; it < iters; ++it) p = ld<FL>(t + p);
    if (p == 0x12345678u) out[gid] = p + pad[threadIdx.x];
}

template <int K>
__global__ __launch_bounds__(kThreads) void chain4(const uint32_t *__restrict__ t, uint64_t n, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    uint32_t p[K];
#pragma unroll
    for (int k = 0; k < K; ++k) p[k] = (uint32_t)pos64(gid, k, n);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) p[k] = t[p[k]];   // table values are < n
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) acc ^= p[k];
    if (acc == 0x12345678u) out[gid] = acc + pad[threadIdx.x];
}

// two dependent reads per hop: rowptr pair (small table of `nodes`+1 entries, consecutive rows of `deg` entries) then
// a neighbour from the big table (values < nodes)
template <int K>
__global__ __launch_bounds__(kThreads) void walk2(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ nbr,
                                                  uint32_t nodes, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    uint32_t cur[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cur[k] = (uint32_t)pos64(gid, k, nodes);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t b = rowptr[cur[k]], e = rowptr[cur[k] + 1];
            const uint32_t r = mix(gid * 131u + it * K + k);
            cur[k] = nbr[b + r % (e - b)];
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) acc ^= cur[k];
    if (acc == 0x12345678u) out[gid] = acc + pad[threadIdx.x];
}

// one dependent 8-byte read per hop: record = {next row begin, degree << 24-ish}; here {x = begin of the next row, y = its degree}
template <int K>
__global__ __launch_bounds__(kThreads) void rec8(const uint2 *__restrict__ rec, uint64_t n8, uint32_t deg, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint32_t gid = blockIdx.x * kThreads + threadIdx.x;
    uint2 cur[K];
#pragma unroll
    for (int k = 0; k < K; ++k) cur[k] = make_uint2((uint32_t)pos64(gid, k, n8 - deg), deg);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t r = mix(gid * 131u + it * K + k);
            cur[k] = rec[cur[k].x + r % cur[k].y];
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) acc ^= cur[k].x;
    if (acc == 0x12345678u) out[gid] = acc + pad[threadIdx.x];
}

__global__ void fill_rec(uint2 *t, uint64_t n8, uint32_t deg) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * blockDim.x)
        t[i] = make_uint2((uint32_t)pos64((uint32_t)i, (uint32_t)(i >> 32) + 77u, n8 - deg), deg);
}
__global__ void fill_rowptr(uint32_t *t, uint32_t nodes, uint32_t deg) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nodes; i += (uint64_t)gridDim.x * blockDim.x)
        t[i] = (uint32_t)(i * deg);
}

static int g_reps = 3;
template <typename F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    double best = 1e30;
    for (int r = 0; r < g_reps; ++r) {
        CK(hipEventRecord(a, 0));
        launch();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

int main(int argc, char **argv) {
    if (getenv("PROBE_REPS")) g_reps = atoi(getenv("PROBE_REPS"));
    const uint64_t max_mb = getenv("PROBE_MAX_MB") ? strtoull(getenv("PROBE_MAX_MB"), nullptr, 10) : 12288;
    const int grid = 2048;
    const uint64_t lanes = (uint64_t)grid * kThreads;
    uint32_t *out;
    CK(hipMalloc(&out, lanes * 4));
    const uint64_t sizes_mb[] = {12, 64, 252, 1024, 12288};
    printf("shape,table_MB,loads_in_flight_per_lane,reads,ms,Greads_per_s,GBs_if_128B_per_read,GBs_if_64B_per_read\n");
    for (uint64_t mb : sizes_mb) {
        if (mb > max_mb) continue;
        const uint64_t n = mb * 1024 * 1024 / 4;
        uint32_t *t;
        CK(hipMalloc(&t, n * 4));
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, t, n, n, 1u);
        CK(hipDeviceSynchronize());
        auto report = [&](const char *shape, int k, uint64_t reads, double ms) {
            const double g = reads / ms / 1e6;
            printf("%s,%llu,%d,%llu,%.4f,%.2f,%.0f,%.0f\n", shape, (unsigned long long)mb, k, (unsigned long long)reads, ms, g,
                   g * 128, g * 64);
            fflush(stdout);
        };
        const int iters = 64;
#define RUN(KERN, K, T, N, IT)                                                                                   \
    do {                                                                                                         \
        double ms = time_ms([&] { hipLaunchKernelGGL((KERN<K>), dim3(grid), dim3(kThreads), kLdsPad, 0, T, N, IT, out); }); \
        report(#KERN, K, lanes * (uint64_t)(IT) * K, ms);                                                        \
    } while (0)
        RUN(gather4, 1, t, n, iters * 4);
        RUN(gather4, 4, t, n, iters);
        RUN(gather4, 8, t, n, iters);
        RUN(gather8, 4, (const uint2 *)t, n / 2, iters);
        RUN(halves, 4, t, n, iters);
        RUN(chain4, 1, t, n, iters);
        RUN(chain4, 2, t, n, iters);
        RUN(chain4, 4, t, n, iters);
        RUN(chainfl, 1, t, n, iters);
        RUN(chainfl, 2, t, n, iters);
        RUN(chainfl, 3, t, n, iters);
        CK(hipFree(t));
    }
    // the walk's two-level hop against the packed one-level hop, cit2-like sizes: 2,927,963 nodes x 21 = 61.5 M entries
    {
        const uint32_t nodes = 2927963, deg = 21;
        const uint64_t nnz = (uint64_t)nodes * deg;
        uint32_t *rowptr, *nbr;
        uint2 *rec;
        CK(hipMalloc(&rowptr, ((uint64_t)nodes + 1) * 4));
        CK(hipMalloc(&nbr, nnz * 4));
        CK(hipMalloc(&rec, nnz * 8));
        hipLaunchKernelGGL(fill_rowptr, dim3(4096), dim3(256), 0, 0, rowptr, nodes, deg);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, nbr, nnz, (uint64_t)nodes, 5u);
        hipLaunchKernelGGL(fill_rec, dim3(4096), dim3(256), 0, 0, rec, nnz, deg);
        CK(hipDeviceSynchronize());
        const int iters = 64;
        auto report = [&](const char *shape, int k, double ms) {
            const uint64_t hops = lanes * (uint64_t)iters * k;
            printf("%s,%llu,%d,%llu,%.4f,%.2f,,\n", shape, (unsigned long long)(nnz * 4 >> 20), k, (unsigned long long)hops, ms,
                   hops / ms / 1e6);
            fflush(stdout);
        };
        for (int pass = 0; pass < 1; ++pass) {
            double ms;
            ms = time_ms([&] { hipLaunchKernelGGL((walk2<1>), dim3(grid), dim3(kThreads), kLdsPad, 0, rowptr, nbr, nodes, iters, out); });
            report("walk2(hops)", 1, ms);
            ms = time_ms([&] { hipLaunchKernelGGL((walk2<2>), dim3(grid), dim3(kThreads), kLdsPad, 0, rowptr, nbr, nodes, iters, out); });
            report("walk2(hops)", 2, ms);
            ms = time_ms([&] { hipLaunchKernelGGL((walk2<4>), dim3(grid), dim3(kThreads), kLdsPad, 0, rowptr, nbr, nodes, iters, out); });
            report("walk2(hops)", 4, ms);
            ms = time_ms([&] { hipLaunchKernelGGL((rec8<1>), dim3(grid), dim3(kThreads), kLdsPad, 0, rec, nnz, deg, iters, out); });
            report("rec8(hops)", 1, ms);
            ms = time_ms([&] { hipLaunchKernelGGL((rec8<2>), dim3(grid), dim3(kThreads), kLdsPad, 0, rec, nnz, deg, iters, out); });
            report("rec8(hops)", 2, ms);
            ms = time_ms([&] { hipLaunchKernelGGL((rec8<4>), dim3(grid), dim3(kThreads), kLdsPad, 0, rec, nnz, deg, iters, out); });
            report("rec8(hops)", 4, ms);
        }
        CK(hipFree(rowptr));
        CK(hipFree(nbr));
        CK(hipFree(rec));
    }
    CK(hipFree(out));
    return 0;
}
