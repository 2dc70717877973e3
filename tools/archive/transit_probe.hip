// transit_probe.hip -- dev-only probe (not product, not a test).  VERDICT r3 "Next #1": would a TRANSIT-parallel last hop
// (walkers binned by the region of the adjacency array they are about to read, each region consumed once out of one XCD's L2,
// results carried back to root-major order) beat the root-parallel last hop of walk_rows_kernel, which fetches one 128-byte
// line per 4-byte read?  Stand-alone program, no torch:
//
//     hipcc -O3 --offload-arch=gfx950 tools/transit_probe.hip -o tools/build/transit_probe && tools/build/transit_probe
//
// The population is the cit2 batch of bench.py: n = 131,072 roots x 200 walkers = 26,214,400 walkers, every one about to
// read ONE uniformly random entry of a 62,994,730-entry int32 adjacency array (a walk in stationarity lands on every
// adjacency entry with the same probability -- DESIGN.md 4.1).  Stages, each timed with HIP events (mean of REPS):
//   direct       res[i] = adj[t[i]]                                     the root-parallel form: one random line per walker
//   fwd          (i, t[i]) -> bins by region of t (tile-sorted in LDS, chunk per (tile, bin) reserved with one atomic)
//   gather       per bin, from workgroups of ONE XCD: id = adj[t] (the region's lines are fetched once into that L2),
//                (i, id) -> bins by root block (again tile-sorted in LDS)
//   back         per root block: res[i] = id, scattered inside a window that is L2-resident
//   gather+scatter   the same gather writing res[i] directly (no back partition): 26 M scattered 4-byte stores
// res is verified against the direct form.  PMC passes: `rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum -- transit_probe` with
// TRANSIT_REPS=1.  What it has to beat: the marginal cost of the last hop INSIDE walk_rows_kernel, 0.185 ms on the cit2 batch
// (profiles/r08_exp_inline_last_hop_emulation.log: 0.6626 -> 0.4772 ms with the last hop's line fetch removed).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e__ = (x);                                                      \
        if (e__ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef unsigned long long u64;
constexpr int NT = 512;              // lanes per workgroup
constexpr int EPT = 16;              // elements per lane and tile
constexpr int TILE = NT * EPT;       // 8,192 elements = 64 KB of LDS staging
constexpr int MAXB = 1024;           // bins per partition

__device__ __forceinline__ u64 splitmix(u64 h) {
    h += 0x9e3779b97f4a7c15ull;
    h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ull;
    h = (h ^ (h >> 27)) * 0x94d049bb133111ebull;
    return h ^ (h >> 31);
}
__global__ void fill_targets(uint32_t *t, u64 n, u64 nnz, u64 salt) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        t[i] = (uint32_t)(((unsigned __int128)splitmix(i + salt) * nnz) >> 64);
}
__global__ void fill_adj(uint32_t *a, u64 nnz, uint32_t nodes) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (u64)gridDim.x * blockDim.x)
        a[i] = (uint32_t)(((unsigned __int128)splitmix(i * 3 + 1) * nodes) >> 64);
}
__global__ __launch_bounds__(256) void direct_gather(const uint32_t *__restrict__ adj, const uint32_t *__restrict__ t, uint32_t *res, u64 n) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) res[i] = adj[t[i]];
}
__global__ void count_mismatch(const uint32_t *a, const uint32_t *b, u64 n, u64 *bad) {
    u64 c = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(bad, c);
}

// One tile of up to TILE (payload, bin) pairs, EPT per lane, leaves the workgroup grouped by bin: LDS histogram whose returning
// atomic is the arrival order, scan, ONE global atomic per (tile, non-empty bin) reserves the chunk, the tile is staged in LDS in
// bin order and written with consecutive lanes on consecutive words.  `binof(payload)` recomputes the bin of a staged element.
template <typename BinOf>
__device__ __forceinline__ void tile_partition(const u64 (&pay)[EPT], const bool (&ok)[EPT], int nb, u64 *out, u64 cap, uint32_t *cursor,
                                               uint32_t *overflow, BinOf binof, u64 *stage, uint32_t *hist, uint32_t *start, uint32_t *gbase) {
    const int tid = threadIdx.x;
    for (int b = tid; b < nb; b += NT) hist[b] = 0;
    __syncthreads();
    uint32_t arr[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) arr[e] = ok[e] ? atomicAdd(&hist[binof(pay[e])], 1u) : 0u;
    __syncthreads();
    // exclusive scan over nb <= MAXB = 2 * NT bins: two bins per lane, wave scan, wave totals through LDS
    __shared__ uint32_t wsum[NT / 64];
    const uint32_t c0 = 2 * tid < nb ? hist[2 * tid] : 0u, c1 = 2 * tid + 1 < nb ? hist[2 * tid + 1] : 0u;
    uint32_t inc = c0 + c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(inc, d, 64);
        if ((tid & 63) >= d) inc += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
    const uint32_t ex = base + inc - (c0 + c1);
    if (2 * tid < nb) {
        start[2 * tid] = ex;
        gbase[2 * tid] = c0 ? atomicAdd(&cursor[2 * tid], c0) : 0u;
    }
    if (2 * tid + 1 < nb) {
        start[2 * tid + 1] = ex + c0;
        gbase[2 * tid + 1] = c1 ? atomicAdd(&cursor[2 * tid + 1], c1) : 0u;
    }
    uint32_t total = 0;
    for (int w = 0; w < NT / 64; ++w) total += wsum[w];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        if (ok[e]) stage[start[binof(pay[e])] + arr[e]] = pay[e];
    __syncthreads();
    for (uint32_t j = tid; j < total; j += NT) {
        const u64 p = stage[j];
        const uint32_t b = binof(p);
        const u64 at = (u64)gbase[b] + (j - start[b]);
        if (at < cap) out[(u64)b * cap + at] = p;
        else atomicOr(overflow, 1u);
    }
    __syncthreads();
}

// fwd: walker i with target t[i] -> payload (i << 32 | t), bin = t >> rsh
__global__ __launch_bounds__(NT) void fwd_partition(const uint32_t *__restrict__ t, u64 n, int rsh, int nb, u64 *bins, u64 cap, uint32_t *cursor, uint32_t *overflow) {
    extern __shared__ __align__(16) unsigned char lds[];
    u64 *stage = (u64 *)lds;
    uint32_t *hist = (uint32_t *)(stage + TILE), *start = hist + MAXB, *gbase = start + MAXB;
    const u64 base = (u64)blockIdx.x * TILE;
    u64 pay[EPT];
    bool ok[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const u64 i = base + (u64)e * NT + threadIdx.x;
        ok[e] = i < n;
        pay[e] = ok[e] ? ((i << 32) | t[i]) : 0ull;
    }
    tile_partition(pay, ok, nb, bins, cap, cursor, overflow, [=](u64 p) { return (uint32_t)p >> rsh; }, stage, hist, start, gbase);
}

// block -> (bin, slice): blocks are dealt round-robin to the 8 XCDs, so block b runs on XCD b % 8.  The blocks of one XCD walk
// its bins (bin % 8 == xcd) one after the other, S slices per bin: the workgroups that are resident on an XCD at one time
// work on a handful of consecutive regions, and a region is only ever read through ONE L2.
__device__ __forceinline__ bool bin_slice(int nb, int S, int &bin, int &slice) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    bin = (k / S) * 8 + xcd;
    slice = k % S;
    return bin < nb;
}

// gather: per region bin, (i, t) -> id = adj[t] -> payload (i << 32 | id), bin = i >> bsh (root block)
template <bool DIRECT>
__global__ __launch_bounds__(NT) void gather_bins(const uint32_t *__restrict__ adj, const u64 *__restrict__ bins, u64 cap, const uint32_t *__restrict__ count, int nb, int S,
                                                  int bsh, int nb2, u64 *bins2, u64 cap2, uint32_t *cursor2, uint32_t *overflow, uint32_t *res) {
    extern __shared__ __align__(16) unsigned char lds[];
    u64 *stage = (u64 *)lds;
    uint32_t *hist = (uint32_t *)(stage + TILE), *start = hist + MAXB, *gbase = start + MAXB;
    int bin, slice;
    if (!bin_slice(nb, S, bin, slice)) return;
    const uint32_t cnt = min(count[bin], (uint32_t)cap);
    const uint32_t per = ((cnt + S - 1) / S + 1) & ~1u;
    const uint32_t lo = min(cnt, (uint32_t)slice * per), hi = min(cnt, lo + per);
    const u64 *src = bins + (u64)bin * cap;
    for (uint32_t tb = lo; tb < hi; tb += TILE) {      // (workgroup-uniform bounds)
        u64 pay[EPT];
        bool ok[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const uint32_t j = tb + (uint32_t)e * NT + threadIdx.x;
            ok[e] = j < hi;
            pay[e] = ok[e] ? src[j] : 0ull;
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const uint32_t id = ok[e] ? adj[(uint32_t)pay[e]] : 0u;
            if (DIRECT) {
                if (ok[e]) res[pay[e] >> 32] = id;
            } else {
                pay[e] = (pay[e] & 0xFFFFFFFF00000000ull) | id;
            }
        }
        if (!DIRECT)
            tile_partition(pay, ok, nb2, bins2, cap2, cursor2, overflow, [=](u64 p) { return (uint32_t)(p >> 32) >> bsh; }, stage, hist, start, gbase);
    }
}

// back: per root block, res[i] = id -- scattered inside a window of 2^bsh words
__global__ __launch_bounds__(NT) void scatter_back(const u64 *__restrict__ bins2, u64 cap2, const uint32_t *__restrict__ count2, int nb2, int S, uint32_t *res) {
    int bin, slice;
    if (!bin_slice(nb2, S, bin, slice)) return;
    const uint32_t cnt = min(count2[bin], (uint32_t)cap2);
    const uint32_t per = (cnt + S - 1) / S;
    const uint32_t lo = min(cnt, (uint32_t)slice * per), hi = min(cnt, lo + per);
    const u64 *src = bins2 + (u64)bin * cap2;
    for (uint32_t j = lo + threadIdx.x; j < hi; j += NT) {
        const u64 p = src[j];
        res[p >> 32] = (uint32_t)p;
    }
}

static int g_reps = 5;
template <typename F>
static double time_ms(F launch, hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();   // warm
    CK(hipStreamSynchronize(s));
    double tot = 0;
    for (int r = 0; r < g_reps; ++r) {
        CK(hipEventRecord(a, s));
        launch();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        tot += ms;
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return tot / g_reps;
}

int main(int argc, char **argv) {
    if (getenv("TRANSIT_REPS")) g_reps = atoi(getenv("TRANSIT_REPS"));
    const u64 n = getenv("TRANSIT_N") ? strtoull(getenv("TRANSIT_N"), 0, 10) : 131072ull * 200ull;
    const u64 nnz = getenv("TRANSIT_NNZ") ? strtoull(getenv("TRANSIT_NNZ"), 0, 10) : 62994730ull;
    const uint32_t nodes = 2927963u;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    uint32_t *adj, *t, *res, *ref, *cursor, *cursor2, *overflow;
    u64 *bad;
    CK(hipMalloc(&adj, nnz * 4));
    CK(hipMalloc(&t, n * 4));
    CK(hipMalloc(&res, n * 4));
    CK(hipMalloc(&ref, n * 4));
    CK(hipMalloc(&cursor, MAXB * 4));
    CK(hipMalloc(&cursor2, MAXB * 4));
    CK(hipMalloc(&overflow, 4));
    CK(hipMalloc(&bad, 8));
    CK(hipMemset(overflow, 0, 4));
    fill_adj<<<4096, 256, 0, s>>>(adj, nnz, nodes);
    fill_targets<<<4096, 256, 0, s>>>(t, n, nnz, 12345);
    CK(hipStreamSynchronize(s));
    const size_t lds = (size_t)TILE * 8 + 3 * MAXB * 4;
    CK(hipFuncSetAttribute((const void *)fwd_partition, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)gather_bins<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)gather_bins<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    printf("stage,region_KB,bins,slices,back_window_KB,back_bins,back_slices,ms,GB_streamed,TBps,note\n");
    const double d_ms = time_ms([&] { direct_gather<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(adj, t, ref, n); }, s);
    printf("direct,,,,,,,%.4f,%.3f,%.2f,%.1f G random reads/s\n", d_ms, n * 8 / 1e9, n * 8 / 1e9 / d_ms, n / d_ms / 1e6);
    fflush(stdout);

    // region sizes (entries = 2^rsh): 2^16 = 256 KB ... 2^18 = 1 MB; the bins of both partitions must stay <= MAXB
    const int only_rsh = getenv("TRANSIT_RSH") ? atoi(getenv("TRANSIT_RSH")) : 0, only_bsh = getenv("TRANSIT_BSH") ? atoi(getenv("TRANSIT_BSH")) : 0;
    const int only_s = getenv("TRANSIT_S") ? atoi(getenv("TRANSIT_S")) : 0;
    for (int rsh = 16; rsh <= 18; ++rsh) {
        if (only_rsh && rsh != only_rsh) continue;
        const int nb = (int)((nnz + (1ull << rsh) - 1) >> rsh);
        if (nb > MAXB) {
            printf("skip,%d,%d,,,,,,,,more than %d bins\n", (4 << rsh) >> 10, nb, MAXB);
            continue;
        }
        const u64 cap = (n / nb) + (n / nb) / 4 + 4096;
        for (int bsh = 16; bsh <= 17; ++bsh) {         // back window: 2^bsh results = 256 / 512 KB
            const int nb2 = (int)((n + (1ull << bsh) - 1) >> bsh);
            if (nb2 > MAXB || (only_bsh && bsh != only_bsh)) continue;
            const u64 cap2 = (n / nb2) + (n / nb2) / 4 + 4096;
            u64 *bins, *bins2;
            CK(hipMalloc(&bins, (u64)nb * cap * 8));
            CK(hipMalloc(&bins2, (u64)nb2 * cap2 * 8));
            for (int S : {4, 8, 16}) {
                if (only_s && S != only_s) continue;
                const int S2 = 16;
                const unsigned gtiles = (unsigned)((n + TILE - 1) / TILE);
                const unsigned gg = (unsigned)(((nb + 7) / 8) * S * 8), gb = (unsigned)(((nb2 + 7) / 8) * S2 * 8);
                auto fwd = [&] {
                    CK(hipMemsetAsync(cursor, 0, MAXB * 4, s));
                    fwd_partition<<<gtiles, NT, lds, s>>>(t, n, rsh, nb, bins, cap, cursor, overflow);
                };
                auto gat = [&] {
                    CK(hipMemsetAsync(cursor2, 0, MAXB * 4, s));
                    gather_bins<false><<<gg, NT, lds, s>>>(adj, bins, cap, cursor, nb, S, bsh, nb2, bins2, cap2, cursor2, overflow, res);
                };
                auto bck = [&] { scatter_back<<<gb, NT, 0, s>>>(bins2, cap2, cursor2, nb2, S2, res); };
                auto gds = [&] { gather_bins<true><<<gg, NT, lds, s>>>(adj, bins, cap, cursor, nb, S, bsh, nb2, bins2, cap2, cursor2, overflow, res); };
                CK(hipMemsetAsync(res, 0xFF, n * 4, s));
                const double f_ms = time_ms(fwd, s);
                const double g_ms = time_ms(gat, s);
                const double b_ms = time_ms(bck, s);
                CK(hipMemsetAsync(bad, 0, 8, s));
                count_mismatch<<<2048, 256, 0, s>>>(res, ref, n, bad);
                u64 hbad = 0;
                uint32_t hover = 0;
                CK(hipMemcpyAsync(&hbad, bad, 8, hipMemcpyDeviceToHost, s));
                CK(hipMemcpyAsync(&hover, overflow, 4, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                CK(hipMemsetAsync(res, 0xFF, n * 4, s));
                const double gd_ms = time_ms(gds, s);
                CK(hipMemsetAsync(bad, 0, 8, s));
                count_mismatch<<<2048, 256, 0, s>>>(res, ref, n, bad);
                u64 hbad2 = 0;
                CK(hipMemcpyAsync(&hbad2, bad, 8, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                const double fGB = n * 12 / 1e9, gGB = (n * 16 + nnz * 4) / 1e9, bGB = n * 12 / 1e9;
                printf("fwd,%d,%d,,,,,%.4f,%.3f,%.2f,\n", (4 << rsh) >> 10, nb, f_ms, fGB, fGB / f_ms);
                printf("gather,%d,%d,%d,%d,%d,,%.4f,%.3f,%.2f,\n", (4 << rsh) >> 10, nb, S, (4 << bsh) >> 10, nb2, g_ms, gGB, gGB / g_ms);
                printf("back,,,,%d,%d,%d,%.4f,%.3f,%.2f,\n", (4 << bsh) >> 10, nb2, S2, b_ms, bGB, bGB / b_ms);
                printf("fwd+gather+back,%d,%d,%d,%d,%d,%d,%.4f,%.3f,%.2f,mismatches %llu overflow %u (direct %.4f ms)\n", (4 << rsh) >> 10, nb, S, (4 << bsh) >> 10, nb2, S2,
                       f_ms + g_ms + b_ms, fGB + gGB + bGB, (fGB + gGB + bGB) / (f_ms + g_ms + b_ms), hbad, hover, d_ms);
                printf("fwd+gather_scatter,%d,%d,%d,,,,%.4f,,,gather writing res[i] itself %.4f ms; mismatches %llu\n", (4 << rsh) >> 10, nb, S, f_ms + gd_ms, gd_ms, hbad2);
                fflush(stdout);
            }
            CK(hipFree(bins));
            CK(hipFree(bins2));
        }
    }
    return 0;
}
