// worklist.hip -- the rows of a batch in ascending order of their root's id, as a work list for the walk kernel (gfx950).
//
// The fused-row walk kernel deals consecutive work items to the same XCD (xcd_item) and walks them at about the same time.  When
// the items are the rows of a batch in BATCH order, neighbours in time are unrelated roots; in order of root id they are
// neighbours in the graph wherever ids have locality (communities of consecutive ids: citation and co-author graphs), and
// repeated endpoints (hubs turn up many times in a batch of edges) stand next to each other: the same hop records are asked for
// by the same L2 within microseconds.  Measured (profiles/r05v_locality_two_level_sort.log): walk kernel 0.75 -> 0.66 ms on
// the structureless cit2-like graph (L2 hits 8.5 M -> 11.1 M per launch), 0.73 -> 0.66 ms and 43 % fewer missed lines on the
// same graph with communities.  The rows themselves stay where the batch has them (row i = endpoint i): only the ORDER in which
// the kernel takes them changes, so nothing observable does.
//
// An exact sort is not needed -- 1,024 buckets of consecutive ids, any order inside a bucket -- so this is ONE radix pass in two
// small launches over 1,024-row tiles.  `count`: every row takes its arrival number inside its bucket from an LDS histogram, the
// block takes a base for each of its non-empty buckets from the global totals (one atomic per block and bucket), and the row's
// place inside its bucket = base + arrival number goes to the workspace.  `place`: every block scans the 1,024 totals (4 KB, no
// third launch) and writes row i at offset[bucket] + place.  The order inside a bucket is the order of arrival of the atomics: it
// differs from run to run, and nothing observable depends on it (the rows themselves stay where the batch has them).
// The totals are zeroed again by the last block of `place` to finish (a completion counter), so a call leaves the workspace as it
// found it: zeroed ONCE by its owner before the first call.  (The first form of this file kept per-block histograms in HBM and made
// every block of the second launch sum 32 of them per bucket: 7 + 18 us for 131,072 rows; this one 4 + 5.)
#include "common.hpp"
#include "blockscan.hpp"

namespace subgacc {

constexpr int kWlThreads = 256, kWlItems = 4, kWlTile = kWlThreads * kWlItems;     // 1,024 rows per block
constexpr int kWlBuckets = 1024;
constexpr int kWlHead = 16;              // workspace: [0] completion counter | 16 B | totals [1,024] | place [n]

__device__ __forceinline__ int wl_bucket(int32_t root, int shift) {
    const uint32_t b = (uint32_t)root >> shift;          // (a root outside the graph -- flagged by the walk kernel -- lands in the last bucket)
    return (int)(b < (uint32_t)kWlBuckets ? b : (uint32_t)kWlBuckets - 1u);
}

__global__ __launch_bounds__(kWlThreads) void worklist_count_kernel(const int32_t *__restrict__ roots, int64_t n, int shift,
                                                                    int32_t *__restrict__ totals, int32_t *__restrict__ place) {
    __shared__ int32_t h[kWlBuckets];
    for (int b = threadIdx.x; b < kWlBuckets; b += kWlThreads) h[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kWlTile;
    int32_t bk[kWlItems], li[kWlItems];
#pragma unroll
    for (int k = 0; k < kWlItems; ++k) {
        const int64_t i = base + (int64_t)k * kWlThreads + threadIdx.x;
        bk[k] = -1, li[k] = 0;
        if (i < n) {
            const int32_t r = roots[i];
            if (r != SUBGACC_NO_ROOT) {      // a repeated endpoint's empty row is not listed
                bk[k] = wl_bucket(r, shift);
                li[k] = atomicAdd(&h[bk[k]], 1);
            }
        }
    }
    __syncthreads();
    {   // the count becomes this block's base inside the bucket (all four atomics of a lane in flight together: empty buckets add 0)
        int32_t c[kWlBuckets / kWlThreads], g[kWlBuckets / kWlThreads];
#pragma unroll
        for (int q = 0; q < kWlBuckets / kWlThreads; ++q) c[q] = h[q * kWlThreads + threadIdx.x];
#pragma unroll
        for (int q = 0; q < kWlBuckets / kWlThreads; ++q) g[q] = atomicAdd(&totals[q * kWlThreads + threadIdx.x], c[q]);
#pragma unroll
        for (int q = 0; q < kWlBuckets / kWlThreads; ++q) h[q * kWlThreads + threadIdx.x] = g[q];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWlItems; ++k) {
        const int64_t i = base + (int64_t)k * kWlThreads + threadIdx.x;
        if (bk[k] >= 0) place[i] = h[bk[k]] + li[k];
    }
}

__global__ __launch_bounds__(kWlThreads) void worklist_place_kernel(const int32_t *__restrict__ roots, int64_t n, int shift,
                                                                    int32_t *__restrict__ totals, const int32_t *__restrict__ place,
                                                                    int32_t *__restrict__ done, int32_t *__restrict__ worklist,
                                                                    int64_t *__restrict__ n_work) {
    __shared__ int32_t off[kWlBuckets];
    __shared__ int32_t last;
    constexpr int PER = kWlBuckets / kWlThreads;       // consecutive buckets per lane
    const int64_t base = (int64_t)blockIdx.x * kWlTile;
    int32_t rt[kWlItems], pl[kWlItems];                // this block's rows: asked for before the scan, needed after it
#pragma unroll
    for (int k = 0; k < kWlItems; ++k) {
        const int64_t i = base + (int64_t)k * kWlThreads + threadIdx.x;
        rt[k] = i < n ? roots[i] : SUBGACC_NO_ROOT;
        pl[k] = i < n ? place[i] : 0;
    }
    int32_t tot[PER];
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        tot[k] = totals[threadIdx.x * PER + k];
        s += tot[k];
    }
    int32_t all;
    int32_t run = block_exclusive_scan<int32_t>(s, &all);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        off[threadIdx.x * PER + k] = run;
        run += tot[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_work = all;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWlItems; ++k) {
        const int64_t i = base + (int64_t)k * kWlThreads + threadIdx.x;
        if (rt[k] != SUBGACC_NO_ROOT) worklist[off[wl_bucket(rt[k], shift)] + pl[k]] = (int32_t)i;
    }
    // the last block to get here zeroes the totals for the next call (every block has READ them by then -- the scan above needed
    // their values -- and that is all the order this needs: no fence, which on this chip is an L2 write-back per block)
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(done, 1) == (int32_t)gridDim.x - 1;
    __syncthreads();
    if (last) {
        for (int b = threadIdx.x; b < kWlBuckets; b += kWlThreads) totals[b] = 0;
        if (threadIdx.x == 0) *done = 0;
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" size_t subgacc_worklist_workspace_bytes(int64_t n) {
    if (n < 0) n = 0;
    return (size_t)kWlHead + (size_t)kWlBuckets * 4 + (size_t)(n > 0 ? n : 1) * 4;
}

extern "C" int subgacc_worklist_by_root(const int32_t *roots, int64_t n, int64_t num_nodes, int32_t *worklist, int64_t *n_work,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    SG_REQUIRE(n >= 0 && n < (1ll << 31) && num_nodes >= 0 && n_work, SUBGACC_ERR_BADARG, "worklist_by_root: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        SG_CHECK_HIP(hipMemsetAsync(n_work, 0, 8, s));
        return SUBGACC_OK;
    }
    SG_REQUIRE(roots && worklist, SUBGACC_ERR_BADARG, "worklist_by_root: null argument");
    SG_REQUIRE(workspace && workspace_bytes >= subgacc_worklist_workspace_bytes(n), SUBGACC_ERR_WORKSPACE, "worklist_by_root: workspace too small");
    const int nblk = (int)ceil_div(n, kWlTile);
    int shift = 0;
    while (((num_nodes > 0 ? num_nodes - 1 : 0) >> shift) >= kWlBuckets) ++shift;
    int32_t *done = (int32_t *)workspace;
    int32_t *totals = (int32_t *)((char *)workspace + kWlHead);
    int32_t *place = totals + kWlBuckets;
    hipLaunchKernelGGL(worklist_count_kernel, dim3(nblk), dim3(kWlThreads), 0, s, roots, n, shift, totals, place);
    hipLaunchKernelGGL(worklist_place_kernel, dim3(nblk), dim3(kWlThreads), 0, s, roots, n, shift, totals, (const int32_t *)place, done,
                       worklist, n_work);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
