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
// launches: per-block histograms in LDS (bucket-major in HBM), then every block sums the few thousand counters in front of
// its own (no third launch for the scan) and scatters its rows.
#include "common.hpp"
#include "blockscan.hpp"

namespace subgacc {

constexpr int kWlThreads = 256, kWlItems = 16, kWlTile = kWlThreads * kWlItems;     // 4,096 rows per block
constexpr int kWlBuckets = 1024;

__device__ __forceinline__ int wl_bucket(int32_t root, int shift) {
    const uint32_t b = (uint32_t)root >> shift;          // (a root outside the graph -- flagged by the walk kernel -- lands in the last bucket)
    return (int)(b < (uint32_t)kWlBuckets ? b : (uint32_t)kWlBuckets - 1u);
}

__global__ __launch_bounds__(kWlThreads) void worklist_hist_kernel(const int32_t *__restrict__ roots, int64_t n, int shift,
                                                                   int32_t *__restrict__ hist, int nblk) {
    __shared__ int32_t h[kWlBuckets];
    for (int b = threadIdx.x; b < kWlBuckets; b += kWlThreads) h[b] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kWlTile;
#pragma unroll
    for (int k = 0; k < kWlItems; ++k) {
        const int64_t i = base + (int64_t)k * kWlThreads + threadIdx.x;
        if (i < n) {
            const int32_t r = roots[i];
            if (r != SUBGACC_NO_ROOT) atomicAdd(&h[wl_bucket(r, shift)], 1);      // a repeated endpoint's empty row is not listed
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kWlBuckets; b += kWlThreads) hist[(int64_t)b * nblk + blockIdx.x] = h[b];
}

__global__ __launch_bounds__(kWlThreads) void worklist_scatter_kernel(const int32_t *__restrict__ roots, int64_t n, int shift,
                                                                      const int32_t *__restrict__ hist, int nblk,
                                                                      int32_t *__restrict__ worklist, int64_t *__restrict__ n_work) {
    __shared__ int32_t cur[kWlBuckets];
    constexpr int PER = kWlBuckets / kWlThreads;       // consecutive buckets per lane
    int32_t tot[PER], before[PER];
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = threadIdx.x * PER + k;
        int32_t t = 0, bf = 0;
        for (int q = 0; q < nblk; ++q) {
            const int32_t c = hist[(int64_t)b * nblk + q];
            t += c;
            bf += q < (int)blockIdx.x ? c : 0;
        }
        tot[k] = t, before[k] = bf;
        s += t;
    }
    int32_t all;
    int32_t run = block_exclusive_scan<int32_t>(s, &all);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        cur[threadIdx.x * PER + k] = run + before[k];      // where this block's rows of the bucket begin
        run += tot[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_work = all;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kWlTile;
#pragma unroll
    for (int k = 0; k < kWlItems; ++k) {
        const int64_t i = base + (int64_t)k * kWlThreads + threadIdx.x;
        if (i < n) {
            const int32_t r = roots[i];
            if (r != SUBGACC_NO_ROOT) worklist[atomicAdd(&cur[wl_bucket(r, shift)], 1)] = (int32_t)i;
        }
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" size_t subgacc_worklist_workspace_bytes(int64_t n) {
    if (n < 0) n = 0;
    return (size_t)ceil_div(n > 0 ? n : 1, kWlTile) * kWlBuckets * 4;
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
    SG_REQUIRE(nblk <= 4096, SUBGACC_ERR_BADARG, "worklist_by_root: %lld rows in one call (at most %d)", (long long)n, 4096 * kWlTile);
    int shift = 0;
    while (((num_nodes > 0 ? num_nodes - 1 : 0) >> shift) >= kWlBuckets) ++shift;
    hipLaunchKernelGGL(worklist_hist_kernel, dim3(nblk), dim3(kWlThreads), 0, s, roots, n, shift, (int32_t *)workspace, nblk);
    hipLaunchKernelGGL(worklist_scatter_kernel, dim3(nblk), dim3(kWlThreads), 0, s, roots, n, shift, (const int32_t *)workspace, nblk,
                       worklist, n_work);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
