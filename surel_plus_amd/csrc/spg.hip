// spg.hip -- SpG build: sort every sampled node set by node id (gfx950).
//
// Replaces the scipy COO->CSR conversion of subg_matrix (reference sampler/random_walks.py:79-80), which
// is a single-threaded global sort.  Here the sets are already grouped by row (one row per root), so the
// "conversion" is a segmented sort, one wave64 workgroup per row, rows dealt to XCDs in contiguous ranges.
//
// Rows hold at most M*m+1 members (<= 801 in every reference configuration) with DISTINCT ids, so the sort
// is a rank sort: the row's ids sit in LDS, every lane keeps its own <= 16 members in registers and counts
// how many ids of the row are smaller (broadcast ds_read_b128, 4 ids per LDS instruction, no bank
// conflicts, no barriers inside the loop); the count IS the output position.  A bitonic network over the
// same data costs log^2(P)/2 full read+write passes over LDS and was LDS-bandwidth bound (1.16 ms vs
// the walk kernel's 1.19 ms on the cit2-like batch).  Rows longer than 1024 fall back to the bitonic path.
#include "common.hpp"

namespace subgacc {

constexpr int kSpgThreads = 64;
constexpr int kRankMaxLen = 16 * kSpgThreads;

// E = members per lane (compile time so that the own ids / counters stay in registers)
template <int E>
__device__ __forceinline__ void rank_sort_row(const int32_t *__restrict__ ids_l, int ns, int ns4, int lane,
                                              const int32_t *__restrict__ sf_row, unsigned long long *out_l) {
    int32_t x[E];
    int32_t cnt[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int e = lane + u * kSpgThreads;
        x[u] = e < ns ? ids_l[e] : 0x7FFFFFFF;
        cnt[u] = 0;
    }
    for (int j = 0; j < ns4; j += 4) {
        const int4 kk = *reinterpret_cast<const int4 *>(ids_l + j);   // same address in every lane: broadcast
#pragma unroll
        for (int u = 0; u < E; ++u)
            cnt[u] += (kk.x < x[u]) + (kk.y < x[u]) + (kk.z < x[u]) + (kk.w < x[u]);
    }
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int e = lane + u * kSpgThreads;
        if (e < ns) out_l[cnt[u]] = ((unsigned long long)(uint32_t)x[u] << 32) | (uint32_t)(sf_row[e] + 1);
    }
}

__global__ __launch_bounds__(kSpgThreads) void spg_rank_kernel(const int64_t *__restrict__ row_off, int64_t n,
                                                                const int32_t *__restrict__ ids,
                                                                const int32_t *__restrict__ sf, int32_t cap,
                                                                int32_t *__restrict__ out_indices,
                                                                int32_t *__restrict__ out_data, int32_t *flags) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    unsigned long long *out_l = (unsigned long long *)lds_raw;   // [cap]   sorted (id, SFptr+1)
    int32_t *ids_l = (int32_t *)(out_l + cap);                   // [cap+4] the row's ids, padded with INT_MAX
    const int64_t i = xcd_item(blockIdx.x, gridDim.x);
    if (i >= n) return;
    const int64_t beg = row_off[i];
    const int64_t ns64 = row_off[i + 1] - beg;
    if (ns64 > cap) {  // the caller under-stated max_len: refuse the row, never overrun LDS
        if (threadIdx.x == 0) atomicOr(&flags[3], 1);
        return;
    }
    const int ns = (int)ns64, lane = threadIdx.x;
    const int ns4 = (ns + 3) & ~3;
    for (int r = lane; r < ns4; r += kSpgThreads) ids_l[r] = r < ns ? ids[beg + r] : 0x7FFFFFFF;
    __syncthreads();
    const int32_t *sf_row = sf + beg;
    switch ((ns + kSpgThreads - 1) / kSpgThreads) {
        case 0: break;
        case 1: rank_sort_row<1>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 2: rank_sort_row<2>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 3: rank_sort_row<3>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 4: rank_sort_row<4>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 5: rank_sort_row<5>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 6: rank_sort_row<6>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 7: rank_sort_row<7>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 8: rank_sort_row<8>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 9: case 10: rank_sort_row<10>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        case 11: case 12: case 13: rank_sort_row<13>(ids_l, ns, ns4, lane, sf_row, out_l); break;
        default: rank_sort_row<16>(ids_l, ns, ns4, lane, sf_row, out_l); break;
    }
    __syncthreads();
    for (int r = lane; r < ns; r += kSpgThreads) {
        const unsigned long long v = out_l[r];
        out_indices[beg + r] = (int32_t)(v >> 32);
        out_data[beg + r] = (int32_t)(uint32_t)v;
    }
}

__global__ __launch_bounds__(kSpgThreads) void spg_build_kernel(const int64_t *__restrict__ row_off, int64_t n,
                                                                 const int32_t *__restrict__ ids,
                                                                 const int32_t *__restrict__ sf, int32_t max_pow2,
                                                                 int32_t *__restrict__ out_indices,
                                                                 int32_t *__restrict__ out_data, int32_t *flags) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    unsigned long long *buf = (unsigned long long *)lds_raw;
    const int64_t i = xcd_item(blockIdx.x, gridDim.x);
    if (i >= n) return;
    const int64_t beg = row_off[i];
    const int64_t ns64 = row_off[i + 1] - beg;
    if (ns64 > max_pow2) {  // the caller under-stated max_len: refuse the row, never overrun LDS
        if (threadIdx.x == 0) atomicOr(&flags[3], 1);
        return;
    }
    const int ns = (int)ns64;
    int P = 1;
    while (P < ns) P <<= 1;
    const int tid = threadIdx.x;
    for (int r = tid; r < P; r += kSpgThreads)
        buf[r] = r < ns ? (((unsigned long long)(uint32_t)ids[beg + r] << 32) | (uint32_t)(sf[beg + r] + 1)) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += kSpgThreads) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const unsigned long long a = buf[lo], b = buf[hi];
                const bool ascending = (lo & k) == 0;
                if ((a > b) == ascending) {
                    buf[lo] = b;
                    buf[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int r = tid; r < ns; r += kSpgThreads) {
        const unsigned long long v = buf[r];
        out_indices[beg + r] = (int32_t)(v >> 32);
        out_data[beg + r] = (int32_t)(uint32_t)v;
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" int subgacc_spg_build(const int64_t *row_off, int64_t n, const int32_t *ids, const int32_t *sf,
                                 int32_t max_len, int32_t *out_indices, int32_t *out_data, int32_t *flags,
                                 void *stream) {
    SG_REQUIRE(n >= 0 && max_len >= 0 && flags, SUBGACC_ERR_BADARG, "spg_build: bad arguments");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(row_off && ids && sf && out_indices && out_data, SUBGACC_ERR_BADARG, "spg_build: null argument");
    const int64_t grid = xcd_grid(n);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "spg_build: too many rows in one call");
    if (max_len <= kRankMaxLen) {
        const int cap = ((max_len > 0 ? max_len : 1) + 3) & ~3;
        const size_t lds_rank = (size_t)cap * 8 + (size_t)(cap + 4) * 4;
        hipLaunchKernelGGL(spg_rank_kernel, dim3((unsigned)grid), dim3(kSpgThreads), lds_rank, (hipStream_t)stream,
                           row_off, n, ids, sf, cap, out_indices, out_data, flags);
        SG_LAUNCH_CHECK();
        return SUBGACC_OK;
    }
    int P = 1;
    while (P < max_len) P <<= 1;
    const size_t lds = (size_t)P * 8;
    SG_REQUIRE(lds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS, "spg_build: rows of %d members do not fit LDS", max_len);
    if (lds > 64 * 1024)
        SG_CHECK_HIP(hipFuncSetAttribute((const void *)spg_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spg_build_kernel, dim3((unsigned)grid), dim3(kSpgThreads), lds, (hipStream_t)stream, row_off, n,
                       ids, sf, P, out_indices, out_data, flags);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
