// spg.hip -- SpG build: sort every sampled node set by node id (gfx950).
//
// Replaces the scipy COO->CSR conversion of subg_matrix (reference sampler/random_walks.py:79-80), which
// is a single-threaded global sort.  Here the sets are already grouped by row (one row per root), so the
// "conversion" is a segmented sort, one wave64 workgroup per row, rows dealt to XCDs in contiguous ranges.
//
// Rows hold at most M*m+1 members (<= 801 in every reference configuration) with DISTINCT ids that are spread
// over the node range, so the sort is a one-pass bucket sort in LDS: bucket = the top bits of (id - row min)
// scaled to the row's id range (monotone in id), histogram with ds_add, wave scan over the buckets, scatter
// with ds_add cursors, and the few members that share a bucket are ordered by counting smaller ids inside
// it.  O(n) LDS operations per row instead of the log^2(P)/2 full read+write passes of a bitonic network
// (which was LDS-bandwidth bound: 1.16 ms vs the walk kernel's 1.19 ms on the cit2-like batch; a register
// rank sort, O(n^2) compares, was slower still at 1.45 ms).  Rows longer than 1024 use the bitonic fallback.
#include "common.hpp"

namespace subgacc {

constexpr int kSpgThreads = 64;

// SFptr of a member: sf[e] itself, or -- when the caller passes the table of distinct LP rows -- the number of the
// table slot stored in sf[e] (saves the separate translate pass over all members)
__device__ __forceinline__ int32_t sf_of(const int32_t *__restrict__ sf, const int32_t *__restrict__ slot_id, int64_t e) {
    const int32_t v = sf[e];
    return slot_id ? slot_id[v] : v;
}
constexpr int kBucketMaxLen = 1024;   // 16 members per lane in registers
constexpr int kBucketMax = 1024;

__device__ __forceinline__ int32_t wave_min_i32(int32_t v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ int32_t wave_max_i32(int32_t v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
    return v;
}

// E = members per lane, compile time so that the row lives in registers: it is read from HBM exactly once
template <int E>
__device__ __forceinline__ void bucket_sort_row(const int32_t *__restrict__ ids, const int32_t *__restrict__ sf,
                                                const int32_t *__restrict__ slot_id,
                                                int64_t beg, int ns, int lane, int bcap, unsigned long long *tmp,
                                                int32_t *start, int32_t *cursor, int32_t *__restrict__ out_indices,
                                                int32_t *__restrict__ out_data) {
    int32_t x[E], v[E];
    int32_t mn = 0x7FFFFFFF, mx = 0;
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int r = lane + u * kSpgThreads;
        x[u] = 0, v[u] = 0;
        if (r < ns) {
            x[u] = ids[beg + r];
            v[u] = sf_of(sf, slot_id, beg + r) + 1;
            mn = min(mn, x[u]);
            mx = max(mx, x[u]);
        }
    }
    mn = wave_min_i32(mn);
    mx = wave_max_i32(mx);
    int logb = 0;
    while ((1 << logb) < ns && (1 << logb) < bcap) ++logb;
    const int B = 1 << logb;
    const uint32_t range = (uint32_t)(mx - mn) + 1u;
    // bucket(id) = ((id - mn) << logb) >> Ls with 2^Ls >= range: monotone in id and < B
    const int Ls = (range <= 1u) ? 0 : (32 - __builtin_clz(range - 1u));
    for (int b = lane; b < B; b += kSpgThreads) cursor[b] = 0;
    __syncthreads();
    uint32_t bk[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        bk[u] = (uint32_t)(((uint64_t)(uint32_t)(x[u] - mn) << logb) >> Ls);
        if (lane + u * kSpgThreads < ns) atomicAdd(&cursor[bk[u]], 1);
    }
    __syncthreads();
    {   // exclusive scan of the histogram: B/64 consecutive buckets per lane + one wave scan
        const int per = (B + kSpgThreads - 1) / kSpgThreads;
        const int b0 = lane * per;
        int32_t s = 0;
        for (int b = b0; b < b0 + per && b < B; ++b) s += cursor[b];
        int32_t inc = s;
#pragma unroll
        for (int dd = 1; dd < kWave; dd <<= 1) {
            const int32_t t = __shfl_up(inc, dd, kWave);
            if (lane >= dd) inc += t;
        }
        int32_t run = inc - s;
        for (int b = b0; b < b0 + per && b < B; ++b) {
            const int32_t c = cursor[b];
            start[b] = run;
            cursor[b] = run;
            run += c;
        }
        if (lane == kSpgThreads - 1) start[B] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; ++u)
        if (lane + u * kSpgThreads < ns) {
            const int slot = atomicAdd(&cursor[bk[u]], 1);
            tmp[slot] = ((unsigned long long)(uint32_t)x[u] << 32) | (uint32_t)v[u];
        }
    __syncthreads();
    // order inside a bucket: final position = bucket start + number of smaller ids in the bucket
    int32_t pos[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        pos[u] = -1;
        if (lane + u * kSpgThreads < ns) {
            const unsigned long long me = ((unsigned long long)(uint32_t)x[u] << 32) | (uint32_t)v[u];
            const int lo = start[bk[u]], hi = start[bk[u] + 1];
            int rank = 0;
            for (int t = lo; t < hi; ++t) rank += (tmp[t] < me) ? 1 : 0;
            pos[u] = lo + rank;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; ++u)
        if (pos[u] >= 0) tmp[pos[u]] = ((unsigned long long)(uint32_t)x[u] << 32) | (uint32_t)v[u];
    __syncthreads();
    for (int r = lane; r < ns; r += kSpgThreads) {   // coalesced write-back of the sorted row
        const unsigned long long w = tmp[r];
        out_indices[beg + r] = (int32_t)(w >> 32);
        out_data[beg + r] = (int32_t)(uint32_t)w;
    }
}

__global__ __launch_bounds__(kSpgThreads) void spg_bucket_kernel(const int64_t *__restrict__ row_off, int64_t n,
                                                                  const int32_t *__restrict__ ids,
                                                                  const int32_t *__restrict__ sf,
                                                                  const int32_t *__restrict__ slot_id, int32_t cap,
                                                                  int32_t bcap, int32_t *__restrict__ out_indices,
                                                                  int32_t *__restrict__ out_data, int32_t *flags) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    unsigned long long *tmp = (unsigned long long *)lds_raw;   // [cap]    members grouped by bucket, then sorted
    int32_t *start = (int32_t *)(tmp + cap);                   // [bcap+1] first slot of every bucket
    int32_t *cursor = start + bcap + 1;                        // [bcap]   histogram, then scatter cursors
    const int64_t i = xcd_item(blockIdx.x, gridDim.x);
    if (i >= n) return;
    const int64_t beg = row_off[i];
    const int64_t ns64 = row_off[i + 1] - beg;
    if (ns64 > cap) {  // the caller under-stated max_len: refuse the row, never overrun LDS
        if (threadIdx.x == 0) atomicOr(&flags[3], 1);
        return;
    }
    const int ns = (int)ns64, lane = threadIdx.x;
#define SG_ROW(EE) bucket_sort_row<EE>(ids, sf, slot_id, beg, ns, lane, bcap, tmp, start, cursor, out_indices, out_data)
    const int e = (ns + kSpgThreads - 1) / kSpgThreads;
    if (e == 0) return;
    else if (e <= 1) SG_ROW(1);
    else if (e <= 2) SG_ROW(2);
    else if (e <= 3) SG_ROW(3);
    else if (e <= 4) SG_ROW(4);
    else if (e <= 5) SG_ROW(5);
    else if (e <= 6) SG_ROW(6);
    else if (e <= 7) SG_ROW(7);
    else if (e <= 8) SG_ROW(8);
    else if (e <= 10) SG_ROW(10);
    else if (e <= 13) SG_ROW(13);
    else SG_ROW(16);
#undef SG_ROW
}

__global__ __launch_bounds__(kSpgThreads) void spg_build_kernel(const int64_t *__restrict__ row_off, int64_t n,
                                                                 const int32_t *__restrict__ ids,
                                                                 const int32_t *__restrict__ sf,
                                                                 const int32_t *__restrict__ slot_id, int32_t max_pow2,
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
        buf[r] = r < ns ? (((unsigned long long)(uint32_t)ids[beg + r] << 32) | (uint32_t)(sf_of(sf, slot_id, beg + r) + 1)) : ~0ull;
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
                                 const void *uniq_table, int64_t uniq_capacity, int32_t max_len, int32_t *out_indices,
                                 int32_t *out_data, int32_t *flags, void *stream) {
    // the id column of the table (layout of uniq_table.hpp: keys u64[cap], mintag u64[cap], id i32[cap])
    const int32_t *slot_id = uniq_table ? (const int32_t *)((const char *)uniq_table + (size_t)uniq_capacity * 16) : nullptr;
    SG_REQUIRE(!uniq_table || uniq_capacity > 0, SUBGACC_ERR_BADARG, "spg_build: table without capacity");
    SG_REQUIRE(n >= 0 && max_len >= 0 && flags, SUBGACC_ERR_BADARG, "spg_build: bad arguments");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(row_off && ids && sf && out_indices && out_data, SUBGACC_ERR_BADARG, "spg_build: null argument");
    const int64_t grid = xcd_grid(n);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "spg_build: too many rows in one call");
    if (max_len <= kBucketMaxLen) {
        const int cap = max_len > 0 ? max_len : 1;
        int bcap = 64;
        while (bcap < cap && bcap < kBucketMax) bcap <<= 1;
        if (bcap > 512) bcap = 512;   // <= 2 members per bucket on average; halves the LDS footprint
        const size_t lds_b = (size_t)cap * 8 + (size_t)(2 * bcap + 1) * 4;
        hipLaunchKernelGGL(spg_bucket_kernel, dim3((unsigned)grid), dim3(kSpgThreads), lds_b, (hipStream_t)stream,
                           row_off, n, ids, sf, slot_id, cap, bcap, out_indices, out_data, flags);
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
                       ids, sf, slot_id, P, out_indices, out_data, flags);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
