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
#include "uniq_table.hpp"

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

// The sort proper: the row (ids x[], payload v[], E members per lane) is in registers; sorted by id it goes to
// out_indices / out_data [beg, beg + ns).
// Level 1: B buckets of equal id width.  Level 2 (only where level 1 left a crowded bucket: ids with locality -- most of a set
// inside one community of consecutive ids -- put hundreds of members into ONE bucket, and the ranking by counting below is
// quadratic in the bucket size): bucket b gets as many sub-buckets as it has members, sub-bucket = offset inside b's id window
// scaled by b's count, so that idx2 = start[b] + sub maps the ids monotonically onto [0, ns) following the row's own
// distribution; `cnt2` holds the ns + 1 level-2 counters, two 16-bit counters per word (walk_rows.hip has the same sort).
constexpr int kFineAbove = 12;
template <int E>
__device__ __forceinline__ void bucket_sort_regs(const int32_t (&x)[E], const int32_t (&v)[E], int32_t mn, int32_t mx,
                                                 int64_t beg, int ns, int lane, int bcap, unsigned long long *tmp,
                                                 int32_t *start, int32_t *cursor, int32_t *__restrict__ out_indices,
                                                 int32_t *__restrict__ out_data, uint32_t *cnt2) {
    int logb = 0;
    while ((1 << logb) < ns && (1 << logb) < bcap) ++logb;
    const int B = 1 << logb;
    const uint32_t range = (uint32_t)(mx - mn) + 1u;
    // bucket(id) = ((id - mn) << logb) >> Ls with 2^Ls >= range: monotone in id and < B
    const int Ls = (range <= 1u) ? 0 : (32 - __builtin_clz(range - 1u));
    for (int b = lane; b < B; b += kSpgThreads) cursor[b] = 0;
    __syncthreads();
    uint32_t bk[E];
    int32_t arr[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        bk[u] = (uint32_t)(((uint64_t)(uint32_t)(x[u] - mn) << logb) >> Ls);
        arr[u] = 0;
        if (lane + u * kSpgThreads < ns) arr[u] = atomicAdd(&cursor[bk[u]], 1);
    }
    __syncthreads();
    int32_t maxc = 0;
    {   // exclusive scan of the histogram: B/64 consecutive buckets per lane + one wave scan
        const int per = (B + kSpgThreads - 1) / kSpgThreads;
        const int b0 = lane * per;
        int32_t s = 0;
        for (int b = b0; b < b0 + per && b < B; ++b) {
            s += cursor[b];
            maxc = max(maxc, cursor[b]);
        }
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
            run += c;
        }
        if (lane == kSpgThreads - 1) start[B] = inc;
        maxc = wave_max_i32(maxc);
    }
    __syncthreads();
    int32_t lo[E], hi[E];
    if (maxc <= kFineAbove || Ls <= logb) {        // evenly spread ids (or one bucket per id): level 1 is the sort
#pragma unroll
        for (int u = 0; u < E; ++u) {
            lo[u] = hi[u] = 0;
            if (lane + u * kSpgThreads < ns) lo[u] = start[bk[u]], hi[u] = start[bk[u] + 1];
        }
    } else {
        const int bshift = Ls - logb;
        const int W2 = (ns + 2) / 2 + 1;
        for (int w = lane; w < W2; w += kSpgThreads) cnt2[w] = 0u;
        __syncthreads();
        uint32_t idx2[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            idx2[u] = 0;
            if (lane + u * kSpgThreads < ns) {
                const uint32_t lo1 = (uint32_t)start[bk[u]], kb = (uint32_t)start[bk[u] + 1] - lo1;
                const uint32_t off = (uint32_t)(x[u] - mn) - (bk[u] << bshift);                     // < 2^bshift
                idx2[u] = lo1 + __umulhi(off << (32 - bshift), kb);                               // floor(off * kb / 2^bshift) < kb
                const uint32_t sh = (idx2[u] & 1u) * 16u;
                arr[u] = (int32_t)((atomicAdd(&cnt2[idx2[u] >> 1], 1u << sh) >> sh) & 0xFFFFu);
            }
        }
        __syncthreads();
        {   // exclusive scan of the level-2 counters in place: consecutive words per lane + one wave scan
            const int per = (W2 + kSpgThreads - 1) / kSpgThreads;
            const int w0 = lane * per;
            int32_t s = 0;
            for (int w = w0; w < w0 + per && w < W2; ++w) s += (int32_t)((cnt2[w] & 0xFFFFu) + (cnt2[w] >> 16));
            int32_t inc = s;
#pragma unroll
            for (int dd = 1; dd < kWave; dd <<= 1) {
                const int32_t t = __shfl_up(inc, dd, kWave);
                if (lane >= dd) inc += t;
            }
            int32_t run = inc - s;
            for (int w = w0; w < w0 + per && w < W2; ++w) {
                const uint32_t c = cnt2[w];
                const uint32_t lo16 = (uint32_t)run;
                run += (int32_t)(c & 0xFFFFu);
                const uint32_t hi16 = (uint32_t)run;
                run += (int32_t)(c >> 16);
                cnt2[w] = lo16 | (hi16 << 16);
            }
        }
        __syncthreads();
        const uint16_t *off2 = (const uint16_t *)cnt2;
#pragma unroll
        for (int u = 0; u < E; ++u) {
            lo[u] = hi[u] = 0;
            if (lane + u * kSpgThreads < ns) lo[u] = off2[idx2[u]], hi[u] = off2[idx2[u] + 1];
        }
    }
#pragma unroll
    for (int u = 0; u < E; ++u)
        if (lane + u * kSpgThreads < ns) tmp[lo[u] + arr[u]] = ((unsigned long long)(uint32_t)x[u] << 32) | (uint32_t)v[u];
    __syncthreads();
    // order inside a (sub-)bucket: final position = its start + number of smaller ids in it
    int32_t pos[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        pos[u] = -1;
        if (lane + u * kSpgThreads < ns) {
            const unsigned long long me = ((unsigned long long)(uint32_t)x[u] << 32) | (uint32_t)v[u];
            int rank = 0;
            for (int t = lo[u]; t < hi[u]; ++t) rank += (tmp[t] < me) ? 1 : 0;
            pos[u] = lo[u] + rank;
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
    bucket_sort_regs<E>(x, v, mn, mx, beg, ns, lane, bcap, tmp, start, cursor, out_indices, out_data, (uint32_t *)(cursor + bcap));
}

// ---------------------------------------------------------------------------------------------------------
// finish_rows: the sets of subgacc_walk_sets (strided staging: ids in first-visit order + packed LP keys) become
// finished SpG rows IN PLACE -- what subgacc_walk_spg emits directly -- in one pass, one wave per root: the set's LP
// keys are folded in a wave-private LDS table (a few dozen distinct rows per set) and registered in the HBM table of
// distinct rows with tag (root_base + i)*stride + first-visit rank (= the member's index in the staging row: the
// numbering of subg_acc.c:957-978), then the members are bucket-sorted by node id from registers
// (random_walks.py:79-80) and written back as (sorted ids, table slots).  Replaces compact_sets + spg_build (two
// passes over all members through HBM and a packed copy) for a batch that is joined from its strided rows.
constexpr int kFinFold = 256;
template <int E>
__device__ __forceinline__ void finish_row(int32_t *__restrict__ row_ids, const unsigned long long *__restrict__ row_keys,
                                           int64_t obase, int ns, int lane, int bcap, UniqTable t, unsigned long long tag0,
                                           unsigned char *lds, int cap, int32_t *__restrict__ row_slot, int32_t *flags) {
    unsigned long long *wk = (unsigned long long *)lds;          // [kFinFold] distinct keys of the set
    uint32_t *wt = (uint32_t *)(wk + kFinFold);                  // [kFinFold] smallest first-visit rank of the key
    int32_t *ws = (int32_t *)(wt + kFinFold);                    // [kFinFold] its slot in the HBM table
    int32_t x[E], v[E];
    unsigned long long key[E];
    int32_t mn = 0x7FFFFFFF, mx = 0;
    for (int s2 = lane; s2 < kFinFold; s2 += kSpgThreads) {
        wk[s2] = kEmptyKey;
        wt[s2] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int r = lane + u * kSpgThreads;
        x[u] = 0, v[u] = -1, key[u] = 0;
        if (r < ns) {
            x[u] = row_ids[obase + r];
            key[u] = row_keys[obase + r];
            mn = min(mn, x[u]);
            mx = max(mx, x[u]);
        }
    }
    mn = wave_min_i32(mn);
    mx = wave_max_i32(mx);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int r = lane + u * kSpgThreads;
        if (r >= ns) continue;
        uint32_t f = (uint32_t)(mix64(key[u]) >> 40) & (kFinFold - 1);
        bool done = false;
        for (int p = 0; p < 16; ++p) {
            unsigned long long cur = wk[f];
            if (cur == kEmptyKey) cur = atomicCAS(&wk[f], kEmptyKey, key[u]);
            if (cur == kEmptyKey || cur == key[u]) {
                if (wt[f] > (uint32_t)r) atomicMin(&wt[f], (uint32_t)r);
                v[u] = -2 - (int32_t)f;      // resolved below
                done = true;
                break;
            }
            f = (f + 1) & (kFinFold - 1);
        }
        if (!done) v[u] = uniq_global_insert(t, key[u], tag0 + (unsigned long long)r, flags);   // crowded fold table
    }
    __syncthreads();
    for (int s2 = lane; s2 < kFinFold; s2 += kSpgThreads)
        if (wk[s2] != kEmptyKey) ws[s2] = uniq_global_insert(t, wk[s2], tag0 + wt[s2], flags);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; ++u)
        if (v[u] <= -2) v[u] = ws[-2 - v[u]];
    __syncthreads();                                             // the fold tables are dead: the sort re-uses the LDS
    unsigned long long *tmp = (unsigned long long *)lds;         // [cap]
    int32_t *start = (int32_t *)(tmp + cap);                     // [bcap+1]
    int32_t *cursor = start + bcap + 1;                          // [bcap]
    bucket_sort_regs<E>(x, v, mn, mx, obase, ns, lane, bcap, tmp, start, cursor, row_ids, row_slot, (uint32_t *)(cursor + bcap));
}

// EMAX = ceil(stride / 64): the longest row the launch can meet.  A compile-time bound so that the register budget
// (and with it the number of resident waves: the kernel is latency x occupancy bound) is that of the rows that occur,
// not of the 1024-member worst case.
template <int EMAX>
__global__ __launch_bounds__(kSpgThreads) void finish_rows_kernel(int32_t *__restrict__ row_ids,
                                                                   const unsigned long long *__restrict__ row_keys,
                                                                   const int32_t *__restrict__ nsize, int64_t n,
                                                                   int32_t stride, int64_t root_base, UniqTable t,
                                                                   int32_t cap, int32_t bcap,
                                                                   int32_t *__restrict__ row_slot, int32_t *flags) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int64_t i = xcd_item(blockIdx.x, gridDim.x);
    if (i >= n) return;
    const int ns = nsize[i], lane = threadIdx.x;
    if (ns > cap) {   // cannot happen for sets of subgacc_walk_sets (ns <= stride <= cap); never overrun LDS
        if (lane == 0) atomicOr(&flags[3], 1);
        return;
    }
    const int64_t obase = i * (int64_t)stride;
    const unsigned long long tag0 = (unsigned long long)((root_base + i) * (int64_t)stride);
#define SG_FIN(EE) finish_row<EE>(row_ids, row_keys, obase, ns, lane, bcap, t, tag0, lds_raw, cap, row_slot, flags)
    const int e = (ns + kSpgThreads - 1) / kSpgThreads;
    if (e == 0) return;
    if constexpr (EMAX <= 4) {
        if (e <= 2) SG_FIN(2);
        else SG_FIN(4);
    } else if constexpr (EMAX <= 7) {
        if (e <= 2) SG_FIN(2);
        else if (e <= 4) SG_FIN(4);
        else SG_FIN(7);
    } else if constexpr (EMAX <= 10) {
        if (e <= 4) SG_FIN(4);
        else if (e <= 7) SG_FIN(7);
        else SG_FIN(10);
    } else {
        if (e <= 4) SG_FIN(4);
        else if (e <= 7) SG_FIN(7);
        else if (e <= 10) SG_FIN(10);
        else if (e <= 13) SG_FIN(13);
        else SG_FIN(16);
    }
#undef SG_FIN
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
        const size_t lds_b = (size_t)cap * 8 + (size_t)(2 * bcap + 1) * 4 + ((size_t)(cap + 2) / 2 + 1) * 4;    // + the level-2 counters
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

extern "C" int subgacc_finish_rows(int32_t *row_ids, const uint64_t *row_keys, const int32_t *nsize, int64_t n,
                                   int32_t stride, int64_t root_base, void *uniq_table, int64_t uniq_capacity,
                                   int32_t *row_slot, int32_t *flags, void *stream) {
    SG_REQUIRE(n >= 0 && stride > 0 && root_base >= 0 && flags, SUBGACC_ERR_BADARG, "finish_rows: bad arguments");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(row_ids && row_keys && nsize && row_slot, SUBGACC_ERR_BADARG, "finish_rows: null argument");
    SG_REQUIRE(uniq_table && uniq_capacity > 0 && (uniq_capacity & (uniq_capacity - 1)) == 0 && uniq_capacity < (1ll << 31),
               SUBGACC_ERR_BADARG, "finish_rows: needs a power-of-two table of distinct rows");
    SG_REQUIRE(stride <= kBucketMaxLen, SUBGACC_ERR_LDS,
               "finish_rows: rows of up to %d members (> %d): use subgacc_compact_sets + subgacc_spg_build", stride,
               kBucketMaxLen);
    const int cap = stride;
    int bcap = 64;       // <= 256 buckets: ~2 members per bucket for the typical set, and the smaller LDS footprint lets
    while (bcap < cap && bcap < 256) bcap <<= 1;   // ~30 rows (waves) be resident per CU -- the kernel is latency x occupancy bound
    const size_t sort_b = (size_t)cap * 8 + (size_t)(2 * bcap + 1) * 4 + ((size_t)(cap + 2) / 2 + 1) * 4, fold_b = (size_t)kFinFold * 16;
    const size_t lds = sort_b > fold_b ? sort_b : fold_b;
    const int64_t grid = xcd_grid(n);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "finish_rows: too many rows in one call");
    const int emax = (cap + kSpgThreads - 1) / kSpgThreads;
#define SG_FIN_LAUNCH(EM)                                                                                               \
    hipLaunchKernelGGL(finish_rows_kernel<EM>, dim3((unsigned)grid), dim3(kSpgThreads), lds, (hipStream_t)stream, row_ids, \
                       (const unsigned long long *)row_keys, nsize, n, stride, root_base,                               \
                       uniq_view(uniq_table, uniq_capacity), cap, bcap, row_slot, flags)
    if (emax <= 4) SG_FIN_LAUNCH(4);
    else if (emax <= 7) SG_FIN_LAUNCH(7);
    else if (emax <= 10) SG_FIN_LAUNCH(10);
    else SG_FIN_LAUNCH(16);
#undef SG_FIN_LAUNCH
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Packed rows -> headed rows (ABI 7): the resident store of a serving loop on whole 128-byte lines (include/subgacc.h: the HEADED
// layout of subgacc_join_desc).  One wavefront per row, four rows per workgroup: the row's length goes into slot 0 of its ids, its
// members behind it, payloads at the same pitch -- a streaming copy (reads and writes of consecutive lanes on consecutive words).
namespace subgacc {
template <typename P>
__global__ __launch_bounds__(256) void rows_to_headed_kernel(const int64_t *__restrict__ row_off, int64_t n, const int32_t *__restrict__ ids,
                                                             const P *__restrict__ payload, int64_t stride, int32_t *__restrict__ out_ids,
                                                             P *__restrict__ out_payload, int32_t *__restrict__ flags) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t r = xcd_item(blockIdx.x, gridDim.x) * 4 + threadIdx.x / kWave;
    if (r >= n) return;
    const int64_t beg = row_off[r];
    int64_t len = row_off[r + 1] - beg;
    if (len > stride - 1) {        // does not fit its slot: cut (never out of bounds), and said
        len = stride - 1;
        if (lane == 0 && flags) atomicOr(&flags[3], 1);
    }
    int32_t *oi = out_ids + r * stride;
    P *op = out_payload + r * stride;
    if (lane == 0) oi[0] = (int32_t)len;
    for (int64_t t = lane; t < len; t += kWave) {
        oi[1 + t] = __builtin_nontemporal_load(&ids[beg + t]);
        op[t] = __builtin_nontemporal_load(&payload[beg + t]);
    }
}
}  // namespace subgacc

extern "C" int subgacc_rows_to_headed(const int64_t *row_off, int64_t n_rows, const int32_t *ids, const void *payload, int32_t payload_bytes,
                                      int64_t row_stride, int32_t *out_ids, void *out_payload, int32_t *flags, void *stream) {
    using namespace subgacc;
    SG_REQUIRE(n_rows >= 0 && row_stride > 1 && row_stride < (1ll << 31) && (payload_bytes == 4 || payload_bytes == 8), SUBGACC_ERR_BADARG,
               "rows_to_headed: bad arguments (row_stride > 1, payload_bytes 4 or 8)");
    if (n_rows == 0) return SUBGACC_OK;
    SG_REQUIRE(row_off && ids && payload && out_ids && out_payload, SUBGACC_ERR_BADARG, "rows_to_headed: null argument");
    const int64_t grid = xcd_grid((n_rows + 3) / 4);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "rows_to_headed: too many rows in one call");
    if (payload_bytes == 4)
        hipLaunchKernelGGL(rows_to_headed_kernel<int32_t>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, row_off, n_rows, ids,
                           (const int32_t *)payload, row_stride, out_ids, (int32_t *)out_payload, flags);
    else
        hipLaunchKernelGGL(rows_to_headed_kernel<unsigned long long>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, row_off, n_rows,
                           ids, (const unsigned long long *)payload, row_stride, out_ids, (unsigned long long *)out_payload, flags);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
