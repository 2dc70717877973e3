// uniq.hip -- global first-occurrence dedup of packed LP rows (gfx950).
//
// Replaces the serial uthash pass of set_sampler (reference subg_acc/subg_acc.c:957-978): every set member's
// 64-bit LP key is mapped to the index of its first occurrence over the concatenated sets, and the distinct
// rows are emitted in that order (subg_acc.c:982-1000).  A sequential "have I seen this key" loop becomes:
//   1. insert: open-addressing table in HBM, key -> min element position (global atomicMin).  The number of
//      distinct LP rows is tiny next to the number of members (493x..19527x compression in the paper), so
//      the table is L2-resident and a coherent load in front of the atomic skips almost all of them.
//   2. number: an element is a "first occurrence" iff the table's min position is its own; an exclusive scan
//      of those flags in element order IS the reference's numbering.  No sort.
//   3. translate: every element reads its key's number.
#include "common.hpp"
#include "blockscan.hpp"
#include "uniq_table.hpp"

namespace subgacc {

__global__ void uniq_reset_kernel(UniqTable t, int64_t cap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        t.keys[i] = kEmptyKey;
        t.mintag[i] = ~0ull;
        t.id[i] = -1;
    }
}

// What an on-demand step (sample the endpoints of B pairs -> rows -> join) needs before its walk kernel, in ONE launch
// instead of four: the table of distinct LP rows reset, the step's status words zeroed, the int64 endpoints of the pairs
// narrowed to the int32 roots the sampler takes (force-cast like the reference's query, subg_acc.c:673).
__global__ void step_prologue_kernel(UniqTable t, int64_t cap, int64_t *zero_words, int64_t n_zero, const int64_t *edge,
                                     int32_t *roots, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {        // cap = 0: no table (key rows)
        t.keys[i] = kEmptyKey;
        t.mintag[i] = ~0ull;
        t.id[i] = -1;
    }
    if (i < n_zero) zero_words[i] = 0;
    if (i < n) {
        const int64_t v = edge[i];     // ids beyond int32 cannot be nodes of an int32 graph: out of range for the walk kernel
        roots[i] = (v < 0 || v > 0x7FFFFFFFll) ? -1 : (int32_t)v;
    }
}

// The same prologue for a step that samples every DISTINCT endpoint once (Philox keys a walk by its root's id: a root's set does
// not depend on where or how often the root appears in the batch).  The n endpoints go into an open-addressing table in HBM
// whose slots are stamped with the step's generation (kept in the workspace; nothing is ever cleared); every occurrence of a
// root raises the slot's value to (generation, ~index) with a 64-bit atomicMax, so that the slot ends up naming the root's
// FIRST occurrence.  The second kernel gives endpoint j the row of that first occurrence -- own[j], and partner[] of the other
// end of its pair -- leaves roots[j] = the root where j is a first occurrence, SUBGACC_NO_ROOT elsewhere, and lists the first
// occurrences (worklist[0 .. *n_distinct), one atomicAdd per wavefront; the list's order is the order of arrival and does not
// matter: entry k names its row).  Rows are sparse in [0, n) but deterministic -- the sets of the batch sit where the plain step
// has them, and the distinct LP rows keep the numbering they would have had with every endpoint sampled (a repeated root never
// is the first to show a row) -- while the walk kernel runs over the dense list and never sees an empty row.
__global__ void step_dedup_claim_kernel(UniqTable t, int64_t cap, int64_t *zero_words, int64_t n_zero, const int64_t *edge,
                                        int32_t *slot_of, unsigned long long *hkeys, unsigned long long *hvals,
                                        uint32_t hmask, int hshift, const int64_t *last_gen, int64_t *n_distinct, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t gen = (uint32_t)*last_gen + 1u;     // the step's stamp lives in the workspace (the map kernel stores it back): a
    if (gen == 0u) gen = 1u;                      // captured step replays with a fresh one; 0 = "never used" is skipped
    if (i < cap) {
        t.keys[i] = kEmptyKey;
        t.mintag[i] = ~0ull;
        t.id[i] = -1;
    }
    if (i < n_zero) zero_words[i] = 0;
    if (i == 0) *n_distinct = 0;                  // counted by the map kernel
    if (i >= n) return;
    const int64_t v = edge[i];
    const int32_t root = (v < 0 || v > 0x7FFFFFFFll) ? -1 : (int32_t)v;
    const unsigned long long key = ((unsigned long long)gen << 32) | (uint32_t)root;
    uint32_t h = ((uint32_t)root * 2654435761u) >> hshift;
    // (bounded: a workspace that was not zeroed before its first use could hold nothing but current-looking stamps; the
    // walk must end there too -- the results are then as undefined as the workspace was)
    for (uint32_t probes = 0; probes <= hmask; ++probes) {
        unsigned long long cur = __hip_atomic_load(&hkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(cur >> 32) != gen) {        // a slot of an earlier step: free
            const unsigned long long prev = atomicCAS(&hkeys[h], cur, key);
            if (prev == cur) break;
            cur = prev;                             // taken meanwhile -- by this step: look at what sits there now
        }
        if (cur == key) break;                      // the root is there already
        h = (h + 1u) & hmask;
    }
    slot_of[i] = (int32_t)h;
    atomicMax(&hvals[h], ((unsigned long long)gen << 32) | (0xFFFFFFFFu - (uint32_t)i));   // later stamp wins, then the smaller index
}

__global__ __launch_bounds__(256) void step_dedup_map_kernel(const int64_t *__restrict__ edge, const int32_t *__restrict__ slot_of,
                                                             const unsigned long long *__restrict__ hvals, int32_t *__restrict__ roots,
                                                             int64_t *__restrict__ own, int64_t *__restrict__ partner,
                                                             int32_t *__restrict__ worklist, int32_t *__restrict__ row_len, int64_t n,
                                                             int64_t *last_gen, int64_t *n_distinct) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool first = false;
    if (j < n) {
        const int64_t row = (int64_t)(0xFFFFFFFFu - (uint32_t)hvals[slot_of[j]]);
        const int64_t half = n / 2;
        own[j] = row;
        partner[j < half ? j + half : j - half] = row;
        first = row == j;
        const int64_t v = edge[j];
        roots[j] = first ? ((v < 0 || v > 0x7FFFFFFFll) ? -1 : (int32_t)v) : SUBGACC_NO_ROOT;
        if (!first) row_len[j] = 0;      // the walk kernel never visits this row: it is empty, not stale
    }
    // the rows that carry a set, as a dense work list for the walk kernel (its order is the order of arrival and does not
    // matter: entry k names its row); ONE atomicAdd per workgroup (one per wavefront were 2,048 returning atomics on one word
    // for a batch of 65,536 pairs, served one after the other: 20 of the kernel's 28 us)
    __shared__ int32_t wcount[256 / kWave];
    __shared__ long long wg_base;
    const unsigned long long m = __ballot(first);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    if (lane == 0) wcount[wv] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t tot = 0;
#pragma unroll
        for (int w = 0; w < 256 / kWave; ++w) tot += wcount[w];
        wg_base = tot ? (long long)atomicAdd((unsigned long long *)n_distinct, (unsigned long long)tot) : 0ll;
    }
    __syncthreads();
    if (first) {
        long long base = wg_base;
        for (int w = 0; w < wv; ++w) base += wcount[w];
        worklist[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)j;
    }
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) {      // the claim kernel of this step is done: its stamp becomes the last one
            uint32_t gen = (uint32_t)*last_gen + 1u;
            if (gen == 0u) gen = 1u;
            *last_gen = (int64_t)gen;
        }
    }
}

// The distinct LP rows are 10^2..10^5 while the members are 10^7..10^9, so almost every member repeats a key
// its neighbours in the tile already carry.  Each block first folds its tile into an LDS table
// (key -> min position, ds_cmpst_b64 / ds_min_u64) and only the block-distinct keys go to the HBM table:
// two orders of magnitude fewer global atomics on the hot keys.  out_slot[e] (optional) receives the HBM table
// slot of every member, so that the later passes never probe again.
__global__ __launch_bounds__(256) void uniq_insert_kernel(UniqTable t, const uint64_t *__restrict__ keys, int64_t n,
                                                           int64_t tag_base, int32_t *__restrict__ out_slot,
                                                           int32_t *flags) {
    __shared__ unsigned long long lk[kInsLds];
    __shared__ unsigned long long lt[kInsLds];
    __shared__ int32_t ls[kInsLds];
    for (int s = threadIdx.x; s < kInsLds; s += 256) {
        lk[s] = kEmptyKey;
        lt[s] = ~0ull;
    }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kInsTile;
    int32_t mine[kInsItems];   // >= 0: LDS slot of the member; < -1: -(global slot)-2; -1: nothing
#pragma unroll
    for (int k = 0; k < kInsItems; ++k) {
        const int64_t e = base + (int64_t)k * 256 + threadIdx.x;
        mine[k] = -1;
        if (e >= n) continue;
        const unsigned long long key = keys[e];
        const unsigned long long tag = (unsigned long long)(tag_base + e);
        uint32_t h = (uint32_t)(mix64(key) >> 40) & (kInsLds - 1);
        bool done = false;
        for (int p = 0; p < 16; ++p) {
            unsigned long long cur = lk[h];
            if (cur == kEmptyKey) cur = atomicCAS(&lk[h], kEmptyKey, key);
            if (cur == kEmptyKey || cur == key) {
                if (lt[h] > tag) atomicMin(&lt[h], tag);
                mine[k] = (int32_t)h;
                done = true;
                break;
            }
            h = (h + 1) & (kInsLds - 1);
        }
        if (!done) {  // crowded neighbourhood: go straight to HBM
            const int32_t g = uniq_global_insert(t, key, tag, flags);
            mine[k] = g >= 0 ? -g - 2 : -1;
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < kInsLds; s += 256)
        if (lk[s] != kEmptyKey) ls[s] = uniq_global_insert(t, lk[s], lt[s], flags);
    if (!out_slot) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kInsItems; ++k) {
        const int64_t e = base + (int64_t)k * 256 + threadIdx.x;
        if (e >= n) continue;
        const int32_t m = mine[k];
        out_slot[e] = m >= 0 ? ls[m] : (m < -1 ? -m - 2 : 0);
    }
}

// ---- numbering: distinct keys ordered by the position of their first occurrence (= ascending mintag)
// collect the occupied slots
__global__ __launch_bounds__(256) void uniq_collect_kernel(UniqTable t, int64_t cap, unsigned long long *list_tag,
                                                            int32_t *list_slot, unsigned long long *count) {
    const int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= cap || t.keys[h] == kEmptyKey) return;
    const unsigned long long i = atomicAdd(count, 1ull);
    list_tag[i] = t.mintag[h];
    list_slot[i] = (int32_t)h;
}

// few distinct rows (the normal case: hundreds..thousands): rank every entry by counting smaller min positions
__global__ __launch_bounds__(256) void uniq_rank_small_kernel(UniqTable t, const unsigned long long *__restrict__ list_tag,
                                                               const int32_t *__restrict__ list_slot,
                                                               const unsigned long long *__restrict__ count,
                                                               int64_t small_limit, uint64_t *__restrict__ out_ukeys,
                                                               int64_t max_unique) {
    __shared__ unsigned long long tile[256];
    const int64_t c = (int64_t)*count;
    if (c > small_limit) return;   // the element-scan path numbers large tables
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if ((int64_t)blockIdx.x * 256 >= c) return;
    const unsigned long long mytag = i < c ? list_tag[i] : ~0ull;
    int64_t rank = 0;
    for (int64_t base = 0; base < c; base += 256) {
        const int64_t u = base + threadIdx.x;
        tile[threadIdx.x] = u < c ? list_tag[u] : ~0ull;
        __syncthreads();
        const int lim = (int)((c - base) < 256 ? (c - base) : 256);
        for (int x = 0; x < lim; ++x) rank += tile[x] < mytag ? 1 : 0;
        __syncthreads();
    }
    if (i < c) {
        const int32_t h = list_slot[i];
        t.id[h] = (int32_t)rank;
        if (rank < max_unique) out_ukeys[rank] = t.keys[h];
    }
}

// many distinct rows: an element is a first occurrence iff the table's min position is its own; an exclusive
// scan of those flags in element order is the numbering.  pass A: count the flags per tile
__global__ __launch_bounds__(kScanThreads) void uniq_count_kernel(UniqTable t, const int32_t *__restrict__ slot,
                                                                   int64_t n, const unsigned long long *__restrict__ count,
                                                                   int64_t small_limit, int32_t *__restrict__ tile_count) {
    if ((int64_t)*count <= small_limit) return;
    const int64_t base = (int64_t)blockIdx.x * kUniqTile;
    int32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < kUniqItems; ++k) {
        const int64_t e = base + (int64_t)k * kScanThreads + threadIdx.x;
        if (e < n) cnt += (t.mintag[slot[e]] == (unsigned long long)e) ? 1 : 0;
    }
    int32_t tot;
    block_exclusive_scan<int32_t>(cnt, &tot);
    if (threadIdx.x == 0) tile_count[blockIdx.x] = tot;
}

// pass B: number the first occurrences in element order
__global__ __launch_bounds__(kScanThreads) void uniq_assign_kernel(UniqTable t, const int32_t *__restrict__ slot,
                                                                    int64_t n, const unsigned long long *__restrict__ count,
                                                                    int64_t small_limit,
                                                                    const int64_t *__restrict__ tile_off,
                                                                    uint64_t *__restrict__ out_ukeys,
                                                                    int64_t max_unique) {
    if ((int64_t)*count <= small_limit) return;
    const int64_t base = (int64_t)blockIdx.x * kUniqTile;
    int64_t run = tile_off[blockIdx.x];
    for (int k = 0; k < kUniqItems; ++k) {
        const int64_t e = base + (int64_t)k * kScanThreads + threadIdx.x;
        int32_t h = -1, flag = 0;
        if (e < n) {
            h = slot[e];
            flag = (t.mintag[h] == (unsigned long long)e) ? 1 : 0;
        }
        int32_t tot;
        const int32_t ex = block_exclusive_scan<int32_t>(flag, &tot);
        if (flag) {
            const int64_t id = run + ex;
            t.id[h] = (int32_t)id;
            if (id < max_unique) out_ukeys[id] = t.keys[h];
        }
        run += tot;
    }
}

// element -> number of its key (+add)
__global__ __launch_bounds__(256) void uniq_translate_kernel(UniqTable t, int64_t n, const int64_t *__restrict__ n_dev,
                                                              int32_t *__restrict__ sf, int32_t add) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev && *n_dev < n) n = *n_dev;   // the element count may still live on the device (no host round trip)
    if (e < n) sf[e] = t.id[sf[e]] + add;
}

__global__ void set_i64_kernel(int64_t *p, int64_t v) { *p = v; }

template <typename OutI>
__device__ __forceinline__ void unpack_row(unsigned long long key, int M, int m, int shift, OutI *row) {
    const unsigned long long fmask = (1ull << shift) - 1ull;
    row[0] = (OutI)(((key >> (m * shift)) & 1ull) ? M : 0);
    for (int j = 1; j <= m; ++j) row[j] = (OutI)((key >> ((m - j) * shift)) & fmask);
}

__global__ __launch_bounds__(256) void unpack_lp_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                         const int64_t *__restrict__ n_dev, int M, int m, int shift,
                                                         int16_t *o16, int32_t *o32, float *of32, int zero_row) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ncol = m + 1;
    if (of32 && zero_row && e < ncol) of32[e] = 0.0f;
    if (e >= n) return;
    if (n_dev && e >= *n_dev) {   // rows past the device-side count: defined (zero), never indexed
        if (o16) for (int j = 0; j < ncol; ++j) o16[e * ncol + j] = 0;
        if (o32) for (int j = 0; j < ncol; ++j) o32[e * ncol + j] = 0;
        if (of32) for (int j = 0; j < ncol; ++j) of32[(e + (zero_row ? 1 : 0)) * ncol + j] = 0.0f;
        return;
    }
    const unsigned long long key = keys[e];
    if (o16) unpack_row<int16_t>(key, M, m, shift, o16 + e * ncol);
    if (o32) unpack_row<int32_t>(key, M, m, shift, o32 + e * ncol);
    if (of32) {
        float *row = of32 + (e + (zero_row ? 1 : 0)) * ncol;
        const unsigned long long fmask = (1ull << shift) - 1ull;
        const float fm = (float)M;
        row[0] = (((key >> (m * shift)) & 1ull) ? fm : 0.0f) / fm;
        for (int j = 1; j <= m; ++j) row[j] = (float)((key >> ((m - j) * shift)) & fmask) / fm;
    }
}

}  // namespace subgacc

using namespace subgacc;

static bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" size_t subgacc_uniq_table_bytes(int64_t capacity) {
    if (capacity <= 0) return 0;
    return (size_t)capacity * (8 + 8 + 4);
}

extern "C" int subgacc_uniq_reset(void *table, int64_t capacity, void *stream) {
    SG_REQUIRE(table && is_pow2(capacity) && capacity < (1ll << 31), SUBGACC_ERR_BADARG,
               "uniq_reset: capacity must be a power of two below 2^31");
    hipLaunchKernelGGL(uniq_reset_kernel, dim3((unsigned)ceil_div(capacity, 256)), dim3(256), 0, (hipStream_t)stream,
                       uniq_view(table, capacity), capacity);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_step_prologue(void *table, int64_t capacity, int64_t *zero_words, int64_t n_zero, const int64_t *edge,
                                     int32_t *roots, int64_t n, void *stream) {
    SG_REQUIRE(!table || (is_pow2(capacity) && capacity < (1ll << 31)), SUBGACC_ERR_BADARG,
               "step_prologue: capacity must be a power of two below 2^31");
    if (!table) capacity = 0;
    SG_REQUIRE(n >= 0 && n_zero >= 0 && (n == 0 || (edge && roots)) && (n_zero == 0 || zero_words), SUBGACC_ERR_BADARG,
               "step_prologue: null argument");
    int64_t span = capacity > n ? capacity : n;
    if (n_zero > span) span = n_zero;
    if (span == 0) return SUBGACC_OK;
    hipLaunchKernelGGL(step_prologue_kernel, dim3((unsigned)ceil_div(span, 256)), dim3(256), 0, (hipStream_t)stream,
                       table ? uniq_view(table, capacity) : UniqTable{nullptr, nullptr, nullptr, 0}, capacity, zero_words, n_zero,
                       edge, roots, n);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

static int64_t dedup_slots(int64_t n) {
    int64_t c = 1024;
    while (c < 2 * n) c <<= 1;
    return c;
}

extern "C" size_t subgacc_step_dedup_workspace_bytes(int64_t n) {
    if (n < 0) n = 0;
    const int64_t c = dedup_slots(n);
    return 256 + 2 * align_up((size_t)c * 8, 256) + align_up((size_t)n * 4, 256);
}

extern "C" int subgacc_step_prologue_dedup(void *table, int64_t capacity, int64_t *zero_words, int64_t n_zero, const int64_t *edge,
                                           int32_t *roots, int64_t *own, int64_t *partner, int32_t *worklist, int32_t *row_len,
                                           int64_t n, void *workspace, size_t workspace_bytes, int64_t *n_distinct, void *stream) {
    SG_REQUIRE(!table || (is_pow2(capacity) && capacity < (1ll << 31)), SUBGACC_ERR_BADARG,
               "step_prologue_dedup: capacity must be a power of two below 2^31");
    if (!table) capacity = 0;
    SG_REQUIRE(n > 0 && n % 2 == 0 && n < (1ll << 30) && n_zero >= 0 && edge && roots && own && partner && worklist && row_len && n_distinct &&
                   (n_zero == 0 || zero_words),
               SUBGACC_ERR_BADARG, "step_prologue_dedup: bad arguments (n = 2B endpoints)");
    SG_REQUIRE(workspace && workspace_bytes >= subgacc_step_dedup_workspace_bytes(n), SUBGACC_ERR_WORKSPACE,
               "step_prologue_dedup: workspace too small");
    const int64_t c = dedup_slots(n);
    char *w = (char *)workspace;
    int64_t *last_gen = (int64_t *)w;                       // the stamp of the previous call on this workspace (0 at first)
    unsigned long long *hkeys = (unsigned long long *)(w + 256);
    unsigned long long *hvals = (unsigned long long *)(w + 256 + align_up((size_t)c * 8, 256));
    int32_t *slot_of = (int32_t *)(w + 256 + 2 * align_up((size_t)c * 8, 256));
    int64_t span = capacity > n ? capacity : n;
    if (n_zero > span) span = n_zero;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(step_dedup_claim_kernel, dim3((unsigned)ceil_div(span, 256)), dim3(256), 0, s,
                       table ? uniq_view(table, capacity) : UniqTable{nullptr, nullptr, nullptr, 0}, capacity, zero_words, n_zero,
                       edge, slot_of, hkeys, hvals, (uint32_t)(c - 1), 32 - (63 - __builtin_clzll((unsigned long long)c)), last_gen,
                       n_distinct, n);
    hipLaunchKernelGGL(step_dedup_map_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, edge, slot_of, hvals, roots, own,
                       partner, worklist, row_len, n, last_gen, n_distinct);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_uniq_insert(void *table, int64_t capacity, const uint64_t *keys, int64_t n, int64_t tag_base,
                                   int32_t *out_slot, int32_t *flags, void *stream) {
    SG_REQUIRE(table && is_pow2(capacity) && capacity < (1ll << 31) && flags && n >= 0 && tag_base >= 0,
               SUBGACC_ERR_BADARG, "uniq_insert: bad arguments");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(keys, SUBGACC_ERR_BADARG, "uniq_insert: null keys");
    const int64_t blocks = ceil_div(n, kInsTile);
    SG_REQUIRE(blocks < (1ll << 31), SUBGACC_ERR_BADARG, "uniq_insert: split the call (n too large)");
    hipLaunchKernelGGL(uniq_insert_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       uniq_view(table, capacity), keys, n, tag_base, out_slot, flags);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

static size_t number_list_bytes(int64_t capacity) {
    return align_up((size_t)capacity * 8, 256) + align_up((size_t)capacity * 4, 256);
}

extern "C" size_t subgacc_uniq_number_workspace_bytes(int64_t capacity, int64_t n) {
    if (n < 0) n = 0;
    if (capacity < 0) capacity = 0;
    const int64_t tiles = ceil_div(n > 0 ? n : 1, kUniqTile);
    return number_list_bytes(capacity) + align_up((size_t)tiles * 4, 256) + align_up((size_t)(tiles + 1) * 8, 256) +
           scan_workspace_bytes(tiles);
}

extern "C" int subgacc_uniq_number(void *table, int64_t capacity, const int32_t *slot, int64_t n,
                                   uint64_t *out_ukeys, int64_t max_unique, int64_t *out_count,
                                   int64_t small_limit, void *workspace, size_t workspace_bytes, void *stream) {
    SG_REQUIRE(table && is_pow2(capacity) && capacity < (1ll << 31) && out_count && n >= 0 && max_unique >= 0,
               SUBGACC_ERR_BADARG, "uniq_number: bad arguments");
    SG_REQUIRE(out_ukeys || max_unique == 0, SUBGACC_ERR_BADARG, "uniq_number: null out_ukeys");
    SG_REQUIRE(workspace && workspace_bytes >= subgacc_uniq_number_workspace_bytes(capacity, n), SUBGACC_ERR_WORKSPACE,
               "uniq_number: workspace too small");
    if (small_limit <= 0) small_limit = 8192;
    if (small_limit > capacity) small_limit = capacity;
    SG_REQUIRE(slot || n == 0, SUBGACC_ERR_BADARG, "uniq_number: null slot array");
    hipStream_t s = (hipStream_t)stream;
    UniqTable t = uniq_view(table, capacity);
    char *ws = (char *)workspace;
    unsigned long long *list_tag = (unsigned long long *)ws;
    ws += align_up((size_t)capacity * 8, 256);
    int32_t *list_slot = (int32_t *)ws;
    ws += align_up((size_t)capacity * 4, 256);
    unsigned long long *count = (unsigned long long *)out_count;
    hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, s, out_count, (int64_t)0);
    hipLaunchKernelGGL(uniq_collect_kernel, dim3((unsigned)ceil_div(capacity, 256)), dim3(256), 0, s, t, capacity, list_tag,
                       list_slot, count);
    SG_LAUNCH_CHECK();
    hipLaunchKernelGGL(uniq_rank_small_kernel, dim3((unsigned)ceil_div(small_limit, 256)), dim3(256), 0, s, t,
                       (const unsigned long long *)list_tag, (const int32_t *)list_slot, (const unsigned long long *)count,
                       small_limit, out_ukeys, max_unique);
    SG_LAUNCH_CHECK();
    if (n > 0) {   // the element-scan path: every kernel returns at once unless count > small_limit
        const int64_t tiles = ceil_div(n, kUniqTile);
        SG_REQUIRE(tiles < (1ll << 31), SUBGACC_ERR_BADARG, "uniq_number: n too large");
        int32_t *tile_count = (int32_t *)ws;
        ws += align_up((size_t)tiles * 4, 256);
        int64_t *tile_off = (int64_t *)ws;
        ws += align_up((size_t)(tiles + 1) * 8, 256);
        hipLaunchKernelGGL(uniq_count_kernel, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, t, slot, n,
                           (const unsigned long long *)count, small_limit, tile_count);
        SG_LAUNCH_CHECK();
        int rc = exclusive_scan_i32(tile_count, tiles, tile_off, ws, workspace_bytes - (size_t)(ws - (char *)workspace), s);
        if (rc != SUBGACC_OK) return rc;
        hipLaunchKernelGGL(uniq_assign_kernel, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, t, slot, n,
                           (const unsigned long long *)count, small_limit, (const int64_t *)tile_off, out_ukeys, max_unique);
        SG_LAUNCH_CHECK();
    }
    return SUBGACC_OK;
}

extern "C" int subgacc_uniq_translate(void *table, int64_t capacity, int32_t *slot_inout, int64_t n,
                                      const int64_t *n_dev, int32_t add, void *stream) {
    SG_REQUIRE(table && is_pow2(capacity) && capacity < (1ll << 31) && n >= 0, SUBGACC_ERR_BADARG,
               "uniq_translate: bad arguments");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(slot_inout, SUBGACC_ERR_BADARG, "uniq_translate: null slot array");
    hipLaunchKernelGGL(uniq_translate_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       uniq_view(table, capacity), n, n_dev, slot_inout, add);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_unpack_lp(const uint64_t *keys, int64_t n, const int64_t *n_dev, int32_t num_walks,
                                 int32_t num_steps, int16_t *out_i16, int32_t *out_i32, float *out_f32, int32_t zero_row,
                                 void *stream) {
    const int shift = subgacc_key_shift(num_walks, num_steps);
    if (shift < 0) return shift;
    SG_REQUIRE(n >= 0 && (keys || n == 0), SUBGACC_ERR_BADARG, "unpack_lp: bad arguments");
    const int64_t work = n > (num_steps + 1) ? n : (num_steps + 1);  // the zero row is written by the first threads
    if (n == 0 && !(out_f32 && zero_row)) return SUBGACC_OK;
    hipLaunchKernelGGL(unpack_lp_kernel, dim3((unsigned)ceil_div(work, 256)), dim3(256), 0, (hipStream_t)stream, keys, n,
                       n_dev, num_walks, num_steps, shift, out_i16, out_i32, out_f32, zero_row);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
