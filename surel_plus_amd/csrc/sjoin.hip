// sjoin.hip -- SpJoin: the set-join structural encoder of SUREL+ (gfx950).
//
// Replaces the SciPy sparse arithmetic of train.py:13-45 (gather), :48-72 (hgather), :75-111
// (bgather/pgather): `x[edge[0]]`, `xr.multiply(lmask) + lmask`, `.data - 1`, `np.stack`, `encode[...]`.
// Those five temporaries and the host->device upload of the index array collapse into one kernel over a
// device-resident SpG.  A join is a list of segments (own row, partner row); every member of the own row leaves as one output
// row (value of the member, value of the same node in the partner row or "absent").  Kernels:
//   sjoin_seg_reduce / sjoin_seg_scan   segment pointers = exclusive scan of the own rows' lengths (train.py:20-22)
//   sjoin_keypair_kernel                mirrored lists (gather / hgather / nb batches at once): one workgroup per PAIR of rows, the
//                                       longer row staged in LDS, the shorter one in registers, ONE sorted-set search per pair;
//                                       payload = LP key (32 / 64 bits: the feature row is unpacked, count / num_walks by an
//                                       fma-refined reciprocal) or -- TAB -- SFptr+1 / table slot with the Z_SF table
//   sjoin_f64pair_kernel                the same plan for the PPR encoder's float payload (train.py:39-43)
//   sjoin_fill_kernel                   any other list: one wave per segment, the partner row in LDS (or searched in place when it
//                                       does not fit)
//   sjoin_counts_kernel / sjoin_pairs_kernel   the count and pair forms of the join
// One entry point, subgacc_sjoin_fill_v2(descriptor) (ABI 6); the entry points of ABI 1-5 forward to it (end of file).
// The [R,2] index array of the reference never exists in memory unless asked for (out_idx).
#include <cstdlib>
#include "common.hpp"
#include "blockscan.hpp"

namespace subgacc {

constexpr int kJoinThreads = 64;

// Segment pointers = exclusive scan of the own rows' lengths (train.py:20-22), as two kernels (one for <= 2048
// segments): the row length is looked up inside the scan's passes (no length array, no separate look-up kernel), and
// every tile adds up the tile sums in front of it itself (at most a few thousand words) instead of a third launch.
// A row number outside [0, n_rows) -- the reference's `x[edge[0]]` raises IndexError for it (train.py:15) -- is never
// dereferenced: the row counts as empty and flags[3] |= 16 tells the host (which raises).
struct SegLen {
    const int64_t *indptr;
    const int32_t *row_len;
    int64_t n_rows;
    const int64_t *own, *partner;
    int32_t *flags;
    int64_t S;
    const int32_t *row_head = nullptr;      // headed rows (ABI 7): the length of row r is row_head[r * row_stride]
    int64_t row_stride = 0;
    __device__ __forceinline__ int64_t len(int64_t a) const {
        return row_len ? (int64_t)row_len[a] : (row_head ? (int64_t)row_head[a * row_stride] : indptr[a + 1] - indptr[a]);
    }
    __device__ __forceinline__ int64_t operator()(int64_t j, bool flag_it) const {
        if (j >= S) return 0;
        const int64_t a = own[j];
        const bool bad = (uint64_t)a >= (uint64_t)n_rows;
        if (flag_it && (bad || (partner && (uint64_t)partner[j] >= (uint64_t)n_rows)) && flags) atomicOr(&flags[3], 16);
        return bad ? 0 : len(a);
    }
};
#ifndef SJ_SEG_ITEMS      // segments per lane of the two size kernels: 2 (131,072 segments = 256 workgroups; 8 per lane left 3/4 of the CUs idle: 17.8 -> 12 us)
#define SJ_SEG_ITEMS 2
#endif
constexpr int kSegItems = SJ_SEG_ITEMS;
constexpr int kSegTile = kScanThreads * kSegItems;

__global__ __launch_bounds__(kScanThreads) void sjoin_seg_reduce_kernel(const SegLen L, int64_t *__restrict__ partial) {
    const int64_t base = (int64_t)blockIdx.x * kSegTile + (int64_t)threadIdx.x * kSegItems;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kSegItems; ++k) s += L(base + k, true);
    int64_t tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// partial == nullptr: a single tile (which then also raises the flag).  ITEMS = kSegItems for the tiles of a large batch;
// 16 for a batch of up to 4,096 segments (the reference's 1,024 pairs: 2,048 segments) as ONE tile in ONE launch.
constexpr int kSegItemsSmall = 16;
template <int ITEMS>
__global__ __launch_bounds__(kScanThreads) void sjoin_seg_scan_kernel(const SegLen L, const int64_t *__restrict__ partial,
                                                                      int64_t *__restrict__ out) {
    constexpr int kSegItems = ITEMS, kSegTile = kScanThreads * ITEMS;
    const int64_t base = (int64_t)blockIdx.x * kSegTile + (int64_t)threadIdx.x * kSegItems;
    int64_t v[kSegItems];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kSegItems; ++k) {
        v[k] = L(base + k, partial == nullptr);
        s += v[k];
    }
    int64_t front = 0;   // sum of the tiles in front of this one
    if (partial) {
        int64_t mine = 0;
        for (int64_t t = threadIdx.x; t < (int64_t)blockIdx.x; t += kScanThreads) mine += partial[t];
        int64_t ignore = block_exclusive_scan(mine, &front);
        (void)ignore;
    }
    int64_t tot;
    int64_t run = block_exclusive_scan(s, &tot) + front;
#pragma unroll
    for (int k = 0; k < kSegItems; ++k) {
        if (base + k < L.S) out[base + k] = run;
        run += v[k];
        if (base + k == L.S - 1) out[L.S] = run;   // the grand total lands in out[S]
    }
}

// ---- the size pass as ONE launch (subgacc_join_desc::options & SUBGACC_JOIN_OPT_SIZES): a single-pass scan with decoupled
// look-back.  The join of a resident store is short work (65,536 pairs of the top-100 PPR store: 64 us of fill): the 16-byte memset
// of the status words, the two size kernels above and the 24-byte read-back -- four more launches of ~4.5 us each, the floor of any
// launch here -- were a quarter of the call.  This kernel is all four: it scans, ORs its status into flags[3] like any other entry
// point (the status of THIS call, which a caller that never zeroes flags wants, is what host_tail gets) and leaves [R, status] in
// pinned host memory.  What it needs in exchange is state that survives between launches -- a ticket, a count of
// finished tiles, one word per tile -- all zero when a launch starts; the LAST tile to finish (every other tile is past its
// look-back by then) zeroes it again, so the caller zeroes it once, when it allocates it.  Tiles take their number from the ticket
// (a tile only ever waits for tiles that run already); a wait is bounded (kSpinLimit polls, never reached with clean state): a dirty
// state -- a launch that was torn down half way -- ends in status bit 64 instead of a hang (a ticket beyond the tiles, a look-back
// that gives up).  That is a best-effort detector, not a recovery: with `done` or the tile words dirty the tile that believes it
// is the last may not be, so after bit 64 the CALLER zeroes the state (CapturedJoin.finish() does) before the next call.
struct SizeState {
    unsigned long long ticket, done, status, total, pad[4];      // 64 bytes; one word per tile follows
};
constexpr unsigned long long kTileAgg = 1ull << 62, kTilePrefix = 2ull << 62, kTileValue = (1ull << 62) - 1;
constexpr int kSpinLimit = 1 << 20;
#ifndef SJ_ONEPASS_ITEMS      // segments per lane: 131,072 segments take 11.8 / 10.5 / 12.4 / 18.4 us with 2 / 4 / 8 / 16 (profiles/r24_onepass_items.log)
#define SJ_ONEPASS_ITEMS 4
#endif
constexpr int kOnePassItems = SJ_ONEPASS_ITEMS;

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

template <int ITEMS>
__global__ __launch_bounds__(kScanThreads) void sjoin_sizes_onepass_kernel(const SegLen L, int64_t *__restrict__ out,
                                                                           unsigned long long *__restrict__ state,
                                                                           int64_t *__restrict__ host_tail, const int nb) {
    constexpr int kTile = kScanThreads * ITEMS;
    SizeState *hd = (SizeState *)state;
    unsigned long long *tile = state + sizeof(SizeState) / 8;
    __shared__ long long s_word[2];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    if (tid == 0) s_word[0] = (long long)atomicAdd(&hd->ticket, 1ull);
    __syncthreads();
    const long long t = s_word[0];
    if (t < nb) {
        const int64_t base = t * kTile + (int64_t)tid * ITEMS;
        int64_t v[ITEMS], s = 0;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int64_t j = base + k;
            v[k] = 0;
            if (j < L.S) {
                const int64_t a = L.own[j];
                const bool oob = (uint64_t)a >= (uint64_t)L.n_rows;
                bad |= oob || (L.partner && (uint64_t)L.partner[j] >= (uint64_t)L.n_rows);
                if (!oob) v[k] = L.len(a);
            }
            s += v[k];
        }
        if (bad) atomicOr(&hd->status, 16ull);
        int64_t tot;
        int64_t run = block_exclusive_scan(s, &tot);
        if (wid == 0) {
            unsigned long long front = 0;
            bool gave_up = false;
            if (t == 0) {
                if (lane == 0) __hip_atomic_store(&tile[0], kTilePrefix | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0) __hip_atomic_store(&tile[t], kTileAgg | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                for (long long top = t - 1; top >= 0 && !gave_up; top -= kWave) {
                    const long long idx = top - lane;
                    unsigned long long x = kTilePrefix;          // in front of tile 0: the prefix 0
                    if (idx >= 0) x = __hip_atomic_load(&tile[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (__ballot((x >> 62) == 0) != 0ull) {
                        if (++spins > kSpinLimit) {
                            gave_up = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        if ((x >> 62) == 0) x = __hip_atomic_load(&tile[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (gave_up) break;
                    const unsigned long long pre = __ballot((x >> 62) == 2);
                    if (pre != 0ull) {         // the nearest tile that knows its whole prefix ends the walk
                        const int first = __ffsll((long long)pre) - 1;
                        front += wave_sum_u64(lane <= first ? (x & kTileValue) : 0ull);
                        break;
                    }
                    front += wave_sum_u64(x & kTileValue);
                }
                if (lane == 0) {
                    if (gave_up) atomicOr(&hd->status, 64ull);
                    __hip_atomic_store(&tile[t], kTilePrefix | ((front + (unsigned long long)tot) & kTileValue), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (lane == 0) s_word[1] = (long long)front;
        }
        __syncthreads();
        run += s_word[1];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            if (base + k < L.S) out[base + k] = run;
            run += v[k];
            if (base + k == L.S - 1) {     // the grand total lands in out[S] -- and in the state, where the finishing tile finds it
                out[L.S] = run;
                atomicExch(&hd->total, (unsigned long long)run);
            }
        }
        if (L.S == 0 && tid == 0) out[0] = 0;
    } else if (tid == 0) {
        atomicOr(&hd->status, 64ull);      // a ticket beyond the tiles: the state was not zero when this launch began
    }
    // Ordering.  Everything one tile learns from another travels through agent-scope atomics on the state's words (the segment
    // pointers themselves are read by the NEXT kernel only) -- but the state must also be left CLEAN, and that needs an order between
    // different addresses: every store / OR / exchange this tile made on tile[t], status and total has to be performed before the
    // finishing tile zeroes those words.  A workgroup barrier alone does not give that (outside tgsplit mode it does not wait for a
    // lane's global atomics in flight: a late OR could land behind the finishing tile's exchange and leak into the next call -- round
    // 5 relied on it).  So: every wave drains its own memory operations (s_waitcnt vmcnt(0): gfx9 counts stores and atomics without
    // return there too), the barrier collects the waves, and only then thread 0 adds to `done` -- with release / acquire semantics at
    // agent scope, so that the tile which reads gridDim.x - 1 there also has the formal edge: its reads and its zeroing stores come
    // after everything every other tile did before ITS increment.
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0)
        s_word[0] = __hip_atomic_fetch_add(&hd->done, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1;
    __syncthreads();
    if (s_word[0]) {       // every other tile is past its look-back, its status and its total: report, and leave the state as it was found
        if (tid == 0) {
            const unsigned long long st = atomicExch(&hd->status, 0ull);
            const long long total = (long long)atomicExch(&hd->total, 0ull);
            if (L.flags && st) atomicOr(&L.flags[3], (int)st);
            if (host_tail) {
                host_tail[0] = (st & 64) ? -1 : total;
                host_tail[1] = (int64_t)st;
            }
            atomicExch(&hd->ticket, 0ull);
            atomicExch(&hd->done, 0ull);
        }
        for (int i = tid; i < nb; i += kScanThreads) __hip_atomic_store(&tile[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct JoinArgs {
    const int64_t *indptr;
    const int32_t *indices;
    const void *data;  // int32 (SFptr+1) or double (PPR score)
    const int64_t *own, *partner, *seg;
    int64_t S;
    int64_t n_rows;   // rows of the store: own / partner values outside [0, n_rows) read as empty rows, flags[3] |= 16
    const float *table;
    int64_t table_rows;
    int32_t k;
    float *out_xz;
    int32_t *out_idx;
    int64_t *out_segid;
    int32_t max_len;
    int32_t *flags;
    // strided rows (subgacc_sjoin_*_rows): row r = [r*row_stride, +row_len[r]) of indices / data, data = table slots
    const int32_t *row_len;
    int64_t row_stride;
    // headed rows (ABI 7: a resident store on whole 128-byte lines): row r = members [r*row_stride, +row_head[r*row_stride]) of
    // indices / data, where `indices` points ONE WORD behind row_head -- slot 0 of a row's ids holds its length, its members follow
    const int32_t *row_head = nullptr;
    int32_t spec_len = 0;     // strided / headed float rows: members asked for before a row's length is known (sjoin_f64pair_kernel)
    // key rows (subgacc_sjoin_fill_keyrows): the rows' payload is the member's 32-bit LP key; a feature row is its unpacked
    // counts / num_walks (lut[c] = float(c) / float(M), built per workgroup), 0xFFFFFFFF = partner absent -> the zero row
    int32_t key_M, key_m, key_shift;
    const int32_t *slot_id;   // slot -> SFptr (id plane of the numbered table of distinct LP rows); NULL with
    int32_t val_add;          // val_add = 1: the feature table is indexed by slot + 1 itself (row 0 = absent)
    bool sized_here = false;  // the segment pointers come from the size pass of this very call: flags[3] & 64 (its state was not
                              // clean, the pointers mean nothing) ends every workgroup before it derives an address from them
    int64_t pb = 0;           // pair_block of a mirrored list: with partner == NULL the partner of segment j is the own row of its mirror
    int32_t split = 1;        // sjoin_pair_kernel: workgroups per pair (small batches: every one stages both rows and emits
                              // its share of the 64-row spans, so that a batch of ~1,000 pairs still fills the chip)
};

// Workgroups per pair of sjoin_pair_kernel: batches far below the chip's ~4,096 resident workgroups are split in two
// (B = 1,024 pairs: 22 -> 18 us; four or eight parts pay more for the repeated row loads than they gain: 21 / 25 us)
static inline int pair_split(int64_t pairs) {
#ifdef SG_DEV_JOIN_SPLIT      // dev builds only (tools/ab.py: -DSG_DEV_JOIN_SPLIT=n)
    return SG_DEV_JOIN_SPLIT;
#endif
    int sp = 1;
    while (sp < 2 && pairs * sp * 2 <= 4096) sp *= 2;
    return sp;
}

// The join's outputs are written once and read by a later kernel, its SpG rows are read once per pair: non-temporal
// (streaming) accesses keep them from displacing each other in L2 -- measured -12 % on the cit2 batch (0.57 -> 0.50 ms).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_store(float4 *p, const float4 &t) {
#ifdef SJ_DEV_PLAIN_STORES      // dev builds: ordinary (cached) stores for the wide words, tools/join_bench.py
    *p = t;
    return;
#endif
    v4f v;
    v.x = t.x, v.y = t.y, v.z = t.z, v.w = t.w;
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
}
__device__ __forceinline__ void stream_store(float2 *p, const float2 &t) {
#ifdef SJ_DEV_SKIP_F64_STORES   // dev builds: what the float join costs without its stores (tools/ppr_join_probe.py)
    if (t.x != -12345.f) return;
#endif
    v2f v;
    v.x = t.x, v.y = t.y;
    __builtin_nontemporal_store(v, reinterpret_cast<v2f *>(p));
}
__device__ __forceinline__ void stream_store(int2 *p, const int2 &t) {
    v2i v;
    v.x = t.x, v.y = t.y;
    __builtin_nontemporal_store(v, reinterpret_cast<v2i *>(p));
}
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) { return __builtin_nontemporal_load(p); }

// partner row of segment j: given, or -- a mirrored list, block 2t+1 = block 2t with own and partner swapped -- the own row of j's mirror
__device__ __forceinline__ int64_t join_partner(const JoinArgs &a, int64_t j) {
    if (a.partner) return a.partner[j];
    return a.own[((j / a.pb) & 1) ? j - a.pb : j + a.pb];
}

__device__ __forceinline__ void join_row(const JoinArgs &a, int64_t r, int64_t &beg, int64_t &len) {
    if ((uint64_t)r >= (uint64_t)a.n_rows) {   // never dereferenced (sjoin_len_kernel gave it length 0 and raised the flag)
        beg = 0, len = 0;
        return;
    }
    if (a.row_stride) {
        beg = r * a.row_stride;
        len = a.row_len ? a.row_len[r] : a.row_head[beg];
    } else {
        beg = a.indptr[r];
        len = a.indptr[r + 1] - beg;
    }
}

// Emit up to 64 consecutive output rows of one segment (one per lane): look the lane's member up in the
// partner row (binary search over sorted ids held in LDS) and write the feature pairs of the whole 64-row span
// with consecutive lanes on consecutive words.  KV = 4: k == 4, rows move as float4; KV = 0: any k <= 16.
// (the one-segment kernel's emit: lists that are NOT mirrored pairs -- the pair kernels below have their own)
template <bool F64, int KV, typename Val>
__device__ __forceinline__ void emit_rows(const JoinArgs &a, int lane, const int32_t *own_ids, const Val *own_val,
                                          int64_t na, const int32_t *pids, const Val *pval, int nb, int64_t t0,
                                          int64_t o, int64_t segj, int k, int k2, uint32_t magic) {
    const int64_t t = t0 + lane;
    const bool live = t < na;
    int32_t id = 0;
    Val va = 0;
    if (live) {
        id = own_ids[t];
        va = own_val[t];
    }
    int lo = 0, hi = live ? nb : 0;
    SJ_HOOK_SEARCH_RANGE(lo, hi);
    while (lo < hi) {   // sorted-set intersection: lower bound in the partner row
        const int mid = (lo + hi) >> 1;
        if (pids[mid] < id) lo = mid + 1;
        else hi = mid;
    }
    const bool hit = live && lo < nb && pids[lo] == id;
    const int64_t row0 = o + t0;
    if (F64) {
        if (live) {
            // the scipy expression computes (partner value or 0) + 1.0 - 1.0 in double, then casts (train.py:33,39-43)
            const double second = ((hit ? (double)pval[lo] : 0.0) + 1.0) - 1.0;
            float2 v;
            v.x = (float)va;
            v.y = (float)second;
            stream_store(reinterpret_cast<float2 *>(a.out_xz) + row0 + lane, v);
        }
    } else {
        int32_t pa = (int32_t)va, pb = hit ? (int32_t)pval[lo] : 0;
        if (a.out_idx && live) {
            int2 v;
            v.x = pa;
            v.y = pb;
            stream_store(reinterpret_cast<int2 *>(a.out_idx) + row0 + lane, v);
        }
        if (a.out_xz) {
            if (live && ((uint64_t)pa >= (uint64_t)a.table_rows || (uint64_t)pb >= (uint64_t)a.table_rows)) {
                atomicOr(&a.flags[3], 2);  // SFptr outside the table: never read out of bounds
                pa = pb = 0;
            }
            const int nrows = (int)((na - t0) < kWave ? (na - t0) : kWave);
            // the 64 rows of this trip are one contiguous span of the output: fetch each row's (pa, pb) from its
            // owner lane by a wave shuffle so that stores are fully coalesced
            if (KV == 4) {
                const float4 *tab4 = reinterpret_cast<const float4 *>(a.table);
                float4 *dst4 = reinterpret_cast<float4 *>(a.out_xz) + row0 * 2;
#pragma unroll
                for (int rnd = 0; rnd < 2; ++rnd) {
                    const int f = rnd * kWave + lane;   // float4 index inside the span
                    const int r = f >> 1;
                    const int spa = __shfl(pa, r, kWave), spb = __shfl(pb, r, kWave);
                    if (r < nrows) stream_store(dst4 + f, tab4[(f & 1) ? spb : spa]);
                }
            } else {
                float *dst = a.out_xz + row0 * k2;
                const int total = nrows * k2;
                for (int f = lane; f < kWave * k2; f += kWave) {
                    const int r = (int)(((uint32_t)f * magic) >> 20);
                    const int c = f - r * k2;
                    const int spa = __shfl(pa, r, kWave), spb = __shfl(pb, r, kWave);
                    if (f < total)
                        __builtin_nontemporal_store(a.table[(int64_t)(c < k ? spa : spb) * k + (c < k ? c : c - k)], dst + f);
                }
            }
        }
    }
    if (a.out_segid && live) __builtin_nontemporal_store(segj, a.out_segid + row0 + lane);
}

// generic: one wave64 workgroup per segment, partner row staged in LDS, own row streamed from HBM.
// STAGE = false: rows too long for LDS (an adjacency-like SpG with hub rows) -- the partner row is searched where it
// lies (L2); slower per look-up, no bound on the row length.
template <bool F64, int KV, bool STAGE = true>
__global__ __launch_bounds__(kJoinThreads) void sjoin_fill_kernel(const JoinArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    using Val = typename std::conditional<F64, double, int32_t>::type;
    Val *pval = (Val *)lds_raw;                       // [max_len]
    int32_t *pids = (int32_t *)(pval + a.max_len);    // [max_len]

    const int64_t j = xcd_item(blockIdx.x, gridDim.x);
    if (j >= a.S) return;
    if (a.sized_here && (a.flags[3] & 64)) return;
    const int lane = threadIdx.x;
    const int64_t ra = a.own[j], rb = join_partner(a, j);
    int64_t ab, na, bb, nb64;
    join_row(a, ra, ab, na);
    join_row(a, rb, bb, nb64);
    if (nb64 > (STAGE ? (int64_t)a.max_len : (int64_t)0x7FFFFFFF)) {
        if (lane == 0) atomicOr(&a.flags[3], 1);
        return;
    }
    const int nb = (int)nb64;
    const Val *data = (const Val *)a.data;
    if (STAGE) {
        for (int r = lane; r < nb; r += kJoinThreads) {   // partner row -> LDS, coalesced
            pids[r] = a.indices[bb + r];
            pval[r] = data[bb + r];
        }
        __syncthreads();
    } else {
        pids = const_cast<int32_t *>(a.indices + bb);
        pval = const_cast<Val *>(data + bb);
    }
    const int64_t o = a.seg[j];
    const int k = a.k, k2 = 2 * k;
    const uint32_t magic = k2 > 0 ? ((1u << 20) + (uint32_t)k2 - 1u) / (uint32_t)k2 : 0u;   // f / k2 for f < 2^11
    for (int64_t t0 = 0; t0 < na; t0 += kWave)
        emit_rows<F64, KV, Val>(a, lane, a.indices + ab, data + ab, na, pids, pval, nb, t0, o, j, k, k2, magic);
}

// paired: the segment list consists of blocks of `pb` segments where block 2t+1 mirrors block 2t (own and
// partner swapped) -- gather's [u.. | v..] and hgather's [U|w ; W|u ; V|w ; W|v].  One workgroup takes the segment (A,B) and
// its mirror (B,A): both SpG rows are read from HBM once and both output blocks are produced from there (the generic kernel
// above reads every row twice).  sjoin_keypair_kernel (LP keys, and -- TAB -- SFptr / table slots with the Z_SF table) and
// sjoin_f64pair_kernel (the PPR payload) below; rounds 1-4's sjoin_pair_kernel staged BOTH rows and searched in BOTH directions.
constexpr int kPairThreads = 256;
#ifndef SJ_PAIR_THREADS      // lanes of sjoin_pair_kernel's workgroups (tools/ab.py --files=sjoin.hip)
#define SJ_PAIR_THREADS 128   // 128 lanes per pair: twice the pairs with their row loads in flight per CU (-4..6 % against 256, r02s)
#endif
constexpr int kPairEmit = SJ_PAIR_THREADS;
// ---------------------------------------------------------------------------------------------------------
// Key rows (the payload of a member is its LP key, 32 or 64 bits): the join of the on-demand step and of a keyed store.
// Same work split as sjoin_pair_kernel -- one workgroup per mirrored pair, both rows staged in LDS, every wave emits whole
// 64-row spans -- rebuilt around what the timing builds of round 5 showed (tools/join_bench.py, profiles/r18_join_*.log): on
// the 2-hop batches the kernel was neither store- nor read-bound; it spent its time (1) in the dependent chain own[] -> row
// length -> rows before the first useful instruction of every workgroup, behind ~1,000 scalar instructions of 64-bit
// divisions, and (2) in an emit loop made of LDS round trips with bank conflicts: a divergent binary search per output row in
// BOTH directions and a table look-up per unpacked field.  Here:
//   * 32-bit index arithmetic, no division on the common paths (one mirrored block, one workgroup per pair);
//   * strided rows: the first NT members of both rows are requested BEFORE the rows' lengths are known (the row slots exist
//     whatever the length): one level less in the dependent chain;
//   * a match is symmetric, so only the SHORTER row S is searched in the longer one T: a hit hands the searcher's key to the
//     member it found (pk[j], one LDS write), and T's spans are emitted afterwards without any search.  Only T is staged in LDS
//     (12 bytes per member instead of 16 for two rows: more pairs resident per CU); S's members are each looked at by the one
//     lane that loaded them and never leave its registers;
//   * the search is a halving lower bound with a wave-uniform trip count (3 vector instructions + 1 LDS read per level, no
//     exec-mask loop);
//   * count / num_walks is computed, not looked up: q0 = c * (1/M), r = fma(-M, q0, c), q = fma(r, 1/M, q0) IS the correctly
//     rounded quotient for every count a key of num_walks < 4,096 can hold (all 11,184,810 cases checked on the host in
//     tests/test_host_logic_cpu.py, every count of a sweep of num_walks on the device in tests/test_gpu_round5.py); larger
//     num_walks divide (main.py:174's IEEE division either way);
//   * partner absent == key 0, whose unpacked row IS the zero row (flag 0, every count 0): no special case;
//   * every width goes through the per-wave staging area and leaves as aligned 16-byte words.
struct KeyQuot {
    float fm, rcp;
    bool divide;
};
__device__ __forceinline__ float lp_quotient(uint32_t c, const KeyQuot &q) {
    const float a = (float)c;
    if (q.divide) return a / q.fm;
    const float q0 = a * q.rcp;
    return __fmaf_rn(__fmaf_rn(-q.fm, q0, a), q.rcp, q0);
}
template <bool K64>
__device__ __forceinline__ float key_field(typename std::conditional<K64, unsigned long long, uint32_t>::type key, int c, int m, int shift,
                                           const KeyQuot &q) {
    if (c == 0) return (float)(uint32_t)((key >> (m * shift)) & 1u);
    return lp_quotient((uint32_t)(key >> ((m - c) * shift)) & ((1u << shift) - 1u), q);
}

// one 64-row span of the output: the lane's row (own key ka, partner key kb; 0 = absent) unpacked into the wave's staging area at
// the span's offset modulo 16 bytes, then out as aligned 16-byte words (a head / tail of up to three floats from one lane each)
// TAB: the payload is not a key but SFptr+1 (0 = absent = the table's zero row): the two feature rows are read from the Z_SF table
// (a few KB..MB, L2-resident; KV = 4: one 16-byte read each), the index pair itself leaves through out_idx when asked for
template <int KV, bool K64, bool TAB, typename Key>
__device__ __forceinline__ void emit_key_span(const JoinArgs &a, int lane, bool live, Key ka, Key kb, int nrows, int64_t row0, int64_t segj,
                                              int kc, const KeyQuot &q, float *stage) {
    const int w = 2 * kc, m = kc - 1, shift = a.key_shift;
    if (TAB) {
        if (a.out_idx && live) {
            int2 v;
            v.x = (int32_t)ka, v.y = (int32_t)kb;
            stream_store(reinterpret_cast<int2 *>(a.out_idx) + row0 + lane, v);
        }
        if (!a.out_xz) {      // index pairs only: no table is consulted
            if (a.out_segid && live) __builtin_nontemporal_store(segj, a.out_segid + row0 + lane);
            return;
        }
        if (live && ((uint64_t)ka >= (uint64_t)a.table_rows || (uint64_t)kb >= (uint64_t)a.table_rows)) {
            atomicOr(&a.flags[3], 2);  // SFptr outside the table: never read out of bounds
            ka = kb = 0;
        }
    }
    float *dst = a.out_xz + row0 * w;
    const int mis = (int)(((uintptr_t)dst >> 2) & 3);       // floats between the 16-byte boundary in front and the span
    const int head = (4 - mis) & 3;                         // floats of the span in front of its first aligned word
    if (live) {
        float *mine = stage + mis + lane * w;
        if (TAB && KV == 4) {
            const float4 fa = reinterpret_cast<const float4 *>(a.table)[(uint32_t)ka], fb = reinterpret_cast<const float4 *>(a.table)[(uint32_t)kb];
            mine[0] = fa.x, mine[1] = fa.y, mine[2] = fa.z, mine[3] = fa.w;
            mine[4] = fb.x, mine[5] = fb.y, mine[6] = fb.z, mine[7] = fb.w;
        } else if (TAB && KV > 0) {      // a compile-time width: all 2*KV reads leave together, then the writes
            const float *ta = a.table + (int64_t)(uint32_t)ka * KV, *tb = a.table + (int64_t)(uint32_t)kb * KV;
            float f[2 * (KV > 0 ? KV : 1)];
#pragma unroll
            for (int c = 0; c < KV; ++c) f[c] = ta[c], f[KV + c] = tb[c];
#pragma unroll
            for (int c = 0; c < 2 * KV; ++c) mine[c] = f[c];
        } else if (TAB) {
            const float *ta = a.table + (int64_t)(uint32_t)ka * kc, *tb = a.table + (int64_t)(uint32_t)kb * kc;
            for (int c = 0; c < kc; ++c) {
                mine[c] = ta[c];
                mine[kc + c] = tb[c];
            }
        } else if (KV > 0) {
            float f[2 * (KV > 0 ? KV : 1)];
#pragma unroll
            for (int c = 0; c < KV; ++c) {
                f[c] = key_field<K64>(ka, c, m, shift, q);
                f[KV + c] = key_field<K64>(kb, c, m, shift, q);
            }
#pragma unroll
            for (int c = 0; c < 2 * KV; ++c) mine[c] = f[c];
        } else
            for (int c = 0; c < kc; ++c) {
                mine[c] = key_field<K64>(ka, c, m, shift, q);
                mine[kc + c] = key_field<K64>(kb, c, m, shift, q);
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int total = nrows * w;                            // floats of the span
    const int nbody = SJ_HOOK_SPAN_STORES(total > head ? (total - head) >> 2 : 0);       // aligned 16-byte words
    const int ntail = total > head ? (total - head) & 3 : 0;
    const float *src = stage + mis;                         // float i of the span
    const float4 *src4 = reinterpret_cast<const float4 *>(src + head);   // stage, or stage + 4: 16-byte aligned
    float4 *dst4 = reinterpret_cast<float4 *>(dst + head);
    if (mis | ntail) {                                      // (never for rows of 16 or 32 bytes in an aligned buffer)
        const int i = kWave - 1 - lane;                     // the last lanes have the fewest body words
        if (i < head && i < total) __builtin_nontemporal_store(src[i], dst + i);
        const int t2 = kWave - 4 - lane, at = head + 4 * nbody;
        if (t2 >= 0 && t2 < ntail) __builtin_nontemporal_store(src[at + t2], dst + at + t2);
    }
    if (KV > 0) {
#pragma unroll
        for (int qd = 0; qd < (KV + 1) / 2; ++qd) {         // 64 rows x 2*KV floats = 32*KV words
            const int f = lane + qd * kWave;
            if (f < nbody) stream_store(dst4 + f, src4[f]);
        }
    } else
        for (int f = lane; f < nbody; f += kWave) stream_store(dst4 + f, src4[f]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next span of this wave re-uses the area
    __builtin_amdgcn_wave_barrier();
    if (a.out_segid && live) __builtin_nontemporal_store(segj, a.out_segid + row0 + lane);
}

template <int KV, int NT, bool K64, bool TAB = false>
__global__ __launch_bounds__(NT) void sjoin_keypair_kernel(const JoinArgs a, uint32_t pb, uint32_t pairs) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    static_assert(!(K64 && TAB), "a table row number is 32 bits");
    using Key = typename std::conditional<K64, unsigned long long, uint32_t>::type;
    constexpr int NW = NT / kWave;
    // TAB: what a member carries becomes SFptr+1 on its way in -- a slot of the numbered table of distinct rows through its id plane
    // (strided rows of the table form), or the payload + val_add (0: packed rows hold SFptr+1; 1: the table is indexed by slot + 1)
    const bool xl = TAB && a.slot_id != nullptr;
    auto sfptr = [&](Key v) -> Key {
        if (!TAB) return v;
        return xl ? (Key)(a.slot_id[(int32_t)v] + 1) : (Key)((int32_t)v + a.val_add);
    };
    const int ML = a.max_len;
    // only T, the LONGER row of the pair, is staged: S's members are looked at by exactly one lane each and stay in registers
    Key *valT = (Key *)lds_raw;                       // [max_len] keys of T
    Key *pk = valT + ML;                              // [max_len] partner keys of T's members (0 = absent)
    int32_t *idsT = (int32_t *)(pk + ML);             // [max_len]
    const int kc = KV > 0 ? KV : a.k, w = 2 * kc;     // floats per output row
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);      // (scalar: everything a span derives from it stays in SGPRs)
    // [4 + 64 rows x 2k] floats per wave, every area on a 16-byte boundary (a span is staged at its output's offset mod 16)
    const size_t stage_off = ((size_t)ML * (2 * sizeof(Key) + 4) + 15) & ~(size_t)15;
    float *stage = (float *)(lds_raw + stage_off) + wave * (kWave * w + 4);

    SJ_HOOK_PAIR_ENTRY();
    if (a.sized_here && (a.flags[3] & 64)) return;
    const uint32_t wg = (uint32_t)(blockIdx.x & (kXcds - 1)) * (gridDim.x / kXcds) + (blockIdx.x / kXcds);    // xcd_item, 32 bits
    uint32_t p = wg, part = 0;
    if (a.split > 1) {
        p = wg / (uint32_t)a.split;
        part = wg - p * (uint32_t)a.split;
    }
    if (p >= pairs) return;
    uint32_t blk = 0, off = p;
    if (pb != pairs) {                 // several mirrored blocks (nb batches in one launch)
        blk = p / pb;
        off = p - blk * pb;
    }
    const int64_t j = (int64_t)blk * 2 * pb + off, j2 = j + pb;
    const int64_t ra = a.own[j];
    int64_t rb;
    if (a.partner) {
        rb = a.partner[j];
        if (a.own[j2] != rb || a.partner[j2] != ra) {   // not a mirrored pair: the caller broke the precondition
            if (tid == 0) atomicOr(&a.flags[3], 4);
            return;
        }
    } else
        rb = a.own[j2];
    const int64_t oA = a.seg[j], oB = a.seg[j2];
    const bool okA = (uint64_t)ra < (uint64_t)a.n_rows, okB = (uint64_t)rb < (uint64_t)a.n_rows;   // else: an empty row, never dereferenced
    const bool same = ra == rb;
    const Key *keys = (const Key *)a.data;
    int64_t ab = 0, bb = 0;
    int na = 0, nb = 0;
    int32_t idA0 = 0, idB0 = 0;
    Key kA0 = 0, kB0 = 0;
    if (a.row_stride) {    // strided / headed rows: ask for the first NT members of both rows now, for their lengths next
        ab = ra * a.row_stride, bb = rb * a.row_stride;
        const bool in = tid < ML;
        if (okA && in) {
            SJ_HOOK_FIRST_TRIP(idA0, kA0, tid) {
                idA0 = stream_load(&a.indices[ab + tid]);
                kA0 = stream_load(&keys[ab + tid]);
            }
        }
        if (okB && !same && in) {
            SJ_HOOK_FIRST_TRIP(idB0, kB0, tid) {
                idB0 = stream_load(&a.indices[bb + tid]);
                kB0 = stream_load(&keys[bb + tid]);
            }
        }
        na = okA ? (a.row_len ? a.row_len[ra] : a.row_head[ab]) : 0;       // (headed: the word in front of the members asked for above)
        nb = okB ? (a.row_len ? a.row_len[rb] : a.row_head[bb]) : 0;
    } else {
        int64_t na64 = 0, nb64 = 0;
        if (okA) {
            ab = a.indptr[ra];
            na64 = a.indptr[ra + 1] - ab;
        }
        if (okB) {
            bb = a.indptr[rb];
            nb64 = a.indptr[rb + 1] - bb;
        }
        na = na64 > ML ? ML + 1 : (int)na64, nb = nb64 > ML ? ML + 1 : (int)nb64;
        if (tid < na && na <= ML) {
            SJ_HOOK_FIRST_TRIP(idA0, kA0, tid) {
                idA0 = stream_load(&a.indices[ab + tid]);
                kA0 = stream_load(&keys[ab + tid]);
            }
        }
        if (!same && tid < nb && nb <= ML) {
            SJ_HOOK_FIRST_TRIP(idB0, kB0, tid) {
                idB0 = stream_load(&a.indices[bb + tid]);
                kB0 = stream_load(&keys[bb + tid]);
            }
        }
    }
    KeyQuot q;
    q.fm = (float)a.key_M, q.rcp = 1.0f / q.fm, q.divide = a.key_M >= 4096;
    if (na > ML || nb > ML) {
        if (tid == 0) atomicOr(&a.flags[3], 1);
        return;
    }
    if (TAB) {      // (only now: a first trip asked for before the lengths were known may hold anything past the row's end)
        kA0 = tid < na ? sfptr(kA0) : (Key)0;
        kB0 = (!same && tid < nb) ? sfptr(kB0) : (Key)0;
    }
    if (same) idB0 = idA0, kB0 = kA0;      // (u,u): the second row is the first
    // roles: S = the shorter row, searched member by member in T = the longer one
    const bool swap = na > nb;
    const int ns = swap ? nb : na, nt = swap ? na : nb;
    const int64_t sb = swap ? bb : ab, tb = swap ? ab : bb;           // where the rows begin in indices / keys
    const int64_t oS = swap ? oB : oA, oT = swap ? oA : oB, jS = swap ? j2 : j, jT = swap ? j : j2;
    // Span c of S = members [64c, 64c + 64) = trip c / NW of wave c % NW: the members a lane needs are the ones it loads itself.
    // Up to kRegTrips trips of S live in registers (rows of up to kRegTrips * NT members: every shape the walk kernels emit);
    // longer rows take the rest span by span (below).  Trips 1.. of S and of T are asked for together: one round trip.
    constexpr int kRegTrips = 4;
    int32_t sid[kRegTrips];
    Key skey[kRegTrips], sgot[kRegTrips];
    sid[0] = swap ? idB0 : idA0, skey[0] = swap ? kB0 : kA0;
    {
        int32_t ti[kRegTrips - 1];
        Key tk[kRegTrips - 1];
#pragma unroll
        for (int u = 1; u < kRegTrips; ++u) {
            const int r = tid + u * NT;
            sid[u] = 0, skey[u] = 0, ti[u - 1] = 0, tk[u - 1] = 0;
            if (r < ns) {
                SJ_HOOK_FIRST_TRIP(sid[u], skey[u], r) {
                    sid[u] = stream_load(&a.indices[sb + r]);
                    skey[u] = sfptr(stream_load(&keys[sb + r]));
                }
            }
            if (r < nt) {
                SJ_HOOK_FIRST_TRIP(ti[u - 1], tk[u - 1], r) {
                    ti[u - 1] = stream_load(&a.indices[tb + r]);
                    tk[u - 1] = sfptr(stream_load(&keys[tb + r]));
                }
            }
        }
        if (tid < nt) {
            idsT[tid] = swap ? idA0 : idB0;
            valT[tid] = swap ? kA0 : kB0;
            pk[tid] = 0;
        }
#pragma unroll
        for (int u = 1; u < kRegTrips; ++u) {
            const int r = tid + u * NT;
            if (r < nt) {
                idsT[r] = ti[u - 1];
                valT[r] = tk[u - 1];
                pk[r] = 0;
            }
        }
    }
    for (int r = tid + kRegTrips * NT; r < nt; r += NT) {
        SJ_HOOK_ROW_LOAD(idsT, valT, tb, r, 3);
        idsT[r] = stream_load(&a.indices[tb + r]);
        valT[r] = sfptr(stream_load(&keys[tb + r]));
        pk[r] = 0;
    }
    __syncthreads();
    SJ_HOOK_PAIR_ROWS_READY();

    // ---- search: every member of S in T (sorted-set intersection: the last member of T that is <= id, by halving; n is
    //      wave-uniform, the trips of a lane are independent chains that share every LDS round trip).  A hit hands the member's
    //      key to the member it found.  With `split` workgroups per pair every one of them searches all of S: its LDS needs every hit.
    const int chunksS = (ns + kWave - 1) / kWave, chunksT = (nt + kWave - 1) / kWave;
    const bool whole = a.split == 1;
    {
        int b[kRegTrips];
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) b[u] = 0;
        int n = nt;
        SJ_HOOK_SEARCH_RANGE(b[0], n);
        const int trips = (chunksS - wave + NW - 1) / NW;       // spans of S this wave holds (<= kRegTrips of them in registers)
        while (n > 1) {
            const int h = n >> 1;
#pragma unroll
            for (int u = 0; u < kRegTrips; ++u)
                if (u < trips) b[u] = idsT[b[u] + h] <= sid[u] ? b[u] + h : b[u];
            n -= h;
        }
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) {
            sgot[u] = 0;
            if (u < trips) {
                const int32_t f = idsT[b[u]];          // (T without members: slot 0 of its LDS array, never used)
                const Key g = valT[b[u]];
                const bool hit = (wave + u * NW) * kWave + lane < ns && n == 1 && f == sid[u];
                if (hit) pk[b[u]] = skey[u], sgot[u] = g;
            }
        }
    }
    // rows longer than the register trips hold: their later spans one by one, searched and emitted on the spot
    for (int c = wave + kRegTrips * NW; c < chunksS; c += NW) {
        const int t0 = c * kWave;
        const bool live = t0 + lane < ns;
        int32_t id = 0;
        Key key = 0;
        if (live) {
            id = stream_load(&a.indices[sb + t0 + lane]);
            key = sfptr(stream_load(&keys[sb + t0 + lane]));
        }
        int bx = 0, n = nt;
        SJ_HOOK_SEARCH_RANGE(bx, n);
        while (n > 1) {
            const int h = n >> 1;
            bx = idsT[bx + h] <= id ? bx + h : bx;
            n -= h;
        }
        const int32_t f = idsT[bx];
        const Key g = valT[bx];
        const bool hit = live && n == 1 && f == id;
        if (hit) pk[bx] = key;
        if (whole || (uint32_t)(c / NW) % (uint32_t)a.split == part)
            emit_key_span<KV, K64, TAB, Key>(a, lane, live, key, hit ? g : (Key)0, ns - t0 < kWave ? ns - t0 : kWave, oS + t0, jS, kc, q, stage);
    }
    __syncthreads();
    // ---- emit: S's spans out of the registers, then T's (every member's partner key is in pk), dealt so that the waves with
    //      fewer spans of S take more of T
#pragma unroll
    for (int u = 0; u < kRegTrips; ++u) {
        const int c = wave + u * NW, t0 = c * kWave;
        if (c < chunksS && (whole || (uint32_t)u % (uint32_t)a.split == part))
            emit_key_span<KV, K64, TAB, Key>(a, lane, t0 + lane < ns, skey[u], sgot[u], ns - t0 < kWave ? ns - t0 : kWave, oS + t0, jS, kc, q, stage);
    }
    const int rot = (NW - chunksS % NW) % NW;       // T's span c goes to wave (c + chunksS) % NW: the round robin simply goes on
    for (int c = (wave + rot) % NW + (int)part * NW; c < chunksT; c += a.split * NW) {
        const int t0 = c * kWave;
        const bool live = t0 + lane < nt;
        const int i = live ? t0 + lane : t0;
        emit_key_span<KV, K64, TAB, Key>(a, lane, live, valT[i], pk[i], nt - t0 < kWave ? nt - t0 : kWave, oT + t0, jT, kc, q, stage);
    }
}

// (Round 5 also measured Q consecutive pairs per workgroup, software-pipelined -- the pairs' row numbers, offsets and lengths read
// once per workgroup, pair q+1's first members on their way into registers while pair q is searched and emitted, two LDS buffers
// in turn: 3-17 % SLOWER than one workgroup per pair on every workload at Q = 4, 8, 16 and with 128 or 256 lanes,
// profiles/r18_join_pipe.log.  The hardware's own interleaving of ~16 resident workgroups per CU already hides the start-up chain;
// fewer, longer-lived workgroups only add barriers.  The code is not kept.)

// ---------------------------------------------------------------------------------------------------------
// Float payload (the PPR encoder's store, train.py:39-43): the same pair join as sjoin_keypair_kernel -- the longer row T staged in
// LDS with a slot per member for its partner's value, the shorter row S in the registers of the lanes that loaded it, one halving
// search of S in T, a hit hands S's value over -- with 8-byte payloads and a two-float output row that needs no staging:
// xz[row] = (float(own), float((partner or 0.0) + 1.0 - 1.0)), the SciPy expression of train.py:33 evaluated in double.
// (Measured and not kept, round 5: 2 / 4 / 8 one-wave pairs per workgroup, every wave on its own pair and its own slice of LDS, no
// barrier between them -- 65,536 one-wave workgroups take 17 us to start when they do nothing else: 66.6 / 66.8-68 / 67.8 us against
// 65.8-66.1, profiles/r24_ppr_pairs_per_wg.log.  Starting the workgroups is hidden behind the ones that run; timing builds
// (profiles/r24_ppr_join_experiments.log: 67 us; 38 without the row loads, 52 without the stores, 37 without both, 55 without the
// search) and the counters (HBM traffic 1.31x the algorithmic bytes = 4.9 TB/s of what a copy reaches here) say the rest.)
template <int NT>
__global__ __launch_bounds__(NT) void sjoin_f64pair_kernel(const JoinArgs a, uint32_t pb, uint32_t pairs) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int NW = NT / kWave;
    const int ML = a.max_len;
    double *valT = (double *)lds_raw;                 // [max_len] values of T
    double *pv = valT + ML;                           // [max_len] partner values of T's members (0.0 = absent)
    int32_t *idsT = (int32_t *)(pv + ML);             // [max_len]
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);

    SJ_HOOK_PAIR_ENTRY();
    if (a.sized_here && (a.flags[3] & 64)) return;
    const uint32_t wg = (uint32_t)(blockIdx.x & (kXcds - 1)) * (gridDim.x / kXcds) + (blockIdx.x / kXcds);    // xcd_item, 32 bits
    uint32_t p = wg, part = 0;
    if (a.split > 1) {
        p = wg / (uint32_t)a.split;
        part = wg - p * (uint32_t)a.split;
    }
    if (p >= pairs) return;
    uint32_t blk = 0, off = p;
    if (pb != pairs) {                 // several mirrored blocks (nb batches in one launch)
        blk = p / pb;
        off = p - blk * pb;
    }
    const int64_t j = (int64_t)blk * 2 * pb + off, j2 = j + pb;
    const double *vals = (const double *)a.data;
    const int64_t ra = a.own[j];
    int64_t rb;
    if (a.partner) {
        rb = a.partner[j];
        if (a.own[j2] != rb || a.partner[j2] != ra) {   // not a mirrored pair: the caller broke the precondition
            if (tid == 0) atomicOr(&a.flags[3], 4);
            return;
        }
    } else
        rb = a.own[j2];
    const int64_t oA = a.seg[j], oB = a.seg[j2];
    const bool okA = (uint64_t)ra < (uint64_t)a.n_rows, okB = (uint64_t)rb < (uint64_t)a.n_rows;   // else: an empty row, never dereferenced
    int64_t ab = 0, bb = 0, na64 = 0, nb64 = 0;
    constexpr int kRegTrips = 4;       // trips of S in registers (rows of up to 4 * NT members; longer ones span by span below)
    int32_t sid[kRegTrips];
    double sval[kRegTrips], sgot[kRegTrips];
    int ns, nt;
    int64_t sb, tb, oS, oT, jS, jT;
    if (a.row_stride) {
        // Strided / headed rows: a row's slot exists whatever its length, so its first a.spec_len members are asked for NOW, together
        // with its length (the word in front of them) -- own[] -> {length, members}: two dependent round trips where packed rows need
        // three (own[] -> row pointers -> members).  This kernel is bound by exactly that chain: 8 one-wave pairs per SIMD, each
        // waiting for its next answer (profiles/r24_ppr_join_experiments.log).  spec_len is the store's typical row length rounded to
        // whole lines of ids (HeadedSpG: 96 for the top-100 PPR store): what lies behind it -- few rows have it -- is asked for once
        // the length is known; a shorter row's speculative tail is read for nothing (its own slot: never out of bounds).
        ab = ra * a.row_stride, bb = rb * a.row_stride;
        const int spec = a.spec_len < ML ? a.spec_len : ML;
        int32_t ia[kRegTrips], ib[kRegTrips];
        double va[kRegTrips], vb[kRegTrips];
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) {
            const int r = tid + u * NT;
            ia[u] = ib[u] = 0, va[u] = vb[u] = 0.0;
            if (u * NT < spec && r < spec) {
                if (okA) ia[u] = stream_load(&a.indices[ab + r]), va[u] = stream_load(&vals[ab + r]);
                if (okB && ra != rb) ib[u] = stream_load(&a.indices[bb + r]), vb[u] = stream_load(&vals[bb + r]);
            }
        }
        if (okA) na64 = a.row_len ? a.row_len[ra] : a.row_head[ab];
        if (okB) nb64 = a.row_len ? a.row_len[rb] : a.row_head[bb];
        if (na64 > ML || nb64 > ML) {
            if (tid == 0) atomicOr(&a.flags[3], 1);
            return;
        }
        const int na = (int)na64, nb = (int)nb64;
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) {      // what lies behind the speculative part
            const int r = tid + u * NT;
            if (r >= spec) {
                if (r < na) ia[u] = stream_load(&a.indices[ab + r]), va[u] = stream_load(&vals[ab + r]);
                if (r < nb && ra != rb) ib[u] = stream_load(&a.indices[bb + r]), vb[u] = stream_load(&vals[bb + r]);
            }
        }
        if (ra == rb) {
#pragma unroll
            for (int u = 0; u < kRegTrips; ++u) ib[u] = ia[u], vb[u] = va[u];
        }
        // roles: S = the shorter row, searched member by member in T = the longer one ((u,u): S and T are the same row)
        const bool swap = na > nb;
        ns = swap ? nb : na, nt = swap ? na : nb;
        sb = swap ? bb : ab, tb = swap ? ab : bb;
        oS = swap ? oB : oA, oT = swap ? oA : oB, jS = swap ? j2 : j, jT = swap ? j : j2;
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) {
            const int r = tid + u * NT;
            sid[u] = swap ? ib[u] : ia[u], sval[u] = swap ? vb[u] : va[u];
            if (r >= ns) sid[u] = 0, sval[u] = 0.0;          // (a speculative read behind the row's end holds anything)
            if (r < nt) {
                idsT[r] = swap ? ia[u] : ib[u];
                valT[r] = swap ? va[u] : vb[u];
                pv[r] = 0.0;
            }
        }
    } else {
        if (okA) {
            ab = a.indptr[ra];
            na64 = a.indptr[ra + 1] - ab;
        }
        if (okB) {
            bb = a.indptr[rb];
            nb64 = a.indptr[rb + 1] - bb;
        }
        if (na64 > ML || nb64 > ML) {
            if (tid == 0) atomicOr(&a.flags[3], 1);
            return;
        }
        const int na = (int)na64, nb = (int)nb64;
        // roles: S = the shorter row, searched member by member in T = the longer one ((u,u): S and T are the same row)
        const bool swap = na > nb;
        ns = swap ? nb : na, nt = swap ? na : nb;
        sb = swap ? bb : ab, tb = swap ? ab : bb;
        oS = swap ? oB : oA, oT = swap ? oA : oB, jS = swap ? j2 : j, jT = swap ? j : j2;
        int32_t ti[kRegTrips];
        double tv[kRegTrips];
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) {      // every trip of both rows asked for together: one round trip
            const int r = tid + u * NT;
            sid[u] = 0, sval[u] = 0.0, ti[u] = 0, tv[u] = 0.0;
            if (r < ns) {
                SJ_HOOK_FIRST_TRIP(sid[u], sval[u], r) {
                    sid[u] = stream_load(&a.indices[sb + r]);
                    sval[u] = stream_load(&vals[sb + r]);
                }
            }
            if (r < nt) {
                SJ_HOOK_FIRST_TRIP(ti[u], tv[u], r) {
                    ti[u] = stream_load(&a.indices[tb + r]);
                    tv[u] = stream_load(&vals[tb + r]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) {
            const int r = tid + u * NT;
            if (r < nt) {
                idsT[r] = ti[u];
                valT[r] = tv[u];
                pv[r] = 0.0;
            }
        }
    }
    for (int r = tid + kRegTrips * NT; r < nt; r += NT) {
        idsT[r] = stream_load(&a.indices[tb + r]);
        valT[r] = stream_load(&vals[tb + r]);
        pv[r] = 0.0;
    }
    __syncthreads();
    SJ_HOOK_PAIR_ROWS_READY();
    const int chunksS = (ns + kWave - 1) / kWave, chunksT = (nt + kWave - 1) / kWave;
    const bool whole = a.split == 1;
    float2 *xz = reinterpret_cast<float2 *>(a.out_xz);
    {
        int b[kRegTrips];
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) b[u] = 0;
        int n = nt;
        SJ_HOOK_SEARCH_RANGE(b[0], n);
        const int trips = (chunksS - wave + NW - 1) / NW;
        while (n > 1) {
            const int h = n >> 1;
#pragma unroll
            for (int u = 0; u < kRegTrips; ++u)
                if (u < trips) b[u] = idsT[b[u] + h] <= sid[u] ? b[u] + h : b[u];
            n -= h;
        }
#pragma unroll
        for (int u = 0; u < kRegTrips; ++u) {
            sgot[u] = 0.0;
            if (u < trips) {
                const int32_t f = idsT[b[u]];
                const double g = valT[b[u]];
                const bool hit = (wave + u * NW) * kWave + lane < ns && n == 1 && f == sid[u];
                if (hit) pv[b[u]] = sval[u], sgot[u] = g;
            }
        }
    }
    for (int c = wave + kRegTrips * NW; c < chunksS; c += NW) {      // rows longer than the register trips hold
        const int t0 = c * kWave;
        const bool live = t0 + lane < ns;
        int32_t id = 0;
        double v = 0.0;
        if (live) id = stream_load(&a.indices[sb + t0 + lane]), v = stream_load(&vals[sb + t0 + lane]);
        int bx = 0, n = nt;
        SJ_HOOK_SEARCH_RANGE(bx, n);
        while (n > 1) {
            const int h = n >> 1;
            bx = idsT[bx + h] <= id ? bx + h : bx;
            n -= h;
        }
        const int32_t f = idsT[bx];
        const double g = valT[bx];
        const bool hit = live && n == 1 && f == id;
        if (hit) pv[bx] = v;
        if (live && (whole || (uint32_t)(c / NW) % (uint32_t)a.split == part)) {
            // the scipy expression computes (partner value or 0) + 1.0 - 1.0 in double, then casts (train.py:33,39-43)
            float2 o;
            o.x = (float)v, o.y = (float)(((hit ? g : 0.0) + 1.0) - 1.0);
            stream_store(xz + oS + t0 + lane, o);
            if (a.out_segid) __builtin_nontemporal_store(jS, a.out_segid + oS + t0 + lane);
        }
    }
    // S's spans leave right away (nobody waits for them); T's after the barrier that completes pv
#pragma unroll
    for (int u = 0; u < kRegTrips; ++u) {
        const int64_t t = (wave + u * NW) * kWave + lane;
        if (t < ns && (whole || (uint32_t)u % (uint32_t)a.split == part)) {
            float2 o;
            o.x = (float)sval[u], o.y = (float)((sgot[u] + 1.0) - 1.0);
            stream_store(xz + oS + t, o);
            if (a.out_segid) __builtin_nontemporal_store(jS, a.out_segid + oS + t);
        }
    }
    __syncthreads();
    const int rot = (NW - chunksS % NW) % NW;       // T's span c goes to wave (c + chunksS) % NW: the round robin simply goes on
    for (int c = (wave + rot) % NW + (int)part * NW; c < chunksT; c += a.split * NW) {
        const int t = c * kWave + lane;
        if (t < nt) {
            float2 o;
            o.x = (float)valT[t], o.y = (float)((pv[t] + 1.0) - 1.0);
            stream_store(xz + oT + t, o);
            if (a.out_segid) __builtin_nontemporal_store(jT, a.out_segid + oT + t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Count form of the join ("next" row f.1 of SURVEY.md section 8: SpJoin fused with the first model stage).
// The reference's Net.forward (model.py:78-83) embeds both feature slots of every output row with the same MLP
// and, for mean aggregation, sums the rows of a segment: segment_sum_j = sum_p C[j,p] * MLP(Z_SF[p]) with
// C[j,p] = how often LP row p occurs in either slot of segment j.  This kernel writes C (dense, one row per
// segment) instead of xz: Z_SF has only c+1 distinct rows, so the [R,2,k] tensor (and the [R,2,H] activations
// behind it) collapse into one [S, c+1] x [c+1, H] GEMM.  Slot value 0 (partner absent) is counted too: the MLP
// of the zero row is not zero.  Mirrored segments are produced together, as in sjoin_pair_kernel.
__global__ __launch_bounds__(kPairThreads) void sjoin_counts_kernel(const JoinArgs a, int64_t pb, float *__restrict__ out_counts) {
    // The plan of sjoin_keypair_kernel: only the longer row T of the pair is staged; the shorter row S is searched in it member by
    // member (one direction: a match is symmetric), and a hit counts for both blocks.  Per block: every own value once, every
    // partner value of a hit once, and "partner absent" (row 0) for the members without one -- n - hits, added when the row is
    // written, not one LDS atomic per member on one address.
    extern __shared__ __align__(16) unsigned char lds_raw[];
    int32_t *valT = (int32_t *)lds_raw;               // [max_len]
    int32_t *idsT = valT + a.max_len;                 // [max_len]
    int32_t *histS = idsT + a.max_len;                // [table_rows]
    int32_t *histT = histS + a.table_rows;            // [table_rows]
    int32_t *nhit = histT + a.table_rows;             // [1]

    const int64_t p = xcd_item(blockIdx.x, gridDim.x);
    if (p >= a.S / 2) return;
    const int64_t j = (p / pb) * 2 * pb + (p % pb), j2 = j + pb;
    const int tid = threadIdx.x;
    const int64_t ra = a.own[j], rb = join_partner(a, j);
    if (a.own[j2] != rb || join_partner(a, j2) != ra) {
        if (tid == 0) atomicOr(&a.flags[3], 4);
        return;
    }
    int64_t ab, na64, bb, nb64;
    join_row(a, ra, ab, na64);
    join_row(a, rb, bb, nb64);
    if (na64 > a.max_len || nb64 > a.max_len) {
        if (tid == 0) atomicOr(&a.flags[3], 1);
        return;
    }
    const bool swap = na64 > nb64;
    const int ns = (int)(swap ? nb64 : na64), nt = (int)(swap ? na64 : nb64);
    const int64_t sb = swap ? bb : ab, tb = swap ? ab : bb, jS = swap ? j2 : j, jT = swap ? j : j2;
    const int32_t *data = (const int32_t *)a.data;
    const int rows = (int)a.table_rows;
    // S's first members are asked for before anything else: they are on their way while T is staged and the histograms are cleared
    constexpr int kTrips = 2;
    int32_t sid[kTrips], sval[kTrips];
#pragma unroll
    for (int u = 0; u < kTrips; ++u) {
        const int r = tid + u * kPairThreads;
        sid[u] = 0, sval[u] = 0;
        if (r < ns) sid[u] = stream_load(&a.indices[sb + r]), sval[u] = stream_load(&data[sb + r]);
    }
    for (int x = tid; x < 2 * rows + 1; x += kPairThreads) histS[x] = 0;   // histS, histT and nhit are contiguous
    for (int r = tid; r < nt; r += kPairThreads) {
        idsT[r] = stream_load(&a.indices[tb + r]);
        valT[r] = stream_load(&data[tb + r]);
    }
    __syncthreads();
    for (int r = tid; r < nt; r += kPairThreads) {      // T's own values
        const int32_t v = valT[r];
        if ((uint32_t)v >= (uint32_t)rows) atomicOr(&a.flags[3], 2);  // SFptr outside the table: never counted out of bounds
        else atomicAdd(&histT[v], 1);
    }
    int hits = 0;
    for (int r0 = 0; r0 < ns; r0 += kPairThreads) {     // S: own value, and -- on a hit -- one partner value for each block
        const int r = r0 + tid, u = r0 / kPairThreads;
        if (r >= ns) break;
        int32_t id, v;
        if (u < kTrips) {
            id = u == 0 ? sid[0] : sid[1];
            v = u == 0 ? sval[0] : sval[1];
        } else {
            id = stream_load(&a.indices[sb + r]);
            v = stream_load(&data[sb + r]);
        }
        int b = 0, n = nt;
        while (n > 1) {
            const int h = n >> 1;
            b = idsT[b + h] <= id ? b + h : b;
            n -= h;
        }
        const bool hit = n == 1 && idsT[b] == id;
        const int32_t pvT = hit ? valT[b] : 0;
        if ((uint32_t)v >= (uint32_t)rows || (uint32_t)pvT >= (uint32_t)rows) {
            atomicOr(&a.flags[3], 2);
            continue;
        }
        atomicAdd(&histS[v], 1);
        if (hit) {
            atomicAdd(&histS[pvT], 1);
            atomicAdd(&histT[v], 1);
            ++hits;
        }
    }
    if (hits) atomicAdd(nhit, hits);
    __syncthreads();
    const int h = *nhit;
    float *outS = out_counts + jS * (int64_t)rows, *outT = out_counts + jT * (int64_t)rows;
    for (int x = tid; x < rows; x += kPairThreads) {
        const int absent_s = x == 0 ? ns - h : 0, absent_t = x == 0 ? nt - h : 0;      // row 0 = partner absent (counted: MLP(0) != 0)
        outS[x] = (float)(histS[x] + absent_s);
        outT[x] = (float)(histT[x] + absent_t);
    }
}


// ---------------------------------------------------------------------------------------------------------
// Pair form of the join (SURVEY.md 8(f).1 for the aggregations that are NOT linear in the rows -- the attention gate of
// model.py:59-62).  Every output row of a segment is the feature pair (table[pa], table[pb]) and the model's first
// stage maps it to pe_embedding(.).sum(-2) = e[pa] + e[pb]: a function of the index pair only.  A segment of ~400
// rows holds a few dozen distinct pairs, so the segment leaves as (pair, multiplicity) rows; gate softmax and the
// weighted sum over the segment are exact with the multiplicities as weights (spjoin.attn_stage).  The distinct pairs
// are found in an ORDERED open-addressing table in LDS (each slot keeps the smallest key that probed it, the larger
// one moves on: Amble-Knuth): its final layout does not depend on the order of the concurrent inserts, so the rows
// leave in a reproducible order (table slot order) without a sort.  Rows of segment j go to [seg[j], seg[j]+cnt[j]).
constexpr unsigned long long kPairEmpty = ~0ull;
__global__ __launch_bounds__(kPairThreads) void sjoin_pairs_kernel(const JoinArgs a, int64_t pb, int ts_log2,
                                                                   int32_t *__restrict__ out_pairs,
                                                                   int32_t *__restrict__ out_mult,
                                                                   int32_t *__restrict__ out_cnt) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int TS = 1 << ts_log2;
    unsigned long long *tabK = (unsigned long long *)lds_raw;   // [2][TS] distinct (pa << 32 | pb) of block A, block B
    int32_t *tabC = (int32_t *)(tabK + 2 * TS);                  // [2][TS] multiplicities
    int32_t *valA = tabC + 2 * TS;                               // [max_len]
    int32_t *valB = valA + a.max_len;
    int32_t *idsA = valB + a.max_len;
    int32_t *idsB = idsA + a.max_len;
    __shared__ int32_t wsum[2][kPairThreads / kWave];

    const int64_t p = xcd_item(blockIdx.x, gridDim.x);
    if (p >= a.S / 2) return;
    const int64_t j = (p / pb) * 2 * pb + (p % pb), j2 = j + pb;
    const int tid = threadIdx.x;
    const int64_t ra = a.own[j], rb = join_partner(a, j);
    if (a.own[j2] != rb || join_partner(a, j2) != ra) {
        if (tid == 0) atomicOr(&a.flags[3], 4);
        return;
    }
    int64_t ab, na64, bb, nb64;
    join_row(a, ra, ab, na64);
    join_row(a, rb, bb, nb64);
    if (na64 > a.max_len || nb64 > a.max_len) {
        if (tid == 0) atomicOr(&a.flags[3], 1);
        return;
    }
    const int na = (int)na64, nb = (int)nb64;
    const int32_t *data = (const int32_t *)a.data;
    for (int x = tid; x < 2 * TS; x += kPairThreads) {
        tabK[x] = kPairEmpty;
        tabC[x] = 0;
    }
    for (int r = tid; r < na; r += kPairThreads) {
        idsA[r] = a.indices[ab + r];
        valA[r] = data[ab + r];
    }
    for (int r = tid; r < nb; r += kPairThreads) {
        idsB[r] = a.indices[bb + r];
        valB[r] = data[bb + r];
    }
    __syncthreads();
    const uint32_t tmask = (uint32_t)TS - 1u;
    // the member's pair (searched once, kept in registers for the counting pass): <= 2 * max_len members, strided
    constexpr int kPer = 8;                                      // 2 * max_len <= 8 * 256 (checked on the host)
    unsigned long long mykey[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int t = tid + u * kPairThreads;
        mykey[u] = kPairEmpty;
        if (t >= na + nb) continue;
        const bool dirB = t >= na;
        const int r = dirB ? t - na : t;
        const int32_t *oid = dirB ? idsB : idsA, *oval = dirB ? valB : valA;
        const int32_t *pid = dirB ? idsA : idsB, *pval = dirB ? valA : valB;
        const int pn = dirB ? na : nb;
        const int32_t id = oid[r];
        int lo = 0, hi = pn;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pid[mid] < id) lo = mid + 1;
            else hi = mid;
        }
        const uint32_t pa = (uint32_t)oval[r], pbv = (lo < pn && pid[lo] == id) ? (uint32_t)pval[lo] : 0u;
        unsigned long long k = ((unsigned long long)pa << 32) | pbv;
        mykey[u] = k;
        unsigned long long *tk = tabK + (dirB ? TS : 0);
        uint32_t h = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> (64 - ts_log2));
        for (int probes = 0; probes < TS; ++probes) {            // ordered insert: the slot keeps the minimum
            const unsigned long long old = atomicMin(&tk[h], k);
            if (old == kPairEmpty || old == k) break;
            k = old > k ? old : k;                               // the larger key moves on
            h = (h + 1u) & tmask;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPer; ++u) {                             // multiplicities: read-only probe, one add
        const int t = tid + u * kPairThreads;
        if (t >= na + nb) continue;
        const bool dirB = t >= na;
        const unsigned long long k = mykey[u];
        const unsigned long long *tk = tabK + (dirB ? TS : 0);
        uint32_t h = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> (64 - ts_log2));
        while (tk[h] != k) h = (h + 1u) & tmask;                 // present by construction
        atomicAdd(&tabC[(dirB ? TS : 0) + h], 1);
    }
    __syncthreads();
    // rows leave in table-slot order: per-thread run of consecutive slots, block-wide exclusive scan of the occupancy
    const int per = TS / kPairThreads > 0 ? TS / kPairThreads : 1;
    for (int dir = 0; dir < 2; ++dir) {
        const unsigned long long *tk = tabK + dir * TS;
        const int32_t *tc = tabC + dir * TS;
        const int s0 = tid * per;
        int mine = 0;
        for (int x = s0; x < s0 + per && x < TS; ++x) mine += tk[x] != kPairEmpty;
        int inc = mine;
#pragma unroll
        for (int dd = 1; dd < kWave; dd <<= 1) {
            const int t2 = __shfl_up(inc, dd, kWave);
            if ((tid & (kWave - 1)) >= dd) inc += t2;
        }
        if ((tid & (kWave - 1)) == kWave - 1) wsum[dir][tid / kWave] = inc;
        __syncthreads();
        int base = 0, total = 0;
        for (int w2 = 0; w2 < kPairThreads / kWave; ++w2) {
            if (w2 < tid / kWave) base += wsum[dir][w2];
            total += wsum[dir][w2];
        }
        const int64_t jj = dir ? j2 : j;
        if (tid == 0) out_cnt[jj] = total;
        int64_t o = a.seg[jj] + base + inc - mine;
        for (int x = s0; x < s0 + per && x < TS; ++x)
            if (tk[x] != kPairEmpty) {
                out_pairs[2 * o] = (int32_t)(tk[x] >> 32);
                out_pairs[2 * o + 1] = (int32_t)(tk[x] & 0xFFFFFFFFu);
                out_mult[o] = tc[x];
                ++o;
            }
    }
}

}  // namespace subgacc

using namespace subgacc;

static size_t onepass_state_bytes(int64_t S);
extern "C" size_t subgacc_sjoin_workspace_bytes(int64_t S) {
    if (S < 0) S = 0;
    const size_t two_step = align_up((size_t)S * 8, 256) + scan_workspace_bytes(S);
    const size_t one_call = onepass_state_bytes(S);       // SUBGACC_JOIN_OPT_SIZES: the single-pass scan's state
    return two_step > one_call ? two_step : one_call;
}

static int join_sizes(const int64_t *spg_indptr, const int32_t *row_len, int64_t n_rows, const int64_t *own,
                      const int64_t *partner, int64_t S, int64_t *out_seg, int32_t *flags, void *workspace,
                      size_t workspace_bytes, void *stream) {
    SG_REQUIRE(S >= 0 && out_seg && n_rows >= 0, SUBGACC_ERR_BADARG, "sjoin_sizes: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (S == 0) return exclusive_scan_i64(nullptr, 0, out_seg, nullptr, 0, s);
    SG_REQUIRE((spg_indptr || row_len) && own, SUBGACC_ERR_BADARG, "sjoin_sizes: null argument");
    SG_REQUIRE(workspace && workspace_bytes >= subgacc_sjoin_workspace_bytes(S), SUBGACC_ERR_WORKSPACE,
               "sjoin_sizes: workspace too small");
    SegLen L{spg_indptr, row_len, n_rows, own, partner, flags, S};
    const int64_t nb = ceil_div(S, kSegTile);
    SG_REQUIRE(nb < (1ll << 31), SUBGACC_ERR_BADARG, "sjoin_sizes: too many segments");
    if (nb == 1) {
        hipLaunchKernelGGL(sjoin_seg_scan_kernel<kSegItems>, dim3(1), dim3(kScanThreads), 0, s, L, (const int64_t *)nullptr, out_seg);
    } else if (S <= (int64_t)kScanThreads * kSegItemsSmall) {
        hipLaunchKernelGGL(sjoin_seg_scan_kernel<kSegItemsSmall>, dim3(1), dim3(kScanThreads), 0, s, L, (const int64_t *)nullptr, out_seg);
    } else {
        int64_t *partial = (int64_t *)workspace;     // nb words <= S words
        hipLaunchKernelGGL(sjoin_seg_reduce_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, s, L, partial);
        hipLaunchKernelGGL(sjoin_seg_scan_kernel<kSegItems>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, L, (const int64_t *)partial,
                           out_seg);
    }
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

// the size pass of subgacc_sjoin_fill_v2(options & SUBGACC_JOIN_OPT_SIZES): one launch, see sjoin_sizes_onepass_kernel
static size_t onepass_state_bytes(int64_t S) {      // the header and one word per tile
    return align_up(sizeof(SizeState) + (size_t)ceil_div(S > 0 ? S : 1, (int64_t)kScanThreads * kOnePassItems) * 8, 256);
}


static int join_sizes_onepass(const subgacc_join_desc *d, hipStream_t s) {
    SG_REQUIRE(d->S >= 0 && d->n_rows >= 0 && d->out_seg && !d->seg, SUBGACC_ERR_BADARG,
               "sjoin_fill_v2: OPT_SIZES writes out_seg [S+1] and reads no seg");
    SG_REQUIRE(d->own || d->S == 0, SUBGACC_ERR_BADARG, "sjoin_fill_v2: null segment list");
    const int64_t nb = d->S > 0 ? ceil_div(d->S, (int64_t)kScanThreads * kOnePassItems) : 1;
    SG_REQUIRE(nb < (1ll << 31), SUBGACC_ERR_BADARG, "sjoin_fill_v2: too many segments");
    SG_REQUIRE(d->size_state && (size_t)d->size_state_bytes >= onepass_state_bytes(d->S), SUBGACC_ERR_WORKSPACE,
               "sjoin_fill_v2: size_state too small (subgacc_sjoin_workspace_bytes(S) bytes, zeroed once)");
    SegLen L{d->row_off, d->row_len, d->n_rows, d->own, d->partner, d->flags, d->S};
    if (!d->row_off && !d->row_len) L.row_head = d->ids, L.row_stride = d->row_stride;      // headed rows
    hipLaunchKernelGGL(sjoin_sizes_onepass_kernel<kOnePassItems>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, L, d->out_seg,
                       (unsigned long long *)d->size_state, d->host_tail, (int)nb);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_sjoin_sizes(const int64_t *spg_indptr, int64_t n_rows, const int64_t *own, const int64_t *partner,
                                   int64_t S, int64_t *out_seg, int32_t *flags, void *workspace, size_t workspace_bytes,
                                   void *stream) {
    SG_REQUIRE(spg_indptr || S == 0, SUBGACC_ERR_BADARG, "sjoin_sizes: null argument");
    return join_sizes(spg_indptr, nullptr, n_rows, own, partner, S, out_seg, flags, workspace, workspace_bytes, stream);
}

extern "C" int subgacc_sjoin_sizes_rows(const int32_t *row_len, int64_t n_rows, const int64_t *own, const int64_t *partner,
                                        int64_t S, int64_t *out_seg, int32_t *flags, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    SG_REQUIRE(row_len || S == 0, SUBGACC_ERR_BADARG, "sjoin_sizes_rows: null argument");
    return join_sizes(nullptr, row_len, n_rows, own, partner, S, out_seg, flags, workspace, workspace_bytes, stream);
}

// LDS of the per-wave staging areas of emit_key_span: [4 + 64 x 2k] floats per wave + the round-up to a 16-byte boundary
static inline size_t key_stage_bytes(int waves, int k) { return 16 + (size_t)waves * (kWave * 2 * k + 4) * 4; }

// Table payload (SFptr+1, or slots of the table of distinct rows) over a mirrored list: the pair kernel of the key rows with the
// feature rows read from the Z_SF table instead of unpacked (round 5: the resident-store join of the reference's own flow,
// gather(edge, z, encode=Z_SF), takes the same plan -- one search per pair, only the longer row in LDS).  vec4: k == 4, 16-byte rows.
static int launch_table_pairs(JoinArgs &a, int64_t S, int64_t pair_block, bool vec4, void *stream, const char *who) {
    const int k = a.out_xz ? a.k : 1;
    const size_t lds = (size_t)a.max_len * 12 + key_stage_bytes(kPairEmit / kWave, k);
    SG_REQUIRE(lds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS, "%s: rows of %d members do not fit LDS", who, (int)a.max_len);
    a.split = pair_split(S / 2);
    const int64_t grid = xcd_grid(S / 2 * a.split);
    SG_REQUIRE(grid < (1ll << 31) && S / 2 < (1ll << 31), SUBGACC_ERR_BADARG, "%s: too many segments in one call", who);
    const uint32_t pairs = (uint32_t)(S / 2), pb = (uint32_t)pair_block;
    hipStream_t s = (hipStream_t)stream;
#define SG_TAB_LAUNCH(KVV)                                                                                                     \
    do {                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                    \
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)sjoin_keypair_kernel<KVV, kPairEmit, false, true>,                   \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
        hipLaunchKernelGGL((sjoin_keypair_kernel<KVV, kPairEmit, false, true>), dim3((unsigned)grid), dim3(kPairEmit), lds, s, a, pb, pairs); \
    } while (0)
    if (vec4) SG_TAB_LAUNCH(4);
    else if (a.out_xz && a.k == 3) SG_TAB_LAUNCH(3);      // the 2-hop configurations (collab-like)
    else if (a.out_xz && a.k == 5) SG_TAB_LAUNCH(5);      // 4 hops
    else SG_TAB_LAUNCH(0);
#undef SG_TAB_LAUNCH
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

// ---------------------------------------------------------------------------------------------------------
// ONE entry point for every form of the join (ABI 6; since ABI 7 the only one -- the seven per-form entry points of ABI 1-5 are gone).
// A descriptor says what the store looks like (packed, strided or headed rows; which payload), which segments to join, what the
// feature rows are made from and which outputs are wanted.  The kernels' arguments are built from it ONCE, here.
// key payload (strided rows of a transient batch, or a packed store whose payload was re-keyed): shared launcher
static int launch_key_join(JoinArgs &a, int32_t num_walks, int32_t num_steps, int64_t S, int64_t pair_block, void *stream,
                           const char *who, bool wide = false) {
    const int shift = subgacc_key_shift(num_walks, num_steps);
    if (shift < 0) return shift;
    SG_REQUIRE(num_steps * shift + 1 <= (wide ? 63 : 31) && num_steps + 1 <= 16, SUBGACC_ERR_KEYWIDTH,
               "%s: LP keys of %d steps x %d bits do not fit %d bits", who, num_steps, shift, wide ? 64 : 32);
    a.table = nullptr, a.table_rows = 0, a.k = num_steps + 1;
    a.out_idx = nullptr;
    a.slot_id = nullptr, a.val_add = 0;
    a.key_M = num_walks, a.key_m = num_steps, a.key_shift = shift;
    // LDS: the longer row of a pair (id + key + partner key per member), one staging area per wave
    const int nt = (a.max_len > 512 && pair_split(S / 2) == 1 && (a.k == 4 || (wide && a.k == 5))) ? 256 : kPairEmit;
    // (rows of 3- and 4-hop sets -- up to 601 / 801 members -- take 256 lanes per pair: the two rows arrive in half the trips and eight
    //  wavefronts emit the ~12-25 spans: cit2 join 0.437 -> 0.429 ms; 2-hop rows, ~120 members, lose 9 % with 256 lanes and keep 128:
    //  profiles/r12_ab_pair_threads.log)
    const size_t lds = (size_t)a.max_len * (wide ? 20 : 12) + key_stage_bytes(nt / kWave, a.k);
    SG_REQUIRE(lds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS, "%s: rows of %d members do not fit LDS", who, (int)a.max_len);
    a.split = pair_split(S / 2);
    const int64_t grid = xcd_grid(S / 2 * a.split);
    SG_REQUIRE(grid < (1ll << 31) && S / 2 < (1ll << 31), SUBGACC_ERR_BADARG, "%s: too many segments in one call", who);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t pairs = (uint32_t)(S / 2), pb = (uint32_t)pair_block;
#define SG_KEY_LAUNCH(KVV, NTT, W)                                                                                             \
    do {                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                    \
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)sjoin_keypair_kernel<KVV, NTT, W>,                                   \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
        hipLaunchKernelGGL((sjoin_keypair_kernel<KVV, NTT, W>), dim3((unsigned)grid), dim3(NTT), lds, s, a, pb, pairs);         \
    } while (0)
    if (wide) {
        if (a.k == 5 && nt == 256) SG_KEY_LAUNCH(5, 256, true);
        else if (a.k == 5) SG_KEY_LAUNCH(5, kPairEmit, true);
        else SG_KEY_LAUNCH(0, kPairEmit, true);
    } else if (a.k == 4 && nt == 256) SG_KEY_LAUNCH(4, 256, false);
    else if (a.k == 4) SG_KEY_LAUNCH(4, kPairEmit, false);      // 3 hops
    else if (a.k == 3) SG_KEY_LAUNCH(3, kPairEmit, false);      // 2 hops (the collab-like configurations)
    else if (a.k == 5) SG_KEY_LAUNCH(5, kPairEmit, false);      // 4 hops with 32-bit keys (M <= 127: the reference's own citation2 setting)
    else SG_KEY_LAUNCH(0, kPairEmit, false);
#undef SG_KEY_LAUNCH
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}


// mirrored lists of a float payload (train.py:39-43)
static int launch_f64_pairs(JoinArgs &a, int64_t S, int64_t pair_block, void *stream) {
    // float rows: T + its partner slots in LDS (20 bytes per member).  Short rows (the top-100 PPR store): ONE wave per pair
    // -- twice the pairs in flight per CU (cit2-PPR join 0.154 -> 0.139 ms in round 2; integer rows were slower that way)
    // (round 3 tried persistent waves with a four-stage software pipeline over their pairs: 91 us against 77; round 5 the
    //  same for key rows, profiles/r18_join_pipe.log: the hardware's interleaving of resident workgroups wins both times)
    const size_t flds = (size_t)a.max_len * 20 + 16;
    SG_REQUIRE(flds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS, "sjoin_fill: float rows of %d members do not fit LDS", (int)a.max_len);
    a.split = pair_split(S / 2);
    const int64_t grid = xcd_grid(S / 2 * a.split);
    SG_REQUIRE(grid < (1ll << 31) && S / 2 < (1ll << 31), SUBGACC_ERR_BADARG, "sjoin_fill: too many segments in one call");
    const uint32_t pairs = (uint32_t)(S / 2), pb32 = (uint32_t)pair_block;
    hipStream_t s = (hipStream_t)stream;
    if (a.max_len <= 2 * kWave) {
        hipLaunchKernelGGL((sjoin_f64pair_kernel<kWave>), dim3((unsigned)grid), dim3(kWave), flds, s, a, pb32, pairs);
    } else {
        if (flds > 64 * 1024)
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)sjoin_f64pair_kernel<kPairEmit>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
        hipLaunchKernelGGL((sjoin_f64pair_kernel<kPairEmit>), dim3((unsigned)grid), dim3(kPairEmit), flds, s, a, pb32, pairs);
    }
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

// any list that is not made of mirrored pairs (or whose two rows do not fit LDS together): one wave per segment
static int launch_segments(JoinArgs &a, bool f64, bool vec4, void *stream) {
    size_t lds = (size_t)a.max_len * (f64 ? 12 : 8);
    const bool staged = lds <= (size_t)kLdsBytes;   // else: rows longer than LDS, searched in place (sjoin_fill_kernel<.., false>)
    if (!staged) lds = 0;
    const int64_t grid = xcd_grid(a.S);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "sjoin_fill: too many segments in one call");
    hipStream_t s = (hipStream_t)stream;
#define SG_JOIN_LAUNCH(F, KVV)                                                                                   \
    do {                                                                                                          \
        if (lds > 64 * 1024)                                                                                      \
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)sjoin_fill_kernel<F, KVV>,                             \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
        hipLaunchKernelGGL((sjoin_fill_kernel<F, KVV>), dim3((unsigned)grid), dim3(kJoinThreads), lds, s, a);     \
    } while (0)
    if (!staged) {
        if (f64) hipLaunchKernelGGL((sjoin_fill_kernel<true, 0, false>), dim3((unsigned)grid), dim3(kJoinThreads), 0, s, a);
        else if (vec4) hipLaunchKernelGGL((sjoin_fill_kernel<false, 4, false>), dim3((unsigned)grid), dim3(kJoinThreads), 0, s, a);
        else hipLaunchKernelGGL((sjoin_fill_kernel<false, 0, false>), dim3((unsigned)grid), dim3(kJoinThreads), 0, s, a);
    } else if (f64) SG_JOIN_LAUNCH(true, 0);
    else if (vec4) SG_JOIN_LAUNCH(false, 4);
    else SG_JOIN_LAUNCH(false, 0);
#undef SG_JOIN_LAUNCH
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

static int launch_counts(JoinArgs &a, int64_t pair_block, float *out_counts, void *stream) {
    const size_t lds = (size_t)a.max_len * 8 + (size_t)a.table_rows * 8 + 16;
    SG_REQUIRE(lds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS,
               "sjoin_counts: %lld distinct LP rows and rows of %d members need %zu B of LDS; use the row form",
               (long long)a.table_rows, (int)a.max_len, lds);
    if (lds > 64 * 1024)
        SG_CHECK_HIP(hipFuncSetAttribute((const void *)sjoin_counts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t grid = xcd_grid(a.S / 2);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "sjoin_counts: too many segments in one call");
    hipLaunchKernelGGL(sjoin_counts_kernel, dim3((unsigned)grid), dim3(kPairThreads), lds, (hipStream_t)stream, a, pair_block, out_counts);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

static int launch_pair_form(JoinArgs &a, int64_t pair_block, int32_t *out_pairs, int32_t *out_mult, int32_t *out_cnt, void *stream) {
    SG_REQUIRE(2 * (int64_t)a.max_len <= 8 * kPairThreads, SUBGACC_ERR_LDS,
               "sjoin_pairs: rows of %d members are too long for the pair form (<= %d); use the row form", (int)a.max_len, 4 * kPairThreads);
    int ts_log2 = 6;                                           // distinct pairs of one block <= its row length
    while ((1 << ts_log2) < a.max_len + a.max_len / 4 + 1) ++ts_log2;
    const size_t lds = (size_t)2 * (1u << ts_log2) * 12 + (size_t)a.max_len * 16;
    SG_REQUIRE(lds + 64 <= (size_t)kLdsBytes, SUBGACC_ERR_LDS, "sjoin_pairs: %zu B of LDS needed", lds);
    if (lds > 64 * 1024)
        SG_CHECK_HIP(hipFuncSetAttribute((const void *)sjoin_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t grid = xcd_grid(a.S / 2);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "sjoin_pairs: too many segments in one call");
    hipLaunchKernelGGL(sjoin_pairs_kernel, dim3((unsigned)grid), dim3(kPairThreads), lds, (hipStream_t)stream, a, pair_block, ts_log2,
                       out_pairs, out_mult, out_cnt);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_sjoin_fill_v2(const subgacc_join_desc *d, void *stream) {
    SG_REQUIRE(d, SUBGACC_ERR_BADARG, "sjoin_fill_v2: null descriptor");
    SG_REQUIRE(d->struct_bytes == (int32_t)sizeof(subgacc_join_desc), SUBGACC_ERR_BADARG,
               "sjoin_fill_v2: descriptor of %d bytes, this library's is %d (set struct_bytes = sizeof(subgacc_join_desc))",
               (int)d->struct_bytes, (int)sizeof(subgacc_join_desc));
    // the store's layout: packed rows (row_off), strided rows (row_len + row_stride), headed rows (neither; row_stride)
    const bool packed = d->row_off != nullptr, strided = d->row_len != nullptr;
    const bool headed = !packed && !strided && d->row_stride > 0;
    SG_REQUIRE((int)packed + (int)strided + (int)headed == 1, SUBGACC_ERR_BADARG,
               "sjoin_fill_v2: exactly one of row_off (packed rows) / row_len (strided rows) / neither, with row_stride (headed rows)");
    SG_REQUIRE(packed || (d->row_stride > (headed ? 1 : 0) && d->row_stride < (1ll << 31)), SUBGACC_ERR_BADARG,
               "sjoin_fill_v2: row_stride = %lld", (long long)d->row_stride);
    SG_REQUIRE((d->options & ~SUBGACC_JOIN_OPT_SIZES) == 0, SUBGACC_ERR_BADARG, "sjoin_fill_v2: unknown option bits %d", (int)d->options);
    SG_REQUIRE(d->form >= SUBGACC_JOIN_ROWS && d->form <= SUBGACC_JOIN_PAIRS, SUBGACC_ERR_BADARG, "sjoin_fill_v2: unknown form %d", (int)d->form);
    SG_REQUIRE(d->payload_kind >= SUBGACC_JOIN_SFPTR && d->payload_kind <= SUBGACC_JOIN_KEY64, SUBGACC_ERR_BADARG,
               "sjoin_fill_v2: unknown payload kind %d", (int)d->payload_kind);
    const int kind = d->payload_kind;
    const bool f64 = kind == SUBGACC_JOIN_F64, keyed = kind == SUBGACC_JOIN_KEY32 || kind == SUBGACC_JOIN_KEY64;
    const bool sized = (d->options & SUBGACC_JOIN_OPT_SIZES) != 0;
    const bool sizes_only = sized && !d->out_xz && !d->out_idx;      // the "count" half of a two-call pattern: out_seg and host_tail only
    const int64_t *seg = sized ? d->out_seg : d->seg;
    const int64_t S = d->S, pb = d->pair_block;
    // ---- what the kernels would not survive is refused here, before anything is launched (with OPT_SIZES: before the size pass has
    //      written out_seg / host_tail)
    SG_REQUIRE(S >= 0 && d->n_rows >= 0 && d->max_len >= 0, SUBGACC_ERR_BADARG, "sjoin_fill_v2: bad arguments");
    SG_REQUIRE(!sized || d->form == SUBGACC_JOIN_ROWS, SUBGACC_ERR_BADARG, "sjoin_fill_v2: OPT_SIZES goes with the row form");
    if (!sizes_only) {
        SG_REQUIRE(d->flags, SUBGACC_ERR_BADARG, "sjoin_fill_v2: null argument (flags)");
        SG_REQUIRE(!(kind == SUBGACC_JOIN_KEY64 && packed) && !(keyed && strided && d->out_segid), SUBGACC_ERR_BADARG,
                   "sjoin_fill_v2: this payload kind does not go with this row layout (64-bit keys: strided or headed rows; keys "
                   "of strided rows -- a transient batch -- are joined with segment pointers)");
        SG_REQUIRE(d->form == SUBGACC_JOIN_ROWS || kind == SUBGACC_JOIN_SFPTR, SUBGACC_ERR_BADARG,
                   "sjoin_fill_v2: the count and pair forms join an SFptr store");
        if (S > 0 || sized) {
            SG_REQUIRE(d->ids && d->payload, SUBGACC_ERR_BADARG, "sjoin_fill_v2: null argument (ids / payload)");
            if (d->form == SUBGACC_JOIN_ROWS) SG_REQUIRE(d->out_xz || d->out_idx, SUBGACC_ERR_BADARG, "sjoin_fill_v2: no output requested");
        }
    }
    if (sized) {       // the whole join of a batch in one call: size pass, then the fill behind it
        const int rc = join_sizes_onepass(d, (hipStream_t)stream);
        if (rc != SUBGACC_OK || sizes_only) return rc;
    }
    if (keyed) {       // (also for S == 0: a caller learns about a key width that does not fit from its first, empty, call)
        const int shift = subgacc_key_shift(d->num_walks, d->num_steps);
        if (shift < 0) return shift;
    }
    if (S == 0) return SUBGACC_OK;
    SG_REQUIRE(d->own && (d->partner || pb > 0), SUBGACC_ERR_BADARG, "sjoin_fill_v2: null argument (segments)");
    SG_REQUIRE(pb >= 0 && (pb == 0 || S % (2 * pb) == 0), SUBGACC_ERR_BADARG,
               "sjoin_fill_v2: S = %lld is not a multiple of 2*pair_block", (long long)S);
    const bool mirrored = pb > 0;
    SG_REQUIRE(d->form == SUBGACC_JOIN_COUNTS || seg, SUBGACC_ERR_BADARG, "sjoin_fill_v2: null argument (seg)");

    JoinArgs a;
    a.sized_here = sized;
    a.indptr = d->row_off, a.indices = headed ? d->ids + 1 : d->ids, a.data = d->payload;
    a.row_len = d->row_len, a.row_stride = packed ? 0 : d->row_stride, a.row_head = headed ? d->ids : nullptr;
    a.pb = pb, a.own = d->own, a.partner = d->partner, a.seg = seg, a.S = S, a.n_rows = d->n_rows;
    a.table = d->table, a.table_rows = d->table_rows, a.k = d->k;
    a.out_xz = d->out_xz, a.out_idx = d->out_idx, a.out_segid = d->out_segid;
    // the longest row: the caller's bound for packed rows, what a row's slot holds otherwise
    a.max_len = packed ? (d->max_len > 0 ? d->max_len : 1) : (int32_t)(headed ? d->row_stride - 1 : d->row_stride);
    a.flags = d->flags;
    a.slot_id = nullptr, a.val_add = 0;
    a.key_M = a.key_m = a.key_shift = 0;

    if (d->form == SUBGACC_JOIN_COUNTS) {
        SG_REQUIRE(mirrored && d->out_counts && d->table_rows > 0, SUBGACC_ERR_BADARG,
                   "sjoin_counts: mirrored blocks (pair_block > 0), out_counts and table_rows > 0");
        a.seg = nullptr, a.table = nullptr, a.k = 0, a.out_xz = nullptr, a.out_idx = nullptr, a.out_segid = nullptr;
        return launch_counts(a, pb, d->out_counts, stream);
    }
    if (d->form == SUBGACC_JOIN_PAIRS) {
        SG_REQUIRE(mirrored && d->out_pairs && d->out_mult && d->out_cnt, SUBGACC_ERR_BADARG,
                   "sjoin_pairs: mirrored blocks (pair_block > 0), out_pairs, out_mult and out_cnt");
        a.table = nullptr, a.table_rows = 0, a.k = 0, a.out_xz = nullptr, a.out_idx = nullptr, a.out_segid = nullptr;
        return launch_pair_form(a, pb, d->out_pairs, d->out_mult, d->out_cnt, stream);
    }
    if (keyed) {
        SG_REQUIRE(mirrored && d->out_xz, SUBGACC_ERR_BADARG,
                   "sjoin_fill_v2: key rows are joined as mirrored blocks (pair_block > 0, S a multiple of 2*pair_block) into out_xz");
        return launch_key_join(a, d->num_walks, d->num_steps, S, pb, stream, "sjoin_fill_v2", kind == SUBGACC_JOIN_KEY64);
    }
    if (f64) {
        SG_REQUIRE(d->out_xz && !d->out_idx && !d->table, SUBGACC_ERR_BADARG,
                   "sjoin_fill_v2: float payload writes out_xz [R,2,1] only (train.py:39-43)");
        a.k = 1, a.table = nullptr, a.table_rows = 0;
        a.spec_len = packed ? 0 : (d->max_len > 0 ? d->max_len : (int32_t)(a.max_len < 128 ? a.max_len : 128));     // (headed / strided rows: max_len is the hint)
        if (mirrored && (size_t)a.max_len * 20 + 16 <= (size_t)kLdsBytes) return launch_f64_pairs(a, S, pb, stream);
        SG_REQUIRE(packed, SUBGACC_ERR_BADARG, "sjoin_fill_v2: strided / headed float rows are joined as mirrored blocks (pair_block > 0)");
        return launch_segments(a, true, false, stream);
    }
    // SFptr+1 (packed / headed rows of a numbered store) or table slots (strided rows of a transient batch) with the Z_SF table
    SG_REQUIRE(!d->out_xz || (d->table && d->table_rows > 0 && d->k > 0 && d->k <= 16), SUBGACC_ERR_BADARG,
               "sjoin_fill_v2: out_xz needs the feature table, k <= 16");
    if (strided) {
        // numbered table given: slot -> SFptr+1 through its id plane (uniq_table.hpp); else the table is indexed by slot+1
        SG_REQUIRE(!d->uniq_table || d->uniq_capacity > 0, SUBGACC_ERR_BADARG, "sjoin_fill_v2: uniq_capacity");
        a.slot_id = d->uniq_table ? (const int32_t *)((const char *)d->uniq_table + (size_t)d->uniq_capacity * 16) : nullptr;
        a.val_add = d->uniq_table ? 0 : 1;
    }
    const bool vec4 = d->out_xz && d->k == 4 && ((uintptr_t)d->table % 16 == 0) && ((uintptr_t)d->out_xz % 16 == 0);
    if (mirrored && (size_t)a.max_len * 16 <= (size_t)kLdsBytes) return launch_table_pairs(a, S, pb, vec4, stream, "sjoin_fill_v2");
    SG_REQUIRE(packed, SUBGACC_ERR_BADARG, "sjoin_fill_v2: strided / headed rows are joined as mirrored blocks (pair_block > 0)");
    return launch_segments(a, false, vec4, stream);
}
