// keyrows.hip -- batched registration of key rows: the table of distinct LP rows for a store that is kept (gfx950).
//
// The reference numbers the distinct LP rows in order of first appearance over the concatenated sets (subg_acc.c:957-978:
// roots in query order, members in first-visit order inside a root).  The table form of the fused-row walk kernel
// (walk_rows_kernel<KR = false>) pays for that inside every root's epilogue: first-visit numbers for every member (an LDS
// atomic per visit, 4 bytes of LDS per slot), a fold table, global atomics -- 1.09 ms per 131,072 roots where the key-rows
// form (rows of (id, 32-bit LP key), no table) takes 0.76 ms.  But only a few thousand of the millions of roots are ever
// the FIRST to show an LP row, and only for those does the order inside the root matter.  So:
//   1. every root is walked with the key-rows kernel;
//   2. ONE streaming pass over the rows' keys (this file) registers every distinct key in the HBM table with the COARSE tag
//      (root index)*stride + stride-1 of the smallest row that shows it -- as late as anything inside that root, before
//      anything of the next -- and lists the rows whose insert lowered a tag: the candidates for a first appearance
//      (a key's true first root always lowers its tag; a few candidates per distinct row and block);
//   3. the candidates alone are walked again by the table form in `tags_only` mode (subgacc_walk_tags): their keys get
//      the exact tag (root index)*stride + first-visit number -- at or below the coarse tag of the same root, so after the
//      atomicMin every distinct key holds exactly the tag the one-pass table form gives it, and
//      subgacc_uniq_number numbers the table as before.
// The same kernel copies the rows to their packed place (strided -> CSR copy of the store): after the numbering with the
// payload SFptr+1 looked up on the way (one chunk of roots), or -- a job of several chunks, numbered at the end -- while it
// registers, with the key kept as payload and translated by one flat pass at the end (subgacc_keyrows_translate).
//
// No streaming loop ever talks to the HBM table.  A store of 10^9 members holds 10^2..10^4 distinct LP rows; asked per row
// (a few dozen distinct keys each) the table would see ~50 dependent probes per row, more requests than the rows' own lines,
// and even a per-wave cache in front of it stalls on a round trip whenever a new key turns up (measured: 4.3 ms per 10^9
// members; a block-wide cache filled per tile of rows behind a vote barrier: 1.8 ms read-only, 5.3 ms copying -- its waves ran
// in lockstep from barrier to barrier).  Instead a block owns a contiguous range of rows and ONE dictionary in LDS:
//   registering: key -> smallest row of the range that shows it (one 8-byte word {row | key}: ds_cmpst_b64 to claim,
//                ds_min_u64 to lower), filled by free-running waves; the dictionary is flushed to the HBM table once, when
//                the range is done (one insert per distinct key of the range);
//   looking up:  the numbered keys (ukeys) are loaded into the dictionary before the first row.
// A row's loads are unconditional (clamped indices), so that the waits between them are counted ones.
#include <stdlib.h>

#include "common.hpp"
#include "uniq_table.hpp"

namespace subgacc {

#ifndef KR_THREADS      // dev-only compile-time knobs (tools/keyrows_bench.py)
#define KR_THREADS 512
#endif
#ifndef KR_UNROLL
#define KR_UNROLL 8      // (8 members per lane in flight: find+copy 1.50 -> 1.28 ms per 10^6 cit2 rows; 2: 1.47, 4: 1.50 -- tools/keyrows_bench.py)
#endif
#ifndef KR_NT        // 1: non-temporal row loads / packed stores -- measured slower here (find+copy 2.21 vs 1.96 ms per 10^6 cit2 rows)
#define KR_NT 0
#endif
#if KR_NT
#define KR_LOAD(p) __builtin_nontemporal_load(p)
#define KR_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define KR_LOAD(p) (*(p))
#define KR_STORE(v, p) (*(p) = (v))
#endif
constexpr int kKrThreads = KR_THREADS, kKrWaves = kKrThreads / kWave;
constexpr int kKrDictBits = 12, kKrDict = 1 << kKrDictBits;     // 4,096 entries x 8 B = 32 KB: 4 blocks = 32 waves per CU
constexpr int kKrDictMax = kKrDict * 3 / 4;                    // entries preloaded at most (the rest is asked for in HBM)
constexpr int kKrProbes = 32;    // (a key that finds no place within this many probes is asked for in HBM every time it turns up)
constexpr int kKrUnroll = KR_UNROLL;                 // members per lane whose loads are in flight together
constexpr int kKrMaxRows = 4096;             // rows per block (candidate bitmap)
enum { KR_REGISTER = 0, KR_REGISTER_COPY = 1, KR_FIND_WRITE = 2 };

// claim-or-find like uniq_global_insert; `lowered` reports whether this call may have lowered the key's tag (then the
// caller's row is a candidate for the key's first appearance; a stale larger `seen` only adds a harmless candidate)
__device__ __forceinline__ int32_t uniq_global_insert_ex(const UniqTable &t, unsigned long long key, unsigned long long tag,
                                                         int32_t *flags, bool &lowered) {
    uint64_t h = mix64(key) & t.mask;
    for (uint64_t probes = 0; probes <= t.mask && probes < kMaxProbes; ++probes) {
        unsigned long long cur = t.keys[h];
        const unsigned long long seen = t.mintag[h];
        if (cur == kEmptyKey) cur = atomicCAS(&t.keys[h], kEmptyKey, key);
        if (cur == kEmptyKey || cur == key) {
            if (seen > tag) {
                atomicMin(&t.mintag[h], tag);
                lowered = true;
            }
            return (int32_t)h;
        }
        h = (h + 1) & t.mask;
    }
    atomicOr(&flags[2], 1);  // table (nearly) full
    return -1;
}

__device__ __forceinline__ int32_t uniq_global_find(const UniqTable &t, unsigned long long key) {
    uint64_t h = mix64(key) & t.mask;
    for (uint64_t probes = 0; probes <= t.mask && probes < kMaxProbes; ++probes) {
        const unsigned long long cur = t.keys[h];
        if (cur == key) return (int32_t)h;
        if (cur == kEmptyKey) return -1;
        h = (h + 1) & t.mask;
    }
    return -1;
}

__device__ __forceinline__ uint32_t kr_home(uint32_t key) {       // LP keys are packed small counts: mix before taking the top bits
    uint32_t h = key * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return h >> (32 - kKrDictBits);
}

// The dictionary: two planes of 32-bit words in LDS -- keys (0xFFFFFFFF = empty) and payloads (SFptr+1, or the smallest row that
// shows the key).  A set's members mostly carry the same dozen LP keys, so most lanes of a wave ask for the same few entries:
// that is what LDS does best (tools/lds_probe.hip: 64 lanes on one word, or on ten hot words, read at the conflict-free rate,
// ds_read_b32 and ds_read_b64 alike; only 64 random words cost 2x).  A translate pass over 3.4e8 members runs at 0.61 ms
// against 0.55 ms for a plain copy of the same bytes (tools/keyrows_bench.py).
struct KrDict {
    uint32_t *k, *v;
};
constexpr uint32_t kKrEmpty = 0xFFFFFFFFu;     // never a key: key rows need m*SHIFT+1 <= 31 bits

__device__ __forceinline__ void kr_clear(const KrDict d, int tid) {
    for (int s = tid; s < kKrDict; s += kKrThreads) d.k[s] = kKrEmpty, d.v[s] = 0xFFFFFFFFu;
}

#if defined(KR_COUNT_MISS)      // dev-only: how often does a look-up leave the dictionary empty-handed? (tools/keyrows_bench.py)
__device__ unsigned long long g_kr_miss;
extern "C" long long subgacc_debug_kr_misses(void) {
    unsigned long long v = 0, z = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_kr_miss), 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kr_miss), &z, 8);
    return (long long)v;
}
#endif
#if defined(KR_ABLATE_LOOKUP) && KR_ABLATE_LOOKUP == 4      // 4: the whole probe loop, but a miss is 0 instead of a question to the HBM table
#define KR_MISS 0
#else
#define KR_MISS (-1)
#endif
// look-up: the payload, or -1 when the key is not there
__device__ __forceinline__ int32_t kr_lookup(const KrDict d, uint32_t key) {
#if defined(KR_ABLATE_LOOKUP) && KR_ABLATE_LOOKUP == 1     // dev-only timing experiments (tools/keyrows_bench.py; results are wrong)
    return (int32_t)(key ^ 1u);                              // 1: no dictionary at all
#elif defined(KR_ABLATE_LOOKUP) && KR_ABLATE_LOOKUP == 2
    return (int32_t)d.v[kr_home(key)];                       // 2: one LDS read, no probe loop, no compare
#elif defined(KR_ABLATE_LOOKUP) && KR_ABLATE_LOOKUP == 3
    {                                                        // 3: the first probe only (key + payload), no loop
        const uint32_t h0 = kr_home(key);
        return d.k[h0] == key ? (int32_t)d.v[h0] : 0;
    }
#endif
    uint32_t h = kr_home(key);
#pragma unroll 1
    for (int p = 0; p < kKrProbes; ++p) {
        const uint32_t e = d.k[h];
        if (e == key) return (int32_t)d.v[h];
        if (e == kKrEmpty) break;
        h = (h + 1) & (kKrDict - 1);
    }
#if defined(KR_COUNT_MISS)
    atomicAdd(&g_kr_miss, 1ull);
#endif
    return KR_MISS;
}

// registering: the entry of `key` ends up holding the smallest row that called; false when the neighbourhood is crowded
__device__ __forceinline__ bool kr_note(const KrDict d, uint32_t key, uint32_t row) {
    uint32_t h = kr_home(key);
#pragma unroll 1
    for (int p = 0; p < kKrProbes; ++p) {
        uint32_t e = d.k[h];
        if (e == kKrEmpty) {
            e = atomicCAS(&d.k[h], kKrEmpty, key);
            if (e == kKrEmpty) e = key;
        }
        if (e == key) {
            if (d.v[h] > row) atomicMin(&d.v[h], row);
            return true;
        }
        h = (h + 1) & (kKrDict - 1);
    }
    return false;
}

// the numbered keys, number + 1 as payload (a barrier must follow)
__device__ __forceinline__ void kr_preload(const KrDict d, const unsigned long long *__restrict__ ukeys, const int64_t *__restrict__ n_ukeys,
                                           int64_t max_ukeys, int tid) {
    int64_t c = *n_ukeys;
    if (c > max_ukeys) c = max_ukeys;
    if (c > kKrDictMax) c = kKrDictMax;
    for (int64_t x = tid; x < c; x += kKrThreads) {
        const uint32_t key = (uint32_t)ukeys[x];
        uint32_t h = kr_home(key);
        for (int p = 0; p < kKrProbes; ++p) {
            if (atomicCAS(&d.k[h], kKrEmpty, key) == kKrEmpty) {
                d.v[h] = (uint32_t)(x + 1);
                break;
            }
            h = (h + 1) & (kKrDict - 1);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(kKrThreads) void keyrows_pass_kernel(const int32_t *__restrict__ row_ids, const int32_t *__restrict__ row_keys,
                                                                  const int32_t *__restrict__ nsize, const int64_t *__restrict__ row_off,
                                                                  int64_t n, int32_t stride, int64_t root_base, int32_t rows_per_block,
                                                                  UniqTable t, const unsigned long long *__restrict__ ukeys,
                                                                  const int64_t *__restrict__ n_ukeys, int64_t max_ukeys,
                                                                  int32_t *__restrict__ out_indices, int32_t *__restrict__ out_data,
                                                                  int32_t *__restrict__ cand, unsigned long long *__restrict__ n_cand,
                                                                  int32_t *flags) {
    __shared__ uint32_t dmem[2 * kKrDict];
    const KrDict dict{dmem, dmem + kKrDict};
    __shared__ uint32_t candbits[kKrMaxRows / 32];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    kr_clear(dict, tid);
    if (MODE != KR_FIND_WRITE)
        for (int s = tid; s < kKrMaxRows / 32; s += kKrThreads) candbits[s] = 0u;
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * rows_per_block;
    const int64_t last = min(first + (int64_t)rows_per_block, n);
    if (MODE == KR_FIND_WRITE) {      // the numbered keys, number + 1 as payload
        kr_preload(dict, ukeys, n_ukeys, max_ukeys, tid);
        __syncthreads();
    }
    // ---- the stream: wave w takes rows first + w, first + w + kKrWaves, ...; nothing below waits for another wave
    int64_t i = first + wave;
    int ns_next = 0;
    int64_t dst_next = 0;
    if (i < last) {
        ns_next = min(nsize[i], stride);
        if (MODE != KR_REGISTER) dst_next = row_off[i];
    }
    for (; i < last; i += kKrWaves) {
        const int ns = ns_next;
        const int64_t dst = dst_next;
        {   // the next row's length and place are on their way while this row is worked on
            const int64_t j = i + kKrWaves < last ? i + kKrWaves : i;
            ns_next = min(nsize[j], stride);
            if (MODE != KR_REGISTER) dst_next = row_off[j];
        }
        const int64_t src = i * (int64_t)stride;
        const uint32_t rowl = (uint32_t)(i - first);
        for (int base = 0; base < ns; base += kKrUnroll * kWave) {
            uint32_t key[kKrUnroll];
            int32_t id[kKrUnroll];
#pragma unroll
            for (int u = 0; u < kKrUnroll; ++u) {
                const int r = min(base + u * kWave + lane, ns - 1);        // clamped: every lane loads, nothing is predicated
                key[u] = (uint32_t)KR_LOAD(&row_keys[src + r]);
                if (MODE != KR_REGISTER) id[u] = KR_LOAD(&row_ids[src + r]);
            }
#pragma unroll
            for (int u = 0; u < kKrUnroll; ++u) {
                const int r = base + u * kWave + lane;
                if (r >= ns) continue;
                int32_t v = (int32_t)key[u];
                if (MODE == KR_FIND_WRITE) {
                    v = kr_lookup(dict, key[u]);
                    if (v < 0) {       // more distinct rows than the dictionary holds: the HBM table knows
                        const int32_t g = uniq_global_find(t, (unsigned long long)key[u]);
                        v = g >= 0 ? t.id[g] + 1 : 0;       // (0: a key nobody registered -- cannot happen after step 1)
                    }
                } else if (!kr_note(dict, key[u], rowl)) {   // a crowded dictionary: this member registers its key by itself
                    bool lowered = false;
                    (void)uniq_global_insert_ex(t, (unsigned long long)key[u],
                                                (unsigned long long)((root_base + i) * (int64_t)stride + (stride - 1)), flags, lowered);
                    if (lowered) atomicOr(&candbits[rowl >> 5], 1u << (rowl & 31u));
                }
                if (MODE != KR_REGISTER) {
                    KR_STORE(id[u], &out_indices[dst + r]);
                    KR_STORE(v, &out_data[dst + r]);      // SFptr+1, or the key itself (translated later)
                }
            }
        }
    }
    if (MODE == KR_FIND_WRITE) return;
    // ---- the range is done: its distinct keys go to the HBM table, each with the coarse tag of the smallest row that shows it
    __syncthreads();
    for (int s = tid; s < kKrDict; s += kKrThreads) {
        const uint32_t e = dict.k[s];
        if (e == kKrEmpty) continue;
        const uint32_t rowl = dict.v[s];
        bool lowered = false;
        (void)uniq_global_insert_ex(t, (unsigned long long)e,
                                    (unsigned long long)((root_base + first + rowl) * (int64_t)stride + (stride - 1)), flags, lowered);
        if (lowered) atomicOr(&candbits[rowl >> 5], 1u << (rowl & 31u));
    }
    __syncthreads();
    for (int s = tid; s < kKrMaxRows / 32; s += kKrThreads) {
        uint32_t bits = candbits[s];
        while (bits) {          // <= one entry per row: cand holds n
            const int b = __ffs(bits) - 1;
            bits &= bits - 1u;
            cand[atomicAdd(n_cand, 1ull)] = (int32_t)(first + s * 32 + b);
        }
    }
}

// flat pass over packed payloads: LP key -> SFptr+1 (the end of a job of several chunks: the numbering is known only now)
__global__ __launch_bounds__(kKrThreads) void keyrows_translate_kernel(int32_t *__restrict__ data, int64_t n, const int64_t *__restrict__ n_dev,
                                                                       UniqTable t, const unsigned long long *__restrict__ ukeys,
                                                                       const int64_t *__restrict__ n_ukeys, int64_t max_ukeys) {
    __shared__ uint32_t dmem[2 * kKrDict];
    const KrDict dict{dmem, dmem + kKrDict};
    const int tid = threadIdx.x;
    kr_clear(dict, tid);
    __syncthreads();
    kr_preload(dict, ukeys, n_ukeys, max_ukeys, tid);
    __syncthreads();
    if (n_dev && *n_dev < n) n = *n_dev;
    const int64_t per = (int64_t)kKrThreads * 4 * 8;           // 8 x 16 bytes per lane and block
    for (int64_t base = (int64_t)blockIdx.x * per; base < n; base += (int64_t)gridDim.x * per) {
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
            const int64_t e0 = base + ((int64_t)q * kKrThreads + tid) * 4;
            if (e0 >= n) break;
            if (e0 + 4 <= n && ((uintptr_t)(data + e0) & 15) == 0) {
                int4 kk = *(const int4 *)(data + e0);
                int32_t *kp = (int32_t *)&kk;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    int32_t v = kr_lookup(dict, (uint32_t)kp[x]);
                    if (v < 0) {
                        const int32_t g = uniq_global_find(t, (unsigned long long)(uint32_t)kp[x]);
                        v = g >= 0 ? t.id[g] + 1 : 0;
                    }
                    kp[x] = v;
                }
                *(int4 *)(data + e0) = kk;
            } else {
                for (int64_t e = e0; e < n && e < e0 + 4; ++e) {
                    int32_t v = kr_lookup(dict, (uint32_t)data[e]);
                    if (v < 0) {
                        const int32_t g = uniq_global_find(t, (unsigned long long)(uint32_t)data[e]);
                        v = g >= 0 ? t.id[g] + 1 : 0;
                    }
                    data[e] = v;
                }
            }
        }
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" int64_t subgacc_keyrows_cand_capacity(int64_t n) { return (n > 0 ? n : 0) + 16; }     // at most one entry per row

static int keyrows_launch(int mode, const int32_t *row_ids, const int32_t *row_keys, const int32_t *nsize, const int64_t *row_off,
                          int64_t n, int32_t stride, int64_t root_base, void *uniq_table, int64_t uniq_capacity,
                          const uint64_t *ukeys, const int64_t *n_ukeys, int64_t max_ukeys, int32_t *out_indices, int32_t *out_data,
                          int32_t *cand, int64_t cand_cap, int64_t *n_cand, int32_t *flags, void *stream) {
    SG_REQUIRE(n >= 0 && stride > 0 && root_base >= 0, SUBGACC_ERR_BADARG, "keyrows: bad sizes");
    SG_REQUIRE(uniq_table && uniq_capacity > 0 && (uniq_capacity & (uniq_capacity - 1)) == 0 && uniq_capacity < (1ll << 31),
               SUBGACC_ERR_BADARG, "keyrows: needs a power-of-two table of distinct rows");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(row_keys && flags && nsize, SUBGACC_ERR_BADARG, "keyrows: null argument");
    SG_REQUIRE(mode == KR_FIND_WRITE || (cand && n_cand && cand_cap >= subgacc_keyrows_cand_capacity(n)), SUBGACC_ERR_BADARG,
               "keyrows: the registering forms list their candidates (cand: subgacc_keyrows_cand_capacity(n) entries)");
    SG_REQUIRE(mode != KR_FIND_WRITE || (ukeys && n_ukeys && max_ukeys >= 0), SUBGACC_ERR_BADARG,
               "keyrows: the look-up form needs the numbered keys (subgacc_uniq_number's out_ukeys / out_count)");
    SG_REQUIRE(mode == KR_REGISTER || (row_ids && row_off && out_indices && out_data), SUBGACC_ERR_BADARG,
               "keyrows: the copying forms need the rows' ids, their packed offsets and the output arrays");
    const UniqTable t = uniq_view(uniq_table, uniq_capacity);
    // rows per block: a range long enough that its one flush of the dictionary does not matter, short enough that the launch has
    // several blocks per resident slot (4 blocks x 256 CUs) to even out what the rows' lengths leave uneven
    int64_t rpb = n / (4 * 256 * 4);
#ifdef SG_DEV_KR_RPB      // dev builds only (tools/keyrows_bench.py: -DSG_DEV_KR_RPB=n)
    rpb = SG_DEV_KR_RPB;
#endif
    rpb = rpb < 32 ? 32 : (rpb > kKrMaxRows ? kKrMaxRows : rpb);
    const int64_t grid = ceil_div(n, rpb);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "keyrows: too many rows in one call");
#define SG_KR_PASS(MODE)                                                                                                 \
    hipLaunchKernelGGL(keyrows_pass_kernel<MODE>, dim3((unsigned)grid), dim3(kKrThreads), 0, (hipStream_t)stream, row_ids, \
                       row_keys, nsize, row_off, n, stride, root_base, (int32_t)rpb, t, (const unsigned long long *)ukeys, \
                       n_ukeys, max_ukeys, out_indices, out_data, cand, (unsigned long long *)n_cand, flags)
    if (mode == KR_REGISTER) SG_KR_PASS(KR_REGISTER);
    else if (mode == KR_REGISTER_COPY) SG_KR_PASS(KR_REGISTER_COPY);
    else SG_KR_PASS(KR_FIND_WRITE);
#undef SG_KR_PASS
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_keyrows_register(const int32_t *row_keys, const int32_t *nsize, int64_t n, int32_t stride,
                                        int64_t root_base, void *uniq_table, int64_t uniq_capacity, int32_t *cand, int64_t cand_cap,
                                        int64_t *n_cand, int32_t *flags, void *stream) {
    return keyrows_launch(KR_REGISTER, nullptr, row_keys, nsize, nullptr, n, stride, root_base, uniq_table, uniq_capacity, nullptr,
                          nullptr, 0, nullptr, nullptr, cand, cand_cap, n_cand, flags, stream);
}

extern "C" int subgacc_keyrows_compact(const int32_t *row_ids, const int32_t *row_keys, const int32_t *nsize, const int64_t *row_off,
                                       int64_t n, int32_t stride, int64_t root_base, void *uniq_table, int64_t uniq_capacity,
                                       const uint64_t *ukeys, const int64_t *n_ukeys, int64_t max_ukeys, int32_t *out_indices,
                                       int32_t *out_data, int32_t *cand, int64_t cand_cap, int64_t *n_cand, int32_t *flags, void *stream) {
    return keyrows_launch(ukeys ? KR_FIND_WRITE : KR_REGISTER_COPY, row_ids, row_keys, nsize, row_off, n, stride, root_base,
                          uniq_table, uniq_capacity, ukeys, n_ukeys, max_ukeys, out_indices, out_data, cand, cand_cap, n_cand, flags,
                          stream);
}

extern "C" int subgacc_keyrows_translate(int32_t *data_inout, int64_t n, const int64_t *n_dev, void *uniq_table, int64_t uniq_capacity,
                                         const uint64_t *ukeys, const int64_t *n_ukeys, int64_t max_ukeys, void *stream) {
    SG_REQUIRE(n >= 0 && uniq_table && uniq_capacity > 0 && (uniq_capacity & (uniq_capacity - 1)) == 0 && ukeys && n_ukeys &&
                   max_ukeys >= 0, SUBGACC_ERR_BADARG, "keyrows_translate: bad arguments");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(data_inout, SUBGACC_ERR_BADARG, "keyrows_translate: null payload");
    int64_t grid = ceil_div(n, (int64_t)kKrThreads * 4 * 8);
    if (grid > 256 * 4 * 4) grid = 256 * 4 * 4;
    hipLaunchKernelGGL(keyrows_translate_kernel, dim3((unsigned)grid), dim3(kKrThreads), 0, (hipStream_t)stream, data_inout, n, n_dev,
                       uniq_view(uniq_table, uniq_capacity), (const unsigned long long *)ukeys, n_ukeys, max_ukeys);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
