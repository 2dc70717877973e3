// uniq_table.hpp -- the HBM open-addressing table of distinct LP rows (key -> min element position, id).
// Shared by uniq.hip (insert / number / translate) and walk.hip (insert fused into the set compaction).
#pragma once
#include "common.hpp"
#include "blockscan.hpp"

namespace subgacc {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kUniqItems = 4;  // elements per thread in the numbering passes
constexpr int kUniqTile = kScanThreads * kUniqItems;

struct UniqTable {
    unsigned long long *keys;    // [cap]
    unsigned long long *mintag;  // [cap]
    int32_t *id;                 // [cap]
    uint64_t mask;
};

__host__ __device__ inline UniqTable uniq_view(void *table, int64_t cap) {
    UniqTable t;
    t.keys = (unsigned long long *)table;
    t.mintag = t.keys + cap;
    t.id = (int32_t *)(t.mintag + cap);
    t.mask = (uint64_t)cap - 1;
    return t;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// Slot choice of the small per-set LDS fold tables (a few dozen distinct LP keys): four VALU instructions instead of
// mix64's two 64-bit multiplies -- the fused walk kernel is bound by VALU issue (profiles/r02e_sq_*.csv).  BITS = log2(slots).
template <int BITS>
__device__ __forceinline__ uint32_t fold_hash(unsigned long long key) {
    return (((uint32_t)key * 0x9E3779B1u) ^ ((uint32_t)(key >> 32) * 0x85EBCA77u)) >> (32 - BITS);
}

constexpr int kInsItems = 8;
constexpr int kInsTile = 256 * kInsItems;
constexpr int kInsLds = 1024;  // block-local table: a tile of 2048 members holds far fewer distinct LP rows
constexpr uint64_t kMaxProbes = 128;  // a longer chain means the table is over-full: report it, grow, retry

// one global insert: claim-or-find the key's slot, then lower its min position; returns the slot (or -1)
__device__ __forceinline__ int32_t uniq_global_insert(const UniqTable &t, unsigned long long key, unsigned long long tag,
                                                      int32_t *flags) {
    uint64_t h = mix64(key) & t.mask;
    for (uint64_t probes = 0; probes <= t.mask && probes < kMaxProbes; ++probes) {
        // Plain (cacheable) loads first.  A slot's key never changes once set and its tag only decreases, so a stale
        // copy in this XCD's L2 / this CU's L1 can only make us take the atomic path below once more than needed --
        // never skip a needed one: stale "empty" -> the CAS returns the real content; stale larger tag -> a redundant
        // atomicMin.  (Device-scope loads, the first version, leave the XCD for the fabric every time: two serial
        // ~2 us round trips per distinct LP row in every root's epilogue.)
        unsigned long long cur = t.keys[h];
        const unsigned long long seen = t.mintag[h];   // asked for together with the key: one round trip, not two
        if (cur == kEmptyKey) cur = atomicCAS(&t.keys[h], kEmptyKey, key);
        if (cur == kEmptyKey || cur == key) {
            if (seen > tag) atomicMin(&t.mintag[h], tag);
            return (int32_t)h;
        }
        h = (h + 1) & t.mask;
    }
    atomicOr(&flags[2], 1);  // table (nearly) full
    return -1;
}


}  // namespace subgacc
