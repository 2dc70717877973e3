// walk_common.hpp -- pieces shared by the walk kernels (walk.hip: one root per workgroup; walk_pipe.hip: persistent,
// software-pipelined): RNGs, CSR row access, kernel arguments, LDS sizing.
#pragma once
#include "common.hpp"
#include "blockscan.hpp"
#include "uniq_table.hpp"

namespace subgacc {

constexpr int kWalkThreads = 256;
constexpr uint32_t kLcgA = 1103515245u, kLcgC = 12345u;
constexpr int kNeighCap = 1000000;  // NEBMAX, subg_acc.c:13

// ---------------------------------------------------------------------------------------- RNGs
__device__ __forceinline__ uint32_t lcg_jump(uint32_t x, uint32_t k) {
    uint32_t a = kLcgA, c = kLcgC;
    while (k) {
        if (k & 1u) x = a * x + c;
        c *= (a + 1u);
        a *= a;
        k >>= 1;
    }
    return x;
}
__device__ __forceinline__ uint32_t rand_r_next(uint32_t &x) {
    uint32_t r;
    x = x * kLcgA + kLcgC;
    r = (x >> 16) & 2047u;
    x = x * kLcgA + kLcgC;
    r = (r << 10) ^ ((x >> 16) & 1023u);
    x = x * kLcgA + kLcgC;
    r = (r << 10) ^ ((x >> 16) & 1023u);
    return r;
}
// Philox2x32-10 (same paper): ONE 32x32 multiply pair per round and two draws per call -- what a walk of up to three hops
// needs after its first hop.  The walk kernels are bound by VALU issue (profiles/r02e_sq_*.csv: 85 % busy); the 4x32
// form cost 72 vector instructions per walk, this one 30.
__device__ __forceinline__ void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t k, uint32_t &o0, uint32_t &o1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p = (unsigned long long)0xD256D193u * c0;      // ONE v_mad_u64_u32: it issues like a single 32-bit multiply
        c0 = (uint32_t)(p >> 32) ^ k ^ c1;
        c1 = (uint32_t)p;
        k += 0x9E3779B9u;
    }
    o0 = c0, o1 = c1;
}
// Stream layout of rng = "philox" (the CPU checker restates it): key = seed, counter word 0 = root id, counter
// word 1 = lane (walk number or shuffle position, < 2^20) | block << 20 (draws 2*block, 2*block+1) | tag << 28
// (0: first hop without replacement, 1: every hop drawn) | shuffle-stream flag << 31.  A draw r picks index
// (r * n) >> 32 below n: one multiply instead of the ~20 instructions of a 32-bit remainder (the rand_r mode keeps the
// reference's %).
constexpr int kPhiloxBlockShift = 20, kPhiloxTagShift = 28;
constexpr uint32_t kPhiloxShuffle = 0x80000000u;
__device__ __forceinline__ uint32_t philox_below(uint32_t r, uint32_t n) { return __umulhi(r, n); }

// neighbour reads of the walk: SG_NT_NEIGH=1 marks them non-temporal (a random 4-byte read of `indices` pulls a line
// that is rarely used again; streaming it keeps the row-pointer lines in L2) -- see tools/ab_walk.sh
#ifndef SG_NT_NEIGH
#define SG_NT_NEIGH 0
#endif
#if SG_NT_NEIGH
#define SG_NEIGH_LOAD(p) __builtin_nontemporal_load(p)
#else
#define SG_NEIGH_LOAD(p) (*(p))
#endif

// ------------------------------------------------------------------- rand_r stream positions
template <bool IDX64>
__device__ __forceinline__ void load_row(const void *indptr, int32_t node, int64_t &beg, int64_t &deg) {
    if (IDX64) {
        const int64_t *p = (const int64_t *)indptr + node;
        beg = p[0];
        deg = p[1] - beg;
    } else {
        const int32_t *p = (const int32_t *)indptr + node;
        const int32_t b = p[0], e = p[1];
        beg = b;
        deg = e - b;
    }
}

// ------------------------------------------------------------------- packed hop records (experiment, round 2)
// rec = [neighbour id : id_bits | its row begin : beg_bits | its degree : 64 - id_bits - beg_bits]: ONE dependent 8-byte read
// per hop instead of a neighbour read followed by a row-pointer read.  A degree field of all ones means "does not fit":
// the walk reads the row pointers for that node.
struct RecFmt {
    int32_t id_bits, beg_bits;
};
__device__ __forceinline__ void rec_unpack(unsigned long long r, const RecFmt &f, int32_t &id, int64_t &beg, int64_t &deg,
                                           bool &escaped) {
    const int deg_bits = 64 - f.id_bits - f.beg_bits;
    const unsigned long long dmask = (1ull << deg_bits) - 1ull;
    id = (int32_t)(r >> (64 - f.id_bits));
    beg = (int64_t)((r >> deg_bits) & ((1ull << f.beg_bits) - 1ull));
    deg = (int64_t)(r & dmask);
    escaped = (r & dmask) == dmask;
}

struct WalkArgs {
    const void *indptr;
    const int32_t *indices;
    const int32_t *query;
    int64_t n;
    const int32_t *worklist;   // optional (fused-row kernel only): the rows to sample, worklist[0 .. *n_work) -- a batch's first
    const int64_t *n_work;     // occurrences (subgacc_step_prologue_dedup); without it row i = block i and query[i] == SUBGACC_NO_ROOT is passed over
    int64_t num_nodes;   // a root outside [0, num_nodes) is never looked up: empty set, flags[3] |= 16 (the host raises)
    const uint32_t *rng_pos, *rng_seed;
    int32_t *set_ids;
    uint64_t *set_keys;
    int32_t *nsize;
    int32_t *walks;
    int32_t *flags;
    int32_t M, m, stride, shift;
    int32_t pitch;       // words between two roots' rows in set_ids / set_slot / set_keys (>= stride; subgacc_walk_cfg::row_pitch)
    int32_t wide_rows;   // every row begins on a 16-byte boundary in set_ids and set_slot (pitch % 4 == 0, aligned bases): 16-byte row stores
    int32_t T, tshift;   // table size (pow2) and 32-log2(T)
    int32_t nwords;      // bitmap words over q in [0, M*m]
    uint32_t seed;
    int32_t wo, step_major, cap_root;
    // SPG mode (walk_spg_kernel): rows leave sorted by node id with the slot of their LP key in the HBM table
    int32_t *set_slot;
    UniqTable table;
    int64_t root_base;   // global index of query[0]: tags (root_base+i)*stride + rank order the first occurrences
    int32_t keyrows;     // fused rows carry the member's 32-bit LP key itself instead of a table slot (no table of distinct rows)
    const unsigned long long *recs;   // packed hop records (walk_rows_kernel<REC>), else NULL
    RecFmt rec;                       // id_bits = 0: the 16-byte form {id : 32 | degree : 32, row begin : 64} (int64 row offsets)
    const uint32_t *walk_pos;   // rand_r on a graph with dead ends (replay.hip): the stream position of every WALK, [n*M]; else NULL
    int64_t work_cap;    // with a work list: the launch covers min(n, work_cap) rows (0 = n); a longer list raises flags[3] |= 32
    int32_t tags_only;   // table-form fused rows (subgacc_walk_tags): register the set's LP keys with their exact first-visit tags
                         // and leave -- no row, no nsize is written (the rows exist already, as key rows)
};

constexpr int kSpgFoldBits = 7;
constexpr int kSpgFold = 1 << kSpgFoldBits;   // block-local table of the set's distinct LP keys
constexpr int kSpgPerLane = 4;     // members per lane kept in registers while LDS is re-used => M*m+1 <= 1024

static inline int table_size_for(int64_t q) {
    int64_t want = q + q / 4 + 1;
    int t = 64;
    while (t < want) t <<= 1;
    return t;
}

static inline size_t walk_lds_bytes(int T, int nwords, int M, int Q, bool spg, bool bucket_truncates) {
    const size_t head = (size_t)T * 16 + (size_t)nwords * 4 + (size_t)(nwords + 1) * 4 + (size_t)M * 4;
    const size_t fold = 8 + (size_t)kSpgFold * 16 + 64;
    if (!spg) return head + (size_t)Q * 2 + 16;
    if (bucket_truncates) return head + (size_t)Q * 2 + fold + 16;
    return head + (fold > (size_t)Q * 2 ? fold : (size_t)Q * 2) + 16;   // the fold table overlays inv
}


// walk_rows_kernel with key rows: where its 16 reduction words live -- behind the walk tables (counts + ids: 8 bytes per slot, 12
// with 64-bit counts; then the Fisher-Yates draws) AND behind what the epilogue's sort lays over them (sort elements: 8 bytes per
// member, + 8 for a 64-bit key; level-1 offsets [NT+1] (+3: the next array begins on 16 bytes); the 16-bit counters of levels 2 and
// 3, 4 * NT words whatever the set's size: a lane zeroes, scans and reads its four as ONE 16-byte LDS access).  The tables usually are the larger part; few walks of many hops (M = 100, m = 4: 401 members in
// a 512-slot table, 400 bytes of draws) and the 512-slot tables with 128 lanes turn that round.
__host__ __device__ static inline size_t kr_red_offset(int T, int M, int stride, int NT, bool wide) {
    const size_t walk = (size_t)T * (wide ? 12 : 8) + (size_t)M * 4 + 8;
    const size_t sort = (size_t)stride * (wide ? 16 : 8) + (size_t)(NT + 4) * 4 + (size_t)(4 * NT) * 4;
    return ((walk > sort ? walk : sort) + 15) & ~(size_t)15;
}

// walk_pipe.hip: persistent software-pipelined form of the walk kernel; returns 1 when it took the launch
int launch_walk_pipe(const WalkArgs &a, bool indptr64, int rng_mode, bool spg, size_t lds, hipStream_t s);
// walk_rows.hip: the fused-row form specialised by hop count and table size; returns 1 when it took the launch
int launch_walk_rows(const WalkArgs &a, bool indptr64, int rng_mode, size_t lds, hipStream_t s);

}  // namespace subgacc
