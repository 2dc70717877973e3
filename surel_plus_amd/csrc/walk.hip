// walk.hip -- random-walk node-set sampling with landing-probability accumulation (gfx950).
//
// Replaces the hot loop of set_sampler (reference subg_acc/subg_acc.c:742-846), random_walk (:144-180),
// random_walk_wo (:183-247) and rpe_encoder (:249-314).  Not a translation: the reference walks one root
// per OpenMP thread with a malloc'd uthash per root; here ONE 256-lane workgroup owns one root, one lane
// per walk, and the per-root node set lives in LDS:
//
//   keys[T]  int32   open-addressing table of visited node ids (T = pow2 > 1.25*(M*m+1))
//   minq[T]  uint32  smallest visit sequence number of the key  (ds_min_u32)
//   pk[T]    uint64  packed landing counts of the key           (ds_add_u64: count of step s lives in
//                    bits [(m-1-s)*SHIFT, +SHIFT), the reference's `bithash` layout, :936-949)
//
// The reference's slot number of a node is its rank in sequential first-visit order.  Parallel lanes
// cannot produce that order directly, so every visit carries its sequence number q (walk-major
// q = w*m+s+1, or step-major q = s*M+w+1 for rpe_encoder; the root is q = 0); after the walks a bitmap
// over q marks the first visits and slot = popcount of the bitmap below minq -- a wave-ballot-style
// prefix sum instead of a sequential hash iteration.  Integer results are bit-exact with the oracle.
//
// RNG: SUBGACC_RNG_RAND_R reproduces glibc rand_r's single sequential stream (the reference at
// nthread=1): the LCG x -> a*x+c is affine, so a lane jumps straight to the stream position of its
// (root, walk) in O(log k).  SUBGACC_RNG_PHILOX is Philox4x32-10 keyed by (seed; root id, walk, step).
#include "common.hpp"
#include "blockscan.hpp"
#include "uniq_table.hpp"
#include "walk_common.hpp"
#include <stdlib.h>

namespace subgacc {

template <bool IDX64>
__global__ void rng_calls_kernel(const void *__restrict__ indptr, const int32_t *__restrict__ query, int64_t n,
                                 int64_t num_nodes, int M, int m, int wo, int cap, int32_t *__restrict__ calls) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t beg, deg = 0;
    if ((uint64_t)(int64_t)query[i] < (uint64_t)num_nodes) load_row<IDX64>(indptr, query[i], beg, deg);   // else: the walk kernel flags it
    if (cap && deg > kNeighCap) deg = kNeighCap;
    int32_t c = 0;
    if (deg > 0) c = wo ? ((deg > M ? M : 0) + M * (m - 1)) : M * m;
    calls[i] = c;
}

// libgomp static schedule: the first n%T threads own ceil(n/T) iterations
__global__ void rng_positions_kernel(const int64_t *__restrict__ calls_excl, int64_t n, int32_t streams,
                                     uint64_t calls_before, uint32_t seed, uint32_t *__restrict__ pos,
                                     uint32_t *__restrict__ sd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t q = n / streams, r = n % streams;
    int64_t t, lo;
    if (i < r * (q + 1)) {
        t = i / (q + 1);
        lo = t * (q + 1);
    } else {
        t = r + (q > 0 ? (i - r * (q + 1)) / q : 0);
        lo = r * (q + 1) + (t - r) * q;
    }
    uint64_t c = (uint64_t)(calls_excl[i] - calls_excl[lo]);
    if (streams == 1) c += calls_before;
    // three LCG steps per rand_r call; the LCG has period 2^32.  The pair leaves NORMALISED: (state of the stream at the root's
    // first draw, 0 further steps) names the same point of the stream as (seed + t, 3c) and spares every workgroup of the walk
    // kernels the 32-round jump from the seed (they compute lcg_jump(seed, pos + ...): 0 rounds for pos = 0)
    pos[i] = 0u;
    sd[i] = lcg_jump(seed + (uint32_t)t, (uint32_t)(3ull * c));
}

// ------------------------------------------------------------------------------ the walk kernel
// <= 80 SGPRs keeps 8 workgroups (32 waves) resident per CU; the allocator would otherwise take ~100 and the
// hardware admits only 6 (MI355X_MICROARCH.md, residency formula) -- worth 14 % on the L2-resident collab graph
#ifndef SG_WALK_SGPR
#define SG_WALK_SGPR 80
#endif
#ifndef SG_WALK_MINW
#define SG_WALK_MINW 8
#endif
template <bool IDX64, int RNG, bool SPG>
__global__ __launch_bounds__(kWalkThreads, SG_WALK_MINW) __attribute__((amdgpu_num_sgpr(SG_WALK_SGPR))) void walk_sets_kernel(const WalkArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    SG_HOOK_KERNEL_ENTRY();
    unsigned long long *pk = (unsigned long long *)lds_raw;     // [T]
    int32_t *keys = (int32_t *)(pk + a.T);                       // [T]
    uint32_t *minq = (uint32_t *)(keys + a.T);                   // [T]
    uint32_t *bitmap = minq + a.T;                               // [nwords]
    uint32_t *prefix = bitmap + a.nwords;                        // [nwords + 1]
    int32_t *sarr = (int32_t *)(prefix + a.nwords + 1);          // [M] Fisher-Yates draws
    uint16_t *inv = (uint16_t *)(sarr + a.M);                    // [M*m+1] table slot of the member ranked r
    // SPG mode only: fold table of distinct LP keys + a reduction scratch.  Without a truncating bucket the
    // ranking arrays (bitmap / prefix / inv) are never touched and the fold table takes their place, which keeps
    // the footprint at 8 workgroups per CU; with a bucket it sits behind inv.
    const bool fold_aliased = SPG && !(a.stride < a.M * a.m + 1);
    unsigned long long *fk = fold_aliased ? (unsigned long long *)(((uintptr_t)(sarr + a.M) + 7) & ~(uintptr_t)7)
                                          : (unsigned long long *)(((uintptr_t)(inv + (a.M * a.m + 1)) + 7) & ~(uintptr_t)7);
    uint32_t *ft = (uint32_t *)(fk + kSpgFold);                  // [kSpgFold] min rank of the key inside the set
    int32_t *fs = (int32_t *)(ft + kSpgFold);                    // [kSpgFold] HBM table slot of the key
    int32_t *red = fs + kSpgFold;                                // [16]

    const int64_t i = xcd_item(blockIdx.x, gridDim.x);
    if (i >= a.n) return;
    const int tid = threadIdx.x;
    const int M = a.M, m = a.m, T = a.T;
    const int32_t root = a.query[i];
    // while the root's two dependent loads (query -> row pointer) are in flight: clear what does not depend on them
    for (int h = tid; h < T; h += kWalkThreads) pk[h] = 0ull;
    for (int x = tid; x < a.nwords; x += kWalkThreads) bitmap[x] = 0u;
    if (SPG) {
        for (int s2 = tid; s2 < kSpgFold; s2 += kWalkThreads) {
            fk[s2] = kEmptyKey;
            ft[s2] = 0xFFFFFFFFu;
        }
        if (tid < 16) red[tid] = tid < 4 ? 0x7FFFFFFF : 0;   // [0..3] min id per wave, [4..7] max id, [8] member count
    }
    if ((uint64_t)(int64_t)root >= (uint64_t)a.num_nodes) {   // the reference would read out of bounds here (no checks)
        if (tid == 0) {
            // (SUBGACC_NO_ROOT marks a row that has no root -- a repeated endpoint of a batch whose first occurrence carries the set: an
            //  empty row, not an error, in this kernel as in the fused-row one; round 5's fallback of subgacc_walk_spg_list flagged it)
            if (root != SUBGACC_NO_ROOT) atomicOr(&a.flags[3], 16);
            a.nsize[i] = 0;
        }
        return;
    }
    int64_t rbeg, rdeg64;
    load_row<IDX64>(a.indptr, root, rbeg, rdeg64);
    {   // keys / minq, with the root already in its slot as member 0 (q = 0): no separate insert phase
        const uint32_t hroot = ((uint32_t)root * 2654435761u) >> a.tshift;
        for (int h = tid; h < T; h += kWalkThreads) {
            const bool isroot = (uint32_t)h == hroot;
            keys[h] = isroot ? root : -1;
            minq[h] = isroot ? 0u : 0xFFFFFFFFu;
        }
    }
    if (a.cap_root && rdeg64 > kNeighCap) rdeg64 = kNeighCap;
    const int64_t obase = i * (int64_t)a.pitch;
    const unsigned long long lead = 1ull << (m * a.shift);

    if (rdeg64 == 0) {  // isolated root: one member, every count = M (subg_acc.c:753-761); id = the root
        if (tid == 0) {
            unsigned long long k = lead;
            for (int s = 0; s < m; ++s) k |= (unsigned long long)M << (s * a.shift);
            a.set_ids[obase] = root;
            if (SPG) a.set_slot[obase] = uniq_global_insert(a.table, k, (unsigned long long)((a.root_base + i) * a.pitch), a.flags);
            else a.set_keys[obase] = k;
            a.nsize[i] = 1;
        }
        if (a.walks)
            for (int x = tid; x < M * (m + 1); x += kWalkThreads) a.walks[i * (int64_t)M * (m + 1) + x] = root;
        return;
    }

    uint32_t rpos = 0, rseed = a.seed;
    if (RNG == SUBGACC_RNG_RAND_R) {
        rpos = a.rng_pos[i];
        rseed = a.rng_seed[i];
    }
    const bool shuffled = a.wo && rdeg64 > M;
    const uint32_t rdeg = (uint32_t)rdeg64;
    const uint32_t ptag = (a.wo ? 0u : 1u) << kPhiloxTagShift;
    if (shuffled) {  // partial Fisher-Yates draws s_k = draw % (deg-k) + k  (subg_acc.c:769-775), one lane per k
        for (int k = tid; k < M; k += kWalkThreads) {
            uint32_t r;
            if (RNG == SUBGACC_RNG_RAND_R) {
                uint32_t x = lcg_jump(rseed, rpos + 3u * (uint32_t)k);
                r = rand_r_next(x);
                sarr[k] = (int32_t)(r % (rdeg - (uint32_t)k)) + k;
            } else {
                uint32_t o1;
                philox2x32_10((uint32_t)root, (uint32_t)k | ptag | kPhiloxShuffle, a.seed, r, o1);
                sarr[k] = (int32_t)philox_below(r, rdeg - (uint32_t)k) + k;
            }
        }
    }
    __syncthreads();
    SG_HOOK_STAMP(0);
    SG_HOOK_STAMP(1);

    const uint32_t tmask = (uint32_t)T - 1u;
    int32_t vmin = root, vmax = root;   // id range of everything this lane visits (SPG mode: bucket scaling)
    for (int w = tid; w < M; w += kWalkThreads) {
        // ---- first hop
        int32_t cur = root;
        uint32_t x = 0;         // rand_r state of this walk
        uint32_t ph[2];         // cached Philox block
        int ph_blk = -1;
        if (RNG == SUBGACC_RNG_RAND_R) {
            const uint32_t per_walk = (uint32_t)(a.wo ? m - 1 : m);
            // (walk_pos: the graph has dead ends, the position of every walk was found by replaying the stream, replay.hip)
            x = a.walk_pos ? lcg_jump(rseed, a.walk_pos[i * (int64_t)M + w])
                           : lcg_jump(rseed, rpos + 3u * ((shuffled ? (uint32_t)M : 0u) + (uint32_t)w * per_walk));
        }
        int32_t *wrow = a.walks ? a.walks + (i * (int64_t)M + w) * (m + 1) : nullptr;
        if (wrow) wrow[0] = root;
        for (int s = 0; s < m; ++s) {
            if (s == 0 && a.wo) {
                uint32_t pick;
                if (shuffled) {
                    // value that the sequential swaps leave at position w: follow the chain of earlier
                    // draws that displaced it (one downward pass over the draws, broadcast LDS reads)
                    int32_t p = sarr[w];
                    for (int j = w - 1; j >= 0; --j)
                        if (sarr[j] == p) p = j;
                    pick = (uint32_t)p;
                } else {
                    pick = (uint32_t)w % rdeg;
                }
                cur = SG_NEIGH_LOAD(&a.indices[rbeg + pick]);
                if (RNG == SUBGACC_RNG_PHILOX && m > 1) {   // the draws of hops 2.. while that load is in flight
                    ph_blk = 0;
                    philox2x32_10((uint32_t)root, (uint32_t)w | ptag, a.seed, ph[0], ph[1]);
                }
            } else {
                SG_HOOK_BEFORE_HOP(cur, w, s);
                int64_t b, d;
                SG_HOOK_LOAD_ROW(IDX64, a.indptr, cur, b, d);
                if (d > 0) {
                    uint32_t r;
                    if (RNG == SUBGACC_RNG_RAND_R) {
                        r = rand_r_next(x);
                    } else {
                        const int idx = a.wo ? s - 1 : s;
                        if ((idx >> 1) != ph_blk) {
                            ph_blk = idx >> 1;
                            philox2x32_10((uint32_t)root, (uint32_t)w | ((uint32_t)ph_blk << kPhiloxBlockShift) | ptag, a.seed,
                                          ph[0], ph[1]);
                        }
                        r = ph[idx & 1];
                    }
                    cur = SG_NEIGH_LOAD(&a.indices[b + (int64_t)(RNG == SUBGACC_RNG_RAND_R ? r % (uint32_t)d
                                                                                           : philox_below(r, (uint32_t)d))]);
                } else if (RNG == SUBGACC_RNG_RAND_R && !a.walk_pos) {
                    atomicOr(&a.flags[0], 1);  // dead end: the positions computed from the degrees no longer hold (the host replays)
                }
            }
            SG_HOOK_VISIT_LABEL
            if (wrow) wrow[s + 1] = cur;
            SG_HOOK_BEFORE_VISIT(cur, pk);
            // ---- visit: insert-or-find, first-visit sequence number, landing count
            uint32_t h = ((uint32_t)cur * 2654435761u) >> a.tshift;
            while (true) {
                const int32_t old = atomicCAS(&keys[h], -1, cur);
                if (old == -1 || old == cur) break;
                h = (h + 1u) & tmask;
            }
            const uint32_t q = a.step_major ? (uint32_t)(s * M + w + 1) : (uint32_t)(w * m + s + 1);
            if (SPG) {
                vmin = min(vmin, cur);
                vmax = max(vmax, cur);
            }
            atomicMin(&minq[h], q);
            atomicAdd(&pk[h], 1ull << ((m - 1 - s) * a.shift));
        }
    }
    if (SPG) {
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            vmin = min(vmin, __shfl_xor(vmin, d, kWave));
            vmax = max(vmax, __shfl_xor(vmax, d, kWave));
        }
        if ((tid & (kWave - 1)) == 0) {
            red[tid / kWave] = vmin;
            red[4 + tid / kWave] = vmax;
        }
    }
    __syncthreads();
    SG_HOOK_STAMP(2);

    // ---- rank the members by first visit: bitmap over q, popcount prefix.  The SPG mode only needs an ORDER of
    // first visits for its tags -- the visit sequence number itself is one -- so it ranks only when a bucket can
    // truncate the set (members ranked >= bucket are dropped, subg_acc.c:814-828)
    const bool need_rank = !SPG || a.stride < M * m + 1;
    int32_t total = 0, ns = 0;
    if (need_rank) {
        for (int h = tid; h < T; h += kWalkThreads)
            if (keys[h] != -1) {
                const uint32_t q = minq[h];
                atomicOr(&bitmap[q >> 5], 1u << (q & 31u));
            }
        __syncthreads();
        for (int x = tid; x <= a.nwords; x += kWalkThreads) {
            uint32_t s = 0;
            for (int j = 0; j < x; ++j) s += __popc(bitmap[j]);
            prefix[x] = s;
        }
        __syncthreads();
        total = (int32_t)prefix[a.nwords];
        ns = total < a.stride ? total : a.stride;
        for (int h = tid; h < T; h += kWalkThreads)
            if (keys[h] != -1) {
                const uint32_t q = minq[h];
                const int32_t r = (int32_t)(prefix[q >> 5] + __popc(bitmap[q >> 5] & ((1u << (q & 31u)) - 1u)));
                if (r < a.stride) {  // members ranked past the bucket are dropped with all their visits (:814-828)
                    if (SPG) {
                        inv[r] = (uint16_t)h;
                    } else {         // straight to the staging row (a second LDS pass to coalesce these stores
                                     // buys nothing: the kernel is bound by its random reads, measured)
                        a.set_ids[obase + r] = keys[h];
                        a.set_keys[obase + r] = pk[h] | (r == 0 ? lead : 0ull);
                    }
                }
            }
        if (SPG) __syncthreads();
    }
    if (!SPG) {
        if (tid == 0) {
            a.nsize[i] = ns;
            if (total > a.stride) atomicAdd(&a.flags[1], 1);
        }
        return;
    }

    // ================= SPG mode: the set leaves as a finished SpG row =================
    // (1) fold the set's LP keys (a few dozen distinct rows) and register them in the HBM table of distinct rows
    //     with tag = (global root index)*stride + first-visit order, which orders first occurrences exactly like the
    //     reference's sequential pass (subg_acc.c:957-978); (2) bucket-sort the members by node id in the LDS the
    //     walk tables occupied (random_walks.py:79-80) and write them to their sorted position.
    const unsigned long long tag0 = (unsigned long long)((a.root_base + i) * (int64_t)a.pitch);
    int32_t idv[kSpgPerLane], slv[kSpgPerLane];
    bool ok[kSpgPerLane];
    int mycount = 0;
    // The lane's members are handled stage by stage, two at a time, not one after the other: a stage issues its LDS
    // reads for both before either is consumed, so their latencies overlap (a workgroup's lifetime is a chain of such
    // round trips; 8 workgroups per CU is all the latency hiding there is).
    constexpr int kIlp = 2;   // members in flight per stage (4 would need more than the 64 VGPRs that keep 8 workgroups per CU)
#pragma unroll
    for (int u0 = 0; u0 < kSpgPerLane; u0 += kIlp) {
        unsigned long long mkey[kIlp], fcur[kIlp];
        uint32_t mtag[kIlp], mf[kIlp], ftag[kIlp];
#pragma unroll
        for (int v = 0; v < kIlp; ++v) {   // stage 1: the member (by rank in bucket mode, else straight from its slot)
            const int u = u0 + v;
            const int x = tid + u * kWalkThreads;
            int h = x;
            if (need_rank) {
                ok[u] = x < ns;
                h = ok[u] ? (int)inv[x] : 0;
                mtag[v] = (uint32_t)x;
                idv[u] = keys[h];
            } else {
                ok[u] = x < T;
                h = ok[u] ? x : 0;
                idv[u] = keys[h];
                mtag[v] = minq[h];
                ok[u] = ok[u] && idv[u] != -1;
            }
            mkey[v] = pk[h];
        }
#pragma unroll
        for (int v = 0; v < kIlp; ++v) {   // stage 2: key, fold slot, first probe
            const int u = u0 + v;
            mkey[v] |= (mtag[v] == 0 ? lead : 0ull);               // the root is rank 0 / visit 0
            mf[v] = fold_hash<kSpgFoldBits>(mkey[v]);
            slv[u] = -1;
            if (!ok[u]) idv[u] = 0;
            mycount += ok[u] ? 1 : 0;
        }
#pragma unroll
        for (int v = 0; v < kIlp; ++v) {
            fcur[v] = fk[mf[v]];
            ftag[v] = ft[mf[v]];
        }
#pragma unroll
        for (int v = 0; v < kIlp; ++v) {   // stage 3: a set holds a few dozen distinct keys -> mostly a hit right away
            const int u = u0 + v;
            if (!ok[u]) continue;
            const unsigned long long key = mkey[v];
            const uint32_t tagoff = mtag[v];
            if (fcur[v] == key) {
                // most lanes meet a tag that is already smaller: the plain read (stale only towards larger values)
                // spares the same-address atomic storm
                if (ftag[v] > tagoff) atomicMin(&ft[mf[v]], tagoff);
                slv[u] = -2 - (int32_t)mf[v];       // resolved to the HBM slot after the fold table is flushed
                continue;
            }
            uint32_t f = mf[v];
            bool done = false;
            for (int p = 0; p < 16; ++p) {
                unsigned long long cur = fk[f];
                if (cur == kEmptyKey) cur = atomicCAS(&fk[f], kEmptyKey, key);
                if (cur == kEmptyKey || cur == key) {
                    if (ft[f] > tagoff) atomicMin(&ft[f], tagoff);
                    slv[u] = -2 - (int32_t)f;
                    done = true;
                    break;
                }
                f = (f + 1) & (kSpgFold - 1);
            }
            if (!done) slv[u] = uniq_global_insert(a.table, key, tag0 + (unsigned long long)tagoff, a.flags);
        }
    }
    if (!need_rank) {
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) mycount += __shfl_xor(mycount, d, kWave);
        if ((tid & (kWave - 1)) == 0) atomicAdd(&red[8], mycount);
    }
    __syncthreads();   // every lane holds its members in registers: the walk tables are free to be re-used
    SG_HOOK_STAMP(3);
    if (!need_rank) total = ns = red[8];
    if (tid == 0) {
        a.nsize[i] = ns;
        if (total > a.stride) atomicAdd(&a.flags[1], 1);
    }
    const int32_t mn = min(min(red[0], red[1]), min(red[2], red[3]));
    const int32_t mx = max(max(red[4], red[5]), max(red[6], red[7]));
    unsigned long long *A = pk;                 // [ns] (id << 32 | slot) grouped by bucket
    int32_t *start = keys;                      // [B+1]
    int32_t *cursor = keys + (T / 4) + 1;       // [B]      B <= min(T/4, 256)
    int logb = 0;
    while ((1 << logb) < ns && (2 << logb) <= T / 4 && (2 << logb) <= kWalkThreads) ++logb;
    const int B = 1 << logb;
    const uint32_t range = (uint32_t)(mx - mn) + 1u;
    const int Ls = (range <= 1u) ? 0 : (32 - __builtin_clz(range - 1u));
    const int bshift = Ls > logb ? Ls - logb : 0;
    if (tid < B) cursor[tid] = 0;
    for (int s2 = tid; s2 < kSpgFold; s2 += kWalkThreads)    // flush the fold table to HBM (latency overlaps the sort)
        if (fk[s2] != kEmptyKey) fs[s2] = SG_HOOK_FLUSH_SLOT(s2, uniq_global_insert(a.table, fk[s2], tag0 + ft[s2], a.flags));
    __syncthreads();
    SG_HOOK_STAMP(4);
    uint32_t bk[kSpgPerLane];
#pragma unroll
    for (int u = 0; u < kSpgPerLane; ++u) {
        bk[u] = (uint32_t)(idv[u] - mn) >> bshift;        // (id - mn) < 2^Ls and Ls >= logb unless the row is a single id
        if (ok[u]) atomicAdd(&cursor[bk[u]], 1);
        if (slv[u] <= -2) slv[u] = fs[-2 - slv[u]];
    }
    __syncthreads();
    SG_HOOK_STAMP(5);
    int32_t maxc;
    {   // exclusive scan over the B <= 256 buckets, one bucket per lane: wave scan, then the wave totals through LDS
        const int32_t c = tid < B ? cursor[tid] : 0;
        int32_t inc = c, mc = c;
#pragma unroll
        for (int dd = 1; dd < kWave; dd <<= 1) {
            const int32_t t2 = __shfl_up(inc, dd, kWave);
            if ((tid & (kWave - 1)) >= dd) inc += t2;
            mc = max(mc, __shfl_xor(mc, dd, kWave));
        }
        if ((tid & (kWave - 1)) == kWave - 1) red[12 + tid / kWave] = inc, red[4 + tid / kWave] = mc;   // (the id range is in registers)
        __syncthreads();
        int32_t base = 0;
#pragma unroll
        for (int w2 = 0; w2 < kWalkThreads / kWave - 1; ++w2) base += w2 < tid / kWave ? red[12 + w2] : 0;      // (spelled out: as a loop over tid / 64 the compiler vectorises it)
        maxc = max(max(red[4], red[5]), max(red[6], red[7]));
        const int32_t excl = base + inc - c;
        if (tid < B) {
            start[tid] = excl;
            cursor[tid] = excl;
        }
        if (tid == B - 1) start[B] = excl + c;
    }
    __syncthreads();
    SG_HOOK_STAMP(6);
    // A crowded bucket -- a set whose ids sit in one community of a graph with id locality -- made the ranking by counting below
    // quadratic (walk_rows.hip met the same).  The same second level here: bucket b gets as many sub-buckets
    // as it has members (sub = offset inside b's id window scaled by b's count), so that start[b] + sub is a monotone map of the ids
    // onto [0, ns) that follows the set's own distribution, and the counting runs over ~1 member.  Evenly spread ids keep the
    // short path.  The level-2 counters (16 bits each) live behind the level-1 offsets and cursors, in the dead id table.
    constexpr int kFineAbove = 12, CW = 2;
    const int W2 = (ns + 2) / 2 + 1;
    uint32_t *cnt2 = (uint32_t *)(keys + T / 2 + 2);                     // [W2] <= T/2 - 2 words (W2 <= 0.4 T + 2)
    int blo[kSpgPerLane], bhi[kSpgPerLane];
    if (!(maxc > kFineAbove && W2 <= CW * kWalkThreads && W2 <= T / 2 - 2)) {
#pragma unroll
        for (int u = 0; u < kSpgPerLane; ++u)
            if (ok[u]) A[atomicAdd(&cursor[bk[u]], 1)] = ((unsigned long long)(uint32_t)idv[u] << 32) | (uint32_t)slv[u];
#pragma unroll
        for (int u = 0; u < kSpgPerLane; ++u) {   // bucket bounds of all the lane's members (overlapping reads)
            blo[u] = ok[u] ? start[bk[u]] : 0;
            bhi[u] = ok[u] ? start[bk[u] + 1] : 0;
        }
    } else {
        for (int x = tid; x < W2; x += kWalkThreads) cnt2[x] = 0u;
        __syncthreads();
        int32_t arr[kSpgPerLane];
#pragma unroll
        for (int u = 0; u < kSpgPerLane; ++u) {      // (bk[u] becomes the member's sub-bucket)
            arr[u] = 0;
            if (!ok[u]) continue;
            const uint32_t lo1 = (uint32_t)start[bk[u]], kb = (uint32_t)start[bk[u] + 1] - lo1;
            const uint32_t off = (uint32_t)(idv[u] - mn) - (bk[u] << bshift);                      // < 2^bshift
            const uint32_t sub = bshift ? __umulhi(off << (32 - bshift), kb) : 0u;                 // floor(off * kb / 2^bshift) < kb
            bk[u] = lo1 + sub;
            const uint32_t sh = (bk[u] & 1u) * 16u;
            arr[u] = (int32_t)((atomicAdd(&cnt2[bk[u] >> 1], 1u << sh) >> sh) & 0xFFFFu);
        }
        __syncthreads();
        {   // exclusive scan of the ns + 1 level-2 counters, in place (offsets <= ns < 2^16): CW consecutive words per lane
            uint32_t w[CW];
            int32_t s2 = 0;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int x = tid * CW + c;
                w[c] = x < W2 ? cnt2[x] : 0u;
                s2 += (int32_t)((w[c] & 0xFFFFu) + (w[c] >> 16));
            }
            int32_t inc = s2;
#pragma unroll
            for (int dd = 1; dd < kWave; dd <<= 1) {
                const int32_t t2 = __shfl_up(inc, dd, kWave);
                if ((tid & (kWave - 1)) >= dd) inc += t2;
            }
            if ((tid & (kWave - 1)) == kWave - 1) red[12 + tid / kWave] = inc;
            __syncthreads();
            int32_t run = inc - s2;
#pragma unroll
            for (int w2 = 0; w2 < kWalkThreads / kWave - 1; ++w2) run += w2 < tid / kWave ? red[12 + w2] : 0;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int x = tid * CW + c;
                const uint32_t lo16 = (uint32_t)run;
                run += (int32_t)(w[c] & 0xFFFFu);
                const uint32_t hi16 = (uint32_t)run;
                run += (int32_t)(w[c] >> 16);
                if (x < W2) cnt2[x] = lo16 | (hi16 << 16);
            }
        }
        __syncthreads();
        const uint16_t *off2 = (const uint16_t *)cnt2;
#pragma unroll
        for (int u = 0; u < kSpgPerLane; ++u) {
            blo[u] = ok[u] ? off2[bk[u]] : 0;
            bhi[u] = ok[u] ? off2[bk[u] + 1] : 0;
            if (ok[u]) A[blo[u] + arr[u]] = ((unsigned long long)(uint32_t)idv[u] << 32) | (uint32_t)slv[u];
        }
    }
    __syncthreads();
    SG_HOOK_STAMP(7);
    // order inside a bucket = number of smaller ids in it -> final position in the row.  The sorted row is assembled
    // in LDS (ids over the dead minq table, slots behind A) and leaves with consecutive lanes on consecutive words.
    int32_t *fin_id = (int32_t *)minq;                 // [ns] <= T
    int32_t *fin_sl = (int32_t *)(pk + a.stride + 1);   // [ns]: A occupies pk[0..ns), ns <= stride; (T - stride - 1) * 8 >= 4 * stride
    const bool staged = (int64_t)(T - a.stride - 1) * 8 >= (int64_t)4 * a.stride;
    const uint32_t *Ahi = (const uint32_t *)A;
#pragma unroll
    for (int u = 0; u < kSpgPerLane; ++u)
        if (ok[u]) {
            const int lo = blo[u], hi = bhi[u];
            int rank = 0;       // ids are distinct within a set: the high word of A decides
            for (int t2 = lo; t2 < hi; ++t2) rank += (Ahi[2 * t2 + 1] < (uint32_t)idv[u]) ? 1 : 0;
            if (staged) {
                fin_id[lo + rank] = idv[u];
                fin_sl[lo + rank] = slv[u];
            } else {
                a.set_ids[obase + lo + rank] = idv[u];
                a.set_slot[obase + lo + rank] = slv[u];
            }
        }
    if (staged) {
        __syncthreads();
        for (int x = tid; x < ns; x += kWalkThreads) {
            a.set_ids[obase + x] = fin_id[x];
            a.set_slot[obase + x] = fin_sl[x];
        }
    }
    SG_HOOK_STAMP(8);
}

// ------------------------------------------------------------------------------- compaction
// One wave per root: copy its members from the strided staging area to the packed arrays (subg_acc.c:870-871).
// INSERT: the same pass also registers every member's LP key in the HBM table of distinct rows
// (subg_acc.c:957-978) -- the keys of one set are folded in a wave-private LDS table first (a set of ~400
// members carries a few dozen distinct rows), only those go to HBM, and out_slot[] receives the table slot of
// every member, so no later pass touches the 8-byte keys again.
constexpr int kCompactThreads = 256;
constexpr int kCompactWaves = kCompactThreads / kWave;
constexpr int kFoldSlots = 256;   // per-wave LDS table

template <bool INSERT>
__global__ __launch_bounds__(kCompactThreads) void compact_sets_kernel(
    const int32_t *__restrict__ set_ids, const uint64_t *__restrict__ set_keys, const int32_t *__restrict__ nsize,
    const int64_t *__restrict__ row_off, int64_t n, int32_t stride, int32_t *__restrict__ out_ids,
    uint64_t *__restrict__ out_keys, UniqTable t, int64_t tag_base, int32_t *__restrict__ out_slot, int32_t *flags) {
    __shared__ unsigned long long lk[INSERT ? kCompactWaves * kFoldSlots : 1];
    __shared__ unsigned long long lt[INSERT ? kCompactWaves * kFoldSlots : 1];
    __shared__ int32_t ls[INSERT ? kCompactWaves * kFoldSlots : 1];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * kCompactWaves + wave;
    const int ns = i < n ? nsize[i] : 0;
    const int64_t src = i * (int64_t)stride, dst = i < n ? row_off[i] : 0;
    if (!INSERT) {
        for (int r = lane; r < ns; r += kWave) {
            out_ids[dst + r] = set_ids[src + r];
            out_keys[dst + r] = set_keys[src + r];
        }
        return;
    }
    unsigned long long *wk = lk + wave * kFoldSlots, *wt = lt + wave * kFoldSlots;
    int32_t *ws = ls + wave * kFoldSlots;
    for (int s = lane; s < kFoldSlots; s += kWave) {
        wk[s] = kEmptyKey;
        wt[s] = ~0ull;
    }
    __syncthreads();
    for (int r = lane; r < ns; r += kWave) {   // fold the set's keys: key -> min position
        const unsigned long long key = set_keys[src + r];
        const unsigned long long tag = (unsigned long long)(tag_base + dst + r);
        uint32_t h = (uint32_t)(mix64(key) >> 40) & (kFoldSlots - 1);
        bool done = false;
        for (int p = 0; p < 16; ++p) {
            unsigned long long cur = wk[h];
            if (cur == kEmptyKey) cur = atomicCAS(&wk[h], kEmptyKey, key);
            if (cur == kEmptyKey || cur == key) {
                if (wt[h] > tag) atomicMin(&wt[h], tag);
                done = true;
                break;
            }
            h = (h + 1) & (kFoldSlots - 1);
        }
        if (!done) uniq_global_insert(t, key, tag, flags);   // crowded: straight to HBM, found again below
    }
    __syncthreads();
    for (int s = lane; s < kFoldSlots; s += kWave)
        if (wk[s] != kEmptyKey) ws[s] = uniq_global_insert(t, wk[s], wt[s], flags);
    __syncthreads();
    for (int r = lane; r < ns; r += kWave) {
        const unsigned long long key = set_keys[src + r];   // second read: L2 hit
        out_ids[dst + r] = set_ids[src + r];
        if (out_keys) out_keys[dst + r] = key;
        uint32_t h = (uint32_t)(mix64(key) >> 40) & (kFoldSlots - 1);
        int32_t slot = -1;
        for (int p = 0; p < 16; ++p) {
            if (wk[h] == key) {
                slot = ws[h];
                break;
            }
            h = (h + 1) & (kFoldSlots - 1);
        }
        if (slot < 0) {   // the key went straight to HBM (or the table is over-full: flags[2] is set, caller retries)
            uint64_t g = mix64(key) & t.mask;
            for (uint64_t probes = 0; probes < kMaxProbes && t.keys[g] != key; ++probes) g = (g + 1) & t.mask;
            slot = (int32_t)g;
        }
        out_slot[dst + r] = slot;
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" int subgacc_key_shift(int32_t num_walks, int32_t num_steps) {
    SG_REQUIRE(num_walks > 0 && num_steps > 0, SUBGACC_ERR_BADARG, "num_walks and num_steps must be positive");
    const int shift = 32 - __builtin_clz((unsigned)num_walks);
    SG_REQUIRE((int64_t)num_steps * shift + 1 <= 64, SUBGACC_ERR_KEYWIDTH,
               "Longer width of type for hasing key needed > INT64.");
    // an all-ones key would collide with the empty marker of the unique table
    SG_REQUIRE(!((int64_t)num_steps * shift + 1 == 64 && num_walks == (1 << shift) - 1), SUBGACC_ERR_KEYWIDTH,
               "key space exhausted: num_steps*SHIFT+1 == 64 with num_walks == 2^SHIFT-1");
    return shift;
}

extern "C" size_t subgacc_rng_positions_workspace_bytes(int64_t n) {
    if (n < 0) n = 0;
    return align_up((size_t)n * 4, 256) + align_up((size_t)(n + 1) * 8, 256) + scan_workspace_bytes(n);
}

extern "C" int subgacc_rng_positions(const subgacc_walk_cfg *cfg, const void *indptr, int64_t num_nodes,
                                     const int32_t *query, int64_t n, int32_t rng_streams, uint64_t calls_before, uint32_t *rng_pos,
                                     uint32_t *rng_seed, void *workspace, size_t workspace_bytes, void *stream) {
    SG_REQUIRE(cfg && indptr && rng_pos && rng_seed && n >= 0 && num_nodes >= 0, SUBGACC_ERR_BADARG,
               "rng_positions: null argument");
    SG_REQUIRE(rng_streams >= 1, SUBGACC_ERR_BADARG, "rng_positions: rng_streams must be >= 1");
    SG_REQUIRE(rng_streams == 1 || calls_before == 0, SUBGACC_ERR_BADARG,
               "rng_positions: calls_before only makes sense for a single stream");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(query, SUBGACC_ERR_BADARG, "rng_positions: null query");
    SG_REQUIRE(workspace && workspace_bytes >= subgacc_rng_positions_workspace_bytes(n), SUBGACC_ERR_WORKSPACE,
               "rng_positions: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int32_t *calls = (int32_t *)ws;
    ws += align_up((size_t)n * 4, 256);
    int64_t *excl = (int64_t *)ws;
    ws += align_up((size_t)(n + 1) * 8, 256);
    const unsigned grid = (unsigned)ceil_div(n, 256);
    if (cfg->indptr64)
        hipLaunchKernelGGL(rng_calls_kernel<true>, dim3(grid), dim3(256), 0, s, indptr, query, n, num_nodes, cfg->num_walks,
                           cfg->num_steps, cfg->first_hop_wo, cfg->cap_root_degree, calls);
    else
        hipLaunchKernelGGL(rng_calls_kernel<false>, dim3(grid), dim3(256), 0, s, indptr, query, n, num_nodes, cfg->num_walks,
                           cfg->num_steps, cfg->first_hop_wo, cfg->cap_root_degree, calls);
    SG_LAUNCH_CHECK();
    int rc = exclusive_scan_i32(calls, n, excl, ws, workspace_bytes - (size_t)(ws - (char *)workspace), s);
    if (rc != SUBGACC_OK) return rc;
    hipLaunchKernelGGL(rng_positions_kernel, dim3(grid), dim3(256), 0, s, (const int64_t *)excl, n, rng_streams,
                       calls_before, cfg->seed, rng_pos, rng_seed);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

template <bool IDX64>
__global__ void hop_records_kernel(const void *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t num_nodes,
                                   int64_t nnz, RecFmt f, unsigned long long *__restrict__ recs) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int32_t v = indices[e];
    int64_t beg = 0, deg = 0;
    if ((uint64_t)(int64_t)v < (uint64_t)num_nodes) load_row<IDX64>(indptr, v, beg, deg);
    const int deg_bits = 64 - f.id_bits - f.beg_bits;
    const unsigned long long dmask = (1ull << deg_bits) - 1ull;
    const unsigned long long dv = (unsigned long long)deg >= dmask ? dmask : (unsigned long long)deg;   // all ones = escape
    recs[e] = ((unsigned long long)(uint32_t)v << (64 - f.id_bits)) | ((unsigned long long)beg << deg_bits) | dv;
}

// 16-byte form for graphs with int64 row offsets: {neighbour id : 32 | its degree : 32 (saturated), its row begin : 64}
template <bool IDX64>
__global__ void hop_records16_kernel(const void *__restrict__ indptr, const int32_t *__restrict__ indices, int64_t num_nodes,
                                     int64_t nnz, ulonglong2 *__restrict__ recs) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int32_t v = indices[e];
    int64_t beg = 0, deg = 0;
    if ((uint64_t)(int64_t)v < (uint64_t)num_nodes) load_row<IDX64>(indptr, v, beg, deg);
    recs[e] = make_ulonglong2(((unsigned long long)(deg > 0xFFFFFFFFll ? 0xFFFFFFFFull : (unsigned long long)deg) << 32) | (uint32_t)v,
                              (unsigned long long)beg);
}

extern "C" int subgacc_hop_records_format(int64_t num_nodes, int64_t nnz, int32_t *id_bits, int32_t *beg_bits) {
    SG_REQUIRE(num_nodes >= 0 && nnz >= 0 && id_bits && beg_bits, SUBGACC_ERR_BADARG, "hop_records_format: bad arguments");
    int ib = 1, bb = 1;
    while (ib < 32 && ((int64_t)1 << ib) < num_nodes) ++ib;
    while (bb < 40 && ((int64_t)1 << bb) <= nnz) ++bb;
    *id_bits = ib, *beg_bits = bb;
    const int deg_bits = 64 - ib - bb;
    SG_REQUIRE(deg_bits >= 8, SUBGACC_ERR_BADARG, "hop_records_format: only %d bits left for the degree", deg_bits);
    return deg_bits;
}

extern "C" int subgacc_hop_records_build(const void *indptr, int32_t indptr64, const int32_t *indices, int64_t num_nodes,
                                         int64_t nnz, int32_t id_bits, int32_t beg_bits, uint64_t *recs, void *stream) {
    SG_REQUIRE(indptr && num_nodes >= 0 && nnz >= 0, SUBGACC_ERR_BADARG, "hop_records_build: bad arguments");
    if (id_bits == 0 && beg_bits == 0) {     // the 16-byte form (2 uint64 per entry)
        if (nnz == 0) return SUBGACC_OK;
        SG_REQUIRE(indices && recs, SUBGACC_ERR_BADARG, "hop_records_build: null argument");
        const int64_t blocks16 = ceil_div(nnz, 256);
        SG_REQUIRE(blocks16 < (1ll << 31), SUBGACC_ERR_BADARG, "hop_records_build: graph too large for one launch");
        if (indptr64)
            hipLaunchKernelGGL(hop_records16_kernel<true>, dim3((unsigned)blocks16), dim3(256), 0, (hipStream_t)stream, indptr,
                               indices, num_nodes, nnz, (ulonglong2 *)recs);
        else
            hipLaunchKernelGGL(hop_records16_kernel<false>, dim3((unsigned)blocks16), dim3(256), 0, (hipStream_t)stream, indptr,
                               indices, num_nodes, nnz, (ulonglong2 *)recs);
        SG_LAUNCH_CHECK();
        return SUBGACC_OK;
    }
    SG_REQUIRE(id_bits >= 1 && beg_bits >= 1 && id_bits + beg_bits <= 56, SUBGACC_ERR_BADARG, "hop_records_build: bad field widths");
    SG_REQUIRE(num_nodes <= ((int64_t)1 << id_bits) && nnz < ((int64_t)1 << beg_bits), SUBGACC_ERR_BADARG,
               "hop_records_build: the graph does not fit id_bits = %d / beg_bits = %d", id_bits, beg_bits);
    if (nnz == 0) return SUBGACC_OK;
    SG_REQUIRE(indices && recs, SUBGACC_ERR_BADARG, "hop_records_build: null argument");
    const RecFmt f{id_bits, beg_bits};
    const int64_t blocks = ceil_div(nnz, 256);
    SG_REQUIRE(blocks < (1ll << 31), SUBGACC_ERR_BADARG, "hop_records_build: graph too large for one launch");
    if (indptr64)
        hipLaunchKernelGGL(hop_records_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, indptr, indices,
                           num_nodes, nnz, f, (unsigned long long *)recs);
    else
        hipLaunchKernelGGL(hop_records_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, indptr, indices,
                           num_nodes, nnz, f, (unsigned long long *)recs);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

static int launch_walk(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                       const int32_t *query, int64_t n, const uint32_t *rng_pos, const uint32_t *rng_seed,
                       int32_t *set_ids, uint64_t *set_keys, int32_t *set_slot, void *uniq_table, int64_t uniq_capacity,
                       int64_t root_base, int32_t *nsize, int32_t *walks, int32_t *flags, void *stream,
                       bool holes = false, const int32_t *worklist = nullptr, const int64_t *n_work = nullptr,
                       bool tags_only = false, int64_t work_cap = 0, bool order_only_list = false) {
    const bool spg = set_slot != nullptr;
    SG_REQUIRE(cfg && indptr && set_ids && (set_keys || spg) && nsize && flags, SUBGACC_ERR_BADARG,
               "walk: null argument");
    SG_REQUIRE(n >= 0 && num_nodes >= 0, SUBGACC_ERR_BADARG, "walk: negative size");
    SG_REQUIRE(cfg->rng_mode == SUBGACC_RNG_RAND_R || cfg->rng_mode == SUBGACC_RNG_PHILOX, SUBGACC_ERR_BADARG,
               "walk: unknown rng_mode %d", cfg->rng_mode);
    const int shift = subgacc_key_shift(cfg->num_walks, cfg->num_steps);
    if (shift < 0) return shift;
    const int M = cfg->num_walks, m = cfg->num_steps;
    SG_REQUIRE((int64_t)M * m + 1 <= (1 << 20), SUBGACC_ERR_LDS, "walk: M*m+1 = %lld too large", (long long)M * m + 1);
    const int Q = M * m + 1;
    const int stride = cfg->bucket > 0 ? cfg->bucket : Q;
    SG_REQUIRE(!cfg->emit_walks || walks, SUBGACC_ERR_BADARG, "walk: emit_walks without a walks buffer");
    SG_REQUIRE(cfg->rng_mode != SUBGACC_RNG_RAND_R || (rng_pos && rng_seed) || n == 0, SUBGACC_ERR_BADARG,
               "walk: RAND_R mode needs rng_pos/rng_seed from subgacc_rng_positions");
    if (spg) {
        SG_REQUIRE(table_size_for(Q) <= kSpgPerLane * kWalkThreads, SUBGACC_ERR_LDS,
                   "walk_spg: M*m+1 = %d needs a per-root table above %d slots; use subgacc_walk_sets + subgacc_spg_build",
                   Q, kSpgPerLane * kWalkThreads);
        SG_REQUIRE(!uniq_table || (uniq_capacity > 0 && (uniq_capacity & (uniq_capacity - 1)) == 0 &&
                                   uniq_capacity < (1ll << 31) && root_base >= 0),
                   SUBGACC_ERR_BADARG, "walk_spg: needs a power-of-two table of distinct rows (or none: key rows)");
    }
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(query, SUBGACC_ERR_BADARG, "walk: null query");   // `indices` may be NULL for an edgeless graph
    SG_REQUIRE(num_nodes >= 1, SUBGACC_ERR_BADARG, "walk: %lld roots but a graph without nodes", (long long)n);

    WalkArgs a;
    a.indptr = indptr, a.indices = indices, a.query = query, a.n = n, a.num_nodes = num_nodes;
    a.worklist = worklist, a.n_work = n_work;
    a.work_cap = work_cap, a.tags_only = tags_only ? 1 : 0;
    a.walk_pos = cfg->rng_mode == SUBGACC_RNG_RAND_R ? cfg->walk_pos : nullptr;
    a.rng_pos = rng_pos, a.rng_seed = rng_seed;
    a.set_ids = set_ids, a.set_keys = set_keys, a.nsize = nsize;
    a.walks = cfg->emit_walks ? walks : nullptr;
    a.flags = flags;
    a.M = M, a.m = m, a.stride = stride, a.shift = shift;
    // rows may lie further apart than they are long (row_pitch: every row on a 128-byte line); tags count in the same unit
    SG_REQUIRE(cfg->row_pitch == 0 || cfg->row_pitch >= stride, SUBGACC_ERR_BADARG, "walk: row_pitch %d < the row capacity %d",
               cfg->row_pitch, stride);
    a.pitch = cfg->row_pitch > 0 ? cfg->row_pitch : stride;
    a.T = table_size_for(Q);
    a.tshift = 32 - (31 - __builtin_clz((unsigned)a.T));
    a.nwords = (Q + 31) / 32;
    a.seed = cfg->seed;
    a.wo = cfg->first_hop_wo ? 1 : 0;
    a.step_major = cfg->order == SUBGACC_ORDER_STEP_MAJOR ? 1 : 0;
    a.cap_root = cfg->cap_root_degree ? 1 : 0;
    a.set_slot = set_slot;
    // key rows leave four members per store where every row begins on a 16-byte boundary (walk_rows_kernel)
    a.wide_rows = (a.pitch % 4 == 0 && (((uintptr_t)set_ids | (uintptr_t)set_slot) & 15u) == 0) ? 1 : 0;
    a.table = (spg && uniq_table) ? uniq_view(uniq_table, uniq_capacity) : UniqTable{nullptr, nullptr, nullptr, 0};
    a.keyrows = (spg && !uniq_table) ? 1 : 0;
    a.root_base = root_base;
    a.recs = (const unsigned long long *)cfg->hop_records;
    a.rec.id_bits = cfg->rec_id_bits, a.rec.beg_bits = cfg->rec_beg_bits;
    SG_REQUIRE(!a.recs || (a.rec.id_bits == 0 && a.rec.beg_bits == 0) ||
                   (a.rec.id_bits >= 1 && a.rec.beg_bits >= 1 && a.rec.id_bits + a.rec.beg_bits <= 56),
               SUBGACC_ERR_BADARG, "walk: hop records with field widths %d / %d", a.rec.id_bits, a.rec.beg_bits);
    if (a.recs && (cfg->indptr64 != 0) != (a.rec.id_bits == 0)) a.recs = nullptr;   // 8-byte form <-> int32 offsets, 16-byte <-> int64
    SG_REQUIRE(a.T <= 65536, SUBGACC_ERR_LDS, "walk: M*m+1 = %d is too large for the per-root LDS tables", Q);
#ifndef SG_DEV_LDS_PAD       // dev builds only: extra dynamic LDS per workgroup, i.e. fewer resident workgroups per CU -- the
#define SG_DEV_LDS_PAD 0     // occupancy response of the kernel (tools/README.md)
#endif
    const size_t lds = walk_lds_bytes(a.T, a.nwords, M, Q, spg, stride < Q) + (size_t)SG_DEV_LDS_PAD;
    SG_REQUIRE(lds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS,
               "walk: per-root tables need %zu B of LDS (> %d): M*m+1 = %d is too large", lds, kLdsBytes, Q);

    hipStream_t s = (hipStream_t)stream;
    // the persistent, software-pipelined form takes every launch it supports (dev builds: -DSG_DEV_NO_WALK_PIPE forces this file's)
#ifdef SG_DEV_NO_WALK_PIPE
    const bool use_pipe = false;
#else
    const bool use_pipe = true;
#endif
    // (replayed stream positions, walk_pos: only the general kernel below reads them)
    if (use_pipe && !holes && !a.walk_pos && launch_walk_pipe(a, cfg->indptr64 != 0, cfg->rng_mode, spg, lds, s)) {
        SG_LAUNCH_CHECK();
        return SUBGACC_OK;
    }
    if (spg && lds <= 64 * 1024 && !a.walk_pos && launch_walk_rows(a, cfg->indptr64 != 0, cfg->rng_mode, lds, s)) {
        SG_LAUNCH_CHECK();
        return SUBGACC_OK;
    }
    SG_REQUIRE(!tags_only, SUBGACC_ERR_BADARG,
               "walk_tags: only the fused-row kernel registers tags alone (2..4 hops, M <= 256, a 512- or 1,024-slot table, no "
               "bucket); M = %d, m = %d", M, m);
    if (holes && order_only_list) {      // a list that names EVERY row only says in which order to take them (subgacc_walk_spg_list): the
        holes = false;                   // general kernel, which reads no list, takes the rows in batch order -- the same rows
        a.worklist = nullptr, a.n_work = nullptr;
    }
    SG_REQUIRE(!holes, SUBGACC_ERR_BADARG,
               "walk_spg_sparse: rows without a root are passed over by the fused-row kernel only (2..4 hops, M <= 256, a 512- or "
               "1,024-slot table, no bucket); M = %d, m = %d", M, m);
    SG_REQUIRE(!a.keyrows, SUBGACC_ERR_BADARG,
               "walk_spg: key rows (no table of distinct rows) need set_sampler order, no bucket, M <= 256, 2 to 4 hops, a 512- or "
               "1,024-slot table and num_steps*SHIFT+1 <= 31 (subgacc_walk_keyrows64: 4 hops, <= 63 bits); M = %d, m = %d", M, m);
    const int64_t grid = xcd_grid(n);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "walk: chunk of %lld roots too large, split it", (long long)n);
#define SG_WALK_LAUNCH(I64, RNGM, SPGM)                                                                           \
    do {                                                                                                          \
        if (lds > 64 * 1024)                                                                                      \
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)walk_sets_kernel<I64, RNGM, SPGM>,                     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
        hipLaunchKernelGGL((walk_sets_kernel<I64, RNGM, SPGM>), dim3((unsigned)grid), dim3(kWalkThreads), lds, s, \
                           a);                                                                                    \
    } while (0)
#define SG_WALK_RNG(I64, SPGM)                                                                                    \
    do {                                                                                                          \
        if (cfg->rng_mode == SUBGACC_RNG_RAND_R) SG_WALK_LAUNCH(I64, SUBGACC_RNG_RAND_R, SPGM);                   \
        else SG_WALK_LAUNCH(I64, SUBGACC_RNG_PHILOX, SPGM);                                                       \
    } while (0)
    if (cfg->indptr64) {
        if (spg) SG_WALK_RNG(true, true);
        else SG_WALK_RNG(true, false);
    } else {
        if (spg) SG_WALK_RNG(false, true);
        else SG_WALK_RNG(false, false);
    }
#undef SG_WALK_RNG
#undef SG_WALK_LAUNCH
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_walk_sets(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices,
                                 int64_t num_nodes, const int32_t *query, int64_t n, const uint32_t *rng_pos,
                                 const uint32_t *rng_seed, int32_t *set_ids, uint64_t *set_keys, int32_t *nsize,
                                 int32_t *walks, int32_t *flags, void *stream) {
    SG_REQUIRE(set_keys, SUBGACC_ERR_BADARG, "walk_sets: null set_keys");
    return launch_walk(cfg, indptr, indices, num_nodes, query, n, rng_pos, rng_seed, set_ids, set_keys, nullptr, nullptr,
                       0, 0, nsize, walks, flags, stream);
}

extern "C" int subgacc_walk_spg(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices,
                                int64_t num_nodes, const int32_t *query, int64_t n, int64_t root_base,
                                const uint32_t *rng_pos, const uint32_t *rng_seed, void *uniq_table,
                                int64_t uniq_capacity, int32_t *row_ids, int32_t *row_slot, int32_t *nsize,
                                int32_t *flags, void *stream) {
    SG_REQUIRE(row_slot, SUBGACC_ERR_BADARG, "walk_spg: null row_slot");
    SG_REQUIRE(cfg && !cfg->emit_walks && cfg->order == SUBGACC_ORDER_WALK_MAJOR, SUBGACC_ERR_BADARG,
               "walk_spg: set_sampler order only, no raw walks");
    return launch_walk(cfg, indptr, indices, num_nodes, query, n, rng_pos, rng_seed, row_ids, nullptr, row_slot,
                       uniq_table, uniq_capacity, root_base, nsize, nullptr, flags, stream);
}

extern "C" int subgacc_walk_spg_sparse(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                                       const int32_t *query, int64_t n, const int32_t *worklist, const int64_t *n_work,
                                       void *uniq_table, int64_t uniq_capacity, int32_t *row_ids, int32_t *row_slot, int32_t *nsize,
                                       int32_t *flags, void *stream) {
    SG_REQUIRE(row_slot && (worklist != nullptr) == (n_work != nullptr), SUBGACC_ERR_BADARG,
               "walk_spg_sparse: null row_slot, or a work list without its length (or the reverse)");
    SG_REQUIRE(cfg && !cfg->emit_walks && cfg->order == SUBGACC_ORDER_WALK_MAJOR && cfg->rng_mode == SUBGACC_RNG_PHILOX,
               SUBGACC_ERR_BADARG, "walk_spg_sparse: set_sampler order, Philox mode (a root's set must not depend on its place in the batch)");
    return launch_walk(cfg, indptr, indices, num_nodes, query, n, nullptr, nullptr, row_ids, nullptr, row_slot, uniq_table,
                       uniq_capacity, 0, nsize, nullptr, flags, stream, true, worklist, n_work);
}

extern "C" int subgacc_walk_spg_list(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                                     const int32_t *query, int64_t n, const uint32_t *rng_pos, const uint32_t *rng_seed,
                                     const int32_t *worklist, const int64_t *n_work, void *uniq_table, int64_t uniq_capacity,
                                     int32_t *row_ids, int32_t *row_slot, int32_t *nsize, int32_t *flags, void *stream) {
    SG_REQUIRE(row_slot && worklist && n_work, SUBGACC_ERR_BADARG, "walk_spg_list: null row_slot / work list / length");
    SG_REQUIRE(cfg && !cfg->emit_walks && cfg->order == SUBGACC_ORDER_WALK_MAJOR, SUBGACC_ERR_BADARG,
               "walk_spg_list: set_sampler order only, no raw walks");
    // (the list of this entry point is an ORDER over all n rows, not a selection: when the fused-row kernel declines the shape --
    //  a dev build without it, a predicate of the caller that drifted from launch_walk_rows' -- the general kernel takes the rows
    //  in batch order instead of refusing the call)
    return launch_walk(cfg, indptr, indices, num_nodes, query, n, rng_pos, rng_seed, row_ids, nullptr, row_slot, uniq_table,
                       uniq_capacity, 0, nsize, nullptr, flags, stream, true, worklist, n_work, false, 0, true);
}

extern "C" int subgacc_walk_keyrows64(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                                      const int32_t *query, int64_t n, const uint32_t *rng_pos, const uint32_t *rng_seed,
                                      const int32_t *worklist, const int64_t *n_work, int32_t *row_ids, uint64_t *row_keys,
                                      int32_t *nsize, int32_t *flags, void *stream) {
    SG_REQUIRE(row_keys && (worklist != nullptr) == (n_work != nullptr), SUBGACC_ERR_BADARG,
               "walk_keyrows64: null row_keys, or a work list without its length (or the reverse)");
    SG_REQUIRE(cfg && !cfg->emit_walks && cfg->order == SUBGACC_ORDER_WALK_MAJOR && cfg->bucket <= 0, SUBGACC_ERR_BADARG,
               "walk_keyrows64: set_sampler order only, no raw walks, no bucket");
    const int shift = subgacc_key_shift(cfg->num_walks, cfg->num_steps);
    if (shift < 0) return shift;
    SG_REQUIRE(cfg->num_steps * shift + 1 > 31, SUBGACC_ERR_BADARG,
               "walk_keyrows64: the keys of M = %d, m = %d fit 32 bits -- subgacc_walk_spg(uniq_table = NULL) writes those",
               cfg->num_walks, cfg->num_steps);
    // (row_slot only has to be non-null: rows of 64-bit keys leave through set_keys; launch_walk_rows takes the launch or nobody does)
    return launch_walk(cfg, indptr, indices, num_nodes, query, n, rng_pos, rng_seed, row_ids, row_keys, (int32_t *)row_keys, nullptr,
                       0, 0, nsize, nullptr, flags, stream, worklist != nullptr, worklist, n_work);
}

extern "C" int subgacc_walk_tags(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                                 const int32_t *query, int64_t n, int64_t root_base, const uint32_t *rng_pos, const uint32_t *rng_seed,
                                 const int32_t *worklist, const int64_t *n_work, int64_t work_cap, void *uniq_table,
                                 int64_t uniq_capacity, int32_t *flags, void *stream) {
    SG_REQUIRE(worklist && n_work && uniq_table && work_cap >= 0, SUBGACC_ERR_BADARG, "walk_tags: needs a work list, its length and a table");
    SG_REQUIRE(cfg && !cfg->emit_walks && cfg->order == SUBGACC_ORDER_WALK_MAJOR && cfg->bucket <= 0, SUBGACC_ERR_BADARG,
               "walk_tags: set_sampler order, no raw walks, no bucket");
    // no row is written in this mode: the kernel's row arguments only have to be non-null
    int32_t *none = (int32_t *)flags;
    return launch_walk(cfg, indptr, indices, num_nodes, query, n, rng_pos, rng_seed, none, nullptr, none, uniq_table, uniq_capacity,
                       root_base, none, nullptr, flags, stream, true, worklist, n_work, true, work_cap);
}

extern "C" int subgacc_compact_sets(const int32_t *set_ids, const uint64_t *set_keys, const int32_t *nsize,
                                    const int64_t *row_off, int64_t n, int32_t stride, int32_t *out_ids,
                                    uint64_t *out_keys, void *uniq_table, int64_t uniq_capacity, int64_t tag_base,
                                    int32_t *out_slot, int32_t *flags, void *stream) {
    SG_REQUIRE(n >= 0 && stride > 0, SUBGACC_ERR_BADARG, "compact_sets: bad sizes");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(set_ids && set_keys && nsize && row_off && out_ids, SUBGACC_ERR_BADARG, "compact_sets: null argument");
    const bool insert = uniq_table != nullptr;
    SG_REQUIRE(insert || out_keys, SUBGACC_ERR_BADARG, "compact_sets: nothing to produce (no out_keys, no table)");
    SG_REQUIRE(!insert || (out_slot && flags && tag_base >= 0 && uniq_capacity > 0 &&
                           (uniq_capacity & (uniq_capacity - 1)) == 0 && uniq_capacity < (1ll << 31)),
               SUBGACC_ERR_BADARG, "compact_sets: the fused insert needs out_slot, flags and a power-of-two table");
    const int64_t blocks = ceil_div(n, kCompactWaves);
    SG_REQUIRE(blocks < (1ll << 31), SUBGACC_ERR_BADARG, "compact_sets: too many roots in one call");
    if (insert)
        hipLaunchKernelGGL(compact_sets_kernel<true>, dim3((unsigned)blocks), dim3(kCompactThreads), 0,
                           (hipStream_t)stream, set_ids, set_keys, nsize, row_off, n, stride, out_ids, out_keys,
                           uniq_view(uniq_table, uniq_capacity), tag_base, out_slot, flags);
    else
        hipLaunchKernelGGL(compact_sets_kernel<false>, dim3((unsigned)blocks), dim3(kCompactThreads), 0,
                           (hipStream_t)stream, set_ids, set_keys, nsize, row_off, n, stride, out_ids, out_keys,
                           UniqTable{nullptr, nullptr, nullptr, 0}, (int64_t)0, (int32_t *)nullptr, (int32_t *)nullptr);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

// SPG mode: rows are already final (sorted ids + table slot); this is a plain strided -> packed copy
__global__ __launch_bounds__(kCompactThreads) void compact_rows_kernel(const int32_t *__restrict__ row_ids,
                                                                        const int32_t *__restrict__ row_slot,
                                                                        const int32_t *__restrict__ nsize,
                                                                        const int64_t *__restrict__ row_off, int64_t n,
                                                                        int32_t stride, int32_t *__restrict__ out_indices,
                                                                        int32_t *__restrict__ out_data,
                                                                        const int32_t *__restrict__ slot_id) {
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * kCompactWaves + wave;
    if (i >= n) return;
    const int ns = nsize[i];
    const int64_t src = i * (int64_t)stride, dst = row_off[i];
    for (int r = lane; r < ns; r += kWave) {
        out_indices[dst + r] = row_ids[src + r];
        const int32_t sl = row_slot[src + r];
        out_data[dst + r] = slot_id ? slot_id[sl] + 1 : sl;   // numbered table given: SFptr+1 at once, no translate pass
    }
}

extern "C" int subgacc_compact_rows(const int32_t *row_ids, const int32_t *row_slot, const int32_t *nsize,
                                    const int64_t *row_off, int64_t n, int32_t stride, int32_t *out_indices,
                                    int32_t *out_data, const void *uniq_table, int64_t uniq_capacity, void *stream) {
    const int32_t *slot_id = uniq_table ? (const int32_t *)((const char *)uniq_table + (size_t)uniq_capacity * 16) : nullptr;
    SG_REQUIRE(!uniq_table || uniq_capacity > 0, SUBGACC_ERR_BADARG, "compact_rows: table without capacity");
    SG_REQUIRE(n >= 0 && stride > 0, SUBGACC_ERR_BADARG, "compact_rows: bad sizes");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(row_ids && row_slot && nsize && row_off && out_indices && out_data, SUBGACC_ERR_BADARG,
               "compact_rows: null argument");
    const int64_t blocks = ceil_div(n, kCompactWaves);
    SG_REQUIRE(blocks < (1ll << 31), SUBGACC_ERR_BADARG, "compact_rows: too many rows in one call");
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)blocks), dim3(kCompactThreads), 0, (hipStream_t)stream,
                       row_ids, row_slot, nsize, row_off, n, stride, out_indices, out_data, slot_id);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
