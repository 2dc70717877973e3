// walk_rows.hip -- the fused-row walk kernel, specialised (gfx950).
//
// walk_sets_kernel<SPG> (walk.hip) serves every configuration with one body: hop count, visit order, first-hop rule,
// bucket truncation and table size are run-time values.  The SQ counters of round 2 (profiles/r02e_sq_cit2.csv) show what
// that costs: 4,800 vector instructions per root, the vector ALUs 85 % busy, 240-320 `v_readlane` of scalar-register
// spills (the kernel must stay under 80 SGPRs to keep 8 workgroups per CU) -- the kernel is bound by VALU issue as much
// as by its random line fetches.  This file is the same algorithm for the shape every reference configuration with
// walks of >= 3 hops has -- set_sampler order, first hop without replacement, no truncating bucket, M <= 256 walks, a
// per-root table of 512 or 1,024 slots -- with the hop count MH and the slots per lane SPL as template parameters:
//   * the walk is straight-line code (no step loop, no per-step mode tests), the later hops' Philox draws are issued
//     while the first hop's neighbour is in flight, and each visit's LDS atomics run under the NEXT hop's row load;
//   * a lane owns SPL consecutive table slots in the epilogue, read with 8/16-byte LDS loads;
//   * the bucket sort takes its in-bucket arrival order from the histogram's returning atomic (one atomic pass, not two);
//   * 32-bit compares / shifts wherever the 64-bit ones of the general kernel were only generality.
// Bit-identical rows, sizes and table tags (tests/test_gpu_parity.py runs both forms against the oracle).
#include <stdlib.h>

#include <type_traits>

#include "walk_common.hpp"
#include "waveops.hpp"

namespace subgacc {

// The value of a load that the program then only uses under a condition: without this the optimizer moves the LOAD under that
// condition too, and a load under a branch is waited for where it stands (see the walk section).  An empty asm that "uses" the
// register keeps the load unconditional; it costs nothing but is a point where the value must have arrived.
#define SG_KEEP_LOAD(v) asm volatile("" : "+v"(v))

// the key rows' final stores: written once, read by the join a kernel later -- SG_NT_ROWS=1 streams them past L2 (A/B: tools/ab.py)
#ifndef SG_NT_ROWS
#define SG_NT_ROWS 0
#endif
#if SG_NT_ROWS
#define SG_ROW_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define SG_ROW_STORE(p, v) (*(p) = (v))
#endif

// Scalar registers: gfx950 holds 800 per SIMD -- up to 100 per wave leave all 8 wave slots usable; the 80 of rounds 2-5 dated from a
// 10-wave device and made the allocator park uniform values in vector lanes (v_writelane / v_readlane) for nothing.
#ifndef SG_ROWS_SGPR
#define SG_ROWS_SGPR 102
#endif
#ifndef SG_SORT_AGAIN_ABOVE   // the two-wave key rows' level 3: from this many members in one level-2 part on
#define SG_SORT_AGAIN_ABOVE 16
#endif
#ifndef SG_SORT_RANK_HEAD   // members of a (sub-)bucket compared without a loop when the sorted position is counted (two-wave key rows)
#define SG_SORT_RANK_HEAD 4
#endif
#ifndef SG_SORT_LEVELS      // finer levels of the key rows' sort behind level 1 (dev builds: 0 / 1 for timing; the result is sorted either way)
#define SG_SORT_LEVELS 2
#endif
#ifndef SG_SORT_ZERO_UPFRONT
#define SG_SORT_ZERO_UPFRONT 0
#endif
#ifndef SG_SORT_RANK_UNROLL
#define SG_SORT_RANK_UNROLL 1
#endif
#ifndef SG_LAST_HOP_ID     // 1: with hop records, the last hop reads the bare id from `indices` (A/B: tools/ab.py)
#define SG_LAST_HOP_ID 1
#endif

// NT lanes per workgroup (256, or 128 with two walks per lane: twice the roots per CU where the 512-slot table leaves the
// LDS for them), SPL table slots per lane, T = NT * SPL.
// REC: 0 = plain CSR; 8 = packed 8-byte hop records (int32 row offsets); 16 = 16-byte records (int64 row offsets)
// K32: the packed landing counts (and the LP keys made of them) fit 32 bits (m*SHIFT+1 <= 31: every reference
// configuration up to 3 hops) -- 12 bytes of LDS per table slot instead of 16, which is what lets the 1,024-slot table of the
// 3-hop configurations run with 128 lanes x 8 slots per lane and 11 roots per CU instead of 8.
// KR ("key rows"): the row's payload is the member's LP key itself -- 32 bits with K32 (a.set_slot), else 64 bits (a.set_keys: the
// 4-hop configurations with M >= 128, 4 x 8 + 1 = 33 bits; round 4).  A batch that is sampled, joined and
// dropped needs neither the table of distinct LP rows nor their numbering -- the join turns a key into its feature row
// arithmetically -- so the whole fold / registration / flush stage (a quarter of the kernel's vector instructions, its
// global atomics) and the first-visit bookkeeping (minq: one LDS atomic per visit, 4 bytes of LDS per slot) fall away.
// EPLP (key rows): members per lane the epilogue's sort provides registers for, ns <= EPLP * NT (0: SPL, or 7 for one wave x 8 slots:
// a 512-slot table holds 408) -- every reference shape has M * m + 1 <= 640, five members per lane of 128: 69 VGPRs instead of 80 + scratch.
template <bool IDX64, int RNG, int MH, int SPL, int NT, int REC = 0, bool K32 = false, bool KR = false, int EPLP = 0>
// (waves per SIMD asked of the register allocator: the table form with 32-bit counts holds 12 bytes of LDS per slot + the fold table,
//  ~15 KB per workgroup of two waves = 5 waves per SIMD whatever the registers; 64-bit counts on 128 lanes, 10.5 KB: 7; the rest 8)
//  key rows with 32-bit counts: 6 asked for -- since round 6 the kernel needs 51 VGPRs (five members per lane in the sort, their state in
//  one word each, no loops over the wave index for the compiler to vectorise): all eight wave slots of a SIMD; how many roots a CU
//  then holds is the launch's choice, through the LDS it asks for: SG_ROWS_KR_LDS_MIN)
__global__ __launch_bounds__(NT, K32 ? (KR ? 6 : 5) : (KR ? 5 : (NT == 128 ? 7 : 8))) __attribute__((amdgpu_num_sgpr(SG_ROWS_SGPR))) void walk_rows_kernel(const WalkArgs a) {
    static_assert(!KR || SPL % 4 == 0, "key rows: 4-slot chunks");
    constexpr bool KR64 = KR && !K32;      // rows of 64-bit LP keys
    extern __shared__ __align__(16) unsigned char lds_raw[];
    using CntT = typename std::conditional<K32, uint32_t, unsigned long long>::type;
    static_assert(SPL % 4 == 0 || (SPL == 2 && !K32), "slot ownership: 4-slot chunks, or two slots with 64-bit counts");
    constexpr CntT kNoKey = (CntT)~(CntT)0;                       // empty marker of the fold table
    constexpr int T = SPL * NT;                                   // 512 or 1,024 slots
    constexpr int TSHIFT = T == 1024 ? 22 : 23;                   // 32 - log2(T)
    constexpr int WPL = kWalkThreads / NT;                        // walks per lane (M <= 256)
    constexpr uint32_t TMASK = (uint32_t)T - 1u;
    CntT *pk = (CntT *)lds_raw;                                   // [T] packed landing counts
    int32_t *keys = (int32_t *)(pk + T);                          // [T] node ids
    uint32_t *minq = (uint32_t *)(keys + T);                      // [T] first visit sequence number
    int32_t *sarr = KR ? (int32_t *)(keys + T) : (int32_t *)(minq + T);   // [M] Fisher-Yates draws (KR: there is no minq)
    CntT *fk = (CntT *)(((uintptr_t)(sarr + a.M) + 7) & ~(uintptr_t)7);   // [kSpgFold]
    uint32_t *ft = (uint32_t *)(fk + kSpgFold);                   // [kSpgFold] min visit number of the key inside the set
    int32_t *fs = (int32_t *)(ft + kSpgFold);                     // [kSpgFold] HBM table slot of the key
    // [16] (KR: no fold table either; behind the walk tables and behind everything the epilogue's sort lays over them)
    int32_t *red = KR ? (int32_t *)(lds_raw + kr_red_offset(T, a.M, a.stride, NT, KR64)) : fs + kSpgFold;

    int64_t i;
    if (a.worklist) {   // a dense list of the rows to sample whose length lives on the device: its first xcd_grid(length) blocks work
        int64_t ne = *a.n_work;
        if (ne > a.n) ne = a.n;
        if (ne > (int64_t)gridDim.x) {     // the caller promised a shorter list (work_cap): say so, stay inside the launch
            if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&a.flags[3], 32);
            ne = (int64_t)gridDim.x / kXcds * kXcds;
        }
        const int64_t ge = (ne + kXcds - 1) / kXcds * kXcds;
        if ((int64_t)blockIdx.x >= ge) return;
        const int64_t k = xcd_item(blockIdx.x, ge);
        if (k >= ne) return;
        // (round 6 measured a record per list entry -- row, root id, the root's row pointers in ONE 16-byte read instead of the three
        //  dependent ones worklist[k] -> query[i] -> indptr[root]: no difference on any workload, profiles/r51_workrec_ab.log; the
        //  hardware's other resident workgroups hide that chain)
        i = a.worklist[k];
    } else {
        i = xcd_item(blockIdx.x, gridDim.x);
        if (i >= a.n) return;
    }
    const int tid = threadIdx.x;
    const int M = a.M;
    const int32_t root = a.query[i];
    if (root == SUBGACC_NO_ROOT) {     // a repeated endpoint of the batch: its first occurrence carries the set (uniq.hip: step dedup)
        if (tid == 0 && !a.tags_only) a.nsize[i] = 0;
        return;
    }
    // while the root's two dependent loads (query -> row pointer) are in flight: clear what does not depend on them
    // Slot ownership (clears, and the epilogue's member fetch): a lane owns the 4-slot chunks g = c*NT + tid, c < SPL/4 --
    // consecutive lanes on consecutive 16-byte words, so the 16-byte LDS accesses below are dense (a lane-contiguous
    // layout, 32 bytes apart, cost 39 % of the LDS cycles in bank conflicts at SPL = 8).  SPL = 2: slots 2*tid, 2*tid+1.
    if (SPL % 4 == 0) {
#pragma unroll
        for (int c = 0; c < SPL / 4; ++c) {
            const int g = c * NT + tid;
            if (K32) {
                ((uint4 *)pk)[g] = make_uint4(0u, 0u, 0u, 0u);
            } else {
                ((uint4 *)pk)[2 * g] = make_uint4(0u, 0u, 0u, 0u);
                ((uint4 *)pk)[2 * g + 1] = make_uint4(0u, 0u, 0u, 0u);
            }
            if (!KR) ((uint4 *)minq)[g] = make_uint4(~0u, ~0u, ~0u, ~0u);
        }
    } else {
        ((uint4 *)pk)[tid] = make_uint4(0u, 0u, 0u, 0u);
        ((uint2 *)minq)[tid] = make_uint2(~0u, ~0u);
    }
    if (!KR && tid < kSpgFold) {
        fk[tid] = kNoKey;
        ft[tid] = 0xFFFFFFFFu;
    }
    if (tid < 16) red[tid] = tid < 4 ? 0x7FFFFFFF : 0;   // [0..3] min id per wave, [4..7] max id, [8] member count
    if ((uint64_t)(int64_t)root >= (uint64_t)a.num_nodes) {   // the reference would read out of bounds here
        if (tid == 0) {
            atomicOr(&a.flags[3], 16);
            if (!a.tags_only) a.nsize[i] = 0;
        }
        return;
    }
    int64_t rbeg, rdeg64;
    load_row<IDX64>(a.indptr, root, rbeg, rdeg64);
    // the last entry of the adjacency array (uniform: a scalar load): what a dead end's dropped load is clamped to, below
    const int64_t last = (IDX64 ? ((const int64_t *)a.indptr)[a.num_nodes] : (int64_t)((const int32_t *)a.indptr)[a.num_nodes]) - 1;
    {   // keys, with the root already in its slot as member 0 (its minq = 0 is stored after the barrier-free clear above:
        // same lane order is not guaranteed across waves, so the root's lane writes BOTH of its words here)
        const uint32_t hroot = ((uint32_t)root * 2654435761u) >> TSHIFT;
        if (SPL % 4 == 0) {
#pragma unroll
            for (int c = 0; c < SPL / 4; ++c) {
                const uint32_t x0 = 4u * (uint32_t)(c * NT + tid);
                ((int4 *)keys)[c * NT + tid] = make_int4(x0 == hroot ? root : -1, x0 + 1u == hroot ? root : -1,
                                                         x0 + 2u == hroot ? root : -1, x0 + 3u == hroot ? root : -1);
            }
            if (!KR && ((hroot >> 2) % (uint32_t)NT) == (uint32_t)tid) minq[hroot] = 0u;   // the lane that cleared this word, after its clear
        } else {
            const uint32_t x0 = 2u * (uint32_t)tid;
            ((int2 *)keys)[tid] = make_int2(x0 == hroot ? root : -1, x0 + 1u == hroot ? root : -1);
            if ((hroot >> 1) == (uint32_t)tid) minq[hroot] = 0u;
        }
    }
    if (a.cap_root && rdeg64 > kNeighCap) rdeg64 = kNeighCap;
    const int64_t obase = i * (int64_t)a.pitch;
    const CntT lead = (CntT)1 << (MH * a.shift);
    const unsigned long long tag0 = (unsigned long long)((a.root_base + i) * (int64_t)a.pitch);

    if (rdeg64 == 0) {  // isolated root: one member, every count = M (subg_acc.c:753-761); id = the root
        if (tid == 0) {
            unsigned long long k = (unsigned long long)lead;
            for (int s = 0; s < MH; ++s) k |= (unsigned long long)M << (s * a.shift);
            if (!KR && a.tags_only) {
                uniq_global_insert(a.table, k, tag0, a.flags);
            } else {
                a.set_ids[obase] = root;
                if (KR64) a.set_keys[obase] = k;
                else a.set_slot[obase] = KR ? (int32_t)k : uniq_global_insert(a.table, k, tag0, a.flags);
                a.nsize[i] = 1;
            }
        }
        return;
    }

    uint32_t rpos = 0, rseed = a.seed;
    if (RNG == SUBGACC_RNG_RAND_R) {
        rpos = a.rng_pos[i];
        rseed = a.rng_seed[i];
    }
    const bool shuffled = rdeg64 > M;
    const uint32_t rdeg = (uint32_t)rdeg64;
    // rand_r: the stream state at the root's first draw, once per workgroup (uniform operands: a scalar loop of up to 32
    // rounds); a lane then jumps the few hundred steps to its own walk (<= 12 rounds) instead of all ~2^28 of them
    const uint32_t xroot = RNG == SUBGACC_RNG_RAND_R ? lcg_jump(rseed, rpos) : 0u;
    uint32_t pickv[WPL];             // the first hop of this lane's walks: an index into the root's row
    uint32_t inv = 0u;               // deg <= M: ceil(2^16 / deg) -- floor(x * inv / 2^16) = floor(x / deg) for x < 256
    if (shuffled) {
        // First hop without replacement (subg_acc.c:769-775): M sequential swaps a[k] <-> a[s_k], s_k = k + draw % (deg - k), over
        // the identity; walk w takes a[w].  Followed backwards, a[w] is found by: p = s_w; then, repeatedly, the LARGEST j below
        // the point reached so far with s_j == p, p = j.  Doing that by scanning j = w-1 .. 0 costs M^2/2 LDS reads per root (the
        // roots that need it are the hubs, 4 % of a collab batch: 11 % of that kernel's vector instructions).  Here instead: the
        // draws are grouped by target in a scratch hash over the still empty count table (key = target, value = the largest
        // member), and the groups are peeled from the top -- round r finds every group's r-th largest member and tells it to the
        // (r-1)-th as its predecessor; groups have one to a handful of members.  Then pred[w] is the first step of the chain and
        // every later step is "the largest member of the group of target j" = what round 0 found, kept in L[j] (j < M).
        constexpr int HS = (int)(T * sizeof(CntT) / 8);                  // hash slots: >= 256 >= M
        constexpr int HSHIFT = HS == 256 ? 24 : (HS == 512 ? 23 : (HS == 1024 ? 22 : 21));
        static_assert(HS == 256 || HS == 512 || HS == 1024 || HS == 2048, "scratch hash over the count table");
        uint32_t *HK = (uint32_t *)pk, *HV = HK + HS;                    // target + 1 (0 = empty) | largest bidder + 1
        int32_t *L = sarr;                                               // [M] largest j with s_j == position, -1 = none
        uint32_t tj[WPL];
        int hs[WPL], stt[WPL], prd[WPL];                                 // stt: 0 bidding, 1 top of its group (waits for its predecessor), 2 done
#pragma unroll
        for (int kk = 0; kk < WPL; ++kk) {
            const int k = tid + kk * NT;
            stt[kk] = k < M ? 0 : 2;
            prd[kk] = -1;
            hs[kk] = 0;
            tj[kk] = 0u;
            if (k >= M) continue;
            uint32_t r;
            if (RNG == SUBGACC_RNG_RAND_R) {
                uint32_t x = lcg_jump(xroot, 3u * (uint32_t)k);      // a jump of < 2^10 steps from the root's state
                r = rand_r_next(x);
                tj[kk] = r % (rdeg - (uint32_t)k) + (uint32_t)k;
            } else {
                uint32_t o1;
                philox2x32_10((uint32_t)root, (uint32_t)k | kPhiloxShuffle, a.seed, r, o1);
                tj[kk] = philox_below(r, rdeg - (uint32_t)k) + (uint32_t)k;
            }
            L[k] = -1;
        }
        __syncthreads();            // the count table is clear
#pragma unroll
        for (int kk = 0; kk < WPL; ++kk) {
            if (stt[kk] == 2) continue;
            uint32_t h = (tj[kk] * 2654435761u) >> HSHIFT;
            while (true) {
                const uint32_t old = atomicCAS(&HK[h], 0u, tj[kk] + 1u);
                if (old == 0u || old == tj[kk] + 1u) break;
                h = (h + 1u) & (uint32_t)(HS - 1);
            }
            hs[kk] = (int)h;
        }
        for (int round = 0;; ++round) {
#pragma unroll
            for (int kk = 0; kk < WPL; ++kk)
                if (stt[kk] == 1) HV[hs[kk]] = 0u;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < WPL; ++kk)
                if (stt[kk] == 0) atomicMax(&HV[hs[kk]], (uint32_t)(tid + kk * NT) + 1u);
            __syncthreads();
            int more = 0;
#pragma unroll
            for (int kk = 0; kk < WPL; ++kk) {
                if (stt[kk] == 2) continue;
                const uint32_t v = HV[hs[kk]];
                if (stt[kk] == 1) {
                    prd[kk] = (int)v - 1;
                    stt[kk] = 2;
                } else {
                    more = 1;
                    if (v == (uint32_t)(tid + kk * NT) + 1u) {
                        stt[kk] = 1;
                        if (round == 0 && tj[kk] < (uint32_t)M) L[tj[kk]] = tid + kk * NT;
                    }
                }
            }
            if (!__syncthreads_or(more)) break;
        }
#pragma unroll
        for (int kk = 0; kk < WPL; ++kk) {
            int32_t j = prd[kk];
            pickv[kk] = tj[kk];
            if (j >= 0) {
                for (int32_t nx = L[j]; nx >= 0; nx = L[j]) j = nx;
                pickv[kk] = (uint32_t)j;
            }
        }
        __syncthreads();
        // the scratch hash goes back to being an empty count table (the same lanes on the same words as the clear above)
        if (SPL % 4 == 0) {
#pragma unroll
            for (int c = 0; c < SPL / 4; ++c) {
                const int g = c * NT + tid;
                if (K32) {
                    ((uint4 *)pk)[g] = make_uint4(0u, 0u, 0u, 0u);
                } else {
                    ((uint4 *)pk)[2 * g] = make_uint4(0u, 0u, 0u, 0u);
                    ((uint4 *)pk)[2 * g + 1] = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        } else {
            ((uint4 *)pk)[tid] = make_uint4(0u, 0u, 0u, 0u);
        }
    } else {
        // w % deg for w < 256 and deg <= M <= 256 without a division per walk: with inv = ceil(2^16 / deg), floor(w * inv / 2^16) is the
        // exact quotient (the error term w * (deg * inv - 2^16) stays below 2^16)
        inv = 65535u / rdeg + 1u;
#pragma unroll
        for (int kk = 0; kk < WPL; ++kk) {
            const uint32_t w = (uint32_t)(tid + kk * NT);
            pickv[kk] = w - ((w * inv) >> 16) * rdeg;
        }
    }
    __syncthreads();
    SG_HOOK_RSTAMP(0);

    // ------------------------------------------------------------------ the walk: WPL walks per lane, straight-line,
    // the walks of a lane interleaved hop by hop (their loads are independent and in flight together)
    {
        // No load of this section stands under a branch.  A lane's walks beyond M (M = 200 on 64 lanes x 4: the fourth walk of 56
        // lanes) are not skipped but walk root -> its first neighbour -> that node's first neighbour: the same address in every
        // such lane, one request.  Only their VISITS are masked.  With `if (!walk) continue` around the loads, the compiler brought
        // each load back through a select that waited for it on the spot (vmcnt is an in-order counter): the row and neighbour
        // loads of a lane's four walks went out one after the other, nine dependent round trips per wavefront instead of three,
        // and the 2-hop kernel was bound by that latency at 8 wavefronts per SIMD, not by its instructions.
        bool wk[WPL];
        int32_t cur[WPL];
        unsigned long long rec[WPL], rec2[WPL];         // REC: the record of the node the walk stands on (rec2: its row begin, 16-byte form)
        uint32_t dr[WPL][MH > 1 ? 2 * (MH / 2) : 2];   // draws of hops 2..MH (Philox: computed while the first load is in flight)
        uint32_t x[WPL];
#pragma unroll
        for (int k = 0; k < WPL; ++k) {
            const int w = tid + k * NT;
            wk[k] = w < M;
            cur[k] = root;
            rec[k] = rec2[k] = 0ull;
            x[k] = 0;
            const uint32_t pick = wk[k] ? pickv[k] : 0u;
            if (REC == 16) {
                const ulonglong2 r2 = ((const ulonglong2 *)a.recs)[rbeg + pick];
                rec[k] = r2.x, rec2[k] = r2.y;
            } else if (REC == 8) {
                rec[k] = a.recs[rbeg + pick];
            } else {
                cur[k] = SG_NEIGH_LOAD(&a.indices[rbeg + pick]);
            }
        }
#pragma unroll
        for (int k = 0; k < WPL; ++k) {
            const int w = tid + k * NT;
            if (RNG == SUBGACC_RNG_PHILOX) {
#pragma unroll
                for (int b = 0; 2 * b < MH - 1; ++b) {
                    philox2x32_10((uint32_t)root, (uint32_t)w | ((uint32_t)b << kPhiloxBlockShift), a.seed, dr[k][2 * b], dr[k][2 * b + 1]);
                    if (!wk[k]) dr[k][2 * b] = dr[k][2 * b + 1] = 0u;      // (a draw of 0 picks a row's first entry)
                }
            } else {
                x[k] = lcg_jump(xroot, 3u * ((shuffled ? (uint32_t)M : 0u) + (uint32_t)w * (uint32_t)(MH - 1)));
            }
        }
        // offset of the next hop inside a row of dk entries.  dk == 0 is a dead end, the walk stays: offset 0, and the load that
        // follows is made all the same and dropped -- its index is clamped to the array's last entry (a trailing empty row begins
        // at nnz) rather than selected, because a select on `live` in front of the load is turned back into a branch around it
        auto next_off = [&](int k, int s, uint32_t dk, bool live) -> uint32_t {
            if (RNG == SUBGACC_RNG_RAND_R) {
                uint32_t xn = x[k];
                const uint32_t r = rand_r_next(xn);
                if (live) x[k] = xn;
                if (!live && wk[k]) atomicOr(&a.flags[0], 1);      // dead end: the sequential stream is no longer reproducible
                return wk[k] ? r % (live ? dk : 1u) : 0u;
            }
            return philox_below(dr[k][s], dk);
        };
#pragma unroll
        for (int s = 0; s < MH; ++s) {
            int64_t b[WPL], d[WPL];
            unsigned long long nr[WPL], nr2[WPL];       // REC: what this hop's loads bring (taken over after the visits, below)
#pragma unroll
            for (int k = 0; k < WPL; ++k) {
                b[k] = d[k] = 0;
                nr[k] = nr2[k] = 0ull;
                if (REC) {      // the record carries the node AND its row: the next record is asked for before the visit
                    bool esc = false;
                    if (SG_LAST_HOP_ID && s + 1 == MH && MH > 1) {
                        cur[k] = (int32_t)(uint32_t)rec[k];       // the last hop fetched the bare neighbour id (below)
                    } else if (REC == 16) {
                        cur[k] = (int32_t)(uint32_t)rec[k];
                        d[k] = (int64_t)(rec[k] >> 32);
                        b[k] = (int64_t)rec2[k];
                    } else {
                        rec_unpack(rec[k], a.rec, cur[k], b[k], d[k], esc);
                    }
                    if (s + 1 < MH) {
                        if (esc) load_row<IDX64>(a.indptr, cur[k], b[k], d[k]);       // (a row too long for the record: rare)
                        const bool live = d[k] > 0;
                        int64_t at = min(b[k] + (int64_t)next_off(k, s, (uint32_t)d[k], live), last);
                        SG_HOOK_HOP_AT(at, s + 2 == MH);
                        if (SG_LAST_HOP_ID && s + 2 == MH) {
                            // the walk ends on this node: its row is never needed, so the 4-byte id from the plain
                            // adjacency array will do -- half the bytes per entry, twice the entries of a hub's row
                            // per line for the walkers that share it
                            nr[k] = (unsigned long long)(uint32_t)SG_NEIGH_LOAD(&a.indices[at]);
                        } else if (REC == 16) {
                            const ulonglong2 r2 = ((const ulonglong2 *)a.recs)[at];
                            nr[k] = r2.x, nr2[k] = r2.y;
                        } else {
                            nr[k] = a.recs[at];
                        }
                    }
                } else if (s + 1 < MH) {
                    load_row<IDX64>(a.indptr, cur[k], b[k], d[k]);   // the next hop's row: in flight under the visit
                }
            }
#pragma unroll
            for (int k = 0; k < WPL; ++k) {
                if (!wk[k]) continue;
                // ---- visit: insert-or-find, first-visit sequence number, landing count
                CntT inc = (CntT)1 << ((MH - 1 - s) * a.shift);
                if (s == 0 && !shuffled) {
                    // deg <= M: walk w takes neighbour w % deg, so neighbour j is landed on by the walks j, j + deg, ... -- exactly
                    // floor((M-1-j) / deg) + 1 of them.  Walk j makes that visit for all of them, the others make none: deg LDS
                    // atomics instead of M, and none of the same-address pile-ups (a root of degree 7 had 28 lanes on one slot)
                    const uint32_t w = (uint32_t)(tid + k * NT);
                    if (w >= rdeg) continue;
                    inc *= (CntT)(((((uint32_t)M - 1u - w) * inv) >> 16) + 1u);
                }
                uint32_t h = ((uint32_t)cur[k] * 2654435761u) >> TSHIFT;
                while (true) {
                    const int32_t old = atomicCAS(&keys[h], -1, cur[k]);
                    if (old == -1 || old == cur[k]) break;
                    h = (h + 1u) & TMASK;
                }
                if (!KR) atomicMin(&minq[h], (uint32_t)((tid + k * NT) * MH + s + 1));
                atomicAdd(&pk[h], inc);
            }
            if (REC && s + 1 < MH) {
                // (the loads' values are taken over here, after every walk of the lane has issued its own and made its visit)
#pragma unroll
                for (int k = 0; k < WPL; ++k) {
                    const bool live = d[k] > 0;
                    SG_KEEP_LOAD(nr[k]);
                    if (SG_LAST_HOP_ID && s + 2 == MH) {
                        rec[k] = live ? nr[k] : (unsigned long long)(uint32_t)cur[k];      // dead end: the walk stays on this node
                    } else {
                        if (REC == 16) {
                            SG_KEEP_LOAD(nr2[k]);
                            rec2[k] = live ? nr2[k] : rec2[k];
                        }
                        rec[k] = live ? nr[k] : rec[k];
                    }
                }
            }
            if (!REC && s + 1 < MH) {
                int32_t nv[WPL];
#pragma unroll
                for (int k = 0; k < WPL; ++k) {
                    const int64_t at = min(b[k] + (int64_t)next_off(k, s, (uint32_t)d[k], d[k] > 0), last);
                    nv[k] = SG_NEIGH_LOAD(&a.indices[at]);
                }
#pragma unroll
                for (int k = 0; k < WPL; ++k) {
                    SG_KEEP_LOAD(nv[k]);
                    cur[k] = d[k] > 0 ? nv[k] : cur[k];
                }
            }
        }
    }
    __syncthreads();
    SG_HOOK_RSTAMP(2);

    if (KR) {
        // ================= key rows: the set leaves sorted by node id with its members' LP keys =================
        // A set fills a fifth to a third of the table: the members are first packed (any order -- the sort below fixes it),
        // so that every later phase runs over ceil(ns / NT) elements per lane instead of SPL mostly empty slots.
        int32_t idv[SPL];
        CntT kv[SPL];
        int cnt = 0;
        uint32_t umn = ~0u;         // an empty slot is -1: the largest unsigned value, the smallest signed one
        int32_t vmax = -1;
#pragma unroll
        for (int c = 0; c < SPL / 4; ++c) {
            const int g = c * NT + tid;
            const int4 kk = ((const int4 *)keys)[g];
            idv[4 * c] = kk.x, idv[4 * c + 1] = kk.y, idv[4 * c + 2] = kk.z, idv[4 * c + 3] = kk.w;
            if (K32) {
                const uint4 v = ((const uint4 *)pk)[g];
                kv[4 * c] = (CntT)v.x, kv[4 * c + 1] = (CntT)v.y, kv[4 * c + 2] = (CntT)v.z, kv[4 * c + 3] = (CntT)v.w;
            } else {
                const uint4 v0 = ((const uint4 *)pk)[2 * g], v1 = ((const uint4 *)pk)[2 * g + 1];
                kv[4 * c] = (CntT)(((unsigned long long)v0.y << 32) | v0.x);
                kv[4 * c + 1] = (CntT)(((unsigned long long)v0.w << 32) | v0.z);
                kv[4 * c + 2] = (CntT)(((unsigned long long)v1.y << 32) | v1.x);
                kv[4 * c + 3] = (CntT)(((unsigned long long)v1.w << 32) | v1.z);
            }
        }
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            cnt += idv[u] != -1 ? 1 : 0;
            umn = min(umn, (uint32_t)idv[u]);
            vmax = max(vmax, idv[u]);
        }
        const int incl = wave_scan_add_i32_incl(cnt);
        umn = wave_red_min_u32(umn);
        vmax = wave_red_max_i32(vmax);
        int wbase = 0;
        if (NT > kWave) {
            if ((tid & (kWave - 1)) == kWave - 1) {
                wbase = atomicAdd(&red[8], incl);
                red[tid / kWave] = (int32_t)min(umn, 0x7FFFFFFFu);
                red[4 + tid / kWave] = vmax;
            }
            wbase = __builtin_amdgcn_readlane(wbase, kWave - 1);
        }
        __syncthreads();   // every lane holds its slots in registers: the walk tables are free to be re-used
        SG_HOOK_RSTAMP(10);
        // (one wavefront per root: the totals are the wave's own reductions -- no trip through the LDS)
        const int32_t ns = NT > kWave ? red[8] : __builtin_amdgcn_readlane(incl, kWave - 1);
        if (tid == 0) a.nsize[i] = ns;
        const int32_t mn = NT > kWave ? min(min(red[0], red[1]), min(red[2], red[3])) : (int32_t)min(umn, 0x7FFFFFFFu);
        const int32_t mx = NT > kWave ? max(max(red[4], red[5]), max(red[6], red[7])) : vmax;
        // The sort: a distribution sort in LDS of up to three levels.
        //   Level 1: B <= NT buckets of equal id WIDTH (the top bits of id - min), one counter per lane: count with a returning LDS
        //     atomic (the return is the member's arrival order), scan into offsets.  Ids spread evenly over their range (a
        //     structureless graph) lie ~3 to a bucket and the sort is done.
        //   Level 2, where a bucket is still crowded (workgroup-uniform: > kFineAbove members): bucket b is cut into as many
        //     sub-buckets of equal width as it has members -- idx = start[b] + floor(offset in b's window * count_b / width), exact
        //     (umulhi) -- a monotone map of the ids onto [0, ns) that follows the set's own distribution (a piecewise-linear
        //     equalisation); 16-bit counters, two per word, scanned in place.  A graph with id locality puts most of a set into ONE
        //     community of consecutive ids = one bucket (32,768 ids wide on a 2.9 M-node graph).
        //   Level 3, where a sub-bucket is STILL crowded (> SG_SORT_AGAIN_ABOVE members; 16 measured against 12 / 24 / never:
        //     profiles/r68_again_ab.log): level 2 cuts a bucket that holds a community of 2,048 ids into sub-buckets
        //     ~130 ids wide, ~16 members each, and the ranking by counting below was an LDS latency loop of up to 30 trips (cit2loc:
        //     0.2 of the kernel's 0.68 ms -- profiles/r40_loc_phase_ms.log).  The same cut once more, of every level-2 sub-bucket by
        //     ITS count, with the position inside the sub-bucket's id window as a float: any monotone map of the ids of one sub-bucket
        //     onto [0, count) will do -- rounding only moves the balance, and float multiply / subtract / truncate are monotone.
        //     (Measured instead and not kept: a level 1 of 1,024 buckets in 16-bit counters, which makes level 3 rare -- its wider
        //     zeroing and scan cost every set more than level 3 costs the crowded ones: cit2loc 0.617 against 0.588 ms, twitter and
        //     collab +3-4 %, profiles/r43_sort_ab.log.)
        // What is left in a (sub-)bucket is ranked by counting.
        unsigned long long *A = (unsigned long long *)lds_raw;            // [ns <= stride] over the counts and the ids
        // 64-bit keys do not fit beside the id in a sort element: they stay where they were packed (KK[p], never moved) and the
        // element carries p -- (id << 32 | p) sorts like (id << 32 | key), ids being distinct
        unsigned long long *KK = A + a.stride;                            // [ns] (KR64 only)
        constexpr int CW = 4;                          // counter words per lane in the scan of levels 2 and 3
        constexpr int kFineAbove = 12;                 // a finer level from this many members in one (sub-)bucket on
        int32_t *start = (int32_t *)(lds_raw + (KR64 ? 16 : 8) * (size_t)a.stride);      // [B+1] level-1 counts, then offsets
        uint32_t *cnt2 = (uint32_t *)(start + NT + 4);                                   // [CW * NT] 16-bit counters of levels 2 and 3 (0 .. ns used), then offsets; 16-byte aligned
        int logb = 0;
        while ((1 << logb) < ns && (2 << logb) <= T / 4 && (2 << logb) <= NT) ++logb;
        const int B = 1 << logb;
        const uint32_t range = (uint32_t)(mx - mn) + 1u;
        const int Ls = (range <= 1u) ? 0 : (32 - __builtin_clz(range - 1u));
        const int bshift = Ls > logb ? Ls - logb : 0;
        if (tid < B) start[tid] = 0;
        if (SG_SORT_ZERO_UPFRONT) ((uint4 *)cnt2)[tid] = make_uint4(0u, 0u, 0u, 0u);
        {
            int p = wbase + incl - cnt;
#pragma unroll
            for (int u = 0; u < SPL; ++u)
                if (idv[u] != -1) {
                    if (KR64) {
                        A[p] = ((unsigned long long)(uint32_t)idv[u] << 32) | (unsigned long long)(uint32_t)p;
                        KK[p] = (unsigned long long)(kv[u] | (idv[u] == root ? lead : (CntT)0));
                    } else {
                        A[p] = ((unsigned long long)(uint32_t)idv[u] << 32) | (uint32_t)(kv[u] | (idv[u] == root ? lead : (CntT)0));
                    }
                    ++p;
                }
        }
        __syncthreads();
        SG_HOOK_RSTAMP(11);
        // exclusive scan of the CW * NT words of 16-bit counters in place (offsets <= ns < 2^16), CW consecutive words per lane: one
        // 16-byte LDS read and one write per lane; the words behind the last counter are zero; -> the largest count
        // (a wave's base = the totals of the waves in front of it: at most NT / 64 - 1 of them, spelled out -- as a loop over
        //  tid / 64 the compiler vectorised it into 16-byte reads with a scalar tail, ~60 instructions for one addition)
        auto scan16 = [&](uint32_t *cw) -> int32_t {
            static_assert(CW == 4, "one uint4 per lane");
            const uint4 v = ((const uint4 *)cw)[tid];
            const uint32_t w[CW] = {v.x, v.y, v.z, v.w};
            int32_t s2 = 0, mc = 0;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                s2 += (int32_t)((w[c] & 0xFFFFu) + (w[c] >> 16));
                mc = max(mc, max((int32_t)(w[c] & 0xFFFFu), (int32_t)(w[c] >> 16)));
            }
            const int32_t inc = wave_scan_add_i32_incl(s2);
            mc = wave_red_max_i32(mc);
            int32_t run = inc - s2;
            if (NT > kWave) {
                if ((tid & (kWave - 1)) == kWave - 1) red[12 + tid / kWave] = inc, red[4 + tid / kWave] = mc;
                __syncthreads();
#pragma unroll
                for (int w2 = 0; w2 < NT / kWave - 1; ++w2) run += w2 < tid / kWave ? red[12 + w2] : 0;
                mc = red[4];
#pragma unroll
                for (int w2 = 1; w2 < NT / kWave; ++w2) mc = max(mc, red[4 + w2]);
            }
            uint32_t o[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const uint32_t lo16 = (uint32_t)run;
                run += (int32_t)(w[c] & 0xFFFFu);
                o[c] = lo16 | ((uint32_t)run << 16);
                run += (int32_t)(w[c] >> 16);
            }
            ((uint4 *)cw)[tid] = make_uint4(o[0], o[1], o[2], o[3]);
            return mc;
        };
        constexpr int EPL = EPLP ? EPLP : (NT == 64 && SPL == 8 ? 7 : SPL);      // members per lane: ns <= EPL * NT (checked at launch)
        unsigned long long el[EPL];
        uint32_t st[EPL];           // the member's state across the phases (below); at the end: its final position in the row
        if (NT == kWave) {
            // ONE wavefront per root (the 2-hop shapes over int32 row offsets: collab): the two-level sort of rounds 3-5 with a member's
            // bucket, arrival order and bucket bounds in registers of their own.  The multi-level form below -- packed state, buckets
            // recomputed, a loop over the finer levels -- costs this shape 4-5 % on the structureless collab-like graph, and none of
            // its parts alone explains it (profiles/r54_collab_bisect.log, r57_collab_levels_ab.log, r59_collab_nt128_ab.log: not the
            // finer levels, not occupancy, not the SGPR cap, not the unrolled ranking loop; two waves per root: +25 %); this form has
            // the registers (66) and its sets are small (<= 408 members: level 2 leaves at most a handful per sub-bucket).
            constexpr int CW1 = 4;
            const int W2 = (ns + 2) / 2 + 1;               // words that hold counters 0 .. ns (counter ns stays 0: its offset is the total)
            uint32_t bk[EPL];
            int32_t pos[EPL];                           // arrival order inside the sub-bucket
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (e * NT >= ns) break;
                const int x = e * NT + tid;
                if (x < ns) {
                    el[e] = A[x];
                    bk[e] = ((uint32_t)(el[e] >> 32) - (uint32_t)mn) >> bshift;
                    pos[e] = atomicAdd(&start[bk[e]], 1);
                }
            }
            __syncthreads();
            SG_HOOK_RSTAMP(12);
            int32_t maxc;
            {
                const int32_t c = tid < B ? start[tid] : 0;
                const int32_t inc = wave_scan_add_i32_incl(c);
                maxc = wave_red_max_i32(c);
                const int32_t excl = inc - c;
                if (tid < B) start[tid] = excl;
                if (tid == B - 1) start[B] = excl + c;
            }
            __syncthreads();
            SG_HOOK_RSTAMP(13);
            int blo[EPL], bhi[EPL];
            if (maxc <= kFineAbove) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if (e * NT >= ns) break;
                    if (e * NT + tid < ns) {
                        blo[e] = start[bk[e]];
                        bhi[e] = start[bk[e] + 1];
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < CW1; ++c)
                    if (c * NT + tid < W2) cnt2[c * NT + tid] = 0u;
                __syncthreads();
                uint32_t idx2[EPL];
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if (e * NT >= ns) break;
                    if (e * NT + tid < ns) {
                        const uint32_t lo1 = (uint32_t)start[bk[e]], kb = (uint32_t)start[bk[e] + 1] - lo1;
                        const uint32_t off = ((uint32_t)(el[e] >> 32) - (uint32_t)mn) - (bk[e] << bshift);      // < 2^bshift
                        const uint32_t sub = bshift ? __umulhi(off << (32 - bshift), kb) : 0u;                 // floor(off * kb / 2^bshift) < kb
                        idx2[e] = lo1 + sub;
                        const uint32_t sh = (idx2[e] & 1u) * 16u;
                        pos[e] = (int32_t)((atomicAdd(&cnt2[idx2[e] >> 1], 1u << sh) >> sh) & 0xFFFFu);
                    }
                }
                __syncthreads();
                {   // exclusive scan of the ns + 1 level-2 counters, in place (offsets <= ns < 2^16): CW1 consecutive words per lane
                    uint32_t w[CW1];
                    int32_t s2 = 0;
#pragma unroll
                    for (int c = 0; c < CW1; ++c) {
                        const int x = tid * CW1 + c;
                        w[c] = x < W2 ? cnt2[x] : 0u;
                        s2 += (int32_t)((w[c] & 0xFFFFu) + (w[c] >> 16));
                    }
                    int32_t run = wave_scan_add_i32_incl(s2) - s2;
#pragma unroll
                    for (int c = 0; c < CW1; ++c) {
                        const int x = tid * CW1 + c;
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
                for (int e = 0; e < EPL; ++e) {
                    if (e * NT >= ns) break;
                    if (e * NT + tid < ns) {
                        blo[e] = off2[idx2[e]];
                        bhi[e] = off2[idx2[e] + 1];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (e * NT >= ns) break;
                if (e * NT + tid < ns) A[blo[e] + pos[e]] = el[e];         // every packed element was read before the barriers above
            }
            __syncthreads();
            SG_HOOK_RSTAMP(14);
            const uint32_t *Ahi = (const uint32_t *)A;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (e * NT >= ns) break;
                if (e * NT + tid < ns) {
                    const uint32_t me = (uint32_t)(el[e] >> 32);
                    int rank = 0;       // ids are distinct within a set: the high word of A decides
#pragma unroll 1
                    for (int t2 = blo[e]; t2 < bhi[e]; ++t2) rank += (Ahi[2 * t2 + 1] < me) ? 1 : 0;
                    st[e] = (uint32_t)(blo[e] + rank);
                }
            }
        } else {
            // A member's state across the phases of the sort: where its (sub-)bucket begins (while counting: the bucket's number), where it
            // ends, the member's arrival order inside it -- positions and counts are <= ns <= 1,020 (checked at launch).  PACK: ONE word per
            // member, 10 bits each, and the member's level-1 bucket worked out again wherever it is needed -- three more registers per
            // member were 14 more VGPRs and a wave slot per SIMD, which the graph with id locality feels.  (!PACK, the plain registers of
            // rounds 3-5, is kept as a form for the one-wave kernel, which loses ~4 % on collab to the new epilogue as a whole
            // (profiles/r54_collab_bisect.log, r57_collab_levels_ab.log: not to the finer levels, not to occupancy, not to the unroll) --
            // but unpacked it needs 80 VGPRs and spills, so it is off.)
            // What an LDS operation returns is never used in the pass that issued it outside the level loop: a lane's reads and returning
            // atomics of a pass are all in flight together.
            constexpr bool PACK = true;
            uint32_t s_hi[PACK ? 1 : EPL], s_pos[PACK ? 1 : EPL], s_bk[PACK ? 1 : EPL];
            auto LO = [&](int e) -> uint32_t { return PACK ? (st[e] & 0x3FFu) : st[e]; };
            auto HI = [&](int e) -> uint32_t { return PACK ? ((st[e] >> 10) & 0x3FFu) : s_hi[e]; };
            auto POS = [&](int e) -> uint32_t { return PACK ? (st[e] >> 20) : s_pos[e]; };
            auto BK = [&](int e) -> uint32_t { return PACK ? (((uint32_t)(el[e] >> 32) - (uint32_t)mn) >> bshift) : s_bk[e]; };
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (e * NT >= ns) break;
                const int x = e * NT + tid;
                if (x < ns) {
                    el[e] = A[x];
                    const uint32_t b1 = ((uint32_t)(el[e] >> 32) - (uint32_t)mn) >> bshift;
                    if (!PACK) s_bk[e] = b1;
                    (PACK ? st[e] : s_pos[e]) = (uint32_t)atomicAdd(&start[b1], 1);     // (the arrival order, as it comes)
                }
            }
            __syncthreads();
            SG_HOOK_RSTAMP(12);
            int32_t maxc;
            {   // level 1: exclusive scan over the B <= NT buckets, one bucket per lane
                const int32_t c = tid < B ? start[tid] : 0;
                const int32_t inc = wave_scan_add_i32_incl(c);
                const int32_t mc = wave_red_max_i32(c);
                int32_t base = 0;
                maxc = mc;
                if (NT > kWave) {
                    if ((tid & (kWave - 1)) == kWave - 1) red[12 + tid / kWave] = inc, red[4 + tid / kWave] = mc;
                    __syncthreads();
#pragma unroll
                    for (int w2 = 0; w2 < NT / kWave - 1; ++w2) base += w2 < tid / kWave ? red[12 + w2] : 0;
                    maxc = red[4];
#pragma unroll
                    for (int w2 = 1; w2 < NT / kWave; ++w2) maxc = max(maxc, red[4 + w2]);
                }
                const int32_t excl = base + inc - c;
                if (tid < B) start[tid] = excl;
                if (tid == B - 1) start[B] = excl + c;
            }
            __syncthreads();
            SG_HOOK_RSTAMP(13);
            const uint16_t *off2 = (const uint16_t *)cnt2;
            // (a crowded set's first trip through the level loop works the level-1 bounds out for itself and overwrites the state:
            //  the pass below is for the sets that skip the loop -- maxc is workgroup-uniform)
            const bool finer = SG_SORT_LEVELS > 0 && maxc > kFineAbove;
            if (finer) {
            } else if (PACK) {
                uint32_t ta[EPL], tb[EPL];         // what the first pass asked the LDS for, until the second packs it
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if (e * NT >= ns) break;
                    if (e * NT + tid < ns) ta[e] = (uint32_t)start[BK(e)], tb[e] = (uint32_t)start[BK(e) + 1];
                }
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if (e * NT >= ns) break;
                    if (e * NT + tid < ns) st[e] = (st[e] << 20) | ta[e] | (tb[e] << 10);
                }
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if (e * NT >= ns) break;
                    if (e * NT + tid < ns) st[e] = (uint32_t)start[BK(e)], s_hi[e] = (uint32_t)start[BK(e) + 1];
                }
            }
            // levels 2 and 3: the counters are the same words, scanned by the same code.  A (sub-)bucket of k members is cut into FINE * k
            // parts, FINE = as many as the counter array holds for this set: the scan covers all 2 * CW * NT counters whatever the set's
            // size, so the finer cut is free -- and on the graph with id locality it is what makes level 3 the exception (a community of
            // 2,048 ids in a window of 32,768 cut into 3 x 250 parts: ~5 members each instead of ~16)
            const uint32_t FINE = (uint32_t)(2 * CW * NT - 1) / (uint32_t)(ns + 1);      // >= 1: ns <= 2 * CW * NT - 2 (checked at launch)
#pragma unroll 1
            for (int lvl = 0; lvl < SG_SORT_LEVELS && maxc > (lvl ? SG_SORT_AGAIN_ABOVE : kFineAbove); ++lvl) {
                if (lvl || !SG_SORT_ZERO_UPFRONT) {
                    if (lvl) __syncthreads();        // every lane has read its offsets of the level before
                    ((uint4 *)cnt2)[tid] = make_uint4(0u, 0u, 0u, 0u);
                    __syncthreads();
                }
                // (two copies of the member loop, not one with `if (lvl)` inside: the compiler turned that branch into selects and every
                //  level-2 pass paid for level 3's float arithmetic -- 150 instructions a root on the one-wave form.  In here a member's
                //  LDS results ARE used on the spot: this loop is the cold path of the structureless workloads.)
                // during the count, LO(e) holds the number of the member's (sub-)bucket
                if (lvl == 0) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        if (e * NT >= ns) break;
                        if (e * NT + tid < ns) {
                            const uint32_t ido = (uint32_t)(el[e] >> 32) - (uint32_t)mn, b1 = ido >> bshift;       // (not BK(e): the cold path keeps no register for it)
                            const uint32_t lo1 = (uint32_t)start[b1], kb = ((uint32_t)start[b1 + 1] - lo1) * FINE;
                            const uint32_t off = ido - (b1 << bshift);                                             // < 2^bshift
                            const uint32_t idx = lo1 * FINE + (bshift ? __umulhi(off << (32 - bshift), kb) : 0u);  // + floor(off * kb / 2^bshift) < kb
                            const uint32_t sh = (idx & 1u) * 16u;
                            const uint32_t arrived = (atomicAdd(&cnt2[idx >> 1], 1u << sh) >> sh) & 0xFFFFu;
                            if (PACK) st[e] = idx | (arrived << 20);
                            else st[e] = idx, s_pos[e] = arrived;
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        if (e * NT >= ns) break;
                        if (e * NT + tid < ns) {
                            const uint32_t ido = (uint32_t)(el[e] >> 32) - (uint32_t)mn, b1 = ido >> bshift;
                            const uint32_t lo1 = (uint32_t)start[b1], kb = ((uint32_t)start[b1 + 1] - lo1) * FINE;
                            const uint32_t off = ido - (b1 << bshift);
                            const uint32_t sub = bshift ? __umulhi(off << (32 - bshift), kb) : 0u;
                            // off * kb / 2^bshift - sub: where inside its level-2 sub-bucket's id window the member lies, [0, 1)
                            const float fr = (float)off * __builtin_ldexpf((float)kb, -bshift) - (float)sub;
                            const uint32_t lo2 = LO(e);
                            const int k2 = ((int)HI(e) - (int)lo2) * (int)FINE;
                            const uint32_t idx = lo2 * FINE + (uint32_t)min(max((int)(fr * (float)k2), 0), k2 - 1);
                            const uint32_t sh = (idx & 1u) * 16u;
                            const uint32_t arrived = (atomicAdd(&cnt2[idx >> 1], 1u << sh) >> sh) & 0xFFFFu;
                            if (PACK) st[e] = idx | (arrived << 20);
                            else st[e] = idx, s_pos[e] = arrived;
                        }
                    }
                }
                __syncthreads();
                maxc = scan16(cnt2);
                __syncthreads();
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if (e * NT >= ns) break;
                    if (e * NT + tid < ns) {
                        const uint32_t idx = LO(e);
                        if (PACK) st[e] = (st[e] & 0x3FF00000u) | (uint32_t)off2[idx] | ((uint32_t)off2[idx + 1] << 10);
                        else st[e] = (uint32_t)off2[idx], s_hi[e] = (uint32_t)off2[idx + 1];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (e * NT >= ns) break;
                if (e * NT + tid < ns) A[LO(e) + POS(e)] = el[e];         // every packed element was read before the barriers above
            }
            __syncthreads();
            SG_HOOK_RSTAMP(14);
            const uint32_t *Ahi = (const uint32_t *)A;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (e * NT >= ns) break;
                if (e * NT + tid < ns) {
                    const uint32_t me = (uint32_t)(el[e] >> 32);
                    // ids are distinct within a set: the high word of A decides.  The first SG_SORT_RANK_HEAD members of the (sub-)bucket without a
                    // loop -- after the finer cuts few hold more, and a loop runs as long as the longest of a wave's 64 lanes needs, with
                    // a branch and an exec-mask round trip per trip: 60 vector instructions a wave on the graph with id locality
                    // (words behind the bucket's end are read and masked; behind the set's end they are still inside this workgroup's LDS)
                    const int lo = (int)LO(e), hi = (int)HI(e);
                    const uint32_t cnt = (uint32_t)(hi - lo);        // >= 1: the member itself
                    int rank = 0;
#pragma unroll
                    for (int i = 0; i < SG_SORT_RANK_HEAD; ++i) rank += (int)((Ahi[2 * (lo + i) + 1] < me) & (i == 0 || cnt > (uint32_t)i));
                    int t2 = lo + SG_SORT_RANK_HEAD;
#if SG_SORT_RANK_UNROLL
#pragma unroll 1         // (four members per trip: the loop is LDS latency, not issue)
                    for (; t2 + 3 < hi; t2 += 4) {
                        const uint32_t a0 = Ahi[2 * t2 + 1], a1 = Ahi[2 * t2 + 3], a2 = Ahi[2 * t2 + 5], a3 = Ahi[2 * t2 + 7];
                        rank += (a0 < me ? 1 : 0) + (a1 < me ? 1 : 0) + (a2 < me ? 1 : 0) + (a3 < me ? 1 : 0);
                    }
#endif
#pragma unroll 1
                    for (; t2 < hi; ++t2) rank += (Ahi[2 * t2 + 1] < me) ? 1 : 0;
                    st[e] = (uint32_t)(lo + rank);          // the member's final position
                }
            }
        }
        __syncthreads();        // every rank is known: the bucket-grouped array can become the sorted one, in place
        SG_HOOK_RSTAMP(15);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            if (e * NT >= ns) break;
            if (e * NT + tid < ns) A[st[e]] = el[e];
        }
        __syncthreads();
        int xs = tid;
        if (!KR64 && a.wide_rows) {      // rows begin on 16-byte boundaries (a pitch of whole lines): four members per lane and store
            typedef int v4i __attribute__((ext_vector_type(4)));
            const int n4 = ns >> 2;
            for (int q = tid; q < n4; q += NT) {
                const ulonglong2 v0 = ((const ulonglong2 *)A)[2 * q], v1 = ((const ulonglong2 *)A)[2 * q + 1];
                const v4i ids4 = {(int32_t)(v0.x >> 32), (int32_t)(v0.y >> 32), (int32_t)(v1.x >> 32), (int32_t)(v1.y >> 32)};
                const v4i sl4 = {(int32_t)(uint32_t)v0.x, (int32_t)(uint32_t)v0.y, (int32_t)(uint32_t)v1.x, (int32_t)(uint32_t)v1.y};
                SG_ROW_STORE((v4i *)&a.set_ids[obase + 4 * q], ids4);
                SG_ROW_STORE((v4i *)&a.set_slot[obase + 4 * q], sl4);
            }
            xs = (n4 << 2) + tid;       // the last ns % 4 members, one by one
        }
        for (int x = xs; x < ns; x += NT) {
            const unsigned long long v = A[x];
            SG_ROW_STORE(&a.set_ids[obase + x], (int32_t)(v >> 32));
            if (KR64) SG_ROW_STORE(&a.set_keys[obase + x], (uint64_t)KK[(uint32_t)v]);
            else SG_ROW_STORE(&a.set_slot[obase + x], (int32_t)(uint32_t)v);
        }
        return;
    }

    // ================= the set leaves as a finished SpG row =================
    // (1) fold the set's LP keys (a few dozen distinct rows) and register them in the HBM table of distinct rows with
    //     tag = (global root index)*stride + first visit number (any monotone function of the first-visit order numbers
    //     the rows like the reference's sequential pass, subg_acc.c:957-978); (2) bucket-sort the members by node id in
    //     the LDS the walk tables occupied (random_walks.py:79-80) and write them to their sorted position.
    int32_t idv[SPL];
    uint32_t mtag[SPL];
    CntT mkey[SPL];
    if (SPL % 4 == 0) {
#pragma unroll
        for (int c = 0; c < SPL / 4; ++c) {
            const int g = c * NT + tid;
            const int4 kk = ((const int4 *)keys)[g];
            const uint4 qq = ((const uint4 *)minq)[g];
            idv[4 * c] = kk.x, idv[4 * c + 1] = kk.y, idv[4 * c + (SPL > 2 ? 2 : 0)] = kk.z, idv[4 * c + (SPL > 2 ? 3 : 1)] = kk.w;
            mtag[4 * c] = qq.x, mtag[4 * c + 1] = qq.y, mtag[4 * c + (SPL > 2 ? 2 : 0)] = qq.z, mtag[4 * c + (SPL > 2 ? 3 : 1)] = qq.w;
            if (K32) {
                const uint4 v = ((const uint4 *)pk)[g];
                mkey[4 * c] = (CntT)v.x, mkey[4 * c + 1] = (CntT)v.y, mkey[4 * c + (SPL > 2 ? 2 : 0)] = (CntT)v.z, mkey[4 * c + (SPL > 2 ? 3 : 1)] = (CntT)v.w;
            } else {
                const uint4 v0 = ((const uint4 *)pk)[2 * g], v1 = ((const uint4 *)pk)[2 * g + 1];
                mkey[4 * c] = (CntT)(((unsigned long long)v0.y << 32) | v0.x);
                mkey[4 * c + 1] = (CntT)(((unsigned long long)v0.w << 32) | v0.z);
                mkey[4 * c + (SPL > 2 ? 2 : 0)] = (CntT)(((unsigned long long)v1.y << 32) | v1.x);
                mkey[4 * c + (SPL > 2 ? 3 : 1)] = (CntT)(((unsigned long long)v1.w << 32) | v1.z);
            }
        }
    } else {
        const int2 kk = ((const int2 *)keys)[tid];
        const uint2 qq = ((const uint2 *)minq)[tid];
        const uint4 v = ((const uint4 *)pk)[tid];
        idv[0] = kk.x, idv[1] = kk.y;
        mtag[0] = qq.x, mtag[1] = qq.y;
        mkey[0] = (CntT)(((unsigned long long)v.y << 32) | v.x);
        mkey[1] = (CntT)(((unsigned long long)v.w << 32) | v.z);
    }
    bool ok[SPL];
    uint32_t mf[SPL];
    int32_t slv[SPL];
    int mycount = 0;
    int32_t vmin = 0x7FFFFFFF, vmax = 0;
#pragma unroll
    for (int u = 0; u < SPL; ++u) {
        ok[u] = idv[u] != -1;
        mkey[u] |= (mtag[u] == 0 ? lead : (CntT)0);              // the root is visit 0
        mf[u] = fold_hash<kSpgFoldBits>(mkey[u]);
        slv[u] = -1;
        mycount += ok[u] ? 1 : 0;
        vmin = min(vmin, ok[u] ? idv[u] : 0x7FFFFFFF);
        vmax = max(vmax, ok[u] ? idv[u] : 0);
    }
    CntT fcur[SPL];
    uint32_t ftag[SPL];
#pragma unroll
    for (int u = 0; u < SPL; ++u) {
        fcur[u] = fk[mf[u]];
        ftag[u] = ft[mf[u]];
    }
#pragma unroll
    for (int u = 0; u < SPL; ++u) {   // a set holds a few dozen distinct keys -> the first probe nearly always settles it
        if (!ok[u]) continue;
        const CntT key = mkey[u];
        const uint32_t tagoff = mtag[u];
        CntT cur = fcur[u];
        // the reads above ran before anybody inserted: "empty" is claimed here, and losing the race to the same key is a hit
        if (cur == kNoKey) cur = atomicCAS(&fk[mf[u]], kNoKey, key);
        if (cur == kNoKey || cur == key) {
            // most lanes meet a tag that is already smaller: the plain read (stale only towards larger values) spares
            // the same-address atomic storm
            if (ftag[u] > tagoff) atomicMin(&ft[mf[u]], tagoff);
            slv[u] = -2 - (int32_t)mf[u];       // resolved to the HBM slot after the fold table is flushed
            continue;
        }
        uint32_t f = (mf[u] + 1) & (kSpgFold - 1);   // a real collision: probe on
        bool done = false;
#pragma unroll 1
        for (int p = 0; p < 15; ++p) {
            cur = fk[f];
            if (cur == kNoKey) cur = atomicCAS(&fk[f], kNoKey, key);
            if (cur == kNoKey || cur == key) {
                if (ft[f] > tagoff) atomicMin(&ft[f], tagoff);
                slv[u] = -2 - (int32_t)f;
                done = true;
                break;
            }
            f = (f + 1) & (kSpgFold - 1);
        }
        if (!done) slv[u] = uniq_global_insert(a.table, (unsigned long long)key, tag0 + (unsigned long long)tagoff, a.flags);
    }
    {
        vmin = wave_red_min_i32(vmin);
        vmax = wave_red_max_i32(vmax);
        mycount = wave_red_add_i32(mycount);
        if ((tid & (kWave - 1)) == 0) {
            red[tid / kWave] = vmin;
            red[4 + tid / kWave] = vmax;
            atomicAdd(&red[8], mycount);
        }
    }
    __syncthreads();   // every lane holds its members in registers: the walk tables are free to be re-used
    SG_HOOK_RSTAMP(3);
    const int32_t ns = red[8];             // no truncating bucket here: every member stays (ns <= M*MH+1 = stride)
    if (tid == 0 && !a.tags_only) a.nsize[i] = ns;
    const int32_t mn = min(min(red[0], red[1]), min(red[2], red[3]));
    const int32_t mx = max(max(red[4], red[5]), max(red[6], red[7]));
    // the walk tables are dead: [ns] (id << 32 | slot) grouped by bucket goes over the counts (and, K32, the ids behind them);
    // the bucket counts / offsets [B+1] (B <= min(T/4, NT)) over the ids -- K32: over the visit numbers
    unsigned long long *A = (unsigned long long *)lds_raw;
    int32_t *start = K32 ? (int32_t *)minq : keys;
    int logb = 0;
    while ((1 << logb) < ns && (2 << logb) <= T / 4 && (2 << logb) <= NT) ++logb;
    const int B = 1 << logb;
    const uint32_t range = (uint32_t)(mx - mn) + 1u;
    const int Ls = (range <= 1u) ? 0 : (32 - __builtin_clz(range - 1u));
    const int bshift = Ls > logb ? Ls - logb : 0;      // (id - mn) < 2^Ls, and ns <= range => logb <= Ls
    if (tid <= B) start[tid] = 0;
    if (tid < kSpgFold)    // flush the fold table to HBM (latency overlaps the sort)
        if (fk[tid] != kNoKey) fs[tid] = uniq_global_insert(a.table, (unsigned long long)fk[tid], tag0 + ft[tid], a.flags);
    if (a.tags_only) return;   // subgacc_walk_tags: the rows exist already (key rows); only the exact tags were wanted
    __syncthreads();
    SG_HOOK_RSTAMP(4);
    uint32_t bk[SPL];
    int32_t arr[SPL];                           // arrival order inside the bucket
#pragma unroll
    for (int u = 0; u < SPL; ++u) {
        bk[u] = (uint32_t)(idv[u] - mn) >> bshift;
        arr[u] = ok[u] ? atomicAdd(&start[bk[u]], 1) : 0;
        if (slv[u] <= -2) slv[u] = fs[-2 - slv[u]];
    }
    __syncthreads();
    SG_HOOK_RSTAMP(5);
    int32_t maxc;
    {   // exclusive scan over the B <= 256 buckets, one bucket per lane: wave scan, then the wave totals through LDS
        const int32_t c = tid < B ? start[tid] : 0;
        const int32_t inc = wave_scan_add_i32_incl(c);     // inclusive scan over the wave
        const int32_t mc = wave_red_max_i32(c);
        if ((tid & (kWave - 1)) == kWave - 1) red[12 + tid / kWave] = inc, red[4 + tid / kWave] = mc;
        __syncthreads();
        int32_t base = 0;
#pragma unroll
        for (int w2 = 0; w2 < NT / kWave - 1; ++w2) base += w2 < tid / kWave ? red[12 + w2] : 0;      // (spelled out: as a loop over tid / 64 the compiler vectorises it)
        maxc = red[4];
        for (int w2 = 1; w2 < NT / kWave; ++w2) maxc = max(maxc, red[4 + w2]);
        const int32_t excl = base + inc - c;
        if (tid < B) start[tid] = excl;
        if (tid == B - 1) start[B] = excl + c;
    }
    __syncthreads();
    SG_HOOK_RSTAMP(6);
    int blo[SPL], bhi[SPL];
    // A crowded bucket (a set whose ids sit in one community of a graph with id locality): the second level of the key-row
    // epilogue's sort, above -- sub-bucket = offset inside the bucket's id window scaled by the bucket's count -- so that the ranking
    // by counting below runs over ~1 member, not over the bucket (quadratic: section 4.10 of DESIGN.md).  The level-2 counters
    // (16 bits each) live where the sorted row is staged later: behind the bucket offsets (K32) / over the visit numbers.
    constexpr int kFineAbove = 12, CW = 4;
    const int W2 = (ns + 2) / 2 + 1;
    uint32_t *cnt2 = K32 ? (uint32_t *)(start + B + 1) : (uint32_t *)minq;
    if (!(maxc > kFineAbove && W2 <= CW * NT && (!K32 || B + 1 + W2 <= T))) {
#pragma unroll
        for (int u = 0; u < SPL; ++u) {   // bucket bounds, then the member goes to bucket start + arrival order
            blo[u] = ok[u] ? start[bk[u]] : 0;
            bhi[u] = ok[u] ? start[bk[u] + 1] : 0;
        }
    } else {
#pragma unroll
        for (int c = 0; c < CW; ++c)
            if (c * NT + tid < W2) cnt2[c * NT + tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SPL; ++u) {      // (bk[u] becomes the member's sub-bucket: the level-1 bucket is not needed again)
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
            const int32_t inc = wave_scan_add_i32_incl(s2);
            if ((tid & (kWave - 1)) == kWave - 1) red[12 + tid / kWave] = inc;
            __syncthreads();
            int32_t run = inc - s2;
#pragma unroll
            for (int w2 = 0; w2 < NT / kWave - 1; ++w2) run += w2 < tid / kWave ? red[12 + w2] : 0;
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
        for (int u = 0; u < SPL; ++u) {
            blo[u] = ok[u] ? off2[bk[u]] : 0;
            bhi[u] = ok[u] ? off2[bk[u] + 1] : 0;
        }
        __syncthreads();     // every bound is in registers before the staging areas (the same words) are written
    }
#pragma unroll
    for (int u = 0; u < SPL; ++u)
        if (ok[u]) A[blo[u] + arr[u]] = ((unsigned long long)(uint32_t)idv[u] << 32) | (uint32_t)slv[u];
    __syncthreads();
    SG_HOOK_RSTAMP(7);
    // order inside a bucket = number of smaller ids in it -> final position in the row.  The sorted row is assembled
    // in LDS (ids over the dead minq table, slots behind the bucket offsets) and leaves with consecutive lanes on
    // consecutive words.
    // (u64 counts: ids over minq, slots behind start[] in keys.  K32: A holds 8*ns <= 8*stride bytes of the 8*T-byte
    //  counts + ids region, the ids go behind it and the slots behind start[] in minq.)
    int32_t *fin_id = K32 ? (int32_t *)(lds_raw + 8 * (size_t)a.stride) : (int32_t *)minq;          // [ns]
    int32_t *fin_sl = start + B + 1;                                                                 // [ns]
    const bool staged = ns + B + 1 <= T && (!K32 || 12 * a.stride <= 8 * T);
    const uint32_t *Ahi = (const uint32_t *)A;
#pragma unroll
    for (int u = 0; u < SPL; ++u)
        if (ok[u]) {
            const int lo = blo[u], hi = bhi[u];
            int rank = 0;       // ids are distinct within a set: the high word of A decides
#pragma unroll 1         // a bucket holds one to a handful of members: an unrolled loop is all prologue
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
        for (int x = tid; x < ns; x += NT) {
            a.set_ids[obase + x] = fin_id[x];
            a.set_slot[obase + x] = fin_sl[x];
        }
    }
}

// returns 1 when this specialised form took the launch (else the caller launches walk_sets_kernel<SPG>)
int launch_walk_rows(const WalkArgs &a, bool indptr64, int rng_mode, size_t lds, hipStream_t s) {
#ifdef SG_DEV_NO_WALK_ROWS      // dev builds only: every shape through walk_sets_kernel<SPG> (A/B)
    return 0;
#endif
    if (!a.wo || a.step_major || a.walks || a.M > kWalkThreads || a.stride != a.M * a.m + 1) return 0;
    if (a.m < 2 || a.m > 4 || (a.T != 512 && a.T != 1024)) return 0;
    const int64_t grid = xcd_grid(a.worklist && a.work_cap > 0 && a.work_cap < a.n ? a.work_cap : a.n);
    if (grid >= (1ll << 31)) return 0;
    const bool rr = rng_mode == SUBGACC_RNG_RAND_R;
    // 512-slot tables: 128 lanes x 2 walks (dev builds: -DSG_DEV_ROWS_NT=256 forces one walk per lane)
#ifndef SG_DEV_ROWS_NT
#define SG_DEV_ROWS_NT 0
#endif
    constexpr bool nt256 = SG_DEV_ROWS_NT == 256;
    // key rows, 512-slot table, 2 hops, int32 row offsets: ONE wave per root (64 lanes x 4 walks x 8 slots, 55 VGPRs) -- no
    // barrier is a real one, the per-wave prologue and the scans are paid once: collab walk 0.296 -> 0.262 ms; neutral with
    // int64 offsets (twitter), and 13 % slower for the 1,024-slot table (16 slots per lane: 95 VGPRs), so only here.
    // (dev builds: -DSG_DEV_ROWS_NT=128 keeps two waves, =64 takes one wave with int64 offsets too)
    constexpr bool nt64 = SG_DEV_ROWS_NT == 64;
    constexpr bool nt128 = SG_DEV_ROWS_NT == 128;
    const bool half = a.T == 512 && !nt256;
    const bool rec = a.recs != nullptr && indptr64 == (a.rec.id_bits == 0);   // hop records: one dependent read per hop
    if (a.keyrows && a.m * a.shift + 1 > 31) {
        // 64-bit key rows (round 4): the 4-hop shapes whose keys need 33+ bits (M = 128 .. 204: 4 x 8 + 1) -- 64-bit counts in the
        // walk tables, the keys leave through set_keys.  1,024-slot table, 128 lanes x 2 walks x 8 slots, ~15 KB of LDS per root.
        if (a.m != 4 || a.m * a.shift + 1 > 63 || a.T != 1024 || !a.set_keys || ((int64_t)a.stride + 2) / 2 + 1 > 4 * 128) return 0;
        const size_t lds64 = kr_red_offset(1024, a.M, a.stride, 128, true) + 64 + 16;
#define SG_KR64W_E(I64, RNGM, EP)                                                                                     \
    do {                                                                                                             \
        if (rec)                                                                                                     \
            hipLaunchKernelGGL((walk_rows_kernel<I64, RNGM, 4, 8, 128, I64 ? 16 : 8, false, true, EP>), dim3((unsigned)grid), dim3(128), lds64, s, a); \
        else                                                                                                         \
            hipLaunchKernelGGL((walk_rows_kernel<I64, RNGM, 4, 8, 128, 0, false, true, EP>), dim3((unsigned)grid), dim3(128), lds64, s, a); \
        return 1;                                                                                                    \
    } while (0)
#define SG_KR64W(I64, RNGM)                                  \
    do {                                                     \
        if (a.stride <= 7 * 128) SG_KR64W_E(I64, RNGM, 7);   \
        SG_KR64W_E(I64, RNGM, 0);                            \
    } while (0)
        if (indptr64) {
            if (rr) SG_KR64W(true, SUBGACC_RNG_RAND_R);
            SG_KR64W(true, SUBGACC_RNG_PHILOX);
        }
        if (rr) SG_KR64W(false, SUBGACC_RNG_RAND_R);
        SG_KR64W(false, SUBGACC_RNG_PHILOX);
#undef SG_KR64W
#undef SG_KR64W_E
    }
    if (a.keyrows) {      // rows that carry the LP key itself: 32-bit counts, 128 lanes, 2 to 4 hops -- or not at all
        // (the epilogue's sort lives over the dead walk tables AND the Fisher-Yates draws behind them: packed members 8*stride,
        //  level-1 offsets 4*(128+1) [start + NT + 1: NT <= 128], level-2 counters 2 bytes each; CW * NT words cover them)
        if (a.m * a.shift + 1 > 31 || a.m > 4 || ((int64_t)a.stride + 2) / 2 + 1 > 4 * (a.T == 512 ? 64 : 128)) return 0;
        // (the reduction words sit behind the walk tables and behind the epilogue's sort, whichever reaches further: kr_red_offset)
        const bool one_wave = (nt64 || (!indptr64 && !nt128)) && a.m == 2 && a.T == 512;
        // (SG_ROWS_KR_LDS_MIN, dev builds: LDS asked for per two-wave workgroup at least -- 160 KB / n roots per CU; A/B of 12 / 14 / 16 roots)
#ifndef SG_ROWS_KR_LDS_MIN
#define SG_ROWS_KR_LDS_MIN 0
#endif
        size_t ldsk = kr_red_offset(a.T, a.M, a.stride, one_wave ? 64 : 128, false) + 64 + 16;
        if (!one_wave && ldsk < (size_t)SG_ROWS_KR_LDS_MIN) ldsk = SG_ROWS_KR_LDS_MIN;
#define SG_KR_E(I64, RNGM, MHH, SPLL, EP)                                                                            \
    do {                                                                                                             \
        if (rec)                                                                                                     \
            hipLaunchKernelGGL((walk_rows_kernel<I64, RNGM, MHH, SPLL, 128, I64 ? 16 : 8, true, true, EP>), dim3((unsigned)grid), dim3(128), ldsk, s, a); \
        else                                                                                                         \
            hipLaunchKernelGGL((walk_rows_kernel<I64, RNGM, MHH, SPLL, 128, 0, true, true, EP>), dim3((unsigned)grid), dim3(128), ldsk, s, a); \
        return 1;                                                                                                    \
    } while (0)
#define SG_KR(I64, RNGM, MHH, SPLL) SG_KR_E(I64, RNGM, MHH, SPLL, 0)
#define SG_KR8(I64, RNGM, MHH)                                           \
    do {                                                                 \
        if (a.stride <= 5 * 128) SG_KR_E(I64, RNGM, MHH, 8, 5);          \
        SG_KR_E(I64, RNGM, MHH, 8, 0);                                   \
    } while (0)
#define SG_KR64(I64, RNGM, MHH, SPLL)                                                                                \
    do {                                                                                                             \
        if (rec)                                                                                                     \
            hipLaunchKernelGGL((walk_rows_kernel<I64, RNGM, MHH, SPLL, 64, I64 ? 16 : 8, true, true>), dim3((unsigned)grid), dim3(64), ldsk, s, a); \
        else                                                                                                         \
            hipLaunchKernelGGL((walk_rows_kernel<I64, RNGM, MHH, SPLL, 64, 0, true, true>), dim3((unsigned)grid), dim3(64), ldsk, s, a); \
        return 1;                                                                                                    \
    } while (0)
#define SG_KR_MH(I64, RNGM)                                  \
    do {                                                     \
        if (a.T == 1024) {                                   \
            if (a.m == 2) SG_KR8(I64, RNGM, 2);              \
            if (a.m == 3) SG_KR8(I64, RNGM, 3);              \
            SG_KR8(I64, RNGM, 4);                            \
        }                                                    \
        if ((nt64 || (!I64 && !nt128)) && a.m == 2) SG_KR64(I64, RNGM, 2, 8); \
        if (a.m == 2) SG_KR(I64, RNGM, 2, 4);                \
        if (a.m == 3) SG_KR(I64, RNGM, 3, 4);                \
        SG_KR(I64, RNGM, 4, 4);                              \
    } while (0)
        if (indptr64) {
            if (rr) SG_KR_MH(true, SUBGACC_RNG_RAND_R);
            SG_KR_MH(true, SUBGACC_RNG_PHILOX);
        }
        if (rr) SG_KR_MH(false, SUBGACC_RNG_RAND_R);
        SG_KR_MH(false, SUBGACC_RNG_PHILOX);
#undef SG_KR_MH
#undef SG_KR64
#undef SG_KR
#undef SG_KR8
#undef SG_KR_E
    }
    // 1,024-slot table with counts that fit 32 bits: 128 lanes x 8 slots, 12 bytes of LDS per slot -> 11 roots per CU
    // (dev builds: -DSG_DEV_ROWS_NT=256 keeps the 256-lane form).  int32 row offsets + 8-byte records / plain CSR only.
    if (a.T == 1024 && !nt256 && !indptr64 && a.m * a.shift + 1 <= 31 && a.m <= 3) {
        const size_t lds32 = (size_t)a.T * 12 + (size_t)a.M * 4 + 8 + (size_t)kSpgFold * 12 + 64 + 16;
#define SG_ROWS32(RNGM, MHH)                                                                                        \
    do {                                                                                                            \
        if (rec)                                                                                                    \
            hipLaunchKernelGGL((walk_rows_kernel<false, RNGM, MHH, 8, 128, 8, true>), dim3((unsigned)grid), dim3(128), lds32, s, a); \
        else                                                                                                        \
            hipLaunchKernelGGL((walk_rows_kernel<false, RNGM, MHH, 8, 128, 0, true>), dim3((unsigned)grid), dim3(128), lds32, s, a); \
        return 1;                                                                                                   \
    } while (0)
        if (rr) {
            if (a.m == 2) SG_ROWS32(SUBGACC_RNG_RAND_R, 2);
            SG_ROWS32(SUBGACC_RNG_RAND_R, 3);
        }
        if (a.m == 2) SG_ROWS32(SUBGACC_RNG_PHILOX, 2);
        SG_ROWS32(SUBGACC_RNG_PHILOX, 3);
#undef SG_ROWS32
    }
#define SG_ROWS(I64, RNGM, MHH, SPLL, NTT)                                                                         \
    do {                                                                                                           \
        if (rec)                                                                                                   \
            hipLaunchKernelGGL((walk_rows_kernel<I64, RNGM, MHH, SPLL, NTT, I64 ? 16 : 8>), dim3((unsigned)grid), dim3(NTT), lds, s, a); \
        else                                                                                                       \
            hipLaunchKernelGGL((walk_rows_kernel<I64, RNGM, MHH, SPLL, NTT>), dim3((unsigned)grid), dim3(NTT), lds, s, a); \
        return 1;                                                                                                  \
    } while (0)
#define SG_ROWS_MH(I64, RNGM, SPLL, NTT)                      \
    do {                                                      \
        if (a.m == 2) SG_ROWS(I64, RNGM, 2, SPLL, NTT);       \
        if (a.m == 3) SG_ROWS(I64, RNGM, 3, SPLL, NTT);       \
        SG_ROWS(I64, RNGM, 4, SPLL, NTT);                     \
    } while (0)
#define SG_ROWS_SPL(I64, RNGM)                                \
    do {                                                      \
        if (a.T == 1024) SG_ROWS_MH(I64, RNGM, 4, 256);       \
        if (half) SG_ROWS_MH(I64, RNGM, 4, 128);              \
        SG_ROWS_MH(I64, RNGM, 2, 256);                        \
    } while (0)
    if (indptr64) {
        if (rr) SG_ROWS_SPL(true, SUBGACC_RNG_RAND_R);
        SG_ROWS_SPL(true, SUBGACC_RNG_PHILOX);
    }
    if (rr) SG_ROWS_SPL(false, SUBGACC_RNG_RAND_R);
    SG_ROWS_SPL(false, SUBGACC_RNG_PHILOX);
#undef SG_ROWS_SPL
#undef SG_ROWS_MH
#undef SG_ROWS
    return 0;
}

}  // namespace subgacc
