// walk_pipe.hip -- persistent, software-pipelined form of the walk kernel (gfx950).
//
// walk_sets_kernel (walk.hip) gives one workgroup to one root: clear the LDS tables, walk, dedup, rank / fold /
// sort, write.  Only ~30-50 % of a workgroup's lifetime issues the random CSR reads the kernel is bound by, LDS
// (8 workgroups per CU) forbids more residents, and so the epilogues show up in the kernel time (measured with
// cycle stamps, tools/walk_phases.py).  Here 8 x 256 CUs workgroups stay resident and each walks a strided
// sequence of roots as a two-stage pipeline:
//
//     walk of root k+1 (one lane per walk, the visited nodes stay in registers, one dependent load in flight)
//        ||  table work of root k (LDS inserts, first-visit rank, row write)
//
// The loads of the next walk are issued at the seams of the table phases and survive the workgroup barriers
// (plain global loads are not drained by __syncthreads; hipcc waits for them at their first use), so every resident
// workgroup always has its M lanes' loads outstanding.  The LDS tables are cleared once; every root hands them
// back clean (only the slots it used are reset).  Results are bit-identical to walk_sets_kernel.
//
// Scope: the set_sampler form (first hop without replacement, walk-major order, no raw walks), M <= 256 walks,
// 1..6 hops, first-visit-ordered sets as output.  Everything else -- including the fused-row form, whose epilogue
// pushed this structure to ~110 VGPRs and half the residency -- takes walk_sets_kernel.
// Measured (round 1): collab-like graph (L2 resident) 0.432 vs 0.463 ms, cit2-like 1.19 vs 1.16 ms (memory bound
// either way).
#include <stdlib.h>
#include <type_traits>

#include "walk_common.hpp"

#ifndef SG_PIPE_MINW   // minimum waves per SIMD asked of the register allocator for the NT < 256 forms
#define SG_PIPE_MINW 8
#endif

namespace subgacc {

// NT lanes per workgroup, WPL = ceil(M / NT) walks per lane (their dependent loads are independent of each other and
// all in flight together): NT = 128 / WPL = 2 puts twice as many roots on a CU in the same 32 wave slots.
template <bool IDX64, int RNG, int MH, int NT, int WPL>
__global__ __launch_bounds__(NT, NT == 256 ? 8 : SG_PIPE_MINW) __attribute__((amdgpu_num_sgpr(80))) void walk_pipe_kernel(const WalkArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    unsigned long long *pk = (unsigned long long *)lds_raw;     // [T]
    int32_t *keys = (int32_t *)(pk + a.T);                       // [T]
    uint32_t *minq = (uint32_t *)(keys + a.T);                   // [T]
    uint32_t *bitmap = minq + a.T;                               // [nwords]
    uint32_t *prefix = bitmap + a.nwords;                        // [nwords + 1]
    int32_t *sarr = (int32_t *)(prefix + a.nwords + 1);          // [M] Fisher-Yates draws of the NEXT root
    uint16_t *inv = (uint16_t *)(sarr + a.M);                    // [M*m+1] table slot of the member ranked r

    const int tid = threadIdx.x;
    const int M = a.M, T = a.T;
    const uint32_t tmask = (uint32_t)T - 1u;
    const unsigned long long lead = 1ull << (MH * a.shift);
    bool walker[WPL];                                            // walk tid + k*NT of the root
#pragma unroll
    for (int k = 0; k < WPL; ++k) walker[k] = tid + k * NT < M;

    for (int h = tid; h < T; h += NT) {                          // the only full clear of the tables
        keys[h] = -1;
        minq[h] = 0xFFFFFFFFu;
        pk[h] = 0ull;
    }
    for (int x = tid; x < a.nwords; x += NT) bitmap[x] = 0u;
    __syncthreads();

    // ------------------------------------------------------------------ state of the walk in flight ("B")
    int64_t iB = blockIdx.x;
    bool hasB = iB < a.n;
    int32_t rootB = 0;
    int64_t rbegB = 0;
    uint32_t rdegB = 0, rposB = 0, rseedB = a.seed;
    bool shufB = false;
    int32_t visB[WPL][MH];
    uint32_t xB[WPL], phB[WPL][2];
    int32_t pendN[WPL];            // node whose load is in flight
    int64_t pendB[WPL], pendD[WPL]; // row (begin, degree) whose load is in flight
#pragma unroll
    for (int k = 0; k < WPL; ++k) {
        xB[k] = 0, pendN[k] = 0, pendB[k] = 0, pendD[k] = 0;
#pragma unroll
        for (int s = 0; s < MH; ++s) visB[k][s] = 0;
    }

    bool sawBad = false;
    auto load_root = [&]() {      // block-uniform: every lane reads the same words
        rootB = a.query[iB];
        // a root outside [0, num_nodes) is never looked up -- without a branch and without an atomic in here: the row
        // load stays unconditional (of row 0) and keeps its place in the software pipeline (either one cost 20-25 % on
        // the cit2 batch)
        const bool bad = (uint64_t)(int64_t)rootB >= (uint64_t)a.num_nodes;
        int64_t d64;
        load_row<IDX64>(a.indptr, bad ? 0 : rootB, rbegB, d64);
        if (bad) d64 = 0;                              // handled like an isolated root ...
        sawBad |= bad;                                 // ... and flagged once, after the loop (no atomic inside it)
        if (a.cap_root && d64 > kNeighCap) d64 = kNeighCap;
        rdegB = (uint32_t)d64;
        if (RNG == SUBGACC_RNG_RAND_R) {
            rposB = a.rng_pos[iB];
            rseedB = a.rng_seed[iB];
        }
        shufB = d64 > M;
    };
    auto draws_to_lds = [&]() {   // partial Fisher-Yates draws s_k = draw % (deg-k) + k (subg_acc.c:769-775)
#pragma unroll
        for (int k = 0; k < WPL; ++k)
            if (walker[k]) {
                const int w = tid + k * NT;
                uint32_t r;
                if (RNG == SUBGACC_RNG_RAND_R) {
                    uint32_t x = lcg_jump(rseedB, rposB + 3u * (uint32_t)w);
                    r = rand_r_next(x);
                    sarr[w] = (int32_t)(r % (rdegB - (uint32_t)w)) + w;
                } else {
                    uint32_t o1;
                    philox2x32_10((uint32_t)rootB, (uint32_t)w | kPhiloxShuffle, a.seed, r, o1);
                    sarr[w] = (int32_t)philox_below(r, rdegB - (uint32_t)w) + w;
                }
            }
    };
    // step J of the chain of dependent loads of walk B.  J = 0: first hop; odd J: the node of hop (J+1)/2 has
    // arrived -> ask for its row; even J: the row has arrived -> draw, ask for the next node; J = 2*MH-1: last node.
    auto bstep = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if (rdegB == 0) return;
#pragma unroll
        for (int k = 0; k < WPL; ++k) {
            if (!walker[k]) continue;
            const int w = tid + k * NT;
            if constexpr (J == 0) {
                uint32_t pick;
                if (shufB) {
                    int32_t p = sarr[w];
                    for (int j = w - 1; j >= 0; --j)
                        if (sarr[j] == p) p = j;
                    pick = (uint32_t)p;
                } else {
                    pick = (uint32_t)w % rdegB;
                }
                if (RNG == SUBGACC_RNG_RAND_R)
                    xB[k] = lcg_jump(rseedB, rposB + 3u * ((shufB ? (uint32_t)M : 0u) + (uint32_t)w * (uint32_t)(MH - 1)));
                pendN[k] = SG_NEIGH_LOAD(&a.indices[rbegB + pick]);
            } else if constexpr (J % 2 == 1) {
                constexpr int hop = (J - 1) / 2;          // 0-based hop whose node just arrived
                visB[k][hop] = pendN[k];
                if constexpr (hop + 1 < MH) load_row<IDX64>(a.indptr, pendN[k], pendB[k], pendD[k]);
            } else {
                constexpr int hop = J / 2;                // hop being taken now (1-based index into the RNG stream - 1)
                if (pendD[k] > 0) {
                    uint32_t r;
                    if (RNG == SUBGACC_RNG_RAND_R) {
                        r = rand_r_next(xB[k]);
                    } else {
                        constexpr int idx = hop - 1;
                        if constexpr ((idx & 1) == 0)
                            philox2x32_10((uint32_t)rootB, (uint32_t)w | ((uint32_t)(idx >> 1) << kPhiloxBlockShift), a.seed,
                                          phB[k][0], phB[k][1]);
                        r = phB[k][idx & 1];
                    }
                    pendN[k] = SG_NEIGH_LOAD(&a.indices[pendB[k] + (int64_t)(RNG == SUBGACC_RNG_RAND_R ? r % (uint32_t)pendD[k]
                                                                                                       : philox_below(r, (uint32_t)pendD[k]))]);
                } else {
                    pendN[k] = visB[k][hop - 1];          // dead end: stay (the rand_r stream is then not reproducible)
                    if (RNG == SUBGACC_RNG_RAND_R) atomicOr(&a.flags[0], 1);
                }
            }
        }
    };
#define SG_BSTEP(JJ)                                                  \
    do {                                                              \
        if constexpr ((JJ) < 2 * MH) {                                \
            if (hasB) bstep(std::integral_constant<int, (JJ) < 2 * MH ? (JJ) : 0>{}); \
        }                                                             \
    } while (0)

    // prologue: the first root of this workgroup walks without anything to hide behind
    if (hasB) {
        load_root();
        if (shufB) draws_to_lds();
    }
    __syncthreads();
    SG_BSTEP(0);
    // BSTEP(0) follows the Fisher-Yates chain through sarr[0..tid) -- reads that cross waves -- and the first loop
    // iteration overwrites sarr with the NEXT root's draws: every wave must be through its chain before that
    // (in the loop the barriers b1..b5 stand between the two)
    __syncthreads();
    SG_BSTEP(1); SG_BSTEP(2); SG_BSTEP(3); SG_BSTEP(4); SG_BSTEP(5);
    SG_BSTEP(6); SG_BSTEP(7); SG_BSTEP(8); SG_BSTEP(9); SG_BSTEP(10); SG_BSTEP(11);

    while (hasB) {
        // ---- root A := finished walk B; B := next root of this workgroup
        const int64_t i = iB;
        const int32_t root = rootB;
        const bool isolated = rdegB == 0;
        int32_t visA[WPL][MH];
#pragma unroll
        for (int k = 0; k < WPL; ++k)
#pragma unroll
            for (int s = 0; s < MH; ++s) visA[k][s] = visB[k][s];
        iB += gridDim.x;
        hasB = iB < a.n;
        if (hasB) {
            load_root();
            if (shufB) draws_to_lds();           // sarr belongs to B alone
        }
        const int64_t obase = i * (int64_t)a.pitch;

        // ---- P1: visits of A -> tables (the root is member 0, q = 0)
        if (!isolated) {
            if (tid == 0) {
                const uint32_t h = ((uint32_t)root * 2654435761u) >> a.tshift;
                // the root may collide with nothing yet: the tables are clean
                keys[h] = root;
                minq[h] = 0u;
            }
        }
        __syncthreads();                          // b0: root slot visible; B's Fisher-Yates draws visible
        SG_BSTEP(0);
        if (!isolated) {
#pragma unroll
            for (int k = 0; k < WPL; ++k) {
                if (!walker[k]) continue;
#pragma unroll
                for (int s = 0; s < MH; ++s) {
                    const int32_t cur = visA[k][s];
                    uint32_t h = ((uint32_t)cur * 2654435761u) >> a.tshift;
                    while (true) {
                        const int32_t old = atomicCAS(&keys[h], -1, cur);
                        if (old == -1 || old == cur) break;
                        h = (h + 1u) & tmask;
                    }
                    atomicMin(&minq[h], (uint32_t)((tid + k * NT) * MH + s + 1));
                    atomicAdd(&pk[h], 1ull << ((MH - 1 - s) * a.shift));
                }
            }
        }
        __syncthreads();                          // b1
        SG_BSTEP(1);

        if (isolated) {   // one member, every count = M (subg_acc.c:753-761); id = the root
            if (tid == 0) {
                unsigned long long k = lead;
                for (int s = 0; s < MH; ++s) k |= (unsigned long long)M << (s * a.shift);
                a.set_ids[obase] = root;
                a.set_keys[obase] = k;
                a.nsize[i] = 1;
            }
            SG_BSTEP(2); SG_BSTEP(3); SG_BSTEP(4); SG_BSTEP(5); SG_BSTEP(6);
            SG_BSTEP(7); SG_BSTEP(8); SG_BSTEP(9); SG_BSTEP(10); SG_BSTEP(11);
            __syncthreads();
            continue;
        }

        // ---- P2..P4: first-visit rank (the set_sampler output order), row write, tables handed back clean
        for (int h = tid; h < T; h += NT)
            if (keys[h] != -1) {
                const uint32_t q = minq[h];
                atomicOr(&bitmap[q >> 5], 1u << (q & 31u));
            }
        __syncthreads();                          // b2
        SG_BSTEP(2);
        for (int x = tid; x <= a.nwords; x += NT) {
            uint32_t s = 0;
            for (int j = 0; j < x; ++j) s += __popc(bitmap[j]);
            prefix[x] = s;
        }
        __syncthreads();                          // b3
        SG_BSTEP(3);
        const int32_t total = (int32_t)prefix[a.nwords];
        const int32_t ns = total < a.stride ? total : a.stride;
        // rank -> table slot first (LDS), then the row leaves with consecutive lanes on consecutive words: scattered
        // 4/8-byte stores cost the memory pipeline as much as the walk's random reads do
        for (int h = tid; h < T; h += NT)
            if (keys[h] != -1) {
                const uint32_t q = minq[h];
                const int32_t r = (int32_t)(prefix[q >> 5] + __popc(bitmap[q >> 5] & ((1u << (q & 31u)) - 1u)));
                if (r < a.stride) {
                    inv[r] = (uint16_t)h;
                } else {                          // members ranked past the bucket are dropped (:814-828)
                    keys[h] = -1;
                    minq[h] = 0xFFFFFFFFu;
                    pk[h] = 0ull;
                }
            }
        __syncthreads();                          // b3': inv complete
        for (int x = tid; x < ns; x += NT) {
            const int h = inv[x];
            a.set_ids[obase + x] = keys[h];
            a.set_keys[obase + x] = pk[h] | (x == 0 ? lead : 0ull);
            keys[h] = -1;                         // hand the slot back clean
            minq[h] = 0xFFFFFFFFu;
            pk[h] = 0ull;
        }
        __syncthreads();                          // b4: tables clean again, bitmap no longer read
        for (int x = tid; x < a.nwords; x += NT) bitmap[x] = 0u;
        if (tid == 0) {
            a.nsize[i] = ns;
            if (total > a.stride) atomicAdd(&a.flags[1], 1);
        }
        SG_BSTEP(4); SG_BSTEP(5); SG_BSTEP(6); SG_BSTEP(7); SG_BSTEP(8); SG_BSTEP(9); SG_BSTEP(10); SG_BSTEP(11);
        __syncthreads();                          // b5: bitmap clean before the next root ranks
    }
#undef SG_BSTEP
    if (sawBad && tid == 0) atomicOr(&a.flags[3], 16);
}

template <bool IDX64, int RNG, int NT, int WPL>
static int launch_hops(const WalkArgs &a, size_t lds, int per_cu, hipStream_t s) {
#define SG_PIPE(MHH)                                                                                              \
    case MHH: {                                                                                                   \
        const void *fn = (const void *)walk_pipe_kernel<IDX64, RNG, MHH, NT, WPL>;                                \
        if (lds > 64 * 1024 &&                                                                                    \
            hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)          \
            return 0;                                                                                             \
        /* the grid is exactly what stays resident (registers may admit fewer workgroups than LDS and wave slots) */ \
        static int occ_lds = -1, occ = 0;                                                                         \
        if (occ_lds != (int)lds) {                                                                                \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, NT, lds) != hipSuccess || occ < 1) occ = per_cu; \
            occ_lds = (int)lds;                                                                                   \
        }                                                                                                         \
        int64_t grid = (int64_t)256 * (occ < per_cu ? occ : per_cu);                                              \
        if (grid > a.n) grid = a.n;                                                                               \
        hipLaunchKernelGGL((walk_pipe_kernel<IDX64, RNG, MHH, NT, WPL>), dim3((unsigned)grid), dim3(NT), lds, s,  \
                           a);                                                                                    \
        return 1;                                                                                                 \
    }
    switch (a.m) {
        SG_PIPE(1)
        SG_PIPE(2)
        SG_PIPE(3)
        SG_PIPE(4)
        SG_PIPE(5)
        SG_PIPE(6)
        default: return 0;
    }
#undef SG_PIPE
}

int launch_walk_pipe(const WalkArgs &a, bool indptr64, int rng_mode, bool spg, size_t lds, hipStream_t s) {
    // The fused-row (SPG) form stays with walk_sets_kernel: pipelined, its epilogue needs ~110 VGPRs, which halves the
    // resident workgroups and made it 35 % slower than the plain form (measured, round 1).
    if (spg || !a.wo || a.step_major || a.walks || a.M > kWalkThreads || a.m < 1 || a.m > 6) return 0;
    if (lds > (size_t)kLdsBytes) return 0;
    // lanes per workgroup (dev builds: -DSG_DEV_PIPE_NT=64|128)
#ifndef SG_DEV_PIPE_NT
#define SG_DEV_PIPE_NT 256
#endif
    const int nt = SG_DEV_PIPE_NT;
    int per_cu = (int)((size_t)kLdsBytes / lds);
    const int slots = 32 / (nt / kWave);                      // 32 waves per CU
    if (per_cu > slots) per_cu = slots;
    if (per_cu < 1) return 0;
    const bool rr = rng_mode == SUBGACC_RNG_RAND_R;
#define SG_PIPE_NT(NTT, WPLL)                                                                      \
    do {                                                                                           \
        if (indptr64)                                                                              \
            return rr ? launch_hops<true, SUBGACC_RNG_RAND_R, NTT, WPLL>(a, lds, per_cu, s)     \
                      : launch_hops<true, SUBGACC_RNG_PHILOX, NTT, WPLL>(a, lds, per_cu, s);    \
        return rr ? launch_hops<false, SUBGACC_RNG_RAND_R, NTT, WPLL>(a, lds, per_cu, s)        \
                  : launch_hops<false, SUBGACC_RNG_PHILOX, NTT, WPLL>(a, lds, per_cu, s);       \
    } while (0)
    if (nt == 64) SG_PIPE_NT(64, 4);
    if (nt == 128) SG_PIPE_NT(128, 2);
    SG_PIPE_NT(256, 1);
#undef SG_PIPE_NT
}

}  // namespace subgacc
