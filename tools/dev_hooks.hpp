// tools/dev_hooks.hpp -- DEV ONLY.  The timing experiments of the walk and join kernels: builds that skip a phase of a kernel
// (and therefore give WRONG results) to see what that phase costs, cycle stamps, "stop after phase k" builds for dynamic
// instruction counts.  Never part of libsubgacc_hip.so: the tools/*.sh scripts force-include this file
//     hipcc ... -include tools/dev_hooks.hpp -DSG_EXPERIMENT=k | -DSJ_EXPERIMENT=k | -DSG_STOP_AFTER=k -c csrc/<file>.hip -o /tmp/...
// into experiment objects under /tmp, linked into /tmp/libsubgacc_*.so and selected with SUBGACC_LIB.
// The product sources only carry the hook POINTS (no-ops: csrc/common.hpp).
//
//   SG_EXPERIMENT (walk.hip)  1 traversal only, no dedup | 2 dedup only, no graph reads after the first hop | 6 no row-pointer read
//                             7 per-phase cycle shares (tools/walk_phases.py) | 8 no registration in the HBM table
//   SG_EXPERIMENT (walk_rows.hip) 9 the last hop's read hits L2 (no missed line) | 10 every later hop's does
//   SG_STOP_AFTER = k         every workgroup of walk_sets_kernel / walk_rows_kernel ends at stamp k (tools/walk_insts.sh)
//   SJ_EXPERIMENT (sjoin.hip) 1 no search | 4 no row loads | 5 ends when the rows stand in LDS | 6 ends at entry
//                             7 spans unpacked and staged but not stored      (rounds 1-4 also had 2 / 3 / 8: store variants of the old kernel)
#pragma once
#define SG_DEV_HOOKS 1
#ifndef SG_EXPERIMENT
#define SG_EXPERIMENT 0
#endif
#ifndef SJ_EXPERIMENT
#define SJ_EXPERIMENT 0
#endif

// ---- walk.hip / walk_rows.hip
#if SG_EXPERIMENT == 7
#define SG_HOOK_KERNEL_ENTRY() unsigned long long t_prev__ = __builtin_readcyclecounter()
#define SG_HOOK_STAMP(k)                                                                        \
    do {                                                                                        \
        if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {   /* 1 workgroup in 64: the atomics stay uncontended */ \
            const unsigned long long now__ = __builtin_readcyclecounter();                      \
            atomicAdd((unsigned long long *)(a.flags + 8) + (k), now__ - t_prev__);             \
            t_prev__ = now__;                                                                   \
        }                                                                                       \
    } while (0)
#elif defined(SG_STOP_AFTER)
#define SG_HOOK_KERNEL_ENTRY()
#define SG_HOOK_STAMP(k)                 \
    do {                                 \
        if (SG_STOP_AFTER == (k)) {      \
            if (threadIdx.x == 0 && !a.tags_only) {   /* a well-formed one-member row, so that the rest of the step stays in bounds (subgacc_walk_tags passes no rows at all) */ \
                a.nsize[i] = 1;          \
                a.set_ids[i * (int64_t)a.pitch] = root; \
                if (SPG) a.set_slot[i * (int64_t)a.pitch] = 0; \
                else a.set_keys[i * (int64_t)a.pitch] = 1ull << (a.m * a.shift); \
            }                            \
            return;                      \
        }                                \
    } while (0)
#else
#define SG_HOOK_KERNEL_ENTRY()
#define SG_HOOK_STAMP(k)
#endif

#if defined(SG_STOP_AFTER)
#define SG_HOOK_RSTAMP(k)                                                                                    \
    do {                                                                                                     \
        if (SG_STOP_AFTER == (k)) {                                                                          \
            if (threadIdx.x == 0 && !a.tags_only) {   /* (subgacc_walk_tags: there are no rows to write) */   \
                a.nsize[i] = 1;                                                                              \
                a.set_ids[i * (int64_t)a.pitch] = root;                                                     \
                a.set_slot[i * (int64_t)a.pitch] = 0;                                                       \
            }                                                                                                \
            return;                                                                                          \
        }                                                                                                    \
    } while (0)
#else
#define SG_HOOK_RSTAMP(k)
#endif

#if SG_EXPERIMENT == 2      // dedup only: no graph reads after the first hop
#define SG_HOOK_BEFORE_HOP(cur, w, s)                                                                                  \
    {                                                                                                                  \
        cur = (int32_t)(((uint32_t)cur * 2654435761u + (uint32_t)(w) * 40503u + (uint32_t)(s)) % 2900000u);           \
        goto visit;                                                                                                    \
    }
#define SG_HOOK_VISIT_LABEL visit:
#else
#define SG_HOOK_BEFORE_HOP(cur, w, s)
#define SG_HOOK_VISIT_LABEL
#endif

#if SG_EXPERIMENT == 6      // upper bound of any row-pointer optimisation: no indptr read at all
#define SG_HOOK_LOAD_ROW(I64, indptr, cur, b, d) (b = ((int64_t)(uint32_t)(cur) * 21) % 62000000, d = 20)
#else
#define SG_HOOK_LOAD_ROW(I64, indptr, cur, b, d) load_row<I64>(indptr, cur, b, d)
#endif

// walk_rows.hip: what the random line fetches of the later hops cost the fused kernel -- the hop's read is folded into a 256 KB
// window of the array (L2 resident: no missed line), 9 = the last hop only, 10 = every later hop.  The ceiling of what ANY
// regrouping of the walkers (transit-parallel hops, tools/transit_probe.hip) could save.
#if SG_EXPERIMENT == 9
#define SG_HOOK_HOP_AT(at, last_hop) do { if (last_hop) at &= 0xFFFF; } while (0)
#elif SG_EXPERIMENT == 10
#define SG_HOOK_HOP_AT(at, last_hop) at &= ((last_hop) ? 0xFFFF : 0x7FFF)
#else
#define SG_HOOK_HOP_AT(at, last_hop)
#endif

#if SG_EXPERIMENT == 1      // traversal only: no dedup
#define SG_HOOK_BEFORE_VISIT(cur, pk)               \
    {                                               \
        if ((cur) == -7) atomicAdd(&(pk)[0], 1ull); \
        continue;                                   \
    }
#else
#define SG_HOOK_BEFORE_VISIT(cur, pk)
#endif

#if SG_EXPERIMENT == 8      // no registration in the HBM table
#define SG_HOOK_FLUSH_SLOT(s2, real) (s2)
#else
#define SG_HOOK_FLUSH_SLOT(s2, real) (real)
#endif

// ---- sjoin.hip
#if SJ_EXPERIMENT == 1
#define SJ_HOOK_SEARCH_RANGE(lo, hi) hi = 0
#else
#define SJ_HOOK_SEARCH_RANGE(lo, hi)
#endif
#if SJ_EXPERIMENT == 4
#define SJ_HOOK_ROW_LOAD(ids, val, row, r, mul)          \
    {                                                    \
        ids[r] = (int32_t)((row) & 1023) + (mul) * (r);  \
        val[r] = (decltype(val[0] + 0))((r) & 127);         \
        continue;                                        \
    }
#define SJ_HOOK_FIRST_TRIP(id, k, t) id = (t) * 3; k = (t) & 127; if (false)
#else
#define SJ_HOOK_ROW_LOAD(ids, val, row, r, mul)
#define SJ_HOOK_FIRST_TRIP(id, k, t)
#endif
// sjoin_pair_kernel: 5 = every workgroup ends once its two rows stand in LDS (the dependent chain own -> row length -> rows alone),
// 6 = ends at entry (what starting the workgroups costs), 7 = the staged spans are unpacked and staged but not stored
#if SJ_EXPERIMENT == 6
#define SJ_HOOK_PAIR_ENTRY() if (a.S >= 0) return
#else
#define SJ_HOOK_PAIR_ENTRY()
#endif
#if SJ_EXPERIMENT == 5
#define SJ_HOOK_PAIR_ROWS_READY() if (a.S >= 0) return
#else
#define SJ_HOOK_PAIR_ROWS_READY()
#endif
#if SJ_EXPERIMENT == 7
#define SJ_HOOK_SPAN_STORES(nbody) ((nbody) == -12345 ? 1 : 0)
#else
#define SJ_HOOK_SPAN_STORES(nbody) (nbody)
#endif
