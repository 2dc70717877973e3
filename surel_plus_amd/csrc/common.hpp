// common.hpp -- shared helpers of libsubgacc_hip.so (gfx950 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/subgacc.h"

// Hook points of the dev-only timing experiments.  The experiments themselves (kernels that skip a phase and therefore give
// WRONG results: "traversal only", "no stores", "stop after phase k", cycle stamps) live in tools/dev_hooks.hpp, which the
// tools/*.sh scripts force-include (-include) into experiment builds under /tmp.  The product build never sees that file:
// every hook below expands to nothing (or to the real expression), so no wrong-result code is compiled into libsubgacc_hip.so.
#ifndef SG_DEV_HOOKS
#define SG_HOOK_KERNEL_ENTRY()
#define SG_HOOK_STAMP(k)
#define SG_HOOK_RSTAMP(k)
#define SG_HOOK_BEFORE_HOP(cur, w, s)
#define SG_HOOK_LOAD_ROW(I64, indptr, cur, b, d) load_row<I64>(indptr, cur, b, d)
#define SG_HOOK_VISIT_LABEL
#define SG_HOOK_HOP_AT(at, last_hop)
#define SG_HOOK_BEFORE_VISIT(cur, pk)
#define SG_HOOK_FLUSH_SLOT(s2, real) (real)
#define SJ_HOOK_SEARCH_RANGE(lo, hi)
#define SJ_HOOK_ROW_LOAD(ids, val, row, r, mul)
#define SJ_HOOK_PAIR_ENTRY()
#define SJ_HOOK_FIRST_TRIP(id, k, t)
#define SJ_HOOK_PAIR_ROWS_READY()
#define SJ_HOOK_SPAN_STORES(nbody) (nbody)
#endif

namespace subgacc {

constexpr int kWave = 64;            // CDNA4 wavefront
constexpr int kXcds = 8;             // MI355X: 8 XCDs, blocks are dealt round-robin over them
constexpr int kLdsBytes = 160 * 1024; // LDS per CU

void set_error(const char *fmt, ...);

#define SG_CHECK_HIP(expr)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            ::subgacc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return SUBGACC_ERR_HIP;                                                                \
        }                                                                                          \
    } while (0)

#define SG_REQUIRE(cond, status, ...)                                                              \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::subgacc::set_error(__VA_ARGS__);                                                     \
            return (status);                                                                       \
        }                                                                                          \
    } while (0)

#define SG_LAUNCH_CHECK() SG_CHECK_HIP(hipGetLastError())

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// XCD-aware block -> work-item map: blocks b and b+8 share an XCD (and its 4 MiB L2), so give each XCD
// one contiguous range of items; neighbouring roots / segments then hit the same L2.  Launch
// xcd_grid(n) blocks and skip when the mapped index is >= n.  Speed only, never correctness.
static inline int64_t xcd_grid(int64_t n) { return ceil_div(n, kXcds) * kXcds; }
__device__ __forceinline__ int64_t xcd_item(int64_t block, int64_t grid) {
    const int64_t per = grid / kXcds;
    return (block % kXcds) * per + block / kXcds;
}

// scan.hip
size_t scan_workspace_bytes(int64_t n);
int exclusive_scan_i32(const int32_t *in, int64_t n, int64_t *out, void *ws, size_t ws_bytes, hipStream_t s);
int exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, void *ws, size_t ws_bytes, hipStream_t s);

}  // namespace subgacc
