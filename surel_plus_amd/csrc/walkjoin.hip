// walkjoin.hip -- walk_join of the legacy SUREL surface (subg_acc/subg_acc.c:509-647) for gfx950.
//
// The reference builds a hash of hashes (root -> {node -> running index}) and, for every query pair (a, b) and every
// position t of the raw walks of a and of b, looks the visited node up in BOTH roots' tables (find_idx, :78-92:
// running index, 0 when absent).  Here the per-root tables are SpG-form rows (node ids sorted, payload = running
// index; built by subgacc_spg_build), one 256-lane workgroup takes one query pair, stages both rows in LDS and
// answers the 4 * stride look-ups by binary search; every lane writes its (own-table, partner-table) pair as one
// 8-byte store, consecutive lanes on consecutive words.
#include "common.hpp"

namespace subgacc {

constexpr int kWjThreads = 256;

__device__ __forceinline__ int32_t wj_find(const int32_t *ids, const int32_t *idx, int32_t len, int32_t node) {
    int32_t lo = 0, hi = len;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (ids[mid] < node) lo = mid + 1;
        else hi = mid;
    }
    return (lo < len && ids[lo] == node) ? idx[lo] : 0;
}

template <bool LDS>
__global__ __launch_bounds__(kWjThreads) void walk_join_kernel(const int32_t *__restrict__ walks, int32_t stride,
                                                               const int64_t *__restrict__ set_off,
                                                               const int32_t *__restrict__ set_ids,
                                                               const int32_t *__restrict__ set_idx,
                                                               const int32_t *__restrict__ qrow, int64_t Q,
                                                               int32_t *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int64_t x = blockIdx.x;
    const int32_t ra = qrow[2 * x], rb = qrow[2 * x + 1];
    const int64_t width = Q * 2 * (int64_t)stride;
    int2 *o0 = (int2 *)(out + 2 * x * (int64_t)stride);             // row of key1's walk
    int2 *o1 = (int2 *)(out + width + 2 * x * (int64_t)stride);     // row of key2's walk
    if (ra < 0 || rb < 0) {      // a query key that is no root (the reference reads out of bounds there)
        for (int t = threadIdx.x; t < stride; t += kWjThreads) o0[t] = o1[t] = make_int2(-1, -1);
        return;
    }
    const int64_t ba = set_off[ra], bb = set_off[rb];
    const int32_t la = (int32_t)(set_off[ra + 1] - ba), lb = (int32_t)(set_off[rb + 1] - bb);
    const int32_t *ia = set_ids + ba, *xa = set_idx + ba, *ib = set_ids + bb, *xb = set_idx + bb;
    if (LDS) {
        int32_t *s = (int32_t *)lds_raw;
        for (int e = threadIdx.x; e < la; e += kWjThreads) s[e] = ia[e], s[la + e] = xa[e];
        for (int e = threadIdx.x; e < lb; e += kWjThreads) s[2 * la + e] = ib[e], s[2 * la + lb + e] = xb[e];
        __syncthreads();
        ia = s, xa = s + la, ib = s + 2 * la, xb = s + 2 * la + lb;
    }
    const int32_t *wa = walks + (int64_t)ra * stride, *wb = walks + (int64_t)rb * stride;
    for (int t = threadIdx.x; t < stride; t += kWjThreads) {
        const int32_t na = wa[t], nb = wb[t];
        o0[t] = make_int2(wj_find(ia, xa, la, na), wj_find(ib, xb, lb, na));    // :625-626
        o1[t] = make_int2(wj_find(ia, xa, la, nb), wj_find(ib, xb, lb, nb));    // :627-628
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" int subgacc_walk_join(const int32_t *walks, int64_t n, int32_t stride, const int64_t *set_off,
                                 const int32_t *set_ids, const int32_t *set_idx, int32_t max_len, const int32_t *qrow,
                                 int64_t Q, int32_t *out, void *stream) {
    SG_REQUIRE(n >= 0 && Q >= 0 && stride > 0 && max_len >= 0, SUBGACC_ERR_BADARG, "walk_join: bad sizes");
    if (Q == 0) return SUBGACC_OK;
    SG_REQUIRE(walks && set_off && qrow && out && (set_ids || max_len == 0) && (set_idx || max_len == 0),
               SUBGACC_ERR_BADARG, "walk_join: null argument");
    SG_REQUIRE(Q < (1ll << 31), SUBGACC_ERR_BADARG, "walk_join: too many query pairs in one call");
    SG_REQUIRE(((uintptr_t)out & 7) == 0, SUBGACC_ERR_BADARG, "walk_join: out must be 8-byte aligned");
    const size_t lds = (size_t)max_len * 16;            // two rows, ids + indices
    if (lds <= (size_t)kLdsBytes / 2) {
        if (lds > 64 * 1024)
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)walk_join_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(walk_join_kernel<true>, dim3((unsigned)Q), dim3(kWjThreads), lds, (hipStream_t)stream, walks,
                           stride, set_off, set_ids, set_idx, qrow, Q, out);
    } else {                                            // rows too long for LDS: search them where they lie (L2)
        hipLaunchKernelGGL(walk_join_kernel<false>, dim3((unsigned)Q), dim3(kWjThreads), 0, (hipStream_t)stream, walks,
                           stride, set_off, set_ids, set_idx, qrow, Q, out);
    }
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
