// spg.hip -- SpG build: sort every sampled node set by node id (gfx950).
//
// Replaces the scipy COO->CSR conversion of subg_matrix (reference sampler/random_walks.py:79-80), which
// is a single-threaded global sort.  Here the sets are already grouped by row (one row per root), so the
// "conversion" is a segmented sort: one wave64 workgroup per row, the row's (id, SFptr+1) pairs packed
// into 64-bit words in LDS and sorted by a bitonic network sized to the row (next pow2 of its length),
// then written back coalesced.  Rows are dealt to XCDs in contiguous ranges.
#include "common.hpp"

namespace subgacc {

constexpr int kSpgThreads = 64;

__global__ __launch_bounds__(kSpgThreads) void spg_build_kernel(const int64_t *__restrict__ row_off, int64_t n,
                                                                 const int32_t *__restrict__ ids,
                                                                 const int32_t *__restrict__ sf, int32_t max_pow2,
                                                                 int32_t *__restrict__ out_indices,
                                                                 int32_t *__restrict__ out_data, int32_t *flags) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    unsigned long long *buf = (unsigned long long *)lds_raw;
    const int64_t i = xcd_item(blockIdx.x, gridDim.x);
    if (i >= n) return;
    const int64_t beg = row_off[i];
    const int64_t ns64 = row_off[i + 1] - beg;
    if (ns64 > max_pow2) {  // the caller under-stated max_len: refuse the row, never overrun LDS
        if (threadIdx.x == 0) atomicOr(&flags[3], 1);
        return;
    }
    const int ns = (int)ns64;
    int P = 1;
    while (P < ns) P <<= 1;
    const int tid = threadIdx.x;
    for (int r = tid; r < P; r += kSpgThreads)
        buf[r] = r < ns ? (((unsigned long long)(uint32_t)ids[beg + r] << 32) | (uint32_t)(sf[beg + r] + 1)) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += kSpgThreads) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const unsigned long long a = buf[lo], b = buf[hi];
                const bool ascending = (lo & k) == 0;
                if ((a > b) == ascending) {
                    buf[lo] = b;
                    buf[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int r = tid; r < ns; r += kSpgThreads) {
        const unsigned long long v = buf[r];
        out_indices[beg + r] = (int32_t)(v >> 32);
        out_data[beg + r] = (int32_t)(uint32_t)v;
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" int subgacc_spg_build(const int64_t *row_off, int64_t n, const int32_t *ids, const int32_t *sf,
                                 int32_t max_len, int32_t *out_indices, int32_t *out_data, int32_t *flags,
                                 void *stream) {
    SG_REQUIRE(n >= 0 && max_len >= 0 && flags, SUBGACC_ERR_BADARG, "spg_build: bad arguments");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(row_off && ids && sf && out_indices && out_data, SUBGACC_ERR_BADARG, "spg_build: null argument");
    int P = 1;
    while (P < max_len) P <<= 1;
    const size_t lds = (size_t)P * 8;
    SG_REQUIRE(lds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS, "spg_build: rows of %d members do not fit LDS", max_len);
    if (lds > 64 * 1024)
        SG_CHECK_HIP(hipFuncSetAttribute((const void *)spg_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t grid = xcd_grid(n);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "spg_build: too many rows in one call");
    hipLaunchKernelGGL(spg_build_kernel, dim3((unsigned)grid), dim3(kSpgThreads), lds, (hipStream_t)stream, row_off, n,
                       ids, sf, P, out_indices, out_data, flags);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
