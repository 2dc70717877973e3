// scan.hip -- device-wide exclusive prefix sums (int32/int64 -> int64) for row offsets.
// Used for: nsize -> SpG row offsets (the serial loop of subg_acc.c:848-851), rand_r call positions,
// SpJoin segment pointers (train.py:20-22).  Three-phase reduce / scan-partials / scan, 2048 items per
// 256-thread block, wave64 shuffles inside a wave and LDS across the 4 waves.
#include "common.hpp"
#include "blockscan.hpp"

namespace subgacc {

constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

template <typename T>
__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const T *__restrict__ in, int64_t n,
                                                                    int64_t *__restrict__ partial) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) s += (int64_t)in[base + k];
    int64_t tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// partial_excl == nullptr => single tile
template <typename T>
__global__ __launch_bounds__(kScanThreads) void scan_tile_kernel(const T *__restrict__ in, int64_t n,
                                                                  const int64_t *__restrict__ partial_excl,
                                                                  int64_t *__restrict__ out) {
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < n) ? (int64_t)in[base + k] : 0;
        s += v[k];
    }
    int64_t tot;
    int64_t run = block_exclusive_scan(s, &tot) + (partial_excl ? partial_excl[blockIdx.x] : 0);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
        if (base + k == n - 1) out[n] = run;  // the grand total lands in out[n]
    }
}

__global__ void scan_empty_kernel(int64_t *out) { out[0] = 0; }

size_t scan_workspace_bytes(int64_t n) {
    size_t bytes = 0;
    int64_t nb = ceil_div(n > 0 ? n : 1, kScanTile);
    while (nb > 1) {
        bytes += align_up((size_t)(2 * nb + 1) * sizeof(int64_t), 256);
        nb = ceil_div(nb, kScanTile);
    }
    return bytes + 256;
}

template <typename T>
static int exclusive_scan_impl(const T *in, int64_t n, int64_t *out, void *ws, size_t ws_bytes, hipStream_t s) {
    if (n <= 0) {
        hipLaunchKernelGGL(scan_empty_kernel, dim3(1), dim3(1), 0, s, out);
        SG_LAUNCH_CHECK();
        return SUBGACC_OK;
    }
    const int64_t nb = ceil_div(n, kScanTile);
    if (nb == 1) {
        hipLaunchKernelGGL(scan_tile_kernel<T>, dim3(1), dim3(kScanThreads), 0, s, in, n, (const int64_t *)nullptr, out);
        SG_LAUNCH_CHECK();
        return SUBGACC_OK;
    }
    const size_t need = align_up((size_t)(2 * nb + 1) * sizeof(int64_t), 256);
    SG_REQUIRE(ws && ws_bytes >= need, SUBGACC_ERR_WORKSPACE, "scan: workspace %zu < %zu bytes", ws_bytes, need);
    int64_t *partial = (int64_t *)ws;
    int64_t *partial_excl = partial + nb;  // nb+1 entries
    hipLaunchKernelGGL(scan_reduce_kernel<T>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, in, n, partial);
    SG_LAUNCH_CHECK();
    int rc = exclusive_scan_impl<int64_t>(partial, nb, partial_excl, (char *)ws + need, ws_bytes - need, s);
    if (rc != SUBGACC_OK) return rc;
    hipLaunchKernelGGL(scan_tile_kernel<T>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, in, n,
                       (const int64_t *)partial_excl, out);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

int exclusive_scan_i32(const int32_t *in, int64_t n, int64_t *out, void *ws, size_t ws_bytes, hipStream_t s) {
    return exclusive_scan_impl<int32_t>(in, n, out, ws, ws_bytes, s);
}
int exclusive_scan_i64(const int64_t *in, int64_t n, int64_t *out, void *ws, size_t ws_bytes, hipStream_t s) {
    return exclusive_scan_impl<int64_t>(in, n, out, ws, ws_bytes, s);
}

}  // namespace subgacc

extern "C" size_t subgacc_scan_workspace_bytes(int64_t n) { return subgacc::scan_workspace_bytes(n); }

extern "C" int subgacc_exclusive_scan_i32(const int32_t *in, int64_t n, int64_t *out, void *workspace,
                                          size_t workspace_bytes, void *stream) {
    SG_REQUIRE(n >= 0 && out && (in || n == 0), SUBGACC_ERR_BADARG, "exclusive_scan_i32: bad arguments");
    return subgacc::exclusive_scan_i32(in, n, out, workspace, workspace_bytes, (hipStream_t)stream);
}

// A few words of status and sizes to pinned host memory by a KERNEL: a step's read-back as hipMemcpyAsync costs the GPU ~10 us of
// idling behind its 4 us blit (the kernel trace of the headline step, profiles/r24_graph_gaps_cit2.txt); one wave storing to
// device-visible host memory does not.
__global__ void publish_words_kernel(const int64_t *__restrict__ src, int n, int64_t *__restrict__ host_dst) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) host_dst[i] = src[i];
}

extern "C" int subgacc_publish_words(const int64_t *src, int64_t n, int64_t *host_dst, void *stream) {
    SG_REQUIRE(n >= 0 && n <= 4096 && (n == 0 || (src && host_dst)), SUBGACC_ERR_BADARG, "publish_words: bad arguments");
    if (n == 0) return SUBGACC_OK;
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, (int)n, host_dst);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
