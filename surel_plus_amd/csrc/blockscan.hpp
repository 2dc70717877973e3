// blockscan.hpp -- 256-thread block exclusive scan: wave64 shuffles inside a wave, LDS across the 4 waves.
#pragma once
#include "common.hpp"

namespace subgacc {

constexpr int kScanThreads = 256;

template <typename V>
__device__ __forceinline__ V wave_inclusive_scan(V v) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        V t = __shfl_up(v, d, kWave);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns the exclusive prefix and
// leaves the block total in *total (valid for every thread).  Safe to call repeatedly.
template <typename V>
__device__ __forceinline__ V block_exclusive_scan(V v, V *total) {
    __shared__ V wave_sums[kScanThreads / kWave];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    V inc = wave_inclusive_scan(v);
    if (lane == kWave - 1) wave_sums[wid] = inc;
    __syncthreads();
    V base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / kWave; ++w) {
        V s = wave_sums[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

}  // namespace subgacc
