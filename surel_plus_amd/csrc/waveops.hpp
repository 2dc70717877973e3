// waveops.hpp -- wave-wide reductions / inclusive scan over the 64 lanes (every lane active at the call sites).
//
// The fast forms are the device library's (ockl): DPP row shifts and broadcasts -- a handful of vector instructions and no
// LDS traffic, where six __shfl steps cost six ds_bpermute round trips each (walk_rows_kernel: cit2 1.055 -> 1.04 ms, collab
// 0.55 -> 0.51 ms, profiles/r03n_ab_dpp_reductions.log).  They are internals of ROCm's device library, not a documented
// interface: the Makefile probes for them at build time (a five-line kernel that must link) and defines
// SG_NO_OCKL_WAVE_OPS when a ROCm update has renamed them; the __shfl forms below then take over (same results, slower).
#pragma once
#include "common.hpp"

namespace subgacc {

#if !defined(SG_NO_OCKL_WAVE_OPS)
extern "C" __device__ __attribute__((const)) int __ockl_wfred_min_i32(int);
extern "C" __device__ __attribute__((const)) int __ockl_wfred_max_i32(int);
extern "C" __device__ __attribute__((const)) unsigned __ockl_wfred_min_u32(unsigned);
extern "C" __device__ __attribute__((const)) int __ockl_wfred_add_i32(int);
extern "C" __device__ __attribute__((const)) int __ockl_wfscan_add_i32(int, bool);
__device__ __forceinline__ int wave_red_min_i32(int v) { return __ockl_wfred_min_i32(v); }
__device__ __forceinline__ int wave_red_max_i32(int v) { return __ockl_wfred_max_i32(v); }
__device__ __forceinline__ unsigned wave_red_min_u32(unsigned v) { return __ockl_wfred_min_u32(v); }
__device__ __forceinline__ int wave_red_add_i32(int v) { return __ockl_wfred_add_i32(v); }
__device__ __forceinline__ int wave_scan_add_i32_incl(int v) { return __ockl_wfscan_add_i32(v, true); }
#else
__device__ __forceinline__ int wave_red_min_i32(int v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ int wave_red_max_i32(int v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ unsigned wave_red_min_u32(unsigned v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v = min(v, (unsigned)__shfl_xor((int)v, d, kWave));
    return v;
}
__device__ __forceinline__ int wave_red_add_i32(int v) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}
__device__ __forceinline__ int wave_scan_add_i32_incl(int v) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int t = __shfl_up(v, d, kWave);
        if (lane >= d) v += t;
    }
    return v;
}
#endif

}  // namespace subgacc
