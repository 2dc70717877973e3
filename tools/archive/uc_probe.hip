// uc_probe.hip -- dev-only: does a random 4-byte read cost less than a 128-byte line when the table is NOT cached in L2?
// The walk kernel is bound by ~55 G random lines/s (tools/line_probe.hip); every load flavour tried so far (nt, sc1, sc0 sc1)
// moves a whole line per miss.  What is left is the memory type of the allocation itself: uncached (hipDeviceMallocUncached) and
// fine-grained (hipDeviceMallocFinegrained) device memory, whose accesses bypass L2 allocation.  Same gather4 shape, 2048 x 256 lanes,
// 4 loads in flight per lane, a 252 MB and a 1 GB table.   hipcc -O3 --offload-arch=gfx950 tools/uc_probe.hip -o tools/build/uc_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(1); } } while (0)
__device__ __forceinline__ uint64_t pos64(uint32_t a, uint32_t b, uint64_t n) {
    uint64_t h = ((uint64_t)a << 32 | b) + 0x9e3779b97f4a7c15ull;
    h = (h ^ (h >> 30)) * 0xbf58476d1ce4e5b9ull;
    h = (h ^ (h >> 27)) * 0x94d049bb133111ebull;
    h ^= h >> 31;
    return (uint64_t)(((unsigned __int128)h * n) >> 64);
}
__global__ void fill(uint32_t *t, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) t[i] = (uint32_t)i * 2654435761u;
}
template <int FL>
__global__ __launch_bounds__(256) void gather4(const uint32_t *__restrict__ t, uint64_t n, int iters, uint32_t *out) {
    extern __shared__ uint32_t pad[];
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t *p = t + pos64(gid, it * 4 + k, n);
            if (FL == 1) v[k] = __builtin_nontemporal_load(p);
            else if (FL == 2) v[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else v[k] = *p;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= v[k];
    }
    if (acc == 0x12345678u) out[gid] = acc + pad[threadIdx.x];
}
template <typename F>
static double best_ms(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    double best = 1e30;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r && ms < best) best = ms;
    }
    return best;
}
int main() {
    uint32_t *out;
    CK(hipMalloc(&out, 2048 * 256 * 4));
    printf("allocation,table_MB,load,reads,ms,Greads_per_s\n");
    for (uint64_t mb : {252ull, 1024ull}) {
        const uint64_t n = mb * 1024 * 1024 / 4;
        for (int kind = 0; kind < 3; ++kind) {
            uint32_t *t = nullptr;
            hipError_t e = kind == 0 ? hipMalloc(&t, n * 4)
                         : hipExtMallocWithFlags((void **)&t, n * 4, kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
            const char *nm = kind == 0 ? "hipMalloc" : (kind == 1 ? "hipDeviceMallocUncached" : "hipDeviceMallocFinegrained");
            if (e != hipSuccess) { printf("%s,%llu,,,,%s\n", nm, (unsigned long long)mb, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            fill<<<4096, 256>>>(t, n);
            CK(hipDeviceSynchronize());
            const int iters = 32;
            const uint64_t reads = 2048ull * 256 * 4 * iters;
            double ms;
            ms = best_ms([&] { gather4<0><<<2048, 256, 16384>>>(t, n, iters, out); });
            printf("%s,%llu,plain,%llu,%.4f,%.2f\n", nm, (unsigned long long)mb, (unsigned long long)reads, ms, reads / ms / 1e6);
            ms = best_ms([&] { gather4<1><<<2048, 256, 16384>>>(t, n, iters, out); });
            printf("%s,%llu,nontemporal,%llu,%.4f,%.2f\n", nm, (unsigned long long)mb, (unsigned long long)reads, ms, reads / ms / 1e6);
            ms = best_ms([&] { gather4<2><<<2048, 256, 16384>>>(t, n, iters, out); });
            printf("%s,%llu,system-scope atomic load,%llu,%.4f,%.2f\n", nm, (unsigned long long)mb, (unsigned long long)reads, ms, reads / ms / 1e6);
            fflush(stdout);
            CK(hipFree(t));
        }
    }
    return 0;
}
