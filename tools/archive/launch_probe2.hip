// Dev-only: does it pay to keep a wavefront for several roots instead of starting a fresh workgroup per root?  A stand-in for the
// 2-hop fused walk kernel: per "root" a chain of `chain` dependent random reads, `alu` x 64 vector instructions and a barrier,
// one wavefront per workgroup with 5 KB of LDS.  131,072 roots in all, `per` of them per workgroup (grid = 131,072 / per).
//   hipcc --offload-arch=gfx950 -O3 tools/launch_probe2.hip -o tools/build/launch_probe2 && tools/build/launch_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void probe(const uint32_t *__restrict__ in, uint32_t *out, int per, int chain, int alu) {
    extern __shared__ uint32_t lds[];
    uint32_t acc = threadIdx.x * 2654435761u;
    for (int r = 0; r < per; ++r) {
        uint32_t p = ((blockIdx.x * per + r) * 64u + threadIdx.x) * 2246822519u;
        for (int c = 0; c < chain; ++c) p = in[(p ^ acc) & 0xFFFFFu] + p * 3u;
        for (int k = 0; k < alu; ++k) {
#pragma unroll
            for (int u = 0; u < 64; ++u) acc = acc * 1664525u + p;
        }
        lds[threadIdx.x] = acc;
        __syncthreads();
        acc += lds[threadIdx.x ^ 1];
    }
    if (acc == 0x12345u) out[0] = acc;
}
static float run(int grid, int lds, int per, int chain, int alu, const uint32_t *in, uint32_t *out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    probe<<<grid / per, 64, lds>>>(in, out, per, chain, alu);
    (void)hipEventRecord(a);
    probe<<<grid / per, 64, lds>>>(in, out, per, chain, alu);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    uint32_t *in, *out;
    (void)hipMalloc(&in, 4 << 20), (void)hipMalloc(&out, 4);
    (void)hipMemset(in, 1, 4 << 20);
    const int grid = 131072;
    for (int chain : {3, 6})
        for (int alu : {4, 12}) {
            printf("chain %d reads, %4d vector instructions per root:", chain, alu * 128);
            for (int per : {1, 2, 4, 8, 16}) printf("  per=%d %.4f ms", per, run(grid, 5120, per, chain, alu, in, out));
            printf("\n");
        }
    return 0;
}
