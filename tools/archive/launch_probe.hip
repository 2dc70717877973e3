// Dev-only: what does it cost just to START the workgroups of the fused walk kernel?  131,072 workgroups of one or two
// wavefronts with ~5 KB of LDS each (the 2-hop key-rows kernel: one root per workgroup).  Variants: an empty body; a body that
// reads one uniform word (scalar) and one per-lane word; `per` roots per workgroup (a loop with a barrier) at grid / per.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_probe.hip -o tools/build/launch_probe && tools/build/launch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void probe(const uint32_t *__restrict__ in, uint32_t *out, int per, int vregs) {
    extern __shared__ uint32_t lds[];
    uint32_t acc = 0;
    for (int r = 0; r < per; ++r) {
        if (MODE >= 1) {
            lds[threadIdx.x] = in[(blockIdx.x * per + r) & 1023];                       // uniform address: a scalar read
            acc += lds[threadIdx.x ^ 1];
        }
        if (MODE >= 2) acc += in[((blockIdx.x * per + r) * 64 + threadIdx.x) & 0xFFFFF];  // a per-lane read, dependent on nothing
        if (MODE >= 3) acc += in[acc & 0xFFFFF];                                          // ... and one that depends on it
        __syncthreads();
    }
    if (acc == 0x12345u) out[0] = acc;
}
template <int MODE>
static float run(int grid, int threads, int lds, int per, const uint32_t *in, uint32_t *out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    probe<MODE><<<grid / per, threads, lds>>>(in, out, per, 0);
    (void)hipEventRecord(a);
    probe<MODE><<<grid / per, threads, lds>>>(in, out, per, 0);
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
    for (int threads : {64, 128})
        for (int lds : {0, 5120, 10240})
            for (int per : {1, 2, 4, 8}) {
                printf("threads %3d lds %5d roots/wg %d:  empty %.4f ms | scalar read %.4f | + lane read %.4f | + dependent read %.4f\n", threads, lds, per,
                       run<0>(grid, threads, lds, per, in, out), run<1>(grid, threads, lds, per, in, out), run<2>(grid, threads, lds, per, in, out),
                       run<3>(grid, threads, lds, per, in, out));
            }
    return 0;
}
