// Dev-only: what does an LDS read cost when many lanes of a wave ask for the SAME word?  (csrc/keyrows.hip: a set's members mostly
// carry the same dozen LP keys.)  256 CUs x 8 blocks x 256 lanes, every lane makes `iters` dependent ds_read_b32 of a table whose
// index comes from a per-lane pattern:  0 distinct words, conflict-free (lane i -> word i) | 1 every lane the same word |
// 2 ten hot words, lanes assigned at random | 3 64 random words of a 4,096-word table | 4 ten hot words but every hot word
// replicated 32 x (lane-private copy: replica = lane & 31) -- the fix under test.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ __launch_bounds__(256) void probe(const uint32_t *__restrict__ idx, int iters, int width, uint32_t *out) {
    __shared__ uint32_t tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) tab[i] = (uint32_t)i;          // a self-loop: the chase stays on its word, and the compiler cannot know
    __syncthreads();
    uint32_t p = idx[threadIdx.x];
    uint32_t acc = 0;
    if (width == 4) {
        for (int k = 0; k < iters; ++k) {
            const uint32_t v = tab[p];                  // dependent chase: the next address is what was read
            acc += v;
            p = v;
        }
    } else {
        const unsigned long long *t8 = (const unsigned long long *)tab;
        for (int k = 0; k < iters; ++k) {
            const unsigned long long v = t8[p >> 1];   // (even p: the low word of the pair is p itself)
            acc += (uint32_t)(v >> 32);
            p = (uint32_t)v;
        }
    }
    if (acc == 0x12345u) out[0] = acc;
}
int main() {
    const int iters = 4096, blocks = 2048;
    uint32_t *d_idx, *d_out;
    hipMalloc(&d_idx, 256 * 4);
    hipMalloc(&d_out, 4);
    const char *names[] = {"64 distinct words, conflict-free", "every lane the same word", "10 hot words, random lanes", "64 random words of 4,096",
                           "10 hot words, each replicated per lane & 31"};
    for (int width : {4, 8})
        for (int pat = 0; pat < 5; ++pat) {
            std::vector<uint32_t> h(256);
            uint32_t x = 12345u;
            for (int i = 0; i < 256; ++i) {
                x = x * 1664525u + 1013904223u;
                const uint32_t r = x >> 8;
                if (pat == 0) h[i] = (i & 63) * (width / 4);
                else if (pat == 1) h[i] = 40;
                else if (pat == 2) h[i] = ((r % 10) * 397u) & 4094u;
                else if (pat == 3) h[i] = (r & 4095u) & ~1u;
                else h[i] = ((r % 10) * 64u + (i & 31) * 2u) & 8190u;
            }
            hipMemcpy(d_idx, h.data(), 1024, hipMemcpyHostToDevice);
            hipEvent_t a, b;
            hipEventCreate(&a), hipEventCreate(&b);
            probe<<<blocks, 256>>>(d_idx, iters, width, d_out);
            hipEventRecord(a);
            probe<<<blocks, 256>>>(d_idx, iters, width, d_out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double reads = (double)blocks * 256 * iters;
            printf("ds_read_b%d  %-46s %8.3f ms  %7.1f G lane-reads/s  (%.2f wave-reads per CU-cycle at 2.4 GHz)\n", width * 8, names[pat], ms,
                   reads / ms / 1e6, reads / 64 / (ms * 1e-3) / 256 / 2.4e9);
        }
    return 0;
}
