// stream_probe.hip -- dev-only probe (not product, not a test): what does the MI355X memory system deliver for the
// traffic mix of the join (sjoin_pair_kernel: ~0.4 GB of rows read, ~1.6 GB of feature rows written per launch)?
//
//     hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/build/stream_probe && tools/build/stream_probe
//
// Every kernel moves 16 bytes per lane and access, consecutive lanes on consecutive words:
//   write      W GB of non-temporal 16-byte stores only                       (the memset yardstick)
//   copy       R = W: one load per store                                       (the copy yardstick)
//   mix1:4     one 16-byte load per FOUR 16-byte stores (the join's ratio), stores carry a value derived from the load
//   mix1:4x2   the same with the output split into two streams far apart (a pair's u-block and v-block)
// Workgroups of 128 lanes, each taking whole 64 KB tiles of the output in a grid-stride loop; sizes as the cit2 batch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e__ = (x);                                                      \
        if (e__ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kLanes = 128;
constexpr int64_t kTile = 4096;   // 16-byte words of output per tile (64 KB)

// RATIO = stores per load (0: no loads); SPLIT: the second half of every tile goes to out + half
template <int RATIO, bool SPLIT, bool NT>
__global__ __launch_bounds__(kLanes) void stream_kernel(const v4f *__restrict__ in, v4f *__restrict__ out, int64_t tiles,
                                                        int64_t half) {
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t o0 = t * kTile;
        for (int64_t x = threadIdx.x; x < kTile; x += kLanes * (RATIO > 0 ? RATIO : 1)) {
            v4f v = {1.f, 2.f, 3.f, 4.f};
            if (RATIO > 0) v = __builtin_nontemporal_load(&in[o0 / RATIO + (x / (kLanes * RATIO)) * kLanes + threadIdx.x]);
#pragma unroll
            for (int r = 0; r < (RATIO > 0 ? RATIO : 1); ++r) {
                const int64_t w = o0 + x + (int64_t)r * kLanes;
                v4f *dst = SPLIT && (x + r * kLanes) >= kTile / 2 ? out + half + w : out + w;
                v.x += 1.f;
                if (NT) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
        }
    }
}

template <int RATIO, bool SPLIT, bool NT>
static void run(const char *name, const v4f *in, v4f *out, int64_t words_out, int grid) {
    const int64_t tiles = words_out / kTile;
    const int64_t half = SPLIT ? words_out : 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream_kernel<RATIO, SPLIT, NT>), dim3(grid), dim3(kLanes), 0, 0, in, out, tiles, half);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<RATIO, SPLIT, NT>), dim3(grid), dim3(kLanes), 0, 0, in, out, tiles, half);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double wb = (double)tiles * kTile * 16, rb = RATIO > 0 ? wb / RATIO : 0;
    printf("%s,%d,%.3f,%.3f,%.4f,%.2f\n", name, grid, rb / 1e9, wb / 1e9, ms, (rb + wb) / ms / 1e9);
}

int main() {
    const int64_t words_out = (int64_t)1600 * 1000 * 1000 / 16 / kTile * kTile;     // ~1.6 GB written
    v4f *in, *out;
    CK(hipMalloc(&in, words_out * 16));
    CK(hipMalloc(&out, words_out * 16 * 2 + 64));
    CK(hipMemset(in, 0, words_out * 16));
    printf("shape,grid,read_GB,written_GB,ms,TB_per_s\n");
    for (int grid : {256 * 8, 256 * 16, 256 * 64}) {
        run<0, false, true>("write_nt", in, out, words_out, grid);
        run<0, false, false>("write_plain", in, out, words_out, grid);
        run<1, false, true>("copy_nt", in, out, words_out, grid);
        run<4, false, true>("mix1:4_nt", in, out, words_out, grid);
        run<4, false, false>("mix1:4_plain", in, out, words_out, grid);
        run<4, true, true>("mix1:4x2_nt", in, out, words_out, grid);
    }
    return 0;
}
