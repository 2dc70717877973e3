// line_roof_lib.hip -- DEV / BENCH ONLY, not part of libsubgacc_hip.so nor of include/subgacc.h (it left the product ABI in round 4):
// built into tools/build/libsubgacc_probe.so by __graft_entry__.build() and loaded by bench.py alone.
// A measurement aid, not part of the reference's surface: the random-line rate of the memory system, measured
// in the same process and on the same table as the walk kernel it is the roof of (bench.py reports it beside the kernel's
// own line rate).  tools/line_probe.hip is the stand-alone study (access shapes, PMC passes: profiles/r02_line_probe_pmc.csv);
// this is its `gather4` shape -- independent random 4-byte reads, 2048 x 256 lanes (the walk kernel's residency), four in
// flight per lane -- over a caller's table.  Every read beyond the caches moves one 128-byte line.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace subgacc {

__global__ __launch_bounds__(256) void line_probe_kernel(const uint32_t *__restrict__ table, uint64_t words, int32_t rounds,
                                                         uint32_t seed, uint32_t *__restrict__ sink) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint64_t i0, i1, i2, i3;
        x = x * 1664525u + 1013904223u; i0 = ((uint64_t)x * words) >> 32;     // (r * n) >> 32: uniform below `words`
        x = x * 1664525u + 1013904223u; i1 = ((uint64_t)x * words) >> 32;
        x = x * 1664525u + 1013904223u; i2 = ((uint64_t)x * words) >> 32;
        x = x * 1664525u + 1013904223u; i3 = ((uint64_t)x * words) >> 32;
        const uint32_t a = table[i0], b = table[i1], c = table[i2], d = table[i3];
        acc += a ^ b ^ c ^ d;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;      // keeps the loads alive; practically never taken
}

}  // namespace subgacc

using namespace subgacc;

// returns 0, or -1 for bad arguments / a failed launch
extern "C" int subgacc_line_probe(const void *table, int64_t table_bytes, int32_t rounds, uint32_t seed, void *sink,
                                  int64_t *reads_out_host, void *stream) {
    if (!(table && sink && table_bytes >= 4 && table_bytes < (1ll << 34) * 4 && rounds > 0)) return -1;
    const unsigned blocks = 2048;
    hipLaunchKernelGGL(line_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)table,
                       (uint64_t)(table_bytes / 4), rounds, seed, (uint32_t *)sink);
    if (hipGetLastError() != hipSuccess) return -1;
    if (reads_out_host) *reads_out_host = (int64_t)blocks * 256 * 4 * rounds;
    return 0;
}
