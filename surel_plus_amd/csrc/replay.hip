// replay.hip -- rand_r stream positions on a graph with dead ends (gfx950).
//
// The reference draws from ONE sequential rand_r stream per OpenMP thread (subg_acc.c:731-732 set_sampler; :157-158, :191-192
// walk_sampler) and a walk that stands on a node without out-edges simply draws nothing (:804-808, :168-172, :236-240): the
// walk stays there, and every later draw of the stream moves up.  On a symmetrised graph that never happens, the number of
// draws per root is a function of its degree, and subgacc_rng_positions places every root in the stream with one scan.  On a
// directed graph the position of a walk depends on how all earlier walks ENDED -- the walk kernels notice (flags[0] |= 1) and
// the host comes here: one WAVE per stream replays it.  Only the count of draws is sequential; what a walk visits is not
// needed here.  So 64 consecutive walks of a root are walked together on the assumption that the ones before them drew
// their full share; the first lane whose walk met a dead end decides how far the assumption held, the positions up to and
// including that walk are final, and the round starts again behind it.  Output: the stream position of every root (its
// shuffle draws) and of every WALK; the walk kernel (walk_sets_kernel, which takes `walk_pos`) then samples the sets in
// parallel as usual, every walk entering the stream exactly where the reference's sequential loop has it.
//
// COST (bounded by the definition, not by the implementation): set_sampler has ONE stream (rng_streams = 1), so one wavefront
// replays all n*M walks, m dependent global loads per round of up to 64 walks, and a walk that ended early restarts the round
// behind it -- ~0.1 us per walk on a graph with few dead ends, more with many: an all-N offline stage over a directed graph
// of millions of nodes takes tens of seconds with the rest of the chip idle.  The host mirror warns from 2^26 walks on
// (sampler.REPLAY_WARN_WALKS) and remembers the discovery on the DeviceCSR, so a batch is never walked twice to find out again;
// rng="philox" is the parallel answer for such graphs.  (A speculative pass per block of roots with a fix-up of the blocks behind a
// short walk would parallelise the common case; not built: the reference's own loader symmetrises every graph, dataloader.py:122-135.)
#include "walk_common.hpp"

namespace subgacc {

template <bool IDX64>
__global__ __launch_bounds__(kWave) void rand_r_replay_kernel(const void *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                              const int32_t *__restrict__ query, int64_t n, int64_t num_nodes,
                                                              int M, int m, int wo, int cap_root, int streams, uint32_t seed,
                                                              uint64_t calls_before, uint32_t *__restrict__ rng_pos,
                                                              uint32_t *__restrict__ rng_seed, uint32_t *__restrict__ walk_pos) {
    extern __shared__ int32_t sarr[];       // [M] Fisher-Yates draws of the root at hand
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;           // the stream: libgomp's static chunk t of the n roots, seed + t
    const int64_t q = n / streams, r = n % streams;
    const int64_t lo = t < r ? t * (q + 1) : r * (q + 1) + (t - r) * q;
    const int64_t hi = lo + (t < r ? q + 1 : q);
    const uint32_t sd = seed + (uint32_t)t;
    uint32_t pos = streams == 1 ? (uint32_t)(3ull * calls_before) : 0u;     // LCG steps taken so far (3 per rand_r call, mod 2^32)
    const int per_walk = wo ? m - 1 : m;
    for (int64_t i = lo; i < hi; ++i) {
        const int32_t root = query[i];
        int64_t rbeg = 0, rdeg64 = 0;
        if ((uint64_t)(int64_t)root < (uint64_t)num_nodes) load_row<IDX64>(indptr, root, rbeg, rdeg64);
        if (cap_root && rdeg64 > kNeighCap) rdeg64 = kNeighCap;
        if (lane == 0) {
            rng_pos[i] = pos;
            rng_seed[i] = sd;
        }
        if (rdeg64 == 0) {       // an isolated (or out-of-range) root draws nothing
            for (int w = lane; w < M; w += kWave) walk_pos[i * (int64_t)M + w] = pos;
            continue;
        }
        const uint32_t rdeg = (uint32_t)rdeg64;
        const bool shuffled = wo && rdeg64 > M;
        if (shuffled) {
            __syncthreads();     // (one wave: a wait for the previous root's readers of sarr)
            for (int k = lane; k < M; k += kWave) {
                uint32_t x = lcg_jump(sd, pos + 3u * (uint32_t)k);
                sarr[k] = (int32_t)(rand_r_next(x) % (rdeg - (uint32_t)k)) + k;
            }
            pos += 3u * (uint32_t)M;
            __syncthreads();
        }
        int w0 = 0;
        while (w0 < M) {
            const int w = w0 + lane;
            const bool live = w < M;
            const uint32_t start = pos + 3u * (uint32_t)per_walk * (uint32_t)lane;
            int used = per_walk;
            if (live && per_walk > 0) {
                uint32_t x = lcg_jump(sd, start);
                int32_t cur = root;
                used = 0;
                for (int s = 0; s < m; ++s) {
                    if (s == 0 && wo) {
                        uint32_t pick;
                        if (shuffled) {
                            int32_t p = sarr[w];
                            for (int j = w - 1; j >= 0; --j)
                                if (sarr[j] == p) p = j;
                            pick = (uint32_t)p;
                        } else {
                            pick = (uint32_t)w % rdeg;
                        }
                        cur = indices[rbeg + pick];
                    } else {
                        int64_t b = 0, d = 0;
                        if ((uint64_t)(int64_t)cur < (uint64_t)num_nodes) load_row<IDX64>(indptr, cur, b, d);
                        if (d <= 0) break;         // a dead end: this walk (and every later step of it) draws nothing more
                        cur = indices[b + (int64_t)(rand_r_next(x) % (uint32_t)d)];
                        ++used;
                    }
                }
            }
            const unsigned long long shortm = __ballot(live && used < per_walk);
            if (shortm == 0ull) {       // every walk of the round drew its full share: all 64 positions were right
                if (live) walk_pos[i * (int64_t)M + w] = start;
                const int done = min(kWave, M - w0);
                pos += 3u * (uint32_t)per_walk * (uint32_t)done;
                w0 += done;
            } else {                    // right up to and including the first short walk
                const int f = __ffsll((long long)shortm) - 1;
                if (lane <= f) walk_pos[i * (int64_t)M + w] = start;
                const int used_f = __shfl(used, f, kWave);
                pos += 3u * ((uint32_t)per_walk * (uint32_t)f + (uint32_t)used_f);
                w0 += f + 1;
            }
        }
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" int subgacc_rng_replay(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                                  const int32_t *query, int64_t n, int32_t rng_streams, uint64_t calls_before, uint32_t *rng_pos,
                                  uint32_t *rng_seed, uint32_t *walk_pos, void *stream) {
    SG_REQUIRE(cfg && indptr && rng_pos && rng_seed && walk_pos && n >= 0 && num_nodes >= 0, SUBGACC_ERR_BADARG, "rng_replay: null argument");
    SG_REQUIRE(cfg->rng_mode == SUBGACC_RNG_RAND_R, SUBGACC_ERR_BADARG, "rng_replay: the rand_r mode only (Philox streams have no positions)");
    SG_REQUIRE(rng_streams >= 1 && rng_streams <= (1 << 20), SUBGACC_ERR_BADARG, "rng_replay: rng_streams must be in [1, 2^20]");
    SG_REQUIRE(rng_streams == 1 || calls_before == 0, SUBGACC_ERR_BADARG, "rng_replay: calls_before only makes sense for a single stream");
    SG_REQUIRE(cfg->num_walks > 0 && cfg->num_steps > 0 && (size_t)cfg->num_walks * 4 <= 64 * 1024, SUBGACC_ERR_BADARG,
               "rng_replay: num_walks = %d", cfg->num_walks);
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(query && (indices || num_nodes == 0), SUBGACC_ERR_BADARG, "rng_replay: null query / indices");
    const size_t lds = (size_t)cfg->num_walks * 4;
    if (cfg->indptr64)
        hipLaunchKernelGGL(rand_r_replay_kernel<true>, dim3((unsigned)rng_streams), dim3(kWave), lds, (hipStream_t)stream, indptr, indices,
                           query, n, num_nodes, cfg->num_walks, cfg->num_steps, cfg->first_hop_wo, cfg->cap_root_degree, rng_streams,
                           cfg->seed, calls_before, rng_pos, rng_seed, walk_pos);
    else
        hipLaunchKernelGGL(rand_r_replay_kernel<false>, dim3((unsigned)rng_streams), dim3(kWave), lds, (hipStream_t)stream, indptr, indices,
                           query, n, num_nodes, cfg->num_walks, cfg->num_steps, cfg->first_hop_wo, cfg->cap_root_degree, rng_streams,
                           cfg->seed, calls_before, rng_pos, rng_seed, walk_pos);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
