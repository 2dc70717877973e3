// batch.hip -- batch_sampler of the legacy SUREL surface (reference subg_acc/subg_acc.c:391-507), gfx950.
//
// The reference forms a mini-batch of nodes: the roots are walked ONE AFTER THE OTHER with one rand_r stream, every
// visited node goes into one insertion-ordered set, and a root stops walking as soon as the set holds (i+1)*thld/n nodes
// (tested after every walk).  The loop over roots is sequential by construction -- where root i+1's draws start depends
// on the walk root i stopped at -- so one workgroup walks the roots in order; what is parallel is everything inside a
// root: one lane per walk (the affine rand_r stream is entered at each walk's position, as in walk.hip), the walk's new
// nodes go into an LDS table with their first-visit number, the stopping walk is the first whose prefix count of first
// visits reaches the threshold, and the first visits up to it are appended in visit order (bitmap rank) and committed to
// the HBM set.  Bit-identical to the reference given its effective seed (seed + getpid(), :421).
// A reached node without out-edges draws nothing in the reference (:466-471), which makes the stream position of every
// later walk data dependent: flags[0] |= 1 (the host raises), like the rand_r mode of the set sampler.
#include "walk_common.hpp"

namespace subgacc {

constexpr int kBatchThreads = 256;

struct BatchArgs {
    const void *indptr;
    const int32_t *indices;
    const int32_t *query;
    int64_t n, num_nodes;
    int32_t M, S, thld;
    uint32_t seed;
    int32_t *out;
    int64_t out_cap;
    int64_t *out_count;
    int32_t *gkeys;      // [gcap] committed members (open addressing, -1 = empty)
    uint32_t gmask;
    int32_t *flags;
    int32_t TL, tlshift;  // local table size (pow2) and 32 - log2(TL)
    int32_t nwords;       // bitmap words over q in [0, M*S]
};

__device__ __forceinline__ int32_t coherent_load(const int32_t *p) {   // other lanes of this workgroup have just CAS-ed it
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool IDX64>
__global__ __launch_bounds__(kBatchThreads) void batch_sampler_kernel(const BatchArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    int32_t *lkeys = (int32_t *)lds_raw;                 // [TL] new nodes of the current root
    uint32_t *lminq = (uint32_t *)(lkeys + a.TL);         // [TL] their first visit number
    uint32_t *bitmap = lminq + a.TL;                      // [nwords]
    uint32_t *prefix = bitmap + a.nwords;                 // [nwords + 1]
    int32_t *sarr = (int32_t *)(prefix + a.nwords + 1);   // [M] Fisher-Yates draws
    int32_t *vis = sarr + a.M;                            // [M*S] nodes of the walks
    int32_t *red = vis + (int64_t)a.M * a.S;              // [4]

    const int tid = threadIdx.x;
    const int M = a.M, S = a.S, TL = a.TL;
    const uint32_t lmask = (uint32_t)TL - 1u;
    for (int h = tid; h < TL; h += kBatchThreads) {
        lkeys[h] = -1;
        lminq[h] = 0xFFFFFFFFu;
    }
    for (int x = tid; x < a.nwords; x += kBatchThreads) bitmap[x] = 0u;
    __syncthreads();

    int64_t base = 0;        // members so far
    uint32_t pos = 0;        // rand_r calls so far (three LCG steps each; the LCG has period 2^32)
    bool overflow = false;
    for (int64_t i = 0; i < a.n; ++i) {
        const int32_t root = a.query[i];
        if ((uint64_t)(int64_t)root >= (uint64_t)a.num_nodes) {   // the reference reads out of bounds here
            if (tid == 0) atomicOr(&a.flags[3], 16);
            continue;
        }
        int64_t rbeg, rdeg64;
        load_row<IDX64>(a.indptr, root, rbeg, rdeg64);
        const uint32_t rdeg = (uint32_t)rdeg64;
        const bool shuffled = rdeg64 > M;
        if (shuffled)
            for (int k = tid; k < M; k += kBatchThreads) {
                uint32_t x = lcg_jump(a.seed, 3u * (pos + (uint32_t)k));
                sarr[k] = (int32_t)(rand_r_next(x) % (rdeg - (uint32_t)k)) + k;
            }
        const uint32_t pos_w = pos + (shuffled ? (uint32_t)M : 0u);
        if (tid == 0) {      // the root itself (add_item before the walks, :448): visit 0 unless it is a member already
            red[0] = 0x7FFFFFFF;
            uint32_t g = ((uint32_t)root * 2654435761u) & a.gmask;
            int32_t cur;
            while ((cur = coherent_load(&a.gkeys[g])) != -1 && cur != root) g = (g + 1u) & a.gmask;
            if (cur == -1) {
                const uint32_t h = ((uint32_t)root * 2654435761u) >> a.tlshift;
                lkeys[h] = root;       // the local table is clean: no collision possible
                lminq[h] = 0u;
            }
        }
        __syncthreads();
        if (rdeg64 >= 1) {
            for (int w = tid; w < M; w += kBatchThreads) {
                uint32_t pick;
                if (shuffled) {   // value the sequential swaps leave at position w (:437-446)
                    int32_t p = sarr[w];
                    for (int j = w - 1; j >= 0; --j)
                        if (sarr[j] == p) p = j;
                    pick = (uint32_t)p;
                } else {
                    pick = (uint32_t)w % rdeg;
                }
                int32_t cur = a.indices[rbeg + pick];
                vis[w * S] = cur;
                uint32_t x = lcg_jump(a.seed, 3u * (pos_w + (uint32_t)w * (uint32_t)(S - 1)));
                for (int s = 1; s < S; ++s) {
                    int64_t b, d;
                    load_row<IDX64>(a.indptr, cur, b, d);
                    if (d > 0) cur = a.indices[b + (int64_t)(rand_r_next(x) % (uint32_t)d)];
                    else atomicOr(&a.flags[0], 1);      // dead end: the sequential stream is not reproducible
                    vis[w * S + s] = cur;
                }
                for (int s = 0; s < S; ++s) {           // visits: members stay out, new nodes get their first visit number
                    const int32_t node = vis[w * S + s];
                    uint32_t g = ((uint32_t)node * 2654435761u) & a.gmask;
                    int32_t cur2;
                    while ((cur2 = coherent_load(&a.gkeys[g])) != -1 && cur2 != node) g = (g + 1u) & a.gmask;
                    if (cur2 == node) continue;
                    uint32_t h = ((uint32_t)node * 2654435761u) >> a.tlshift;
                    while (true) {
                        const int32_t old = atomicCAS(&lkeys[h], -1, node);
                        if (old == -1 || old == node) break;
                        h = (h + 1u) & lmask;
                    }
                    atomicMin(&lminq[h], (uint32_t)(w * S + s + 1));
                }
            }
        }
        __syncthreads();
        // first visits in visit order: bitmap over the visit numbers, popcount prefix
        for (int h = tid; h < TL; h += kBatchThreads)
            if (lkeys[h] != -1) {
                const uint32_t q = lminq[h];
                atomicOr(&bitmap[q >> 5], 1u << (q & 31u));
            }
        __syncthreads();
        for (int x = tid; x <= a.nwords; x += kBatchThreads) {
            uint32_t s = 0;
            for (int j = 0; j < x; ++j) s += __popc(bitmap[j]);
            prefix[x] = s;
        }
        __syncthreads();
        // the walk the reference stops after: the first w with  members + first visits up to its last step >= (i+1)*thld/n
        const int64_t thr = (i + 1) * (int64_t)a.thld / a.n;
        if (rdeg64 >= 1)
            for (int w = tid; w < M; w += kBatchThreads) {
                const uint32_t q = (uint32_t)((w + 1) * S);                       // last visit number of walk w
                const uint32_t upto = prefix[q >> 5] + __popc(bitmap[q >> 5] & (0xFFFFFFFFu >> (31u - (q & 31u))));
                if (base + (int64_t)upto >= thr) atomicMin(&red[0], w);
            }
        __syncthreads();
        const int wb = rdeg64 >= 1 ? (red[0] < M ? red[0] : M - 1) : -1;
        const uint32_t qmax = (uint32_t)((wb + 1) * S);
        const uint32_t added = prefix[qmax >> 5] + __popc(bitmap[qmax >> 5] & (0xFFFFFFFFu >> (31u - (qmax & 31u))));
        if (base + (int64_t)added > a.out_cap) overflow = true;
        for (int h = tid; h < TL; h += kBatchThreads) {
            const int32_t node = lkeys[h];
            if (node == -1) continue;
            const uint32_t q = lminq[h];
            if (q <= qmax && !overflow) {     // commit: position = rank of the first visit, and into the set of members
                const uint32_t r = prefix[q >> 5] + __popc(bitmap[q >> 5] & ((1u << (q & 31u)) - 1u));
                a.out[base + r] = node;
                uint32_t g = ((uint32_t)node * 2654435761u) & a.gmask;
                while (atomicCAS(&a.gkeys[g], -1, node) != -1) g = (g + 1u) & a.gmask;
            }
            lkeys[h] = -1;                    // the local table is handed back clean
            lminq[h] = 0xFFFFFFFFu;
        }
        __syncthreads();
        for (int x = tid; x < a.nwords; x += kBatchThreads) bitmap[x] = 0u;
        if (overflow) break;
        base += added;
        pos = pos_w + (uint32_t)(wb + 1) * (uint32_t)(S - 1);
        __threadfence();       // the commits are visible before the next root's look-ups
        __syncthreads();
    }
    if (tid == 0) {
        if (overflow) atomicOr(&a.flags[1], 1);
        *a.out_count = base;
    }
}

__global__ void batch_fill_kernel(int32_t *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = -1;
}

static inline int64_t batch_table_slots(int64_t out_cap) {
    int64_t c = 1024;
    while (c < 2 * (out_cap + 1)) c <<= 1;
    return c;
}

}  // namespace subgacc

using namespace subgacc;

extern "C" size_t subgacc_batch_sampler_workspace_bytes(int64_t out_cap) {
    if (out_cap < 0) out_cap = 0;
    return (size_t)batch_table_slots(out_cap) * 4;
}

extern "C" int subgacc_batch_sampler(const void *indptr, int32_t indptr64, const int32_t *indices, int64_t num_nodes,
                                     const int32_t *query, int64_t n, int32_t num_walks, int32_t num_steps, int32_t thld,
                                     uint32_t seed_eff, int32_t *out, int64_t out_cap, int64_t *out_count, void *workspace,
                                     size_t workspace_bytes, int32_t *flags, void *stream) {
    SG_REQUIRE(indptr && out && out_count && flags && n >= 0 && num_nodes >= 0 && out_cap >= 0, SUBGACC_ERR_BADARG,
               "batch_sampler: null argument or negative size");
    SG_REQUIRE(num_walks > 0 && num_steps > 0, SUBGACC_ERR_BADARG, "batch_sampler: num_walks and num_steps must be positive");
    const int64_t Q = (int64_t)num_walks * num_steps + 1;
    SG_REQUIRE(Q <= 6000, SUBGACC_ERR_LDS, "batch_sampler: num_walks*num_steps+1 = %lld needs more LDS than a workgroup has",
               (long long)Q);
    hipStream_t s = (hipStream_t)stream;
    SG_CHECK_HIP(hipMemsetAsync(out_count, 0, sizeof(int64_t), s));
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(query && num_nodes >= 1, SUBGACC_ERR_BADARG, "batch_sampler: roots without a graph");
    SG_REQUIRE(out_cap < (1ll << 30), SUBGACC_ERR_BADARG, "batch_sampler: out_cap too large");
    const int64_t gcap = batch_table_slots(out_cap);
    SG_REQUIRE(workspace && workspace_bytes >= (size_t)gcap * 4, SUBGACC_ERR_WORKSPACE, "batch_sampler: workspace too small");
    BatchArgs a;
    a.indptr = indptr, a.indices = indices, a.query = query, a.n = n, a.num_nodes = num_nodes;
    a.M = num_walks, a.S = num_steps, a.thld = thld, a.seed = seed_eff;
    a.out = out, a.out_cap = out_cap, a.out_count = out_count;
    a.gkeys = (int32_t *)workspace, a.gmask = (uint32_t)(gcap - 1), a.flags = flags;
    a.TL = table_size_for(Q);
    a.tlshift = 32 - (31 - __builtin_clz((unsigned)a.TL));
    a.nwords = (int)((Q + 32) / 32);       // visit numbers 0 .. M*S inclusive
    const size_t lds = (size_t)a.TL * 8 + (size_t)a.nwords * 4 + (size_t)(a.nwords + 1) * 4 + (size_t)num_walks * 4 +
                       (size_t)num_walks * num_steps * 4 + 16;
    SG_REQUIRE(lds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS, "batch_sampler: %zu B of LDS needed", lds);
    hipLaunchKernelGGL(batch_fill_kernel, dim3((unsigned)ceil_div(gcap, 256)), dim3(256), 0, s, a.gkeys, gcap);
    if (indptr64) {
        if (lds > 64 * 1024)
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)batch_sampler_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(batch_sampler_kernel<true>, dim3(1), dim3(kBatchThreads), lds, s, a);
    } else {
        if (lds > 64 * 1024)
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)batch_sampler_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(batch_sampler_kernel<false>, dim3(1), dim3(kBatchThreads), lds, s, a);
    }
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
