// ppr.hip -- top-K approximate personalised PageRank sets for gfx950.
//
// Reference: sampler/pprgo.py:9-38 (_calc_ppr_node: Andersen-Chung-Lang push with a LIFO work list),
// :53-63 (top-k per root), :85-111 (degree normalisation), utils.py:35-36 (encoding 'PPR').
//
// The push ORDER fixes every float32 rounding of the result, so the kernel keeps the reference's order exactly and
// takes its parallelism from (a) the roots -- one wavefront per root, thousands resident -- and (b) the neighbour
// loop of one push, whose iterations touch distinct nodes of a simple graph and are independent: the 64 lanes take
// 64 consecutive neighbours, insert-or-find them in the wave's private hash table, add the same increment, test the
// threshold and append the nodes that pass to the work list in lane order (ballot + prefix popcount), which is the
// order the sequential loop appends them in.
//
// State of one root (dicts p, r and list q of the reference) lives in a per-wave slab of HBM sized for the
// worst case of the approximation (<= 1/(alpha*eps) touched nodes; 288 GB make 8192 x MBs affordable) -- open
// addressing, cleaned by the list of touched slots, so a slab is cleared once per launch, not once per root.
// All slab accesses are workgroup-scope atomics: lanes of the wave hand values to each other through the XCD's L2.
#include "common.hpp"

namespace subgacc {

constexpr int kPprThreads = 64;                 // one wavefront per workgroup, one root per wavefront at a time
constexpr int kPprUnroll = 4;                   // neighbours per lane and trip of the push loop
constexpr uint32_t kInQueue = 0x80000000u;      // sign bit of the residual word: the node is on the work list

struct PprArgs {
    const void *indptr;
    const int32_t *indices;
    const int32_t *roots;
    int64_t n;
    float alpha, alpha_eps;
    int32_t topk;
    int32_t cap, hshift;      // table slots per wave (pow2), 32 - log2(cap)
    int32_t *slab;            // [gridDim.x][6][cap]: keys | r | p | meta | stack | touched
    int32_t *out_count;       // [n]   entries of row i, -1: table too small for this root
    int32_t *out_ids;         // [n * topk]  ascending node id
    float *out_vals;          // [n * topk]
    int32_t *flags;
    unsigned long long *pushes;
};

// Slab accesses: WORKGROUP-scope atomic loads / stores (no read-modify-writes).  A slab belongs to one single-wave
// workgroup for the whole launch, so its lines live in that XCD's L2; workgroup scope keeps the accesses there
// (device scope sends them past the per-XCD L2s to the fabric: 87 % L2 misses, measured).
constexpr auto kScope = __HIP_MEMORY_SCOPE_WORKGROUP;
template <typename T>
__device__ __forceinline__ T ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, kScope); }
template <typename T>
__device__ __forceinline__ void st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, kScope); }

template <bool IDX64>
__device__ __forceinline__ int64_t row_of(const void *indptr, int32_t node, int64_t &beg) {
    if (IDX64) {
        const int64_t *q = (const int64_t *)indptr + node;
        beg = q[0];
        return q[1] - beg;
    }
    const int32_t *q = (const int32_t *)indptr + node;
    const int32_t b = q[0], e = q[1];
    beg = b;
    return (int64_t)(e - b);
}

// One node of the dicts p / r of the reference = one slot of two parallel planes:
//   kr[slot]  low: node id (0xFFFFFFFF = free); high: residual r (float bits, r >= 0) | kInQueue   -- the hot plane
//   pm[slot]  low: score p (float bits); high: 1-based order of entry into p, 0 = never popped     -- touched by pops only
constexpr unsigned long long kFreeSlot = 0x00000000FFFFFFFFull;

__device__ __forceinline__ unsigned long long pack2(uint32_t lo, uint32_t hi) { return ((unsigned long long)hi << 32) | lo; }

template <bool IDX64>
__global__ __launch_bounds__(kPprThreads) void ppr_push_kernel(const PprArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    uint32_t *hist = (uint32_t *)lds_raw;                 // [256]
    int32_t *sel_id = (int32_t *)(hist + 256);            // [topk]
    float *sel_val = (float *)(sel_id + a.topk);          // [topk]

    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t mask = (uint32_t)a.cap - 1u;
    unsigned long long *kr = (unsigned long long *)(a.slab + (size_t)blockIdx.x * 6u * (size_t)a.cap);   // [cap]
    unsigned long long *pm = kr + a.cap;                                              // [cap]
    int32_t *stack = (int32_t *)(pm + a.cap);                                         // [cap]
    int32_t *touched = stack + a.cap;                                                 // [cap] grows upwards
    int32_t *plist = touched + (a.cap - 1);      // slots of the popped nodes in order of entry into p; grows DOWNWARDS
                                                 // from the end of `touched`; np + ntouched <= cap is part of the
                                                 // overflow test, so the two lists never meet
    const double one_minus_alpha = 1.0 - (double)a.alpha;     // int64 - float32 -> float64 (numba)
    unsigned long long my_pushes = 0, my_touched = 0;

    for (int64_t i = blockIdx.x; i < a.n; i += gridDim.x) {
        const int32_t inode = a.roots[i];
        int32_t ntouched = 1, np = 1, ns = 1;
        unsigned long long root_pushes = 0;
        bool overflow = false;
        if (lane == 0) {      // p = {inode: 0}; r[inode] = alpha; q = [inode]   (pprgo.py:13-16); the slab is clean
            const int32_t s0 = (int32_t)(((uint32_t)inode * 2654435761u) >> a.hshift);
            st(&kr[s0], pack2((uint32_t)inode, __float_as_uint(a.alpha) | kInQueue));
            st(&pm[s0], pack2(0u, 1u));
            st(&touched[0], s0);
            st(&plist[0], s0);
            st(&stack[0], s0);
        }
        __syncthreads();
        while (ns > 0) {
            if (np + 1 + ntouched > a.cap) {   // `plist` would run into `touched` (pops of sink nodes add no trip test)
                overflow = true;
                break;
            }
            --ns;
            const int32_t su = ld(&stack[ns]);                 // q.pop()  (:18)
            const unsigned long long ekr = ld(&kr[su]);
            const unsigned long long epm = ld(&pm[su]);
            const int32_t unode = (int32_t)(uint32_t)ekr;
            const float res = __uint_as_float((uint32_t)(ekr >> 32) & ~kInQueue);   // :20
            const uint32_t mu = (uint32_t)(epm >> 32);
            const bool first = mu == 0u;
            if (first) ++np;
            int64_t b;
            const int64_t d = row_of<IDX64>(a.indptr, unode, b);
            if (lane == 0) {   // p[unode] += res (:21-24); r[unode] = 0 (:25); off the work list
                st(&kr[su], pack2((uint32_t)unode, 0u));
                st(&pm[su], pack2(__float_as_uint(__uint_as_float((uint32_t)epm) + res), first ? (uint32_t)np : mu));
                if (first) st(&plist[-(np - 1)], su);
                ++root_pushes;
            }
            const float val = d > 0 ? (float)(one_minus_alpha * (double)res / (double)d) : 0.0f;   // :27
            __syncthreads();
            // kPprUnroll x 64 neighbours per trip, their loads in flight together.  No read-modify-write atomics: on
            // this part every global atomic is executed at the memory side of the fabric, not in the XCD's L2 (measured:
            // TCC_EA0_ATOMIC == TCC_ATOMIC), and that rate bounded the kernel.  The lanes hold DISTINCT nodes (simple
            // graph), so an entry has one writer per trip; a free slot is claimed by storing the finished entry and
            // reading it back after the wave's stores have landed -- of several claimants the last store wins, the
            // others probe on.
            for (int64_t base = 0; base < d; base += kPprUnroll * kPprThreads) {
                bool act[kPprUnroll], isnew[kPprUnroll], pass[kPprUnroll], open[kPprUnroll], claim[kPprUnroll];
                int32_t v[kPprUnroll];
                int64_t dv[kPprUnroll];
                uint32_t h[kPprUnroll];
                unsigned long long e[kPprUnroll];
#pragma unroll
                for (int u = 0; u < kPprUnroll; ++u) {
                    const int64_t j = base + u * kPprThreads + lane;
                    act[u] = j < d;
                    v[u] = act[u] ? a.indices[b + j] : 0;
                }
#pragma unroll
                for (int u = 0; u < kPprUnroll; ++u) {
                    int64_t bv;
                    dv[u] = act[u] ? row_of<IDX64>(a.indptr, v[u], bv) : 0;
                    h[u] = ((uint32_t)v[u] * 2654435761u) >> a.hshift;
                    e[u] = act[u] ? ld(&kr[h[u]]) : 0ull;
                    open[u] = act[u];
                    isnew[u] = pass[u] = false;
                }
                bool any_open = true;
                while (any_open) {
                    // (1) settle what the loaded entry allows: found -> update in place; free -> claim; else probe on
#pragma unroll
                    for (int u = 0; u < kPprUnroll; ++u) {
                        claim[u] = false;
                        if (!open[u]) continue;
                        const int32_t k = (int32_t)(uint32_t)e[u];
                        if (k != v[u] && k != -1) continue;
                        const uint32_t rq = k == v[u] ? (uint32_t)(e[u] >> 32) : 0u;
                        const float rv = __uint_as_float(rq & ~kInQueue) + val;                      // :28-31
                        const bool inq = (rq & kInQueue) != 0u;
                        pass[u] = !inq && (double)rv >= (double)a.alpha_eps * (double)dv[u];     // :33-35
                        st(&kr[h[u]], pack2((uint32_t)v[u], __float_as_uint(rv) | ((inq || pass[u]) ? kInQueue : 0u)));
                        if (k == v[u]) open[u] = false;
                        else claim[u] = true;
                    }
                    __syncthreads();
                    // (2) claimants read back; the others step to the next slot
                    any_open = false;
#pragma unroll
                    for (int u = 0; u < kPprUnroll; ++u) {
                        if (!open[u]) continue;
                        if (!claim[u]) h[u] = (h[u] + 1u) & mask;
                        e[u] = ld(&kr[h[u]]);
                    }
#pragma unroll
                    for (int u = 0; u < kPprUnroll; ++u) {
                        if (!open[u]) continue;
                        if (claim[u] && (int32_t)(uint32_t)e[u] == v[u]) {
                            open[u] = false;
                            isnew[u] = true;
                        } else {
                            pass[u] = false;     // lost the slot (or still probing): e[u] holds what to look at next
                            any_open = true;
                        }
                    }
                    any_open = __any(any_open);
                }
#pragma unroll
                for (int u = 0; u < kPprUnroll; ++u) {
                    const unsigned long long newm = __ballot(isnew[u]), passm = __ballot(pass[u]);
                    if (isnew[u]) st(&touched[ntouched + __popcll(newm & lt)], (int32_t)h[u]);
                    if (pass[u]) st(&stack[ns + __popcll(passm & lt)], (int32_t)h[u]);          // :36 q.append
                    ntouched += __popcll(newm);
                    ns += __popcll(passm);
                }
                __syncthreads();
                if (((int64_t)ntouched + kPprUnroll * kPprThreads) * 4 > 3 * (int64_t)a.cap ||
                    np + 1 + ntouched + kPprUnroll * kPprThreads > a.cap) {   // the next trip / pop could pass 3/4 of the table
                    overflow = true;
                    ns = 0;
                    break;
                }
            }
        }
        if (!overflow) my_pushes += root_pushes, my_touched += (unsigned long long)ntouched;   // abandoned attempts do not count

        // ---- top-k of p by (value, order of entry) -- np.argsort(val)[-topk:] (:59-61), ties keep the later entry
        int32_t nsel = np < a.topk ? np : a.topk;
        if (overflow) {
            nsel = 0;
            if (lane == 0) {
                a.out_count[i] = -1;
                atomicOr(&a.flags[2], 1);
            }
        } else {
            unsigned long long thr = 0ull;
            if (np > a.topk) {
                unsigned long long prefix = 0ull;
                uint32_t kk = (uint32_t)a.topk;
                for (int shift = 56; shift >= 0; shift -= 8) {
                    for (int x = lane; x < 256; x += kPprThreads) hist[x] = 0u;
                    __syncthreads();
                    for (int x = lane; x < np; x += kPprThreads) {
                        const int32_t s = ld(&plist[-x]);
                        const unsigned long long epm = ld(&pm[s]);   // (score bits, order)
                        const uint32_t m = (uint32_t)(epm >> 32);
                        if (m) {   // key64 = (score bits << 32) | order of entry
                            const unsigned long long k64 = (epm << 32) | m;
                            if (shift == 56 || (k64 >> (shift + 8)) == prefix) atomicAdd(&hist[(k64 >> shift) & 255u], 1u);
                        }
                    }
                    __syncthreads();
                    // digit of the kk-th largest key among the candidates: lanes own 4 bins each
                    const uint32_t c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
                    uint32_t suf = c0 + c1 + c2 + c3;          // inclusive suffix sum over lanes
#pragma unroll
                    for (int o = 1; o < kPprThreads; o <<= 1) {
                        const uint32_t t = __shfl_down(suf, o, kPprThreads);
                        if (lane + o < kPprThreads) suf += t;
                    }
                    const unsigned long long ge = __ballot(suf >= kk);
                    const int L = 63 - __clzll(ge);            // highest lane whose suffix still holds kk keys
                    uint32_t above = __shfl(suf, L, kPprThreads) - __shfl(c0 + c1 + c2 + c3, L, kPprThreads);
                    int digit = 4 * L;
                    for (int dd = 3; dd >= 0; --dd) {
                        const uint32_t c = hist[4 * L + dd];
                        if (above + c >= kk) {
                            digit = 4 * L + dd;
                            break;
                        }
                        above += c;
                    }
                    kk -= above;
                    prefix = (prefix << 8) | (unsigned long long)digit;
                    __syncthreads();
                }
                thr = prefix;
            }
            int32_t base = 0;
            for (int x0 = 0; x0 < np; x0 += kPprThreads) {
                const int x = x0 + lane;
                bool sel = false;
                int32_t s = 0;
                float pv = 0.f;
                if (x < np) {
                    s = ld(&plist[-x]);
                    const unsigned long long epm = ld(&pm[s]);
                    const uint32_t m = (uint32_t)(epm >> 32);
                    if (m) {
                        pv = __uint_as_float((uint32_t)epm);
                        sel = ((epm << 32) | m) >= thr;
                    }
                }
                const unsigned long long sm = __ballot(sel);
                if (sel) {
                    const int32_t pos = base + __popcll(sm & lt);
                    sel_id[pos] = (int32_t)(uint32_t)ld(&kr[s]);
                    sel_val[pos] = pv;
                }
                base += __popcll(sm);
            }
            __syncthreads();
            // rows leave sorted by node id (coo -> csr, :66-82,:87): rank = number of smaller ids
            for (int e = lane; e < nsel; e += kPprThreads) {
                const int32_t id = sel_id[e];
                int32_t rank = 0;
                for (int f = 0; f < nsel; ++f) rank += sel_id[f] < id;
                a.out_ids[i * a.topk + rank] = id;
                a.out_vals[i * a.topk + rank] = sel_val[e];
            }
            if (lane == 0) a.out_count[i] = nsel;
        }
        // ---- hand the slab back clean
        for (int x = lane; x < ntouched; x += kPprThreads) {
            const int32_t s = ld(&touched[x]);
            st(&kr[s], kFreeSlot);
            st(&pm[s], 0ull);
        }
        __syncthreads();
    }
    if (a.pushes && lane == 0 && my_touched) {
        atomicAdd(&a.pushes[0], my_pushes);
        atomicAdd(&a.pushes[1], my_touched);
    }
}

__global__ __launch_bounds__(256) void ppr_slab_reset_kernel(unsigned long long *slab, int64_t cap, int64_t waves) {
    // the two entry planes at the head of every 24*cap-byte slab: kr = free, pm = 0
    const int64_t total = 2 * cap * waves;
    for (int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x; x < total; x += (int64_t)gridDim.x * 256) {
        const int64_t w = x / (2 * cap), e = x - w * 2 * cap;
        slab[w * 3 * cap + e] = e < cap ? kFreeSlot : 0ull;
    }
}

// pprgo.py:88-108 + optional row_of_entry search: one thread per entry, row found by binary search in row_off
template <bool IDX64>
__global__ __launch_bounds__(256) void ppr_normalize_kernel(const void *indptr, const int32_t *roots, int64_t n,
                                                            const int64_t *row_off, const int32_t *ids, const float *vals,
                                                            int mode, double *out, unsigned long long *max_bits) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nnz = row_off[n];
    double v = 0.0;
    if (x < nnz) {
        int64_t lo = 0, hi = n;                      // last row with row_off[row] <= x
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (row_off[mid] <= x) lo = mid;
            else hi = mid;
        }
        int64_t t;
        const double dr = (double)row_of<IDX64>(indptr, roots[lo], t);
        const double dc = (double)row_of<IDX64>(indptr, ids[x], t);
        v = (double)vals[x];
        if (mode == 1) v = sqrt(fmax(dr, 1e-12)) * v * (1.0 / sqrt(fmax(dc, 1e-12)));
        else if (mode == 2) v = dr * v * (1.0 / fmax(dc, 1e-12));
        out[x] = v;
    }
    if (max_bits) {   // non-negative doubles order like their bit patterns
        double m = x < nnz ? v : 0.0;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, kWave));
        if ((threadIdx.x & (kWave - 1)) == 0 && m > 0.0) atomicMax(max_bits, (unsigned long long)__double_as_longlong(m));
    }
}

// utils.py:35-36: x.data = (x.data + 0.1) / (x.data.max() + 0.1)
__global__ __launch_bounds__(256) void ppr_encode_kernel(double *data, const int64_t *nnz_dev, const unsigned long long *max_bits) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= *nnz_dev) return;
    const double mx = __longlong_as_double((long long)*max_bits);
    data[x] = (data[x] + 0.1) / (mx + 0.1);
}

}  // namespace subgacc

using namespace subgacc;

extern "C" size_t subgacc_ppr_slab_bytes(int32_t table_log2, int32_t num_waves) {
    if (table_log2 < 10 || table_log2 > 26 || num_waves < 1) return 0;   // half of the table + 256 entries of slack must fit
    return (size_t)num_waves * 6u * sizeof(int32_t) * ((size_t)1 << table_log2);
}

extern "C" int subgacc_ppr_slab_reset(void *slab, int32_t table_log2, int32_t num_waves, void *stream) {
    SG_REQUIRE(slab && subgacc_ppr_slab_bytes(table_log2, num_waves) > 0, SUBGACC_ERR_BADARG, "ppr_slab_reset: bad arguments");
    const int64_t cap = (int64_t)1 << table_log2;
    int64_t blocks = ceil_div(2 * cap * num_waves, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(ppr_slab_reset_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long *)slab, cap,
                       (int64_t)num_waves);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_ppr_topk(const void *indptr, int32_t indptr64, const int32_t *indices, int64_t num_nodes,
                                const int32_t *roots, int64_t n, float alpha, float epsilon, int32_t topk,
                                void *slab, int32_t table_log2, int32_t num_waves, int32_t *out_count, int32_t *out_ids,
                                float *out_vals, int32_t *flags, uint64_t *pushes, void *stream) {
    SG_REQUIRE(indptr && out_count && out_ids && out_vals && flags && slab, SUBGACC_ERR_BADARG, "ppr_topk: null argument");
    SG_REQUIRE(n >= 0 && num_nodes >= 0 && topk >= 1 && topk <= 4096, SUBGACC_ERR_BADARG, "ppr_topk: topk must be in [1, 4096]");
    SG_REQUIRE(subgacc_ppr_slab_bytes(table_log2, num_waves) > 0, SUBGACC_ERR_BADARG, "ppr_topk: table_log2 in [10, 26], num_waves >= 1");
    SG_REQUIRE(alpha > 0.f && alpha <= 1.f && epsilon > 0.f, SUBGACC_ERR_BADARG, "ppr_topk: alpha in (0,1], epsilon > 0");
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(roots, SUBGACC_ERR_BADARG, "ppr_topk: null roots");
    PprArgs a;
    a.indptr = indptr, a.indices = indices, a.roots = roots, a.n = n;
    a.alpha = alpha;
    a.alpha_eps = alpha * epsilon;          // float32 product, as numba types it
    a.topk = topk;
    a.cap = 1 << table_log2, a.hshift = 32 - table_log2;
    a.slab = (int32_t *)slab;
    a.out_count = out_count, a.out_ids = out_ids, a.out_vals = out_vals, a.flags = flags;
    a.pushes = (unsigned long long *)pushes;
    const int64_t grid = n < num_waves ? n : num_waves;
    const size_t lds = 1024 + (size_t)topk * 8;
    if (indptr64) hipLaunchKernelGGL(ppr_push_kernel<true>, dim3((unsigned)grid), dim3(kPprThreads), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ppr_push_kernel<false>, dim3((unsigned)grid), dim3(kPprThreads), lds, (hipStream_t)stream, a);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_ppr_normalize(const void *indptr, int32_t indptr64, const int32_t *roots, int64_t n,
                                     const int64_t *row_off, int64_t max_nnz, const int32_t *ids, const float *vals,
                                     int32_t mode, double *out, uint64_t *max_bits, void *stream) {
    SG_REQUIRE(mode >= 0 && mode <= 2, SUBGACC_ERR_BADARG, "ppr_normalize: mode 0 (row), 1 (sym) or 2 (col)");
    SG_REQUIRE(n >= 0 && max_nnz >= 0, SUBGACC_ERR_BADARG, "ppr_normalize: negative size");
    if (n == 0 || max_nnz == 0) return SUBGACC_OK;
    SG_REQUIRE(indptr && roots && row_off && ids && vals && out, SUBGACC_ERR_BADARG, "ppr_normalize: null argument");
    const unsigned grid = (unsigned)ceil_div(max_nnz, 256);
    if (indptr64) hipLaunchKernelGGL(ppr_normalize_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, indptr, roots, n, row_off, ids, vals, mode, out, (unsigned long long *)max_bits);
    else hipLaunchKernelGGL(ppr_normalize_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, indptr, roots, n, row_off, ids, vals, mode, out, (unsigned long long *)max_bits);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_ppr_encode(double *data, int64_t max_nnz, const int64_t *nnz_dev, const uint64_t *max_bits, void *stream) {
    SG_REQUIRE(max_nnz >= 0, SUBGACC_ERR_BADARG, "ppr_encode: negative size");
    if (max_nnz == 0) return SUBGACC_OK;
    SG_REQUIRE(data && nnz_dev && max_bits, SUBGACC_ERR_BADARG, "ppr_encode: null argument");
    hipLaunchKernelGGL(ppr_encode_kernel, dim3((unsigned)ceil_div(max_nnz, 256)), dim3(256), 0, (hipStream_t)stream, data, nnz_dev,
                       (const unsigned long long *)max_bits);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
