// encode.hip -- the DEG and SPD structural encoders over a PPR node-set store (gfx950).
//
// Reference: utils.py:22-34 (`encoding(x, adj, 'DEG' | 'SPD')`), SciPy sparse algebra over x = the top-K PPR matrix
// (one row per node) and adj = the (symmetric, unweighted) adjacency:
//   DEG  x += normalize(adj, 'l1', axis=1);  x_deg = log(x.getnnz(1) + 1);  agg = x.copy();
//        x.data = (x > 0).multiply(x_deg).data            -> pattern X u A, value(i,j) = log(nnz(row j of X u A) + 1)
//   SPD  x0 = x > 0; x1 = adj > 0; x2 = x1 ** 2;  x = x1 + x0.multiply(x2 * 0.5) + x0 * 0.3;  x.setdiag(2.3)
//                                                         -> pattern A u X u diag, value = 1[A] + 0.5[X and 2-path] + 0.3[X]
// Both are a per-row UNION of two sorted id lists with a value rule.  One wave64 workgroup per row: the short list
// (the PPR row, <= top-K) goes to LDS, every element finds its slot in the merged row by a binary search in the
// other list plus a prefix count, so the merged row is written in one pass, sorted, without a sort.
// The 2-path test of SPD (x2 = A*A as a boolean) is a sorted-set intersection N(i) n N(j): lanes walk the shorter
// adjacency row, binary-search the longer one in place and stop at the first hit.
#include "common.hpp"

namespace subgacc {

constexpr int kEncThreads = 64;

struct EncArgs {
    const int64_t *x_off;      // [n+1]
    const int32_t *x_ids;      // sorted per row
    const double *x_val;
    int64_t n;                 // rows = nodes
    const void *indptr;        // adjacency, rows sorted, no repeated entries
    const int32_t *indices;
    int32_t mode;              // 1 DEG, 2 SPD
    int32_t kmax;              // longest x row
    int32_t *row_len;          // sizes pass: out; DEG fill: in (union size of every row, no diagonal)
    const int64_t *out_off;    // [n+1]
    int32_t *out_ids;
    double *out_val;
    double *out_agg;           // DEG: x + D^-1 A
    const double *log_table;   // DEG: log_table[d] = log(d + 1), computed on the host with the reference's libm
    int64_t log_len;
    int32_t *flags;
};

template <bool IDX64>
__device__ __forceinline__ int64_t adj_row(const void *indptr, int64_t node, int64_t &beg) {
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

__device__ __forceinline__ int64_t lower_bound(const int32_t *a, int64_t n, int32_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// is N(i) n N(j) non-empty?  (wave-cooperative; rows sorted)
__device__ __forceinline__ bool rows_intersect(const int32_t *a, int64_t na, const int32_t *b, int64_t nb, int lane) {
    if (na > nb) {
        const int32_t *t = a; a = b; b = t;
        const int64_t tn = na; na = nb; nb = tn;
    }
    for (int64_t base = 0; base < na; base += kEncThreads) {
        bool hit = false;
        if (base + lane < na) {
            const int32_t e = a[base + lane];
            const int64_t p = lower_bound(b, nb, e);
            hit = p < nb && b[p] == e;
        }
        if (__any(hit)) return true;
    }
    return false;
}

// PASS 1: row_len[i] = |X_i u A_i| (+1 when SPD adds a missing diagonal entry)
template <bool IDX64>
__global__ __launch_bounds__(kEncThreads) void enc_sizes_kernel(const EncArgs a) {
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t xb = a.x_off[i], nx = a.x_off[i + 1] - xb;
    int64_t ab;
    const int64_t na = adj_row<IDX64>(a.indptr, i, ab);
    const int32_t *A = a.indices + ab, *X = a.x_ids + xb;
    int both = 0;
    bool diag = false;
    for (int64_t j = lane; j < nx; j += kEncThreads) {
        const int32_t x = X[j];
        const int64_t p = lower_bound(A, na, x);
        both += (p < na && A[p] == x);
        diag |= x == (int32_t)i;
    }
#pragma unroll
    for (int o = kEncThreads / 2; o > 0; o >>= 1) both += __shfl_xor(both, o, kEncThreads);
    diag = __any(diag);
    if (lane == 0) {
        if (!diag) {
            const int64_t p = lower_bound(A, na, (int32_t)i);
            diag = p < na && A[p] == (int32_t)i;
        }
        a.row_len[i] = (int32_t)(nx + na - both + ((a.mode == 2 && !diag) ? 1 : 0));
    }
}

// PASS 2: the merged row
template <bool IDX64>
__global__ __launch_bounds__(kEncThreads) void enc_fill_kernel(const EncArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    int32_t *xs = (int32_t *)lds_raw;                    // [kmax]   ids of the PPR row
    int32_t *pref = xs + a.kmax;                         // [kmax+1] PPR-only entries before position j
    uint8_t *inA = (uint8_t *)(pref + a.kmax + 1);       // [kmax]   also a neighbour
    uint8_t *two = inA + a.kmax;                         // [kmax]   SPD: a 2-path i -> k -> j exists

    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t xb = a.x_off[i];
    const int nx = (int)(a.x_off[i + 1] - xb);
    int64_t ab;
    const int64_t na = adj_row<IDX64>(a.indptr, i, ab);
    const int32_t *A = a.indices + ab;
    const double *xv = a.x_val + xb;
    const int64_t ob = a.out_off[i];

    for (int j = lane; j < nx; j += kEncThreads) {
        const int32_t x = a.x_ids[xb + j];
        xs[j] = x;
        const int64_t p = lower_bound(A, na, x);
        inA[j] = p < na && A[p] == x;
        two[j] = 0;
    }
    __syncthreads();
    if (a.mode == 2) {      // x2 = x1 ** 2 restricted to the PPR pattern: N(i) n N(j) != {} (adjacency taken as symmetric)
        for (int j = 0; j < nx; ++j) {
            int64_t jb;
            const int64_t nj = adj_row<IDX64>(a.indptr, xs[j], jb);
            const bool hit = rows_intersect(A, na, a.indices + jb, nj, lane);
            if (lane == 0) two[j] = hit;
        }
    }
    if (lane == 0) {        // <= top-K entries: a serial prefix is cheaper than a scan here
        int c = 0;
        for (int j = 0; j < nx; ++j) {
            pref[j] = c;
            c += !inA[j];
        }
        pref[nx] = c;
    }
    __syncthreads();
    // SPD puts (i, 2.3) on the diagonal: present -> overwritten, absent -> inserted (everything above shifts by one)
    const int32_t self = (int32_t)i;
    bool has_diag = false;
    int64_t dpos = 0;
    if (a.mode == 2) {
        const int64_t pa = lower_bound(A, na, self);
        const int64_t px = lower_bound(xs, nx, self);
        has_diag = (pa < na && A[pa] == self) || (px < nx && xs[px] == self);
        dpos = pa + pref[px];
    }
    const int64_t shift_from = (a.mode == 2 && !has_diag) ? (int64_t)self : (int64_t)0x7FFFFFFF;   // ids above it move by one
    const double inv_deg = na > 0 ? 1.0 / (double)na : 0.0;      // normalize(adj, 'l1'): 1 / row sum
    // neighbours (in A, maybe also in X)
    for (int64_t ja = lane; ja < na; ja += kEncThreads) {
        const int32_t v = A[ja];
        const int64_t lb = lower_bound(xs, nx, v);
        const bool inx = lb < nx && xs[lb] == v;
        const int64_t pos = ob + ja + pref[lb] + (v > shift_from ? 1 : 0);
        a.out_ids[pos] = v;
        if (a.mode == 1) {
            const int32_t d = a.row_len[v];
            a.out_val[pos] = (int64_t)d < a.log_len ? a.log_table[d] : 0.0;
            if ((int64_t)d >= a.log_len) atomicOr(&a.flags[3], 8);
            a.out_agg[pos] = inx ? xv[lb] + inv_deg : inv_deg;
        } else {
            double val = 1.0;
            if (inx) {
                if (two[lb]) val += 0.5;
                val += 0.3;
            }
            a.out_val[pos] = v == self ? 2.3 : val;
        }
    }
    // PPR-only entries
    for (int j = lane; j < nx; j += kEncThreads) {
        if (inA[j]) continue;
        const int32_t v = xs[j];
        const int64_t pos = ob + lower_bound(A, na, v) + pref[j] + (v > shift_from ? 1 : 0);
        a.out_ids[pos] = v;
        if (a.mode == 1) {
            const int32_t d = a.row_len[v];
            a.out_val[pos] = (int64_t)d < a.log_len ? a.log_table[d] : 0.0;
            if ((int64_t)d >= a.log_len) atomicOr(&a.flags[3], 8);
            a.out_agg[pos] = xv[j];
        } else {
            const double val = two[j] ? 0.5 + 0.3 : 0.3;
            a.out_val[pos] = v == self ? 2.3 : val;
        }
    }
    if (a.mode == 2 && !has_diag && lane == 0) {
        a.out_ids[ob + dpos] = self;
        a.out_val[ob + dpos] = 2.3;
    }
}

}  // namespace subgacc

using namespace subgacc;

static int enc_args(EncArgs &a, const int64_t *x_off, const int32_t *x_ids, const double *x_val, int64_t n, const void *indptr,
                    const int32_t *indices, int32_t mode, int32_t kmax, int32_t *flags) {
    SG_REQUIRE(n >= 0 && (mode == 1 || mode == 2) && kmax >= 0 && kmax <= 8192 && flags, SUBGACC_ERR_BADARG,
               "encode: mode 1 (DEG) or 2 (SPD), rows of the node-set store <= 8192 entries");
    SG_REQUIRE(n == 0 || (x_off && indptr), SUBGACC_ERR_BADARG, "encode: null argument");
    SG_REQUIRE(n < (1ll << 31), SUBGACC_ERR_BADARG, "encode: too many rows");
    a.x_off = x_off, a.x_ids = x_ids, a.x_val = x_val, a.n = n;
    a.indptr = indptr, a.indices = indices, a.mode = mode, a.kmax = kmax > 0 ? kmax : 1;
    a.row_len = nullptr, a.out_off = nullptr, a.out_ids = nullptr, a.out_val = nullptr, a.out_agg = nullptr;
    a.log_table = nullptr, a.log_len = 0, a.flags = flags;
    return SUBGACC_OK;
}

extern "C" int subgacc_encode_sizes(const int64_t *x_off, const int32_t *x_ids, int64_t n, const void *indptr,
                                    int32_t indptr64, const int32_t *indices, int32_t mode, int32_t *row_len,
                                    int32_t *flags, void *stream) {
    EncArgs a;
    int rc = enc_args(a, x_off, x_ids, nullptr, n, indptr, indices, mode, 1, flags);
    if (rc != SUBGACC_OK) return rc;
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(row_len, SUBGACC_ERR_BADARG, "encode_sizes: null row_len");
    a.row_len = row_len;
    if (indptr64) hipLaunchKernelGGL(enc_sizes_kernel<true>, dim3((unsigned)n), dim3(kEncThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(enc_sizes_kernel<false>, dim3((unsigned)n), dim3(kEncThreads), 0, (hipStream_t)stream, a);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}

extern "C" int subgacc_encode_fill(const int64_t *x_off, const int32_t *x_ids, const double *x_val, int64_t n, int32_t kmax,
                                   const void *indptr, int32_t indptr64, const int32_t *indices, int32_t mode,
                                   const int32_t *deg_row_len, const double *log_table, int64_t log_len,
                                   const int64_t *out_off, int32_t *out_ids, double *out_val, double *out_agg,
                                   int32_t *flags, void *stream) {
    EncArgs a;
    int rc = enc_args(a, x_off, x_ids, x_val, n, indptr, indices, mode, kmax, flags);
    if (rc != SUBGACC_OK) return rc;
    if (n == 0) return SUBGACC_OK;
    SG_REQUIRE(out_off && out_ids && out_val && x_val, SUBGACC_ERR_BADARG, "encode_fill: null argument");
    SG_REQUIRE(mode != 1 || (deg_row_len && log_table && log_len > 0 && out_agg), SUBGACC_ERR_BADARG,
               "encode_fill: DEG needs the union sizes of all rows, the log table and out_agg");
    a.row_len = const_cast<int32_t *>(deg_row_len);
    a.log_table = log_table, a.log_len = log_len;
    a.out_off = out_off, a.out_ids = out_ids, a.out_val = out_val, a.out_agg = out_agg;
    const size_t lds = (size_t)a.kmax * 4 + ((size_t)a.kmax + 1) * 4 + (size_t)a.kmax * 2 + 16;
    if (lds > 64 * 1024) {
        SG_CHECK_HIP(hipFuncSetAttribute((const void *)enc_fill_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SG_CHECK_HIP(hipFuncSetAttribute((const void *)enc_fill_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (indptr64) hipLaunchKernelGGL(enc_fill_kernel<true>, dim3((unsigned)n), dim3(kEncThreads), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(enc_fill_kernel<false>, dim3((unsigned)n), dim3(kEncThreads), lds, (hipStream_t)stream, a);
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
