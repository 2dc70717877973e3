// sjoin.hip -- SpJoin: the set-join structural encoder of SUREL+ (gfx950).
//
// Replaces the SciPy sparse arithmetic of train.py:13-45 (gather), :48-72 (hgather), :75-111
// (bgather/pgather): `x[edge[0]]`, `xr.multiply(lmask) + lmask`, `.data - 1`, `np.stack`, `encode[...]`.
// Those five temporaries and the host->device upload of the index array collapse into one kernel over a
// device-resident SpG:  one wave64 workgroup per output segment (own row, partner row); the partner row's
// sorted ids and payload are staged in LDS with coalesced loads, every lane takes one member of the own
// row, finds it in the partner row by binary search in LDS (sorted-set intersection), and writes the
// feature pair straight from the Z_SF table (L2-resident, a few KB..MB) -- the [R,2] index array of the
// reference never exists in memory unless asked for.
#include "common.hpp"

namespace subgacc {

constexpr int kJoinThreads = 64;

__global__ void sjoin_len_kernel(const int64_t *__restrict__ indptr, const int64_t *__restrict__ own, int64_t S,
                                 int64_t *__restrict__ len) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < S) {
        const int64_t a = own[j];
        len[j] = indptr[a + 1] - indptr[a];
    }
}

struct JoinArgs {
    const int64_t *indptr;
    const int32_t *indices;
    const void *data;  // int32 (SFptr+1) or double (PPR score)
    const int64_t *own, *partner, *seg;
    int64_t S;
    const float *table;
    int64_t table_rows;
    int32_t k;
    float *out_xz;
    int32_t *out_idx;
    int64_t *out_segid;
    int32_t max_len;
    int32_t *flags;
};

template <bool F64>
__global__ __launch_bounds__(kJoinThreads) void sjoin_fill_kernel(const JoinArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    using Val = typename std::conditional<F64, double, int32_t>::type;
    Val *pval = (Val *)lds_raw;                       // [max_len]
    int32_t *pids = (int32_t *)(pval + a.max_len);    // [max_len]

    const int64_t j = xcd_item(blockIdx.x, gridDim.x);
    if (j >= a.S) return;
    const int tid = threadIdx.x;
    const int64_t ra = a.own[j], rb = a.partner[j];
    const int64_t ab = a.indptr[ra], na = a.indptr[ra + 1] - ab;
    const int64_t bb = a.indptr[rb], nb64 = a.indptr[rb + 1] - bb;
    if (nb64 > a.max_len) {
        if (tid == 0) atomicOr(&a.flags[3], 1);
        return;
    }
    const int nb = (int)nb64;
    const Val *data = (const Val *)a.data;
    for (int r = tid; r < nb; r += kJoinThreads) {
        pids[r] = a.indices[bb + r];
        pval[r] = data[bb + r];
    }
    __syncthreads();
    const int64_t o = a.seg[j];
    const int k = a.k;
    for (int64_t t = tid; t < na; t += kJoinThreads) {
        const int32_t id = a.indices[ab + t];
        const Val va = data[ab + t];
        int lo = 0, hi = nb;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pids[mid] < id) lo = mid + 1;
            else hi = mid;
        }
        const bool hit = lo < nb && pids[lo] == id;
        const int64_t row = o + t;
        if (F64) {
            // the scipy expression computes (partner value or 0) + 1.0 - 1.0 in double, then casts (train.py:33,39-43)
            const double second = ((hit ? (double)pval[lo] : 0.0) + 1.0) - 1.0;
            a.out_xz[2 * row] = (float)va;
            a.out_xz[2 * row + 1] = (float)second;
        } else {
            int32_t pa = (int32_t)va, pb = hit ? (int32_t)pval[lo] : 0;
            if (a.out_idx) {
                a.out_idx[2 * row] = pa;
                a.out_idx[2 * row + 1] = pb;
            }
            if (a.out_xz) {
                if ((uint64_t)pa >= (uint64_t)a.table_rows || (uint64_t)pb >= (uint64_t)a.table_rows) {
                    atomicOr(&a.flags[3], 2);  // SFptr outside the table: never read out of bounds
                    pa = pb = 0;
                }
                float *dst = a.out_xz + row * 2 * k;
                const float *ta = a.table + (int64_t)pa * k, *tb = a.table + (int64_t)pb * k;
                for (int c = 0; c < k; ++c) dst[c] = ta[c];
                for (int c = 0; c < k; ++c) dst[k + c] = tb[c];
            }
        }
        if (a.out_segid) a.out_segid[row] = j;
    }
}

}  // namespace subgacc

using namespace subgacc;

extern "C" size_t subgacc_sjoin_workspace_bytes(int64_t S) {
    if (S < 0) S = 0;
    return align_up((size_t)S * 8, 256) + scan_workspace_bytes(S);
}

extern "C" int subgacc_sjoin_sizes(const int64_t *spg_indptr, const int64_t *own, int64_t S, int64_t *out_seg,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    SG_REQUIRE(S >= 0 && out_seg, SUBGACC_ERR_BADARG, "sjoin_sizes: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (S == 0) return exclusive_scan_i64(nullptr, 0, out_seg, nullptr, 0, s);
    SG_REQUIRE(spg_indptr && own, SUBGACC_ERR_BADARG, "sjoin_sizes: null argument");
    SG_REQUIRE(workspace && workspace_bytes >= subgacc_sjoin_workspace_bytes(S), SUBGACC_ERR_WORKSPACE,
               "sjoin_sizes: workspace too small");
    int64_t *len = (int64_t *)workspace;
    char *ws = (char *)workspace + align_up((size_t)S * 8, 256);
    hipLaunchKernelGGL(sjoin_len_kernel, dim3((unsigned)ceil_div(S, 256)), dim3(256), 0, s, spg_indptr, own, S, len);
    SG_LAUNCH_CHECK();
    return exclusive_scan_i64(len, S, out_seg, ws, workspace_bytes - align_up((size_t)S * 8, 256), s);
}

extern "C" int subgacc_sjoin_fill(const int64_t *spg_indptr, const int32_t *spg_indices, const int32_t *spg_data_i32,
                                  const double *spg_data_f64, const int64_t *own, const int64_t *partner, int64_t S,
                                  const int64_t *seg, const float *table, int64_t table_rows, int32_t k,
                                  float *out_xz, int32_t *out_idx, int64_t *out_segid, int32_t max_len,
                                  int32_t *flags, void *stream) {
    SG_REQUIRE(S >= 0 && max_len >= 0 && flags, SUBGACC_ERR_BADARG, "sjoin_fill: bad arguments");
    if (S == 0) return SUBGACC_OK;
    SG_REQUIRE(spg_indptr && spg_indices && own && partner && seg, SUBGACC_ERR_BADARG, "sjoin_fill: null argument");
    SG_REQUIRE((spg_data_i32 != nullptr) != (spg_data_f64 != nullptr), SUBGACC_ERR_BADARG,
               "sjoin_fill: exactly one of spg_data_i32 / spg_data_f64");
    const bool f64 = spg_data_f64 != nullptr;
    if (f64) {
        SG_REQUIRE(out_xz && !out_idx && !table, SUBGACC_ERR_BADARG,
                   "sjoin_fill: float payload writes out_xz [R,2,1] only (train.py:39-43)");
    } else {
        SG_REQUIRE(out_xz || out_idx, SUBGACC_ERR_BADARG, "sjoin_fill: no output requested");
        SG_REQUIRE(!out_xz || (table && table_rows > 0 && k > 0), SUBGACC_ERR_BADARG,
                   "sjoin_fill: out_xz needs the feature table");
    }
    JoinArgs a;
    a.indptr = spg_indptr, a.indices = spg_indices;
    a.data = f64 ? (const void *)spg_data_f64 : (const void *)spg_data_i32;
    a.own = own, a.partner = partner, a.seg = seg, a.S = S;
    a.table = table, a.table_rows = table_rows, a.k = k;
    a.out_xz = out_xz, a.out_idx = out_idx, a.out_segid = out_segid;
    a.max_len = max_len > 0 ? max_len : 1;
    a.flags = flags;
    const size_t lds = (size_t)a.max_len * (f64 ? 12 : 8);
    SG_REQUIRE(lds <= (size_t)kLdsBytes, SUBGACC_ERR_LDS, "sjoin_fill: rows of %d members do not fit LDS", max_len);
    const int64_t grid = xcd_grid(S);
    SG_REQUIRE(grid < (1ll << 31), SUBGACC_ERR_BADARG, "sjoin_fill: too many segments in one call");
    hipStream_t s = (hipStream_t)stream;
    if (f64) {
        if (lds > 64 * 1024)
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)sjoin_fill_kernel<true>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sjoin_fill_kernel<true>, dim3((unsigned)grid), dim3(kJoinThreads), lds, s, a);
    } else {
        if (lds > 64 * 1024)
            SG_CHECK_HIP(hipFuncSetAttribute((const void *)sjoin_fill_kernel<false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(sjoin_fill_kernel<false>, dim3((unsigned)grid), dim3(kJoinThreads), lds, s, a);
    }
    SG_LAUNCH_CHECK();
    return SUBGACC_OK;
}
