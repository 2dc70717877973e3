/* Stand-alone C client of include/subgacc.h -- no Python, no torch: proves that the shared library is a plain
 * C-ABI drop-in (what a cgo / JNI / ctypes / CPython-module binding would call).  Built and run by
 * tests/test_gpu_cabi_c.py on the GPU box:
 *     gcc cabi_smoke.c -I../../include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64 -ldl
 * It samples node sets on a 6-node graph with subgacc_walk_sets (Philox), compacts them, numbers the LP rows,
 * builds the SpG and joins one pair, and checks the invariants of subg_acc/test/test.py:34-45 on the result. */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "subgacc.h"

#define CHECK(x)                                                        \
    do {                                                                \
        if (!(x)) {                                                     \
            fprintf(stderr, "FAILED %s (line %d)\n", #x, __LINE__);     \
            return 1;                                                   \
        }                                                               \
    } while (0)
#define HIP(x) CHECK((x) == hipSuccess)

static void *dev(size_t bytes, const void *src) {
    void *p = NULL;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return NULL;
    if (src) hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
    else hipMemset(p, 0, bytes ? bytes : 16);
    return p;
}

int main(int argc, char **argv) {
    void *lib = dlopen(argc > 1 ? argv[1] : "libsubgacc_hip.so", RTLD_NOW);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); CHECK(p_##name != NULL)
    SYM(subgacc_abi_version); SYM(subgacc_device_count); SYM(subgacc_last_error); SYM(subgacc_key_shift);
    SYM(subgacc_walk_sets); SYM(subgacc_scan_workspace_bytes); SYM(subgacc_exclusive_scan_i32); SYM(subgacc_compact_sets);
    SYM(subgacc_uniq_table_bytes); SYM(subgacc_uniq_reset); SYM(subgacc_uniq_number_workspace_bytes);
    SYM(subgacc_uniq_number); SYM(subgacc_spg_build); SYM(subgacc_sjoin_workspace_bytes); SYM(subgacc_sjoin_sizes);
    SYM(subgacc_sjoin_fill_v2); SYM(subgacc_unpack_lp); SYM(subgacc_rows_to_headed);
    SYM(subgacc_step_prologue); SYM(subgacc_walk_keyrows64); SYM(subgacc_walk_spg); SYM(subgacc_sjoin_sizes_rows);

    CHECK(p_subgacc_abi_version() == SUBGACC_ABI_VERSION);
    CHECK(p_subgacc_device_count() >= 1);
    CHECK(p_subgacc_key_shift(8, 2) == 4);
    CHECK(p_subgacc_key_shift(200, 9) == SUBGACC_ERR_KEYWIDTH && strlen(p_subgacc_last_error()) > 0);

    /* ring of 6 nodes + chord 0-3, symmetric, sorted */
    const int32_t indptr_h[7] = {0, 3, 5, 7, 10, 12, 14};
    const int32_t indices_h[14] = {1, 3, 5, 0, 2, 1, 3, 0, 2, 4, 3, 5, 0, 4};
    const int32_t query_h[4] = {0, 3, 3, 5};
    enum { N = 6, n = 4, M = 8, m = 2, STRIDE = M * m + 1, CAP = 1024 };
    int32_t *indptr = dev(sizeof indptr_h, indptr_h), *indices = dev(sizeof indices_h, indices_h);
    int32_t *query = dev(sizeof query_h, query_h);
    int32_t *st_ids = dev(n * STRIDE * 4, NULL), *nsize = dev(n * 4, NULL), *flags = dev(16, NULL);
    uint64_t *st_keys = dev(n * STRIDE * 8, NULL);
    subgacc_walk_cfg cfg = {M, m, -1, SUBGACC_RNG_PHILOX, 7u, 1, SUBGACC_ORDER_WALK_MAJOR, 1, 0, 0};
    CHECK(p_subgacc_walk_sets(&cfg, indptr, indices, N, query, n, NULL, NULL, st_ids, st_keys, nsize, NULL, flags, NULL) == 0);
    size_t wsb = p_subgacc_scan_workspace_bytes(n);
    void *ws = dev(wsb, NULL);
    int64_t *row_off = dev((n + 1) * 8, NULL);
    CHECK(p_subgacc_exclusive_scan_i32(nsize, n, row_off, ws, wsb, NULL) == 0);
    int64_t off_h[n + 1];
    HIP(hipMemcpy(off_h, row_off, sizeof off_h, hipMemcpyDeviceToHost));
    const int64_t X = off_h[n];
    CHECK(X >= n && X <= n * STRIDE);

    void *table = dev(p_subgacc_uniq_table_bytes(CAP), NULL);
    CHECK(p_subgacc_uniq_reset(table, CAP, NULL) == 0);
    int32_t *ids = dev(X * 4, NULL), *slot = dev(X * 4, NULL);
    CHECK(p_subgacc_compact_sets(st_ids, st_keys, nsize, row_off, n, STRIDE, ids, NULL, table, CAP, 0, slot, flags, NULL) == 0);
    uint64_t *ukeys = dev(CAP * 8, NULL);
    int64_t *count = dev(8, NULL);
    size_t nwb = p_subgacc_uniq_number_workspace_bytes(CAP, X);
    void *nws = dev(nwb, NULL);
    CHECK(p_subgacc_uniq_number(table, CAP, slot, X, ukeys, CAP, count, 0, nws, nwb, NULL) == 0);
    int64_t c = 0;
    HIP(hipMemcpy(&c, count, 8, hipMemcpyDeviceToHost));
    CHECK(c >= 1 && c <= X);
    int32_t *z_idx = dev(X * 4, NULL), *z_dat = dev(X * 4, NULL);
    CHECK(p_subgacc_spg_build(row_off, n, ids, slot, table, CAP, STRIDE, z_idx, z_dat, flags, NULL) == 0);
    float *tab = dev((c + 1) * (m + 1) * 4, NULL);
    CHECK(p_subgacc_unpack_lp(ukeys, c, NULL, M, m, NULL, NULL, tab, 1, NULL) == 0);

    /* join SpG rows (0,1) and (2,3): rows 1 and 2 are both the set of node 3 */
    const int64_t own_h[4] = {0, 2, 1, 3}, partner_h[4] = {1, 3, 0, 2};
    int64_t *own = dev(sizeof own_h, own_h), *partner = dev(sizeof partner_h, partner_h), *seg = dev(5 * 8, NULL);
    size_t jwb = p_subgacc_sjoin_workspace_bytes(4);
    void *jws = dev(jwb, NULL);
    CHECK(p_subgacc_sjoin_sizes(row_off, n, own, partner, 4, seg, flags, jws, jwb, NULL) == 0);
    int64_t seg_h[5];
    HIP(hipMemcpy(seg_h, seg, sizeof seg_h, hipMemcpyDeviceToHost));
    const int64_t R = seg_h[4];
    float *xz = dev(R * 2 * (m + 1) * 4, NULL);
    /* ABI 6: every form of the join goes through ONE entry point; the descriptor says what the store and the outputs are */
    {
        subgacc_join_desc d;
        memset(&d, 0, sizeof d);
        d.struct_bytes = (int32_t)sizeof d, d.form = SUBGACC_JOIN_ROWS, d.payload_kind = SUBGACC_JOIN_SFPTR;
        d.row_off = row_off, d.n_rows = n, d.ids = z_idx, d.payload = z_dat, d.max_len = STRIDE;
        d.own = own, d.partner = partner, d.S = 4, d.seg = seg, d.pair_block = 2;
        d.table = tab, d.table_rows = c + 1, d.k = m + 1, d.out_xz = xz, d.flags = flags;
        CHECK(p_subgacc_sjoin_fill_v2(&d, NULL) == 0);
        /* the same join as ONE call (a serving loop's form): the size pass runs first, as a single launch, then the fill; the row
         * count and the status word arrive in pinned host memory by themselves.  size_state: zeroed once (dev() does), left clean */
        int64_t *seg1 = dev(5 * 8, NULL), *tail = NULL, seg1_h[5];
        float *xz1 = dev(4 * STRIDE * 2 * (m + 1) * 4, NULL), *a_h = malloc(R * 2 * (m + 1) * 4), *b_h = malloc(R * 2 * (m + 1) * 4);
        void *state = dev(jwb, NULL);
        HIP(hipHostMalloc((void **)&tail, 16, 0));
        tail[0] = tail[1] = -7;
        d.seg = NULL, d.options = SUBGACC_JOIN_OPT_SIZES, d.out_seg = seg1, d.size_state = state, d.size_state_bytes = (int64_t)jwb;
        d.host_tail = tail, d.out_xz = xz1;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(p_subgacc_sjoin_fill_v2(&d, NULL) == 0);
            HIP(hipDeviceSynchronize());
            CHECK(tail[0] == R && tail[1] == 0);
            HIP(hipMemcpy(seg1_h, seg1, sizeof seg1_h, hipMemcpyDeviceToHost));
            CHECK(memcmp(seg1_h, seg_h, sizeof seg_h) == 0);
            HIP(hipMemcpy(a_h, xz, R * 2 * (m + 1) * 4, hipMemcpyDeviceToHost));
            HIP(hipMemcpy(b_h, xz1, R * 2 * (m + 1) * 4, hipMemcpyDeviceToHost));
            CHECK(memcmp(a_h, b_h, R * 2 * (m + 1) * 4) == 0);
            tail[0] = -7;
        }
        d.size_state_bytes = 8;                              /* a state too small is refused */
        CHECK(p_subgacc_sjoin_fill_v2(&d, NULL) == SUBGACC_ERR_WORKSPACE);
        d.struct_bytes = 8;                                  /* a descriptor of another layout is refused, not misread */
        CHECK(p_subgacc_sjoin_fill_v2(&d, NULL) == SUBGACC_ERR_BADARG);
        free(a_h), free(b_h);
    }
    HIP(hipDeviceSynchronize());

    /* checks on the host */
    int32_t fl[4], ns_h[n], *ids_h = malloc(X * 4), *zi = malloc(X * 4), *zd = malloc(X * 4);
    float *xz_h = malloc(R * 2 * (m + 1) * 4), *tab_h = malloc((c + 1) * (m + 1) * 4);
    HIP(hipMemcpy(fl, flags, 16, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ns_h, nsize, sizeof ns_h, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(ids_h, ids, X * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(zi, z_idx, X * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(zd, z_dat, X * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(xz_h, xz, R * 2 * (m + 1) * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(tab_h, tab, (c + 1) * (m + 1) * 4, hipMemcpyDeviceToHost));
    CHECK(fl[0] == 0 && fl[1] == 0 && fl[2] == 0 && fl[3] == 0);
    for (int i = 0; i < n; ++i) {
        CHECK(off_h[i + 1] - off_h[i] == ns_h[i] && ns_h[i] >= 1);
        CHECK(ids_h[off_h[i]] == query_h[i]);                                   /* the root is member 0 */
        float col[m + 1];
        memset(col, 0, sizeof col);
        for (int64_t e = off_h[i]; e < off_h[i + 1]; ++e) {
            if (e > off_h[i]) CHECK(zi[e] > zi[e - 1]);                         /* SpG row sorted, distinct */
            CHECK(zd[e] >= 1 && zd[e] <= c);
            for (int j = 0; j <= m; ++j) col[j] += tab_h[zd[e] * (m + 1) + j];
        }
        for (int j = 0; j <= m; ++j) CHECK(col[j] > 0.999f && col[j] < 1.001f);  /* every LP column sums to M (/M) */
    }
    /* the same join from the store re-keyed on the host (payload = the member's LP key = the low word of its row's 64-bit key):
     * subgacc_sjoin_fill_keys unpacks the feature rows itself -- bit for bit the table join's xz */
    {
        uint64_t *uk_h = malloc(c * 8);
        int32_t *zk_h = malloc(X * 4);
        HIP(hipMemcpy(uk_h, ukeys, c * 8, hipMemcpyDeviceToHost));
        for (int64_t e = 0; e < X; ++e) zk_h[e] = (int32_t)(uint32_t)uk_h[zd[e] - 1];
        int32_t *z_key = dev(X * 4, zk_h);
        float *xz2 = dev(R * 2 * (m + 1) * 4, NULL), *xz2_h = malloc(R * 2 * (m + 1) * 4);
        subgacc_join_desc d;
        memset(&d, 0, sizeof d);
        d.struct_bytes = (int32_t)sizeof d, d.form = SUBGACC_JOIN_ROWS, d.payload_kind = SUBGACC_JOIN_KEY32;
        d.row_off = row_off, d.n_rows = n, d.ids = z_idx, d.payload = z_key, d.max_len = STRIDE;
        d.own = own, d.partner = partner, d.S = 4, d.seg = seg, d.pair_block = 2;
        d.num_walks = M, d.num_steps = m, d.out_xz = xz2, d.flags = flags;
        CHECK(p_subgacc_sjoin_fill_v2(&d, NULL) == 0);
        HIP(hipDeviceSynchronize());
        HIP(hipMemcpy(xz2_h, xz2, R * 2 * (m + 1) * 4, hipMemcpyDeviceToHost));
        CHECK(memcmp(xz_h, xz2_h, R * 2 * (m + 1) * 4) == 0);
        /* ABI 7: the same store laid out on whole 128-byte lines -- HEADED rows: a row's length in slot 0 of its ids, no row pointers --
         * and joined in ONE call (SUBGACC_JOIN_OPT_SIZES: size pass + fill; [R, status] land in pinned host memory): what a C serving
         * loop over a resident store does per batch */
        {
            enum { PITCH = 32 };                                                    /* >= STRIDE + 1, a multiple of 32 words */
            int32_t *h_ids = dev(n * PITCH * 4, NULL), *h_key = dev(n * PITCH * 4, NULL);
            CHECK(p_subgacc_rows_to_headed(row_off, n, z_idx, z_key, 4, PITCH, h_ids, h_key, flags, NULL) == 0);
            const size_t sbytes = p_subgacc_sjoin_workspace_bytes(4);
            void *state = dev(sbytes, NULL);                                        /* zeroed ONCE */
            int64_t *seg3 = dev(5 * 8, NULL), *tail = NULL, seg3_h[5];
            HIP(hipHostMalloc((void **)&tail, 16, 0));
            const size_t cap = (size_t)4 * (PITCH - 1) * 2 * (m + 1) * 4;            /* the worst case: R is not known beforehand */
            float *xz3 = dev(cap, NULL);
            subgacc_join_desc h;
            memset(&h, 0, sizeof h);
            h.struct_bytes = (int32_t)sizeof h, h.form = SUBGACC_JOIN_ROWS, h.payload_kind = SUBGACC_JOIN_KEY32;
            h.row_stride = PITCH, h.n_rows = n, h.ids = h_ids, h.payload = h_key;
            h.own = own, h.partner = partner, h.S = 4, h.pair_block = 2, h.num_walks = M, h.num_steps = m;
            h.options = SUBGACC_JOIN_OPT_SIZES, h.out_seg = seg3, h.size_state = state, h.size_state_bytes = (int64_t)sbytes, h.host_tail = tail;
            h.flags = flags;
            for (int rep = 0; rep < 2; ++rep) {                                     /* (the state cleans itself: call after call) */
                tail[0] = tail[1] = -7;
                h.out_xz = rep ? xz3 : NULL;                                        /* first the size pass alone, then the whole join */
                CHECK(p_subgacc_sjoin_fill_v2(&h, NULL) == 0);
                HIP(hipDeviceSynchronize());
                CHECK(tail[0] == R && tail[1] == 0);
            }
            HIP(hipMemcpy(seg3_h, seg3, sizeof seg3_h, hipMemcpyDeviceToHost));
            CHECK(memcmp(seg3_h, seg_h, sizeof seg3_h) == 0);
            HIP(hipMemcpy(xz2_h, xz3, R * 2 * (m + 1) * 4, hipMemcpyDeviceToHost));
            CHECK(memcmp(xz_h, xz2_h, R * 2 * (m + 1) * 4) == 0);
        }
        HIP(hipMemcpy(fl, flags, 16, hipMemcpyDeviceToHost));
        CHECK(fl[3] == 0);
    }
    /* ABI 5: the on-demand step with KEY ROWS, as a C serving loop runs it -- endpoints of two pairs -> prologue (int64 -> int32 roots,
     * status zeroed) -> fused-row walk whose rows carry the members' LP keys (4 hops: M = 200 needs 33 bits -> subgacc_walk_keyrows64;
     * M = 100 fits 32 -> subgacc_walk_spg without a table) -> segment sizes -> join that unpacks the keys itself.  Checked here: the
     * reference's invariants (subg_acc/test/test.py:34-45) on rows and on xz; parity with the oracle is the Python suite's job. */
    for (int wide = 0; wide < 2; ++wide) {
        enum { M4 = 0, m4 = 4 };
        const int Mw = wide ? 200 : 100, S4 = Mw * m4 + 1, sh = wide ? 8 : 7;
        const int64_t edge_h[4] = {0, 3, 3, 5};                                  /* pairs (0,3) and (3,5): [u0 u1 | v0 v1] */
        int64_t *edge = dev(sizeof edge_h, edge_h), *status = dev(32, NULL), *seg2 = dev(5 * 8, NULL);
        int32_t *roots = dev(4 * 4, NULL), *rid = dev(4 * S4 * 4, NULL), *ns2 = dev(4 * 4, NULL);
        void *rkey = dev((size_t)4 * S4 * 8, NULL);
        subgacc_walk_cfg c4 = {Mw, m4, -1, SUBGACC_RNG_PHILOX, 7u, 1, SUBGACC_ORDER_WALK_MAJOR, 1, 0, 0};
        CHECK(p_subgacc_step_prologue(NULL, 0, status, 4, edge, roots, 4, NULL) == 0);
        if (wide)
            CHECK(p_subgacc_walk_keyrows64(&c4, indptr, indices, N, roots, 4, NULL, NULL, NULL, NULL, rid, (uint64_t *)rkey, ns2,
                                           (int32_t *)status, NULL) == 0);
        else
            CHECK(p_subgacc_walk_spg(&c4, indptr, indices, N, roots, 4, 0, NULL, NULL, NULL, 0, rid, (int32_t *)rkey, ns2,
                                     (int32_t *)status, NULL) == 0);
        const int64_t own2_h[4] = {0, 1, 2, 3};                                   /* row i = endpoint i; partner = the mirror block */
        int64_t *own2 = dev(sizeof own2_h, own2_h);
        CHECK(p_subgacc_sjoin_sizes_rows(ns2, 4, own2, NULL, 4, seg2, (int32_t *)status, jws, jwb, NULL) == 0);
        int64_t s2[5];
        HIP(hipMemcpy(s2, seg2, sizeof s2, hipMemcpyDeviceToHost));
        const int64_t R2 = s2[4];
        float *xzk = dev(R2 * 2 * (m4 + 1) * 4, NULL), *xzk_h = malloc(R2 * 2 * (m4 + 1) * 4);
        {
            subgacc_join_desc d;
            memset(&d, 0, sizeof d);
            d.struct_bytes = (int32_t)sizeof d, d.form = SUBGACC_JOIN_ROWS, d.payload_kind = wide ? SUBGACC_JOIN_KEY64 : SUBGACC_JOIN_KEY32;
            d.row_len = ns2, d.n_rows = 4, d.row_stride = S4, d.ids = rid, d.payload = rkey;
            d.own = own2, d.S = 4, d.seg = seg2, d.pair_block = 2;             /* partner = NULL: the mirror block */
            d.num_walks = Mw, d.num_steps = m4, d.out_xz = xzk, d.flags = (int32_t *)status;
            CHECK(p_subgacc_sjoin_fill_v2(&d, NULL) == 0);
        }
        HIP(hipDeviceSynchronize());
        int32_t nsh[4], st32[8], *rid_h = malloc(4 * S4 * 4);
        uint64_t *rk_h = malloc((size_t)4 * S4 * 8);
        HIP(hipMemcpy(nsh, ns2, sizeof nsh, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(st32, status, 32, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(rid_h, rid, 4 * S4 * 4, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(rk_h, rkey, (size_t)4 * S4 * (wide ? 8 : 4), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(xzk_h, xzk, R2 * 2 * (m4 + 1) * 4, hipMemcpyDeviceToHost));
        CHECK(st32[0] == 0 && st32[1] == 0 && st32[2] == 0 && st32[3] == 0);
        for (int i = 0; i < 4; ++i) {
            CHECK(nsh[i] >= 1 && nsh[i] <= N && s2[i + 1] - s2[i] == nsh[i]);
            long col[m4 + 1];
            memset(col, 0, sizeof col);
            int has_root = 0;
            for (int e = 0; e < nsh[i]; ++e) {
                const int32_t id = rid_h[i * S4 + e];
                const uint64_t key = wide ? rk_h[(size_t)i * S4 + e] : (uint64_t)((const uint32_t *)rk_h)[(size_t)i * S4 + e];
                if (e) CHECK(id > rid_h[i * S4 + e - 1]);                        /* sorted, distinct */
                const int lead = (int)((key >> (m4 * sh)) & 1u);
                CHECK(lead == (id == (int32_t)edge_h[i]));                       /* the root's row carries the LEAD bit, nobody else */
                has_root |= lead;
                for (int j = 1; j <= m4; ++j) col[j] += (long)((key >> ((m4 - j) * sh)) & ((1u << sh) - 1u));
            }
            CHECK(has_root);
            for (int j = 1; j <= m4; ++j) CHECK(col[j] == Mw);                   /* every landing-count column sums to M */
            float fc[m4 + 1];                                                    /* ... and so do the own-slot feature rows of xz (/M) */
            memset(fc, 0, sizeof fc);
            for (int64_t r = s2[i]; r < s2[i + 1]; ++r)
                for (int j = 0; j <= m4; ++j) fc[j] += xzk_h[(r * 2 + 0) * (m4 + 1) + j];
            CHECK(fc[0] > 0.999f && fc[0] < 1.001f);
            for (int j = 1; j <= m4; ++j) CHECK(fc[j] > 0.999f && fc[j] < 1.001f);
        }
        CHECK(nsh[1] == nsh[2]);                                                 /* endpoints 1 and 2 are both node 3: the same set */
        for (int e = 0; e < nsh[1]; ++e) CHECK(rid_h[1 * S4 + e] == rid_h[2 * S4 + e]);
        printf("cabi_smoke key rows (%d-bit keys): M=%d m=%d R=%lld ok\n", wide ? 64 : 32, Mw, m4, (long long)R2);
    }
    /* rows 1 and 2 are the same set (Philox is keyed by the root id): joined with each other both slots agree */
    CHECK(ns_h[1] == ns_h[2]);
    CHECK(seg_h[2] - seg_h[1] == ns_h[2] && seg_h[4] - seg_h[3] == ns_h[3]);
    for (int64_t r = seg_h[0]; r < seg_h[1]; ++r) (void)r;
    printf("cabi_smoke ok: X=%lld c=%lld R=%lld\n", (long long)X, (long long)c, (long long)R);
    return 0;
}
