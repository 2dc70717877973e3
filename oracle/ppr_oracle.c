/*
 * oracle/ppr_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Sequential CPU restatement of the reference's top-K approximate-PPR set sampler
 * (SURVEY 8(f).3): sampler/pprgo.py:9-38 (_calc_ppr_node), :53-63 (calc_ppr_topk_parallel),
 * :66-82 (ppr_topk / construct_sparse), :85-111 (topk_ppr_matrix normalisation) and
 * utils.py:35-36 (encoding 'PPR').  Nothing under surel_plus_amd/ may import, link or call this file.
 *
 * Parity status: PARITY UNPINNED.  The reference function is numba-compiled and numba is not
 * in this image, the reference holds no test or golden vector for it, so this restatement could
 * not be run against the reference.  It follows the published algorithm line by line with
 * numba's documented typing of the expressions:
 *   alpha, epsilon               float32 arguments (pprgo.py:72)
 *   alpha_eps = alpha*epsilon    float32 * float32 -> float32
 *   p, r                         dict int64 -> float32 (first stores are float32)
 *   res, _val, res_vnode         float32 (the `locals=` of the decorator, :9)
 *   (1 - alpha) * res / deg[u]   int64 - float32 -> float64; * float32 -> float64;
 *                                / int64 -> float64; rounded to float32 on the store to _val
 *   res_vnode >= alpha_eps*deg   float32 * int64 -> float64, compared in float64
 *   q                            Python list used as a LIFO stack (q.pop()), membership `vnode not in q`
 * The push order (and with it every float32 rounding) is therefore fully determined; the one
 * thing numba leaves open is which of several EQUAL values np.argsort(val)[-topk:] keeps at the
 * cut (its quicksort is not stable).  This file keeps the later-inserted ones (stable ascending
 * sort, last topk) and so does the HIP path; tests/ also check the result against the defining
 * property of the approximation (0 <= ppr_exact - p <= eps-bound), which does not depend on any of this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
    int32_t *key; /* -1 = empty */
    float *r, *p;
    int32_t *pord; /* 0 = not in p, else 1-based insertion order into p */
    uint8_t *inq;
    uint32_t mask;
    int32_t *touched;
    int64_t ntouched;
} ppr_tab;

static int tab_init(ppr_tab *t, uint32_t cap)
{
    t->mask = cap - 1;
    t->key = malloc(sizeof(int32_t) * cap);
    t->r = calloc(cap, sizeof(float));
    t->p = calloc(cap, sizeof(float));
    t->pord = calloc(cap, sizeof(int32_t));
    t->inq = calloc(cap, 1);
    t->touched = malloc(sizeof(int32_t) * cap);
    t->ntouched = 0;
    if (!t->key || !t->r || !t->p || !t->pord || !t->inq || !t->touched)
        return -1;
    memset(t->key, 0xFF, sizeof(int32_t) * cap);
    return 0;
}
static void tab_free(ppr_tab *t)
{
    free(t->key), free(t->r), free(t->p), free(t->pord), free(t->inq), free(t->touched);
}
static void tab_reset(ppr_tab *t)
{
    for (int64_t x = 0; x < t->ntouched; ++x)
    {
        int32_t s = t->touched[x];
        t->key[s] = -1, t->r[s] = 0.f, t->p[s] = 0.f, t->pord[s] = 0, t->inq[s] = 0;
    }
    t->ntouched = 0;
}
/* slot of `node`, inserted when absent; -1 when the table is half full */
static int32_t tab_slot(ppr_tab *t, int32_t node)
{
    uint32_t h = ((uint32_t)node * 2654435761u) & t->mask;
    while (t->key[h] != -1 && t->key[h] != node)
        h = (h + 1) & t->mask;
    if (t->key[h] == -1)
    {
        if ((uint64_t)t->ntouched * 2 >= (uint64_t)t->mask + 1)
            return -1;
        t->key[h] = node;
        t->touched[t->ntouched++] = (int32_t)h;
    }
    return (int32_t)h;
}

typedef struct
{
    float val;
    int32_t ord, id;
} ppr_ent;
static int ent_cmp_val(const void *a, const void *b)
{
    const ppr_ent *x = a, *y = b;
    if (x->val != y->val)
        return x->val < y->val ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord);
}
static int ent_cmp_id(const void *a, const void *b)
{
    const ppr_ent *x = a, *y = b;
    return x->id < y->id ? -1 : (x->id > y->id);
}

/* pprgo.py:9-38 for every root + :59-61 (top-k) ; rows leave sorted by node id (coo->csr, :66-82,:87).
 * out_ids/out_vals are [n*topk]; out_count[i] = entries of row i.  Returns 0, -1 (memory) or -2 (table). */
int orc_ppr_topk(const int64_t *indptr, const int32_t *indices, const int32_t *roots, int64_t n, float alpha,
                 float epsilon, int32_t topk, int32_t table_log2, int32_t *out_count, int32_t *out_ids, float *out_vals,
                 int64_t *out_pushes)
{
    ppr_tab t;
    const uint32_t cap = 1u << table_log2;
    if (tab_init(&t, cap))
        return -1;
    int32_t *stack = malloc(sizeof(int32_t) * cap);
    ppr_ent *ent = malloc(sizeof(ppr_ent) * cap);
    if (!stack || !ent)
        return -1;
    const float alpha_eps = alpha * epsilon; /* float32 product */
    int rc = 0;
    int64_t pushes = 0;
    for (int64_t i = 0; i < n && rc == 0; ++i)
    {
        const int32_t inode = roots[i];
        int32_t np = 0;
        int64_t ns = 0;
        int32_t s0 = tab_slot(&t, inode);
        t.p[s0] = 0.f, t.pord[s0] = ++np; /* p = {inode: 0} */
        t.r[s0] = alpha;                  /* r[inode] = alpha */
        stack[ns++] = s0, t.inq[s0] = 1;  /* q = [inode] */
        while (ns > 0)
        {
            const int32_t su = stack[--ns]; /* q.pop() */
            t.inq[su] = 0;
            const int32_t unode = t.key[su];
            const float res = t.r[su];
            if (!t.pord[su])
                t.pord[su] = ++np;
            t.p[su] += res;
            t.r[su] = 0.f;
            ++pushes;
            const int64_t b = indptr[unode], e = indptr[unode + 1];
            const int64_t degu = e - b;
            for (int64_t j = b; j < e; ++j)
            {
                const int32_t vnode = indices[j];
                const float val = (float)((1.0 - (double)alpha) * (double)res / (double)degu);
                const int32_t sv = tab_slot(&t, vnode);
                if (sv < 0)
                {
                    rc = -2;
                    break;
                }
                t.r[sv] += val;
                const float res_v = t.r[sv];
                const int64_t degv = indptr[vnode + 1] - indptr[vnode];
                if ((double)res_v >= (double)alpha_eps * (double)degv && !t.inq[sv])
                    stack[ns++] = sv, t.inq[sv] = 1;
            }
            if (rc)
                break;
        }
        if (rc == 0)
        {
            int64_t m = 0;
            for (int64_t x = 0; x < t.ntouched; ++x)
            {
                const int32_t s = t.touched[x];
                if (t.pord[s])
                    ent[m].val = t.p[s], ent[m].ord = t.pord[s], ent[m].id = t.key[s], ++m;
            }
            qsort(ent, (size_t)m, sizeof(ppr_ent), ent_cmp_val); /* ascending, ties by insertion order */
            const int64_t keep = m < topk ? m : topk;
            ppr_ent *top = ent + (m - keep); /* [-topk:] */
            qsort(top, (size_t)keep, sizeof(ppr_ent), ent_cmp_id);
            out_count[i] = (int32_t)keep;
            for (int64_t x = 0; x < keep; ++x)
                out_ids[i * topk + x] = top[x].id, out_vals[i * topk + x] = top[x].val;
        }
        tab_reset(&t);
    }
    if (out_pushes)
        *out_pushes = pushes;
    free(stack), free(ent);
    tab_free(&t);
    return rc;
}

/* pprgo.py:88-108: 'sym'  data = deg_sqrt[idx[row]] * data * deg_inv_sqrt[col]   (float64)
 *                  'col'  data = deg[idx[row]] * data * (1/max(deg,1e-12))[col]
 *                  'row'  unchanged (float32 values widened)
 * deg = adj.sum(1) of an unweighted graph = the row length. mode: 0 row, 1 sym, 2 col. */
void orc_ppr_normalize(const int64_t *indptr, const int32_t *roots, int64_t n, const int64_t *row_off,
                       const int32_t *ids, const float *vals, int mode, double *out)
{
    for (int64_t i = 0; i < n; ++i)
    {
        const double dr = (double)(indptr[roots[i] + 1] - indptr[roots[i]]);
        for (int64_t x = row_off[i]; x < row_off[i + 1]; ++x)
        {
            const double dc = (double)(indptr[ids[x] + 1] - indptr[ids[x]]);
            const double v = (double)vals[x];
            if (mode == 1)
            {
                const double sr = sqrt(fmax(dr, 1e-12)), sc = sqrt(fmax(dc, 1e-12));
                out[x] = sr * v * (1.0 / sc);
            }
            else if (mode == 2)
                out[x] = dr * v * (1.0 / fmax(dc, 1e-12));
            else
                out[x] = v;
        }
    }
}

/* utils.py:35-36: x.data = (x.data + 0.1) / (x.data.max() + 0.1) */
void orc_ppr_encode(double *data, int64_t nnz)
{
    if (nnz == 0)
        return;
    double mx = data[0];
    for (int64_t x = 1; x < nnz; ++x)
        mx = data[x] > mx ? data[x] : mx;
    for (int64_t x = 0; x < nnz; ++x)
        data[x] = (data[x] + 0.1) / (mx + 0.1);
}
