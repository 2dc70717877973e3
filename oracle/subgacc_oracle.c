/*
 * oracle/subgacc_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain, sequential CPU restatement of the SubGAcc hot path of SUREL+
 * (Graph-COM/SUREL_Plus).  It exists only so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check / time the HIP path against it.
 * Nothing under surel_plus_amd/ may import, link or call this file.
 *
 * Parity status: PINNED.  In RNG mode ORC_RNG_RAND_R this restatement is checked
 * bit-for-bit against (a) the golden vectors in tests/golden/ generated from the
 * compiled reference (oracle/gen_golden.py) and (b), when oracle/_ref exists,
 * the compiled reference itself (tests/test_oracle_vs_ref.py).
 *
 * What is restated (reference file:line, relative to /root/reference):
 *   orc_rand_r            glibc 2.35 stdlib/rand_r.c (third-party; the reference calls it at
 *                         subg_acc/subg_acc.c:173,209,240,771,807)
 *   orc_gset_sampler      subg_acc/subg_acc.c:649-1034 (set_sampler)
 *   orc_walk_sampler      subg_acc/subg_acc.c:144-180 (random_walk), :183-247 (random_walk_wo),
 *                         :249-314 (rpe_encoder), :316-389 (walk_sampler)
 *   orc_spg_build         sampler/random_walks.py:77-81 (subg_matrix; scipy COO->CSR + sort)
 *   orc_sjoin             train.py:13-45 (gather), :48-72 (hgather), :75-111 (bgather/pgather)
 *
 * Deliberate deviations from the reference (documented in DESIGN.md):
 *   - an isolated root (degree 0) emits its own id as the single set member; the reference
 *     leaves that id uninitialised (subg_acc.c:753-761 vs :838-844).
 *   - a second RNG mode (ORC_RNG_PHILOX, counter-based Philox2x32-10: key = seed,
 *     counter = (root id, walk | draw block | stream tag); a draw r picks neighbour
 *     (r * degree) >> 32) that the reference does not have; it is the
 *     schedule-independent mode of the HIP path and has to be checkable too.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_RNG_RAND_R 0
#define ORC_RNG_PHILOX 1
#define ORC_NEIGH_CAP 1000000 /* NEBMAX, subg_acc.c:13,750 */

/* ------------------------------------------------------------------ RNGs */

/* glibc rand_r: three rounds of x = x*1103515245+12345, taking 11+10+10 bits. */
uint32_t orc_rand_r(uint32_t *state)
{
    uint32_t x = *state, r;
    x = x * 1103515245u + 12345u;
    r = (x >> 16) & 2047u;
    x = x * 1103515245u + 12345u;
    r = (r << 10) ^ ((x >> 16) & 1023u);
    x = x * 1103515245u + 12345u;
    r = (r << 10) ^ ((x >> 16) & 1023u);
    *state = x;
    return r;
}

/* Philox4x32-10 (Salmon et al., SC'11), published constants. */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round)
    {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

/* Philox2x32-10 (same paper): one 32x32 multiply per round; two draws per call -- what a walk of up to three
 * hops needs after its first (deterministic) hop. */
static void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t k, uint32_t out[2])
{
    for (int round = 0; round < 10; ++round)
    {
        uint64_t p = (uint64_t)0xD256D193u * c0;
        uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
        c1 = (uint32_t)p;
        c0 = n0;
        k += 0x9E3779B9u;
    }
    out[0] = c0, out[1] = c1;
}

/* Stream layout of the Philox mode (shared with csrc/walk_common.hpp): key = seed, counter word 0 = root id,
 * counter word 1 = lane (walk number or shuffle position, < 2^20) | block << 20 (draws 2*block, 2*block+1) |
 * tag << 28 | shuffle-stream flag << 31. */
#define ORC_PHILOX_BLOCK_SHIFT 20
#define ORC_PHILOX_TAG_SHIFT 28
#define ORC_PHILOX_SHUFFLE 0x80000000u

/* draw #idx (0-based) of the stream (root, lane, tag) */
static uint32_t philox_draw(uint32_t seed, uint32_t root, uint32_t lane, uint32_t idx, uint32_t tag)
{
    uint32_t out[2];
    philox2x32_10(root, lane | ((idx >> 1) << ORC_PHILOX_BLOCK_SHIFT) | (tag << ORC_PHILOX_TAG_SHIFT), seed, out);
    return out[idx & 1u];
}
static uint32_t philox_shuffle_draw(uint32_t seed, uint32_t root, uint32_t k, uint32_t tag)
{
    uint32_t out[2];
    philox2x32_10(root, k | (tag << ORC_PHILOX_TAG_SHIFT) | ORC_PHILOX_SHUFFLE, seed, out);
    return out[0];
}
/* a draw -> an index below n: the high word of r*n (one multiply on the GPU; the rand_r mode keeps the reference's %) */
static inline uint32_t philox_below(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

/* exposed for the RNG known-answer tests */
void orc_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }
void orc_philox2x32_10(const uint32_t *ctr, uint32_t key, uint32_t *out) { philox2x32_10(ctr[0], ctr[1], key, out); }
void orc_rand_r_stream(uint32_t seed, int64_t count, uint32_t *out)
{
    for (int64_t i = 0; i < count; ++i)
        out[i] = orc_rand_r(&seed);
}

/* ------------------------------------------------------- small containers */

typedef struct
{
    int32_t *key;
    int32_t *val;
    uint32_t mask;
} nodeset_t; /* open addressing, key -1 = empty */

static int nodeset_init(nodeset_t *s, int64_t max_items)
{
    uint32_t cap = 16;
    while ((int64_t)cap < 2 * (max_items + 1))
        cap <<= 1;
    s->key = (int32_t *)malloc(sizeof(int32_t) * cap);
    s->val = (int32_t *)malloc(sizeof(int32_t) * cap);
    s->mask = cap - 1;
    return (s->key && s->val) ? 0 : -1;
}
static void nodeset_clear(nodeset_t *s) { memset(s->key, 0xFF, sizeof(int32_t) * ((size_t)s->mask + 1)); }
static void nodeset_free(nodeset_t *s)
{
    free(s->key);
    free(s->val);
}
/* returns slot index in table for `key` (existing or empty) */
static uint32_t nodeset_probe(const nodeset_t *s, int32_t key)
{
    uint32_t h = ((uint32_t)key * 2654435761u) & s->mask;
    while (s->key[h] != -1 && s->key[h] != key)
        h = (h + 1) & s->mask;
    return h;
}

typedef struct
{
    int32_t *nsize;  /* [n]            */
    int32_t *ids;    /* [X] node ids, per-root first-visit order  */
    int16_t *counts; /* [X*ncol] raw landing counts, col 0 = M on the root row */
    int64_t X, cap;
} setbuf_t;

static int setbuf_reserve(setbuf_t *b, int64_t need, int ncol)
{
    if (need <= b->cap)
        return 0;
    int64_t cap = b->cap ? b->cap : 1024;
    while (cap < need)
        cap += cap / 2 + 1;
    int32_t *ids = (int32_t *)realloc(b->ids, sizeof(int32_t) * cap);
    if (!ids)
        return -1;
    b->ids = ids;
    int16_t *cnt = (int16_t *)realloc(b->counts, sizeof(int16_t) * cap * ncol);
    if (!cnt)
        return -1;
    b->counts = cnt;
    b->cap = cap;
    return 0;
}

/* ---------------------------------------------------- first hop selection */

/* The reference draws the first hop without replacement by a partial Fisher-Yates over
 * [0,deg) (subg_acc.c:763-776): for k<M: s = draw % (deg-k) + k; swap(perm[k], perm[s]).
 * Only perm[0..M) is read afterwards, so a sparse map of displaced entries is enough. */
typedef struct
{
    nodeset_t moved; /* position -> value, for positions whose content differs from identity */
} sparse_perm_t;

static int32_t perm_get(const sparse_perm_t *p, int32_t pos)
{
    uint32_t h = nodeset_probe(&p->moved, pos);
    return p->moved.key[h] == pos ? p->moved.val[h] : pos;
}
static void perm_set(sparse_perm_t *p, int32_t pos, int32_t v)
{
    uint32_t h = nodeset_probe(&p->moved, pos);
    p->moved.key[h] = pos;
    p->moved.val[h] = v;
}

/* ------------------------------------------------------------ walk engine */

typedef struct
{
    const int64_t *indptr;
    const int32_t *indices;
    int M, m, rng_mode;
    uint32_t seed;
    uint32_t tag; /* philox stream tag: 0 = first hop w/o replacement (+gset), 1 = plain walks */
    int cap_first; /* clamp the root degree to NEBMAX (set_sampler only, subg_acc.c:750) */
} walk_cfg_t;

/* Walk all M walks of one root; emit(node, walk, step, user) is called for step = 0..m-1 in
 * walk-major order.  `state` is the sequential rand_r state (RAND_R mode only).
 * without_repl: first hop distinct-neighbour rule (random_walk_wo / set_sampler) vs every
 * step drawn (random_walk). */
typedef void (*emit_fn)(int32_t node, int walk, int step, void *user);

static void walk_root(const walk_cfg_t *c, int32_t root, int without_repl, uint32_t *state,
                      sparse_perm_t *perm, int32_t *first, emit_fn emit, void *user)
{
    const int M = c->M, m = c->m;
    int64_t rbeg = c->indptr[root];
    int64_t rdeg = c->indptr[root + 1] - rbeg;
    int32_t deg = (int32_t)((c->cap_first && rdeg > ORC_NEIGH_CAP) ? ORC_NEIGH_CAP : rdeg);

    if (without_repl && deg > M)
    {
        nodeset_clear(&perm->moved);
        for (int k = 0; k < M; ++k)
        {
            uint32_t r = c->rng_mode == ORC_RNG_RAND_R ? orc_rand_r(state)
                                                       : philox_shuffle_draw(c->seed, (uint32_t)root, (uint32_t)k, c->tag);
            int32_t s = (int32_t)(c->rng_mode == ORC_RNG_RAND_R ? r % (uint32_t)(deg - k) : philox_below(r, (uint32_t)(deg - k))) + k;
            int32_t vk = perm_get(perm, k), vs = perm_get(perm, s);
            perm_set(perm, k, vs);
            perm_set(perm, s, vk);
            first[k] = vs;
        }
    }

    for (int w = 0; w < M; ++w)
    {
        int32_t cur = root;
        for (int s = 0; s < m; ++s)
        {
            if (s == 0 && without_repl)
            {
                if (deg > 0)
                    cur = c->indices[rbeg + (deg > M ? first[w] : w % deg)];
            }
            else
            {
                int64_t b = c->indptr[cur];
                int64_t d = c->indptr[cur + 1] - b;
                if (d > 0)
                {
                    uint32_t r;
                    if (c->rng_mode == ORC_RNG_RAND_R)
                        r = orc_rand_r(state);
                    else
                        r = philox_draw(c->seed, (uint32_t)root, (uint32_t)w, (uint32_t)(without_repl ? s - 1 : s), c->tag);
                    cur = c->indices[b + (int64_t)(c->rng_mode == ORC_RNG_RAND_R ? r % (uint32_t)d : philox_below(r, (uint32_t)d))];
                }
            }
            emit(cur, w, s, user);
        }
    }
}

/* ----------------------------------------------------------- gset_sampler */

typedef struct
{
    nodeset_t *set;
    int32_t *ids;    /* scratch [stride]        */
    int16_t *counts; /* scratch [stride*ncol]   */
    int count, stride, ncol, overflow;
} gset_acc_t;

static void gset_emit(int32_t node, int walk, int step, void *user)
{
    (void)walk;
    gset_acc_t *a = (gset_acc_t *)user;
    uint32_t h = nodeset_probe(a->set, node);
    int slot;
    if (a->set->key[h] == node)
        slot = a->set->val[h];
    else
    {
        if (a->count >= a->stride)
        { /* bucket full: the visit is dropped, subg_acc.c:814-828 */
            a->overflow = 1;
            return;
        }
        slot = a->count++;
        a->set->key[h] = node;
        a->set->val[h] = slot;
        a->ids[slot] = node;
        memset(a->counts + (size_t)slot * a->ncol, 0, sizeof(int16_t) * a->ncol);
    }
    a->counts[(size_t)slot * a->ncol + step + 1]++;
}

static int gset_range(const walk_cfg_t *cfg, const int32_t *query, int64_t lo, int64_t hi, int stride,
                      uint32_t *state, setbuf_t *out, int64_t *n_overflow)
{
    const int M = cfg->M, m = cfg->m, ncol = m + 1;
    nodeset_t set;
    sparse_perm_t perm;
    if (nodeset_init(&set, stride) || nodeset_init(&perm.moved, 2 * (int64_t)M))
        return -1;
    int32_t *first = (int32_t *)malloc(sizeof(int32_t) * (M > 0 ? M : 1));
    gset_acc_t acc;
    acc.set = &set;
    acc.ids = (int32_t *)malloc(sizeof(int32_t) * stride);
    acc.counts = (int16_t *)malloc(sizeof(int16_t) * (size_t)stride * ncol);
    acc.stride = stride;
    acc.ncol = ncol;
    int rc = 0;
    for (int64_t i = lo; i < hi && rc == 0; ++i)
    {
        int32_t root = query[i];
        nodeset_clear(&set);
        acc.count = 1;
        acc.overflow = 0;
        { /* the root is always member 0, flagged by col0 = M (subg_acc.c:751,779-782) */
            uint32_t h = nodeset_probe(&set, root);
            set.key[h] = root;
            set.val[h] = 0;
            acc.ids[0] = root;
            memset(acc.counts, 0, sizeof(int16_t) * ncol);
            acc.counts[0] = (int16_t)M;
        }
        if (cfg->indptr[root + 1] - cfg->indptr[root] == 0)
        { /* isolated root: [M, M, ..., M], subg_acc.c:753-761 */
            for (int s = 1; s < ncol; ++s)
                acc.counts[s] = (int16_t)M;
        }
        else
            walk_root(cfg, root, 1, state, &perm, first, gset_emit, &acc);
        *n_overflow += acc.overflow;
        if (setbuf_reserve(out, out->X + acc.count, ncol))
        {
            rc = -1;
            break;
        }
        memcpy(out->ids + out->X, acc.ids, sizeof(int32_t) * acc.count);
        memcpy(out->counts + out->X * ncol, acc.counts, sizeof(int16_t) * (size_t)acc.count * ncol);
        out->nsize[i - lo] = acc.count;
        out->X += acc.count;
    }
    free(first);
    free(acc.ids);
    free(acc.counts);
    nodeset_free(&set);
    nodeset_free(&perm.moved);
    return rc;
}

/* 64-bit key of one LP row: counts of steps 1..m, SHIFT bits each, most significant first,
 * plus a LEAD bit on root rows (subg_acc.c:900-955). */
static int lp_shift(int M) { return 32 - __builtin_clz((unsigned)M); }

typedef struct
{
    uint64_t *key;
    int32_t *val;
    uint64_t mask;
    int64_t count;
} u64map_t;

static int u64map_init(u64map_t *t, uint64_t cap)
{
    t->key = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    t->val = (int32_t *)malloc(sizeof(int32_t) * cap);
    if (!t->key || !t->val)
        return -1;
    memset(t->val, 0xFF, sizeof(int32_t) * cap); /* val -1 = empty */
    t->mask = cap - 1;
    t->count = 0;
    return 0;
}
static uint64_t mix64(uint64_t x)
{
    x ^= x >> 33, x *= 0xff51afd7ed558ccdULL, x ^= x >> 33, x *= 0xc4ceb9fe1a85ec53ULL, x ^= x >> 33;
    return x;
}
static int u64map_grow(u64map_t *t);
static int32_t u64map_get_or_add(u64map_t *t, uint64_t key, int *added)
{
    if ((uint64_t)t->count * 2 > t->mask)
        if (u64map_grow(t))
            return -2;
    uint64_t h = mix64(key) & t->mask;
    while (t->val[h] != -1 && t->key[h] != key)
        h = (h + 1) & t->mask;
    if (t->val[h] == -1)
    {
        t->key[h] = key;
        t->val[h] = (int32_t)t->count++;
        *added = 1;
    }
    else
        *added = 0;
    return t->val[h];
}
static int u64map_grow(u64map_t *t)
{
    u64map_t n;
    if (u64map_init(&n, (t->mask + 1) * 2))
        return -1;
    for (uint64_t i = 0; i <= t->mask; ++i)
        if (t->val[i] != -1)
        {
            uint64_t h = mix64(t->key[i]) & n.mask;
            while (n.val[h] != -1)
                h = (h + 1) & n.mask;
            n.key[h] = t->key[i];
            n.val[h] = t->val[i];
        }
    n.count = t->count;
    free(t->key);
    free(t->val);
    *t = n;
    return 0;
}

typedef struct
{
    int32_t *nsize;  /* [n]       */
    int32_t *ids;    /* [X]   remap row 0 */
    int32_t *sf;     /* [X]   remap row 1 (index into enc) */
    int16_t *enc;    /* [c*ncol]  */
    int16_t *raw;    /* [X*ncol]  (debug>0 output of the reference) */
    int64_t X, c, n_overflow;
} orc_gset_result;

void orc_gset_free(orc_gset_result *r)
{
    free(r->nsize), free(r->ids), free(r->sf), free(r->enc), free(r->raw);
    memset(r, 0, sizeof(*r));
}

/* status: 0 ok, -1 out of memory, -2 key too wide (needs m*SHIFT+1 <= 64, subg_acc.c:905-915) */
int orc_gset_sampler(const int64_t *indptr, const int32_t *indices, const int32_t *query, int64_t n,
                     int M, int m, int bucket, uint32_t seed, int rng_mode, int nthreads, orc_gset_result *res)
{
    const int ncol = m + 1;
    const int stride = bucket < 0 ? M * m + 1 : bucket;
    const int SHIFT = lp_shift(M);
    memset(res, 0, sizeof(*res));
    if ((int64_t)m * SHIFT + 1 > 64)
        return -2;
    walk_cfg_t cfg = {indptr, indices, M, m, rng_mode, seed, 0u, 1};

    int T = 1;
#ifdef _OPENMP
    if (rng_mode == ORC_RNG_PHILOX && nthreads > 1)
        T = nthreads;
#endif
    (void)nthreads;
    res->nsize = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
    setbuf_t *parts = (setbuf_t *)calloc((size_t)T, sizeof(setbuf_t));
    int64_t *ovf = (int64_t *)calloc((size_t)T, sizeof(int64_t));
    int rc = 0;
    uint32_t state = seed; /* one sequential stream for all roots: the nthread=1 behaviour of :731-732 */
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t)
    {
        int64_t lo = n * t / T, hi = n * (t + 1) / T;
        parts[t].nsize = res->nsize + lo;
        int r = gset_range(&cfg, query, lo, hi, stride, &state, &parts[t], &ovf[t]);
        if (r)
            rc = r;
    }
    int64_t X = 0;
    for (int t = 0; t < T; ++t)
        X += parts[t].X, res->n_overflow += ovf[t];
    res->X = X;
    res->ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(X > 0 ? X : 1));
    res->sf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(X > 0 ? X : 1));
    res->raw = (int16_t *)malloc(sizeof(int16_t) * (size_t)(X > 0 ? X : 1) * ncol);
    if (!res->ids || !res->sf || !res->raw)
        rc = -1;
    if (rc == 0)
    {
        int64_t off = 0;
        for (int t = 0; t < T; ++t)
        {
            memcpy(res->ids + off, parts[t].ids, sizeof(int32_t) * parts[t].X);
            memcpy(res->raw + off * ncol, parts[t].counts, sizeof(int16_t) * parts[t].X * ncol);
            off += parts[t].X;
        }
    }
    for (int t = 0; t < T; ++t)
        free(parts[t].ids), free(parts[t].counts);
    free(parts);
    free(ovf);
    if (rc)
        return rc;

    /* global first-occurrence dedup of LP rows through the packed key, subg_acc.c:936-978 */
    u64map_t uniq;
    if (u64map_init(&uniq, 1024))
        return -1;
    int64_t enc_cap = 1024;
    res->enc = (int16_t *)malloc(sizeof(int16_t) * enc_cap * ncol);
    for (int64_t e = 0; e < X; ++e)
    {
        const int16_t *row = res->raw + e * ncol;
        uint64_t key = 0;
        for (int j = 1; j < ncol; ++j)
            key = (key << SHIFT) | (uint64_t)(uint16_t)row[j];
        if (row[0] != 0) /* root row: col0 == M */
            key |= (uint64_t)1 << ((ncol - 1) * SHIFT);
        int added;
        int32_t id = u64map_get_or_add(&uniq, key, &added);
        if (id < 0)
            return -1;
        if (added)
        {
            if (uniq.count > enc_cap)
            {
                enc_cap *= 2;
                res->enc = (int16_t *)realloc(res->enc, sizeof(int16_t) * enc_cap * ncol);
                if (!res->enc)
                    return -1;
            }
            memcpy(res->enc + (size_t)id * ncol, row, sizeof(int16_t) * ncol);
        }
        res->sf[e] = id;
    }
    res->c = uniq.count;
    free(uniq.key);
    free(uniq.val);
    return 0;
}

/* ----------------------------------------------------------- walk_sampler */

typedef struct
{
    int32_t *walks;  /* [n * M*(m+1)]  */
    int32_t *nsize;  /* [n]   unique nodes per root */
    int32_t *ids;    /* [X]   step-major first-visit order, root first */
    int32_t *counts; /* [X*(m+1)] */
    int64_t X;
} orc_walk_result;

void orc_walk_free(orc_walk_result *r)
{
    free(r->walks), free(r->nsize), free(r->ids), free(r->counts);
    memset(r, 0, sizeof(*r));
}

typedef struct
{
    int32_t *row; /* walks of this root, [M*(m+1)] */
    int m;
} walk_rec_t;
static void walk_emit(int32_t node, int walk, int step, void *user)
{
    walk_rec_t *r = (walk_rec_t *)user;
    r->row[(size_t)walk * (r->m + 1) + step + 1] = node;
}

/* libgomp static schedule: the first n%T threads take ceil(n/T) iterations. */
static void static_chunk(int64_t n, int T, int t, int64_t *lo, int64_t *hi)
{
    int64_t q = n / T, r = n % T;
    if (t < r)
        *lo = t * (q + 1), *hi = *lo + q + 1;
    else
        *lo = r * (q + 1) + (t - r) * q, *hi = *lo + q;
}

int orc_walk_sampler(const int64_t *indptr, const int32_t *indices, const int32_t *query, int64_t n,
                     int M, int m, uint32_t seed, int nthread_streams, int without_repl, int rng_mode,
                     orc_walk_result *res)
{
    const int W = M * (m + 1), ncol = m + 1;
    memset(res, 0, sizeof(*res));
    res->walks = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1) * W);
    res->nsize = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
    if (!res->walks || !res->nsize)
        return -1;
    walk_cfg_t cfg = {indptr, indices, M, m, rng_mode, seed, without_repl ? 0u : 1u, 0};
    int T = nthread_streams > 0 ? nthread_streams : 1;
    sparse_perm_t perm;
    if (nodeset_init(&perm.moved, 2 * (int64_t)M))
        return -1;
    int32_t *first = (int32_t *)malloc(sizeof(int32_t) * (M > 0 ? M : 1));
    for (int t = 0; t < T; ++t)
    { /* thread t of the reference owns one contiguous chunk and the stream seeded seed+t (:157-158,:191-192) */
        int64_t lo, hi;
        static_chunk(n, T, t, &lo, &hi);
        uint32_t state = seed + (uint32_t)t;
        for (int64_t i = lo; i < hi; ++i)
        {
            walk_rec_t rec = {res->walks + (size_t)i * W, m};
            for (int w = 0; w < M; ++w)
                rec.row[(size_t)w * ncol] = query[i];
            walk_root(&cfg, query[i], without_repl, &state, &perm, first, walk_emit, &rec);
        }
    }
    free(first);
    nodeset_free(&perm.moved);

    /* rpe_encoder (:249-314): unique ids in step-major first-visit order, counts[slot][step] */
    nodeset_t set;
    if (nodeset_init(&set, (int64_t)M * m + 1))
        return -1;
    int64_t cap = 0, X = 0;
    for (int64_t i = 0; i < n; ++i)
    {
        const int32_t *row = res->walks + (size_t)i * W;
        int64_t need = X + (int64_t)M * m + 1;
        if (need > cap)
        {
            cap = need + cap / 2;
            res->ids = (int32_t *)realloc(res->ids, sizeof(int32_t) * cap);
            res->counts = (int32_t *)realloc(res->counts, sizeof(int32_t) * cap * ncol);
            if (!res->ids || !res->counts)
                return -1;
        }
        nodeset_clear(&set);
        int32_t *ids = res->ids + X;
        int32_t *cnt = res->counts + X * ncol;
        int count = 1;
        {
            uint32_t h = nodeset_probe(&set, row[0]);
            set.key[h] = row[0], set.val[h] = 0;
            ids[0] = row[0];
            memset(cnt, 0, sizeof(int32_t) * ncol);
            cnt[0] = M;
        }
        for (int s = 1; s <= m; ++s)
            for (int w = 0; w < M; ++w)
            {
                int32_t node = row[(size_t)w * ncol + s];
                uint32_t h = nodeset_probe(&set, node);
                int slot;
                if (set.key[h] == node)
                    slot = set.val[h];
                else
                {
                    slot = count++;
                    set.key[h] = node, set.val[h] = slot;
                    ids[slot] = node;
                    memset(cnt + (size_t)slot * ncol, 0, sizeof(int32_t) * ncol);
                }
                cnt[(size_t)slot * ncol + s]++;
            }
        res->nsize[i] = count;
        X += count;
    }
    res->X = X;
    nodeset_free(&set);
    return 0;
}

/* --------------------------------------------------------------- SpG build */

typedef struct
{
    int32_t id, val;
} idval_t;
static int idval_cmp(const void *a, const void *b)
{
    int32_t x = ((const idval_t *)a)->id, y = ((const idval_t *)b)->id;
    return (x > y) - (x < y);
}

/* rows = query positions; per row the members sorted by node id; data = sf + 1.
 * (scipy's COO->CSR of random_walks.py:79 does exactly this for unique (row, col) pairs.) */
int orc_spg_build(const int32_t *nsize, int64_t n, const int32_t *ids, const int32_t *sf,
                  int64_t *out_indptr, int32_t *out_indices, int32_t *out_data, int nthreads)
{
    int64_t maxn = 1;
    out_indptr[0] = 0;
    for (int64_t i = 0; i < n; ++i)
    {
        out_indptr[i + 1] = out_indptr[i] + nsize[i];
        if (nsize[i] > maxn)
            maxn = nsize[i];
    }
    int rc = 0;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
    {
        idval_t *tmp = (idval_t *)malloc(sizeof(idval_t) * (size_t)maxn);
        if (!tmp)
            rc = -1;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < n; ++i)
        {
            if (!tmp)
                continue;
            const int64_t off = out_indptr[i];
            const int ns = nsize[i];
            for (int r = 0; r < ns; ++r)
                tmp[r].id = ids[off + r], tmp[r].val = sf[off + r] + 1;
            qsort(tmp, (size_t)ns, sizeof(idval_t), idval_cmp);
            for (int r = 0; r < ns; ++r)
                out_indices[off + r] = tmp[r].id, out_data[off + r] = tmp[r].val;
        }
        free(tmp);
    }
    return rc;
}

/* ------------------------------------------------------------------ SpJoin */

/* Generic segment join.  Segment j lists the members of SpG row own[j] in ascending id order;
 * each output row carries (value in own row, value of the same node in row partner[j] or 0).
 * gather(edge)  = own [u0..uB-1, v0..vB-1], partner [v.., u..]           (train.py:13-45)
 * hgather(hedge)= own [u.., w.., v.., w..], partner [w.., u.., w.., v..]  (train.py:48-72)
 * out_seg[S+1] = exclusive scan of the segment sizes.
 * int mode: out_idx int32 [R,2].  float mode (data_f64 != NULL): out_val float32 [R,2] with the
 * second slot computed as (partner_value_or_0 + 1.0) - 1.0 in double, as the scipy expression
 * `x.multiply(mask) + mask` followed by `.data - 1` does. */
int64_t orc_sjoin_count(const int64_t *indptr, const int64_t *own, int64_t S, int64_t *out_seg)
{
    int64_t R = 0;
    out_seg[0] = 0;
    for (int64_t j = 0; j < S; ++j)
    {
        R += indptr[own[j] + 1] - indptr[own[j]];
        out_seg[j + 1] = R;
    }
    return R;
}

void orc_sjoin_fill(const int64_t *indptr, const int32_t *indices, const int32_t *data_i32,
                    const double *data_f64, const int64_t *own, const int64_t *partner, int64_t S,
                    const int64_t *seg, int32_t *out_idx, float *out_val, int nthreads)
{
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(dynamic, 64)
    for (int64_t j = 0; j < S; ++j)
    {
        int64_t a = indptr[own[j]], ae = indptr[own[j] + 1];
        int64_t b = indptr[partner[j]], be = indptr[partner[j] + 1];
        int64_t o = seg[j];
        for (; a < ae; ++a, ++o)
        {
            int32_t id = indices[a];
            while (b < be && indices[b] < id)
                ++b;
            int hit = (b < be && indices[b] == id);
            if (data_f64)
            {
                out_val[2 * o] = (float)data_f64[a];
                out_val[2 * o + 1] = (float)(((hit ? data_f64[b] : 0.0) + 1.0) - 1.0);
            }
            else
            {
                out_idx[2 * o] = data_i32[a];
                out_idx[2 * o + 1] = hit ? data_i32[b] : 0;
            }
        }
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ walk_join (legacy SUREL join over raw walks)
 * subg_acc/subg_acc.c:509-647.  walk int32[n, stride] (column 0 = the root), key lists concatenated in `key_ids`
 * with offsets key_off[n+1]; member j of the concatenation has index j+1 (:573-584, `idx` runs over all lists).
 * query int32[Q,2] node ids.  out int32[2, Q*2*stride]; xrow int32[Q,2] = row of each query key (the LAST row
 * whose root equals it: uthash prepends to its bucket chain, so HASH_FIND meets the latest duplicate first; -1 when
 * the key is no root -- the reference then reads out of bounds, here the 4 values of that query are left -1).
 * find_idx(:78-92): index of the node in the key's list, 0 when absent.  The lists hold distinct ids. */
typedef struct
{
    int32_t id, idx;
} wj_ent;
static int wj_cmp(const void *a, const void *b)
{
    const wj_ent *x = a, *y = b;
    return x->id < y->id ? -1 : (x->id > y->id);
}
static int32_t wj_find(const wj_ent *row, int64_t len, int32_t node)
{
    int64_t lo = 0, hi = len;
    while (lo < hi)
    {
        int64_t mid = (lo + hi) >> 1;
        if (row[mid].id < node)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < len && row[lo].id == node) ? row[lo].idx : 0;
}
int orc_walk_join(const int32_t *walk, int64_t n, int32_t stride, const int64_t *key_off, const int32_t *key_ids,
                  const int32_t *query, int64_t Q, int32_t *out, int32_t *xrow)
{
    const int64_t X = key_off[n];
    wj_ent *ent = malloc(sizeof(wj_ent) * (size_t)(X > 0 ? X : 1));
    wj_ent *roots = malloc(sizeof(wj_ent) * (size_t)(n > 0 ? n : 1));
    if (!ent || !roots)
        return -1;
    for (int64_t i = 0; i < n; ++i)
    {
        for (int64_t j = key_off[i]; j < key_off[i + 1]; ++j)
            ent[j].id = key_ids[j], ent[j].idx = (int32_t)(j + 1);
        qsort(ent + key_off[i], (size_t)(key_off[i + 1] - key_off[i]), sizeof(wj_ent), wj_cmp);
        roots[i].id = walk[i * (int64_t)stride], roots[i].idx = (int32_t)i;
    }
    /* stable by construction: sort by (id, row) and take the last of a run */
    for (int64_t i = 1; i < n; ++i)
    { /* insertion sort keeps equal ids in row order; n is a batch (thousands) */
        wj_ent t = roots[i];
        int64_t j = i - 1;
        while (j >= 0 && roots[j].id > t.id)
            roots[j + 1] = roots[j], --j;
        roots[j + 1] = t;
    }
    const int64_t width = Q * 2 * (int64_t)stride;
    for (int64_t x = 0; x < Q; ++x)
    {
        int32_t row[2];
        for (int s = 0; s < 2; ++s)
        {
            const int32_t key = query[2 * x + s];
            int64_t lo = 0, hi = n; /* last position with id <= key */
            while (lo < hi)
            {
                int64_t mid = (lo + hi) >> 1;
                if (roots[mid].id <= key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            row[s] = (lo > 0 && roots[lo - 1].id == key) ? roots[lo - 1].idx : -1;
            xrow[2 * x + s] = row[s];
        }
        for (int32_t t = 0; t < stride; ++t)
            for (int s = 0; s < 2; ++s)
            {
                int32_t *o = out + s * width + 2 * x * (int64_t)stride + 2 * t;
                if (row[0] < 0 || row[1] < 0)
                {
                    o[0] = o[1] = -1;
                    continue;
                }
                const int32_t node = walk[row[s] * (int64_t)stride + t];
                o[0] = wj_find(ent + key_off[row[0]], key_off[row[0] + 1] - key_off[row[0]], node);
                o[1] = wj_find(ent + key_off[row[1]], key_off[row[1] + 1] - key_off[row[1]], node);
            }
    }
    free(ent), free(roots);
    return 0;
}

/* ------------------------------------------------------------ batch_sampler
 * Restates subg_acc/subg_acc.c:391-507 (legacy SUREL mini-batch former): the roots are walked one after the other with
 * ONE rand_r stream, every visited node goes into one insertion-ordered set, and a root stops walking once the set holds
 * (i+1)*thld/n nodes (checked after every walk, :474).  `seed_eff` is the state the reference builds as seed + getpid()
 * (:421) -- the caller passes the sum, so a run of the reference in a known process can be reproduced.
 * out (caller-allocated, out_cap entries) receives the nodes in insertion order; returns their number, or -1 when
 * out_cap is too small, -2 on allocation failure.  A node without out-edges draws nothing (:466-471). */
int64_t orc_batch_sampler(const int64_t *indptr, const int32_t *indices, const int32_t *query, int64_t n, int M, int S,
                          int thld, uint32_t seed_eff, int32_t *out, int64_t out_cap)
{
    nodeset_t set;
    if (nodeset_init(&set, out_cap))
        return -2;
    nodeset_clear(&set);
    int64_t count = 0;
    uint32_t state = seed_eff;
    int32_t *rseq = NULL;
    int64_t rcap = 0;
    int64_t rc = 0;
#define ORC_BATCH_ADD(v)                                 \
    do                                                   \
    {                                                    \
        uint32_t h__ = nodeset_probe(&set, (v));         \
        if (set.key[h__] == -1)                          \
        {                                                \
            if (count >= out_cap)                        \
            {                                            \
                rc = -1;                                 \
                goto done;                               \
            }                                            \
            set.key[h__] = (v);                          \
            out[count++] = (v);                          \
        }                                                \
    } while (0)
    for (int64_t i = 0; i < n; ++i)
    {
        const int32_t root = query[i];
        const int64_t rbeg = indptr[root];
        const int32_t deg = (int32_t)(indptr[root + 1] - rbeg);
        if (deg > M)
        { /* partial Fisher-Yates over 0..deg-1 (:434-446) */
            if (deg > rcap)
            {
                free(rseq);
                rseq = (int32_t *)malloc(sizeof(int32_t) * (size_t)deg);
                rcap = deg;
                if (!rseq)
                {
                    rc = -2;
                    goto done;
                }
            }
            for (int32_t j = 0; j < deg; ++j)
                rseq[j] = j;
            for (int k = 0; k < M; ++k)
            {
                int32_t s = (int32_t)(orc_rand_r(&state) % (uint32_t)(deg - k)) + k;
                int32_t t = rseq[k];
                rseq[k] = rseq[s];
                rseq[s] = t;
            }
        }
        ORC_BATCH_ADD(root);
        for (int w = 0; w < M; ++w)
        {
            int32_t cur = root;
            if (deg < 1)
                break;
            cur = indices[rbeg + (deg <= M ? w % deg : rseq[w])];
            ORC_BATCH_ADD(cur);
            for (int step = 1; step < S; ++step)
            {
                const int64_t b = indptr[cur];
                const int32_t d = (int32_t)(indptr[cur + 1] - b);
                if (d > 0)
                {
                    cur = indices[b + (int64_t)(orc_rand_r(&state) % (uint32_t)d)];
                    ORC_BATCH_ADD(cur);
                }
            }
            if ((int)count >= (int)((i + 1) * thld / n))
                break;
        }
    }
    rc = count;
done:
#undef ORC_BATCH_ADD
    free(rseq);
    nodeset_free(&set);
    return rc;
}
