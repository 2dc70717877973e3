/*
 * subgacc.h -- C ABI of the MI355X-native SubGAcc hot path (libsubgacc_hip.so, gfx950).
 *
 * This is the drop-in boundary for the SUREL+ path
 *     sample (random-walk node sets + landing-probability counts)  ->  SpG  ->  SpJoin.
 * Every entry point names the reference interface it replaces (paths relative to the reference
 * tree, Graph-COM/SUREL_Plus).  The reference binds its C half through a CPython module
 * (`subg_acc`, subg_acc/subg_acc.c:1036-1059) and runs SpJoin in SciPy (train.py:13-111); the host
 * mirror of both lives in surel_plus_amd/ and is a thin ctypes layer over the functions below.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / numpy / HIP types in any signature.
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and nothing
 *     synchronises unless the function says so.  No allocation happens inside: scratch comes
 *     from the caller (`*_workspace_bytes`), so calls are graph-capturable.
 *   - return value: SUBGACC_OK or a negative subgacc_status; subgacc_last_error() gives the text
 *     (thread local).
 *   - node ids are int32 (the reference is int32-only, subg_acc.c:663-676); CSR row offsets may be
 *     int32 or int64 (`indptr64`), SpG row offsets are always int64.
 */
#ifndef SUBGACC_H
#define SUBGACC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUBGACC_ABI_VERSION 7

typedef enum subgacc_status {
    SUBGACC_OK = 0,
    SUBGACC_ERR_BADARG = -1,    /* TypeError("Input parsing error.") territory, subg_acc.c:658 */
    SUBGACC_ERR_WORKSPACE = -2, /* caller's scratch too small (the reference's MemoryError paths) */
    SUBGACC_ERR_KEYWIDTH = -3,  /* m*SHIFT+1 > 64: AssertionError, subg_acc.c:905-915 */
    SUBGACC_ERR_CAPACITY = -4,  /* unique-row table full: retry with a larger one */
    SUBGACC_ERR_HIP = -5,       /* a HIP runtime call failed */
    SUBGACC_ERR_LDS = -6,       /* M*m+1 too large for the per-root LDS tables */
    SUBGACC_ERR_NODEVICE = -7   /* no gfx950 device visible */
} subgacc_status;

enum { SUBGACC_RNG_RAND_R = 0, /* glibc rand_r stream, bit-exact with the reference at nthread=1 */
       SUBGACC_RNG_PHILOX = 1  /* Philox2x32-10, key = seed, counter = (root id, walk | draw block | stream tag): schedule independent */ };
enum { SUBGACC_ORDER_WALK_MAJOR = 0, /* set_sampler's first-visit order, subg_acc.c:785-832 */
       SUBGACC_ORDER_STEP_MAJOR = 1  /* rpe_encoder's first-visit order, subg_acc.c:263-278 */ };

int subgacc_abi_version(void);
const char *subgacc_last_error(void);
/* number of visible HIP devices whose arch is gfx950 (0 => every compute call fails loudly) */
int subgacc_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Sampler: walks + per-root node-set dedup + landing-probability counts.
 * Replaces the hot loop of set_sampler (subg_acc.c:742-846), random_walk (:144-180),
 * random_walk_wo (:183-247) and rpe_encoder (:249-314).
 * ------------------------------------------------------------------------------------------- */
typedef struct subgacc_walk_cfg {
    int32_t num_walks;       /* M                                                             */
    int32_t num_steps;       /* m = hops per walk (gset_sampler's num_steps; CLI --num_steps-1) */
    int32_t bucket;          /* cap on the set size; <=0 => M*m+1 (subg_acc.c:680)              */
    int32_t rng_mode;        /* SUBGACC_RNG_*                                                   */
    uint32_t seed;
    int32_t first_hop_wo;    /* 1: first hop without replacement (set_sampler, random_walk_wo)  */
    int32_t order;           /* SUBGACC_ORDER_*                                                 */
    int32_t cap_root_degree; /* 1: clamp the root degree to 1e6 (NEBMAX, subg_acc.c:750)        */
    int32_t indptr64;        /* CSR row offsets are int64 (else int32)                          */
    int32_t emit_walks;      /* 1: also write raw walks int32[n, M*(m+1)] (walk_sampler)        */
    /* ABI 3: optional packed hop records of the graph (subgacc_hop_records_build), NULL = none.  The fused-row kernel
     * then makes ONE dependent 8-byte read per hop (neighbour id + its row begin + its degree) instead of a neighbour
     * read followed by a row-pointer read; results are identical.  rec_id_bits / rec_beg_bits as given to the build;
     * both 0 = the 16-byte form.  The 8-byte form goes with int32 row offsets, the 16-byte form with int64 (a mismatch is
     * ignored: the plain CSR is walked). */
    const void *hop_records;
    int32_t rec_id_bits, rec_beg_bits;
    /* ABI 4: RAND_R on a graph with dead ends -- the stream position (LCG steps from the stream's seed) of every WALK,
     * uint32 [n * num_walks] for the n roots of the call, from subgacc_rng_replay; NULL = positions follow from the
     * degrees (subgacc_rng_positions), the symmetrised graphs of the reference's loader. */
    const uint32_t *walk_pos;
    /* ABI 6: words between the rows of two consecutive roots in the strided outputs (set_ids / set_keys / row_ids / row_slot); 0 =
     * the row capacity itself (bucket, or M*m+1).  A multiple of 32 puts every row on a 128-byte line: the join reads and the walk
     * kernels write whole lines.  Whoever reads the rows afterwards takes the same number as its `stride` / `row_stride`. */
    int32_t row_pitch;
} subgacc_walk_cfg;

/* Hop records: rec[e] = [indices[e] : id_bits | indptr[indices[e]] : beg_bits | degree(indices[e]) : rest], one uint64 per
 * CSR entry (8*nnz bytes).  subgacc_hop_records_format picks the field widths for a graph and returns the bits left for
 * the degree (a degree that does not fit is stored as all ones: the kernel reads the row pointers for that node), or
 * SUBGACC_ERR_BADARG when fewer than 8 are left (records are not worth building then).
 * id_bits = beg_bits = 0 builds the 16-byte form {id : 32 | degree : 32, row begin : 64} (recs: 2 uint64 per entry, 16*nnz
 * bytes) -- what a graph with int64 row offsets takes. */
int subgacc_hop_records_format(int64_t num_nodes, int64_t nnz, int32_t *id_bits, int32_t *beg_bits);
int subgacc_hop_records_build(const void *indptr, int32_t indptr64, const int32_t *indices, int64_t num_nodes, int64_t nnz,
                              int32_t id_bits, int32_t beg_bits, uint64_t *recs, void *stream);

/* LP rows are carried as one packed 64-bit key: count of step j in bits [(m-j)*SHIFT, +SHIFT),
 * SHIFT = 32-clz(M), plus bit m*SHIFT (LEAD) on the root row -- the reference's `bithash`
 * (subg_acc.c:900-955).  Returns SHIFT, or SUBGACC_ERR_KEYWIDTH. */
int subgacc_key_shift(int32_t num_walks, int32_t num_steps);

/* RAND_R mode only: per-root position in the sequential rand_r stream.  The reference consumes
 * calls(r) = (deg>M ? M : 0) + M*(m-1) draws per non-isolated root (first_hop_wo) or M*m
 * (plain walks), in root order; stream t of `rng_streams` owns libgomp's static chunk t of the n
 * roots and starts from seed+t (subg_acc.c:157-158,191-192; one stream for set_sampler :731-732).
 * Writes the pair (rng_seed[i], rng_pos[i]) = (an LCG state, LCG steps to take from it) that names root i's first draw; the
 * walk entry points only ever use it as lcg_jump(rng_seed[i], rng_pos[i] + ...).  It leaves NORMALISED -- rng_seed[i] = the
 * state of stream t after the draws before root i, rng_pos[i] = 0 -- so that no workgroup has to make the jump again.
 * `calls_before` = draws consumed before query[0] (for sharded / chunked callers). */
size_t subgacc_rng_positions_workspace_bytes(int64_t n);
int subgacc_rng_positions(const subgacc_walk_cfg *cfg, const void *indptr, int64_t num_nodes, const int32_t *query, int64_t n,
                          int32_t rng_streams, uint64_t calls_before, uint32_t *rng_pos, uint32_t *rng_seed,
                          void *workspace, size_t workspace_bytes, void *stream);

/* The same positions for a graph WITH dead ends (a reached node without out-edges draws nothing in the reference and the
 * walk stays there, subg_acc.c:804-808, :168-172, :236-240: the position of every later draw then depends on how the earlier
 * walks ended).  One wavefront per stream replays it -- 64 walks at a time, right up to the first walk that ended early --
 * and writes rng_pos / rng_seed as above plus walk_pos [n * num_walks], which the walk entry points take through
 * cfg->walk_pos (they then run their general kernel).  Costs one dependent read per hop per walk of the sequential stream:
 * a fall-back for inputs the reference accepts, taken by the host mirror when a walk kernel reports flags[0] & 1. */
int subgacc_rng_replay(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                       const int32_t *query, int64_t n, int32_t rng_streams, uint64_t calls_before, uint32_t *rng_pos,
                       uint32_t *rng_seed, uint32_t *walk_pos, void *stream);

/* Sample n roots.  Outputs per root i, at fixed offsets i*stride (stride = bucket or M*m+1):
 *   set_ids [n*stride] int32   members in first-visit order (root first)
 *   set_keys[n*stride] uint64  packed LP row of each member
 *   nsize   [n]        int32   set size (<= stride)
 *   walks   [n*M*(m+1)] int32  (emit_walks only, else NULL)
 *   flags   [4]        int32   [0] |= 1 when RAND_R mode met a dead end (degree-0 non-root: the call
 *                              count is then data dependent and the stream cannot be reproduced in
 *                              parallel); [1] += roots whose set overflowed `bucket`; [3] |= 16 when a root
 *                              lies outside [0, num_nodes): it is never looked up (the reference reads out of
 *                              bounds there), its set is empty and the host mirror raises IndexError.
 *                              Caller zeroes.
 * rng_pos / rng_seed: from subgacc_rng_positions (RAND_R) or NULL (PHILOX). */
int subgacc_walk_sets(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                      const int32_t *query, int64_t n, const uint32_t *rng_pos, const uint32_t *rng_seed,
                      int32_t *set_ids, uint64_t *set_keys, int32_t *nsize, int32_t *walks, int32_t *flags,
                      void *stream);

/* Fused form for the SpG pipeline: sample n roots and leave every set as a FINISHED SpG row at offset i*stride:
 *   row_ids [n*stride] int32  members sorted by node id (random_walks.py:79-80)
 *   row_slot[n*stride] int32  slot of the member's LP key in `uniq_table` (translate after subgacc_uniq_number)
 * The keys are registered in the table with tag (root_base+i)*stride + first-visit rank, which numbers the
 * distinct rows exactly like the reference's sequential pass (subg_acc.c:957-978); root_base = global index of
 * query[0] when a job is split into chunks.  Needs M*m+1 <= 818 (SUBGACC_ERR_LDS otherwise: use
 * subgacc_walk_sets + subgacc_compact_sets + subgacc_spg_build).  flags as for subgacc_walk_sets, [2] |= 1
 * when the table is (nearly) full. */
int subgacc_walk_spg(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                     const int32_t *query, int64_t n, int64_t root_base, const uint32_t *rng_pos,
                     const uint32_t *rng_seed, void *uniq_table, int64_t uniq_capacity, int32_t *row_ids,
                     int32_t *row_slot, int32_t *nsize, int32_t *flags, void *stream);
/* The same finished rows from the sets of subgacc_walk_sets (for the configurations where that kernel is the faster
 * walk -- short walks over a cache-resident graph): row i of the staging area, row_ids / row_keys [i*stride, +nsize[i])
 * in first-visit order, becomes (ids sorted by node id, slots in `uniq_table`) IN PLACE (row_ids) and in row_slot,
 * the LP keys registered with tag (root_base+i)*stride + first-visit rank -- one pass, instead of
 * subgacc_compact_sets + subgacc_uniq_number + subgacc_spg_build and a packed copy.  stride <= 1024. */
int subgacc_finish_rows(int32_t *row_ids, const uint64_t *row_keys, const int32_t *nsize, int64_t n, int32_t stride,
                        int64_t root_base, void *uniq_table, int64_t uniq_capacity, int32_t *row_slot, int32_t *flags,
                        void *stream);
/* strided -> packed copy of the rows of subgacc_walk_spg: row i goes to [row_off[i], +nsize[i]).  With uniq_table
 * (already numbered by subgacc_uniq_number) out_data receives SFptr+1 directly; without it the raw table slots. */
int subgacc_compact_rows(const int32_t *row_ids, const int32_t *row_slot, const int32_t *nsize, const int64_t *row_off,
                         int64_t n, int32_t stride, int32_t *out_indices, int32_t *out_data, const void *uniq_table,
                         int64_t uniq_capacity, void *stream);

/* Exclusive scan int32 -> int64, out[n] = total.  (Prefix of nsize, subg_acc.c:848-851.) */
size_t subgacc_scan_workspace_bytes(int64_t n);
int subgacc_exclusive_scan_i32(const int32_t *in, int64_t n, int64_t *out, void *workspace, size_t workspace_bytes,
                               void *stream);

/* n <= 4,096 words of device memory (sizes, status words) to pinned, device-visible host memory by a one-wave kernel -- the read-back
 * of a lazily resolved step without a copy engine in the stream (the reference returns its sizes by value, subg_acc.c:1017-1024). */
int subgacc_publish_words(const int64_t *src, int64_t n, int64_t *host_dst, void *stream);

/* Left-compact the strided per-root sets (subg_acc.c:870-871): row i goes to [row_off[i], +nsize[i]).
 * out_keys may be NULL when uniq_table is given.  With uniq_table (see below; `uniq_capacity` slots) the pass
 * also inserts every member's key with position tag_base + row_off[i] + r (subg_acc.c:957-978) and writes the
 * table slot of every member to out_slot[] -- this replaces a separate subgacc_uniq_insert over out_keys. */
int subgacc_compact_sets(const int32_t *set_ids, const uint64_t *set_keys, const int32_t *nsize,
                         const int64_t *row_off, int64_t n, int32_t stride, int32_t *out_ids, uint64_t *out_keys,
                         void *uniq_table, int64_t uniq_capacity, int64_t tag_base, int32_t *out_slot, int32_t *flags,
                         void *stream);

/* ---------------------------------------------------------------------------------------------
 * Global first-occurrence dedup of LP rows (subg_acc.c:957-978): key -> index in order of first
 * appearance over the concatenated sets.  Open-addressing table in HBM, `capacity` a power of two.
 * ------------------------------------------------------------------------------------------- */
size_t subgacc_uniq_table_bytes(int64_t capacity);
int subgacc_uniq_reset(void *table, int64_t capacity, void *stream);
/* insert keys[0..n) whose global element positions are tag_base+0..n-1 (the position the element has in the
 * concatenated `slot` array later handed to subgacc_uniq_number); out_slot[e] (optional) = table slot of
 * keys[e]; flags[2] |= 1 when the table is (nearly) full */
int subgacc_uniq_insert(void *table, int64_t capacity, const uint64_t *keys, int64_t n, int64_t tag_base,
                        int32_t *out_slot, int32_t *flags, void *stream);
/* number the distinct keys by first occurrence (ascending minimum position):
 *   out_ukeys[] uint64 the distinct keys in index order (at most max_unique are written)
 *   out_count   int64  number of distinct keys (device scalar)
 * Tables with at most `small_limit` (<=0: 8192) distinct keys are ranked directly; larger ones are numbered by
 * a scan over the n elements (`slot` = the out_slot of the inserts, positions == tags). */
size_t subgacc_uniq_number_workspace_bytes(int64_t capacity, int64_t n);
int subgacc_uniq_number(void *table, int64_t capacity, const int32_t *slot, int64_t n, uint64_t *out_ukeys,
                        int64_t max_unique, int64_t *out_count, int64_t small_limit, void *workspace,
                        size_t workspace_bytes, void *stream);
/* slot_inout[e] <- index of the element's key (+add): remap[1] of the reference with add = 0, the SpG payload
 * SFptr+1 with add = 1.  n_dev (optional, device scalar): process min(n, *n_dev) elements -- lets a caller that
 * has not read the element count back yet launch over its buffer capacity. */
int subgacc_uniq_translate(void *table, int64_t capacity, int32_t *slot_inout, int64_t n, const int64_t *n_dev,
                           int32_t add, void *stream);

/* Unpack keys to LP rows [n, m+1]: col 0 = M on LEAD rows else 0 (subg_acc.c:751,982-1000).
 * out_i16 / out_i32 / out_f32 may each be NULL; out_f32 is float(count)/float(M) (main.py:174) and,
 * with zero_row != 0, gets an all-zero row prepended (random_walks.py:81) => [n+1, m+1].
 * n_dev (optional, device scalar): only the first min(n, *n_dev) keys are valid, later rows are zero-filled. */
int subgacc_unpack_lp(const uint64_t *keys, int64_t n, const int64_t *n_dev, int32_t num_walks, int32_t num_steps,
                      int16_t *out_i16, int32_t *out_i32, float *out_f32, int32_t zero_row, void *stream);

/* ---------------------------------------------------------------------------------------------
 * SpG build (sampler/random_walks.py:79-80: scipy COO->CSR): sort each row's members by node id.
 *   out_indices[X] int32 sorted ids, out_data[X] int32 = sf+1.  With uniq_table != NULL, sf[] holds table slots
 *   (out_slot of the inserts) and is translated on the fly, after subgacc_uniq_number.
 *   max_len >= max nsize; a longer row is
 *   left unwritten and flags[3] |= 1 (flags: the int32[4] word array of subgacc_walk_sets).
 * ------------------------------------------------------------------------------------------- */
int subgacc_spg_build(const int64_t *row_off, int64_t n, const int32_t *ids, const int32_t *sf,
                      const void *uniq_table, int64_t uniq_capacity, int32_t max_len, int32_t *out_indices,
                      int32_t *out_data, int32_t *flags, void *stream);

/* ---------------------------------------------------------------------------------------------
 * SpJoin (train.py:13-45 gather, :48-72 hgather, :75-111 bgather/pgather).
 * A join is a list of S segments (own[j], partner[j]) of SpG row numbers: segment j emits one row
 * per member w of row own[j] in ascending id order with the pair
 *     ( value_own(w), value_partner(w) or 0 ).
 * gather(edge[2,B]):  own = [u..,v..], partner = [v..,u..];  hgather: [u,w,v,w] / [w,u,w,v].
 * ABI 4: with a mirrored list (pair_block > 0, see subgacc_join_desc) `partner` may be NULL in every fill entry point: the
 * partner of segment j is then the own row of j's mirror (own[j + pair_block] in an even block, own[j - pair_block] in an odd
 * one) -- gather() passes its [2,B] endpoint tensor as `own` as it stands and builds no second list.
 * ------------------------------------------------------------------------------------------- */
/* out_seg[S+1] int64 = exclusive scan of the segment sizes (`indptr` of train.py:20-22).
 * n_rows = rows of the store.  A row number outside [0, n_rows) in own / partner (partner may be NULL: not checked
 * then) -- an IndexError of `x[edge[0]]` in the reference, train.py:15 -- is never dereferenced by any join entry
 * point: the row reads as empty and flags[3] |= 16 (flags: int32[4], caller zeroes; may be NULL). */
size_t subgacc_sjoin_workspace_bytes(int64_t S);
int subgacc_sjoin_sizes(const int64_t *spg_indptr, int64_t n_rows, const int64_t *own, const int64_t *partner, int64_t S,
                        int64_t *out_seg, int32_t *flags, void *workspace, size_t workspace_bytes, void *stream);
int subgacc_sjoin_sizes_rows(const int32_t *row_len, int64_t n_rows, const int64_t *own, const int64_t *partner, int64_t S,
                             int64_t *out_seg, int32_t *flags, void *workspace, size_t workspace_bytes, void *stream);   /* strided rows */

/* ABI 6 -- ONE entry point fills the R = out_seg[S] rows of every form of the join: the descriptor states what the store looks
 * like, what a member's payload is, which segments to join and which outputs are wanted; subgacc_sjoin_fill_v2 dispatches.
 * (ABI 7: it is the ONLY fill entry point -- the seven per-form entry points of ABI 1-5, forwards since ABI 6, are gone.)
 *
 *   store      exactly one of three row layouts.  PACKED rows: row_off [n_rows+1] (the SpG of random_walks.py:79).  STRIDED rows:
 *              row_len [n_rows] + row_stride (the form subgacc_walk_spg leaves a transient batch in -- row r = [r*row_stride,
 *              +row_len[r]) -- joined where they lie, no packed copy).  HEADED rows (ABI 7; row_off = row_len = NULL, row_stride > 1):
 *              a RESIDENT store on whole 128-byte lines (subgacc_rows_to_headed writes it; row_stride a multiple of 32) -- slot 0 of
 *              row r's ids, ids[r*row_stride], holds the row's LENGTH, its members follow at ids[r*row_stride + 1 + t], member t's
 *              payload is payload[r*row_stride + t]: a row needs no row pointer (its place is r*row_stride, its length arrives
 *              with its first members), begins on a line and ends inside its own last one -- what a packed store reads beyond
 *              the algorithmic bytes (rows that begin and end inside lines, a line per row pointer) is gone, for row_stride /
 *              (mean length) of its size.  ids: member ids, ascending inside a row; max_len >= the longest row touched (packed
 *              rows; a longer row sets flags[3] |= 1 and its segment is skipped; strided rows: row_stride is the bound, headed
 *              rows: row_stride - 1).  Mirrored lists of rows that do not fit LDS (~13k members) and lists that are not mirrored
 *              take a one-segment-per-wave kernel (packed rows only); rows beyond that (~20k int / ~13k float members) are
 *              searched in place.
 *   payload    SUBGACC_JOIN_SFPTR  int32: SFptr+1 (packed rows) or a slot of `uniq_table` (strided rows: slots become SFptr+1 through
 *                                  the numbered table's id plane on their way in; uniq_table = NULL: `table` is indexed by slot+1
 *                                  itself, subgacc_unpack_lp(zero_row = 1)); feature rows are gathered from table f32 [table_rows, k]
 *                                  (Z_SF with the zero row; an SFptr outside it is never read: flags[3] |= 2)
 *              SUBGACC_JOIN_F64    double: the PPR encoder's score (train.py:39-43); xz is [R,2,1] = (float(own), float((partner or
 *                                  0.0) + 1.0 - 1.0)), the SciPy expression of train.py:33 in double
 *              SUBGACC_JOIN_KEY32  int32 LP key (key = sum_j count_j << ((num_steps - j) * SHIFT), bit num_steps*SHIFT set on a root's
 *                                  own row, subg_acc.c:900-955): a feature row is the key's unpacked counts / num_walks -- what
 *                                  subgacc_unpack_lp writes into the feature table, computed on the fly (main.py:174's IEEE division);
 *                                  partner absent = the zero row.  Needs num_steps*SHIFT+1 <= 31.  Packed rows: a store re-keyed once
 *                                  (SpG.keyed); strided rows: what subgacc_walk_spg(uniq_table = NULL) writes
 *              SUBGACC_JOIN_KEY64  uint64 LP key: the strided rows of subgacc_walk_keyrows64 (4 hops, 32..63 bits); strided / headed rows
 *   segments   own / partner [S], seg [S+1] from subgacc_sjoin_sizes(_rows); pair_block = 0: independent segments (SFPTR / F64, packed
 *              rows); pair_block = P > 0: blocks of P segments, block 2t+1 mirrors block 2t (own / partner swapped) -- gather passes
 *              P = B, hgather P = B, nb batches at once P = B -- the two rows of a pair are read once and both blocks produced from
 *              there (flags[3] |= 4 if the list is not mirrored like that); `partner` may then be NULL.  Strided rows, keys, the count
 *              and the pair form: mirrored lists only
 *   form       SUBGACC_JOIN_ROWS   out_xz f32 [R,2,k] and / or out_idx i32 [R,2] (the raw index pairs, SFPTR only), out_segid i64 [R]
 *                                  (segment id of every row, `ptr=False`, train.py:25-30; optional; not with strided rows)
 *              SUBGACC_JOIN_COUNTS out_counts f32 [S, table_rows]: how often LP row p (SFptr+1, 0 = partner absent) occurs in either
 *                                  feature slot of segment j, so that segment_sum_j(MLP(xz).sum(-2)) == out_counts[j] @ MLP(Z_SF)
 *                                  (model.py:78-83); 8*max_len + 8*table_rows + 16 bytes of LDS <= 160 KiB (SUBGACC_ERR_LDS)
 *              SUBGACC_JOIN_PAIRS  for aggregations that are not linear in the rows (the attention gate, model.py:59-62): segment j as
 *                                  its DISTINCT index pairs with multiplicities, in a reproducible order, at rows [seg[j], seg[j] +
 *                                  out_cnt[j]) of out_pairs i32 [R,2], out_mult i32 [R]; out_cnt i32 [S]; max_len <= 1024
 *   options    SUBGACC_JOIN_OPT_SIZES (row form): the WHOLE join of a batch in this one call -- the size pass runs first, as ONE
 *              launch, and the fill behind it.  seg = NULL; out_seg [S+1] is written (what subgacc_sjoin_sizes writes) and read by
 *              the fill; the outputs must hold the worst case (S * max_len rows, R is not known to the host beforehand); the size
 *              pass ORs its status into flags[3] (16: a row number outside the store; 64: size_state was not clean -- the segment pointers
 *              mean nothing, the fill of this call, and of every later one that finds the bit set, writes no row: zero the state
 *              and flags and call again); host_tail (optional; int64[2] of pinned, device-visible host memory) receives [R, the status of THIS
 *              size pass] when it ends (R = -1 with bit 64), so that a serving loop needs neither a memset in front of the join
 *              nor a copy behind it.  size_state: subgacc_sjoin_workspace_bytes(S) bytes of device memory the caller zeroes ONCE, when allocating
 *              it: every call leaves it zeroed again (a single-pass scan keeps its ticket and one word per tile there); one
 *              state serves one join at a time (calls on one stream; one state per stream otherwise).
 *              With NO output (out_xz = out_idx = NULL) the call is the size pass alone: out_seg and host_tail are written and
 *              nothing else -- the "count" half of a two-call pattern (then: allocate R rows, call again with seg = that out_seg
 *              and without the option) for callers that do not hold a worst-case buffer.
 *   struct_bytes = sizeof(subgacc_join_desc): a descriptor of another size is refused (SUBGACC_ERR_BADARG); fields a form does not
 *   read must be zero / NULL. */
enum { SUBGACC_JOIN_SFPTR = 0, SUBGACC_JOIN_F64 = 1, SUBGACC_JOIN_KEY32 = 2, SUBGACC_JOIN_KEY64 = 3 };
enum { SUBGACC_JOIN_ROWS = 0, SUBGACC_JOIN_COUNTS = 1, SUBGACC_JOIN_PAIRS = 2 };
enum { SUBGACC_JOIN_OPT_SIZES = 1 };
typedef struct subgacc_join_desc {
    int32_t struct_bytes, form, payload_kind, max_len;
    const int64_t *row_off;
    const int32_t *row_len;
    int64_t row_stride, n_rows;
    const int32_t *ids;
    const void *payload;
    const void *uniq_table;
    int64_t uniq_capacity;
    const int64_t *own, *partner;
    int64_t S;
    const int64_t *seg;
    int64_t pair_block;
    const float *table;
    int64_t table_rows;
    int32_t k, num_walks, num_steps, options;
    float *out_xz;
    int32_t *out_idx;
    int64_t *out_segid;
    float *out_counts;
    int32_t *out_pairs, *out_mult, *out_cnt;
    int32_t *flags;
    int64_t *out_seg;           /* options & SUBGACC_JOIN_OPT_SIZES */
    void *size_state;
    int64_t size_state_bytes;
    int64_t *host_tail;
} subgacc_join_desc;
int subgacc_sjoin_fill_v2(const subgacc_join_desc *d, void *stream);


/* Packed rows -> headed rows (ABI 7): the resident store of a serving loop laid out on whole lines -- the rows random_walks.py:79-81
 * builds as a SciPy CSR and train.py:17-18 / :39-43 slice one by one (x[edge[0]]), in the layout the pair kernels read with one
 * dependent trip less.  row_off [n+1], ids, payload
 * (payload_bytes = 4: SFptr+1 / 32-bit keys, 8: PPR scores / 64-bit keys) as SpG holds them; out_ids [n*row_stride] int32 and
 * out_payload [n*row_stride] of the same element size, row_stride >= (longest row) + 1 (a longer row is cut and flags[3] |= 1), a
 * multiple of 32 for rows on whole 128-byte lines.  Slots behind a row's members are left as they are. */
int subgacc_rows_to_headed(const int64_t *row_off, int64_t n_rows, const int32_t *ids, const void *payload, int32_t payload_bytes,
                           int64_t row_stride, int32_t *out_ids, void *out_payload, int32_t *flags, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Top-K approximate-PPR node sets -- replaces sampler/pprgo.py:9-38 (_calc_ppr_node, the
 * Andersen-Chung-Lang push with a LIFO work list), :53-63 (calc_ppr_topk_parallel), :85-111 (topk_ppr_matrix
 * normalisation) and utils.py:35-36 (encoding 'PPR').  The graph is an unweighted CSR without repeated entries
 * in a row (deg = row length, as np.sum(adj > 0, 1) / adj.sum(1) of pprgo.py:69,92 give for such a graph).
 * The push order and every float32 rounding follow the reference (numba typing: alpha, epsilon, p, r float32;
 * (1 - alpha) * res / deg in float64, rounded on the store), so rows are reproducible bit for bit; equal scores
 * at the top-k cut keep the node that entered p later.
 *
 * One wavefront works on one root at a time in a private `slab` of 24 << table_log2 bytes (hash table of p, r
 * and the work list).  num_waves slabs = num_waves resident wavefronts; reset the slabs once, the kernel hands
 * them back clean (table_log2 in [10, 26]).  A root that touches more than (1 << table_log2) / 2 nodes (bounded by 1/(alpha*epsilon))
 * gets out_count = -1 and flags[2] |= 1: run those roots again with a larger table.
 *   out_count [n] int32, out_ids [n*topk] int32 ascending ids of row i at i*topk, out_vals [n*topk] float32
 *   pushes (optional, device uint64[2]) accumulates the number of pushes and of touched nodes.  topk <= 4096.
 * ------------------------------------------------------------------------------------------- */
size_t subgacc_ppr_slab_bytes(int32_t table_log2, int32_t num_waves);
int subgacc_ppr_slab_reset(void *slab, int32_t table_log2, int32_t num_waves, void *stream);
int subgacc_ppr_topk(const void *indptr, int32_t indptr64, const int32_t *indices, int64_t num_nodes,
                     const int32_t *roots, int64_t n, float alpha, float epsilon, int32_t topk, void *slab,
                     int32_t table_log2, int32_t num_waves, int32_t *out_count, int32_t *out_ids, float *out_vals,
                     int32_t *flags, uint64_t *pushes, void *stream);
/* Packed rows (row_off[n+1], ids, vals) -> float64 payload.  mode 0 'row': unchanged; 1 'sym':
 * sqrt(max(deg_root,1e-12)) * v * (1/sqrt(max(deg_col,1e-12))); 2 'col': deg_root * v * (1/max(deg_col,1e-12))
 * (pprgo.py:88-108, evaluated left to right in float64).  max_nnz >= row_off[n] sizes the launch.  max_bits
 * (optional, device uint64, zeroed by the caller) receives the bit pattern of the largest output value. */
int subgacc_ppr_normalize(const void *indptr, int32_t indptr64, const int32_t *roots, int64_t n,
                          const int64_t *row_off, int64_t max_nnz, const int32_t *ids, const float *vals, int32_t mode,
                          double *out, uint64_t *max_bits, void *stream);
/* utils.py:35-36: data = (data + 0.1) / (max + 0.1) over the first *nnz_dev entries; max from max_bits. */
int subgacc_ppr_encode(double *data, int64_t max_nnz, const int64_t *nnz_dev, const uint64_t *max_bits, void *stream);

/* ---------------------------------------------------------------------------------------------
 * DEG and SPD structural encoders over a PPR node-set store (utils.py:22-34, `encoding(x, adj, 'DEG' | 'SPD')`):
 * per row the union of the PPR row x_i (x_off / x_ids / x_val, ids ascending) and the adjacency row N(i), with
 *   mode 1 DEG  value(i,j) = log(|x_j u N(j)| + 1)  (log_table[d] = log(d+1), made by the caller with the libm the
 *               reference uses); out_agg(i,j) = x_ij + 1/deg(i)  (`x += normalize(adj, 'l1')`, the `agg` return)
 *   mode 2 SPD  value(i,j) = 1[j in N(i)] + 0.5[j in x_i and N(i) n N(j) != {}] + 0.3[j in x_i], diagonal = 2.3
 *               (x1 + x0.multiply(x1**2 * 0.5) + x0 * 0.3; setdiag(2.3)); the adjacency is taken as symmetric.
 * Rows = nodes (n = number of nodes), adjacency rows ascending without repeats.  Two calls: subgacc_encode_sizes
 * gives row_len[n] (for DEG these are also the d of the value rule); scan them into out_off[n+1]; then
 * subgacc_encode_fill writes the merged rows (ids ascending) -- kmax >= longest x row (<= 8192).
 * flags[3] |= 8: a row length outside log_table.
 * ------------------------------------------------------------------------------------------- */
int subgacc_encode_sizes(const int64_t *x_off, const int32_t *x_ids, int64_t n, const void *indptr, int32_t indptr64,
                         const int32_t *indices, int32_t mode, int32_t *row_len, int32_t *flags, void *stream);
int subgacc_encode_fill(const int64_t *x_off, const int32_t *x_ids, const double *x_val, int64_t n, int32_t kmax,
                        const void *indptr, int32_t indptr64, const int32_t *indices, int32_t mode,
                        const int32_t *deg_row_len, const double *log_table, int64_t log_len, const int64_t *out_off,
                        int32_t *out_ids, double *out_val, double *out_agg, int32_t *flags, void *stream);

/* ---------------------------------------------------------------------------------------------
 * walk_join of the legacy SUREL surface (subg_acc/subg_acc.c:509-647): for every query pair (a, b) of roots and
 * every position t of their raw walks, the running index of the visited node in a's key list and in b's key list
 * (find_idx, :78-92; 0 = absent).
 *   walks   int32 [n, stride]           raw walks, row i belongs to root i (walk_sampler's first output)
 *   set_*   the n key lists as SpG-form rows: set_off int64[n+1], set_ids int32 ascending per row, set_idx int32 =
 *           1 + position of the member in the concatenation of the lists as given (:573-584) -- what
 *           subgacc_spg_build makes of (row_off, ids, sf = 0..X-1)
 *   qrow    int32 [Q,2] row numbers of the query keys (find_key_item, :617; -1 = not a root: that pair yields -1s)
 *   out     int32 [2, Q*2*stride], 8-byte aligned: out[s][2*x*stride + 2*t + {0,1}] = index of walks[row_s(x)][t] in
 *           the list of {key1, key2} of pair x  (:621-629)
 * max_len >= longest key list (sizes the LDS staging; rows past 5120 members are searched in place).
 * ------------------------------------------------------------------------------------------- */
int subgacc_walk_join(const int32_t *walks, int64_t n, int32_t stride, const int64_t *set_off, const int32_t *set_ids,
                      const int32_t *set_idx, int32_t max_len, const int32_t *qrow, int64_t Q, int32_t *out,
                      void *stream);

/* ---------------------------------------------------------------------------------------------
 * Key rows: a batch that is sampled, joined and dropped needs neither the table of distinct LP rows nor their numbering.
 * subgacc_walk_spg with uniq_table = NULL writes the member's 32-bit LP key itself as the row's payload (needs
 * num_steps*SHIFT+1 <= 31, 2 to 4 hops, set_sampler order, no bucket, M <= 256: SUBGACC_ERR_BADARG otherwise); such rows are
 * joined with payload SUBGACC_JOIN_KEY32 (subgacc_join_desc): same (xz, seg) as the table path, sizes by subgacc_sjoin_sizes_rows.
 * A PACKED store whose payload was re-keyed once (SpG.keyed) is joined the same way: for the reference's flow -- subg_matrix over all
 * nodes once, main.py:172-178, then one join per training batch, train.py:120-127 -- that takes the gather from the Z_SF table out of
 * every output row; xz is bit-identical to the SFPTR join with table = float32(enc) / num_walks, main.py:174.
 *
 * ABI 5 -- key rows for the 4-hop configurations.  The paper's sampler figure is citation2 with m = 4, M = 200 (Fig. 6a): its LP
 * key -- the reference's 64-bit `bithash`, subg_acc.c:900-955 -- takes 4 x 8 + 1 = 33 bits.  subgacc_walk_spg(uniq_table = NULL)
 * serves 2 to 4 hops while num_steps*SHIFT+1 <= 31 (4 hops: M <= 127, e.g. the reference's own citation2 setting M = 100,
 * README.md:82-96); beyond that, subgacc_walk_keyrows64 writes the same rows with the whole 64-bit key as payload:
 *   row_ids [n*stride] int32 (sorted by node id), row_keys [n*stride] uint64, nsize [n]; stride = M*m+1.
 * Shapes: 4 hops, 32 <= num_steps*SHIFT+1 <= 63, a 1,024-slot table (M*4+1 <= 818), set_sampler order, no bucket.  worklist /
 * n_work: optional (both or neither), as for subgacc_walk_spg_list / _sparse (rows that are not listed are left alone: zero nsize
 * first); rng_pos / rng_seed as for subgacc_walk_spg.  Joined with payload SUBGACC_JOIN_KEY64: bit for bit the xz of the table path.
 * ------------------------------------------------------------------------------------------- */
int subgacc_walk_keyrows64(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                           const int32_t *query, int64_t n, const uint32_t *rng_pos, const uint32_t *rng_seed,
                           const int32_t *worklist, const int64_t *n_work, int32_t *row_ids, uint64_t *row_keys,
                           int32_t *nsize, int32_t *flags, void *stream);
/* ---------------------------------------------------------------------------------------------
 * Prologue of one on-demand step (sample both endpoints of B query pairs -> rows -> SpJoin; train.py:120-127 calls the
 * join once per batch of main.py:32's 1,024 pairs) in ONE launch: subgacc_uniq_reset of the table of distinct LP rows,
 * `n_zero` status words zeroed, and the n = 2B endpoints `edge` (int64, [u.. | v..]) narrowed to the int32 roots the
 * sampler takes (an id outside int32 becomes -1: out of range for the walk kernel, which flags it).
 * ------------------------------------------------------------------------------------------- */
int subgacc_step_prologue(void *uniq_table, int64_t capacity, int64_t *zero_words, int64_t n_zero, const int64_t *edge,
                          int32_t *roots, int64_t n, void *stream);   /* uniq_table may be NULL (key rows: no table) */

/* The same prologue for a step that samples every DISTINCT endpoint once.  Philox keys a walk by (seed, root id, walk, hop), so a
 * root's set does not depend on where or how often the root stands in the batch (the reference samples every node once, offline,
 * main.py:172-178; its sequential rand_r stream has no such property, so this form is Philox only).  Two launches:
 *   roots       int32 [n]: roots[j] = endpoint j where j is the FIRST occurrence of its node in the batch, SUBGACC_NO_ROOT
 *               elsewhere -- rows of the batch's sets stay where the plain step has them, the rows of repeated endpoints
 *               stay empty (subgacc_walk_spg_sparse passes over them), and the distinct LP rows keep their numbering
 *   own, partner int64 [n]: gather()'s mirrored segment lists over those rows -- own[j] = row of endpoint j's first occurrence,
 *               partner[j] = that of the other end of its pair (j +- n/2); for subgacc_sjoin_sizes_rows / _fill_rows / _fill_keyrows
 *   worklist    int32 [n]: the first occurrences, worklist[0 .. *n_distinct), in order of arrival (not reproducible, and it does
 *               not matter: entry k names its row) -- what subgacc_walk_spg_sparse runs over
 *   row_len     int32 [n]: the rows' lengths (the sampler's nsize): set to 0 for the rows of repeated endpoints, which the
 *               sampler will not visit
 *   n_distinct  int64 [1] (device): the number of distinct endpoints = the length of the work list
 *   workspace   subgacc_step_dedup_workspace_bytes(n) bytes, ZEROED once by the caller before its first use and left alone
 *               afterwards: it keeps the stamp of the last step (slots are stamped, never cleared), so a captured (replayed)
 *               step works like a launched one
 * ------------------------------------------------------------------------------------------- */
#define SUBGACC_NO_ROOT (-2147483647 - 1)
size_t subgacc_step_dedup_workspace_bytes(int64_t n);
int subgacc_step_prologue_dedup(void *uniq_table, int64_t capacity, int64_t *zero_words, int64_t n_zero, const int64_t *edge,
                                int32_t *roots, int64_t *own, int64_t *partner, int32_t *worklist, int32_t *row_len, int64_t n,
                                void *workspace, size_t workspace_bytes, int64_t *n_distinct, void *stream);
/* subgacc_walk_spg over some of the rows only: the rows worklist[0 .. *n_work) (both on the device; the launch covers n rows,
 * blocks past the list's length leave at once), or -- worklist = n_work = NULL -- every row i whose query[i] is not
 * SUBGACC_NO_ROOT (such a row gets nsize[i] = 0 and nothing else is touched; rows that are not on the work list are not touched
 * at all).  Philox mode, set_sampler order, shapes the fused-row kernel serves (2..4 hops, M <= 256, M*m+1 <= 818, no bucket):
 * SUBGACC_ERR_BADARG otherwise.  Row i belongs to query[i]; tags of the table of distinct rows start at 0. */
int subgacc_walk_spg_sparse(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                            const int32_t *query, int64_t n, const int32_t *worklist, const int64_t *n_work, void *uniq_table,
                            int64_t uniq_capacity, int32_t *row_ids, int32_t *row_slot, int32_t *nsize, int32_t *flags,
                            void *stream);

/* ---------------------------------------------------------------------------------------------
 * ABI 4 -- batched registration of key rows: the table of distinct LP rows (subg_acc.c:957-978) for a store that is KEPT
 * (subg_matrix, sampler/random_walks.py:74-82; main.py:172-178 samples all N nodes once), built from the rows of the fast
 * key-rows form of subgacc_walk_spg (uniq_table = NULL) instead of inside every root's walk:
 *   1. subgacc_keyrows_register: one pass over the rows' 32-bit LP keys (row i = row_keys[i*stride, +nsize[i])); every
 *      distinct key of a row goes to `uniq_table` with the COARSE tag (root_base+i)*stride + stride-1, and the rows whose
 *      insert lowered a key's tag -- the candidates for a key's first appearance, always including the true one -- are
 *      listed in cand[0 .. *n_cand) (int32 row numbers in arrival order; cand holds n entries, *n_cand is a device counter
 *      the caller zeroes);
 *   2. subgacc_walk_tags: the candidates alone (worklist = cand, *n_work = *n_cand; the launch covers min(n, work_cap)
 *      rows, work_cap = 0: n; a longer list raises flags[3] |= 32) are walked again by the table form of the fused-row
 *      kernel, which lowers their keys' tags to the exact (root_base+i)*stride + first-visit number and writes nothing
 *      else -- after it every distinct key carries the tag subgacc_walk_spg(uniq_table) would have given it, and
 *      subgacc_uniq_number numbers the table like the reference's sequential pass;
 *   3. subgacc_keyrows_compact: strided key rows -> packed rows at row_off[i] (the CSR copy of the store).  With `ukeys`
 *      (out_ukeys / out_count / max_unique of subgacc_uniq_number: one chunk of roots -- steps 1, 2, the numbering, then
 *      this) the payload is SFptr+1, looked up on the way.  With ukeys = NULL (a job of several chunks, numbered at the
 *      end) the pass registers the chunk's keys itself (as step 1, candidates listed; step 2 follows) and keeps the KEY as
 *      payload; subgacc_keyrows_translate turns the packed payloads of all chunks into SFptr+1 once the table is numbered
 *      (n_dev, optional: only the first min(n, *n_dev) entries).
 * Shapes: what the key-rows form of subgacc_walk_spg serves (num_steps*SHIFT+1 <= 31, 2 to 4 hops, M <= 256, no bucket).
 * rng_pos / rng_seed as for subgacc_walk_spg (RAND_R: positions of ALL n roots of the chunk; row i reads entry i).
 * cand: subgacc_keyrows_cand_capacity(n) entries (cand_cap says how many there are; every row is listed at most once). */
int64_t subgacc_keyrows_cand_capacity(int64_t n);
int subgacc_keyrows_register(const int32_t *row_keys, const int32_t *nsize, int64_t n, int32_t stride,
                             int64_t root_base, void *uniq_table, int64_t uniq_capacity, int32_t *cand, int64_t cand_cap,
                             int64_t *n_cand, int32_t *flags, void *stream);
int subgacc_walk_tags(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                      const int32_t *query, int64_t n, int64_t root_base, const uint32_t *rng_pos, const uint32_t *rng_seed,
                      const int32_t *worklist, const int64_t *n_work, int64_t work_cap, void *uniq_table,
                      int64_t uniq_capacity, int32_t *flags, void *stream);
int subgacc_keyrows_compact(const int32_t *row_ids, const int32_t *row_keys, const int32_t *nsize, const int64_t *row_off,
                            int64_t n, int32_t stride, int64_t root_base, void *uniq_table, int64_t uniq_capacity,
                            const uint64_t *ukeys, const int64_t *n_ukeys, int64_t max_ukeys, int32_t *out_indices,
                            int32_t *out_data, int32_t *cand, int64_t cand_cap, int64_t *n_cand, int32_t *flags, void *stream);
int subgacc_keyrows_translate(int32_t *data_inout, int64_t n, const int64_t *n_dev, void *uniq_table, int64_t uniq_capacity,
                              const uint64_t *ukeys, const int64_t *n_ukeys, int64_t max_ukeys, void *stream);

/* ABI 4: subgacc_walk_spg over ALL n rows but in the order of a work list (worklist[0 .. *n_work) names every row once:
 * subgacc_worklist_by_root) -- either RNG mode: row i keeps its place in the batch AND in the rand_r stream (rng_pos[i] /
 * rng_seed[i] from subgacc_rng_positions over the n roots in batch order; NULL for Philox), only the order in which the kernel
 * takes the rows changes.  Tags of the table of distinct rows start at 0.  Shapes: any that subgacc_walk_spg takes with a
 * table of distinct rows -- where the fused-row kernel declines (a truncating bucket, M > 256, ...) the general kernel walks
 * the rows in batch order, which gives the same rows; key rows (uniq_table = NULL) need the fused-row kernel's shapes. */
int subgacc_walk_spg_list(const subgacc_walk_cfg *cfg, const void *indptr, const int32_t *indices, int64_t num_nodes,
                          const int32_t *query, int64_t n, const uint32_t *rng_pos, const uint32_t *rng_seed,
                          const int32_t *worklist, const int64_t *n_work, void *uniq_table, int64_t uniq_capacity,
                          int32_t *row_ids, int32_t *row_slot, int32_t *nsize, int32_t *flags, void *stream);

/* ---------------------------------------------------------------------------------------------
 * batch_sampler of the legacy SUREL surface (subg_acc/subg_acc.c:391-507): one insertion-ordered set of nodes grown by
 * walking the roots one after the other (num_walks walks of num_steps nodes each, first hop without replacement,
 * one rand_r stream); root i stops walking once the set holds (i+1)*thld/n nodes (:474).
 *   seed_eff   the state the reference starts from: its `seed` argument + getpid() (:421)
 *   out        int32 [out_cap] nodes in insertion order; *out_count (device) receives their number.
 *              A sufficient out_cap is min(num_nodes, n*(num_walks*num_steps+1)).  flags[1] |= 1 when it was too small,
 *              flags[0] |= 1 when a walk reached a node without out-edges (the stream position of the later walks is then
 *              data dependent and not reproduced), flags[3] |= 16 for a root outside [0, num_nodes) (skipped).
 *   workspace  subgacc_batch_sampler_workspace_bytes(out_cap) bytes (the set's hash table)
 * One workgroup: the loop over roots is sequential by definition; walks, dedup and ordering inside a root are parallel.
 * ------------------------------------------------------------------------------------------- */
size_t subgacc_batch_sampler_workspace_bytes(int64_t out_cap);
int subgacc_batch_sampler(const void *indptr, int32_t indptr64, const int32_t *indices, int64_t num_nodes,
                          const int32_t *query, int64_t n, int32_t num_walks, int32_t num_steps, int32_t thld,
                          uint32_t seed_eff, int32_t *out, int64_t out_cap, int64_t *out_count, void *workspace,
                          size_t workspace_bytes, int32_t *flags, void *stream);

/* ABI 4: the rows 0 .. n-1 of a batch as a work list in ascending order of their root's id (1,024 buckets of consecutive
 * ids, any order inside a bucket; rows whose root is SUBGACC_NO_ROOT -- the repeated endpoints subgacc_step_prologue_dedup
 * marks -- are left out; *n_work = the rows listed) -- what subgacc_walk_spg_sparse / _list then runs over: roots that are neighbours in id
 * space (the same community of a graph with id locality) or equal (repeated endpoints) are walked at the same time on the
 * same XCD and share its L2.  The rows stay where they are; only the order of the walk changes, so no result does (the order
 * INSIDE a bucket is the arrival order of atomics and differs from run to run).  Two small launches; workspace =
 * subgacc_worklist_workspace_bytes(n) bytes, ZEROED ONCE by its owner before the first call -- every call leaves it zeroed for
 * the next (one workspace serves one stream at a time). */
size_t subgacc_worklist_workspace_bytes(int64_t n);
int subgacc_worklist_by_root(const int32_t *roots, int64_t n, int64_t num_nodes, int32_t *worklist, int64_t *n_work,
                             void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SUBGACC_H */
