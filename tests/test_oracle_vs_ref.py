"""CPU, build container only: randomized sweep of the oracle (oracle/subgacc_oracle.c) against the REFERENCE ITSELF
(oracle/_ref = /root/reference/subg_acc/subg_acc.c compiled by `make -C oracle ref`), beyond the 46 committed
fixtures: random graphs (isolated nodes, star hubs, K2 components), random M / m / bucket / seeds / query shapes.

    gset_sampler  nthread=1 (the only reproducible setting, subg_acc.c:731-732)        subg_acc.c:649-1034
    walk_sampler  nthread 1..8, both first-hop modes                                   subg_acc.c:316-389
    walk_join     over walk_sampler's own output                                       subg_acc.c:509-647
    batch_sampler seeded with seed + getpid() by the reference (:421): same process, so the sum is known   subg_acc.c:391-507

Skips cleanly where oracle/_ref does not exist (the GPU box may or may not carry it; nothing here needs a GPU).
Every case has its own seeded stream: a failing case number reproduces alone."""
import ctypes

import numpy as np
import pytest

import oracle

ref = oracle.ref_module()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (needs /root/reference; `make -C oracle ref`)")

CASES = 240


def rand_graph(rng):
    import scipy.sparse as sps
    N = int(rng.integers(4, 400))
    E = int(rng.integers(1, 6 * N))
    iso = int(rng.integers(0, 4))
    star = int(rng.integers(0, N - 1)) if rng.random() < 0.4 else 0
    k2 = rng.random() < 0.3
    r, c = rng.integers(0, N, E), rng.integers(0, N, E)
    if star:
        r = np.concatenate([r, np.zeros(star, int)])
        c = np.concatenate([c, rng.choice(np.arange(1, N), star, replace=False)])
    tot = N + iso + (2 if k2 else 0)
    if k2:
        r, c = np.concatenate([r, [tot - 2]]), np.concatenate([c, [tot - 1]])
    A = sps.csr_matrix((np.ones(len(r)), (r, c)), shape=(tot, tot))
    A = sps.csr_matrix(A + A.T)          # symmetrised like dataloader.py:122-135: no dead ends
    A.setdiag(0)
    A.eliminate_zeros()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


def rand_query(rng, n_nodes):
    kind = rng.integers(0, 4)
    if kind == 0:
        return np.arange(n_nodes)
    if kind == 1:                                            # repeated, unordered roots
        return rng.integers(0, n_nodes, int(rng.integers(1, 2 * n_nodes)))
    if kind == 2:
        return np.sort(rng.choice(n_nodes, int(rng.integers(1, n_nodes + 1)), replace=False))
    return rng.integers(0, n_nodes, int(rng.integers(1, 9)))  # a handful (fewer roots than threads, below)


def rand_walk_params(rng):
    M = int(rng.choice([1, 2, 3, 5, 8, 16, 20, 31, 32, 64, 100, 200]))
    shift = int(M).bit_length()
    m = int(rng.integers(1, min(6, 63 // shift) + 1))       # m*SHIFT+1 <= 64 (subg_acc.c:905-915)
    return M, m


def mask_isolated(ptr, query, nsize, remap):
    """the reference leaves the id word of an isolated root uninitialised (subg_acc.c:753-761)"""
    remap = remap.copy()
    deg = np.diff(ptr)[query]
    off = np.concatenate([[0], np.cumsum(nsize)])[:-1]
    remap[0, off[deg == 0]] = np.asarray(query)[deg == 0]
    return remap


@pytest.mark.parametrize("block", range(8))
def test_gset_sampler_random_sweep(block, capfd):
    for case in range(block * CASES // 8, (block + 1) * CASES // 8):
        rng = np.random.default_rng(10_000 + case)
        ptr, idx = rand_graph(rng)
        q = rand_query(rng, len(ptr) - 1)
        M, m = rand_walk_params(rng)
        bucket = int(rng.integers(2, M * m + 2)) if rng.random() < 0.3 else -1
        seed = int(rng.integers(0, 2**31 - 1))
        nsize, remap, enc, raw = ref.gset_sampler(ptr, idx, q, num_walks=M, num_steps=m, bucket=bucket, nthread=1,
                                                  seed=seed, debug=1)
        remap = mask_isolated(ptr, q, nsize, remap)
        o = oracle.gset_sampler(ptr, idx, q, num_walks=M, num_steps=m, bucket=bucket, seed=seed, debug=True)
        what = f"case {case}: N={len(ptr) - 1} n={len(q)} M={M} m={m} bucket={bucket} seed={seed}"
        assert np.array_equal(o[0], nsize), what
        assert np.array_equal(o[1], remap), what
        assert np.array_equal(o[2], enc), what
        assert np.array_equal(o[3], raw), what
    ctypes.CDLL(None).fflush(None)
    capfd.readouterr()        # the reference printf()s statistics


@pytest.mark.parametrize("block", range(8))
def test_walk_sampler_and_walk_join_random_sweep(block, capfd):
    for case in range(block * CASES // 8, (block + 1) * CASES // 8):
        rng = np.random.default_rng(20_000 + case)
        ptr, idx = rand_graph(rng)
        q = rand_query(rng, len(ptr) - 1).astype(np.int32)
        M, m = rand_walk_params(rng)
        T = int(rng.integers(1, 9))
        rep = bool(rng.integers(0, 2))
        seed = int(rng.integers(0, 2**31 - 1))
        walks, obj = ref.walk_sampler(ptr, idx, q, num_walks=M, num_steps=m, nthread=T, seed=seed, replacement=rep)
        ow, onsize, oids, ocounts = oracle.walk_sampler(ptr, idx, q, num_walks=M, num_steps=m, nthread=T, seed=seed,
                                                        replacement=rep)
        what = f"case {case}: N={len(ptr) - 1} n={len(q)} M={M} m={m} T={T} rep={rep} seed={seed}"
        assert np.array_equal(ow, walks), what
        assert np.array_equal(onsize, [len(obj[i, 0]) for i in range(len(q))]), what
        assert np.array_equal(oids, np.concatenate([obj[i, 0] for i in range(len(q))])), what
        assert np.array_equal(ocounts, np.concatenate([obj[i, 1] for i in range(len(q))])), what
        # the legacy join over these walks (pairs of sampled roots, (u, u) included).  Distinct roots only: with a root
        # repeated in `walk` the reference adds duplicate keys to a uthash table (HASH_ADD_INT, subg_acc.c:563) --
        # which of them HASH_FIND returns flips with every bucket expansion (expansion re-chains a bucket in reverse),
        # i.e. depends on how many other roots there are.  The goldens' small `dup` case (no expansion: the last row
        # wins) is what the oracle and the HIP path implement; the reference's callers pass unique nodes.
        first = np.sort(np.unique(q, return_index=True)[1])
        uq, uw, ukeys = q[first], walks[first], [obj[i, 0] for i in first]
        Q = int(rng.integers(1, 40))
        pairs = uq[rng.integers(0, len(uq), (Q, 2))]
        pairs[0] = (uq[0], uq[0])
        out, xrow = ref.walk_join(uw, ukeys, pairs, nthread=1, return_idx=True)
        oout, oxrow = oracle.walk_join(uw, ukeys, pairs, return_idx=True)
        assert np.array_equal(oout, out) and np.array_equal(oxrow, xrow), what
    ctypes.CDLL(None).fflush(None)
    capfd.readouterr()


@pytest.mark.parametrize("block", range(4))
def test_batch_sampler_random_sweep(block):
    import os
    for case in range(block * CASES // 4, (block + 1) * CASES // 4):
        rng = np.random.default_rng(30_000 + case)
        ptr, idx = rand_graph(rng)
        N = len(ptr) - 1
        q = rand_query(rng, N).astype(np.int32)
        M = int(rng.integers(1, 60))
        S = int(rng.integers(1, 10))
        thld = int(rng.integers(1, 4 * N))
        seed = int(rng.integers(0, 2**30))
        r = ref.batch_sampler(ptr, idx, q, num_walks=M, num_steps=S, thld=thld, seed=seed)
        o = oracle.batch_sampler(ptr, idx, q, num_walks=M, num_steps=S, thld=thld, seed_eff=seed + os.getpid())
        assert np.array_equal(o, r), f"case {case}: N={N} n={len(q)} M={M} S={S} thld={thld} seed={seed}"
