"""GPU: the path at BASELINE.json's full sizes, checked through size-independent properties (and, on a random
subset of the roots, bit-exactly against the oracle).  These are the slow tests (tens of seconds each)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    import surel_plus_amd
    return surel_plus_amd


def _row_checks(z, table, roots, M, sample):
    """properties of SpG rows `sample` (row i holds the set of roots[i])"""
    ip = z.indptr
    for i in sample.tolist():
        lo, hi = int(ip[i]), int(ip[i + 1])
        ids = z.indices[lo:hi]
        assert hi > lo and bool((ids[1:] > ids[:-1]).all())                       # sorted, distinct
        assert bool((ids == roots[i]).any())                                      # the root is a member
        cols = table[z.data[lo:hi].long()].sum(0)
        assert torch.allclose(cols, torch.ones_like(cols), atol=1e-4)             # every LP column sums to M (/M)


def test_cit2_scale_all_roots_then_a_million_pairs(sp):
    """N = 2,927,963 roots (the offline stage of main.py:172-178) in several chunks, 1.1e9 set members."""
    from surel_plus_amd.graphs import preset_graph, query_pairs
    csr = preset_graph("cit2")
    N, M, m = csr.num_nodes, 200, 3
    roots = torch.arange(N, dtype=torch.int32, device="cuda")
    z, sets = sp.sample_spg(csr, roots, num_walks=M, num_steps=m, seed=5, rng="philox", fused=True)
    X = z.nnz
    assert X == int(sets.nsize.sum()) and X > 10**9 and z.indptr.dtype == torch.int64
    assert sets.c < 5000 and int(z.data.max()) == sets.c and int(z.data.min()) >= 1
    table = sets.feature_table()
    g = torch.Generator(device="cuda").manual_seed(0)
    sample = torch.randint(0, N, (300,), device="cuda", generator=g)
    _row_checks(z, table, roots, M, sample)
    # isolated roots keep exactly themselves
    deg = (csr.indptr[1:] - csr.indptr[:-1])
    iso = torch.nonzero(deg == 0).flatten()[:50]
    assert bool((sets.nsize[iso] == 1).all()) and bool((z.indices[z.indptr[iso]] == iso.int()).all())
    # bit-exact against the oracle on a random subset of roots (Philox sets do not depend on the batch)
    sub = sample[:120].cpu().numpy()
    o_nsize, o_remap, o_enc = oracle.gset_sampler(csr.indptr.cpu().numpy(), csr.indices.cpu().numpy(), sub, num_walks=M,
                                                  num_steps=m, seed=5, rng="philox", nthreads=8)
    oi, ox, od = oracle.spg_build(o_nsize, o_remap)
    enc_full = sets.enc_int16().cpu().numpy()
    for j, r in enumerate(sub.tolist()):
        lo, hi = int(z.indptr[r]), int(z.indptr[r + 1])
        assert np.array_equal(z.indices[lo:hi].cpu().numpy(), ox[oi[j]:oi[j + 1]])
        # LP rows agree (the numbering differs: the oracle only saw the subset)
        assert np.array_equal(enc_full[z.data[lo:hi].cpu().numpy() - 1], o_enc[od[oi[j]:oi[j + 1]] - 1])
    # the online stage: 2^20 pairs in 16 batches of 65,536
    tot = 0
    for b in range(16):
        edge = query_pairs(csr, 65536, seed=100 + b)
        xz, ind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
        lens = z.indptr[1:] - z.indptr[:-1]
        assert torch.equal(ind[1:] - ind[:-1], torch.cat([lens[edge[0]], lens[edge[1]]]))
        assert xz.shape[0] == int(ind[-1])
        if b == 0:   # first slot of every row is the member's own LP row: never the zero row
            assert bool((xz[:, 0, :].abs().sum(-1) > 0).all())
        tot += xz.shape[0]
    assert tot > 3 * 10**8
    # on demand == offline (main.py:172-178 samples all N once; sample_and_gather samples the endpoints of the batch): the
    # batch of 65,536 pairs as rows of LP keys, through preallocated step buffers, against the join over the resident store
    # (Philox sets are functions of the seed and the root) -- and against the table form of the same batch
    edge = query_pairs(csr, 65536, seed=100)
    wxz, wind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    bufs = sp.StepBuffers(csr, 65536, num_walks=M, num_steps=m)
    from surel_plus_amd import sampler as _s
    assert bufs.keyrows and (csr.hop_records() is not None) == (_s.HOP_RECORDS != "0")   # 252 MB of adjacency
    xz, ind, bsets = sp.sample_and_gather(csr, edge, num_walks=M, num_steps=m, seed=5, rng="philox", buffers=bufs)
    bsets.prefetch().resolve()
    rows = int(bsets.extra[0])
    assert rows == wxz.shape[0] and torch.equal(ind, wind) and torch.equal(xz[:rows], wxz)
    txz, tind, tsets = sp.sample_and_gather(csr, edge, num_walks=M, num_steps=m, seed=5, rng="philox", key_rows=False)
    assert not tsets.keyrows and torch.equal(tind, wind) and torch.equal(txz, wxz)


def test_twitter_scale_int64_offsets(sp):
    """41.65 M nodes, 2.9e9 adjacency entries (int64 CSR offsets); 10 M roots -> more than 2^31 set members, so
    the SpG row offsets leave the int32 range as well."""
    from surel_plus_amd.graphs import preset_graph
    csr = preset_graph("twitter")
    assert csr.indptr.dtype == torch.int64 and csr.nnz > 2**31
    M, m = 200, 2
    g = torch.Generator(device="cuda").manual_seed(3)
    roots = torch.randint(0, csr.num_nodes, (10_000_000,), device="cuda", generator=g, dtype=torch.int64).int()
    z, sets = sp.sample_spg(csr, roots, num_walks=M, num_steps=m, seed=9, rng="philox")
    X = z.nnz
    assert X > 2**31 and X == int(sets.nsize.long().sum()) and int(z.indptr[-1]) == X
    table = sets.feature_table()
    sample = torch.cat([torch.randint(0, roots.numel(), (150,), device="cuda", generator=g),
                        torch.arange(roots.numel() - 50, roots.numel(), device="cuda")])     # incl. rows past 2^31
    assert int(z.indptr[sample[-1]]) > 2**31
    _row_checks(z, table, roots, M, sample)
    # the same roots again in one small batch give the same rows (schedule independence at scale)
    sub = sample[:64]
    z2, s2 = sp.sample_spg(csr, roots[sub], num_walks=M, num_steps=m, seed=9, rng="philox")
    e1, e2 = sets.enc_int16(), s2.enc_int16()
    for j, i in enumerate(sub.tolist()):
        a = slice(int(z.indptr[i]), int(z.indptr[i + 1]))
        b = slice(int(z2.indptr[j]), int(z2.indptr[j + 1]))
        assert torch.equal(z.indices[a], z2.indices[b])
        assert torch.equal(e1[z.data[a].long() - 1], e2[z2.data[b].long() - 1])
    edge = torch.stack([sample[:100], sample[50:150]])
    xz, ind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    assert xz.shape[0] == int(ind[-1])
