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


@pytest.mark.parametrize("preset,M,m,pos_frac", [("collab", 200, 2, 0.5), ("ppa", 200, 3, 1.0 / 21.0)])
def test_collab_and_ppa_presets_at_full_size(sp, preset, M, m, pos_frac):
    """BASELINE.json configs[1] / [2] at their full N (235,868 / 576,289 nodes) under pytest, not only in bench.py: all N roots
    -> the resident store (the reference's invariants, subg_acc/test/test.py:34-45, on a sample of rows; a random subset of roots
    bit-exact against the oracle), then a batch of 65,536 pairs joined from the store, from the keyed store and on demand through
    the step buffers -- the three must agree bit for bit (Philox sets are functions of (seed, root))."""
    from surel_plus_amd.graphs import preset_graph, query_pairs
    csr = preset_graph(preset)
    N = csr.num_nodes
    roots = torch.arange(N, dtype=torch.int32, device="cuda")
    z, sets = sp.sample_spg(csr, roots, num_walks=M, num_steps=m, seed=5, rng="philox", fused=True)
    X = z.nnz
    assert X == int(sets.nsize.long().sum()) and int(z.data.max()) == sets.c and int(z.data.min()) >= 1
    table = sets.feature_table()
    g = torch.Generator(device="cuda").manual_seed(1)
    sample = torch.randint(0, N, (200,), device="cuda", generator=g)
    _row_checks(z, table, roots, M, sample)
    sub = sample[:100].cpu().numpy()
    o_nsize, o_remap, o_enc = oracle.gset_sampler(csr.indptr.cpu().numpy(), csr.indices.cpu().numpy(), sub, num_walks=M,
                                                  num_steps=m, seed=5, rng="philox", nthreads=8)
    oi, ox, od = oracle.spg_build(o_nsize, o_remap)
    enc_full = sets.enc_int16().cpu().numpy()
    for j, r in enumerate(sub.tolist()):
        lo, hi = int(z.indptr[r]), int(z.indptr[r + 1])
        assert np.array_equal(z.indices[lo:hi].cpu().numpy(), ox[oi[j]:oi[j + 1]])
        assert np.array_equal(enc_full[z.data[lo:hi].cpu().numpy() - 1], o_enc[od[oi[j]:oi[j + 1]] - 1])
    edge = query_pairs(csr, 65536, seed=100, pos_frac=pos_frac)
    wxz, wind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    lens = z.indptr[1:] - z.indptr[:-1]
    assert torch.equal(wind[1:] - wind[:-1], torch.cat([lens[edge[0]], lens[edge[1]]])) and wxz.shape[0] == int(wind[-1])
    enc0 = torch.cat([torch.zeros((1, m + 1), dtype=torch.int16, device="cuda"), sets.enc_int16()])
    zk = z.keyed(enc0, M)
    kxz, kind = sp.gather(edge, zk, "cuda", ptr=True, encode=zk.slot_table())
    assert torch.equal(kind, wind) and torch.equal(kxz, wxz)
    bufs = sp.StepBuffers(csr, 65536, num_walks=M, num_steps=m)
    xz, ind, bsets = sp.sample_and_gather(csr, edge, num_walks=M, num_steps=m, seed=5, rng="philox", buffers=bufs)
    bsets.prefetch().resolve()
    rows = int(bsets.extra[0])
    assert rows == wxz.shape[0] and torch.equal(ind, wind) and torch.equal(xz[:rows], wxz)
    # ... and as 64 reference-sized batches of 1,024 pairs in one launch sequence (round 4)
    many = sp.gather_many(edge.view(2, 64, 1024).permute(1, 0, 2).contiguous(), zk, "cuda", encode=zk.slot_table())
    e3 = edge.view(2, 64, 1024)
    for b in (0, 17, 63):
        gx, gi = sp.gather(e3[:, b], zk, "cuda", ptr=True, encode=zk.slot_table())
        assert torch.equal(many[b][0], gx) and torch.equal(many[b][1], gi)


def test_twitter_scale_int64_offsets(sp):
    """41.65 M nodes, ~2.9e9 adjacency entries (int64 CSR offsets), the graph as dataloader.py:122-135 would hand it over (G + G.T:
    undirected, simple, rows sorted -- built by graphs.symmetric_powerlaw_graph_big, round 4); 10 M roots -> more than 2^31 set
    members, so the SpG row offsets leave the int32 range as well."""
    from surel_plus_amd.graphs import preset_graph
    csr = preset_graph("twitter")
    assert csr.indptr.dtype == torch.int64 and csr.nnz > 2**31
    # the loader's guarantees, checked on a slice of rows (a global check would need a second copy of the 12 GB adjacency)
    ip = csr.indptr
    for r0 in (0, 20_000_000, csr.num_nodes - 200_000):
        lo, hi = int(ip[r0]), int(ip[r0 + 200_000])
        seg = csr.indices[lo:hi].long()
        row = torch.repeat_interleave(torch.arange(r0, r0 + 200_000, device="cuda"), ip[r0 + 1:r0 + 200_001] - ip[r0:r0 + 200_000])
        key = row * csr.num_nodes + seg
        assert bool((key[1:] > key[:-1]).all()) and bool((row != seg).all())            # sorted, simple, no self loops
        # symmetric: every (u, v) of the slice has its (v, u) -- looked up by binary search in v's sorted row
        pick = torch.randint(0, seg.numel(), (20000,), device="cuda")
        u, v = row[pick], seg[pick]
        vb, ve = ip[v], ip[v + 1]
        steps = int(torch.log2((ve - vb).max().float()).ceil().item()) + 1
        lo_, hi_ = vb.clone(), ve.clone()
        for _ in range(steps):
            mid = (lo_ + hi_) // 2
            act = lo_ < hi_
            go = csr.indices[mid.clamp(max=csr.nnz - 1)].long() < u
            lo_, hi_ = torch.where(go & act, mid + 1, lo_), torch.where(~go & act, mid, hi_)
        assert bool(((lo_ < ve) & (csr.indices[lo_.clamp(max=csr.nnz - 1)].long() == u)).all())
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


def test_cit2_scale_four_hop_batch_with_64_bit_key_rows(sp):
    """The paper's sampler setting (Fig. 6a: citation2, m = 4, M = 200) at the bench's batch size: 65,536 pairs on demand through the
    step buffers -- rows of 64-bit LP keys (subgacc_walk_keyrows64 / subgacc_sjoin_fill_keyrows64) -- against the table form of the
    same batch, bit for bit; a random subset of the endpoints against the oracle; the invariants of subg_acc/test/test.py:34-45."""
    from surel_plus_amd.graphs import preset_graph, query_pairs
    csr = preset_graph("cit2")
    M, m, B = 200, 4, 65536
    edge = query_pairs(csr, B, seed=321)
    bufs = sp.StepBuffers(csr, B, num_walks=M, num_steps=m)
    assert bufs.keyrows and bufs.key64
    xz, ind, sets = sp.sample_and_gather(csr, edge, num_walks=M, num_steps=m, seed=5, rng="philox", buffers=bufs)
    sets.prefetch().resolve()
    rows = int(sets.extra[0])
    txz, tind, tsets = sp.sample_and_gather(csr, edge, num_walks=M, num_steps=m, seed=5, rng="philox", key_rows=False)
    assert not tsets.keyrows and rows == txz.shape[0] and torch.equal(ind, tind) and torch.equal(xz[:rows], txz)
    # own-slot feature rows of every segment: the root flag once, every landing-count column sums to 1 (= M / M)
    seg_sum = torch.segment_reduce(xz[:rows, 0, :].double(), "sum", offsets=ind, axis=0)
    assert torch.allclose(seg_sum, torch.ones_like(seg_sum), atol=1e-6)
    # 64 endpoints against the oracle (Philox sets are functions of (seed, root)): set sizes and sorted members
    g = torch.Generator(device="cuda").manual_seed(2)
    pick = torch.randint(0, 2 * B, (64,), device="cuda", generator=g)
    roots = edge.reshape(-1)[pick].cpu().numpy()
    o_nsize, o_remap, o_enc = oracle.gset_sampler(csr.indptr.cpu().numpy(), csr.indices.cpu().numpy(), roots, num_walks=M, num_steps=m,
                                                  seed=5, rng="philox", nthreads=8)
    oi, ox, od = oracle.spg_build(o_nsize, o_remap)
    stride = bufs.stride                   # (rows lie on whole 128-byte lines: M*m+1 rounded up, StepBuffers)
    for j, i in enumerate(pick.tolist()):
        n_i = int(bufs.nsize[i])
        assert n_i == int(o_nsize[j])
        assert np.array_equal(bufs.ids[i * stride: i * stride + n_i].cpu().numpy(), ox[oi[j]:oi[j + 1]])
