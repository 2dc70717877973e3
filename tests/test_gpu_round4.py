"""GPU (MI355X): round-4 additions, through the C ABI, against the oracle / the per-batch forms.
  * gather_many / sample_and_gather_many / StepBuffers(batch=) / CapturedStep(batch=): nb reference-sized batches in one launch
    sequence == nb calls of the reference-shaped entry points (train.py:120-127 at main.py:32's batch size), and == the oracle;
  * the sorted work list is only taken where walk_rows_kernel runs (bucket > 0 with >= 16,384 roots);
  * rand_r dead ends are remembered on the DeviceCSR; stale member counts are refused."""
import os

import numpy as np
import pytest
import torch

import oracle
from gpu_helpers import _oracle_spg, dir_graph, sp, sym_graph  # noqa: F401

pytestmark = pytest.mark.gpu


def _store(sp, N=3000, E=12000, M=40, k=4, seed=5):
    ptr_, idx = sym_graph(N, E, seed=seed, hubs=1)
    csr = sp.DeviceCSR(ptr_, idx)
    z, enc = sp.subg_matrix(csr, np.arange(N), num_walks=M, num_steps=k, rng="philox", seed=9)
    o_nsize, o_remap, o_enc = oracle.gset_sampler(ptr_, idx, np.arange(N), num_walks=M, num_steps=k - 1, rng="philox", seed=9, nthreads=8)
    o_spg = oracle.spg_build(o_nsize, o_remap)
    return csr, z, enc, o_spg, (ptr_, idx)


@pytest.mark.parametrize("ptr", [True, False])
@pytest.mark.parametrize("store", ["table", "keyed", "float"])
def test_gather_many_is_gather_batch_by_batch(sp, ptr, store):
    N, M = 3000, 40
    csr, z, enc, o_spg, _ = _store(sp, N=N, M=M)
    rs = np.random.default_rng(3)
    nb, B = 7, 53
    edges = rs.integers(0, N, (nb, 2, B))
    edges[2, :, 5] = edges[2, 0, 5]            # a (u, u) pair
    edges[4, 1, :10] = edges[4, 0, :10][::-1]  # repeated endpoints
    if store == "float":
        import scipy.sparse as sps
        A = sps.random(N, N, density=0.004, format="csr", random_state=4, dtype=np.float64)
        A.sort_indices()
        x, table, otab = sp.SpG.from_scipy(A), None, None
        ospg = (A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data)
    else:
        zsf = enc.astype(np.float32) / np.float32(M)
        ospg, otab = o_spg, zsf
        if store == "keyed":
            x = z.keyed(enc, M)
            table = x.slot_table()
        else:
            x, table = z, torch.from_numpy(zsf).cuda()
    many = sp.gather_many(edges, x, "cuda", ptr=ptr, encode=table)
    assert len(many) == nb
    for b in range(nb):
        xz1, ind1 = sp.gather(edges[b], x, "cuda", ptr=ptr, encode=table)
        assert torch.equal(many[b][0], xz1) and torch.equal(many[b][1], ind1), b
        oxz, oind = oracle.gather(edges[b], ospg, ptr=ptr, encode=otab)
        assert np.array_equal(many[b][0].cpu().numpy(), oxz) and np.array_equal(many[b][1].cpu().numpy(), oind), b
    # a list with a short last batch: runs of equal size are fused, the rest joined singly
    lst = [edges[0], edges[1], edges[2][:, :17]]
    got = sp.gather_many(lst, x, "cuda", ptr=ptr, encode=table)
    for e, (xz_b, ind_b) in zip(lst, got):
        xz1, ind1 = sp.gather(e, x, "cuda", ptr=ptr, encode=table)
        assert torch.equal(xz_b, xz1) and torch.equal(ind_b, ind1)
    with pytest.raises(IndexError):
        bad = edges.copy()
        bad[3, 1, 2] = N + 5
        sp.gather_many(bad, x, "cuda", ptr=ptr, encode=table)
    # edge cases: no batches, empty batches, a single batch
    assert sp.gather_many(edges[:0], x, "cuda", ptr=ptr, encode=table) == []
    empty = sp.gather_many(edges[:2, :, :0], x, "cuda", ptr=ptr, encode=table)
    assert len(empty) == 2 and all(e_[0].shape[0] == 0 for e_ in empty)
    one = sp.gather_many(edges[:1], x, "cuda", ptr=ptr, encode=table)
    xz1, ind1 = sp.gather(edges[0], x, "cuda", ptr=ptr, encode=table)
    assert len(one) == 1 and torch.equal(one[0][0], xz1) and torch.equal(one[0][1], ind1)


@pytest.mark.parametrize("store", ["table", "keyed"])
def test_hgather_many_is_hgather_batch_by_batch(sp, store):
    """train.py:48-72 at main_horder.py:33's batch size, many batches per launch sequence: per batch the blocks [U|w ; W|u ; V|w ; W|v]
    with their segment ids, bit for bit the single-batch call and the oracle"""
    N, M = 3000, 40
    csr, z, enc, o_spg, _ = _store(sp, N=N, M=M)
    zsf = enc.astype(np.float32) / np.float32(M)
    x, table = (z.keyed(enc, M), None) if store == "keyed" else (z, torch.from_numpy(zsf).cuda())
    if store == "keyed":
        table = x.slot_table()
    hedges = np.random.default_rng(6).integers(0, N, (5, 3, 41))
    hedges[1, :, 3] = hedges[1, 0, 3]
    many = sp.hgather_many(hedges, x, "cuda", encode=table)
    assert len(many) == 5
    for b in range(5):
        xz1, ids1 = sp.hgather(hedges[b], x, "cuda", encode=table)
        assert torch.equal(many[b][0], xz1) and torch.equal(many[b][1], ids1), b
        oxz, oids = oracle.hgather(hedges[b], o_spg, zsf)
        assert np.array_equal(many[b][0].cpu().numpy(), oxz) and np.array_equal(many[b][1].cpu().numpy(), oids), b
    with pytest.raises(NotImplementedError):
        sp.hgather_many(hedges, x, "cuda", encode=None)


@pytest.mark.parametrize("M,hops,idx64", [(40, 3, False), (200, 2, True), (24, 4, False)])
def test_sample_and_gather_many_is_sample_and_gather_batch_by_batch(sp, M, hops, idx64):
    """nb batches of B pairs sampled by ONE walk launch and joined by ONE join launch: every batch's (xz, indptr) is what the
    single-batch call gives (Philox: a root's set is a function of (seed, root id)) and what the oracle's resident store gives."""
    N = 2500
    ptr_, idx = sym_graph(N, 10000, seed=2, hubs=1)
    csr = sp.DeviceCSR(ptr_.astype(np.int64) if idx64 else ptr_, idx)
    o_nsize, o_remap, o_enc = oracle.gset_sampler(ptr_, idx, np.arange(N), num_walks=M, num_steps=hops, rng="philox", seed=21, nthreads=8)
    o_spg = oracle.spg_build(o_nsize, o_remap)
    zsf = oracle.enc_table(o_enc).astype(np.float32) / np.float32(M)
    nb, B = 6, 64
    edges = torch.from_numpy(np.random.default_rng(8).integers(0, N, (nb, 2, B))).cuda()
    forms = {"allocating": {}}
    try:
        forms["buffers"] = {"buffers": sp.StepBuffers(csr, nb * B, num_walks=M, num_steps=hops, batch=B)}
    except ValueError:
        pass
    for name, kw in forms.items():
        parts, sets = sp.sample_and_gather_many(csr, edges, num_walks=M, num_steps=hops, seed=21, rng="philox", **kw)
        assert len(parts) == nb
        for b in range(nb):
            xz1, ind1, _ = sp.sample_and_gather(csr, edges[b], num_walks=M, num_steps=hops, seed=21, rng="philox")
            assert torch.equal(parts[b][0], xz1) and torch.equal(parts[b][1], ind1), (name, b)
            oxz, oind = oracle.gather(edges[b].cpu().numpy(), o_spg, ptr=True, encode=zsf)
            assert np.array_equal(parts[b][0].cpu().numpy(), oxz) and np.array_equal(parts[b][1].cpu().numpy(), oind), (name, b)
    if "buffers" in forms:      # ... and as ONE HIP graph replayed with new batches
        cap = sp.CapturedStep(csr, nb * B, num_walks=M, num_steps=hops, seed=21, rng="philox", batch=B)
        for rep in range(2):
            e2 = torch.from_numpy(np.random.default_rng(80 + rep).integers(0, N, (nb, 2, B))).cuda()
            got = cap(e2).finish_batches()
            for b in range(nb):
                oxz, oind = oracle.gather(e2[b].cpu().numpy(), o_spg, ptr=True, encode=zsf)
                assert np.array_equal(got[b][0].cpu().numpy(), oxz) and np.array_equal(got[b][1].cpu().numpy(), oind), (rep, b)
        with pytest.raises(ValueError):
            cap(edges[:, :, :5])
    with pytest.raises(ValueError):
        sp.sample_and_gather_many(csr, edges, num_walks=M, num_steps=hops, rng="rand_r")
    with pytest.raises(ValueError):
        sp.StepBuffers(csr, 100, num_walks=M, num_steps=hops, batch=33)


@pytest.mark.parametrize("rng", ["philox", "rand_r"])
def test_a_truncating_bucket_with_many_roots_takes_the_general_fused_kernel(sp, rng):
    """with bucket > 0 walk_rows_kernel declines the launch; a single chunk of >= 16,384 roots used to be
    handed to it with a work list all the same and raised BADARG.  The gate now asks the same question the launcher does."""
    N, M, m, bucket = 20000, 200, 2, 50
    ptr_, idx = sym_graph(N, 60000, seed=6, hubs=2)
    csr = sp.DeviceCSR(ptr_, idx)
    q = np.random.default_rng(1).permutation(N)[:17000]
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, m, 4, rng, bucket)
    for kw in ({"fused": True}, {"fused": True, "strided": True}):
        z, sets = sp.sample_spg(csr, q, num_walks=M, num_steps=m, seed=4, rng=rng, bucket=bucket, **kw)
        if isinstance(z, sp.StridedSpG):
            z = z.to_csr()
        nnz = z.nnz
        assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices[:nnz].cpu().numpy(), ox), kw
        assert np.array_equal(z.data[:nnz].cpu().numpy(), od) and np.array_equal(sets.enc_int16().cpu().numpy(), oenc), kw
    from surel_plus_amd.sampler import rows_kernel_takes, walk_kernel_name
    assert not rows_kernel_takes(M, m, bucket) and rows_kernel_takes(M, m, -1)
    assert walk_kernel_name(csr, M, m, True, bucket) == "walk_sets_kernel<SPG>"
    # the same roots without the bucket go through the sorted work list and walk_rows_kernel, with and without the order
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, m, 4, rng, -1)
    for sort_roots in (True, False):
        z, sets = sp.sample_spg(csr, q, num_walks=M, num_steps=m, seed=4, rng=rng, fused=True, sort_roots=sort_roots)
        assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices[: z.nnz].cpu().numpy(), ox), sort_roots
        assert np.array_equal(z.data[: z.nnz].cpu().numpy(), od), sort_roots


def test_dead_ends_are_remembered_on_the_graph(sp):
    """rand_r on a directed graph used to walk EVERY batch twice (once to find the dead end, once replayed);
    the discovery now sticks to the DeviceCSR, and the buffered step -- which cannot replay -- refuses such a graph by name."""
    ptr_, idx = dir_graph(600, 1500, seed=3, hubs=1)
    csr = sp.DeviceCSR(ptr_, idx)
    q = np.arange(600)
    ref = oracle.gset_sampler(ptr_, idx, q, num_walks=20, num_steps=3, rng="rand_r")
    assert not getattr(csr, "_rand_r_dead_ends", False)
    from surel_plus_amd import sampler
    calls = []
    real = sampler.lib().subgacc_rng_replay
    for rep in range(2):
        s = sp.sample_sets(csr, q, num_walks=20, num_steps=3, rng="rand_r")
        assert np.array_equal(s.nsize.cpu().numpy(), ref[0]) and np.array_equal(s.ids.cpu().numpy(), ref[1][0])
        calls.append(getattr(csr, "_rand_r_dead_ends", False))
    assert calls == [True, True] and real is not None
    with pytest.raises(ValueError, match="dead ends"):
        sp.StepBuffers(csr, 64, num_walks=20, num_steps=3, rng="rand_r")
    # Philox is untouched by the note
    s = sp.sample_sets(csr, q, num_walks=20, num_steps=3, rng="philox")
    o = oracle.gset_sampler(ptr_, idx, q, num_walks=20, num_steps=3, rng="philox")
    assert np.array_equal(s.nsize.cpu().numpy(), o[0])


def test_member_count_of_a_deduplicated_step_is_refused_once_the_buffers_moved_on(sp):
    """SampledSets.X of a root-dedup step is counted lazily from the buffers' sizes; after the buffers took a
    later batch that count would silently describe the later batch."""
    N, M, hops, B = 4000, 200, 3, 256
    ptr_, idx = sym_graph(N, 16000, seed=12)
    csr = sp.DeviceCSR(ptr_, idx)
    bufs = sp.StepBuffers(csr, B, num_walks=M, num_steps=hops, dedup_roots=True)
    rs = np.random.default_rng(5)
    e1 = torch.from_numpy(rs.integers(0, 50, (2, B))).cuda()       # many repeated endpoints
    e2 = torch.from_numpy(rs.integers(0, N, (2, B))).cuda()
    _, _, s1 = sp.sample_and_gather(csr, e1, num_walks=M, num_steps=hops, rng="philox", buffers=bufs, dedup_roots=True)
    s1.resolve()
    x1 = s1.X                                                      # asked for in time: counted from this batch's sizes
    o = oracle.gset_sampler(ptr_, idx, np.unique(e1.cpu().numpy()), num_walks=M, num_steps=hops, rng="philox", nthreads=8)
    assert x1 == int(o[0].sum())
    _, _, s2 = sp.sample_and_gather(csr, e1, num_walks=M, num_steps=hops, rng="philox", buffers=bufs, dedup_roots=True)
    s2.resolve()
    sp.sample_and_gather(csr, e2, num_walks=M, num_steps=hops, rng="philox", buffers=bufs, dedup_roots=True)[2].resolve()
    with pytest.raises(sp.SubgAccError, match="later batch"):
        s2.X


def test_twitter_like_generator_is_symmetric_simple_and_sorted(sp):
    """graphs.symmetric_powerlaw_graph_big: G + G.T as dataloader.py:122-135 would hand it over, built chunk by chunk."""
    from surel_plus_amd.graphs import symmetric_powerlaw_graph_big
    g = symmetric_powerlaw_graph_big(60000, 30.0, seed=3, row_chunks=5)
    ip, ix = g.indptr.long(), g.indices.long()
    N = g.num_nodes
    row = torch.repeat_interleave(torch.arange(N, device=ip.device), ip[1:] - ip[:-1])
    key = row * N + ix
    assert bool((key[1:] > key[:-1]).all()) and bool((row != ix).all())
    assert torch.equal(torch.sort(ix * N + row).values, key)
    assert 25.0 < g.nnz / N < 31.0


def test_the_epoch_loop_of_integration_md(sp):
    """INTEGRATION.md, "the reference's loop at the reference's batch size": train.py:120-128 regrouped -- the DataLoader's permutation
    drawn up front, 64 batches per gather_many call, a short last batch -- gives every batch exactly what the reference's loop body
    (`pgather(edge, g, device, rpe, bgather, ptr)` per batch, train.py:127) gives, and a first model stage consumes it unchanged."""
    from torch.utils.data import DataLoader
    N, M, batch_size = 3000, 40, 128
    csr, g, enc, o_spg, _ = _store(sp, N=N, M=M)
    rpe = torch.from_numpy(enc).cuda().float() / M                     # main.py:174
    edges = torch.from_numpy(np.random.default_rng(12).integers(0, N, (2, 1000)))      # 7 full batches + one of 104
    torch.manual_seed(3)
    perms = list(DataLoader(range(edges.size(1)), batch_size, shuffle=True))
    embed = torch.nn.Linear(enc.shape[1], 8).cuda()
    seen = 0
    for lo in range(0, len(perms), 64):
        group = [edges[:, p] for p in perms[lo:lo + 64]]
        for perm, (x, ind) in zip(perms[lo:lo + 64], sp.gather_many(group, g, "cuda", ptr=True, encode=rpe)):
            rx, rind = sp.pgather(edges[:, perm], g, "cuda", rpe, sp.bgather, ptr=True)          # the reference's loop body
            assert torch.equal(x, rx) and torch.equal(ind, rind)
            oxz, oind = oracle.gather(edges[:, perm].numpy(), o_spg, ptr=True, encode=rpe.cpu().numpy())
            assert np.array_equal(x.cpu().numpy(), oxz) and np.array_equal(ind.cpu().numpy(), oind)
            h = torch.segment_reduce(embed(x).sum(dim=-2), "mean", offsets=ind, axis=0).view(2, -1, 8)     # model.py:78-83, mean aggregation
            assert h.shape[1] == perm.numel() and bool(torch.isfinite(h).all())
            seen += perm.numel()
    assert seen == edges.size(1)


def test_gather_many_lazy_reads_nothing_until_a_batch_is_taken(sp):
    """gather_many(out=, lazy=True): the call queues the join without a host read; the batch boundaries (and the status word) arrive
    with the first batch that is taken -- same results, and a row number outside the store raises THERE"""
    N, M = 3000, 40
    csr, z, enc, o_spg, _ = _store(sp, N=N, M=M)
    zk = z.keyed(enc, M)
    nb, B = 6, 64
    edges = torch.from_numpy(np.random.default_rng(2).integers(0, N, (nb, 2, B))).cuda()
    out = torch.empty(nb * 2 * B * zk.max_len * 2 * enc.shape[1], dtype=torch.float32, device="cuda")
    lazy = sp.gather_many(edges, zk, "cuda", encode=zk.slot_table(), out=out, lazy=True)
    assert lazy._bounds is None                                   # nothing has been read back yet
    eager = sp.gather_many(edges, zk, "cuda", encode=zk.slot_table())
    for (a, ai), (b_, bi) in zip(lazy, eager):
        assert torch.equal(a, b_) and torch.equal(ai, bi)
    assert len(lazy[1:3]) == 2 and torch.equal(lazy[-1][0], eager[nb - 1][0])
    bad = edges.clone()
    bad[2, 0, 5] = N + 3
    if os.environ.get("SUBGACC_DEBUG", "0") == "1":                 # (debug mode reads the status word after every join: it raises at the call)
        with pytest.raises(IndexError):
            sp.gather_many(bad, zk, "cuda", encode=zk.slot_table(), out=out, lazy=True)
    else:
        q = sp.gather_many(bad, zk, "cuda", encode=zk.slot_table(), out=out, lazy=True)
        with pytest.raises(IndexError):
            q[0]
    with pytest.raises(ValueError):
        sp.gather_many(edges, zk, "cuda", ptr=False, encode=zk.slot_table(), out=out, lazy=True)
    with pytest.raises(ValueError):
        sp.gather_many(edges, zk, "cuda", encode=zk.slot_table(), out=out[:1000], lazy=True)
