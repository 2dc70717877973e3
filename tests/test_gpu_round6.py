"""GPU (MI355X): round-6 additions, through the C ABI, against the oracle.
  * the key rows' distribution sort (csrc/walk_rows.hip) on sets whose ids are so unevenly spread that its first level (equal id
    width) and its second (equal width inside a crowded bucket) both leave crowded buckets: dense islands of consecutive ids in a
    wide id range.  VERDICT r5 #2: the walk kernel was SLOWER on a graph with id locality than on a structureless one."""
import numpy as np
import pytest
import torch

import oracle
from gpu_helpers import _oracle_spg, sp  # noqa: F401

pytestmark = pytest.mark.gpu


def island_graph(N, islands, width, deg, seed, p_out=0.1, spread=None):
    """`islands` blocks of `width` CONSECUTIVE ids, far apart in [0, N): every island node has ~deg neighbours, 1 - p_out of them inside
    its island; every other id is an isolated node.  Symmetric, simple, rows sorted (dataloader.py:122-135 hands such a graph over)."""
    import scipy.sparse as sps
    rng = np.random.default_rng(seed)
    spread = spread or N // islands
    base = (np.arange(islands, dtype=np.int64) * spread + rng.integers(0, max(spread - width, 1), islands))
    nodes = (base[:, None] + np.arange(width)[None, :]).reshape(-1)
    E = nodes.size * deg // 2
    src_i = rng.integers(0, nodes.size, E)
    inside = rng.random(E) >= p_out
    dst_i = np.where(inside, (src_i // width) * width + rng.integers(0, width, E), rng.integers(0, nodes.size, E))
    r, c = nodes[src_i], nodes[dst_i]
    A = sps.csr_matrix((np.ones(E, dtype=np.int8), (r, c)), shape=(N, N))
    A = sps.csr_matrix(A + A.T)
    A.setdiag(0)
    A.eliminate_zeros()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), nodes


@pytest.mark.parametrize("M,hops,width", [(200, 3, 300), (200, 2, 300), (200, 4, 500), (100, 4, 300), (200, 3, 24)])
@pytest.mark.parametrize("i64", [False, True])
def test_sort_levels_on_dense_islands_in_a_wide_id_range(sp, M, hops, width, i64):
    """6 M ids, islands of `width` consecutive ids: a set has most of its members inside ONE island (one level-1 bucket of the sort: 8,192
    ids wide for 1,024 buckets) next to a few dozen members anywhere; level 2 cuts that bucket into sub-buckets ~30 ids wide that
    still hold ~30 members (width 300 / 500) -- level 3 --, or every island into one sub-bucket (width 24).  Every kernel shape of
    walk_rows_kernel's key rows (1,024 / 512 slots, one and two waves, 32- and 64-bit keys, int32 and int64 row offsets), the table
    form and the general kernel, all against the oracle."""
    from surel_plus_amd.spg import sample_spg
    N = 6_000_000
    ptr_, idx, nodes = island_graph(N, 40, width, 24, seed=width + hops)
    q = np.random.default_rng(2).choice(nodes, 1500)
    q[:4] = [0, N - 1, nodes[0], nodes[-1]]                          # (isolated roots at both ends of the id range too)
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, hops, 21, "philox", -1)
    sizes = np.diff(oi)
    if width >= 300:    # the shape this test is about: a big set, most of it in a window far narrower than range / 1,024
        big = int(np.argmax(sizes))
        row = ox[oi[big]:oi[big + 1]]
        assert sizes[big] > 150 and int(row.max()) - int(row.min()) > (1 << 22)
        assert int(np.max(np.bincount((row - row.min()) >> 13))) > 100
    csr = sp.DeviceCSR(ptr_.astype(np.int64) if i64 else ptr_, idx)
    for kw in ({"strided": True, "number_rows": False}, {"strided": True}, {"fused": True}, {"fused": False}, {"strided": True, "fused": False}):
        z, sets = sample_spg(csr, q, num_walks=M, num_steps=hops, seed=21, rng="philox", **kw)
        if isinstance(z, sp.StridedSpG):
            if kw.get("number_rows", True) is False:
                assert sets.keyrows                                  # really the key-row epilogue
            z = z.to_csr()
        assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices[: z.nnz].cpu().numpy(), ox), kw
        assert np.array_equal(z.data[: z.nnz].cpu().numpy(), od) and np.array_equal(sets.enc_int16().cpu().numpy(), oenc), kw


def test_buffered_step_on_dense_islands_joins_like_the_oracle(sp):
    """the on-demand step (StepBuffers: key rows on whole lines, sorted work list) over the island graph: (xz, indptr) == the oracle's
    join of the oracle's sets -- a row that left the sort out of order would miss its partners in the join's search"""
    from surel_plus_amd.graphs import query_pairs
    N = 6_000_000
    ptr_, idx, nodes = island_graph(N, 40, 300, 24, seed=5)
    csr = sp.DeviceCSR(ptr_, idx)
    M, hops, B = 200, 3, 9000
    rng = np.random.default_rng(3)
    e = np.stack([rng.choice(nodes, B), rng.choice(nodes, B)]).astype(np.int64)
    e[1, : B // 2] = e[0, : B // 2] + rng.integers(-20, 20, B // 2)          # half the pairs inside one island (or next to it)
    e = np.clip(e, 0, N - 1)
    edge = torch.from_numpy(e).cuda()
    bufs = sp.StepBuffers(csr, B, num_walks=M, num_steps=hops, rng="philox")
    xz, ind, sets = sp.sample_and_gather(csr, edge, num_walks=M, num_steps=hops, seed=9, rng="philox", buffers=bufs)
    sets.resolve()                                                  # (the buffered step is lazy: xz is a view of the worst-case buffer)
    xz = xz[: int(ind[-1])]
    o_nsize, o_remap, o_enc = oracle.gset_sampler(ptr_, idx, e.reshape(-1), num_walks=M, num_steps=hops, rng="philox", seed=9, nthreads=8)
    o_spg = oracle.spg_build(o_nsize, o_remap)
    zsf = oracle.enc_table(o_enc).astype(np.float32) / np.float32(M)
    B2 = 2 * B
    pairs = np.stack([np.arange(B), np.arange(B, B2)])
    oxz, oind = oracle.gather(pairs, o_spg, ptr=True, encode=zsf)
    assert np.array_equal(ind.cpu().numpy(), oind) and np.array_equal(xz.cpu().numpy(), oxz)


def test_status_of_the_first_tile_survives_a_thousand_one_call_joins(sp):
    """VERDICT r5 #8b: the one-pass size scan leaves its state clean by itself, which needs an ORDER between words at different
    addresses -- every tile's status OR and prefix store performed before the finishing tile's exchange and zeroing stores (round 6:
    every wave drains its memory operations before the barrier in front of the `done` increment, which is a release / acquire at agent
    scope).  1,000 calls over 64 tiles with a bad row in tile 0, the tile that starts first and whose OR is the oldest in flight:
    bit 16 reaches host_tail every time, and the state is all zero after every call -- a flag that had lost the race would be missing
    from this call and found in the next."""
    from surel_plus_amd.graphs import ppr_like_spg
    N, B = 5000, 32768
    zf = ppr_like_spg(N, 40, seed=5)
    cj = sp.CapturedJoin(zf, B)
    rs = np.random.default_rng(3)
    good = torch.from_numpy(rs.integers(0, N, (2, B))).cuda()
    bad = good.clone()
    bad[0, 7] = N + 3                                       # segment 7: tile 0
    words = 8 + -(-2 * B // 1024)
    for it in range(1000):
        with pytest.raises(IndexError):
            cj(bad).finish()
        assert int(cj._host[1]) & 16
        if it % 50 == 0:        # (a read-back of the state is a synchronisation of its own: now and then, and after the clean call below)
            assert int(cj._state.view(torch.int64)[:words].abs().sum().item()) == 0, it
    xz, ind = cj(good).finish()                             # nothing leaked into the call behind
    assert int(cj._host[1]) == 0 and int(cj._state.view(torch.int64)[:words].abs().sum().item()) == 0
    wxz, wind = sp.gather(good, zf, "cuda", ptr=True, encode=None)
    assert torch.equal(ind, wind) and torch.equal(xz, wxz)


def test_four_threads_of_lazy_steps_on_four_streams(sp):
    """"safe to call from the 4 pgather-style Python threads" (train.py:88-111).  Four threads, each with its own HIP
    stream and its own pair of step buffers, queue 2,000 lazy on-demand steps each -- sizes and status published to pinned host
    memory by a kernel behind every step (_lib.publish / keep_until), resolved one step behind -- and check every step's sizes
    against the single-threaded result of the same batch."""
    import threading

    from surel_plus_amd.graphs import query_pairs
    from gpu_helpers import sym_graph
    ptr_, idx = sym_graph(20000, 120000, seed=13, hubs=3)
    csr = sp.DeviceCSR(ptr_, idx)
    M, hops, B, STEPS = 64, 2, 256, 2000
    edges = [query_pairs(csr, B, seed=70 + s) for s in range(8)]
    want = []
    for e in edges:
        xz, ind, sets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=4, rng="philox")
        want.append((int(xz.shape[0]), int(sets.X)))
    errors = []

    def worker(t):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                bufs = [sp.StepBuffers(csr, B, num_walks=M, num_steps=hops, rng="philox") for _ in range(2)]
                pending = None
                for s in range(STEPS):
                    k = (s + 3 * t) % len(edges)
                    xz, ind, sets = sp.sample_and_gather(csr, edges[k], num_walks=M, num_steps=hops, seed=4, rng="philox", lazy=True,
                                                         buffers=bufs[s & 1])
                    sets.prefetch(extra=ind[-1:])
                    if pending is not None:
                        pk, psets = pending
                        psets.resolve()
                        if (int(psets.extra[0]), int(psets.X)) != want[pk]:
                            errors.append((t, s - 1, int(psets.extra[0]), int(psets.X), want[pk]))
                            return
                    pending = (k, sets)
                pending[1].resolve()
                stream.synchronize()
        except Exception as ex:      # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(ex)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


# ------------------------------------------------------------------ resident stores on whole lines (SpG.aligned(): headed rows, ABI 7)
@pytest.mark.parametrize("name", ["sjoin_int.npz", "sjoin_float.npz", "sjoin_int_emptyrows.npz"])
def test_aligned_store_matches_reference_golden(sp, name):
    """the reference's own outputs (train.gather run in the build container, tests/golden/) through the store laid out again on whole
    128-byte lines -- SpG.aligned(): rows at a fixed pitch, their lengths in their first slots, no row pointers -- eagerly (size
    pass alone, then the fill), lazily (one call), as CapturedJoin / CapturedJoinPool, with segment pointers and segment ids"""
    from gpu_helpers import _load, _spg_from_golden
    g = _load(name)
    z = _spg_from_golden(sp, g)
    za = z.aligned()
    assert za.pitch % 32 == 0 and za.pitch > z.max_len and torch.equal(za.row_lengths().long(), z.indptr[1:] - z.indptr[:-1])
    back = za.to_spg()
    nnz = z.nnz
    assert torch.equal(back.indptr, z.indptr) and torch.equal(back.indices, z.indices[:nnz]) and torch.equal(back.data, z.data[:nnz])
    enc = torch.from_numpy(g["encode"]).cuda() if g["encode"].size else None
    edge = np.asarray(g["edge"])
    for e in (edge, torch.from_numpy(edge).cuda()):
        xz, ind = sp.gather(e, za, "cuda", ptr=True, encode=enc)
        assert xz.dtype == torch.float32 and ind.dtype == torch.int64
        assert np.array_equal(xz.cpu().numpy(), g["xz_ptr1"]) and np.array_equal(ind.cpu().numpy(), g["ind_ptr1"])
        xz, ids = sp.gather(e, za, "cuda", ptr=False, encode=enc)
        assert np.array_equal(xz.cpu().numpy(), g["xz_ptr0"]) and np.array_equal(ids.cpu().numpy(), g["ind_ptr0"])
    k = 1 if enc is None else enc.shape[1]
    buf = torch.empty(2 * edge.shape[1] * za.max_len * 2 * k, dtype=torch.float32, device="cuda")
    xz, ind = sp.gather(edge, za, "cuda", ptr=True, encode=enc, out=buf, lazy=True)
    R = int(ind[-1])
    assert np.array_equal(xz[:R].cpu().numpy(), g["xz_ptr1"]) and np.array_equal(ind.cpu().numpy(), g["ind_ptr1"])
    cj = sp.CapturedJoin(za, edge.shape[1], encode=enc)
    xz, ind = cj(torch.from_numpy(edge).cuda()).finish()
    assert np.array_equal(xz.cpu().numpy(), g["xz_ptr1"]) and np.array_equal(ind.cpu().numpy(), g["ind_ptr1"])
    pool = sp.CapturedJoinPool(za, edge.shape[1], lanes=2, encode=enc)
    xz, ind = pool.finish(pool.submit(torch.from_numpy(edge).cuda()))
    assert np.array_equal(xz.cpu().numpy(), g["xz_ptr1"]) and np.array_equal(ind.cpu().numpy(), g["ind_ptr1"])


def test_aligned_hgather_matches_reference_golden(sp):
    """train.hgather's own output (tests/golden/hjoin_int.npz) from the aligned store, eagerly and as CapturedJoin(triplets=True)"""
    from gpu_helpers import _load, _spg_from_golden
    g = _load("hjoin_int.npz")
    za = _spg_from_golden(sp, g).aligned()
    hedge = np.asarray(g["hedge"])
    enc = torch.from_numpy(g["encode"]).cuda()
    xz, ids = sp.hgather(hedge, za, "cuda", encode=enc)
    assert np.array_equal(xz.cpu().numpy(), g["xz"]) and np.array_equal(ids.cpu().numpy(), g["ind"])
    cj = sp.CapturedJoin(za, hedge.shape[1], encode=enc, triplets=True)
    xz, ids = cj(torch.from_numpy(hedge).cuda()).finish()
    assert np.array_equal(xz.cpu().numpy(), g["xz"]) and np.array_equal(ids.cpu().numpy(), g["ind"])


@pytest.mark.parametrize("payload", ["sfptr", "keyed", "float"])
@pytest.mark.parametrize("max_len", [1, 31, 32, 63, 64, 65, 127, 128, 129, 255, 257, 600])
def test_aligned_store_at_every_span_and_trip_boundary(sp, payload, max_len):
    """rows of every length around the kernels' span (64) and trip (128 / 256 lanes) boundaries and around the pitch's own (a row of
    31 members fills its 32-word slot exactly, one of 32 needs the next line): the three payload forms of a resident store, packed
    against aligned, against the oracle's join of the same store -- (u,u) pairs, empty rows and out-of-range detection included"""
    rs = np.random.default_rng(max_len)
    N, span = 600, 4000
    lens = rs.integers(0, max_len + 1, N)
    lens[:4] = [max_len, max_len, 0, max(max_len - 1, 0)]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = np.concatenate([np.sort(rs.choice(span, n_, replace=False)) for n_ in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    M, m = 200, 2
    if payload == "float":
        data = (rs.random(ids.size) + 0.1) / 1.1
        z = sp.SpG(torch.from_numpy(indptr).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(data).cuda(), max_len=max_len, shape=(N, span))
        enc = o_enc = None
    else:
        c = 37
        enc_i = np.zeros((c + 1, m + 1), dtype=np.int16)
        enc_i[1:, 1:] = rs.integers(0, M + 1, (c, m))
        enc_i[1:, 0] = np.where(rs.random(c) < 0.1, M, 0)
        data = rs.integers(1, c + 1, ids.size).astype(np.int32)
        z = sp.SpG(torch.from_numpy(indptr).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(data).cuda(), max_len=max_len, shape=(N, span))
        o_enc = enc_i.astype(np.float32) / np.float32(M)
        enc = torch.from_numpy(o_enc).cuda()
        if payload == "keyed":
            z = z.keyed(torch.from_numpy(enc_i).cuda(), M)
            enc = z.slot_table()
    za = z.aligned()
    B = 3000
    e = rs.integers(0, N, (2, B))
    e[:, :4] = [[0, 1, 2, 5], [1, 1, 0, 2]]
    oxz, oind = oracle.gather(e, (indptr, ids, data), ptr=True, encode=o_enc)
    for store, encode in ((z, enc), (za, za.slot_table() if payload == "keyed" else enc)):
        xz, ind = sp.gather(e, store, "cuda", ptr=True, encode=encode)
        assert np.array_equal(ind.cpu().numpy(), oind) and np.array_equal(xz.cpu().numpy(), oxz), type(store).__name__
    cj = sp.CapturedJoin(za, B, encode=za.slot_table() if payload == "keyed" else enc)
    xz, ind = cj(torch.from_numpy(e).cuda()).finish()
    assert np.array_equal(ind.cpu().numpy(), oind) and np.array_equal(xz.cpu().numpy(), oxz)
    e[1, 17] = N + 5                                           # a row outside the store: IndexError as scipy raises it (train.py:15)
    with pytest.raises(IndexError):
        sp.gather(e, za, "cuda", ptr=True, encode=za.slot_table() if payload == "keyed" else enc)
    with pytest.raises(IndexError):
        cj(torch.from_numpy(e).cuda()).finish()


def test_rows_without_a_root_stay_empty_when_the_general_kernel_takes_a_work_list_call(sp):
    """subgacc_walk_spg_list hands the launch to the general kernel when the fused-row kernel declines it (here: a
    truncating bucket); that kernel reads no list and walks all n rows -- a row whose root is SUBGACC_NO_ROOT (what the list had left
    out) must stay an EMPTY row there too, not be reported as a root outside the graph."""
    from gpu_helpers import sym_graph
    N, M, m, bucket = 20000, 200, 2, 50
    ptr_, idx = sym_graph(N, 60000, seed=6, hubs=2)
    csr = sp.DeviceCSR(ptr_, idx)
    q = np.random.default_rng(1).permutation(N)[:17000].astype(np.int32)
    holes = np.random.default_rng(2).random(q.size) < 0.1
    qh = q.copy()
    qh[holes] = -2 ** 31
    z, sets = sp.sample_spg(csr, qh, num_walks=M, num_steps=m, seed=4, rng="philox", bucket=bucket, fused=True)
    nsize = sets.nsize.cpu().numpy()
    assert (nsize[holes] == 0).all() and (nsize[~holes] > 0).all()
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q[~holes], M, m, 4, "philox", bucket)
    keep = np.flatnonzero(~holes)
    got_ptr = z.indptr.cpu().numpy()
    assert np.array_equal(np.diff(got_ptr)[keep], np.diff(oi)) and np.array_equal(z.indices[: z.nnz].cpu().numpy(), ox)


@pytest.mark.parametrize("M,hops", [(200, 3), (37, 2), (50, 4), (203, 4)])
def test_rows_written_four_members_per_store_equal_rows_written_one_by_one(sp, M, hops):
    """walk_rows_kernel writes a key row four members per 16-byte store where every row begins on a 16-byte boundary (StepBuffers: a
    pitch of whole 128-byte lines) and one member per store otherwise (align_rows=False: rows M*m+1 words apart, odd) -- same rows,
    same join, for set sizes of every remainder modulo 4; 32-bit keys and (M = 203, 4 hops) 64-bit keys, which stay on the narrow path."""
    from gpu_helpers import sym_graph
    ptr_, idx = sym_graph(30_000, 150_000, seed=41, hubs=3)
    csr = sp.DeviceCSR(ptr_, idx)
    B = 3000
    rng = np.random.default_rng(11)
    edge = torch.from_numpy(rng.integers(0, 30_000, (2, B)).astype(np.int64)).cuda()
    out = []
    for aligned in (True, False):
        bufs = sp.StepBuffers(csr, B, num_walks=M, num_steps=hops, rng="philox", align_rows=aligned)
        assert (bufs.stride % 4 == 0) == aligned
        xz, ind, sets = sp.sample_and_gather(csr, edge, num_walks=M, num_steps=hops, seed=5, rng="philox", buffers=bufs)
        sets.resolve()
        n = sets.nsize.cpu().numpy()
        assert len({int(v) % 4 for v in n}) == 4           # every remainder of a row's length occurs
        out.append((xz[: int(ind[-1])].cpu().numpy().copy(), ind.cpu().numpy().copy(), n.copy()))
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
