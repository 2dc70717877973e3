"""GPU (MI355X): round-5 additions, through the C ABI, against the oracle / the reference's goldens / the other forms.
  * a 64-bit key-rows batch that is numbered after the fact is sampled again with the configuration AND the
    rand_r stream positions of its first sampling (calls_before, rng_streams, cap_root_degree), not with defaults."""
import numpy as np
import pytest
import torch

import oracle
from gpu_helpers import _oracle_spg, dir_graph, sp, sym_graph  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,hops", [(200, 4), (100, 3)])
@pytest.mark.parametrize("kw", [dict(prefix=40), dict(cap_root_degree=False), dict(prefix=7, cap_root_degree=False)])
def test_key_rows_numbered_after_the_fact_describe_the_batch_that_was_joined(sp, M, hops, kw):
    """sets.number() / enc / to_csr() of a key-rows batch (64-bit keys: sampled again with the table form; 32-bit keys: registered
    from the rows, tags from walk_tags) == the table-form sample made with the SAME keyword arguments, rand_r stream included"""
    ptr_, idx = sym_graph(6000, 50000, seed=17, hubs=3)          # hubs: a root of degree > M is where cap_root_degree matters
    csr = sp.DeviceCSR(ptr_, idx)
    q = np.random.default_rng(4).integers(0, 6000, 900).astype(np.int32)
    deg = np.diff(ptr_)
    q[:3] = np.argsort(-deg)[:3]                                   # the hubs themselves are roots
    kw = dict(kw)
    prefix = q[::-1][: kw.pop("prefix", 0)].copy()                 # roots "sampled before" this batch on the same rand_r stream
    if prefix.size:
        from surel_plus_amd.shard import rand_r_calls
        kw["calls_before"] = rand_r_calls(csr.indptr, prefix, M, hops)
        assert kw["calls_before"] > 0
    zk, sk = sp.sample_spg(csr, q, num_walks=M, num_steps=hops, seed=19, rng="rand_r", strided=True, number_rows=False, **kw)
    assert sk.keyrows and sk.key64 == (hops == 4 and M >= 128)
    zt, st_ = sp.sample_spg(csr, q, num_walks=M, num_steps=hops, seed=19, rng="rand_r", strided=True, number_rows=True,
                            key_rows=False, **kw)
    assert not st_.keyrows
    assert torch.equal(sk.nsize, st_.nsize)
    assert sk.c == st_.c and torch.equal(sk.number().ukeys, st_.number().ukeys)
    assert torch.equal(sk.enc_int16(), st_.enc_int16())
    a, b = zk.to_csr(), zt.to_csr()
    assert torch.equal(a.indptr, b.indptr) and torch.equal(a.indices, b.indices) and torch.equal(a.data, b.data)
    # ... and what the reference's one sequential stream gives these roots when `prefix` was sampled in front of them
    o_nsize, o_remap, _ = oracle.gset_sampler(ptr_, idx, np.concatenate([prefix, q]), num_walks=M, num_steps=hops, rng="rand_r", seed=19)
    assert np.array_equal(sk.nsize.cpu().numpy(), o_nsize[prefix.size:])
    skip = int(o_nsize[: prefix.size].sum())
    rows = np.split(o_remap[0][skip:], np.cumsum(o_nsize[prefix.size:])[:-1])
    assert np.array_equal(a.indices.cpu().numpy(), np.concatenate([np.sort(r) for r in rows]))


@pytest.mark.parametrize("M", [1, 2, 3, 5, 7, 10, 100, 127, 128, 200, 255, 256, 300, 1000, 4095, 4096, 5000])
def test_keyed_join_divides_every_count_like_main_py(sp, M):
    """the key join unpacks count / num_walks itself (csrc/sjoin.hip:lp_quotient, an fma-refined reciprocal below 4,096 walks, a
    division beyond): every value a key field of this num_walks can hold, in both fields and both slots, against numpy's float32
    division (main.py:174: `float32(enc) / num_walks`)"""
    from surel_plus_amd._lib import check, lib
    m = 2
    shift = check(lib().subgacc_key_shift(M, m))
    assert m * shift + 1 <= 31
    vals = np.arange(1 << shift, dtype=np.int64)
    c1, c2 = vals, (vals * 7 + 3) & ((1 << shift) - 1)
    keys = ((c1 << shift) | c2) | (np.int64(1) << (m * shift)) * (vals & 1)          # (every other member carries the root flag)
    per = min(256, vals.size)
    n_rows = vals.size // per
    indptr = torch.arange(0, vals.size + 1, per, dtype=torch.int64, device="cuda")
    ids = torch.arange(per, dtype=torch.int32, device="cuda").repeat(n_rows)
    z = sp.SpG(indptr, ids, torch.from_numpy(keys.astype(np.int32)).cuda(), max_len=per, shape=(n_rows, per), max_data=0)
    z.keyrows, z.key_M, z.key_m = True, M, m
    rows = torch.arange(n_rows, device="cuda")
    xz, ind = sp.gather(torch.stack([rows, rows]), z, "cuda", ptr=True, encode=z.slot_table())
    got = xz.cpu().numpy()
    assert got.shape == (2 * vals.size, 2, m + 1)
    want = np.stack([(vals & 1).astype(np.float32), c1.astype(np.float32) / np.float32(M), c2.astype(np.float32) / np.float32(M)], axis=1)
    for half in (got[: vals.size], got[vals.size:]):              # (u, u): own row and partner row are the same member
        assert np.array_equal(half[:, 0, :], want) and np.array_equal(half[:, 1, :], want)


@pytest.mark.parametrize("key_rows", [True, False])
def test_stale_member_count_is_refused_for_table_rows_too(sp, key_rows):
    """the freshness stamp of a buffered step's sets is not a privilege of key rows -- a root-dedup step over the
    TABLE form of the rows (key_rows=False) counts its members lazily from the same buffers"""
    N, M, hops, B = 4000, 200, 3, 256
    ptr_, idx = sym_graph(N, 16000, seed=12)
    csr = sp.DeviceCSR(ptr_, idx)
    bufs = sp.StepBuffers(csr, B, num_walks=M, num_steps=hops, dedup_roots=True, key_rows=key_rows)
    assert bufs.keyrows == key_rows
    rs = np.random.default_rng(5)
    e1 = torch.from_numpy(rs.integers(0, 50, (2, B))).cuda()
    e2 = torch.from_numpy(rs.integers(0, N, (2, B))).cuda()
    _, _, s1 = sp.sample_and_gather(csr, e1, num_walks=M, num_steps=hops, rng="philox", buffers=bufs, dedup_roots=True)
    s1.resolve()
    o = oracle.gset_sampler(ptr_, idx, np.unique(e1.cpu().numpy()), num_walks=M, num_steps=hops, rng="philox", nthreads=8)
    assert s1.X == int(o[0].sum())
    _, _, s2 = sp.sample_and_gather(csr, e1, num_walks=M, num_steps=hops, rng="philox", buffers=bufs, dedup_roots=True)
    s2.resolve()
    sp.sample_and_gather(csr, e2, num_walks=M, num_steps=hops, rng="philox", buffers=bufs, dedup_roots=True)[2].resolve()
    with pytest.raises(sp.SubgAccError, match="later batch"):
        s2.X


def test_gather_many_refuses_what_it_cannot_honour_and_captured_join_takes_a_stream(sp):
    """a LIST of batches with out= / lazy=True used to ignore both silently; CapturedJoin.__call__ lost its
    stream= keyword (CapturedStep / CapturedStepPool.submit have it)"""
    from test_gpu_round4 import _store
    N = 3000
    csr, z, enc, o_spg, _ = _store(sp, N=N, M=40)
    table = torch.from_numpy(enc.astype(np.float32) / np.float32(40)).cuda()          # Z_SF as main.py:174 makes it
    rs = np.random.default_rng(2)
    batches = [torch.from_numpy(rs.integers(0, N, (2, 64))).cuda() for _ in range(3)]
    ref = [sp.gather(b, z, "cuda", ptr=True, encode=table) for b in batches]
    got = sp.gather_many(batches, z, "cuda", ptr=True, encode=table)
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got, ref))
    buf = torch.empty(3 * 128 * z.max_len * 2 * table.shape[1], dtype=torch.float32, device="cuda")
    with pytest.raises(ValueError, match="one \\[nb, 2, B\\] array"):
        sp.gather_many(batches, z, "cuda", ptr=True, encode=table, out=buf, lazy=True)
    with pytest.raises(ValueError):
        sp.gather_many(batches, z, "cuda", ptr=True, encode=table, lazy=True)
    lazy = sp.gather_many(torch.stack(batches), z, "cuda", ptr=True, encode=table, out=buf, lazy=True)      # the array form honours both
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(lazy, ref))
    cj = sp.CapturedJoin(z, 64, encode=table)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    xz, ind = cj(batches[1], stream=side).finish()
    assert torch.equal(xz, ref[1][0]) and torch.equal(ind, ref[1][1])
    xz, ind = cj(batches[2]).finish()                                  # and without: the current stream, as before
    assert torch.equal(xz, ref[2][0]) and torch.equal(ind, ref[2][1])


@pytest.mark.parametrize("payload", ["table3", "table4", "table7", "keyed", "float"])
@pytest.mark.parametrize("B", [37, 3000])          # 37 pairs: two workgroups per pair (split), 3,000: one
def test_pair_join_at_every_span_and_trip_boundary(sp, payload, B):
    """sjoin_keypair_kernel / sjoin_f64pair_kernel hold the shorter row S of a pair in registers (four trips of NT members), stage
    the longer row T, search once, then emit 64-row spans: rows of 0 / 1 / 63 / 64 / 65 / 127 / 128 / 129 / 255 / 256 / 257 / 511 /
    512 / 513 / 600 / 1,030 / 1,500 / 3,000 members -- every span and trip boundary, rows beyond the register trips (the
    span-by-span path), empty rows, (u,u) pairs, short-with-long and long-with-short pairs -- against the oracle, for every payload
    form of the resident store, with segment pointers and with segment ids."""
    lens = np.array([0, 1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 600, 1030, 1500, 3000])
    rs = np.random.default_rng(7)
    n_rows, n_cols = 400, 6000
    row_len = rs.choice(lens, n_rows)
    row_len[: lens.size] = lens
    indptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(row_len, out=indptr[1:])
    ids = np.concatenate([np.sort(rs.choice(n_cols, L, replace=False)) for L in row_len]).astype(np.int32)
    X = int(indptr[-1])
    e = rs.integers(0, n_rows, (2, B))
    e[:, : lens.size] = np.stack([np.arange(lens.size), np.arange(lens.size)[::-1]])        # every length against every other
    e[1, lens.size: lens.size + 6] = e[0, lens.size: lens.size + 6]                         # (u,u)
    M = 200
    if payload == "float":
        data = rs.random(X) * 3.0
        z = sp.SpG(torch.from_numpy(indptr).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(data).cuda())
        enc_dev, enc_host = None, None
    else:
        k = {"table3": 3, "table4": 4, "table7": 7, "keyed": 4}[payload]
        c = 500
        counts = rs.integers(0, M + 1, (c + 1, k)).astype(np.int64)
        counts[0] = 0
        counts[:, 0] = (rs.integers(0, 2, c + 1) * M)
        counts[0, 0] = 0
        counts[1:, 1] = np.maximum(counts[1:, 1], 1)                                         # a real LP row is never all zero
        data = rs.integers(1, c + 1, X).astype(np.int32)
        z = sp.SpG(torch.from_numpy(indptr).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(data).cuda())
        enc_host = counts.astype(np.float32) / np.float32(M)
        if payload == "keyed":
            z = z.keyed(counts.astype(np.int16), M)
            enc_dev = z.slot_table()
        else:
            enc_dev = torch.from_numpy(enc_host).cuda()
    spg_host = (indptr, ids, data)
    edge = torch.from_numpy(e).cuda()
    for ptr in (True, False):
        xz, ind = sp.gather(edge, z, "cuda", ptr=ptr, encode=enc_dev)
        oxz, oind = oracle.gather(e, spg_host, ptr=ptr, encode=enc_host)
        assert np.array_equal(ind.cpu().numpy(), oind)
        assert np.array_equal(xz.cpu().numpy(), oxz)
    if payload == "table3":        # the count form over the same pairs (sjoin_counts_kernel: the same one-directional plan, rows of up
        # to two register trips and beyond): C[j, p] = occurrences of LP row p in either slot of segment j, from the oracle's pairs
        own, partner = oracle.pair_segments(e)
        seg, pairs = oracle.sjoin(indptr, ids, data, own, partner)
        want = np.zeros((2 * B, c + 1), np.float32)
        np.add.at(want, (np.repeat(np.arange(2 * B), np.diff(seg))[:, None].repeat(2, 1), pairs), 1.0)
        C, sizes = sp.gather_counts(edge, z, c + 1)
        assert np.array_equal(sizes.cpu().numpy(), np.diff(seg)) and np.array_equal(C.cpu().numpy(), want)


@pytest.mark.parametrize("graph", [False, True])
def test_one_call_join_sizes_in_one_launch(sp, graph):
    """subgacc_sjoin_fill_v2 with SUBGACC_JOIN_OPT_SIZES (CapturedJoin's call): the size pass as ONE launch -- a single-pass scan
    whose state cleans itself -- then the fill.  Segment pointers and rows equal gather()'s for 1 tile (2,048 segments), 2 tiles,
    a full look-back window and beyond it (70,000 pairs = 69 tiles), call after call on the same state; a row outside the store
    raises IndexError at finish(); a state that is not clean ends in an error (never a hang) and works again afterwards."""
    from surel_plus_amd.graphs import ppr_like_spg
    N = 5000
    zf = ppr_like_spg(N, 40, seed=5)
    rs = np.random.default_rng(11)
    for B in (0, 1, 1024, 1025, 70000):
        cj = sp.CapturedJoin(zf, B, graph=graph)
        for rep in range(3):
            e = torch.from_numpy(rs.integers(0, N, (2, B))).cuda()
            xz, ind = cj(e).finish()
            wxz, wind = sp.gather(e, zf, "cuda", ptr=True, encode=None)
            assert torch.equal(ind, wind) and torch.equal(xz, wxz), (B, rep)
            assert int(cj._state.view(torch.int64)[: 8 + -(-2 * B // 1024)].abs().sum().item()) == 0   # header + one word per tile: left clean
    e[0, B // 2] = N                                        # a row outside the store: flagged by the size pass, never dereferenced
    with pytest.raises(IndexError):
        cj(e).finish()
    e[0, B // 2] = 0
    cj._state.view(torch.int64)[0] = 5                      # a ticket left behind by a launch that never finished
    with pytest.raises(sp.SubgAccError, match="not clean"):
        cj(e).finish()
    xz, ind = cj(e).finish()
    wxz, wind = sp.gather(e, zf, "cuda", ptr=True, encode=None)
    assert torch.equal(ind, wind) and torch.equal(xz, wxz)


def test_one_call_join_over_strided_key_rows_matches_the_two_step_form(sp):
    """OPT_SIZES over the strided key rows of a transient batch (row_len + row_stride: the on-demand step's store), through the
    descriptor itself: out_seg, the rows and host_tail equal subgacc_sjoin_sizes_rows + the plain fill; flags[3] is ORed into"""
    from surel_plus_amd import _lib
    from surel_plus_amd.graphs import query_pairs
    from surel_plus_amd.spg import sample_spg
    from surel_plus_amd.spjoin import _arange_segments, sjoin
    from gpu_helpers import sym_graph
    ptr_, idx = sym_graph(4000, 30000, seed=2, hubs=2)
    csr = sp.DeviceCSR(ptr_, idx)
    B = 3000
    e = query_pairs(csr, B, seed=4)
    z, sets = sample_spg(csr, e.reshape(-1).to(torch.int32), num_walks=100, num_steps=3, seed=9, rng="philox", strided=True, fused=True,
                         number_rows=False)
    assert sets.keyrows and getattr(z, "keyrows", False)
    own, partner = _arange_segments(B, "cuda")
    xz, ind, _ = sjoin(z, own, partner, z.slot_table(), pair_block=B)
    S = 2 * B
    out = torch.empty_like(xz)
    seg = torch.zeros(S + 1, dtype=torch.int64, device="cuda")
    state = torch.zeros(_lib.lib().subgacc_sjoin_workspace_bytes(S), dtype=torch.uint8, device="cuda")
    host = torch.full((2,), -7, dtype=torch.int64).pin_memory()
    flags = torch.tensor([7, 7, 7, 128], dtype=torch.int32, device="cuda")
    for _ in range(2):
        _lib.join_fill(_lib.JOIN_ROWS, _lib.JOIN_KEY64 if z.sets.key64 else _lib.JOIN_KEY32, options=_lib.JOIN_OPT_SIZES, row_len=z.nsize,
                       n_rows=z.n_rows, row_stride=z.stride, ids=z.indices, payload=z.slot, own=own, partner=partner, S=S, pair_block=B,
                       num_walks=100, num_steps=3, out_xz=out, flags=flags, out_seg=seg, size_state=state, size_state_bytes=state.numel(),
                       host_tail=host)
        torch.cuda.synchronize()
        assert torch.equal(seg, ind) and torch.equal(out, xz)
        assert host.tolist() == [xz.shape[0], 0] and flags.tolist() == [7, 7, 7, 128]
        out.zero_(), seg.zero_()
    # what the descriptor refuses: a seg next to OPT_SIZES, no state, a state too small, another form
    kw = dict(row_len=z.nsize, n_rows=z.n_rows, row_stride=z.stride, ids=z.indices, payload=z.slot, own=own, S=S, pair_block=B, num_walks=100,
              num_steps=3, out_xz=out, flags=flags, options=_lib.JOIN_OPT_SIZES)
    for bad in (dict(out_seg=seg, seg=seg, size_state=state, size_state_bytes=state.numel()), dict(out_seg=seg),
                dict(out_seg=seg, size_state=state, size_state_bytes=64), dict(size_state=state, size_state_bytes=state.numel())):
        with pytest.raises((TypeError, MemoryError, sp.SubgAccError)):      # (the mirror raises the reference's exception types)
            _lib.join_fill(_lib.JOIN_ROWS, _lib.JOIN_KEY32, **kw, **bad)


def test_join_pool_lanes_on_their_own_streams_equal_gather(sp):
    """CapturedJoinPool: CapturedJoins taken in turn, each on its own stream (the serving loop of short-row joins) -- every batch's
    (xz, indptr) equals gather()'s, whatever is still in flight on the other lanes; a lane that is busy refuses"""
    from surel_plus_amd.graphs import ppr_like_spg
    N, B = 4000, 3000
    zf = ppr_like_spg(N, 60, seed=8)
    rs = np.random.default_rng(5)
    batches = [torch.from_numpy(rs.integers(0, N, (2, B))).cuda() for _ in range(7)]
    want = [sp.gather(b, zf, "cuda", ptr=True, encode=None) for b in batches]
    torch.cuda.synchronize()
    pool = sp.CapturedJoinPool(zf, B, lanes=3)
    pend = []
    for i, b in enumerate(batches):
        if len(pend) == 3:
            j, t = pend.pop(0)
            xz, ind = pool.finish(t)
            assert torch.equal(ind, want[j][1]) and torch.equal(xz, want[j][0]), j
        pend.append((i, pool.submit(b)))
    with pytest.raises(RuntimeError, match="in flight"):
        pool.submit(batches[0])
    for j, t in pend:
        xz, ind = pool.finish(t)
        assert torch.equal(ind, want[j][1]) and torch.equal(xz, want[j][0]), j


def test_publish_words_reaches_pinned_memory(sp):
    """subgacc_publish_words: a step's sizes and status words to pinned host memory by a one-wave kernel (no copy engine in the stream)"""
    from surel_plus_amd import _lib
    for n in (1, 5, 64, 1000, 4096):
        src = torch.arange(n, dtype=torch.int64, device="cuda") * 3 - 7
        host = torch.full((n,), -1, dtype=torch.int64).pin_memory()
        _lib.publish(src, host)
        torch.cuda.synchronize()
        assert torch.equal(host, src.cpu())
    with pytest.raises((TypeError, sp.SubgAccError)):
        _lib.check(_lib.lib().subgacc_publish_words(_lib.ptr(src), 5000, host.data_ptr(), _lib.stream_ptr()))


def test_one_call_hgather_equals_hgather(sp):
    """CapturedJoin(triplets=True) / CapturedJoinPool(..., triplets=True): hgather(hedge [3, B]) from a resident store as one library call
    per batch -- (xz, segment ids) bit for bit what hgather() returns (train.py:48-72), table store and keyed store"""
    from test_gpu_round4 import _store
    N = 3000
    csr, z, enc, o_spg, _ = _store(sp, N=N, M=40)
    table = torch.from_numpy(enc.astype(np.float32) / np.float32(40)).cuda()
    zk = z.keyed(enc, 40)
    rs = np.random.default_rng(3)
    B = 700
    hs = [torch.from_numpy(rs.integers(0, N, (3, B))).cuda() for _ in range(4)]
    hs[1][2, 5] = hs[1][0, 5]                                   # w == u
    for store, encode in ((z, table), (zk, zk.slot_table())):
        want = [sp.hgather(h, store, "cuda", encode=encode) for h in hs]
        cj = sp.CapturedJoin(store, B, encode=encode, triplets=True)
        for h, (wxz, wids) in zip(hs, want):
            xz, ids = cj(h).finish()
            assert torch.equal(ids, wids) and torch.equal(xz, wxz)
        xz, ids = cj(hs[0].cpu().numpy()).finish()                 # host input goes through the static buffer
        assert torch.equal(ids, want[0][1]) and torch.equal(xz, want[0][0])
        pool = sp.CapturedJoinPool(store, B, lanes=2, encode=encode, triplets=True)
        tickets = [pool.submit(hs[0]), pool.submit(hs[1])]
        for t, (wxz, wids) in zip(tickets, want[:2]):
            xz, ids = pool.finish(t)
            assert torch.equal(ids, wids) and torch.equal(xz, wxz)
    with pytest.raises(NotImplementedError):
        sp.CapturedJoin(z, B, triplets=True)


def test_one_call_join_size_pass_beyond_the_resident_grid(sp):
    """the single-pass scan of SUBGACC_JOIN_OPT_SIZES with more tiles than the chip holds at once (3,000,000 segments = 2,930
    tiles of 256 lanes; ~2,048 are resident): tiles take their numbers from a ticket, so a tile only ever waits for tiles that
    run already -- the segment pointers equal a cumulative sum, the rows equal gather()'s, twice on the same state"""
    from surel_plus_amd.graphs import ppr_like_spg
    N, B = 4000, 1_500_000
    zf = ppr_like_spg(N, 12, seed=6)
    e = torch.randint(0, N, (2, B), device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    cj = sp.CapturedJoin(zf, B)
    wxz, wind = sp.gather(e, zf, "cuda", ptr=True, encode=None)
    for _ in (0, 1):
        xz, ind = cj(e).finish()
        assert torch.equal(ind, wind) and torch.equal(xz, wxz)
    assert int(cj._state.view(torch.int64)[: 8 + 2930].abs().sum().item()) == 0


@pytest.mark.parametrize("name", ["sjoin_int.npz", "sjoin_float.npz", "sjoin_int_emptyrows.npz"])
def test_one_call_join_matches_reference_golden(sp, name):
    """the reference's own outputs (train.gather run in the build container, tests/golden/): CapturedJoin / CapturedJoinPool -- one
    library call per batch -- give the same (xz, indptr), bit for bit, as the fixtures hold for ptr=True"""
    from gpu_helpers import _load, _spg_from_golden
    g = _load(name)
    z = _spg_from_golden(sp, g)
    enc = torch.from_numpy(g["encode"]).cuda() if g["encode"].size else None
    edge = np.asarray(g["edge"])
    cj = sp.CapturedJoin(z, edge.shape[1], encode=enc)
    for e in (edge, torch.from_numpy(edge).cuda()):                # NumPy endpoints (as the fixtures have them) and a device tensor
        xz, ind = cj(e).finish()
        assert xz.dtype == torch.float32 and ind.dtype == torch.int64
        assert np.array_equal(xz.cpu().numpy(), g["xz_ptr1"]) and np.array_equal(ind.cpu().numpy(), g["ind_ptr1"])
    pool = sp.CapturedJoinPool(z, edge.shape[1], lanes=2, encode=enc)
    xz, ind = pool.finish(pool.submit(torch.from_numpy(edge).cuda()))
    assert np.array_equal(xz.cpu().numpy(), g["xz_ptr1"]) and np.array_equal(ind.cpu().numpy(), g["ind_ptr1"])


def test_one_call_hgather_matches_reference_golden(sp):
    """train.hgather's own output (tests/golden/hjoin_int.npz) through CapturedJoin(triplets=True)"""
    from gpu_helpers import _load, _spg_from_golden
    g = _load("hjoin_int.npz")
    z = _spg_from_golden(sp, g)
    hedge = np.asarray(g["hedge"])
    cj = sp.CapturedJoin(z, hedge.shape[1], encode=torch.from_numpy(g["encode"]).cuda(), triplets=True)
    xz, ids = cj(torch.from_numpy(hedge).cuda()).finish()
    assert np.array_equal(xz.cpu().numpy(), g["xz"]) and np.array_equal(ids.cpu().numpy(), g["ind"])
