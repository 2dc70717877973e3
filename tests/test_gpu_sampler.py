"""GPU (MI355X), SURVEY 8(a) rows a1-a8: the sampler -- gset_sampler / walk_sampler, the fused-row and general walk kernels, hop records,
the table of distinct LP rows and its numbering, the work list -- through the C ABI against the reference's golden vectors and the oracle.
Integer / index outputs are compared BIT-EXACT."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, golden_files
from gpu_helpers import _load, _oracle_counts, _oracle_spg, _reference_style_attn, _reference_style_lstm, _spg_from_golden, _walkjoin_inputs, dir_graph, sp, sym_graph  # noqa: F401

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ scan
@pytest.mark.parametrize("n", [0, 1, 7, 2047, 2048, 2049, 100000, 2048 * 2048 + 5])
def test_exclusive_scan(sp, n):
    from surel_plus_amd._lib import check, lib, ptr, stream_ptr
    L = lib()
    x = torch.randint(0, 1000, (n,), dtype=torch.int32, device="cuda")
    out = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ws = torch.empty(L.subgacc_scan_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    check(L.subgacc_exclusive_scan_i32(ptr(x), n, ptr(out), ptr(ws), ws.numel(), stream_ptr()))
    ref = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum(x.long(), 0)])
    assert torch.equal(out, ref)


# ------------------------------------------------------------------------------- gset_sampler
@pytest.mark.parametrize("name", golden_files("gset_"))
def test_gset_sampler_matches_reference_golden(sp, name):
    g = _load(name)
    out = sp.gset_sampler(g["indptr"], g["indices"], g["query"], num_walks=int(g["M"]), num_steps=int(g["m"]),
                          bucket=int(g["bucket"]), seed=int(g["seed"]), debug=1)
    assert out[0].dtype == np.int32 and out[1].dtype == np.int32 and out[2].dtype == np.int16
    assert np.array_equal(out[0], g["nsize"])
    assert np.array_equal(out[1], g["remap"])
    assert np.array_equal(out[2], g["enc"])
    assert np.array_equal(out[3], g["raw"])


@pytest.mark.parametrize("rng", ["rand_r", "philox"])
@pytest.mark.parametrize("M,m,N,E,hubs", [(200, 2, 20000, 80000, 4), (200, 3, 6000, 200000, 0), (100, 4, 3000, 9000, 2),
                                          (7, 5, 500, 1500, 1), (300, 2, 2000, 300000, 0)])
def test_gset_sampler_matches_oracle(sp, rng, M, m, N, E, hubs):
    ptr_, idx = sym_graph(N, E, seed=M + m, hubs=hubs)
    q = np.random.default_rng(3).permutation(N)[: min(N, 4000)]
    a = sp.gset_sampler(ptr_, idx, q, num_walks=M, num_steps=m, seed=99, debug=1, rng=rng)
    b = oracle.gset_sampler(ptr_, idx, q, num_walks=M, num_steps=m, seed=99, debug=True, rng=rng)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("rng", ["rand_r", "philox"])
def test_pipelined_walk_with_every_root_shuffled(sp, rng):
    """Every degree > M = 200 and more roots than resident workgroups (2048): each workgroup of the persistent walk
    kernel pipelines several roots whose first hop is the partial Fisher-Yates of subg_acc.c:763-776 -- the draws of
    root k+1 land in the LDS array root k's lanes are still chasing unless a barrier separates them."""
    ptr_, idx = sym_graph(6000, 900000, seed=21)
    assert int(np.diff(ptr_).min()) > 200
    q = np.random.default_rng(5).permutation(6000)
    a = sp.gset_sampler(ptr_, idx, q, num_walks=200, num_steps=2, seed=31, debug=1, rng=rng)
    b = oracle.gset_sampler(ptr_, idx, q, num_walks=200, num_steps=2, seed=31, debug=True, rng=rng)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    from surel_plus_amd.spg import sample_spg
    csr = sp.DeviceCSR(ptr_, idx)
    for fused in (False, True):          # the fused-row kernel on the same all-shuffled batch
        z, sets = sample_spg(csr, q, num_walks=200, num_steps=2, seed=31, rng=rng, fused=fused)
        oi, od, ov = oracle.spg_build(b[0], b[1])
        assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices[: z.nnz].cpu().numpy(), od)
        assert np.array_equal(z.data[: z.nnz].cpu().numpy(), ov)


def test_gset_multichunk_and_int64_indptr(sp):
    ptr_, idx = sym_graph(5000, 30000, seed=5, hubs=1)
    q = np.arange(5000)
    from surel_plus_amd.sampler import DeviceCSR, sample_sets
    b = oracle.gset_sampler(ptr_, idx, q, num_walks=50, num_steps=3, seed=4, debug=True)
    for ip in (ptr_, ptr_.astype(np.int64)):
        csr = DeviceCSR(ip, idx)
        s = sample_sets(csr, q, num_walks=50, num_steps=3, seed=4, staging_bytes=151 * 12 * 700)   # 8 chunks
        assert np.array_equal(s.nsize.cpu().numpy(), b[0])
        assert np.array_equal(torch.stack([s.ids, s.get_sf()]).cpu().numpy(), b[1])
        assert np.array_equal(s.enc_int16().cpu().numpy(), b[2])
        tab = s.feature_table().cpu().numpy()
        assert np.array_equal(tab, oracle.enc_table(b[2]).astype(np.float32) / np.float32(50))


def test_philox_is_schedule_independent(sp):
    """Sets of a root do not depend on the batch it is sampled in (counter = seed, root id, walk, step)."""
    ptr_, idx = sym_graph(3000, 20000, seed=8)
    from surel_plus_amd.sampler import DeviceCSR, sample_sets
    csr = DeviceCSR(ptr_, idx)
    full = sample_sets(csr, np.arange(3000), num_walks=64, num_steps=3, seed=1, rng="philox", dedup=False)
    part = sample_sets(csr, np.arange(1000, 1200), num_walks=64, num_steps=3, seed=1, rng="philox", dedup=False)
    off = full.row_off.cpu().numpy()
    assert np.array_equal(full.ids.cpu().numpy()[off[1000]:off[1200]], part.ids.cpu().numpy())
    assert np.array_equal(full.keys.cpu().numpy()[off[1000]:off[1200]], part.keys.cpu().numpy())


def test_rand_r_dead_end_is_replayed(sp):
    """A directed graph with a sink: the reference draws nothing on it and carries on (subg_acc.c:804-808), so the stream
    positions are data dependent -- the walk kernel reports it, the host replays the stream (subgacc_rng_replay) and the
    result is the reference's (the directed goldens of tests/golden are checked by the golden tests above)."""
    indptr = np.array([0, 2, 3, 3], np.int32)      # node 2 has no out-edges
    indices = np.array([1, 2, 2], np.int32)
    for rng in ("rand_r", "philox"):
        out = sp.gset_sampler(indptr, indices, np.array([0, 1, 2]), num_walks=4, num_steps=3, rng=rng, debug=1)
        ref = oracle.gset_sampler(indptr, indices, np.array([0, 1, 2]), num_walks=4, num_steps=3, rng=rng, debug=True)
        for x, y in zip(out, ref):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("M,m,N,E,hubs,idx64", [(200, 2, 3000, 9000, 2, False), (64, 3, 5000, 40000, 0, True), (20, 4, 800, 1500, 1, False),
                                                (300, 2, 1500, 200000, 0, False), (7, 5, 400, 900, 3, False)])
def test_rand_r_on_directed_graphs_matches_the_sequential_stream(sp, M, m, N, E, hubs, idx64):
    """rng='rand_r' accepts what the reference accepts: graphs with dead ends.  gset_sampler, the SpG pipeline (fused and
    general, several chunks) and walk_sampler with 1..5 streams against the oracle's sequential loops; a lazy batch cannot
    replay by itself and says so at resolve()."""
    from surel_plus_amd import sampler
    from surel_plus_amd.spg import sample_spg
    ptr_, idx = dir_graph(N, E, seed=M + m, hubs=hubs)
    q = np.random.default_rng(2).permutation(N)[: min(N, 2500)]
    ip = ptr_.astype(np.int64) if idx64 else ptr_
    a = sp.gset_sampler(ip, idx, q, num_walks=M, num_steps=m, seed=17, debug=1)
    b = oracle.gset_sampler(ptr_, idx, q, num_walks=M, num_steps=m, seed=17, debug=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    csr = sp.DeviceCSR(ip, idx)
    oi, ox, od = oracle.spg_build(b[0], b[1])
    for kw in ({"fused": True}, {"fused": False}, {"fused": True, "staging_bytes": (M * m + 1) * 8 * (len(q) // 5 + 1)}):
        z, sets = sample_spg(csr, q, num_walks=M, num_steps=m, seed=17, rng="rand_r", **kw)
        assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices[: z.nnz].cpu().numpy(), ox)
        assert np.array_equal(z.data[: z.nnz].cpu().numpy(), od) and np.array_equal(sets.enc_int16().cpu().numpy(), b[2])
    for T, rep in ((1, False), (3, True), (5, True)):
        w, obj = sp.walk_sampler(ip, idx, q, num_walks=M, num_steps=m, nthread=T, seed=5, replacement=rep)
        ow, on, oi_, oc = oracle.walk_sampler(ptr_, idx, q, num_walks=M, num_steps=m, nthread=T, seed=5, replacement=rep)
        assert np.array_equal(w, ow)
        off = np.concatenate([[0], np.cumsum(on)])
        assert all(np.array_equal(obj[i, 0], oi_[off[i]:off[i + 1]]) and np.array_equal(obj[i, 1], oc[off[i]:off[i + 1]])
                   for i in range(len(q)))
    # a lazy batch on a graph nobody has walked yet cannot replay by itself and says so at resolve() ...
    fresh = sp.DeviceCSR(ip, idx)
    lz = sampler.sample_sets(fresh, q, num_walks=M, num_steps=m, seed=17, rng="rand_r", lazy=True)
    with pytest.raises(sampler.RandRDeadEnd, match="lazy=False"):
        lz.resolve()
    # ... while the graph above REMEMBERS its dead ends (round 4): the lazy batch replays the stream from the start
    assert csr._rand_r_dead_ends
    lz = sampler.sample_sets(csr, q, num_walks=M, num_steps=m, seed=17, rng="rand_r", lazy=True).resolve()
    assert np.array_equal(lz.nsize.cpu().numpy(), b[0]) and np.array_equal(lz.ids.cpu().numpy(), b[1][0])


def test_reference_invariants_at_scale(sp):
    """subg_acc/test/test.py:34-45 on a 200k-root run (size-independent properties)."""
    from surel_plus_amd.graphs import powerlaw_graph
    from surel_plus_amd.sampler import sample_sets
    csr = powerlaw_graph(200_000, 8.2, seed=0)
    M, m = 200, 2
    s = sample_sets(csr, torch.arange(200_000, device="cuda", dtype=torch.int32), num_walks=M, num_steps=m, rng="philox")
    assert int(s.nsize.sum()) == s.X
    assert int(s.get_sf().max()) == s.c - 1
    enc = s.enc_int16().long()
    rows = enc[s.get_sf().long()]
    assert int((rows[:, 0] == M).sum()) == 200_000
    seg = torch.repeat_interleave(torch.arange(200_000, device="cuda"), s.nsize.long())
    colsum = torch.zeros((200_000, m + 1), dtype=torch.int64, device="cuda").index_add_(0, seg, rows)
    assert bool((colsum == M).all())
    assert bool((s.ids[s.row_off[:-1]] == torch.arange(200_000, device="cuda", dtype=torch.int32)).all())
    # members are unique inside a set
    key = seg * csr.num_nodes + s.ids.long()
    assert torch.unique(key).numel() == s.X


# ------------------------------------------------------------------------------- walk_sampler
@pytest.mark.parametrize("name", golden_files("walk_"))
def test_walk_sampler_matches_reference_golden(sp, name):
    g = _load(name)
    walks, obj = sp.walk_sampler(g["indptr"], g["indices"], g["query"], num_walks=int(g["M"]), num_steps=int(g["m"]),
                                 nthread=int(g["nthread"]), seed=int(g["seed"]), replacement=bool(g["replacement"]))
    assert walks.dtype == np.int32 and np.array_equal(walks, g["walks"])
    off = np.concatenate([[0], np.cumsum(g["nsize"])])
    for i in range(len(g["query"])):
        assert obj[i, 0].dtype == np.int32 and obj[i, 1].dtype == np.int32
        assert np.array_equal(obj[i, 0], g["ids"][off[i]:off[i + 1]])
        assert np.array_equal(obj[i, 1], g["counts"][off[i]:off[i + 1]])


def test_unique_table_grows_on_overflow(sp):
    """A deliberately tiny unique-row table (64 slots) must be detected as over-full and retried larger."""
    ptr_, idx = sym_graph(3000, 9000, seed=104, hubs=2)
    from surel_plus_amd.sampler import DeviceCSR, sample_sets
    s = sample_sets(DeviceCSR(ptr_, idx), np.arange(3000), num_walks=100, num_steps=4, seed=99, uniq_capacity=64)
    b = oracle.gset_sampler(ptr_, idx, np.arange(3000), num_walks=100, num_steps=4, seed=99)
    assert s.c > 64
    assert np.array_equal(torch.stack([s.ids, s.get_sf()]).cpu().numpy(), b[1])
    assert np.array_equal(s.enc_int16().cpu().numpy(), b[2])


def test_unique_numbering_large_table_path(sp):
    """force the element-scan numbering (used when the distinct LP rows exceed `small_limit`) and compare it
    with the direct ranking and with the oracle."""
    ptr_, idx = sym_graph(3000, 9000, seed=104, hubs=2)
    from surel_plus_amd.sampler import DeviceCSR, sample_sets
    csr = DeviceCSR(ptr_, idx)
    b = oracle.gset_sampler(ptr_, idx, np.arange(3000), num_walks=100, num_steps=4, seed=7)
    for limit in (0, 16):       # 0 -> default 8192 (direct ranking); 16 -> scan path since c >> 16
        s = sample_sets(csr, np.arange(3000), num_walks=100, num_steps=4, seed=7, uniq_small_limit=limit)
        assert s.c > 16
        assert np.array_equal(torch.stack([s.ids, s.get_sf()]).cpu().numpy(), b[1])
        assert np.array_equal(s.enc_int16().cpu().numpy(), b[2])


def test_standalone_dedup_of_packed_keys_matches_the_fused_path(sp):
    """subgacc_uniq_insert over already packed keys (dedup_lp_rows) == the insert fused into the compaction."""
    ptr_, idx = sym_graph(4000, 20000, seed=21, hubs=1)
    from surel_plus_amd.sampler import DeviceCSR, dedup_lp_rows, sample_sets
    csr = DeviceCSR(ptr_, idx)
    fused = sample_sets(csr, np.arange(4000), num_walks=64, num_steps=3, seed=2, rng="philox", keep_keys=True)
    plain = sample_sets(csr, np.arange(4000), num_walks=64, num_steps=3, seed=2, rng="philox", dedup=False)
    assert torch.equal(fused.keys, plain.keys) and torch.equal(fused.ids, plain.ids)
    dedup_lp_rows(plain)
    assert torch.equal(plain.sf, fused.get_sf()) and torch.equal(plain.ukeys, fused.ukeys)


@pytest.mark.parametrize("M,m,bucket", [(1, 1, -1), (1, 5, -1), (3, 1, -1), (1000, 2, -1), (64, 3, 1), (64, 3, 2), (255, 7, -1)])
@pytest.mark.parametrize("rng", ["rand_r", "philox"])
def test_gset_parameter_corners(sp, M, m, bucket, rng):
    """one walk, one hop, a 64 KB LDS table (M*m+1 = 2001), buckets that keep only the root, SHIFT*m+1 = 57 bits."""
    ptr_, idx = sym_graph(600, 3000, seed=M * 7 + m, hubs=1)
    q = np.concatenate([np.arange(600), [5, 5, 599]])
    a = sp.gset_sampler(ptr_, idx, q, num_walks=M, num_steps=m, bucket=bucket, seed=3, debug=1, rng=rng)
    b = oracle.gset_sampler(ptr_, idx, q, num_walks=M, num_steps=m, bucket=bucket, seed=3, debug=True, rng=rng)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_graph_of_isolated_nodes(sp):
    """every root isolated: sets are {root}, LP rows are [M, M, ..., M] (subg_acc.c:753-761)."""
    indptr = np.zeros(11, np.int32)
    indices = np.zeros(0, np.int32)
    for rng in ("rand_r", "philox"):
        a = sp.gset_sampler(indptr, indices, np.arange(10), num_walks=9, num_steps=3, debug=1, rng=rng)
        b = oracle.gset_sampler(indptr, indices, np.arange(10), num_walks=9, num_steps=3, debug=True, rng=rng)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert np.array_equal(a[1][0], np.arange(10)) and a[2].tolist() == [[9, 9, 9, 9]]
    w, obj = sp.walk_sampler(indptr, indices, np.arange(10), num_walks=4, num_steps=2, replacement=True)
    assert np.array_equal(w, np.repeat(np.arange(10), 12).reshape(10, 12))


def test_key_width_errors_match_the_reference(sp):
    ptr_, idx = sym_graph(50, 100, seed=1)
    with pytest.raises(AssertionError, match="hasing key"):      # 9 hops * 8 bits + 1 > 64, subg_acc.c:911-915
        sp.gset_sampler(ptr_, idx, np.arange(5), num_walks=200, num_steps=9)
    with pytest.raises(TypeError):                                 # float CSR cannot be safely cast, subg_acc.c:663
        sp.gset_sampler(ptr_.astype(np.float64), idx, np.arange(5))
    with pytest.raises(TypeError):
        sp.gset_sampler(ptr_, idx.astype(np.int64), np.arange(5))
    # the query is force-cast (subg_acc.c:673): float ids are accepted and truncated
    a = sp.gset_sampler(ptr_, idx, np.arange(5, dtype=np.float64), num_walks=4, num_steps=2)
    b = oracle.gset_sampler(ptr_, idx, np.arange(5), num_walks=4, num_steps=2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("replacement", [False, True])
def test_walk_sampler_philox_and_many_streams(sp, replacement):
    ptr_, idx = sym_graph(700, 5000, seed=9, hubs=1)
    q = np.arange(700)
    for rng, T in (("philox", 1), ("rand_r", 7), ("rand_r", 700), ("rand_r", 1000)):
        walks, obj = sp.walk_sampler(ptr_, idx, q, num_walks=33, num_steps=4, nthread=T, seed=5, replacement=replacement,
                                     rng=rng)
        ow, on, oi, oc = oracle.walk_sampler(ptr_, idx, q, num_walks=33, num_steps=4, nthread=T, seed=5,
                                             replacement=replacement, rng=rng)
        assert np.array_equal(walks, ow)
        off = np.concatenate([[0], np.cumsum(on)])
        assert all(np.array_equal(obj[i, 0], oi[off[i]:off[i + 1]]) and np.array_equal(obj[i, 1], oc[off[i]:off[i + 1]])
                   for i in range(700))


@pytest.mark.parametrize("rng", ["rand_r", "philox"])
@pytest.mark.parametrize("M,m,N,E,hubs,bucket", [(200, 2, 20000, 80000, 4, -1), (200, 3, 6000, 200000, 0, -1),
                                                 (100, 4, 3000, 9000, 2, -1), (7, 5, 500, 1500, 1, -1),
                                                 (200, 4, 2000, 100000, 1, -1), (64, 3, 3000, 9000, 2, 10), (1, 1, 300, 900, 0, -1)])
def test_fused_spg_pipeline_matches_oracle(sp, rng, M, m, N, E, hubs, bucket):
    """one kernel per root: walk + dedup + LP + unique-row registration + sort by id (csrc/walk.hip, SPG mode)."""
    ptr_, idx = sym_graph(N, E, seed=M + m, hubs=hubs)
    q = np.concatenate([np.random.default_rng(3).permutation(N)[: min(N, 3000)], [0, 0, 1]])
    from surel_plus_amd.sampler import DeviceCSR
    csr = DeviceCSR(ptr_, idx)
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, m, 11, rng, bucket)
    for fused in (True, False):
        z, info = sp.sample_spg(csr, q, num_walks=M, num_steps=m, seed=11, rng=rng, bucket=bucket, fused=fused)
        assert np.array_equal(z.indptr.cpu().numpy(), oi), fused
        assert np.array_equal(z.indices.cpu().numpy(), ox), fused
        assert np.array_equal(z.data.cpu().numpy(), od), fused
        assert np.array_equal(info.enc_int16().cpu().numpy(), oenc), fused
        assert z.max_data == oenc.shape[0]


def test_fused_spg_multichunk_overflow_and_fallbacks(sp):
    ptr_, idx = sym_graph(4000, 16000, seed=31, hubs=2)
    ptr_[-1:]  # keep flake quiet
    q = np.arange(4000)
    from surel_plus_amd.sampler import DeviceCSR, sample_sets
    csr = DeviceCSR(ptr_.astype(np.int64), idx)                    # int64 row offsets
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, 100, 4, 5, "rand_r")
    # 9 chunks + a 64-slot table that must be regrown
    z, info = sp.sample_spg(csr, q, num_walks=100, num_steps=4, seed=5, fused=True, staging_bytes=401 * 8 * 450,
                            uniq_capacity=64)
    assert info.data is not None                                   # really the fused-row form
    assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices.cpu().numpy(), ox)
    assert np.array_equal(z.data.cpu().numpy(), od) and np.array_equal(info.enc_int16().cpu().numpy(), oenc)
    # more distinct rows than the direct ranking is allowed to handle -> None -> general pipeline
    assert sample_sets(csr, q, num_walks=100, num_steps=4, seed=5, uniq_small_limit=16, fused_rows=True) is None
    # M*m+1 > 1024 does not fit the fused kernel
    assert sample_sets(csr, q[:10], num_walks=300, num_steps=4, seed=5, fused_rows=True) is None
    z2, _ = sp.sample_spg(csr, q[:300], num_walks=300, num_steps=4, seed=5, rng="philox", fused=True)
    (oi2, ox2, od2), _ = _oracle_spg(ptr_, idx, q[:300], 300, 4, 5, "philox")
    assert np.array_equal(z2.indices.cpu().numpy(), ox2) and np.array_equal(z2.data.cpu().numpy(), od2)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("rng", ["rand_r", "philox"])
def test_lazy_pipeline_has_no_host_round_trip_and_the_same_result(sp, fused, rng):
    """lazy=True leaves every size on the device: sample -> SpG -> table -> SpJoin queue up asynchronously and give
    the same bytes as the eager form / the oracle once resolve() has read the sizes back."""
    ptr_, idx = sym_graph(6000, 40000, seed=13, hubs=2)
    q = np.random.default_rng(1).permutation(6000)[:2500]
    from surel_plus_amd.sampler import DeviceCSR
    csr = DeviceCSR(ptr_, idx)
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, 150, 3, 8, rng)
    z, sets = sp.sample_spg(csr, q, num_walks=150, num_steps=3, seed=8, rng=rng, fused=fused, lazy=True)
    assert sets.pending and z.indices.numel() == len(q) * 451          # capacity-sized, nothing read back yet
    table = sets.feature_table()
    edge = np.random.default_rng(2).integers(0, len(q), (2, 3000))
    xz, ind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    otab = oracle.enc_table(oenc).astype(np.float32) / np.float32(150)
    oxz, oind = oracle.gather(edge, (oi, ox, od), ptr=True, encode=otab)
    assert np.array_equal(xz.cpu().numpy(), oxz) and np.array_equal(ind.cpu().numpy(), oind)
    sets.resolve()
    assert not sets.pending and sets.c == oenc.shape[0] and sets.X == len(ox)
    X = sets.X
    assert np.array_equal(z.indptr.cpu().numpy(), oi)
    assert np.array_equal(z.indices[:X].cpu().numpy(), ox) and np.array_equal(z.data[:X].cpu().numpy(), od)
    assert np.array_equal(sets.enc_int16().cpu().numpy(), oenc)
    assert z.nnz == X and z.to_scipy().nnz == X


@pytest.mark.parametrize("rng", ["rand_r", "philox"])
@pytest.mark.parametrize("M,m,N,E,hubs,bucket", [(200, 2, 20000, 80000, 4, -1), (200, 3, 6000, 200000, 0, -1),
                                                 (100, 4, 3000, 9000, 2, -1), (7, 5, 500, 1500, 1, -1),
                                                 (255, 4, 2000, 100000, 1, -1), (64, 3, 3000, 9000, 2, 10), (1, 1, 300, 900, 0, -1)])
def test_finished_rows_from_the_general_walk_kernel_match_oracle(sp, rng, M, m, N, E, hubs, bucket):
    """subgacc_walk_sets + subgacc_finish_rows (the strided rows of the configurations where the general walk kernel is
    the faster one): rows, numbering of the distinct LP rows and LP table are the oracle's, bit for bit; the join from
    these rows equals the join from the oracle's SpG."""
    from surel_plus_amd.spg import StridedSpG, sample_spg
    ptr_, idx = sym_graph(N, E, seed=M + m, hubs=hubs)
    q = np.random.default_rng(3).permutation(N)[: min(N, 3000)]
    (oi, od, ov), oenc = _oracle_spg(ptr_, idx, q, M, m, 99, rng, bucket)
    csr = sp.DeviceCSR(ptr_, idx)
    for lazy in (False, True):
        z, sets = sample_spg(csr, q, num_walks=M, num_steps=m, seed=99, rng=rng, bucket=bucket, fused=False, strided=True,
                             lazy=lazy)
        assert isinstance(z, StridedSpG) and sets.strided
        zc = z.to_csr()
        assert np.array_equal(zc.indptr.cpu().numpy(), oi) and np.array_equal(zc.indices.cpu().numpy(), od)
        assert np.array_equal(zc.data.cpu().numpy(), ov)
        assert np.array_equal(sets.enc_int16().cpu().numpy(), oenc)
    edge = np.random.default_rng(5).integers(0, len(q), (2, 500))
    table = oracle.enc_table(oenc).astype(np.float32) / np.float32(M)
    oxz, oind = oracle.gather(edge, (oi, od, ov), ptr=True, encode=table)
    xz, ind = sp.gather(edge, z, None, ptr=True, encode=z.slot_table())
    assert np.array_equal(ind.cpu().numpy(), oind) and np.array_equal(xz.cpu().numpy(), oxz)


@pytest.mark.parametrize("rng", ["rand_r", "philox"])
@pytest.mark.parametrize("M,m,idx64", [(200, 3, False), (200, 4, True), (120, 4, False), (100, 3, True), (256, 2, False)])
def test_specialised_fused_row_kernel_on_shuffled_roots(sp, rng, M, m, idx64):
    """csrc/walk_rows.hip: every instantiation family (2..4 hops x 512 / 1,024 table slots x int32 / int64 row offsets)
    on a graph whose roots all take the Fisher-Yates first hop (every degree > M), rows and numbering against the oracle"""
    ptr_, idx = sym_graph(4000, 700000, seed=23)
    assert int(np.diff(ptr_).min()) > M
    q = np.random.default_rng(7).permutation(4000)[:2500]
    from surel_plus_amd.sampler import DeviceCSR, walk_kernel_name
    assert walk_kernel_name(None, M, m, True) == "walk_rows_kernel"
    csr = DeviceCSR(ptr_.astype(np.int64) if idx64 else ptr_, idx)
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, m, 13, rng, -1)
    z, info = sp.sample_spg(csr, q, num_walks=M, num_steps=m, seed=13, rng=rng, fused=True)
    assert np.array_equal(z.indptr.cpu().numpy(), oi)
    assert np.array_equal(z.indices.cpu().numpy(), ox)
    assert np.array_equal(z.data.cpu().numpy(), od)
    assert np.array_equal(info.enc_int16().cpu().numpy(), oenc)


@pytest.mark.parametrize("rng", ["rand_r", "philox"])
@pytest.mark.parametrize("M,m,bits", [(200, 3, None), (100, 2, None), (200, 4, (24, 32)), (64, 4, (20, 36)), (200, 2, "i64"),
                                      (100, 3, "i64")])
def test_hop_records_give_the_same_rows(sp, rng, M, m, bits):
    """DeviceCSR.hop_records(): one packed 8-byte record per CSR entry (neighbour, its row begin, its degree) lets the
    fused-row kernel make one dependent read per hop; rows, sizes and numbering are what the plain CSR gives -- also when
    the degree field is so narrow (bits=: 8 bits left) that hubs take the escape path back to the row pointers"""
    ptr_, idx = sym_graph(6000, 60000, seed=29, hubs=4)            # hubs of degree ~1,500 next to degree-20 nodes
    q = np.concatenate([np.random.default_rng(3).permutation(6000)[:3000], [0, 1, 2, 3]])
    from surel_plus_amd.sampler import DeviceCSR
    wide = bits == "i64"                                           # int64 row offsets: the 16-byte form of the records
    bits = None if wide else bits
    ptr_w = ptr_.astype(np.int64) if wide else ptr_
    plain, recs = DeviceCSR(ptr_w, idx), DeviceCSR(ptr_w, idx)
    assert plain.hop_records(force=False) is None
    r = recs.hop_records(force=True, bits=bits)
    assert r is not None and r[0].numel() == recs.nnz * (2 if wide else 1) and recs.hop_records() is r
    assert (r[1] == 0) == wide
    za, ia = sp.sample_spg(plain, q, num_walks=M, num_steps=m, seed=17, rng=rng, fused=True)
    zb, ib = sp.sample_spg(recs, q, num_walks=M, num_steps=m, seed=17, rng=rng, fused=True)
    for a_, b_ in ((za.indptr, zb.indptr), (za.indices, zb.indices), (za.data, zb.data), (ia.enc_int16(), ib.enc_int16())):
        assert torch.equal(a_, b_)
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, m, 17, rng, -1)
    assert np.array_equal(zb.indptr.cpu().numpy(), oi) and np.array_equal(zb.indices.cpu().numpy(), ox)
    assert np.array_equal(zb.data.cpu().numpy(), od) and np.array_equal(ib.enc_int16().cpu().numpy(), oenc)


@pytest.mark.parametrize("M,m,wide", [(200, 3, False), (200, 2, True), (64, 4, False), (256, 3, False)])
def test_hop_records_on_a_graph_with_dead_ends(sp, M, m, wide):
    """a directed graph: walks reach nodes without out-edges and stay there -- also when the hop that would have fetched
    the last node's bare id finds nothing to fetch (a regression: the stale record was read as an id); Philox only
    (rand_r: tested above, through the replayed stream)"""
    import scipy.sparse as sps
    rng0 = np.random.default_rng(11)
    N = 3000
    r, c = rng0.integers(0, N, 9000), rng0.integers(0, N // 2, 9000)      # the upper half of the ids has no out-edges ... mostly
    A = sps.csr_matrix((np.ones(len(r)), (r, c)), shape=(N, N))
    A.sum_duplicates(); A.setdiag(0); A.eliminate_zeros(); A.sort_indices()
    ptr_ = A.indptr.astype(np.int64 if wide else np.int32)
    idx = A.indices.astype(np.int32)
    assert int((np.diff(A.indptr) == 0).sum()) > 100
    q = rng0.permutation(N)[:1500]
    from surel_plus_amd.sampler import DeviceCSR
    csr = DeviceCSR(ptr_, idx)
    assert csr.hop_records(force=True) is not None
    (oi, ox, od), oenc = _oracle_spg(A.indptr.astype(np.int32), idx, q, M, m, 19, "philox", -1)
    z, info = sp.sample_spg(csr, q, num_walks=M, num_steps=m, seed=19, rng="philox", fused=True)
    assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices.cpu().numpy(), ox)
    assert np.array_equal(z.data.cpu().numpy(), od) and np.array_equal(info.enc_int16().cpu().numpy(), oenc)
    zk, sk = sp.sample_spg(csr, q, num_walks=M, num_steps=m, seed=19, rng="philox", strided=True, number_rows=False)
    zc = zk.to_csr()
    assert np.array_equal(zc.indptr.cpu().numpy(), oi) and np.array_equal(zc.indices.cpu().numpy(), ox)


# ------------------------------------------------------------------ batched registration of key rows (csrc/keyrows.hip, ABI 4)
@pytest.mark.parametrize("rng", ["rand_r", "philox"])
@pytest.mark.parametrize("M,hops,N,E,hubs,idx64", [(200, 3, 9000, 90000, 3, False), (200, 2, 20000, 80000, 4, True),
                                                  (80, 3, 3000, 9000, 2, False), (255, 3, 1500, 200000, 0, False),
                                                  (120, 2, 400, 700, 1, False)])
def test_batched_registration_numbers_the_store_like_the_reference(sp, rng, M, hops, N, E, hubs, idx64):
    """The store that is kept (subg_matrix: main.py:172-178) sampled with the key-rows kernel, its LP rows registered by one
    pass over the rows and numbered after the candidate roots were walked again (subgacc_keyrows_register / subgacc_walk_tags /
    subgacc_keyrows_compact): bit for bit the oracle's (nsize, SpG, enc) -- i.e. the reference's first-occurrence numbering,
    subg_acc.c:957-978 -- and the table form's; one chunk, several chunks, sizes left on the device."""
    from surel_plus_amd import sampler
    from surel_plus_amd.spg import sample_spg
    assert sampler.key_rows_ok(M, hops)
    ptr_, idx = sym_graph(N, E, seed=M + hops, hubs=hubs)
    q = np.random.default_rng(11).permutation(N)
    csr = sp.DeviceCSR(ptr_.astype(np.int64) if idx64 else ptr_, idx)
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, hops, 77, rng, -1)
    stride = M * hops + 1
    for kw in ({}, {"staging_bytes": stride * 8 * (N // 7 + 1)}, {"lazy": True}):
        z, sets = sample_spg(csr, q, num_walks=M, num_steps=hops, seed=77, rng=rng, fused=True, **kw)
        sets.resolve()
        nnz = z.nnz
        assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices[:nnz].cpu().numpy(), ox)
        assert np.array_equal(z.data[:nnz].cpu().numpy(), od)
        assert np.array_equal(sets.enc_int16().cpu().numpy(), oenc)
    # the table form of the walk kernel (every root registers its own rows)
    zt, tsets = sample_spg(csr, q, num_walks=M, num_steps=hops, seed=77, rng=rng, fused=True, batched_registration=False)
    assert torch.equal(zt.indptr, z.indptr) and torch.equal(zt.data[: zt.nnz], z.data[:nnz]) and torch.equal(tsets.ukeys, sets.ukeys)


def test_batched_registration_regrows_a_small_table_and_numbers_transient_batches(sp):
    """(a) a table of distinct rows that is too small is grown and the job repeated, as for the table form; (b) a key-rows
    batch (StepBuffers) is numbered after the fact from the rows in its buffers -- and refused once the buffers moved on."""
    from surel_plus_amd.graphs import query_pairs
    from surel_plus_amd.spg import sample_spg
    ptr_, idx = sym_graph(6000, 60000, seed=8, hubs=2)
    csr = sp.DeviceCSR(ptr_, idx)
    q = np.arange(6000)
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, 200, 3, 5, "philox", -1)
    z, sets = sample_spg(csr, q, num_walks=200, num_steps=3, seed=5, rng="philox", fused=True, uniq_capacity=64)
    assert sets.capacity > 64 and np.array_equal(z.data[: z.nnz].cpu().numpy(), od) and np.array_equal(sets.enc_int16().cpu().numpy(), oenc)
    bufs = sp.StepBuffers(csr, 256, num_walks=200, num_steps=3)
    e1, e2 = query_pairs(csr, 256, seed=1), query_pairs(csr, 256, seed=2)
    xz, ind, s1 = sp.sample_and_gather(csr, e1, num_walks=200, num_steps=3, seed=5, rng="philox", buffers=bufs)
    s1.resolve()
    (_, _, _), oenc1 = _oracle_spg(ptr_, idx, e1.reshape(-1).cpu().numpy(), 200, 3, 5, "philox", -1)
    assert np.array_equal(s1.enc_int16().cpu().numpy(), oenc1)
    xz, ind, s2 = sp.sample_and_gather(csr, e2, num_walks=200, num_steps=3, seed=5, rng="philox", buffers=bufs)
    s2.resolve()
    s1.ukeys = None
    with pytest.raises(sp.SubgAccError):
        s1.number()
    (_, _, _), oenc2 = _oracle_spg(ptr_, idx, e2.reshape(-1).cpu().numpy(), 200, 3, 5, "philox", -1)
    assert np.array_equal(s2.enc_int16().cpu().numpy(), oenc2)


def test_shfl_fallback_of_the_wave_reductions_gives_the_same_rows():
    """csrc/waveops.hpp: the DPP wave reductions are internals of ROCm's device library; should an update rename them, the
    Makefile's probe builds the __shfl forms instead (-DSG_NO_OCKL_WAVE_OPS).  That build is made here (walk_rows.hip only,
    into /tmp) and the key-rows / fused-row parity tests are run through it in a child process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    try:
        from ab import build_variant
    finally:
        sys.path.pop(0)
    lib = build_variant("-DSG_NO_OCKL_WAVE_OPS", ["walk_rows.hip"])
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_sampler.py"),
                        os.path.join(root, "tests", "test_gpu_join.py"), "-x", "-q", "-k",
                        "key_rows_join_like or specialised_fused_row or batched_registration_numbers"],
                       env=dict(os.environ, SUBGACC_LIB=lib), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("M,hops", [(200, 3), (200, 2), (100, 4), (200, 4),
                                    # shapes that walk_rows_kernel does not take: the general fused kernel (walk.hip), whose epilogue
                                    # got the second sort level in round 4 -- more than 256 walks, and (below) a truncating bucket
                                    (300, 2), (260, 3)])
def test_sets_with_id_locality_sort_like_any_other(sp, M, hops):
    """A graph whose communities are blocks of consecutive ids (graphs.community_graph): most of a set lies inside one block, i.e.
    inside ONE bucket of a sort that buckets by equal id width.  The two-level distribution sort (walk_rows.hip key rows,
    spg.hip bucket_sort_regs) must give the same rows as ever -- every path against the oracle -- and take its fine level here."""
    from surel_plus_amd.graphs import community_graph, query_pairs
    from surel_plus_amd.spg import sample_spg
    csr = community_graph(30000, 20.7, seed=4, block=512, p_in=0.85)
    ptr_, idx = csr.indptr.cpu().numpy(), csr.indices.cpu().numpy()
    q = np.random.default_rng(1).permutation(30000)[:3000]
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, hops, 13, "philox", -1)
    assert int(np.diff(oi).max()) > 150                      # (sets big enough for a crowded bucket)
    # ... and with the table form of the fused walk kernel (every root registers its own rows: 32-bit counts for 2 and 3 hops,
    # 64-bit for 4), whose epilogue has the same two levels
    for batched in (True, False):
        for kw in ({"fused": True}, {"fused": False}, {"strided": True}, {"strided": True, "fused": False}):
            z, sets = sample_spg(csr, q, num_walks=M, num_steps=hops, seed=13, rng="philox", batched_registration=batched, **kw)
            if isinstance(z, sp.StridedSpG):
                z = z.to_csr()
            assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices[: z.nnz].cpu().numpy(), ox), (kw, batched)
            assert np.array_equal(z.data[: z.nnz].cpu().numpy(), od) and np.array_equal(sets.enc_int16().cpu().numpy(), oenc), (kw, batched)
    # a bucket that truncates (members ranked past it are dropped, subg_acc.c:814-828): walk_sets_kernel<SPG> in its ranking form,
    # crowded buckets included (the first `bucket` members of a set still lie in one community)
    bucket = 180
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, q, M, hops, 13, "philox", bucket)
    assert int(np.diff(oi).max()) == bucket
    z, sets = sample_spg(csr, q, num_walks=M, num_steps=hops, seed=13, rng="philox", bucket=bucket, fused=True)
    assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices[: z.nnz].cpu().numpy(), ox)
    assert np.array_equal(z.data[: z.nnz].cpu().numpy(), od) and np.array_equal(sets.enc_int16().cpu().numpy(), oenc)


@pytest.mark.parametrize("n,num_nodes", [(1, 5), (4097, 1000), (70000, 3_000_000), (20000, 1024)])
def test_worklist_by_root_lists_every_live_row_once_in_bucket_order(sp, n, num_nodes):
    """subgacc_worklist_by_root: a permutation of the rows whose root is not SUBGACC_NO_ROOT, ascending in (root >> shift) with
    1,024 buckets over [0, num_nodes); *n_work = the rows listed.  Checked through the C-ABI on its own."""
    from surel_plus_amd._lib import lib, check, ptr
    L = lib()
    g = np.random.default_rng(n)
    roots = g.integers(0, num_nodes, n).astype(np.int32)
    dead = g.random(n) < 0.2 if n > 1 else np.zeros(1, bool)
    roots[dead] = -2147483648
    r = torch.from_numpy(roots).cuda()
    wl = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    nw = torch.zeros(1, dtype=torch.int64, device="cuda")
    ws = torch.zeros(L.subgacc_worklist_workspace_bytes(n), dtype=torch.uint8, device="cuda")       # zeroed once by its owner
    for _ in range(3):       # (every call leaves the workspace ready for the next)
        wl.fill_(-7)
        check(L.subgacc_worklist_by_root(ptr(r), n, num_nodes, ptr(wl), ptr(nw), ptr(ws), ws.numel(), None))
    torch.cuda.synchronize()
    k = int(nw.item())
    assert int(ws[: 16 + 4096].view(torch.int32).abs().sum().item()) == 0
    assert k == int((~dead).sum())
    got = wl[:k].cpu().numpy()
    assert np.array_equal(np.sort(got), np.flatnonzero(~dead)) and (wl[k:] == -7).all()
    shift = 0
    while ((num_nodes - 1) >> shift) >= 1024:
        shift += 1
    assert (np.diff(roots[got] >> shift) >= 0).all()
