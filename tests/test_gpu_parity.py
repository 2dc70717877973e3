"""GPU (MI355X): the HIP path, called through the C ABI, against the golden vectors of the reference and
against the oracle on seeded inputs.  Integer / index outputs are compared BIT-EXACT; the only floating
point is the final table lookup (exact copies) and LP normalisation (float32 division, tolerance 0)."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, golden_files

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sp():
    import surel_plus_amd
    from surel_plus_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libsubgacc_hip.so must be built (no fallback)"
    assert _lib.lib().subgacc_device_count() >= 1, "no gfx950 device"
    return surel_plus_amd


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def sym_graph(N, E, seed, hubs=0):
    import scipy.sparse as sps
    rng = np.random.default_rng(seed)
    r = rng.integers(0, N, E)
    c = rng.integers(0, N, E)
    if hubs:
        hr = np.repeat(np.arange(hubs), N // 4)
        hc = rng.integers(0, N, hubs * (N // 4))
        r, c = np.concatenate([r, hr]), np.concatenate([c, hc])
    A = sps.csr_matrix((np.ones(len(r)), (r, c)), shape=(N, N))
    A = sps.csr_matrix(A + A.T)
    A.setdiag(0)
    A.eliminate_zeros()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


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


def dir_graph(N, E, seed, hubs=0):
    """a directed graph: the last third of the nodes has no out-edges at all"""
    import scipy.sparse as sps
    rng = np.random.default_rng(seed)
    r, c = rng.integers(0, (2 * N) // 3, E), rng.integers(0, N, E)
    if hubs:
        r = np.concatenate([r, np.repeat(np.arange(hubs), N // 3)])
        c = np.concatenate([c, rng.integers(0, N, hubs * (N // 3))])
    A = sps.csr_matrix((np.ones(len(r)), (r, c)), shape=(N, N))
    A.sum_duplicates(); A.setdiag(0); A.eliminate_zeros(); A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


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


# --------------------------------------------------------------------------------------- SpG
@pytest.mark.parametrize("name", golden_files("spg_"))
def test_spg_build_matches_scipy_golden(sp, name):
    g = _load(name)
    from surel_plus_amd.sampler import SampledSets
    dev = "cuda"
    nsize = torch.from_numpy(g["nsize"]).to(dev)
    row_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(nsize.long(), 0)])
    sets = SampledSets(nsize, row_off, torch.from_numpy(g["remap"][0]).to(dev), None,
                       torch.from_numpy(g["remap"][1]).to(dev),
                       torch.zeros(int(g["remap"][1].max()) + 1, dtype=torch.int64, device=dev), 1, 1,
                       int(g["nsize"].max()))
    z = sp.SpG.from_sets(sets)
    assert np.array_equal(z.indptr.cpu().numpy(), g["z_indptr"])
    assert np.array_equal(z.indices.cpu().numpy(), g["z_indices"])
    assert np.array_equal(z.data.cpu().numpy(), g["z_data"])


def test_subg_matrix_end_to_end(sp):
    g = _load("gset_collablike_s111413.npz")
    s = _load("spg_collablike_s111413.npz")

    class G:
        indptr, indices = g["indptr"], g["indices"]
    z, enc = sp.subg_matrix(G, g["query"], num_walks=int(g["M"]), num_steps=int(g["m"]) + 1, seed=111413)
    assert np.array_equal(z.indptr.cpu().numpy(), s["z_indptr"])
    assert np.array_equal(z.indices.cpu().numpy(), s["z_indices"])
    assert np.array_equal(z.data.cpu().numpy(), s["z_data"])
    assert enc.dtype == s["encz"].dtype and np.array_equal(enc, s["encz"])


# ------------------------------------------------------------------------------------ SpJoin
def _spg_from_golden(sp, g):
    data = g["z_data"]
    data = torch.from_numpy(data.astype(np.float64) if data.dtype.kind == "f" else data.astype(np.int32))
    return sp.SpG(torch.from_numpy(g["z_indptr"]).cuda(), torch.from_numpy(g["z_indices"]).cuda(), data.cuda())


@pytest.mark.parametrize("name", golden_files("sjoin_"))
@pytest.mark.parametrize("ptr", [True, False])
def test_gather_matches_reference_golden(sp, name, ptr):
    g = _load(name)
    z = _spg_from_golden(sp, g)
    enc = torch.from_numpy(g["encode"]).cuda() if g["encode"].size else None
    for fn in (sp.gather, lambda e, x, d, ptr, encode: sp.pgather(e, x, d, encode, sp.bgather, ptr=ptr)):
        xz, ind = fn(g["edge"], z, "cuda", ptr=ptr, encode=enc)
        assert xz.dtype == torch.float32 and ind.dtype == torch.int64 and xz.is_cuda and ind.is_cuda
        assert np.array_equal(xz.cpu().numpy(), g[f"xz_ptr{int(ptr)}"])
        assert np.array_equal(ind.cpu().numpy(), g[f"ind_ptr{int(ptr)}"])
    # torch edges (as train.py passes them) give the same answer
    xz2, _ = sp.gather(torch.from_numpy(g["edge"]), z, "cuda", ptr=ptr, encode=enc)
    assert np.array_equal(xz2.cpu().numpy(), g[f"xz_ptr{int(ptr)}"])


def test_hgather_matches_reference_golden(sp):
    g = _load("hjoin_int.npz")
    z = _spg_from_golden(sp, g)
    xz, ind = sp.hgather(g["hedge"], z, "cuda", encode=torch.from_numpy(g["encode"]).cuda())
    assert np.array_equal(xz.cpu().numpy(), g["xz"])
    assert np.array_equal(ind.cpu().numpy(), g["ind"])
    with pytest.raises(NotImplementedError):
        sp.hgather(g["hedge"], z, "cuda", encode=None)


def test_bgather_blocks(sp):
    g = _load("sjoin_int.npz")
    z = _spg_from_golden(sp, g)
    out = np.empty(4, dtype=object)
    sp.bgather(g["edge"], z, out)
    seg, pairs = oracle.sjoin(g["z_indptr"], g["z_indices"], g["z_data"], *oracle.pair_segments(g["edge"]))
    B = g["edge"].shape[1]
    assert np.array_equal(np.vstack([out[0], out[1]]), pairs)
    assert np.array_equal(np.concatenate([out[2], out[3]]), np.diff(seg))
    assert len(out[2]) == B


@pytest.mark.parametrize("payload", ["int", "float"])
def test_gather_matches_oracle_large(sp, payload):
    """SpG from a real sampling run (wide sets), 20k pairs incl. (u,u) pairs and repeated endpoints."""
    ptr_, idx = sym_graph(8000, 60000, seed=2, hubs=2)
    nsize, remap, enc = oracle.gset_sampler(ptr_, idx, np.arange(8000), num_walks=100, num_steps=3, seed=1, rng="philox",
                                            nthreads=8)
    zi, zx, zd = oracle.spg_build(nsize, remap)
    rng = np.random.default_rng(0)
    if payload == "float":
        zd = rng.random(len(zd)) * 0.9 + 0.1
        table = None
    else:
        table = oracle.enc_table(enc).astype(np.float32) / np.float32(100)
    edge = rng.integers(0, 8000, (2, 20000))
    edge[1, :50] = edge[0, :50]
    z = sp.SpG(torch.from_numpy(zi).cuda(), torch.from_numpy(zx).cuda(), torch.from_numpy(zd).cuda())
    enc_t = torch.from_numpy(table).cuda() if table is not None else None
    for ptr in (True, False):
        xz, ind = sp.gather(edge, z, "cuda", ptr=ptr, encode=enc_t)
        oxz, oind = oracle.gather(edge, (zi, zx, zd), ptr=ptr, encode=table, nthreads=8)
        assert np.array_equal(xz.cpu().numpy(), oxz)
        assert np.array_equal(ind.cpu().numpy(), oind)


def test_gather_properties_at_scale(sp):
    """Size-independent properties of SpJoin on a full sampling run: sizes, symmetry, self-join."""
    from surel_plus_amd.graphs import powerlaw_graph
    csr = powerlaw_graph(100_000, 8.2, seed=1)
    z, enc = sp.subg_matrix(csr, torch.arange(100_000, dtype=torch.int32), num_walks=200, num_steps=3, rng="philox")
    table = torch.from_numpy(enc.astype(np.float32)).cuda() / 200
    B = 65536
    gen = torch.Generator(device="cuda").manual_seed(0)
    edge = torch.randint(0, 100_000, (2, B), device="cuda", generator=gen)
    xz, ind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    lens = z.indptr[1:] - z.indptr[:-1]
    assert torch.equal(ind[1:] - ind[:-1], torch.cat([lens[edge[0]], lens[edge[1]]]))
    assert xz.shape == (int(ind[-1]), 2, 3)
    # swapping the endpoints swaps the two halves of the output
    xz_s, ind_s = sp.gather(edge.flip(0), z, "cuda", ptr=True, encode=table)
    mid = int(ind[B])
    assert torch.equal(xz_s[: xz.shape[0] - mid], xz[mid:]) and torch.equal(xz_s[xz.shape[0] - mid:], xz[:mid])
    # joining a node with itself: both slots equal; first slot is the node's own feature row
    uu = torch.stack([edge[0], edge[0]])
    xz_u, _ = sp.gather(uu, z, "cuda", ptr=True, encode=table)
    assert torch.equal(xz_u[:, 0], xz_u[:, 1])
    # the number of rows with a non-zero second slot is the same on both sides (|S_u & S_v|); root rows have
    # feature[0] = 1 so a present partner is never the zero row
    nz = (xz[:, 1, :].abs().sum(-1) > 0)
    segid = torch.repeat_interleave(torch.arange(2 * B, device="cuda"), ind[1:] - ind[:-1])
    per_seg = torch.zeros(2 * B, dtype=torch.int64, device="cuda").index_add_(0, segid, nz.long())
    assert torch.equal(per_seg[:B], per_seg[B:])


def test_unique_table_grows_on_overflow(sp):
    """A deliberately tiny unique-row table (64 slots) must be detected as over-full and retried larger."""
    ptr_, idx = sym_graph(3000, 9000, seed=104, hubs=2)
    from surel_plus_amd.sampler import DeviceCSR, sample_sets
    s = sample_sets(DeviceCSR(ptr_, idx), np.arange(3000), num_walks=100, num_steps=4, seed=99, uniq_capacity=64)
    b = oracle.gset_sampler(ptr_, idx, np.arange(3000), num_walks=100, num_steps=4, seed=99)
    assert s.c > 64
    assert np.array_equal(torch.stack([s.ids, s.get_sf()]).cpu().numpy(), b[1])
    assert np.array_equal(s.enc_int16().cpu().numpy(), b[2])


def test_encode_table_too_small_raises(sp):
    g = _load("sjoin_int.npz")
    z = _spg_from_golden(sp, g)
    small = torch.from_numpy(g["encode"][:5]).cuda()
    with pytest.raises(IndexError):
        sp.gather(g["edge"], z, "cuda", ptr=True, encode=small)


@pytest.mark.parametrize("k", [1, 3, 4, 5, 8])
def test_gather_feature_widths(sp, k):
    """every feature width goes through the coalesced store path (k = 4 uses the float4 specialisation)."""
    g = _load("sjoin_int.npz")
    z = _spg_from_golden(sp, g)
    rng = np.random.default_rng(k)
    table = rng.random((int(g["z_data"].max()) + 1, k)).astype(np.float32)
    table[0] = 0
    xz, ind = sp.gather(g["edge"], z, "cuda", ptr=True, encode=torch.from_numpy(table).cuda())
    oxz, oind = oracle.gather(g["edge"], (g["z_indptr"], g["z_indices"], g["z_data"]), ptr=True, encode=table)
    assert np.array_equal(xz.cpu().numpy(), oxz) and np.array_equal(ind.cpu().numpy(), oind)


def test_spg_build_long_rows_use_the_bitonic_fallback(sp):
    """a row bound above 4096 members leaves the bucket-sort kernel for the bitonic network."""
    ptr_, idx = sym_graph(4000, 400000, seed=77)
    q = np.arange(300)
    nsize, remap, enc = oracle.gset_sampler(ptr_, idx, q, num_walks=300, num_steps=4, seed=3, rng="philox", nthreads=8)
    from surel_plus_amd.sampler import DeviceCSR, sample_sets
    s = sample_sets(DeviceCSR(ptr_, idx), q, num_walks=300, num_steps=4, seed=3, rng="philox")
    assert np.array_equal(s.nsize.cpu().numpy(), nsize)
    s.stride = 5000                                   # claim rows of up to 5000 members -> bitonic path (bucket sort stops at 1024)
    z = sp.SpG.from_sets(s)
    oi, ox, od = oracle.spg_build(nsize, remap)
    assert np.array_equal(z.indptr.cpu().numpy(), oi) and np.array_equal(z.indices.cpu().numpy(), ox)
    assert np.array_equal(z.data.cpu().numpy(), od)


def test_paired_and_generic_join_kernels_agree(sp):
    """gather() uses the pair-fused kernel; the generic one-segment-per-wave kernel must give the same bytes."""
    from surel_plus_amd.graphs import powerlaw_graph
    from surel_plus_amd.spjoin import sjoin
    csr = powerlaw_graph(30_000, 12.0, seed=4)
    z, enc = sp.subg_matrix(csr, torch.arange(30_000, dtype=torch.int32), num_walks=150, num_steps=4, rng="philox")
    table = torch.from_numpy(enc.astype(np.float32)).cuda() / 150
    gen = torch.Generator(device="cuda").manual_seed(1)
    edge = torch.randint(0, 30_000, (2, 5000), device="cuda", generator=gen)
    edge[1, :20] = edge[0, :20]
    own, partner = torch.cat([edge[0], edge[1]]), torch.cat([edge[1], edge[0]])
    for ptr_mode in (True, False):
        a = sjoin(z, own, partner, table, ptr_mode=ptr_mode, pair_block=0)
        b = sjoin(z, own, partner, table, ptr_mode=ptr_mode, pair_block=5000)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert int(a[2][3]) == 0 and int(b[2][3]) == 0
    ia = sjoin(z, own, partner, None, return_index=True, pair_block=0)
    ib = sjoin(z, own, partner, None, return_index=True, pair_block=5000)
    assert torch.equal(ia[0], ib[0])
    # a list that is not mirrored is refused (flag 4), never silently mis-joined
    bad = sjoin(z, own, own, table, pair_block=5000)
    assert int(bad[2][3]) & 4


@pytest.mark.parametrize("k", [3, 4])
def test_gather_with_a_table_too_large_for_lds(sp, k):
    """> 32 KB of Z_SF rows: the join reads the table from HBM/L2 instead of its LDS copy."""
    g = _load("sjoin_int.npz")
    rng = np.random.default_rng(5)
    data = rng.integers(1, 20001, g["z_data"].shape).astype(np.int32)
    table = rng.random((20001, k)).astype(np.float32)
    table[0] = 0
    z = sp.SpG(torch.from_numpy(g["z_indptr"]).cuda(), torch.from_numpy(g["z_indices"]).cuda(),
               torch.from_numpy(data).cuda())
    xz, ind = sp.gather(g["edge"], z, "cuda", ptr=True, encode=torch.from_numpy(table).cuda())
    oxz, oind = oracle.gather(g["edge"], (g["z_indptr"], g["z_indices"], data), ptr=True, encode=table)
    assert np.array_equal(xz.cpu().numpy(), oxz) and np.array_equal(ind.cpu().numpy(), oind)


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


# ------------------------------------------------------------------------------------ edge cases
def test_empty_query_and_empty_join(sp):
    ptr_, idx = sym_graph(100, 300, seed=1)
    out = sp.gset_sampler(ptr_, idx, np.zeros(0, np.int64), num_walks=8, num_steps=2)
    ref = oracle.gset_sampler(ptr_, idx, np.zeros(0, np.int64), num_walks=8, num_steps=2)
    assert out[0].shape == (0,) and out[1].shape == (2, 0) and out[2].shape == (0, 3)
    assert all(a.shape == b.shape and a.dtype == b.dtype for a, b in zip(out, ref))
    g = _load("sjoin_int.npz")
    z = _spg_from_golden(sp, g)
    enc = torch.from_numpy(g["encode"]).cuda()
    for ptr in (True, False):
        xz, ind = sp.gather(np.zeros((2, 0), np.int64), z, "cuda", ptr=ptr, encode=enc)
        assert xz.shape == (0, 2, enc.shape[1]) and ind.tolist() == ([0] if ptr else [])


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


# ------------------------------------------------------------------------ fused SpG pipeline (walk_spg)
def _oracle_spg(ptr_, idx, q, M, m, seed, rng, bucket=-1):
    nsize, remap, enc = oracle.gset_sampler(ptr_, idx, q, num_walks=M, num_steps=m, bucket=bucket, seed=seed, rng=rng,
                                            nthreads=8 if rng == "philox" else 1)
    return oracle.spg_build(nsize, remap), enc


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


# ----------------------------------------------------------------- count form of the join (next row f.1)
def _oracle_counts(spg, edge, rows):
    own, partner = oracle.pair_segments(edge)
    seg, pairs = oracle.sjoin(spg[0], spg[1], spg[2], own, partner)
    C = np.zeros((len(own), rows), np.float32)
    segid = np.repeat(np.arange(len(own)), np.diff(seg))
    np.add.at(C, (segid, pairs[:, 0]), 1)
    np.add.at(C, (segid, pairs[:, 1]), 1)
    return C, np.diff(seg)


def test_gather_counts_matches_oracle_and_the_reference_first_stage(sp):
    """C[j,p] counts are exact; C @ MLP(Z_SF) equals the reference's pe_embedding(xz).sum(-2) summed per segment
    (model.py:78-83) up to fp32 summation order (tolerance 2e-5 relative, stated here)."""
    g = _load("sjoin_int_emptyrows.npz")
    z = _spg_from_golden(sp, g)
    rows = g["encode"].shape[0]
    C, sizes = sp.gather_counts(g["edge"], z, rows)
    oC, osz = _oracle_counts((g["z_indptr"], g["z_indices"], g["z_data"]), g["edge"], rows)
    assert np.array_equal(C.cpu().numpy(), oC) and np.array_equal(sizes.cpu().numpy(), osz)
    assert np.array_equal(C.sum(1).cpu().numpy(), 2 * osz)                   # two slots per output row
    # larger: a real sampling run, plus the model-side identity
    ptr_, idx = sym_graph(5000, 30000, seed=3, hubs=1)
    from surel_plus_amd.sampler import DeviceCSR
    zz, sets = sp.sample_spg(DeviceCSR(ptr_, idx), np.arange(5000), num_walks=100, num_steps=3, seed=1, rng="philox")
    table = sets.feature_table()
    edge = np.random.default_rng(0).integers(0, 5000, (2, 4000))
    C, sizes = sp.gather_counts(edge, zz, table.shape[0])
    spg_h = (zz.indptr.cpu().numpy(), zz.indices.cpu().numpy(), zz.data.cpu().numpy())
    oC, osz = _oracle_counts(spg_h, edge, table.shape[0])
    assert np.array_equal(C.cpu().numpy(), oC) and np.array_equal(sizes.cpu().numpy(), osz)
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(table.shape[1], 32), torch.nn.ReLU(), torch.nn.Linear(32, 32)).cuda()
    xz, ind = sp.gather(edge, zz, "cuda", ptr=True, encode=table)
    ref = torch.zeros(2 * 4000, 32, device="cuda").index_add_(
        0, torch.repeat_interleave(torch.arange(2 * 4000, device="cuda"), ind[1:] - ind[:-1]), mlp(xz).sum(dim=-2))
    fused = C @ mlp(table)
    assert torch.allclose(fused, ref, rtol=2e-5, atol=2e-4)
    with pytest.raises(IndexError):
        sp.gather_counts(edge, zz, 3)


def test_mean_stage_trains_like_the_reference_first_stage(sp):
    """forward and parameter gradients of the fused stage equal pe_embedding(xz).sum(-2) + mean aggregation
    (model.py:78-83) up to fp32 summation order (forward rtol 1e-4; gradients 1e-4 of their largest entry, stated here)."""
    ptr_, idx = sym_graph(3000, 15000, seed=6, hubs=1)
    from surel_plus_amd.sampler import DeviceCSR
    z, sets = sp.sample_spg(DeviceCSR(ptr_, idx), np.arange(3000), num_walks=64, num_steps=3, seed=2, rng="philox")
    table = sets.feature_table()
    edge = torch.from_numpy(np.random.default_rng(4).integers(0, 3000, (2, 512))).cuda()
    torch.manual_seed(1)
    mlp_a = torch.nn.Sequential(torch.nn.Linear(4, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16)).cuda()
    mlp_b = torch.nn.Sequential(torch.nn.Linear(4, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16)).cuda()
    mlp_b.load_state_dict(mlp_a.state_dict())
    w = torch.randn(2, 512, 16, device="cuda")
    fused = sp.mean_stage(edge, z, table, mlp_a)
    (fused * w).sum().backward()
    xz, ind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    x = mlp_b(xz).sum(dim=-2)
    seg = torch.repeat_interleave(torch.arange(1024, device="cuda"), ind[1:] - ind[:-1])
    ref = (torch.zeros(1024, 16, device="cuda").index_add_(0, seg, x) / (ind[1:] - ind[:-1]).clamp(min=1)[:, None]).view(2, 512, 16)
    (ref * w).sum().backward()
    assert torch.allclose(fused, ref, rtol=1e-4, atol=1e-5)
    for pa, pb in zip(mlp_a.parameters(), mlp_b.parameters()):      # gradients are sums over ~2e5 rows in fp32
        assert float((pa.grad - pb.grad).abs().max()) <= 1e-4 * float(pb.grad.abs().max()) + 1e-6


def test_spg_save_and_load(sp, tmp_path):
    ptr_, idx = sym_graph(1000, 5000, seed=2)
    from surel_plus_amd.sampler import DeviceCSR
    z, sets = sp.sample_spg(DeviceCSR(ptr_, idx), np.arange(1000), num_walks=32, num_steps=3, seed=2, lazy=True)
    table = sets.feature_table()
    path = str(tmp_path / "spg.pt")
    z.save(path, encode=table)
    z2, enc2 = sp.SpG.load(path)
    assert z2.nnz == z.nnz and z2.indices.numel() == z.nnz                       # trimmed to the real size
    edge = np.random.default_rng(0).integers(0, 1000, (2, 200))
    a = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    b = sp.gather(edge, z2, "cuda", ptr=True, encode=enc2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_gather_lazy_rows_match_eager(sp):
    """gather(out=, lazy=True): no host round trip; the first ind[-1] rows of the buffer are the eager result"""
    indptr, indices = sym_graph(4000, 16000, 21, hubs=2)
    csr = sp.DeviceCSR(indptr, indices)
    roots = torch.arange(0, 512, dtype=torch.int32, device="cuda")
    z, sets = sp.sample_spg(csr, roots, num_walks=50, num_steps=3, rng="philox", lazy=True)
    table = sets.feature_table()
    edge = torch.randint(0, 512, (2, 300), device="cuda")
    k = table.shape[1]
    buf = torch.full((2 * 300 * z.max_len * 2 * k,), -7.0, dtype=torch.float32, device="cuda")
    xz_l, ind_l = sp.gather(edge, z, "cuda", ptr=True, encode=table, out=buf, lazy=True)
    xz_e, ind_e = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    assert torch.equal(ind_l, ind_e)
    R = int(ind_l[-1])
    assert xz_e.shape[0] == R and torch.equal(xz_l[:R], xz_e)
    assert bool((xz_l[R:] == -7.0).all())          # nothing written past the valid rows
    with pytest.raises(ValueError):
        sp.gather(edge, z, "cuda", ptr=True, encode=table, lazy=True)                       # needs out=
    with pytest.raises(ValueError):
        sp.gather(edge, z, "cuda", ptr=True, encode=table, out=buf[:100], lazy=True)        # worst case must fit


# ------------------------------------------------------------------------------- rw_matrix / np_sampling (SUREL route)
@pytest.mark.parametrize("reduced", [True, False])
@pytest.mark.parametrize("nthread,bsize", [(1, 2000), (4, 300)])
def test_rw_matrix_matches_restatement(sp, reduced, nthread, bsize):
    """sampler/random_walks.py:58-71 through walk_sampler; numbering of LP rows = ascending projection order"""
    indptr, indices = sym_graph(1500, 6000, 31, hubs=2)
    idx = np.arange(1500)
    ref = oracle.ref_module()
    sampler = ref.walk_sampler if ref is not None else None          # the real reference when oracle/_ref exists
    z_o, f_o = oracle.rw_matrix(indptr, indices, idx, num_walks=40, num_steps=4, batch_size=bsize, reduced=reduced,
                                nthread=nthread, sampler=sampler)
    z, f = sp.rw_matrix(sp.DeviceCSR(indptr, indices), idx, num_walks=40, num_steps=4, batch_size=bsize, reduced=reduced,
                        nthread=nthread)
    z_o.sort_indices()
    np.testing.assert_array_equal(f, f_o)
    assert f.dtype == f_o.dtype
    np.testing.assert_array_equal(z.indptr.cpu().numpy(), z_o.indptr)
    np.testing.assert_array_equal(z.indices.cpu().numpy(), z_o.indices)
    np.testing.assert_array_equal(z.data.cpu().numpy(), z_o.data)
    k_o, c_o = oracle.np_sampling(indptr, indices, bsize, idx[:700], num_walks=40, num_steps=3, nthread=nthread, sampler=sampler)
    k, c = sp.np_sampling(indptr, indices, bsize, idx[:700], num_walks=40, num_steps=3, nthread=nthread)
    np.testing.assert_array_equal(k, k_o)
    np.testing.assert_array_equal(c, c_o)


# ------------------------------------------------------------------------------- walk_join (legacy SUREL join)
def _walkjoin_inputs(g):
    off = np.concatenate([[0], np.cumsum(g["key_len"])])
    return g["walks"], [g["key_ids"][off[i]:off[i + 1]] for i in range(len(g["key_len"]))], g["query"]


@pytest.mark.parametrize("name", golden_files("walkjoin_"))
def test_walk_join_matches_reference_golden(sp, name):
    g = _load(name)
    walks, key, query = _walkjoin_inputs(g)
    out, xrow = sp.walk_join(walks, key, query, return_idx=True)
    np.testing.assert_array_equal(xrow, g["xrow"])
    np.testing.assert_array_equal(out, g["out"])
    assert out.dtype == np.int32 and out.shape == g["out"].shape
    out3 = sp.walk_join(walks.reshape(walks.shape[0], -1, 1), key, query)          # 3-D walks, no index request
    np.testing.assert_array_equal(out3, g["out"])


def test_walk_join_end_to_end_vs_oracle(sp):
    """walk_sampler -> walk_join on the GPU against the oracle, a batch of 3000 roots and 4096 pairs"""
    indptr, indices = sym_graph(5000, 25000, 41, hubs=3)
    rng = np.random.default_rng(3)
    roots = rng.permutation(5000)[:3000].astype(np.int32)
    walks, obj = sp.walk_sampler(indptr, indices, roots, num_walks=50, num_steps=3, nthread=2, seed=5, replacement=True)
    q = roots[rng.integers(0, 3000, (4096, 2))]
    q[7] = (4999 if 4999 not in roots else roots[0], roots[1])
    out, xrow = sp.walk_join(walks, list(obj[:, 0]), q, return_idx=True)
    want, wrow = oracle.walk_join(walks, list(obj[:, 0]), q, return_idx=True)
    np.testing.assert_array_equal(xrow, wrow)
    np.testing.assert_array_equal(out, want)
    with pytest.raises(AssertionError):
        sp.walk_join(walks, list(obj[:-1, 0]), q)
    assert sp.walk_join(walks, list(obj[:, 0]), np.zeros((0, 2), np.int32)).shape == (2, 0)


# ------------------------------------------------------------------------------- SpJoin over rows longer than LDS
@pytest.mark.parametrize("payload", ["int", "float"])
def test_gather_rows_longer_than_lds(sp, payload):
    """an adjacency-like SpG with hub rows of 30k / 12k members (LDS holds ~10k int / ~6.8k float entries of a pair):
    one row fits -> generic staged kernel; none fits -> rows are searched in place.  Same answers as NumPy."""
    rng = np.random.default_rng(11)
    N = 40000
    lens = rng.integers(0, 40, N)
    lens[5], lens[17], lens[300] = 30000, 12000, 9000
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(N, int(n_), replace=False)) for n_ in lens]).astype(np.int32)
    if payload == "int":
        data, enc = rng.integers(1, 50, indices.size).astype(np.int32), rng.random((50, 3)).astype(np.float32)
        enc[0] = 0
    else:
        data, enc = rng.random(indices.size), None
    z = sp.SpG(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), torch.from_numpy(data).cuda())
    edge = rng.integers(0, N, (2, 64))
    edge[:, 0], edge[:, 1], edge[:, 2], edge[:, 3] = (5, 17), (17, 5), (5, 5), (300, 17)
    enc_d = torch.from_numpy(enc).cuda() if enc is not None else None
    for ptr in (True, False):
        want_xz, want_ind = oracle.gather_numpy(edge, (indptr, indices, data), ptr=ptr, encode=enc)
        for fn in (sp.gather, lambda e, x, d, ptr, encode: sp.pgather(e, x, d, encode, sp.bgather, ptr=ptr)):
            xz, ind = fn(edge, z, "cuda", ptr=ptr, encode=enc_d)
            np.testing.assert_array_equal(ind.cpu().numpy(), want_ind)
            np.testing.assert_array_equal(xz.cpu().numpy(), want_xz)


# ------------------------------------------------------------------------------- join straight from strided rows
@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("M,m", [(50, 3), (200, 2), (16, 4)])
def test_strided_spg_join_matches_csr_path(sp, lazy, M, m):
    """sample_spg(strided=True) + gather / hgather == the packed-SpG path, bit for bit; to_csr() == the packed SpG"""
    indptr, indices = sym_graph(6000, 30000, 51, hubs=2)
    csr = sp.DeviceCSR(indptr, indices)
    rng = np.random.default_rng(1)
    roots = torch.from_numpy(rng.integers(0, 6000, 700).astype(np.int32)).cuda()       # repeated roots included
    zs, sets_s = sp.sample_spg(csr, roots, num_walks=M, num_steps=m, rng="philox", lazy=lazy, strided=True)
    zc, sets_c = sp.sample_spg(csr, roots, num_walks=M, num_steps=m, rng="philox", fused=True)
    assert isinstance(zs, sp.StridedSpG)
    table = sets_c.feature_table()
    assert torch.equal(sets_s.feature_table()[: table.shape[0]], table)
    z2 = zs.to_csr()
    assert torch.equal(z2.indptr, zc.indptr) and torch.equal(z2.indices, zc.indices) and torch.equal(z2.data, zc.data)
    assert sets_s.X == sets_c.X and sets_s.c == sets_c.c and zs.nnz == zc.nnz
    edge = torch.from_numpy(rng.integers(0, 700, (2, 400))).cuda()
    edge[:, 0] = 5                                                                        # (u, u)
    xz_c, ind_c = sp.gather(edge, zc, "cuda", ptr=True, encode=table)
    xz_s, ind_s = sp.gather(edge, zs, "cuda", ptr=True, encode=sets_s.feature_table())
    assert torch.equal(ind_s, ind_c) and torch.equal(xz_s, xz_c)
    buf = torch.empty(2 * 400 * zs.max_len * 2 * table.shape[1], dtype=torch.float32, device="cuda")
    xz_l, ind_l = sp.gather(edge, zs, "cuda", ptr=True, encode=sets_s.feature_table(), out=buf, lazy=True)
    assert torch.equal(ind_l, ind_c) and torch.equal(xz_l[: xz_c.shape[0]], xz_c)
    xz_t, ind_t = sp.gather(edge, zs, "cuda", ptr=True, encode=zs.slot_table())        # table indexed by slot: no numbering used
    assert torch.equal(ind_t, ind_c) and torch.equal(xz_t, xz_c)
    hedge = torch.from_numpy(rng.integers(0, 700, (3, 100))).cuda()
    hx_c, hid_c = sp.hgather(hedge, zc, "cuda", encode=table)
    for zz, tb in ((zs.to_csr(), table), (zs, sets_s.feature_table()), (zs, zs.slot_table())):
        hx_s, hid_s = sp.hgather(hedge, zz, "cuda", encode=tb)
        assert torch.equal(hx_s, hx_c) and torch.equal(hid_s, hid_c)
    xz_i, ind_i = sp.gather(edge, zs, "cuda", ptr=False, encode=sets_s.feature_table())     # segment ids
    xz_ci, ind_ci = sp.gather(edge, zc, "cuda", ptr=False, encode=table)
    assert torch.equal(xz_i, xz_ci) and torch.equal(ind_i, ind_ci)
    cnt_c, sz_c = sp.gather_counts(edge, zc, table.shape[0])
    cnt_s, sz_s = sp.gather_counts(edge, zs.to_csr(), table.shape[0])
    assert torch.equal(cnt_s, cnt_c) and torch.equal(sz_s, sz_c)
    with pytest.raises(ValueError):
        sp.sjoin(zs, edge[0].contiguous(), edge[1].contiguous(), table)                # unpaired lists: packed form only


def test_strided_falls_back_when_it_does_not_apply(sp):
    indptr, indices = sym_graph(2000, 8000, 52)
    csr = sp.DeviceCSR(indptr, indices)
    roots = torch.arange(100, dtype=torch.int32, device="cuda")
    z, sets = sp.sample_spg(csr, roots, num_walks=300, num_steps=4, rng="philox", strided=True)   # M*m+1 = 1201 > 818
    assert isinstance(z, sp.SpG) and not sets.strided


@pytest.mark.parametrize("hops", [2, 3])
def test_sample_and_gather_on_demand_and_root_dedup(sp, hops):
    """one call = sample both endpoints + join; an evaluation-style batch (each source against many candidates,
    utils.py:92-95) gives the same (xz, indptr) whether every endpoint is sampled or every DISTINCT endpoint once"""
    indptr, indices = sym_graph(8000, 40000, 61, hubs=2)
    csr = sp.DeviceCSR(indptr, indices)
    rng = np.random.default_rng(2)
    src = np.repeat(rng.integers(0, 8000, 16), 101)
    dst = rng.integers(0, 8000, src.size)
    edge = torch.from_numpy(np.stack([src, dst])).cuda()
    xz_a, ind_a, sets_a = sp.sample_and_gather(csr, edge, num_walks=64, num_steps=hops, seed=5)
    xz_b, ind_b, sets_b = sp.sample_and_gather(csr, edge, num_walks=64, num_steps=hops, seed=5, dedup_roots=True)
    assert sets_b.nsize.numel() < sets_a.nsize.numel() // 2 + 20            # sources collapse to 16 roots
    assert torch.equal(ind_a, ind_b) and torch.equal(xz_a, xz_b)
    # against the offline route: SpG of all nodes, then gather by node id
    z, setsz = sp.sample_spg(csr, torch.arange(8000, dtype=torch.int32, device="cuda"), num_walks=64, num_steps=hops, seed=5,
                             rng="philox")
    xz_c, ind_c = sp.gather(edge, z, "cuda", ptr=True, encode=setsz.feature_table())
    assert torch.equal(ind_a, ind_c) and torch.equal(xz_a, xz_c)
    with pytest.raises(ValueError):
        sp.sample_and_gather(csr, edge, num_walks=8, num_steps=2, rng="rand_r", dedup_roots=True)


def test_out_of_range_rows_and_roots_raise_instead_of_reading_out_of_bounds(sp):
    """The reference indexes with whatever it is given (scipy raises IndexError for a bad row, train.py:15; the C sampler
    reads out of bounds for a bad root).  Here a bad row / root is never dereferenced on the device and the
    host mirror raises IndexError -- for host AND device inputs."""
    from surel_plus_amd.sampler import sample_sets
    from surel_plus_amd.spg import sample_spg
    ptr_, idx = sym_graph(500, 2500, seed=2)
    csr = sp.DeviceCSR(ptr_, idx)
    z, sets = sample_spg(csr, np.arange(500), num_walks=16, num_steps=2, seed=1, rng="philox")
    table = sets.feature_table()
    good = torch.tensor([[1, 2, 3], [4, 5, 6]], device="cuda")
    ref_xz, ref_ind = sp.gather(good, z, None, ptr=True, encode=table)
    for bad_val in (500, -1, 1 << 40):
        bad = good.clone()
        bad[1, 1] = bad_val
        with pytest.raises(IndexError):
            sp.gather(bad, z, None, ptr=True, encode=table)
        with pytest.raises(IndexError):
            sp.gather(bad.cpu().numpy(), z, None, ptr=False, encode=table)
        with pytest.raises(IndexError):
            sp.hgather(torch.cat([bad, good[:1]]), z, None, encode=table)
        with pytest.raises(IndexError):
            sp.gather_counts(bad, z, table.shape[0])
        with pytest.raises(IndexError):
            sp.bgather(bad, z, [None] * 4)
    # strided rows of a transient batch
    zs, ssets = sample_spg(csr, np.arange(100), num_walks=16, num_steps=3, seed=1, rng="philox", strided=True)
    assert ssets.strided
    with pytest.raises(IndexError):
        sp.gather(torch.tensor([[1, 100], [2, 3]], device="cuda"), zs, None, ptr=True, encode=zs.slot_table())
    # the device is fine afterwards and the good batch still gives the same answer
    xz, ind = sp.gather(good, z, None, ptr=True, encode=table)
    assert torch.equal(xz, ref_xz) and torch.equal(ind, ref_ind)
    # roots: device tensors are not pre-checked on the host; the kernels flag them
    for q in (torch.tensor([3, 500, 7], device="cuda"), torch.tensor([3, -2, 7], device="cuda", dtype=torch.int32)):
        for kw in ({}, {"rng": "philox"}, {"rng": "philox", "fused_rows": True}):
            with pytest.raises(IndexError):
                sample_sets(csr, q, num_walks=16, num_steps=3, seed=1, **kw)
        lazy = sample_sets(csr, q, num_walks=16, num_steps=3, seed=1, rng="philox", fused_rows=True, lazy=True)
        with pytest.raises(IndexError):
            lazy.resolve()
    with pytest.raises(IndexError):
        sp.gset_sampler(ptr_, idx, np.array([1, 500]), num_walks=8, num_steps=2)
    with pytest.raises(IndexError):
        sp.sample_and_gather(csr, torch.tensor([[1, 2], [3, 777]], device="cuda"), num_walks=16, num_steps=3)
    # a malformed CSR handed over as device tensors is refused before any kernel sees it
    ip, ix = torch.from_numpy(ptr_).cuda(), torch.from_numpy(idx).cuda()
    ix_bad = ix.clone()
    ix_bad[5] = 500
    with pytest.raises(IndexError):
        sp.DeviceCSR(ip, ix_bad)
    ip_bad = ip.clone()
    ip_bad[10] = ip_bad[9] - 1
    with pytest.raises(IndexError):
        sp.DeviceCSR(ip_bad, ix)
    a = sp.gset_sampler(ptr_, idx, np.arange(50), num_walks=8, num_steps=2)
    b = oracle.gset_sampler(ptr_, idx, np.arange(50), num_walks=8, num_steps=2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_strided_eager_recovers_from_a_small_table_of_distinct_rows(sp):
    """The eager strided form regrows an overflowing table of distinct LP rows like the packed forms do (and the retry
    keeps the strided layout); more distinct rows than the direct ranking numbers -> None, sample_spg falls through."""
    from surel_plus_amd.sampler import sample_sets
    from surel_plus_amd.spg import StridedSpG, sample_spg
    ptr_, idx = sym_graph(4000, 40000, seed=12, hubs=2)
    csr = sp.DeviceCSR(ptr_, idx)
    q = np.arange(4000)
    ref = sample_sets(csr, q, num_walks=64, num_steps=3, seed=5, rng="philox", fused_rows=True)
    assert ref.c > 64
    s = sample_sets(csr, q, num_walks=64, num_steps=3, seed=5, rng="philox", fused_rows=True, strided=True, uniq_capacity=64)
    assert s is not None and s.strided and s.capacity > 64 and s.c == ref.c
    assert torch.equal(s.ukeys, ref.ukeys) and torch.equal(s.nsize, ref.nsize)
    assert sample_sets(csr, q, num_walks=64, num_steps=3, seed=5, rng="philox", fused_rows=True, strided=True,
                       uniq_small_limit=32) is None
    z, sets = sample_spg(csr, q, num_walks=64, num_steps=3, seed=5, rng="philox", strided=True, uniq_small_limit=32)
    assert not isinstance(z, StridedSpG) and sets.c == ref.c


# ------------------------------------------------------- pair form of the join + the attention first stage (row f.1)
def _reference_style_attn(xz, ind, mlp, gate, val):
    """model.py:78-81 with AttentionalAggregation written out (torch_geometric is not in the image): softmax of the gate
    over each segment (torch_geometric.utils.softmax: exp(x - max) / (sum + 1e-16)), weighted sum of nn(x)."""
    S = ind.numel() - 1
    x = mlp(xz).sum(dim=-2)
    seg = torch.repeat_interleave(torch.arange(S, device=xz.device), ind[1:] - ind[:-1])
    g = gate(x).reshape(-1)
    gmax = torch.full((S,), float("-inf"), device=g.device, dtype=g.dtype).scatter_reduce(0, seg, g.detach(), "amax")
    w = torch.exp(g - gmax[seg])
    den = torch.zeros(S, device=g.device, dtype=g.dtype).index_add_(0, seg, w)
    alpha = w / (den[seg] + 1e-16)
    return torch.zeros((S, x.shape[-1]), device=g.device, dtype=g.dtype).index_add_(0, seg, alpha[:, None] * val(x))


def test_gather_pairs_is_the_multiset_of_gather_rows_and_is_reproducible(sp):
    """per segment: expanding (pair, multiplicity) gives exactly the index pairs of gather(); two runs agree bit for bit
    (the ordered hash table's layout does not depend on the order of the concurrent inserts)."""
    g = _load("sjoin_int_emptyrows.npz")
    z = _spg_from_golden(sp, g)
    own, partner = oracle.pair_segments(g["edge"])
    oseg, opairs = oracle.sjoin(g["z_indptr"], g["z_indices"], g["z_data"], own, partner)
    pairs, mult, ptr_ = sp.gather_pairs(g["edge"], z)
    pairs, mult, ptr_ = pairs.cpu().numpy(), mult.cpu().numpy(), ptr_.cpu().numpy()
    for j in range(len(own)):
        want = np.unique(opairs[oseg[j]:oseg[j + 1]], axis=0, return_counts=True)
        got = pairs[ptr_[j]:ptr_[j + 1]]
        order = np.lexsort((got[:, 1], got[:, 0]))
        assert np.array_equal(got[order], want[0]) and np.array_equal(mult[ptr_[j]:ptr_[j + 1]][order], want[1])
    ptr2, idx2 = sym_graph(6000, 40000, seed=31, hubs=2)
    zz, sets = sp.sample_spg(sp.DeviceCSR(ptr2, idx2), np.arange(6000), num_walks=200, num_steps=3, seed=3, rng="philox")
    edge = np.random.default_rng(2).integers(0, 6000, (2, 5000))
    edge[:, 0] = (17, 17)
    a = sp.gather_pairs(edge, zz)
    b = sp.gather_pairs(edge, zz)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    idxp, ind = sp.sjoin(zz, *[torch.from_numpy(v).cuda() for v in oracle.pair_segments(edge)], None, ptr_mode=True,
                         return_index=True, pair_block=5000)[:2]
    assert int(a[1].sum()) == idxp.shape[0] and a[0].shape[0] < idxp.shape[0] // 4      # same rows, far fewer of them
    # expand and compare as multisets per segment, on the device
    segc = torch.repeat_interleave(torch.arange(10000, device="cuda"), a[2][1:] - a[2][:-1])
    key_c = (segc.repeat_interleave(a[1].long()) << 42) | (a[0][:, 0].long().repeat_interleave(a[1].long()) << 21) | \
        a[0][:, 1].long().repeat_interleave(a[1].long())
    segx = torch.repeat_interleave(torch.arange(10000, device="cuda"), ind[1:] - ind[:-1])
    key_x = (segx << 42) | (idxp[:, 0].long() << 21) | idxp[:, 1].long()
    assert torch.equal(torch.sort(key_c)[0], torch.sort(key_x)[0])
    with pytest.raises(IndexError):
        sp.gather_pairs(np.array([[1], [6000]]), zz)


def test_attn_stage_trains_like_the_reference_first_stage(sp):
    """Forward and parameter gradients of the fused attention stage against pe_embedding(xz).sum(-2) + attentional
    aggregation (model.py:59-62,78-81) evaluated in float64 on the full xz.  Tolerances, stated here: forward within
    2e-5 of the largest entry; every parameter gradient within 5e-4 of its largest entry AND no worse than 4x the error
    the reference-style float32 stage itself makes against float64 (the gate's gradient is a sum with heavy
    cancellation -- the float32 reference is off by ~2e-4 there; the fused form sums ~20x fewer terms and measures
    1e-6..7e-5 depending on the atomics' order).  The gate bias has zero gradient (softmax is shift invariant): absolute bound."""
    ptr_, idx = sym_graph(3000, 15000, seed=6, hubs=1)
    z, sets = sp.sample_spg(sp.DeviceCSR(ptr_, idx), np.arange(3000), num_walks=64, num_steps=3, seed=2, rng="philox")
    table = sets.feature_table()
    edge = torch.from_numpy(np.random.default_rng(4).integers(0, 3000, (2, 512))).cuda()

    def nets(dtype):
        torch.manual_seed(7)
        return [mod.to(dtype) for mod in (
            torch.nn.Sequential(torch.nn.Linear(4, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16)).cuda(),
            torch.nn.Sequential(torch.nn.Linear(16, 1)).cuda(),
            torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.ReLU()).cuda())]
    fa, fb, f64 = nets(torch.float32), nets(torch.float32), nets(torch.float64)
    torch.manual_seed(1)
    w = torch.randn(2, 512, 16, device="cuda")
    fused = sp.attn_stage(edge, z, table, *fa)
    (fused * w).sum().backward()
    xz, ind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    ref32 = _reference_style_attn(xz, ind, *fb).view(2, -1, 16)
    (ref32 * w).sum().backward()
    truth = _reference_style_attn(xz.double(), ind, *f64).view(2, -1, 16)
    (truth * w.double()).sum().backward()
    scale = float(truth.detach().abs().max())
    assert float((fused.detach().double() - truth.detach()).abs().max()) <= 2e-5 * scale
    assert float((ref32.detach().double() - truth.detach()).abs().max()) <= 2e-5 * scale
    params = [[(n, p) for mod in ms for n, p in mod.named_parameters()] for ms in (fa, fb, f64)]
    for (n, pa), (_, pb), (_, pc) in zip(*params):
        gs = float(pc.grad.abs().max())
        if gs < 1e-9:                       # the gate bias: analytically zero
            assert float(pa.grad.abs().max()) < 1e-5
            continue
        err_fused = float((pa.grad.double() - pc.grad).abs().max()) / gs
        err_ref32 = float((pb.grad.double() - pc.grad).abs().max()) / gs
        assert err_fused <= 5e-4, (n, err_fused)
        assert err_fused <= max(4 * err_ref32, 1e-4), (n, err_fused, err_ref32)


@pytest.mark.parametrize("B,M,hops,idx64", [(2048, 200, 3, False), (700, 200, 2, False), (512, 100, 3, True), (300, 64, 4, False), (400, 200, 4, False), (300, 110, 4, True)])
def test_step_with_root_dedup_equals_the_plain_step(sp, B, M, hops, idx64):
    """StepBuffers(dedup_roots=True): every distinct endpoint sampled once, in the row of its first occurrence
    (subgacc_step_prologue_dedup + subgacc_walk_spg_sparse) -- bit for bit the (xz, indptr) of the step that samples every
    endpoint, batch after batch through the same buffers (the generation-stamped hash is never cleared), for batches full of
    repeated endpoints; first occurrences, segment lists, distinct count and the numbering of the LP rows are exact"""
    ptr_, idx = sym_graph(3000, 40000, seed=23, hubs=3)
    csr = sp.DeviceCSR(ptr_.astype(np.int64) if idx64 else ptr_, idx)
    plain = sp.StepBuffers(csr, B, num_walks=M, num_steps=hops)
    dd = sp.StepBuffers(csr, B, num_walks=M, num_steps=hops, dedup_roots=True)
    rng = np.random.default_rng(3)
    for s in range(4):
        hi = (40, 3000, 300, 3000)[s]                       # 40 distinct nodes: nearly every endpoint repeats
        e = torch.from_numpy(rng.integers(0, hi, (2, B))).cuda()
        e[:, 3] = e[0, 3]                                   # a (u, u) pair
        xz, ind, sets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=9, rng="philox", buffers=plain)
        sets.resolve()
        R = int(ind[-1].item())
        dxz, dind, dsets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=9, rng="philox", buffers=dd,
                                                dedup_roots=True)
        dsets.prefetch().resolve()
        assert torch.equal(ind, dind) and torch.equal(xz[:R], dxz[:R])
        flat = e.reshape(-1).cpu().numpy()
        _, first_idx = np.unique(flat, return_index=True)
        is_first = np.zeros(2 * B, bool)
        is_first[first_idx] = True
        roots = dd.roots.cpu().numpy()
        assert dsets.n_distinct == len(first_idx)
        assert np.array_equal(roots != -2 ** 31, is_first) and np.array_equal(roots[is_first], flat[is_first])
        assert np.array_equal(dd.own.cpu().numpy(), np.array([np.flatnonzero(flat == v)[0] for v in flat]) if B <= 700 else dd.own.cpu().numpy())
        assert dsets.X == int(dd.nsize.sum().item()) and dsets.X <= sets.X and int(dd.nsize[torch.from_numpy(~is_first).cuda()].sum()) == 0
        if s in (0, 1):       # the distinct LP rows keep the numbering of the whole batch (a repeated root is never first to show a row)
            assert torch.equal(dsets.number().ukeys, sets.number().ukeys)
    with pytest.raises(ValueError):
        sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=9, rng="philox", buffers=plain, dedup_roots=True)
    # ... and as ONE captured HIP graph per step: the stamp of the hash lives on the device, every replay gets a fresh one
    if B <= 1024:
        step = sp.CapturedStep(csr, B, num_walks=M, num_steps=hops, seed=9, dedup_roots=True)
        for s in range(3):
            e = torch.from_numpy(rng.integers(0, (60, 3000, 500)[s], (2, B))).cuda()
            xz, ind, _ = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=9, rng="philox", buffers=plain)
            cxz, cind = step(e).finish()
            assert torch.equal(ind, cind) and torch.equal(xz[: cxz.shape[0]], cxz) and cxz.shape[0] == int(ind[-1].item())
            assert step.distinct_roots == torch.unique(e).numel()


@pytest.mark.parametrize("M,hops", [(200, 3), (200, 2), (64, 3), (100, 4), (255, 4)])
def test_keyed_store_joins_like_the_table_join(sp, M, hops):
    """SpG.keyed(): the resident store re-keyed once (payload = LP key instead of SFptr+1, subgacc_sjoin_fill_keys) gives bit for
    bit the (xz, indptr) of the reference-style join with Z_SF = float32(enc) / M (main.py:174) -- and of the oracle"""
    ptr_, idx = sym_graph(4000, 30000, seed=17, hubs=2)
    csr = sp.DeviceCSR(ptr_, idx)
    z, sets = sp.sample_spg(csr, np.arange(4000), num_walks=M, num_steps=hops, seed=3, rng="philox")
    table = sets.feature_table()
    enc = oracle.enc_table(sets.enc_int16().cpu().numpy())              # [c+1, k] counts, zero row in front
    if hops * M.bit_length() + 1 > 31:                                  # the key does not fit 32 bits: refused, as the C ABI does
        with pytest.raises(AssertionError):
            z.keyed(enc, M)
        return
    zk = z.keyed(enc, M)
    assert zk.keyrows and zk.indices.data_ptr() == z.indices.data_ptr()
    edge = torch.from_numpy(np.random.default_rng(5).integers(0, 4000, (2, 3000))).cuda()
    edge[:, 7] = edge[0, 7]
    xz, ind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    kxz, kind = sp.gather(edge, zk, "cuda", ptr=True, encode=zk.slot_table())
    assert torch.equal(ind, kind) and torch.equal(xz, kxz)
    oxz, oind = oracle.gather(edge.cpu().numpy(), tuple(t.cpu().numpy() for t in (z.indptr, z.indices, z.data)), ptr=True,
                              encode=enc.astype(np.float32) / np.float32(M))
    assert np.array_equal(kxz.cpu().numpy(), oxz) and np.array_equal(kind.cpu().numpy(), oind)
    buf = torch.empty(int(xz.numel()) + 64, dtype=torch.float32, device="cuda")
    bxz, _ = sp.gather(edge, zk, "cuda", ptr=True, encode=zk.slot_table(), out=buf)
    assert torch.equal(bxz, xz) and bxz.data_ptr() == buf.data_ptr()
    sxz, sid = sp.gather(edge, z, "cuda", ptr=False, encode=table)                 # segment ids instead of pointers
    ksxz, ksid = sp.gather(edge, zk, "cuda", ptr=False, encode=zk.slot_table())
    assert torch.equal(sid, ksid) and torch.equal(sxz, ksxz)
    hedge = torch.from_numpy(np.random.default_rng(6).integers(0, 4000, (3, 500))).cuda()     # hgather (train.py:48-72)
    hxz, hid = sp.hgather(hedge, z, "cuda", encode=table)
    khxz, khid = sp.hgather(hedge, zk, "cuda", encode=zk.slot_table())
    assert torch.equal(hid, khid) and torch.equal(hxz, khxz)
    for bad in (lambda: sp.gather(edge, zk, "cuda", ptr=True, encode=table),
                lambda: sp.gather_counts(edge, zk, table.shape[0]), lambda: sp.gather_pairs(edge, zk)):
        with pytest.raises((ValueError, TypeError)):
            bad()


def _reference_style_lstm(xz, ptr, embed, lstm):
    """model.py:78-83 with LSTMAggregation as torch_geometric 2.x defines it: to_dense_batch (zero padding to the longest
    segment) -> lstm -> the output at the last position"""
    x = embed(xz).sum(dim=-2)
    S = ptr.numel() - 1
    lens = ptr[1:] - ptr[:-1]
    dense = x.new_zeros((S, int(lens.max()), x.shape[-1]))
    for j in range(S):
        dense[j, : int(lens[j])] = x[int(ptr[j]): int(ptr[j + 1])]
    return lstm(dense)[0][:, -1]


def test_gather_index_and_lstm_stage_match_the_reference_first_stage(sp):
    """Index form of the join == the index pairs behind gather()'s rows (bit-exact), from the packed, the strided and the
    key-rows form of the batch; lstm_stage == pe_embedding(xz).sum(-2) + LSTM aggregation on the full xz (model.py:63-65,
    78-83).  Tolerances: forward within 2e-5 of the largest entry of the float64 evaluation, parameter gradients within 5e-4
    of their largest entry (the same bounds as the attention stage; an LSTM over <= 200 positions is well conditioned)."""
    ptr_, idx = sym_graph(2000, 9000, seed=8, hubs=1)
    csr = sp.DeviceCSR(ptr_, idx)
    z, sets = sp.sample_spg(csr, np.arange(2000), num_walks=32, num_steps=3, seed=5, rng="philox")
    table = sets.feature_table()
    edge = torch.from_numpy(np.random.default_rng(9).integers(0, 2000, (2, 96))).cuda()
    edge[:, 5] = edge[0, 5]                                            # a (u, u) pair
    pairs, ind = sp.gather_index(edge, z)
    xz, ind2 = sp.gather(edge, z, "cuda", ptr=True, encode=table)
    assert pairs.dtype == torch.int32 and torch.equal(ind, ind2)
    assert torch.equal(table[pairs.long()], xz)
    o_xz, o_ind = oracle.gather(edge.cpu().numpy(), tuple(t.cpu().numpy() for t in (z.indptr, z.indices, z.data)), ptr=True,
                                encode=None)
    # ... and against the oracle's index form (train.py:25-33 before the table lookup)
    assert np.array_equal(pairs.cpu().numpy(), o_xz.astype(np.int32).reshape(-1, 2)) and np.array_equal(ind.cpu().numpy(), o_ind)

    # the transient forms of a batch (strided table rows, key rows) give the index pairs of their packed store
    from surel_plus_amd.graphs import query_pairs
    e = query_pairs(csr, 200, seed=3)
    rows = torch.arange(400, device="cuda").view(2, 200)
    for kr in (False, True):
        _, bind, bsets = sp.sample_and_gather(csr, e, num_walks=32, num_steps=3, seed=5, rng="philox", key_rows=kr)
        zs = sp.StridedSpG(bsets, csr.num_nodes)
        bp, bi = sp.gather_index(rows, zs)
        cp, ci = sp.gather_index(rows, zs.to_csr())
        assert torch.equal(bp, cp) and torch.equal(bi, ci) and torch.equal(bi, bind)

    def nets(dtype):
        torch.manual_seed(11)
        return [torch.nn.Sequential(torch.nn.Linear(4, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16)).cuda().to(dtype),
                torch.nn.LSTM(16, 16, batch_first=True).cuda().to(dtype)]
    fa, fb, f64 = nets(torch.float32), nets(torch.float32), nets(torch.float64)
    torch.manual_seed(2)
    w = torch.randn(2, 96, 16, device="cuda")
    fused = sp.lstm_stage(edge, z, table, *fa)
    (fused * w).sum().backward()
    ref32 = _reference_style_lstm(xz, ind, *fb).view(2, -1, 16)
    (ref32 * w).sum().backward()
    truth = _reference_style_lstm(xz.double(), ind, *f64).view(2, -1, 16)
    (truth * w.double()).sum().backward()
    scale = float(truth.detach().abs().max())
    assert float((fused.detach().double() - truth.detach()).abs().max()) <= 2e-5 * scale
    for (n, pa), (_, pc) in zip([(n, p) for mod in fa for n, p in mod.named_parameters()],
                                [(n, p) for mod in f64 for n, p in mod.named_parameters()]):
        gs = float(pc.grad.abs().max())
        assert float((pa.grad.double() - pc.grad).abs().max()) <= 5e-4 * max(gs, 1e-6), n


@pytest.mark.parametrize("B,hops,rng", [(1024, 3, "philox"), (1024, 2, "philox"), (300, 3, "rand_r")])
def test_captured_step_replays_equal_the_eager_step(sp, B, hops, rng):
    """sample -> SpG rows -> join as ONE HIP graph (stepgraph.CapturedStep, the reference's batch size of 1,024 pairs,
    main.py:32): every replay gives bit for bit what the eager calls give for the same pairs, also after the static
    buffers have carried other batches; a bad node id in a replayed batch is still an IndexError."""
    from surel_plus_amd.graphs import query_pairs
    ptr_, idx = sym_graph(20000, 120000, seed=13, hubs=3)
    csr = sp.DeviceCSR(ptr_, idx)
    step = sp.CapturedStep(csr, B, num_walks=100, num_steps=hops, seed=9, rng=rng)
    for s in (1, 2, 3, 1):
        e = query_pairs(csr, B, seed=s)
        xz, ind = step(e).finish()
        exz, eind, esets = sp.sample_and_gather(csr, e, num_walks=100, num_steps=hops, seed=9, rng=rng)
        assert torch.equal(ind, eind) and torch.equal(xz, exz)
        assert step.members == esets.X
        if esets.strided:       # joined by table slot: the distinct LP rows are never numbered ... until somebody asks
            assert step.distinct_rows is None and esets.ukeys is None and esets.c == esets.number().ukeys.numel() > 0
        else:
            assert step.distinct_rows == esets.c
    bad = query_pairs(csr, B, seed=5)
    bad[1, 7] = 20000
    with pytest.raises(IndexError):
        step(bad).finish()
    xz, ind = step(query_pairs(csr, B, seed=2)).finish()          # and the step is usable afterwards
    exz, eind, _ = sp.sample_and_gather(csr, query_pairs(csr, B, seed=2), num_walks=100, num_steps=hops, seed=9, rng=rng)
    assert torch.equal(ind, eind) and torch.equal(xz, exz)


@pytest.mark.parametrize("rng", ["rand_r", "philox"])
@pytest.mark.parametrize("M,hops,idx64", [(200, 3, False), (200, 2, True), (100, 3, True), (120, 2, False),
                                          # round 4: 4 hops -- 32-bit keys up to M = 127 (512- and 1,024-slot tables), 64-bit keys beyond
                                          # (the paper's Fig. 6a setting m = 4, M = 200: 33 bits; subgacc_walk_keyrows64)
                                          (100, 4, False), (120, 4, True), (200, 4, False), (128, 4, True), (204, 4, False)])
def test_key_rows_join_like_table_rows(sp, rng, M, hops, idx64):
    """rows that carry the LP key instead of a table slot (csrc/walk_rows.hip KR form, subgacc_sjoin_fill_keyrows / _keyrows64):
    the same (xz, indptr) as the table form and as the oracle; numbering, enc and the packed CSR on demand (sampled again)"""
    from surel_plus_amd.graphs import query_pairs
    from surel_plus_amd.sampler import key_rows_form
    assert key_rows_form(M, hops) == (64 if (hops == 4 and M >= 128) else 32)
    ptr_, idx = sym_graph(8000, 70000, seed=31, hubs=3)
    csr = sp.DeviceCSR(ptr_.astype(np.int64) if idx64 else ptr_, idx)
    e = query_pairs(csr, 700, seed=3)
    e[:, 5] = e[0, 5]                                              # a (u, u) pair
    xz, ind, sets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=21, rng=rng)
    assert sets.keyrows and sets.table is None and sets.key64 == (key_rows_form(M, hops) == 64)
    txz, tind, tsets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=21, rng=rng, key_rows=False)
    assert not tsets.keyrows and torch.equal(ind, tind) and torch.equal(xz, txz)
    # against the oracle: sets of both endpoints -> SpG -> gather with Z_SF = enc / M
    roots = e.reshape(-1).cpu().numpy()
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, roots, M, hops, 21, rng, -1)
    table = oracle.enc_table(oenc).astype(np.float32) / np.float32(M)
    rows = np.arange(2 * 700, dtype=np.int64).reshape(2, 700)
    oxz, oind = oracle.gather(rows, (oi, ox, od), ptr=True, encode=table)
    assert np.array_equal(ind.cpu().numpy(), oind) and np.array_equal(xz.cpu().numpy(), oxz)
    # what the key rows do not carry is sampled again on demand: numbering, enc, the packed store
    assert sets.c == tsets.c and torch.equal(sets.number().ukeys, tsets.number().ukeys)
    assert torch.equal(sets.enc_int16(), tsets.enc_int16())
    z = sp.StridedSpG(sets, csr.num_nodes)
    zc, tc = z.to_csr(), sp.StridedSpG(tsets, csr.num_nodes).to_csr()
    assert torch.equal(zc.indptr, tc.indptr) and torch.equal(zc.indices, tc.indices) and torch.equal(zc.data, tc.data)


@pytest.mark.parametrize("B,M,hops", [(512, 100, 3), (300, 200, 2), (64, 50, 4), (300, 200, 4), (200, 100, 4)])
def test_buffered_step_equals_the_allocating_step(sp, B, M, hops):
    """spjoin.StepBuffers: the on-demand step as six launches over preallocated buffers gives bit for bit what the general
    form gives, batch after batch through the same buffers; errors surface at resolve() as they do for lazy=True"""
    from surel_plus_amd.graphs import query_pairs
    ptr_, idx = sym_graph(20000, 120000, seed=13, hubs=3)
    csr = sp.DeviceCSR(ptr_, idx)
    bufs = sp.StepBuffers(csr, B, num_walks=M, num_steps=hops)
    for s in (1, 2, 3):
        e = query_pairs(csr, B, seed=s)
        xz, ind, sets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=9, rng="philox", buffers=bufs)
        sets.prefetch().resolve()
        exz, eind, esets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=9, rng="philox")
        rows = int(sets.extra[0])
        assert rows == exz.shape[0] == sets.X == esets.X
        assert torch.equal(ind, eind) and torch.equal(xz[:rows], exz)
        assert sets.c == esets.c and torch.equal(sets.number().ukeys, esets.number().ukeys)
    bad = query_pairs(csr, B, seed=4)
    bad[0, 3] = 20000
    _, _, sets = sp.sample_and_gather(csr, bad, num_walks=M, num_steps=hops, seed=9, rng="philox", buffers=bufs)
    with pytest.raises(IndexError):
        sets.resolve()
    with pytest.raises(ValueError):
        sp.sample_and_gather(csr, bad, num_walks=M, num_steps=hops, seed=9, rng="rand_r", buffers=bufs)
    with pytest.raises(ValueError):
        sp.StepBuffers(csr, 8, num_walks=300, num_steps=4)             # 1,201 members: beyond the fused-row kernel


def test_captured_step_pool_keeps_batches_in_flight_on_their_own_streams(sp):
    """stepgraph.CapturedStepPool: four captured steps on four streams, three batches in flight; every batch equals the
    eager step for the same pairs, whatever was in flight around it"""
    from surel_plus_amd.graphs import query_pairs
    ptr_, idx = sym_graph(20000, 120000, seed=13, hubs=3)
    csr = sp.DeviceCSR(ptr_, idx)
    B = 512
    pool = sp.CapturedStepPool(csr, B, lanes=4, num_walks=100, num_steps=3, seed=9, rng="philox")
    edges = [query_pairs(csr, B, seed=s) for s in range(10)]
    want = [sp.sample_and_gather(csr, e, num_walks=100, num_steps=3, seed=9, rng="philox")[:2] for e in edges]
    inflight, got = [], []
    for s, e in enumerate(edges):
        inflight.append((s, pool.submit(e)))
        if len(inflight) == 4:
            with pytest.raises(RuntimeError):
                pool.submit(e)                                   # all four lanes busy
            s0, t0 = inflight.pop(0)
            xz, ind = pool.finish(t0)
            got.append((s0, xz.clone(), ind.clone()))
    for s0, t0 in inflight:
        xz, ind = pool.finish(t0)
        got.append((s0, xz.clone(), ind.clone()))
    assert [g[0] for g in got] == list(range(10))
    for s0, xz, ind in got:
        assert torch.equal(ind, want[s0][1]) and torch.equal(xz, want[s0][0]), s0


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


# ------------------------------------------------------------------------------- batch_sampler (legacy SUREL mini-batches)
@pytest.mark.parametrize("name", golden_files("batch_"))
def test_batch_sampler_matches_reference_golden(sp, name):
    """subg_acc.c:391-507; the fixture records the effective seed (seed + getpid()) of the reference run that made it"""
    g = _load(name)
    out = sp.batch_sampler(g["indptr"], g["indices"], g["query"], num_walks=int(g["M"]), num_steps=int(g["S"]),
                           thld=int(g["thld"]), seed=int(g["seed_eff"]), pid=0)
    assert out.dtype == np.int32
    np.testing.assert_array_equal(out, g["out"])


def test_batch_sampler_vs_oracle_and_its_process_seed(sp):
    indptr, indices = sym_graph(20000, 120000, 17, hubs=2)          # hubs: degree > num_walks -> Fisher-Yates first hops
    rng = np.random.default_rng(5)
    for n, M, S, thld in ((64, 200, 8, 1000), (700, 50, 4, 5000), (33, 300, 6, 20000), (5, 7, 1, 10), (1, 1, 1, 1)):
        q = rng.integers(0, 20000, n).astype(np.int32)
        q[0] = 0                                                     # a hub root
        for seed in (111413, 3):
            out = sp.batch_sampler(indptr, indices, q, num_walks=M, num_steps=S, thld=thld, seed=seed, pid=12345)
            ref_ = oracle.batch_sampler(indptr, indices, q, num_walks=M, num_steps=S, thld=thld, seed_eff=seed + 12345)
            np.testing.assert_array_equal(out, ref_)
    # pid=None: the reference's seed + getpid() (subg_acc.c:421)
    out = sp.batch_sampler(indptr, indices, q, num_walks=9, num_steps=3, thld=50, seed=1)
    np.testing.assert_array_equal(out, oracle.batch_sampler(indptr, indices, q, num_walks=9, num_steps=3, thld=50,
                                                            seed_eff=1 + os.getpid()))
    # int64 row offsets, device-resident graph, a root out of range, a graph with a dead end
    from surel_plus_amd import DeviceCSR
    csr64 = DeviceCSR(indptr.astype(np.int64), indices)
    out64 = sp.batch_sampler(csr64, None, q, num_walks=9, num_steps=3, thld=50, seed=1)
    np.testing.assert_array_equal(out64, out)
    with pytest.raises(IndexError):
        sp.batch_sampler(indptr, indices, np.array([1, 20000]), num_walks=4, num_steps=2)
    dp = np.array([0, 1, 1], np.int32)                               # 0 -> 1, node 1 has no out-edges
    with pytest.raises(sp.SubgAccError, match="out-edges"):
        sp.batch_sampler(dp, np.array([1], np.int32), np.array([0]), num_walks=2, num_steps=3)


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


# ------------------------------------------------------------------ callable from pgather-style Python threads
def test_four_threads_on_their_own_streams_give_the_serial_results(sp):
    """The reference's pgather calls bgather from 4 Python threads (train.py:88-99); a maintainer who keeps that function and
    imports only the join gets exactly this: threads, each on its own HIP stream, calling gather / bgather / sample_and_gather /
    subg_matrix-style sampling concurrently on SHARED SpG / DeviceCSR objects (and on the module's shared caches: segment
    lists, hop records).  Every result must equal the serial one, bit for bit."""
    import threading
    from surel_plus_amd import spjoin
    from surel_plus_amd.graphs import query_pairs
    ptr_, idx = sym_graph(12000, 90000, seed=4, hubs=3)
    M, hops, B, T, ROUNDS = 200, 3, 384, 4, 6
    csr = sp.DeviceCSR(ptr_, idx)
    z, enc = sp.subg_matrix(csr, np.arange(12000), num_walks=M, num_steps=hops + 1, rng="philox")
    table = torch.from_numpy(enc).cuda().float() / M
    edges = [query_pairs(csr, B, seed=50 + j) for j in range(T * ROUNDS)]
    torch.cuda.synchronize()

    def one(e):
        xz, ind = sp.gather(e, z, "cuda", ptr=True, encode=table)
        out = np.empty(4, dtype=object)                                     # a row of pgather's out_blocks (train.py:89)
        sp.bgather(e, z, out)                                               # train.py:75-85, the piece pgather's threads run
        dxz, dind, _ = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=111413, rng="philox")
        return (xz.cpu(), ind.cpu(), dxz.cpu(), dind.cpu()) + tuple(torch.from_numpy(np.ascontiguousarray(o)) for o in out)

    want = [one(e) for e in edges]
    for own_streams in (True, False):     # each thread on a stream of its own / all of them on the default stream, as the reference's are
        spjoin._ARANGE_SEGMENTS.clear()
        if hasattr(csr, "_recs"):
            del csr._recs                 # the shared caches start cold: the threads race to fill them
        results, errors = [None] * len(edges), []

        def worker(t):
            try:
                if own_streams:
                    st = torch.cuda.Stream()
                    with torch.cuda.stream(st):
                        for j in range(t, len(edges), T):
                            results[j] = one(edges[j])
                else:
                    for j in range(t, len(edges), T):
                        results[j] = one(edges[j])
            except Exception as ex:       # noqa: BLE001 -- reported below
                errors.append(repr(ex))
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors
        torch.cuda.synchronize()
        for j in range(len(edges)):
            assert all(torch.equal(a, b) for a, b in zip(results[j], want[j])), f"batch {j} (own streams: {own_streams})"


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
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-x", "-q", "-k",
                        "key_rows_join_like or specialised_fused_row or batched_registration_numbers"],
                       env=dict(os.environ, SUBGACC_LIB=lib), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_captured_join_over_a_resident_store_equals_gather(sp):
    """stepgraph.CapturedJoin: the join of a resident store with everything built once, ONE library call per batch (and graph=True:
    the same launches replayed as a HIP graph), batch after batch --
    float payload (the PPR store), the Z_SF-table store and the keyed store; same (xz, indptr) as gather(), IndexError for a
    row outside the store at finish()."""
    from surel_plus_amd.graphs import ppr_like_spg, query_pairs
    ptr_, idx = sym_graph(6000, 50000, seed=3, hubs=2)
    csr = sp.DeviceCSR(ptr_, idx)
    z, enc = sp.subg_matrix(csr, np.arange(6000), num_walks=100, num_steps=4, rng="philox")
    table = torch.from_numpy(enc).cuda().float() / 100
    zk = z.keyed(enc, 100)
    zf = ppr_like_spg(6000, 100, seed=3)
    B = 500
    for store, encode in ((zf, None), (z, table), (zk, zk.slot_table())):
        for graph in (False, True):
            cj = sp.CapturedJoin(store, B, encode=encode, graph=graph)
            for s_ in (1, 2, 3):
                e = query_pairs(csr, B, seed=s_)
                xz, ind = cj(e).finish()
                wxz, wind = sp.gather(e, store, "cuda", ptr=True, encode=encode)
                assert torch.equal(ind, wind) and torch.equal(xz, wxz)
    bad = query_pairs(csr, B, seed=9)
    bad[1, 7] = 6000
    with pytest.raises(IndexError):
        cj(bad).finish()


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


@pytest.mark.parametrize("B,M,hops", [(9000, 200, 3), (300, 200, 2)])
def test_buffered_step_in_rand_r_mode_is_the_reference_stream(sp, B, M, hops):
    """StepBuffers(rng='rand_r'): the allocation-free step in the reference's own RNG mode -- the rows' places in the sequential
    stream come from subgacc_rng_positions per step, the walk takes them in order of root id (B >= 8,192: the sorted work
    list) or in batch order -- bit for bit the allocating step and the oracle's sequential sampler + join."""
    from surel_plus_amd.graphs import query_pairs
    ptr_, idx = sym_graph(20000, 120000, seed=13, hubs=3)
    csr = sp.DeviceCSR(ptr_, idx)
    bufs = sp.StepBuffers(csr, B, num_walks=M, num_steps=hops, rng="rand_r")
    for s_ in (1, 2):
        e = query_pairs(csr, B, seed=s_)
        xz, ind, sets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=9, rng="rand_r", buffers=bufs)
        sets.prefetch().resolve()
        R = int(ind[-1].item())
        wxz, wind, _ = sp.sample_and_gather(csr, e, num_walks=M, num_steps=hops, seed=9, rng="rand_r")
        assert torch.equal(ind, wind) and torch.equal(xz[:R], wxz)
    roots = e.reshape(-1).cpu().numpy()
    (oi, ox, od), oenc = _oracle_spg(ptr_, idx, roots, M, hops, 9, "rand_r", -1)
    table = oracle.enc_table(oenc).astype(np.float32) / np.float32(M)
    oxz, oind = oracle.gather(np.arange(2 * B, dtype=np.int64).reshape(2, B), (oi, ox, od), ptr=True, encode=table)
    assert np.array_equal(ind.cpu().numpy(), oind) and np.array_equal(xz[:R].cpu().numpy(), oxz)
    assert np.array_equal(sets.enc_int16().cpu().numpy(), oenc)


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
