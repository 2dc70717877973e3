"""GPU (MI355X), SURVEY 8(a) rows a10-a12 and 8(f).1-2: SpJoin -- gather / bgather / pgather / hgather, the strided and keyed forms, the
on-demand step (buffers, captured steps, pools), the count / pair / index forms with the first model stage -- against the reference's
golden vectors and the oracle.  Bit-exact; the only floating point is the table lookup (exact copies) and LP normalisation (float32
division, tolerance 0)."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, golden_files
from gpu_helpers import _load, _oracle_counts, _oracle_spg, _reference_style_attn, _reference_style_lstm, _spg_from_golden, _walkjoin_inputs, dir_graph, sp, sym_graph  # noqa: F401

pytestmark = pytest.mark.gpu


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
