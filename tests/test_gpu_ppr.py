"""GPU (MI355X): top-K approximate-PPR sets through the C ABI against oracle/ppr_oracle.c.

The oracle restates sampler/pprgo.py with numba's typing; it is UNPINNED (numba is not in the image), so besides
bit-exact agreement HIP <-> oracle the tests check the defining property of the approximation against an exact
PPR computed by power iteration (float64, tolerance = the push threshold eps * deg)."""
import os

import numpy as np
import pytest
import scipy.sparse as sps
import torch

from oracle import oracle as orc
from gpu_helpers import sym_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ppr():
    from surel_plus_amd import _lib, ppr
    assert os.path.exists(_lib.LIB_PATH), "libsubgacc_hip.so must be built (no fallback)"
    assert _lib.lib().subgacc_device_count() >= 1, "no gfx950 device"
    return ppr


def directed_graph(N, E, seed):
    rng = np.random.default_rng(seed)
    A = sps.csr_matrix((np.ones(E), (rng.integers(0, N // 2, E), rng.integers(0, N, E))), shape=(N, N))   # rows >= N/2: no out-edges
    A.sum_duplicates()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


def _graph(kind):
    if kind == "sparse":
        return sym_graph(3000, 9000, 1)
    if kind == "hubs":
        return sym_graph(2000, 6000, 2, hubs=3)
    if kind == "dense":
        return sym_graph(400, 12000, 3)
    if kind == "directed":
        return directed_graph(1500, 9000, 4)
    if kind == "star":
        N = 300
        A = sps.lil_matrix((N, N))
        A[0, 1:] = 1
        A[1:, 0] = 1
        A = A.tocsr()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32)
    if kind == "edgeless":
        return np.zeros(11, np.int32), np.zeros(0, np.int32)
    raise KeyError(kind)


def _check_rows(got, want):
    off, ids, vals = got[0].cpu().numpy(), got[1].cpu().numpy(), got[2].cpu().numpy()
    np.testing.assert_array_equal(off, want[0])
    np.testing.assert_array_equal(ids, want[1])
    np.testing.assert_array_equal(vals.view(np.int32), want[2].view(np.int32))     # float32 scores, bit for bit


@pytest.mark.parametrize("kind", ["sparse", "hubs", "dense", "directed", "star", "edgeless"])
@pytest.mark.parametrize("alpha,eps,topk", [(0.5, 1e-4, 100), (0.1, 1e-4, 100), (0.7, 1e-3, 5), (0.15, 1e-5, 32)])
def test_ppr_topk_matches_oracle(ppr, kind, alpha, eps, topk):
    from surel_plus_amd import DeviceCSR
    indptr, indices = _graph(kind)
    N = len(indptr) - 1
    if kind == "dense" and eps < 1e-4:
        pytest.skip("covered by the sparse graphs")
    roots = np.arange(N, dtype=np.int32)
    want = orc.ppr_topk(indptr, indices, roots, alpha, eps, topk, table_log2=18)
    got = ppr.ppr_topk(DeviceCSR(indptr, indices), alpha, eps, roots, topk)
    _check_rows(got, want)
    assert got[3] == want[3]                                    # same number of pushes


def test_ppr_int64_offsets_duplicate_and_shuffled_roots(ppr):
    from surel_plus_amd import DeviceCSR
    indptr, indices = _graph("sparse")
    rng = np.random.default_rng(5)
    roots = rng.integers(0, len(indptr) - 1, 5000).astype(np.int32)      # repeats, arbitrary order
    want = orc.ppr_topk(indptr, indices, roots, 0.5, 1e-4, 20, table_log2=16)
    got = ppr.ppr_topk(DeviceCSR(indptr.astype(np.int64), indices), 0.5, 1e-4, roots, 20)
    _check_rows(got, want)


def test_ppr_small_tables_are_retried(ppr):
    """table_log2 = 10 holds 512 nodes: most roots overflow and are re-run with larger tables."""
    from surel_plus_amd import DeviceCSR
    indptr, indices = _graph("hubs")
    roots = np.arange(len(indptr) - 1, dtype=np.int32)
    want = orc.ppr_topk(indptr, indices, roots, 0.3, 1e-4, 50, table_log2=18)
    got = ppr.ppr_topk(DeviceCSR(indptr, indices), 0.3, 1e-4, roots, 50, table_log2=10)
    _check_rows(got, want)


@pytest.mark.parametrize("norm", ["row", "sym", "col"])
def test_topk_ppr_matrix_and_encoding(ppr, norm):
    from surel_plus_amd import DeviceCSR
    indptr, indices = _graph("directed")       # rows of degree 0 exercise max(deg, 1e-12)
    idx = np.arange(len(indptr) - 1, dtype=np.int32)
    off, ids, data = orc.topk_ppr_matrix(indptr, indices, 0.5, 1e-4, idx, 30, normalization=norm, table_log2=16)
    z = ppr.topk_ppr_matrix(DeviceCSR(indptr, indices), 0.5, 1e-4, idx, 30, normalization=norm)
    np.testing.assert_array_equal(z.indptr.cpu().numpy(), off)
    np.testing.assert_array_equal(z.indices.cpu().numpy(), ids)
    np.testing.assert_array_equal(z.data.cpu().numpy().view(np.int64), data.view(np.int64))      # float64, bit for bit
    enc = orc.ppr_encode(data)
    z2 = ppr.topk_ppr_matrix(DeviceCSR(indptr, indices), 0.5, 1e-4, idx, 30, normalization=norm, encode=True)
    np.testing.assert_array_equal(z2.data.cpu().numpy().view(np.int64), enc.view(np.int64))
    z3, agg = ppr.encoding(z, None, "PPR")
    assert agg is None
    np.testing.assert_array_equal(z3.data.cpu().numpy().view(np.int64), enc.view(np.int64))
    with pytest.raises(ValueError):
        ppr.topk_ppr_matrix(DeviceCSR(indptr, indices), 0.5, 1e-4, idx, 30, normalization="nope")


def test_ppr_approximation_property(ppr):
    """0 <= ppr_exact(v) - p(v) <= eps * deg(v) for the kept entries (Andersen-Chung-Lang guarantee); this does not
    depend on the oracle."""
    from surel_plus_amd import DeviceCSR
    indptr, indices = sym_graph(1500, 5000, 7)
    N = len(indptr) - 1
    alpha, eps = 0.5, 1e-4
    roots = np.array([0, 3, 77, 512, 1499], dtype=np.int32)
    off, ids, vals, _ = ppr.ppr_topk(DeviceCSR(indptr, indices), alpha, eps, roots, N)     # keep everything
    off, ids, vals = off.cpu().numpy(), ids.cpu().numpy(), vals.cpu().numpy()
    A = sps.csr_matrix((np.ones(len(indices)), indices, indptr), shape=(N, N))
    deg = np.maximum(np.diff(indptr), 1).astype(np.float64)
    P = sps.diags(1.0 / deg) @ A
    for i, s in enumerate(roots):
        x = np.zeros(N)
        x[s] = 1.0
        pi = np.zeros(N)
        for _ in range(120):
            pi += alpha * x
            x = (1 - alpha) * (P.T @ x)
        row = slice(off[i], off[i + 1])
        err = pi[ids[row]] - vals[row].astype(np.float64)
        assert err.min() > -1e-6
        assert (err / (eps * deg[ids[row]])).max() < 1.0 + 1e-3
        missing = np.setdiff1d(np.arange(N), ids[row])
        assert (pi[missing] <= eps * deg[missing] * (1 + 1e-3)).all()       # never-pushed nodes hold less than the threshold


def test_ppr_spg_feeds_gather(ppr):
    """the float64 SpG goes straight into SpJoin's float mode (train.py:39-43)"""
    import surel_plus_amd as sp
    from surel_plus_amd import DeviceCSR
    indptr, indices = _graph("sparse")
    N = len(indptr) - 1
    z = ppr.topk_ppr_matrix(DeviceCSR(indptr, indices), 0.5, 1e-4, np.arange(N), 100, normalization="sym", encode=True)
    rng = np.random.default_rng(0)
    edge = rng.integers(0, N, (2, 512))
    xz, ip = sp.gather(torch.from_numpy(edge).cuda(), z, "cuda", ptr=True, encode=None)
    want_xz, want_ip = orc.gather_numpy(edge, (z.indptr.cpu().numpy(), z.indices.cpu().numpy(), z.data.cpu().numpy()),
                                        ptr=True, encode=None)
    np.testing.assert_array_equal(ip.cpu().numpy(), want_ip)
    np.testing.assert_array_equal(xz.cpu().numpy(), want_xz)


@pytest.mark.parametrize("enc", ["DEG", "SPD"])
@pytest.mark.parametrize("kind", ["sparse", "hubs", "dense", "star"])
def test_deg_spd_encoders_match_scipy(ppr, enc, kind):
    """utils.py:22-34 executed by SciPy (oracle.encoding_scipy) vs the union kernels: pattern and float64 values, bit for bit"""
    from surel_plus_amd import DeviceCSR
    indptr, indices = _graph(kind)
    N = len(indptr) - 1
    adj = sps.csr_matrix((np.ones(len(indices)), indices, indptr), shape=(N, N))
    csr = DeviceCSR(indptr, indices)
    x = ppr.topk_ppr_matrix(csr, 0.5, 1e-4, np.arange(N), 30, normalization="sym")
    want, wagg = orc.encoding_scipy(x.to_scipy(), adj, enc)
    z, agg = ppr.encoding(x, csr, enc)
    got = z.to_scipy()
    np.testing.assert_array_equal(got.indptr, want.indptr)
    np.testing.assert_array_equal(got.indices, want.indices)
    np.testing.assert_array_equal(got.data.view(np.int64), want.data.astype(np.float64).view(np.int64))
    if enc == "DEG":
        ga = agg.to_scipy()
        np.testing.assert_array_equal(ga.indices, wagg.indices)
        np.testing.assert_array_equal(ga.data.view(np.int64), wagg.data.view(np.int64))
    else:
        assert agg is None and wagg is None
    # and the result feeds SpJoin's float mode, hub rows included (rows longer than LDS are searched in place)
    import surel_plus_amd as sp
    edge = np.random.default_rng(0).integers(0, N, (2, 200))
    xz, ip = sp.gather(torch.from_numpy(edge).cuda(), z, "cuda", ptr=True, encode=None)
    wxz, wip = orc.gather_numpy(edge, (want.indptr.astype(np.int64), want.indices, want.data.astype(np.float64)), ptr=True, encode=None)
    np.testing.assert_array_equal(ip.cpu().numpy(), wip)
    np.testing.assert_array_equal(xz.cpu().numpy(), wxz)
