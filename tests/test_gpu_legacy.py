"""GPU (MI355X), SURVEY 8(f).4: the legacy SUREL surface -- walk_join, batch_sampler, rw_matrix / np_sampling -- against the reference's
golden vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, golden_files
from gpu_helpers import _load, _oracle_counts, _oracle_spg, _reference_style_attn, _reference_style_lstm, _spg_from_golden, _walkjoin_inputs, dir_graph, sp, sym_graph  # noqa: F401

pytestmark = pytest.mark.gpu


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
