"""GPU (MI355X), SURVEY 8(a) row a9: the SpG build (random_walks.py:74-82) against SciPy goldens and the oracle, persistence."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, golden_files
from gpu_helpers import _load, _oracle_counts, _oracle_spg, _reference_style_attn, _reference_style_lstm, _spg_from_golden, _walkjoin_inputs, dir_graph, sp, sym_graph  # noqa: F401

pytestmark = pytest.mark.gpu


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
