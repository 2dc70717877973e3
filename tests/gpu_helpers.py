"""Shared by the GPU test files: the `sp` fixture (the package with its library built and loaded), graph builders, fixture loaders and
the oracle-side helpers the parity tests compare against."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, golden_files


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


# ------------------------------------------------------------------------------------ SpJoin
def _spg_from_golden(sp, g):
    data = g["z_data"]
    data = torch.from_numpy(data.astype(np.float64) if data.dtype.kind == "f" else data.astype(np.int32))
    return sp.SpG(torch.from_numpy(g["z_indptr"]).cuda(), torch.from_numpy(g["z_indices"]).cuda(), data.cuda())


# ------------------------------------------------------------------------ fused SpG pipeline (walk_spg)
def _oracle_spg(ptr_, idx, q, M, m, seed, rng, bucket=-1):
    nsize, remap, enc = oracle.gset_sampler(ptr_, idx, q, num_walks=M, num_steps=m, bucket=bucket, seed=seed, rng=rng,
                                            nthreads=8 if rng == "philox" else 1)
    return oracle.spg_build(nsize, remap), enc


# ----------------------------------------------------------------- count form of the join (next row f.1)
def _oracle_counts(spg, edge, rows):
    own, partner = oracle.pair_segments(edge)
    seg, pairs = oracle.sjoin(spg[0], spg[1], spg[2], own, partner)
    C = np.zeros((len(own), rows), np.float32)
    segid = np.repeat(np.arange(len(own)), np.diff(seg))
    np.add.at(C, (segid, pairs[:, 0]), 1)
    np.add.at(C, (segid, pairs[:, 1]), 1)
    return C, np.diff(seg)


# ------------------------------------------------------------------------------- walk_join (legacy SUREL join)
def _walkjoin_inputs(g):
    off = np.concatenate([[0], np.cumsum(g["key_len"])])
    return g["walks"], [g["key_ids"][off[i]:off[i + 1]] for i in range(len(g["key_len"]))], g["query"]


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
