"""CPU: oracle/ppr_oracle.c (the checker of the GPU PPR sampler).  The reference function is numba code that cannot
run here (parity UNPINNED), so the restatement is pinned to what can be checked without it: the defining property of
the Andersen-Chung-Lang approximation against an exact PPR, committed vectors of the push (tests/golden/ppr_push.npz, made by
oracle/gen_ppr_fixture.py: a pure-Python second statement with NumPy scalars typed as numba types them), and the
normalisation / encoding formulas against their NumPy one-liners (pprgo.py:88-108, utils.py:35-36)."""
import numpy as np
import scipy.sparse as sps

from oracle import oracle as orc


def _graph(N=400, E=1400, seed=0):
    rng = np.random.default_rng(seed)
    r, c = rng.integers(0, N, E), rng.integers(0, N, E)
    A = sps.csr_matrix((np.ones(2 * E), (np.r_[r, c], np.r_[c, r])), shape=(N, N))
    A.setdiag(0)
    A.eliminate_zeros()
    A.data[:] = 1
    A.sort_indices()
    return A


def test_oracle_matches_the_committed_push_vectors():
    """tests/golden/ppr_push.npz: the push of sampler/pprgo.py:9-38 evaluated once by a pure-Python second statement (dicts and
    a LIFO list, NumPy scalars typed as numba types them) in the build container -- oracle/gen_ppr_fixture.py, committed
    beside its output.  Node sets and score BIT PATTERNS must agree."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "ppr_push.npz"))
    n = len(g["indptr"]) - 1
    for j, (alpha, eps) in enumerate(g["params"]):
        off, ids, vals, _ = orc.ppr_topk(g["indptr"], g["indices"], g["roots"], float(alpha), float(eps), n, table_log2=12)
        np.testing.assert_array_equal(off, g[f"off_{j}"])
        np.testing.assert_array_equal(ids, g[f"ids_{j}"])
        np.testing.assert_array_equal(vals.view(np.int32), g[f"score_bits_{j}"])


def test_oracle_topk_keeps_largest_scores_rows_sorted():
    A = _graph(300, 2500, 3)
    N = A.shape[0]
    full = orc.ppr_topk(A.indptr, A.indices, np.arange(N), 0.3, 1e-4, N, table_log2=12)
    top = orc.ppr_topk(A.indptr, A.indices, np.arange(N), 0.3, 1e-4, 10, table_log2=12)
    for i in range(N):
        fi, fv = full[1][full[0][i]:full[0][i + 1]], full[2][full[0][i]:full[0][i + 1]]
        ti, tv = top[1][top[0][i]:top[0][i + 1]], top[2][top[0][i]:top[0][i + 1]]
        assert len(ti) == min(10, len(fi)) and (np.diff(ti) > 0).all()
        assert np.isin(ti, fi).all()
        rest = fv[~np.isin(fi, ti)]
        assert rest.size == 0 or rest.max() <= tv.min()
        np.testing.assert_array_equal(tv, fv[np.isin(fi, ti)])


def test_oracle_approximation_property():
    A = _graph(500, 2000, 5)
    N = A.shape[0]
    alpha, eps = 0.5, 1e-4
    deg = np.maximum(np.diff(A.indptr), 1).astype(np.float64)
    P = sps.diags(1.0 / deg) @ A
    roots = np.array([0, 10, 250], np.int32)
    off, ids, vals, _ = orc.ppr_topk(A.indptr, A.indices, roots, alpha, eps, N, table_log2=12)
    for i, s in enumerate(roots):
        x = np.zeros(N)
        x[s] = 1.0
        pi = np.zeros(N)
        for _ in range(120):
            pi += alpha * x
            x = (1 - alpha) * (P.T @ x)
        row = slice(off[i], off[i + 1])
        err = pi[ids[row]] - vals[row]
        assert err.min() > -1e-6 and (err / (eps * deg[ids[row]])).max() < 1.001


def test_oracle_normalisation_and_encoding_formulas():
    A = _graph(200, 500, 9)
    N = A.shape[0]
    idx = np.arange(N, dtype=np.int32)
    off, ids, vals, _ = orc.ppr_topk(A.indptr, A.indices, idx, 0.5, 1e-4, 20, table_log2=12)
    M = sps.csr_matrix((vals, ids, off), shape=(N, N))
    deg = np.asarray(A.sum(1)).ravel()
    row, col = M.nonzero()
    deg_sqrt = np.sqrt(np.maximum(deg, 1e-12))
    want = {"row": M.data.astype(np.float64),
            "sym": deg_sqrt[idx[row]] * M.data * (1.0 / deg_sqrt)[col],                     # pprgo.py:88-98
            "col": deg[idx[row]] * M.data * (1.0 / np.maximum(deg, 1e-12))[col]}             # pprgo.py:99-108
    for mode, w in want.items():
        o2, i2, d2 = orc.topk_ppr_matrix(A.indptr, A.indices, 0.5, 1e-4, idx, 20, normalization=mode, table_log2=12)
        np.testing.assert_array_equal(o2, off)
        np.testing.assert_array_equal(i2, ids)
        np.testing.assert_array_equal(d2.view(np.int64), np.asarray(w, np.float64).view(np.int64))
        enc = orc.ppr_encode(d2)
        np.testing.assert_array_equal(enc.view(np.int64), ((w + 0.1) / (w.max() + 0.1)).view(np.int64))   # utils.py:35-36
