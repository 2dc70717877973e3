"""CPU: oracle/ppr_oracle.c (the checker of the GPU PPR sampler).  The reference function is numba code that cannot
run here (parity UNPINNED), so the restatement is pinned to what can be checked without it: the defining property of
the Andersen-Chung-Lang approximation against an exact PPR, the dict/list semantics of sampler/pprgo.py:9-38 against a
literal Python transcription with NumPy float32 scalars where numba's typing and NumPy's agree, and the
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


def _calc_ppr_node(inode, indptr, indices, deg, alpha, epsilon):
    """pprgo.py:9-38 transcribed; float32 where numba says float32, float64 where it says float64."""
    f32 = np.float32
    alpha, epsilon = f32(alpha), f32(epsilon)
    alpha_eps = f32(alpha * epsilon)
    p, r, q = {inode: f32(0)}, {inode: alpha}, [inode]
    while q:
        unode = q.pop()
        res = r.get(unode, f32(0))
        p[unode] = f32(p.get(unode, f32(0)) + res)
        r[unode] = f32(0)
        for vnode in indices[indptr[unode]:indptr[unode + 1]]:
            vnode = int(vnode)
            _val = f32((1.0 - float(alpha)) * float(res) / float(deg[unode]))
            r[vnode] = f32(r.get(vnode, f32(0)) + _val)
            if float(r[vnode]) >= float(alpha_eps) * float(deg[vnode]) and vnode not in q:
                q.append(vnode)
    return list(p.keys()), list(p.values())


def test_oracle_matches_python_transcription():
    A = _graph()
    deg = np.diff(A.indptr)
    roots = np.array([0, 1, 7, 99, 399], np.int32)
    for alpha, eps in ((0.5, 1e-4), (0.15, 1e-3)):
        off, ids, vals, _ = orc.ppr_topk(A.indptr, A.indices, roots, alpha, eps, A.shape[0], table_log2=12)
        for i, s in enumerate(roots):
            keys, pv = _calc_ppr_node(int(s), A.indptr, A.indices, deg, alpha, eps)
            order = np.argsort(keys)
            np.testing.assert_array_equal(ids[off[i]:off[i + 1]], np.asarray(keys)[order])
            np.testing.assert_array_equal(vals[off[i]:off[i + 1]].view(np.int32),
                                          np.asarray(pv, np.float32)[order].view(np.int32))


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
