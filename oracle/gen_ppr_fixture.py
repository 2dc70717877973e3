#!/usr/bin/env python3
"""Generate tests/golden/ppr_push.npz -- TEST INFRASTRUCTURE, run in the build container.

The reference's PPR push (sampler/pprgo.py:9-38, `_calc_ppr_node`) is numba code; numba is not in the image, so the
reference itself cannot produce a vector for it (parity of the PPR sampler is UNPINNED).  What can be produced
without numba is the same push with the SAME dict / LIFO-list semantics and explicit NumPy scalar types where numba's
typing says float32 / float64 -- an independent second statement of the algorithm (pure Python dicts, not the C hash
tables of oracle/ppr_oracle.c), evaluated here once and committed as data.  tests/test_ppr_oracle_cpu.py checks
oracle/ppr_oracle.c against it bit for bit (score bit patterns), the GPU tests check csrc/ppr.hip against the oracle.
"""
import os

import numpy as np
import scipy.sparse as sps

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ppr_push.npz")


def push_scores(inode, indptr, indices, deg, alpha, epsilon):
    """approximate PPR of one root by residual pushes with a LIFO work list; float32 p, r, alpha, epsilon and
    (1 - alpha) * res / deg evaluated in float64 then rounded on the store, as numba types pprgo.py:9-38"""
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


def graph(N=400, E=1400, seed=0):
    rng = np.random.default_rng(seed)
    r, c = rng.integers(0, N, E), rng.integers(0, N, E)
    A = sps.csr_matrix((np.ones(2 * E), (np.r_[r, c], np.r_[c, r])), shape=(N, N))
    A.setdiag(0)
    A.eliminate_zeros()
    A.data[:] = 1
    A.sort_indices()
    return A


if __name__ == "__main__":
    A = graph()
    deg = np.diff(A.indptr)
    roots = np.array([0, 1, 7, 99, 399], np.int32)
    blob = {"indptr": A.indptr.astype(np.int32), "indices": A.indices.astype(np.int32), "roots": roots,
            "params": np.array([[0.5, 1e-4], [0.15, 1e-3]], np.float64)}
    for j, (alpha, eps) in enumerate(blob["params"]):
        off, ids, bits = [0], [], []
        for s in roots:
            keys, pv = push_scores(int(s), A.indptr, A.indices, deg, alpha, eps)
            order = np.argsort(keys)
            ids.append(np.asarray(keys, np.int32)[order])
            bits.append(np.asarray(pv, np.float32)[order].view(np.int32))
            off.append(off[-1] + len(keys))
        blob[f"off_{j}"], blob[f"ids_{j}"], blob[f"score_bits_{j}"] = np.asarray(off, np.int64), np.concatenate(ids), np.concatenate(bits)
    np.savez_compressed(OUT, **blob)
    print(f"wrote {OUT}: {os.path.getsize(OUT)} bytes")
