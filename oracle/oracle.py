"""ctypes front-end of oracle/subgacc_oracle.c plus an independent NumPy restatement of SpJoin.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Every function names the reference lines it
follows (paths relative to /root/reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RNG_RAND_R = 0
RNG_PHILOX = 1
_RNG = {"rand_r": RNG_RAND_R, "philox": RNG_PHILOX, RNG_RAND_R: RNG_RAND_R, RNG_PHILOX: RNG_PHILOX}


def build(force=False):
    """gcc-compile the C restatement into oracle/_build/liboracle.so (and oracle/_ref when the
    reference tree is present -- build container only)."""
    so = os.path.join(_HERE, "_build", "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("subgacc_oracle.c", "ppr_oracle.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if os.path.exists("/root/reference/subg_acc/subg_acc.c"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_sjoin_count.restype = C.c_int64
    return _LIB


def ref_module():
    """The compiled reference extension (oracle/_ref), or None when it was never built."""
    import glob
    import importlib.util

    hits = glob.glob(os.path.join(_HERE, "_ref", "subg_acc*.so"))
    if not hits:
        return None
    import sys
    prev = sys.modules.get("subg_acc")          # the drop-in mirror at the repo root has the same module name:
    spec = importlib.util.spec_from_file_location("subg_acc", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if prev is not None:                        # a single-phase extension registers itself in sys.modules -- the
        sys.modules["subg_acc"] = prev          # checker must never shadow (or be mistaken for) the product module
    else:
        sys.modules.pop("subg_acc", None)
    return mod


class _GsetResult(C.Structure):
    _fields_ = [("nsize", C.POINTER(C.c_int32)), ("ids", C.POINTER(C.c_int32)), ("sf", C.POINTER(C.c_int32)),
                ("enc", C.POINTER(C.c_int16)), ("raw", C.POINTER(C.c_int16)),
                ("X", C.c_int64), ("c", C.c_int64), ("n_overflow", C.c_int64)]


class _WalkResult(C.Structure):
    _fields_ = [("walks", C.POINTER(C.c_int32)), ("nsize", C.POINTER(C.c_int32)), ("ids", C.POINTER(C.c_int32)),
                ("counts", C.POINTER(C.c_int32)), ("X", C.c_int64)]


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _take(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _csr(indptr, indices):
    return (np.ascontiguousarray(indptr, dtype=np.int64), np.ascontiguousarray(indices, dtype=np.int32))


def rand_r_stream(seed, count):
    out = np.empty(count, np.uint32)
    lib().orc_rand_r_stream(C.c_uint32(seed & 0xFFFFFFFF), C.c_int64(count), _p(out, C.c_uint32))
    return out


def philox4x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, np.uint32)
    key = np.ascontiguousarray(key, np.uint32)
    out = np.empty(4, np.uint32)
    lib().orc_philox4x32_10(_p(ctr, C.c_uint32), _p(key, C.c_uint32), _p(out, C.c_uint32))
    return out


def philox2x32_10(ctr, key):
    ctr = np.ascontiguousarray(ctr, np.uint32)
    out = np.empty(2, np.uint32)
    lib().orc_philox2x32_10(_p(ctr, C.c_uint32), C.c_uint32(int(key) & 0xFFFFFFFF), _p(out, C.c_uint32))
    return out


def gset_sampler(indptr, indices, query, num_walks=100, num_steps=3, bucket=-1, seed=111413, rng="rand_r",
                 nthreads=1, debug=False):
    """subg_acc.c:649-1034.  Returns [nsize int32[n], remap int32[2,X], enc int16[c,m+1]] (+ raw when debug)."""
    ip, ix = _csr(indptr, indices)
    q = np.ascontiguousarray(np.asarray(query).astype(np.int32))
    res = _GsetResult()
    rc = lib().orc_gset_sampler(_p(ip, C.c_int64), _p(ix, C.c_int32), _p(q, C.c_int32), C.c_int64(q.size),
                                C.c_int(num_walks), C.c_int(num_steps), C.c_int(bucket), C.c_uint32(seed & 0xFFFFFFFF),
                                C.c_int(_RNG[rng]), C.c_int(nthreads), C.byref(res))
    if rc == -2:
        raise AssertionError("Longer width of type for hasing key needed > INT64.")
    if rc != 0:
        raise MemoryError("oracle gset_sampler failed")
    ncol = num_steps + 1
    X, c = res.X, res.c
    nsize = _take(res.nsize, q.size, np.int32)
    remap = np.stack([_take(res.ids, X, np.int32), _take(res.sf, X, np.int32)])
    enc = _take(res.enc, c * ncol, np.int16).reshape(c, ncol)
    raw = _take(res.raw, X * ncol, np.int16).reshape(X, ncol)
    lib().orc_gset_free(C.byref(res))
    return [nsize, remap, enc, raw] if debug else [nsize, remap, enc]


def walk_sampler(ptr, neighs, query, num_walks=100, num_steps=3, nthread=1, seed=111413, replacement=False,
                 rng="rand_r"):
    """subg_acc.c:316-389.  `replacement=True` selects the WITHOUT-replacement first hop, as the
    reference does (:354-362).  Returns (walks int32[n, M*(m+1)], nsize, ids, counts int32[X, m+1])."""
    ip, ix = _csr(ptr, neighs)
    q = np.ascontiguousarray(np.asarray(query).astype(np.int32))
    res = _WalkResult()
    rc = lib().orc_walk_sampler(_p(ip, C.c_int64), _p(ix, C.c_int32), _p(q, C.c_int32), C.c_int64(q.size),
                                C.c_int(num_walks), C.c_int(num_steps), C.c_uint32(seed & 0xFFFFFFFF),
                                C.c_int(nthread), C.c_int(1 if replacement else 0), C.c_int(_RNG[rng]), C.byref(res))
    if rc != 0:
        raise MemoryError("oracle walk_sampler failed")
    W = num_walks * (num_steps + 1)
    walks = _take(res.walks, q.size * W, np.int32).reshape(q.size, W)
    nsize = _take(res.nsize, q.size, np.int32)
    ids = _take(res.ids, res.X, np.int32)
    counts = _take(res.counts, res.X * (num_steps + 1), np.int32).reshape(res.X, num_steps + 1)
    lib().orc_walk_free(C.byref(res))
    return walks, nsize, ids, counts


def spg_build(nsize, remap, nthreads=1):
    """sampler/random_walks.py:79-80 with rows = query positions: (indptr int64, indices int32, data int32)."""
    nsize = np.ascontiguousarray(nsize, np.int32)
    ids = np.ascontiguousarray(remap[0], np.int32)
    sf = np.ascontiguousarray(remap[1], np.int32)
    n, X = nsize.size, ids.size
    indptr = np.empty(n + 1, np.int64)
    indices = np.empty(X, np.int32)
    data = np.empty(X, np.int32)
    rc = lib().orc_spg_build(_p(nsize, C.c_int32), C.c_int64(n), _p(ids, C.c_int32), _p(sf, C.c_int32),
                             _p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_int32), C.c_int(nthreads))
    assert rc == 0
    return indptr, indices, data


def enc_table(enc):
    """random_walks.py:81: prepend the all-zero 'absent' row."""
    enc = np.asarray(enc)
    return np.concatenate([np.zeros((1, enc.shape[1]), enc.dtype), enc], axis=0)


def sjoin(indptr, indices, data, own, partner, nthreads=1):
    """Generic segment join (C merge join).  Returns (seg int64[S+1], pairs) where pairs is
    int32[R,2] for integer SpG data and float32[R,2] for float64 data (train.py:13-45,48-72)."""
    indptr = np.ascontiguousarray(indptr, np.int64)
    indices = np.ascontiguousarray(indices, np.int32)
    own = np.ascontiguousarray(own, np.int64)
    partner = np.ascontiguousarray(partner, np.int64)
    S = own.size
    seg = np.empty(S + 1, np.int64)
    R = lib().orc_sjoin_count(_p(indptr, C.c_int64), _p(own, C.c_int64), C.c_int64(S), _p(seg, C.c_int64))
    data = np.ascontiguousarray(data)
    if data.dtype == np.float64:
        out = np.empty((R, 2), np.float32)
        lib().orc_sjoin_fill(_p(indptr, C.c_int64), _p(indices, C.c_int32), None, _p(data, C.c_double),
                             _p(own, C.c_int64), _p(partner, C.c_int64), C.c_int64(S), _p(seg, C.c_int64),
                             None, _p(out, C.c_float), C.c_int(nthreads))
    else:
        data = data.astype(np.int32, copy=False)
        out = np.empty((R, 2), np.int32)
        lib().orc_sjoin_fill(_p(indptr, C.c_int64), _p(indices, C.c_int32), _p(data, C.c_int32), None,
                             _p(own, C.c_int64), _p(partner, C.c_int64), C.c_int64(S), _p(seg, C.c_int64),
                             _p(out, C.c_int32), None, C.c_int(nthreads))
    return seg, out


def pair_segments(edge):
    """own/partner lists of gather(): left blocks then right blocks (train.py:15,33-37)."""
    edge = np.asarray(edge, np.int64)
    return np.concatenate([edge[0], edge[1]]), np.concatenate([edge[1], edge[0]])


def triplet_segments(hedge):
    """own/partner lists of hgather(): [U|w ; W|u ; V|w ; W|v] (train.py:50-67)."""
    h = np.asarray(hedge, np.int64)
    u, v, w = h[0], h[1], h[2]
    return np.concatenate([u, w, v, w]), np.concatenate([w, u, w, v])


def gather(edge, spg, ptr=True, encode=None, nthreads=1):
    """train.py:13-45 on an SpG given as (indptr, indices, data).  Returns (xz float32, indptr-or-segment-ids int64)."""
    own, partner = pair_segments(edge)
    return _finish(spg, own, partner, ptr, encode, nthreads)


def hgather(hedge, spg, encode, nthreads=1):
    """train.py:48-72: always segment ids, encode mandatory."""
    own, partner = triplet_segments(hedge)
    return _finish(spg, own, partner, False, encode, nthreads)


def _finish(spg, own, partner, ptr, encode, nthreads):
    indptr, indices, data = spg
    seg, pairs = sjoin(indptr, indices, data, own, partner, nthreads)
    if encode is not None:
        xz = np.asarray(encode, np.float32)[pairs]            # [R,2,k]
    else:
        xz = pairs.astype(np.float32)[..., None]              # [R,2,1]
    if ptr:
        return xz, seg
    return xz, np.repeat(np.arange(own.size, dtype=np.int64), np.diff(seg))


def gather_numpy(edge, spg, ptr=True, encode=None):
    """Independent pure-NumPy restatement of SURVEY.md section 3.3 (searchsorted per pair); used to
    cross-check the C merge join on small cases."""
    indptr, indices, data = spg
    edge = np.asarray(edge, np.int64)
    B = edge.shape[1]
    isf = np.asarray(data).dtype == np.float64
    blocks, sizes = [], []
    for side in (0, 1):
        for b in range(B):
            a, p = edge[side, b], edge[1 - side, b]
            ia, va = indices[indptr[a]:indptr[a + 1]], data[indptr[a]:indptr[a + 1]]
            ip_, vp = indices[indptr[p]:indptr[p + 1]], data[indptr[p]:indptr[p + 1]]
            pos = np.searchsorted(ip_, ia)
            pos_c = np.minimum(pos, max(len(ip_) - 1, 0))
            hit = (pos < len(ip_)) & (ip_[pos_c] == ia) if len(ip_) else np.zeros(len(ia), bool)
            if isf:
                second = np.where(hit, vp[pos_c] if len(ip_) else 0.0, 0.0)
                second = (second + 1.0) - 1.0
                blk = np.stack([va.astype(np.float64), second], axis=1).astype(np.float32)
            else:
                second = np.where(hit, vp[pos_c] if len(ip_) else 0, 0)
                blk = np.stack([va, second], axis=1).astype(np.int32)
            blocks.append(blk)
            sizes.append(len(ia))
    pairs = np.concatenate(blocks) if blocks else np.zeros((0, 2), np.float32 if isf else np.int32)
    xz = np.asarray(encode, np.float32)[pairs] if encode is not None else pairs.astype(np.float32)[..., None]
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    if ptr:
        return xz, seg
    return xz, np.repeat(np.arange(2 * B, dtype=np.int64), sizes)


def num_threads():
    return int(lib().orc_num_threads())


# ------------------------------------------------------------------ top-K PPR sets (oracle/ppr_oracle.c; parity UNPINNED)
def ppr_topk(indptr, indices, roots, alpha, epsilon, topk, table_log2=20):
    """sampler/pprgo.py:9-38,53-82: rows of the top-`topk` approximate-PPR entries of every root, sorted by node id.
    Returns (row_off int64[n+1], ids int32[X], vals float32[X], pushes)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    roots = np.ascontiguousarray(roots, dtype=np.int32)
    n = len(roots)
    cnt = np.zeros(n, np.int32)
    ids = np.zeros(n * topk, np.int32)
    vals = np.zeros(n * topk, np.float32)
    pushes = C.c_int64(0)
    rc = lib().orc_ppr_topk(_p(indptr, C.c_int64), _p(indices, C.c_int32), _p(roots, C.c_int32), C.c_int64(n),
                            C.c_float(alpha), C.c_float(epsilon), C.c_int32(topk), C.c_int32(table_log2),
                            _p(cnt, C.c_int32), _p(ids, C.c_int32), _p(vals, C.c_float), C.byref(pushes))
    if rc:
        raise MemoryError(f"orc_ppr_topk failed ({rc})")
    off = np.zeros(n + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    keep = (np.arange(topk)[None, :] < cnt[:, None]).ravel()
    return off, ids[keep], vals[keep], pushes.value


_PPR_NORM = {"row": 0, "sym": 1, "col": 2}


def topk_ppr_matrix(indptr, indices, alpha, eps, idx, topk, normalization="row", table_log2=20):
    """sampler/pprgo.py:85-111 on an unweighted CSR graph: (row_off, ids, data float64)."""
    if normalization not in _PPR_NORM:
        raise ValueError(f"Unknown PPR normalization: {normalization}")
    off, ids, vals, _ = ppr_topk(indptr, indices, idx, alpha, eps, topk, table_log2)
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    roots = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.zeros(len(ids), np.float64)
    lib().orc_ppr_normalize(_p(indptr, C.c_int64), _p(roots, C.c_int32), C.c_int64(len(roots)), _p(off, C.c_int64),
                            _p(ids, C.c_int32), _p(vals, C.c_float), C.c_int(_PPR_NORM[normalization]), _p(out, C.c_double))
    return off, ids, out


def ppr_encode(data):
    """utils.py:35-36 (encoding 'PPR'): (x + 0.1) / (max + 0.1), float64."""
    data = np.array(data, dtype=np.float64, copy=True)
    lib().orc_ppr_encode(_p(data, C.c_double), C.c_int64(len(data)))
    return data


# ------------------------------------------------------------------ walk_sampler route to the SpG (random_walks.py:35-71)
def np_sampling(ptr, neighs, bsize, target, num_walks=200, num_steps=4, nthread=1, seed=111413, sampler=None):
    """sampler/random_walks.py:35-47.  `sampler` = a walk_sampler with the reference's return shape ([walks, obj]);
    default: this module's C restatement (pinned to the reference's goldens)."""
    key, freq = [], []
    target = np.asarray(target)
    for lo in range(0, len(target), bsize):                     # gen_batch(target, bsize, keep=True), :25-29
        batch = target[lo:lo + bsize]
        if sampler is None:
            _, nsize, ids, counts = walk_sampler(ptr, neighs, batch, num_walks=num_walks, num_steps=num_steps,
                                                 nthread=nthread, seed=seed, replacement=True)
            key.append(ids)
            freq.append(counts)
        else:
            _, freqs = sampler(ptr, neighs, batch, num_walks=num_walks, num_steps=num_steps, nthread=nthread, seed=seed,
                               replacement=True)
            key.append(np.concatenate(list(freqs[:, 0])))
            freq.append(np.vstack(list(freqs[:, 1])))
    return np.concatenate(key), np.vstack(freq)


def rw_matrix(indptr, indices, train_idx, num_walks=200, num_steps=4, batch_size=2000, reduced=True, nthread=1,
              seed=111413, sampler=None):
    """sampler/random_walks.py:58-71 with np.unique standing in for fastremap.unique (same contract: sorted distinct
    values + index of the first occurrence).  Returns (scipy csr z with rows = positions in train_idx, freqs)."""
    import scipy.sparse as sps
    gsize = len(indptr) - 1
    neighbors, freqs = np_sampling(indptr, indices, batch_size, train_idx, num_walks=num_walks, num_steps=num_steps - 1,
                                   nthread=nthread, seed=seed, sampler=sampler)
    # sizes of the sets, in order (construct_sparse, :50-55, takes the ragged list; np_sampling above returns it flat)
    sizes = []
    for lo in range(0, len(train_idx), batch_size):
        batch = np.asarray(train_idx)[lo:lo + batch_size]
        sizes.append(walk_sampler(indptr, indices, batch, num_walks=num_walks, num_steps=num_steps - 1, nthread=nthread,
                                  seed=seed, replacement=True)[1])
    sizes = np.concatenate(sizes) if sizes else np.zeros(0, np.int32)
    if reduced:
        proj = np.array([(num_walks + 1) ** i for i in reversed(range(num_steps))], dtype=np.int64)
        idy = freqs.astype(np.int64) @ proj
        val, idx = np.unique(idy, return_index=True)
        idy = np.searchsorted(val, idy)
        freqs = freqs[idx]
    else:
        idy = np.arange(len(freqs))
    i = np.repeat(np.arange(len(sizes)), sizes)
    z = sps.csr_matrix((idy + 1, (i, neighbors)), shape=(gsize, gsize))
    freqs = np.insert(freqs, 0, np.zeros((1, num_steps), freqs.dtype), axis=0)
    return z, freqs


# ------------------------------------------------------------------ walk_join (subg_acc.c:509-647, legacy SUREL)
def walk_join(walk, key, query, nthread=-1, return_idx=False):
    """Returns out int32[2, Q*2*stride] (and xrow int32[Q,2] with return_idx), as the reference does."""
    walk = np.ascontiguousarray(walk, np.int32)
    n = walk.shape[0]
    stride = int(np.prod(walk.shape[1:]))
    walk = walk.reshape(n, stride)
    key = [np.asarray(k, np.int32).ravel() for k in key]
    assert len(key) == n, "Dims do not match between num of walks and keys."
    off = np.zeros(n + 1, np.int64)
    np.cumsum([len(k) for k in key], out=off[1:])
    ids = np.ascontiguousarray(np.concatenate(key) if n else np.zeros(0, np.int32), np.int32)
    q = np.ascontiguousarray(np.asarray(query).astype(np.int32)).reshape(-1, 2)
    Q = q.shape[0]
    out = np.empty((2, Q * 2 * stride), np.int32)
    xrow = np.empty((Q, 2), np.int32)
    rc = lib().orc_walk_join(_p(walk, C.c_int32), C.c_int64(n), C.c_int32(stride), _p(off, C.c_int64), _p(ids, C.c_int32),
                             _p(q, C.c_int32), C.c_int64(Q), _p(out, C.c_int32), _p(xrow, C.c_int32))
    assert rc == 0
    return [out, xrow] if return_idx else out


def batch_sampler(ptr, neighs, query, num_walks=200, num_steps=8, thld=1000, seed_eff=111413):
    """subg_acc.batch_sampler (subg_acc.c:391-507) with the rand_r state given outright: the reference starts from
    seed + getpid(), pass that sum as `seed_eff`.  Returns int32[#unique] in insertion order."""
    ptr = np.ascontiguousarray(ptr, np.int64)
    neighs = np.ascontiguousarray(neighs, np.int32)
    q = np.ascontiguousarray(np.asarray(query).astype(np.int32)).ravel()
    n = q.size
    cap = int(n * (num_walks * num_steps + 1)) + 1
    cap = min(cap, int(ptr.size - 1) + 1)            # a set of node ids never exceeds the node count
    out = np.empty(max(cap, 1), np.int32)
    f = lib().orc_batch_sampler
    f.restype = C.c_int64
    rc = f(_p(ptr, C.c_int64), _p(neighs, C.c_int32), _p(q, C.c_int32), C.c_int64(n), C.c_int(num_walks), C.c_int(num_steps),
           C.c_int(thld), C.c_uint32(int(seed_eff) & 0xFFFFFFFF), _p(out, C.c_int32), C.c_int64(out.size))
    assert rc >= 0, rc
    return out[:rc].copy()


# ------------------------------------------------------------------ DEG / SPD encoders (utils.py:22-34) with SciPy
def encoding_scipy(x, adj, encoding):
    """The reference's few lines of sparse algebra, executed by SciPy itself (utils.py cannot be imported here: its
    module header pulls in torch_geometric).  x, adj: scipy CSR.  Returns (x', agg) with canonical (sorted) rows."""
    import scipy.sparse as sps
    from sklearn.preprocessing import normalize
    x = sps.csr_matrix(x, copy=True)
    agg = None
    if encoding == "DEG":                                            # utils.py:24-30
        x += normalize(adj, norm="l1", axis=1)
        x_deg = x.getnnz(axis=1)
        x_deg = np.log(x_deg + 1)
        agg = x.copy()
        x.data = (x > 0).multiply(x_deg).data
    elif encoding == "SPD":                                          # utils.py:31-36
        x0 = x > 0
        x1 = adj > 0
        x2 = x1 ** 2
        x = x1 + x0.multiply(x2 * 0.5) + x0 * 0.3
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x.setdiag(2.3)
    else:
        raise NotImplementedError
    x = sps.csr_matrix(x)
    x.sort_indices()
    if agg is not None:
        agg = sps.csr_matrix(agg)
        agg.sort_indices()
    return x, agg
