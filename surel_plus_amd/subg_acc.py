"""Host mirror of the reference's `subg_acc` extension module (subg_acc/subg_acc.c:1036-1059).

Same function names, keyword arguments, defaults, return conventions and dtypes as the CPython module:
NumPy (host) arrays in, a list of NumPy arrays out, with the sampling itself done by the HIP kernels.
`run` (a `system()` wrapper, subg_acc.c:119-128) is intentionally not provided.

Extensions over the reference: `rng="philox"` (schedule-independent counter RNG), int64 `indptr`.
`nthread` keeps its place in the signature: for gset_sampler it is ignored (the kernels always reproduce
the deterministic nthread=1 stream; the reference's multi-thread output is a data race, subg_acc.c:731-732),
for walk_sampler it selects the number of rand_r streams exactly like the OpenMP team size does.
"""
import numpy as np
import torch

from . import _lib
from .sampler import DeviceCSR, sample_sets

__all__ = ["gset_sampler", "walk_sampler", "walk_join", "batch_sampler", "add", "sjoin"]


def _csr_from_host(indptr, indices):
    ip = np.asarray(indptr)
    ix = np.asarray(indices)
    # PyArray_FROM_OTF(..., NPY_INT, NPY_ARRAY_IN_ARRAY) is a SAFE cast (subg_acc.c:663,668)
    if ip.dtype != np.int64 and not np.can_cast(ip.dtype, np.int32, "safe"):
        raise TypeError("Input parsing error. (indptr cannot be safely cast to int32)")
    if not np.can_cast(ix.dtype, np.int32, "safe"):
        raise TypeError("Input parsing error. (indices cannot be safely cast to int32)")
    ip = np.ascontiguousarray(ip, dtype=np.int64 if ip.dtype == np.int64 else np.int32)
    ix = np.ascontiguousarray(ix, dtype=np.int32)
    if ip.ndim != 1 or ip.size < 1 or ix.ndim != 1:
        raise TypeError("Input parsing error. (indptr / indices must be 1-D, indptr non-empty)")
    # The reference reads whatever the CSR says (no bounds checks, a bad graph is a segfault); on the GPU a stray
    # address can take the device down.  DeviceCSR validates the arrays on the device right after the upload (a few
    # reductions over HBM-resident data) -- not with O(nnz) NumPy passes on the host in every call.
    return DeviceCSR(ip, ix)


def _checked_query(query, num_nodes):
    q = np.asarray(query)
    if q.size and (int(q.min()) < 0 or int(q.max()) >= num_nodes):
        raise IndexError(f"query node ids outside [0, {num_nodes})")
    return q


def gset_sampler(indptr, indices, query, num_walks=100, num_steps=3, bucket=-1, nthread=-1, seed=111413, debug=-1,
                 rng="rand_r"):
    """subg_acc.c:649-1034.  Returns [nsize int32[n], remap int32[2,X], enc int16[c,num_steps+1]]
    (+ raw_enc int16[X,num_steps+1] when debug > 0)."""
    csr = indptr if isinstance(indptr, DeviceCSR) else _csr_from_host(indptr, indices)
    if not torch.is_tensor(query):
        query = _checked_query(query, csr.num_nodes)
    sets = sample_sets(csr, query, num_walks=num_walks, num_steps=num_steps, bucket=bucket, seed=seed, rng=rng)
    # the hand-over (subg_acc.c:1017-1024): ids and SFptr go straight into the two rows of ONE pinned [2, X] host array, each by
    # its own asynchronous copy; nsize and enc ride behind them on the same stream -- one wait for all four
    sf = sets.get_sf()
    enc_dev = sets.enc_int16()
    raw_dev = enc_dev[sf.long()] if debug > 0 else None
    remap, (nsize, enc, raw) = _rows_to_host([sets.ids, sf], [sets.nsize, enc_dev, raw_dev])
    if _lib.VERBOSE:
        print(f"#SubGAcc: #total {sets.X}; #enc_unique {sets.c}; compression ratio {sets.X / max(sets.c, 1):.2f}")
    if debug > 0:
        return [nsize, remap, enc, raw]
    return [nsize, remap, enc]


PINNED_LIMIT = 256 << 20      # bytes: results beyond this leave through pageable memory


def _rows_to_host(rows, others=()):
    """device rows (1-D, same dtype and length) -> one NumPy [len(rows), X] array over pinned memory, each row copied by its own
    async D2H (no stacked device copy, no pageable staging); `others`: more device tensors (or None) brought along the same way.
    The arrays keep their pinned blocks alive; torch's host allocator takes the blocks back when they are dropped -- but it rounds a
    request up to a power of two and never returns a block to the OS, and these results are the caller's to keep (the offline stage
    over all nodes hands over GBs): an array above PINNED_LIMIT is a pageable one (a slower copy, no page-locked memory left behind)."""
    host = None
    if rows:
        big = len(rows) * rows[0].numel() * rows[0].element_size() > PINNED_LIMIT
        host = torch.empty((len(rows), rows[0].numel()), dtype=rows[0].dtype, pin_memory=not big)
        for i, r in enumerate(rows):
            host[i].copy_(r, non_blocking=not big)
    extra = []
    for t in others:
        if t is None:
            extra.append(None)
            continue
        big = t.numel() * t.element_size() > PINNED_LIMIT
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=not big)
        h.copy_(t, non_blocking=not big)
        extra.append(h)
    devs = {t.device for t in list(rows) + [t for t in others if t is not None] if t.is_cuda}
    for dev in devs:           # (every copy went to the current stream of ITS tensor's device, which need not be the current device)
        torch.cuda.current_stream(dev).synchronize()
    return (None if host is None else host.numpy()), [None if h is None else h.numpy() for h in extra]


def walk_sampler(ptr, neighs, query, num_walks=100, num_steps=3, nthread=-1, seed=111413, replacement=False,
                 rng="rand_r"):
    """subg_acc.c:316-389.  Returns [walks int32[n, M*(m+1)], obj] with obj an object array [n,2] of
    (ids int32[count], counts int32[count, m+1]).  As in the reference, `replacement=True` selects the
    first hop WITHOUT replacement (subg_acc.c:354-362)."""
    csr = ptr if isinstance(ptr, DeviceCSR) else _csr_from_host(ptr, neighs)
    if not torch.is_tensor(query):
        query = _checked_query(query, csr.num_nodes)
    sets = sample_sets(csr, query, num_walks=num_walks, num_steps=num_steps, seed=seed, rng=rng,
                       first_hop_wo=bool(replacement), order=_lib.ORDER_STEP_MAJOR, cap_root_degree=False,
                       emit_walks=True, rng_streams=max(int(nthread), 1), dedup=False)
    _, (walks, off, ids, counts) = _rows_to_host([], [sets.walks, sets.row_off, sets.ids, sets.counts_int32()])
    n = len(off) - 1
    # rpe_encoder's per-root (ids, counts) pairs (subg_acc.c:371-380) as views of the two packed arrays, cut in one C loop
    obj = np.empty((n, 2), dtype=object)
    if n:
        lo, hi = off[:-1].astype(object), off[1:].astype(object)
        obj[:, 0] = np.frompyfunc(lambda a, b: ids[a:b], 2, 1)(lo, hi)
        obj[:, 1] = np.frompyfunc(lambda a, b: counts[a:b], 2, 1)(lo, hi)
    return [walks, obj]


def walk_join(walk, key, query, nthread=-1, return_idx=False):
    """subg_acc.c:509-647 (legacy SUREL join).  walk int32 [n, stride] or [n, M, m+1] -- walk_sampler's first output;
    key -- n sequences of node ids (walk_sampler's obj[:, 0]); query -- [Q, 2] node ids, each a root of `walk`.
    Returns out int32 [2, Q*2*stride] (and, with return_idx, int32 [Q, 2] row numbers of the query keys).  For each
    pair and walk position: (index of the node in key1's list, in key2's list), 1-based over the concatenation of all
    lists, 0 = absent; out[0] follows key1's walk, out[1] key2's.  A repeated root resolves to its LAST row, as the
    reference's hash does; a query key that is no root yields -1s (the reference reads out of bounds)."""
    from ._lib import check, lib, ptr, stream_ptr
    dev = _lib.require_device()
    w = walk if torch.is_tensor(walk) else torch.from_numpy(np.ascontiguousarray(np.asarray(walk), dtype=np.int32))
    if w.dim() < 2:
        raise TypeError("Input parsing error. (walk must be [n, stride] or [n, M, m+1])")
    n = w.shape[0]
    w = w.to(device=dev, dtype=torch.int32).reshape(n, -1).contiguous()
    stride = w.shape[1]
    key = list(key)
    if len(key) != n:
        raise AssertionError("Dims do not match between num of walks and keys.")
    lens = np.fromiter((len(k) for k in key), dtype=np.int64, count=n)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    X = int(off[-1])
    ids_h = np.concatenate([np.asarray(k).astype(np.int32, copy=False).ravel() for k in key]) if X else np.zeros(0, np.int32)
    row_off = torch.from_numpy(off).to(dev)
    ids = torch.from_numpy(np.ascontiguousarray(ids_h, dtype=np.int32)).to(dev)
    max_len = int(lens.max()) if n else 0
    st = stream_ptr()
    # the key lists become SpG-form rows: ids ascending, payload = 1 + position in the concatenation (:573-584)
    set_ids, set_idx = torch.empty_like(ids), torch.empty_like(ids)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    if X:
        pos = torch.arange(X, dtype=torch.int32, device=dev)
        check(lib().subgacc_spg_build(ptr(row_off), n, ptr(ids), ptr(pos), None, 0, max_len, ptr(set_ids), ptr(set_idx),
                                      ptr(flags), st))
    q = torch.as_tensor(np.asarray(query) if not torch.is_tensor(query) else query).to(device=dev, dtype=torch.int32).reshape(-1, 2)
    Q = q.shape[0]
    if n:      # find_key_item (:617): row of the root, the last one among equals
        roots_sorted, order = torch.sort(w[:, 0].contiguous(), stable=True)
        p = torch.searchsorted(roots_sorted, q.reshape(-1).contiguous(), right=True) - 1
        hit = (p >= 0) & (roots_sorted[p.clamp(min=0)] == q.reshape(-1))
        qrow = torch.where(hit, order[p.clamp(min=0)], torch.full_like(p, -1)).to(torch.int32).reshape(Q, 2).contiguous()
    else:
        qrow = torch.full((Q, 2), -1, dtype=torch.int32, device=dev)
    out = torch.empty((2, Q * 2 * stride), dtype=torch.int32, device=dev)
    check(lib().subgacc_walk_join(ptr(w), n, stride, ptr(row_off), ptr(set_ids), ptr(set_idx), max_len, ptr(qrow), Q,
                                  ptr(out), st))
    out = out.cpu().numpy()
    return [out, qrow.cpu().numpy()] if return_idx else out


def batch_sampler(ptr, neighs, query, num_walks=200, num_steps=8, thld=1000, nthread=-1, seed=111413, pid=None):
    """subg_acc.c:391-507 (legacy SUREL mini-batch former).  Returns int32 [#unique]: the nodes met by walking the roots
    one after the other (first hop without replacement, num_steps nodes per walk), in order of first appearance; root i
    stops walking once the set holds (i+1)*thld/n nodes.  As in the reference the rand_r stream starts from
    seed + getpid() (:421) -- results differ from process to process by design; `pid` (extension) pins that term so that a
    run can be reproduced (`pid=0`: the stream starts from `seed` itself).  `nthread` is accepted and unused, as in the
    reference (the loop is sequential there too).  The graph must have no reachable dead ends (SubgAccError otherwise:
    a node without out-edges draws nothing, which makes every later position of the stream data dependent)."""
    import os

    from ._lib import check, lib, stream_ptr
    from ._lib import ptr as dptr
    csr = ptr if isinstance(ptr, DeviceCSR) else _csr_from_host(ptr, neighs)
    if num_walks <= 0 or num_steps <= 0:
        raise TypeError("Input parsing error. (num_walks and num_steps must be positive)")
    dev = csr.device
    if torch.is_tensor(query):
        q = query.to(device=dev, dtype=torch.int32).contiguous().view(-1)
    else:
        q = torch.from_numpy(np.ascontiguousarray(np.asarray(_checked_query(query, csr.num_nodes)).astype(np.int32)).ravel()).to(dev)
    n = q.numel()
    seed_eff = (int(seed) + (os.getpid() if pid is None else int(pid))) & 0xFFFFFFFF
    # every walk adds at most num_steps nodes and a root stops at (i+1)*thld/n: |set| <= thld + n*(num_steps+1)
    cap = max(1, min(csr.num_nodes, n * (int(num_walks) * int(num_steps) + 1), max(int(thld), 0) + n * (int(num_steps) + 1) + 1))
    L = lib()
    out = torch.empty(cap, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    ws = torch.empty(L.subgacc_batch_sampler_workspace_bytes(cap), dtype=torch.uint8, device=dev)
    check(L.subgacc_batch_sampler(dptr(csr.indptr), 1 if csr.indptr64 else 0, dptr(csr.indices), csr.num_nodes, dptr(q), n,
                                  int(num_walks), int(num_steps), int(thld), seed_eff, dptr(out), cap, dptr(count), dptr(ws),
                                  ws.numel(), dptr(flags), stream_ptr()))
    st = torch.cat([flags.long(), count]).tolist()
    if st[3] & 16:
        raise IndexError("query node ids outside [0, num_nodes) (the reference reads out of bounds for them)")
    if st[0]:
        raise _lib.SubgAccError("batch_sampler: a walk reached a node without out-edges; the sequential rand_r stream cannot "
                                "be reproduced on such a graph (the reference's graphs are symmetrised, dataloader.py:122-135)")
    if st[1]:
        raise _lib.SubgAccError("batch_sampler: output capacity exceeded (internal sizing error)")
    return out[: st[4]].cpu().numpy()


def add(i, j):
    """subg_acc.c:109-117 (demo function)."""
    return int(i) * 2 + int(j) * 7


def sjoin(edge, x, device=None, ptr=True, encode=None):
    """The paper's SpJoin operator (no symbol in the reference module; contract = train.py:13-45)."""
    from .spjoin import gather
    return gather(edge, x, device, ptr=ptr, encode=encode)
