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

__all__ = ["gset_sampler", "walk_sampler", "add", "sjoin"]


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
    # The reference reads whatever the CSR says (no bounds checks, a bad graph is a segfault); on the GPU a stray
    # address can take the device down, so the host arrays are validated before they are uploaded.
    if ip.ndim != 1 or ip.size < 1 or ix.ndim != 1:
        raise TypeError("Input parsing error. (indptr / indices must be 1-D, indptr non-empty)")
    if ip[0] != 0 or ip[-1] > ix.size or (ip.size > 1 and bool((np.diff(ip) < 0).any())):
        raise IndexError("CSR row offsets are not monotone within [0, len(indices)]")
    if ix.size and (int(ix.min()) < 0 or int(ix.max()) >= ip.size - 1):
        raise IndexError("CSR neighbour ids outside [0, num_nodes)")
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
    nsize = sets.nsize.cpu().numpy()
    sf = sets.get_sf()
    remap = torch.stack([sets.ids, sf]).cpu().numpy()
    enc_dev = sets.enc_int16()
    enc = enc_dev.cpu().numpy()
    if _lib.VERBOSE:
        print(f"#SubGAcc: #total {sets.X}; #enc_unique {sets.c}; compression ratio {sets.X / max(sets.c, 1):.2f}")
    if debug > 0:
        raw = enc_dev[sf.long()].cpu().numpy()
        return [nsize, remap, enc, raw]
    return [nsize, remap, enc]


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
    walks = sets.walks.cpu().numpy()
    off = sets.row_off.cpu().numpy()
    ids = sets.ids.cpu().numpy()
    counts = sets.counts_int32().cpu().numpy()
    n = len(off) - 1
    obj = np.empty((n, 2), dtype=object)
    for i in range(n):
        obj[i, 0] = ids[off[i]:off[i + 1]]
        obj[i, 1] = counts[off[i]:off[i + 1]]
    return [walks, obj]


def add(i, j):
    """subg_acc.c:109-117 (demo function)."""
    return int(i) * 2 + int(j) * 7


def sjoin(edge, x, device=None, ptr=True, encode=None):
    """The paper's SpJoin operator (no symbol in the reference module; contract = train.py:13-45)."""
    from .spjoin import gather
    return gather(edge, x, device, ptr=ptr, encode=encode)
