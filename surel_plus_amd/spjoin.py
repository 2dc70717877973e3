"""SpJoin on the GPU: drop-ins for train.py's gather / bgather / pgather / hgather.

Reference: train.py:13-45 (gather), :48-72 (hgather), :75-85 (bgather), :88-111 (pgather).  Same names,
same argument meaning, same return values (xz float32 [R,2,k], indptr-or-segment-ids int64 on `device`);
`x` is an SpG (surel_plus_amd.spg.SpG) or a scipy CSR (uploaded once and cached), `encode` the Z_SF table
as a float32 CUDA tensor or None for a float payload.  The work is done by csrc/sjoin.hip.
"""
import os
import threading
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import JOIN_COUNTS, JOIN_F64, JOIN_KEY32, JOIN_KEY64, JOIN_PAIRS, JOIN_ROWS, JOIN_SFPTR, check, join_fill, lib, ptr, stream_ptr

NO_ROOT = -2 ** 31      # include/subgacc.h: SUBGACC_NO_ROOT
from .sampler import _timed
from .spg import HeadedSpG, SpG, StridedSpG

_scipy_cache = weakref.WeakKeyDictionary()


def _as_spg(x):
    if isinstance(x, (SpG, StridedSpG, HeadedSpG)):
        return x
    try:
        hit = _scipy_cache.get(x)
    except TypeError:
        hit = None
    if hit is None:
        with _CACHE_LOCK:       # uploaded once, complete before another thread / stream can pick it up
            try:
                hit = _scipy_cache.get(x)
            except TypeError:
                hit = None
            if hit is None:
                hit = SpG.from_scipy(x)
                torch.cuda.current_stream(hit.device).synchronize()
                try:
                    _scipy_cache[x] = hit
                except TypeError:
                    pass
    return hit


def _as_rows(edge, device):
    """[r, B] endpoints as an int64 device tensor (the reference takes torch or numpy integer arrays)."""
    if torch.is_tensor(edge):
        return edge.to(device=device, dtype=torch.int64)
    return torch.from_numpy(np.ascontiguousarray(np.asarray(edge)).astype(np.int64)).to(device)


def sjoin(spg, own, partner, encode=None, ptr_mode=True, return_index=False, pair_block=0, out=None, lazy=False):
    """Generic segment join (include/subgacc.h: subgacc_sjoin_sizes + subgacc_sjoin_fill).

    own/partner: int64 device tensors of SpG row numbers, one segment each.  pair_block = P > 0 promises that
    the list is made of blocks of P segments with block 2t+1 the mirror of block 2t (see include/subgacc.h).
    out: optional preallocated float32 buffer with room for the R output rows (a steady-state caller re-uses one
    buffer instead of asking the allocator for a fresh GB-sized block per batch); the result is a view of it.
    lazy=True (needs out=, segment pointers, an integer SpG): no host round trip at all -- the number of rows R stays
    on the device as ind[-1] and xz is the whole buffer viewed as [capacity, 2, k], of which the first R rows are valid.
    Returns (xz, ind): xz float32 [R,2,k] (or int32 [R,2] index pairs when return_index), ind = int64 [S+1]
    segment pointers (ptr_mode) or int64 [R] segment ids.
    """
    L = lib()
    dev = spg.device
    st = stream_ptr()
    S = own.numel()
    own = own.contiguous()
    if partner is None:
        if pair_block <= 0 and S > 0:
            raise ValueError("partner=None needs a mirrored segment list (pair_block > 0)")
    else:
        partner = partner.contiguous()
    seg, flags = _seg_and_flags(S, dev)
    ws = torch.empty(L.subgacc_sjoin_workspace_bytes(S), dtype=torch.uint8, device=dev)
    if isinstance(spg, StridedSpG):
        return _sjoin_strided(spg, own, partner, seg, flags, ws, encode, ptr_mode, return_index, pair_block, out, lazy)
    if isinstance(spg, HeadedSpG):
        return _sjoin_headed(spg, own, partner, seg, flags, encode, ptr_mode, return_index, pair_block, out, lazy)
    check(L.subgacc_sjoin_sizes(ptr(spg.indptr), spg.n_rows, ptr(own), ptr(partner), S, ptr(seg), ptr(flags), ptr(ws),
                                ws.numel(), st))
    is_f64 = spg.data.dtype == torch.float64
    if getattr(spg, "keyrows", False):       # SpG.keyed(): the payload is the LP key, the join unpacks it (no table)
        from .spg import KEY_ROWS_ENCODE
        if encode is not KEY_ROWS_ENCODE or return_index or (pair_block <= 0 and S > 0) or (lazy and not ptr_mode):
            raise ValueError("a keyed() store is joined by gather / hgather(…, encode=zk.slot_table())")
        k = spg.key_m + 1
        R = None if lazy else _size_and_row_check(seg, S, flags, spg.n_rows)
        if lazy:
            if out is None or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev or \
                    out.numel() < S * spg.max_len * 2 * k:
                raise ValueError("lazy out= must hold S * SpG.max_len * 2 * k float32 on the SpG's device")
            rows = out.numel() // (2 * k)
            res = out.view(-1)[: rows * 2 * k].view(rows, 2, k)
        elif out is not None:
            if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < R * 2 * k or out.device != dev:
                raise ValueError("out= must be a contiguous float32 buffer on the SpG's device with >= R*2*k elements")
            res = out.view(-1)[: R * 2 * k].view(R, 2, k)
        else:
            res = torch.empty((R, 2, k), dtype=torch.float32, device=dev)
        segid = None if ptr_mode else torch.empty(R, dtype=torch.int64, device=dev)
        with _timed("sjoin_fill"):
            join_fill(JOIN_ROWS, JOIN_KEY32, row_off=spg.indptr, n_rows=spg.n_rows, ids=spg.indices, payload=spg.data, max_len=spg.max_len,
                      own=own, partner=partner, S=S, seg=seg, pair_block=pair_block, num_walks=spg.key_M, num_steps=spg.key_m,
                      out_xz=res, out_segid=segid, flags=flags)
        return res, (seg if ptr_mode else _with_pointers(segid, seg)), flags
    if lazy and (out is None or not ptr_mode or return_index or (encode is None and not is_f64)):
        raise ValueError("lazy=True needs out=, ptr=True and an integer SpG with its encode table (or a float-payload SpG)")
    R = None if lazy else _size_and_row_check(seg, S, flags, spg.n_rows)     # the one host round trip
    segid = None if ptr_mode else torch.empty(R, dtype=torch.int64, device=dev)
    if is_f64:
        if encode is not None:
            raise TypeError("a float-payload SpG is joined without an encode table (train.py:39-43)")
        if lazy:      # worst case: every segment as long as the longest SpG row
            if out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev or out.numel() < S * spg.max_len * 2:
                raise ValueError("lazy out= must hold S * SpG.max_len * 2 float32 on the SpG's device")
            rows = out.numel() // 2
            xz = out.view(-1)[: rows * 2].view(rows, 2, 1)
        elif out is not None:
            if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < R * 2 or out.device != dev:
                raise ValueError("out= must be a contiguous float32 buffer on the SpG's device with >= R*2 elements")
            xz = out.view(-1)[: R * 2].view(R, 2, 1)
        else:
            xz = torch.empty((R, 2, 1), dtype=torch.float32, device=dev)
        with _timed("sjoin_fill"):
            join_fill(JOIN_ROWS, JOIN_F64, row_off=spg.indptr, n_rows=spg.n_rows, ids=spg.indices, payload=spg.data, max_len=spg.max_len,
                      own=own, partner=partner, S=S, seg=seg, pair_block=pair_block, out_xz=xz, out_segid=segid, flags=flags)
        out = xz
    elif return_index:
        out = torch.empty((R, 2), dtype=torch.int32, device=dev)
        join_fill(JOIN_ROWS, JOIN_SFPTR, row_off=spg.indptr, n_rows=spg.n_rows, ids=spg.indices, payload=spg.data, max_len=spg.max_len,
                  own=own, partner=partner, S=S, seg=seg, pair_block=pair_block, out_idx=out, out_segid=segid, flags=flags)
    else:
        if encode is None:
            raise NotImplementedError("an integer SpG needs the encode table")
        enc = encode.to(device=dev, dtype=torch.float32).contiguous()
        k = enc.shape[1]
        if enc.shape[0] <= spg.max_data:       # host-side bound check: no device round trip on the hot path
            raise IndexError(f"index {spg.max_data} is out of bounds for the encode table with {enc.shape[0]} rows")
        if lazy:      # worst case: every segment as long as the longest SpG row
            if out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev or \
                    out.numel() < S * spg.max_len * 2 * k:
                raise ValueError("lazy out= must hold S * SpG.max_len * 2 * k float32 on the SpG's device")
            rows = out.numel() // (2 * k)
            out = out.view(-1)[: rows * 2 * k].view(rows, 2, k)
        elif out is not None:
            if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < R * 2 * k or out.device != dev:
                raise ValueError("out= must be a contiguous float32 buffer on the SpG's device with >= R*2*k elements")
            out = out.view(-1)[: R * 2 * k].view(R, 2, k)
        else:
            out = torch.empty((R, 2, k), dtype=torch.float32, device=dev)
        with _timed("sjoin_fill"):
            join_fill(JOIN_ROWS, JOIN_SFPTR, row_off=spg.indptr, n_rows=spg.n_rows, ids=spg.indices, payload=spg.data, max_len=spg.max_len,
                      own=own, partner=partner, S=S, seg=seg, pair_block=pair_block, table=enc, table_rows=enc.shape[0], k=k,
                      out_xz=out, out_segid=segid, flags=flags)
    return out, (seg if ptr_mode else _with_pointers(segid, seg)), flags


def _sjoin_strided(spg, own, partner, seg, flags, ws, encode, ptr_mode, return_index, pair_block, out, lazy):
    """sjoin over a StridedSpG (rows where the fused walk kernel left them): mirrored lists, segment pointers."""
    L, dev, st, S = lib(), spg.device, stream_ptr(), own.numel()
    if pair_block <= 0:
        raise ValueError("a StridedSpG is joined by gather / hgather (mirrored segment lists); use .to_csr() for the other forms")
    if lazy and not ptr_mode:
        raise ValueError("lazy=True needs ptr=True")
    check(L.subgacc_sjoin_sizes_rows(ptr(spg.nsize), spg.n_rows, ptr(own), ptr(partner), S, ptr(seg), ptr(flags), ptr(ws),
                                     ws.numel(), st))
    R = None if lazy else _size_and_row_check(seg, S, flags, spg.n_rows)
    segid = None if ptr_mode else torch.empty(R, dtype=torch.int64, device=dev)
    if return_index:
        if lazy:
            raise ValueError("lazy=True needs the encode table")
        if getattr(spg, "keyrows", False):
            raise ValueError("a key-rows batch has no row numbers to return: join z.to_csr() (gather_index does)")
        spg.sets.number()               # the pairs are SFptr+1: a transient batch is numbered only now
        res = torch.empty((R, 2), dtype=torch.int32, device=dev)
        join_fill(JOIN_ROWS, JOIN_SFPTR, row_len=spg.nsize, n_rows=spg.n_rows, row_stride=spg.stride, ids=spg.indices, payload=spg.slot,
                  uniq_table=spg.table, uniq_capacity=spg.capacity, own=own, partner=partner, S=S, seg=seg, pair_block=pair_block,
                  out_idx=res, out_segid=segid, flags=flags)
        return res, (seg if ptr_mode else _with_pointers(segid, seg)), flags
    if encode is None:
        raise NotImplementedError("an integer SpG needs the encode table")
    if getattr(spg, "keyrows", False):      # rows of LP keys: the feature rows are unpacked from the keys by the join itself
        from .spg import KEY_ROWS_ENCODE
        if encode is not KEY_ROWS_ENCODE:
            raise ValueError("a key-rows batch is joined with encode=z.slot_table(); use z.to_csr() for another table")
        M_, m_ = spg.sets.num_walks, spg.sets.num_steps
        k = m_ + 1
        if lazy:
            if out is None or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev or \
                    out.numel() < S * spg.max_len * 2 * k:
                raise ValueError("lazy out= must hold S * SpG.max_len * 2 * k float32 on the SpG's device")
            rows = out.numel() // (2 * k)
            res = out.view(-1)[: rows * 2 * k].view(rows, 2, k)
        elif out is not None:
            if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < R * 2 * k or out.device != dev:
                raise ValueError("out= must be a contiguous float32 buffer on the SpG's device with >= R*2*k elements")
            res = out.view(-1)[: R * 2 * k].view(R, 2, k)
        else:
            res = torch.empty((R, 2, k), dtype=torch.float32, device=dev)
        if not ptr_mode:
            raise ValueError("a key-rows batch is joined with segment pointers (ptr=True); use z.to_csr() for segment ids")
        with _timed("sjoin_fill"):
            join_fill(JOIN_ROWS, JOIN_KEY64 if spg.sets.key64 else JOIN_KEY32, row_len=spg.nsize, n_rows=spg.n_rows, row_stride=spg.stride,
                      ids=spg.indices, payload=spg.slot, own=own, partner=partner, S=S, seg=seg, pair_block=pair_block,
                      num_walks=M_, num_steps=m_, out_xz=res, flags=flags)
        return res, seg, flags
    by_slot = encode is spg._slot_table and encode is not None      # StridedSpG.slot_table(): indexed by slot + 1
    enc = encode if by_slot else encode.to(device=dev, dtype=torch.float32).contiguous()
    k = enc.shape[1]
    if not by_slot and enc.shape[0] <= spg.max_data:
        raise IndexError(f"index {spg.max_data} is out of bounds for the encode table with {enc.shape[0]} rows")
    tab, cap = (None, 0) if by_slot else (spg.table, spg.capacity)
    if lazy:
        if out is None or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev or \
                out.numel() < S * spg.max_len * 2 * k:
            raise ValueError("lazy out= must hold S * SpG.max_len * 2 * k float32 on the SpG's device")
        rows = out.numel() // (2 * k)
        res = out.view(-1)[: rows * 2 * k].view(rows, 2, k)
    elif out is not None:
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < R * 2 * k or out.device != dev:
            raise ValueError("out= must be a contiguous float32 buffer on the SpG's device with >= R*2*k elements")
        res = out.view(-1)[: R * 2 * k].view(R, 2, k)
    else:
        res = torch.empty((R, 2, k), dtype=torch.float32, device=dev)
    with _timed("sjoin_fill"):
        join_fill(JOIN_ROWS, JOIN_SFPTR, row_len=spg.nsize, n_rows=spg.n_rows, row_stride=spg.stride, ids=spg.indices, payload=spg.slot,
                  uniq_table=tab, uniq_capacity=cap, own=own, partner=partner, S=S, seg=seg, pair_block=pair_block,
                  table=enc, table_rows=enc.shape[0], k=k, out_xz=res, out_segid=segid, flags=flags)
    return res, (seg if ptr_mode else _with_pointers(segid, seg)), flags


def _sjoin_headed(spg, own, partner, seg, flags, encode, ptr_mode, return_index, pair_block, out, lazy):
    """sjoin over a HeadedSpG (a resident store on whole lines, include/subgacc.h: headed rows): mirrored lists.  The size pass is
    the library's one-launch form (SUBGACC_JOIN_OPT_SIZES): lazily it and the fill are ONE call into a worst-case `out`; eagerly it
    runs alone first (no output: the "count" call), the host reads [R, status] from pinned memory, and the fill follows into R rows."""
    from .spg import KEY_ROWS_ENCODE
    dev, S = spg.device, own.numel()
    if pair_block <= 0 and S > 0:
        raise ValueError("a HeadedSpG is joined by gather / hgather (mirrored segment lists); keep the packed store for the other forms")
    if return_index:
        raise ValueError("index pairs come from the packed store (gather_index)")
    kw = dict(row_stride=spg.pitch, n_rows=spg.n_rows, ids=spg.ids, payload=spg.data, own=own, partner=partner, S=S, pair_block=pair_block,
              flags=flags)
    if spg.keyrows:
        if encode is not KEY_ROWS_ENCODE:
            raise ValueError("a keyed() store is joined by gather / hgather(…, encode=zk.slot_table())")
        kind, k = JOIN_KEY32, spg.key_m + 1
        kw.update(num_walks=spg.key_M, num_steps=spg.key_m)
    elif spg.data.dtype == torch.float64:
        if encode is not None:
            raise TypeError("a float-payload SpG is joined without an encode table (train.py:39-43)")
        kind, k = JOIN_F64, 1
    else:
        if encode is None:
            raise NotImplementedError("an integer SpG needs the encode table")
        enc = encode.to(device=dev, dtype=torch.float32).contiguous()
        if enc.shape[0] <= spg.max_data:
            raise IndexError(f"index {spg.max_data} is out of bounds for the encode table with {enc.shape[0]} rows")
        kind, k = JOIN_SFPTR, int(enc.shape[1])
        kw.update(table=enc, table_rows=enc.shape[0], k=k)
    if lazy and (out is None or not ptr_mode):
        raise ValueError("lazy=True needs out= and ptr=True")
    state = torch.zeros(lib().subgacc_sjoin_workspace_bytes(S), dtype=torch.uint8, device=dev)
    host = torch.empty(2, dtype=torch.int64, pin_memory=True)
    sized = dict(options=_lib.JOIN_OPT_SIZES, out_seg=seg, size_state=state, size_state_bytes=state.numel(), host_tail=host)
    if lazy:
        if out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev or out.numel() < S * spg.max_len * 2 * k:
            raise ValueError("lazy out= must hold S * SpG.max_len * 2 * k float32 on the SpG's device")
        rows = out.numel() // (2 * k)
        res = out.view(-1)[: rows * 2 * k].view(rows, 2, k)
        with _timed("sjoin_fill"):
            join_fill(JOIN_ROWS, kind, out_xz=res, **kw, **sized)
        ev = torch.cuda.Event()
        ev.record()
        _lib.keep_until(ev, (host, state))          # the kernels write both after this function has returned
        return res, seg, flags
    join_fill(JOIN_ROWS, kind, **kw, **sized)        # no output: the size pass alone
    torch.cuda.current_stream(dev).synchronize()
    R, status = (int(v) for v in host.tolist())
    if status & 64:
        raise _lib.SubgAccError("the join's size state was not clean")
    if status & 16:
        raise IndexError(f"row index out of range for an SpG with {spg.n_rows} rows")
    if out is not None:
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() < R * 2 * k or out.device != dev:
            raise ValueError("out= must be a contiguous float32 buffer on the SpG's device with >= R*2*k elements")
        res = out.view(-1)[: R * 2 * k].view(R, 2, k)
    else:
        res = torch.empty((R, 2, k), dtype=torch.float32, device=dev)
    segid = None if ptr_mode else torch.empty(R, dtype=torch.int64, device=dev)
    with _timed("sjoin_fill"):
        join_fill(JOIN_ROWS, kind, seg=seg, out_xz=res, out_segid=segid, **kw)
    return res, (seg if ptr_mode else _with_pointers(segid, seg)), flags


def _with_pointers(segid, seg):
    """segment ids as the join's second result; the pointers they were made from ride along (gather_many / hgather_many cut by them)"""
    segid.seg_pointers = seg
    return segid


def _seg_and_flags(S, dev):
    """Segment pointers int64[S+1] and the join's int32[4] status words as ONE allocation: what the host reads of a join --
    the output size seg[S] and the status word flags[3] -- is then one contiguous 24-byte copy (`seg.join_tail`), with no
    cast / concatenate kernels in front of it (they were two of the seven nodes of a captured join)."""
    buf = torch.empty(S + 3, dtype=torch.int64, device=dev)
    flags = buf[S + 1:].view(torch.int32)
    flags.zero_()
    seg = buf[:S + 1]
    seg.join_tail = buf[S:]         # [R | flags[0], flags[1] | flags[2], flags[3]]
    return seg, flags


def tail_words(tail):
    """(R, status word) of a join_tail read back as three int64"""
    R, _, w = tail
    return int(R), (int(w) >> 32) & 0xFFFFFFFF


def _size_and_row_check(seg, S, flags, n_rows):
    """The join's one host read: the output size R = seg[S] and, in the same copy, the status word of the size pass --
    a row number outside the store is an IndexError here as it is in the reference (scipy's x[edge[0]], train.py:15);
    the kernels never dereference such a row (it reads as empty), so nothing out of bounds has happened by now."""
    R, status = tail_words(seg.join_tail.tolist())
    if status & 16:
        raise IndexError(f"row index out of range for an SpG with {n_rows} rows")
    return int(R)


# SpG.max_len / SpG.max_data make the kernel's other guards (flags[3] & 1, & 2) unreachable; SUBGACC_DEBUG=1 reads them
# back after every join anyway (one extra host sync per call).  Row numbers out of range: raised by the eager forms
# above; a lazy join (no host read at all) returns empty segments for them and leaves flags[3] & 16 set.
_DEBUG_FLAGS = os.environ.get("SUBGACC_DEBUG", "0") == "1"


def lazy_join_status(ind):
    """A lazy join reads nothing back, so a row number outside the store -- an IndexError of scipy's x[edge[0]] in the reference,
    train.py:15 -- cannot be raised when the join is queued: such a row reads as empty and the join's status word keeps the
    fact.  This reads that word for the `ind` a lazy gather() returned (one small host read, whenever the caller resolves the
    batch) and raises like the eager form; sample_and_gather(lazy=True) carries the word in its sets' status: resolve() raises."""
    flags = getattr(ind, "join_flags", None)
    if flags is not None and int(flags[3].item()) & 16:
        raise IndexError("row index out of range for the SpG (lazy join: the row was joined as an empty row)")
    return ind


def _checked(out, ind, flags, lazy=False):
    if lazy:
        ind.join_flags = flags       # see lazy_join_status()
    if _DEBUG_FLAGS and not torch.cuda.is_current_stream_capturing():      # (a capture cannot read back: the replay's finish() does)
        f = int(flags[3].item())
        if f & 16:
            raise IndexError("row index out of range for the SpG")
        if f & 1:
            raise _lib.SubgAccError("SpG row longer than SpG.max_len")
        if f & 2:
            raise IndexError("SFptr outside the encode table")
    return out, ind


def gather(edge, x, device=None, ptr=True, encode=None, out=None, lazy=False):
    """train.py:13-45.  Left blocks (S_u with S_v looked up) then right blocks, per pair in batch order.
    out= / lazy=: see sjoin (a serving loop's forms: caller-owned output buffer, no host round trip)."""
    spg = _as_spg(x)
    e = _as_rows(edge, spg.device)
    # own = [u.. | v..] is the contiguous [2, B] tensor itself; the mirrored partner list [v.. | u..] is derived by the kernels
    own = e.contiguous().view(-1)
    return _checked(*sjoin(spg, own, None, encode, ptr_mode=ptr, pair_block=e.shape[1], out=out, lazy=lazy), lazy=lazy)


class BatchViews:
    """The nb reference-shaped results of a join over nb batches laid out [u_0 | v_0 | u_1 | v_1 | ...]: a sequence of
    (xz_b, indptr_b) -- xz_b float32 [R_b, 2, k] the rows of batch b (a view of the one output buffer), indptr_b int64 [2B+1]
    starting at 0 (train.py:21-22; a row of ONE [nb, 2B+1] tensor), or segment ids 0..P-1 where the caller asked for ids.
    The nb+1 batch boundaries (and the join's status word) are read from the device when the first batch is TAKEN, in one small
    copy, and the views are made batch by batch as they are taken: a loop that queues the next group of batches before it
    consumes this one never waits for the GPU in between.  A row number outside the store raises IndexError at that point."""

    def __init__(self, xz, seg, seg_per_batch, ids=None, flags=None, n_rows=None, prefetch=False):
        P = int(seg_per_batch)
        S = seg.numel() - 1
        if P <= 0 or S % P:
            raise ValueError(f"{S} segments are not a whole number of batches of {P} segments")
        self.xz, self.seg, self.P, self.nb, self.ids = xz, seg, P, S // P, ids
        self._flags, self._n_rows, self._bounds, self._ptrs, self._pending = flags, n_rows, None, None, None
        if prefetch and seg.is_cuda:
            # the boundaries start their way to pinned host memory NOW, behind this join's kernels: taking the first batch later waits
            # for exactly this copy, not for whatever the stream holds by then (the next group of batches, queued in between)
            src = self._words()
            host = torch.empty(src.numel(), dtype=torch.int64, pin_memory=True)
            _lib.publish(src, host)
            ev = torch.cuda.Event()
            ev.record()
            _lib.keep_until(ev, host)
            self._pending = (host, ev, src)

    def _words(self):
        src = self.seg[::self.P]                               # [nb+1]: first row of every batch, and the total
        if self._flags is not None:
            src = torch.cat([src, self._flags[3:4].to(torch.int64)])
        return src

    def _resolve(self):
        if self._bounds is None:
            if self._pending is not None:
                host, ev, _ = self._pending
                ev.synchronize()
                words = host.tolist()
                self._pending = None
            else:
                words = self._words().tolist()
            if self._flags is not None:
                if words.pop() & 16:
                    raise IndexError(f"row index out of range for an SpG with {self._n_rows} rows")
            self._bounds = words
            if self.ids is None:
                self._ptrs = self.seg.as_strided((self.nb, self.P + 1), (self.P, 1)) - self.seg[:-1:self.P][:, None]
        return self._bounds

    def __len__(self):
        return self.nb

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.nb))]
        if i < 0:
            i += self.nb
        if not 0 <= i < self.nb:
            raise IndexError(i)
        b = self._resolve()
        if self.ids is not None:
            return self.xz[b[i]:b[i + 1]], self.ids[b[i]:b[i + 1]]
        return self.xz[b[i]:b[i + 1]], self._ptrs[i]

    def __iter__(self):
        return (self[i] for i in range(self.nb))

    def __eq__(self, other):                    # (an empty result compares equal to []: gather_many(edges[:0]) == [])
        return list(self) == other if isinstance(other, list) else NotImplemented


def split_batches(xz, seg, batch_pairs):
    """The result of a join over nb batches of `batch_pairs` pairs (gather_many, sample_and_gather_many, a StepBuffers made with
    batch=) as its nb reference-shaped pieces: see BatchViews."""
    return BatchViews(xz, seg, 2 * int(batch_pairs))


def gather_many(edges, x, device=None, ptr=True, encode=None, out=None, lazy=False):
    """gather() for MANY reference-sized batches at once: `edges` [nb, 2, B] (the batches of an epoch are known when it starts:
    train.py:120 draws the DataLoader permutation up front) joined in ONE launch sequence -- one size pass, one scan, one fill
    over nb*B pairs -- instead of nb times three launches of 1,024 pairs that cannot fill the chip (main.py:32).  Returns
    [(xz_b, ind_b)] * nb, bit for bit what `gather(edges[b], x, device, ptr, encode)` returns for every b: xz_b float32
    [R_b, 2, k] (views of one buffer, `out` if given), ind_b int64 [2B+1] segment pointers from 0 (ptr=True) or int64 [R_b]
    segment ids 0..2B-1 (ptr=False).  One small host read when the call is made (the total number of rows, to size xz) and one when
    the first batch is taken (BatchViews); lazy=True (needs out= for the worst case -- nb*2B * SpG.max_len * 2k float32 -- and
    ptr=True) makes the call itself free of host reads: the next group can be queued before this one is consumed.
    `edges` may also be a list of [2, B_i] arrays: runs of equal B are fused, the rest (an epoch's short last batch) joined singly
    -- eagerly: out= / lazy=True with a list raise ValueError."""
    spg = _as_spg(x)
    if isinstance(edges, (list, tuple)):
        if out is not None or lazy:
            # (a list is cut into runs of equal batch size, each its own join with its own row count: one caller's buffer cannot
            #  be handed out before those counts are read, which is exactly the host read lazy=True promises not to make)
            raise ValueError("gather_many: out= / lazy=True need the batches as one [nb, 2, B] array (a list of batches is joined "
                             "run by run, eagerly); stack equal-sized batches, join a short last batch with gather()")
        res, i = [], 0
        while i < len(edges):
            j = i
            while j + 1 < len(edges) and tuple(edges[j + 1].shape) == tuple(edges[i].shape):
                j += 1
            if j == i:
                res.append(gather(edges[i], spg, device, ptr=ptr, encode=encode))
            else:
                stack = torch.stack([_as_rows(e, spg.device) for e in edges[i:j + 1]])
                res.extend(gather_many(stack, spg, device, ptr=ptr, encode=encode))
            i = j + 1
        return res
    e = _as_rows(edges, spg.device)
    if e.dim() != 3 or e.shape[1] != 2:
        raise ValueError("gather_many: edges must be [nb, 2, B]")
    nb, _, B = e.shape
    if nb == 0 or B == 0:
        return [gather(e[b], spg, device, ptr=ptr, encode=encode) for b in range(nb)]
    own = e.contiguous().view(-1)                              # [u_0 | v_0 | u_1 | v_1 | ...]: mirrored blocks of B segments
    if lazy and not ptr:
        raise ValueError("gather_many(lazy=True) needs ptr=True (segment ids are sized by the row count)")
    xz, ind, flags = sjoin(spg, own, None, encode, ptr_mode=ptr, pair_block=B, out=out, lazy=lazy)
    _checked(xz, ind, flags)
    if ptr:
        return BatchViews(xz, ind, 2 * B, flags=flags if lazy else None, n_rows=spg.n_rows, prefetch=lazy)
    # segment ids (train.py:25-30, the LSTM aggregator): the kernel wrote the ids over ALL segments; inside a batch they are 0..2B-1
    ind.remainder_(2 * B)
    return BatchViews(xz, ind.seg_pointers, 2 * B, ids=ind)


def hgather(hedge, x, device=None, encode=None):
    """train.py:48-72.  Blocks [U|w ; W|u ; V|w ; W|v], always segment ids; encode is mandatory."""
    if encode is None:
        raise NotImplementedError
    spg = _as_spg(x)
    h = _as_rows(hedge, spg.device)
    u, v, w = h[0], h[1], h[2]
    own = torch.cat([u, w, v, w])          # the mirrored partner list [w, u, w, v] is derived by the kernels
    xz, ind = _checked(*sjoin(spg, own, None, encode, ptr_mode=False, pair_block=h.shape[1]))
    assert xz.size(0) == ind.size(0)
    return xz, ind


def hgather_many(hedges, x, device=None, encode=None):
    """hgather() for MANY batches of triplets at once (main_horder.py:33: 2,048 triplets per batch, ~8k one-pair workgroups that
    cannot fill the chip): `hedges` [nb, 3, B] joined in one launch sequence; returns [(xz_b, ids_b)] * nb, bit for bit what
    hgather(hedges[b], x, device, encode) returns (segment ids 0..4B-1 per batch, train.py:57-68)."""
    if encode is None:
        raise NotImplementedError
    spg = _as_spg(x)
    h = _as_rows(hedges, spg.device)
    if h.dim() != 3 or h.shape[1] != 3:
        raise ValueError("hgather_many: hedges must be [nb, 3, B]")
    nb, _, B = h.shape
    if nb == 0 or B == 0:
        return [hgather(h[b], spg, device, encode) for b in range(nb)]
    own = torch.stack([h[:, 0], h[:, 2], h[:, 1], h[:, 2]], dim=1).contiguous().view(-1)      # per batch [u | w | v | w]: two mirrored pairs of blocks
    xz, ids, flags = sjoin(spg, own, None, encode, ptr_mode=False, pair_block=B)
    _checked(xz, ids, flags)
    ids.remainder_(4 * B)                                      # the kernel numbered the segments of all batches: 0..4B-1 inside a batch
    return BatchViews(xz, ids.seg_pointers, 4 * B, ids=ids)


def bgather(edge, x, out):
    """train.py:75-85: the per-thread block worker of pgather.  Kept for signature compatibility: fills
    out[0..3] with (left pairs, right pairs, left sizes, right sizes) as NumPy arrays."""
    spg = _as_spg(x)
    e = _as_rows(edge, spg.device)
    B = e.shape[1]
    own = torch.cat([e[0], e[1]])
    partner = torch.cat([e[1], e[0]])
    if spg.data.dtype == torch.float64:
        pairs, seg = _checked(*sjoin(spg, own, partner, None, ptr_mode=True))
        pairs = pairs.view(-1, 2)
    else:
        pairs, seg = _checked(*sjoin(spg, own, partner, None, ptr_mode=True, return_index=True))
    sizes = (seg[1:] - seg[:-1]).cpu().numpy()
    mid = int(seg[B].item())
    pairs = pairs.cpu().numpy()
    out[0], out[1] = pairs[:mid], pairs[mid:]
    out[2], out[3] = sizes[:B], sizes[B:]


def pgather(edge, M, device=None, encode=None, gather_func=None, ptr=True, njobs=4):
    """train.py:88-111.  The reference splits the batch over `njobs` Python threads; one kernel launch
    covers the whole batch here, so `gather_func` / `njobs` only keep the call signature.  The result is
    identical to gather() (as it is in the reference)."""
    return gather(edge, M, device, ptr=ptr, encode=encode)


class StepBuffers:
    """Everything one on-demand step (sample_and_gather) touches, allocated once for a fixed (B, M, m): the int32 roots, the
    strided rows and their sizes, the table of distinct LP rows and the feature table indexed by its slots, the segment
    pointers with the step's status words right behind them (one small read-back carries sizes, flags and the join's row
    count) and the output buffer.  With it a step is SIX launches -- prologue (table reset + status + root narrowing),
    walk, segment reduce, segment scan, LP unpack, join -- and no allocation; without it torch's allocator and a dozen
    few-microsecond helper kernels sit between them (1,024 pairs: ~100 us of which the walk and the join are 55).
    Reuse is the caller's business: a buffer set is busy until its step has been resolved (bench.py alternates two).
    dedup_roots=True: every DISTINCT endpoint of the batch is sampled once (subgacc_step_prologue_dedup: a generation-stamped
    hash of the endpoints names every node's first occurrence, one more small launch; the walk kernel runs over the list of
    first occurrences) -- Philox keys a walk by its root's id, so (xz, indptr) do not change; the sets of the batch sit in the
    rows of the first occurrences (sets.n_distinct of them; bufs.roots == NO_ROOT elsewhere), the other rows are empty.
    The hash is stamped with a per-step generation kept on the device: a captured step replays correctly."""

    def __init__(self, csr, pairs, num_walks=200, num_steps=3, uniq_capacity=1 << 17, out=None, dedup_roots=False, rng="philox",
                 key_rows=True, sort_roots=True, batch=None, align_rows=True):
        from .sampler import FUSED_MAX_Q
        L, dev = lib(), csr.device
        self.B, self.M, self.m = int(pairs), int(num_walks), int(num_steps)
        # batch=b: the `pairs` of a step are pairs/b reference-sized batches of b pairs each, handed over as [nb, 2, b] (rows
        # [u_0 | v_0 | u_1 | v_1 | ...]); the join pairs row j with its mirror inside ITS batch, and split_batches() cuts the
        # result into the nb reference-shaped (xz, indptr).  None: one batch, [2, pairs].
        self.batch = self.B if batch is None else int(batch)
        if self.batch <= 0 or self.B % self.batch:
            raise ValueError(f"StepBuffers: pairs = {self.B} is not a whole number of batches of {batch}")
        if self.batch != self.B and dedup_roots:
            raise ValueError("StepBuffers: root dedup works on one batch (batch=None)")
        n, self.Q, self.k = 2 * self.B, self.M * self.m + 1, self.m + 1
        if self.Q > FUSED_MAX_Q or self.m < 1:
            raise ValueError(f"StepBuffers: num_walks*num_steps+1 = {self.Q} exceeds what the fused-row walk kernel holds")
        # the rows of two roots lie `stride` words apart: M*m+1 rounded up to whole 128-byte lines (subgacc_walk_cfg::row_pitch) --
        # the join reads, and the walk kernel writes, whole lines (the join alone: 3-5 % on every workload, profiles/r28_join_pitch.log)
        # (align_rows=False: rows M*m+1 words apart, the layout of rounds 1-4)
        self.stride = (self.Q + 31) // 32 * 32 if align_rows else self.Q
        self.capacity = int(uniq_capacity)
        self.roots = torch.empty(n, dtype=torch.int32, device=dev)
        self.nsize = torch.empty(n, dtype=torch.int32, device=dev)
        self.ids = torch.empty(n * self.stride, dtype=torch.int32, device=dev)
        from .sampler import key_rows_form
        form = key_rows_form(self.M, self.m) if key_rows else 0
        self.keyrows = bool(form)                 # rows of LP keys: no table, no feature table
        self.key64 = form == 64                   # ... 64-bit keys (4-hop walks with M >= 128): `slot` is int64
        self.slot = torch.empty(n * self.stride, dtype=torch.int64 if self.key64 else torch.int32, device=dev)
        self.sort_roots = bool(sort_roots)     # the walk kernel takes the rows in ascending order of root id (csrc/worklist.hip)
        self.table = None if self.keyrows else torch.empty(L.subgacc_uniq_table_bytes(self.capacity), dtype=torch.uint8, device=dev)
        self.tail = torch.zeros(n + 1 + 4 + 1, dtype=torch.int64, device=dev)  # seg [n+1] | status [4] | distinct roots [1]
        self.seg, self.status, self.n_distinct = self.tail[: n + 1], self.tail[n + 1: n + 5], self.tail[n + 5:]
        self.dedup = bool(dedup_roots)
        self.rng = rng
        if rng not in ("philox", "rand_r") or (rng == "rand_r" and self.dedup):
            raise ValueError("StepBuffers: rng is 'philox' or 'rand_r'; root dedup needs 'philox' (a rand_r set depends on its place in the stream)")
        if rng == "rand_r" and getattr(csr, "_rand_r_dead_ends", False):
            raise ValueError("StepBuffers(rng='rand_r'): this graph has dead ends (a walk reached a node without out-edges), its "
                             "rand_r stream has to be replayed per batch -- sample_and_gather(..., rng='rand_r') without buffers= "
                             "does that; the buffered step cannot")
        if rng == "rand_r":      # the rows' places in the reference's sequential stream (subgacc_rng_positions), per step
            self.rng_pos = torch.empty(n, dtype=torch.int32, device=dev)
            self.rng_seed = torch.empty(n, dtype=torch.int32, device=dev)
            self.rng_ws = torch.empty(L.subgacc_rng_positions_workspace_bytes(n), dtype=torch.uint8, device=dev)
        if self.dedup:
            from .sampler import walk_kernel_name
            if walk_kernel_name(csr, self.M, self.m, True) != "walk_rows_kernel":
                raise ValueError("StepBuffers(dedup_roots=True) needs a shape the fused-row walk kernel serves (2..4 hops, M <= 256)")
            self.own = torch.empty(n, dtype=torch.int64, device=dev)
            self.partner = torch.empty(n, dtype=torch.int64, device=dev)
            self.worklist = torch.empty(n, dtype=torch.int32, device=dev)
            self.dedup_ws = torch.zeros(L.subgacc_step_dedup_workspace_bytes(n), dtype=torch.uint8, device=dev)
            self.dedup_steps = 0
        self.ws = torch.empty(max(L.subgacc_sjoin_workspace_bytes(n), 8), dtype=torch.uint8, device=dev)
        self.feat = None if self.keyrows else torch.empty((self.capacity + 1, self.k), dtype=torch.float32, device=dev)
        need = n * self.Q * 2 * self.k
        if out is not None and (out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev or out.numel() < need):
            raise ValueError("StepBuffers: out= must hold 2B * (M*m+1) * 2 * (m+1) float32 on the graph's device")
        self.out = out if out is not None else torch.empty(need, dtype=torch.float32, device=dev)


def _dedup_tick(bufs):
    """the hash's 32-bit generation must never wrap (a stale stamp would outrank a fresh one): long before it could, the
    workspace is zeroed again -- host side, once in 2^31 steps (every queued step of these buffers is behind it on the stream)"""
    bufs.dedup_steps += 1
    if bufs.dedup_steps >= (1 << 31):
        bufs.dedup_ws.zero_()
        bufs.dedup_steps = 0


def _sorted_list(bufs, csr, n, L, st):
    """the rows of the step as a work list in ascending order of root id (csrc/worklist.hip); its buffers are made on first use"""
    if not hasattr(bufs, "sorted_list"):
        dev = csr.device
        bufs.sorted_list = torch.empty(n, dtype=torch.int32, device=dev)
        bufs.n_all = torch.zeros(1, dtype=torch.int64, device=dev)
        bufs.sort_ws = torch.zeros(L.subgacc_worklist_workspace_bytes(n), dtype=torch.uint8, device=dev)      # (zeroed once: every call leaves it so)
    check(L.subgacc_worklist_by_root(ptr(bufs.roots), n, csr.num_nodes, ptr(bufs.sorted_list), ptr(bufs.n_all), ptr(bufs.sort_ws),
                                     bufs.sort_ws.numel(), st))


def _buffered_step(csr, e, bufs, seed, out):
    """sample_and_gather through a StepBuffers: six launches, nothing allocated, nothing read back"""
    from .sampler import SampledSets, _timed, make_cfg, walk_kernel_name
    L, st, dev = lib(), stream_ptr(), csr.device
    B, M, m, k, n = bufs.B, bufs.M, bufs.m, bufs.k, 2 * bufs.B
    PB = bufs.batch              # pairs per mirrored block of the segment list
    if tuple(e.shape) != ((2, B) if PB == B else (B // PB, 2, PB)):
        raise ValueError(f"these StepBuffers were made for [2, {B}] pairs" if PB == B else
                         f"these StepBuffers were made for [{B // PB}, 2, {PB}] pairs")
    e = e.contiguous()
    flags = bufs.status.view(torch.int32)[:4]
    cfg = make_cfg(csr, M, m, -1, seed, bufs.rng, records=(2 <= m <= 4),     # (only the fused-row kernel of 2..4 hops reads hop records)
                   row_pitch=bufs.stride if bufs.stride != bufs.Q else 0)
    rr = bufs.rng == "rand_r"
    rp, rs = (ptr(bufs.rng_pos), ptr(bufs.rng_seed)) if rr else (None, None)
    check(L.subgacc_key_shift(cfg.num_walks, cfg.num_steps))
    kr = bufs.keyrows
    bufs.step_id = step_id = getattr(bufs, "step_id", 0) + 1
    if bufs.dedup:      # first occurrences only: the other rows stay empty, the segment lists point at the first occurrence
        if not torch.cuda.is_current_stream_capturing():
            _dedup_tick(bufs)
        check(L.subgacc_step_prologue_dedup(ptr(bufs.table), 0 if kr else bufs.capacity, ptr(bufs.status), 4, ptr(e), ptr(bufs.roots),
                                            ptr(bufs.own), ptr(bufs.partner), ptr(bufs.worklist), ptr(bufs.nsize), n,
                                            ptr(bufs.dedup_ws), bufs.dedup_ws.numel(), ptr(bufs.n_distinct), st))
        # (no sorted list here: what the order buys is mostly repeated endpoints standing next to each other, and those are gone --
        # measured, cit2 walk kernel 0.647 ms either way, and the sort costs its 25 us)
        with _timed("walk_sets"):
            if bufs.key64:
                check(L.subgacc_walk_keyrows64(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(bufs.roots), n, None, None,
                                               ptr(bufs.worklist), ptr(bufs.n_distinct), ptr(bufs.ids), ptr(bufs.slot), ptr(bufs.nsize),
                                               ptr(flags), st))
            else:
                check(L.subgacc_walk_spg_sparse(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(bufs.roots), n,
                                                ptr(bufs.worklist), ptr(bufs.n_distinct), ptr(bufs.table), 0 if kr else bufs.capacity,
                                                ptr(bufs.ids), ptr(bufs.slot), ptr(bufs.nsize), ptr(flags), st))
        own, partner = bufs.own, bufs.partner
    elif bufs.sort_roots and n >= SORT_ROOTS_MIN and walk_kernel_name(csr, M, m, True) == "walk_rows_kernel":
        # the rows stay where the batch has them; the walk kernel takes them in ascending order of their root's id (a work list):
        # roots that are neighbours in id space -- the same community of a graph with id locality -- are walked at the same time on
        # the same XCD and share its L2
        check(L.subgacc_step_prologue(ptr(bufs.table), 0 if kr else bufs.capacity, ptr(bufs.status), 4, ptr(e), ptr(bufs.roots), n, st))
        _sorted_list(bufs, csr, n, L, st)
        if rr:
            check(L.subgacc_rng_positions(cfg, ptr(csr.indptr), csr.num_nodes, ptr(bufs.roots), n, 1, 0, ptr(bufs.rng_pos),
                                          ptr(bufs.rng_seed), ptr(bufs.rng_ws), bufs.rng_ws.numel(), st))
        with _timed("walk_sets"):
            if bufs.key64:
                check(L.subgacc_walk_keyrows64(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(bufs.roots), n, rp, rs,
                                               ptr(bufs.sorted_list), ptr(bufs.n_all), ptr(bufs.ids), ptr(bufs.slot), ptr(bufs.nsize),
                                               ptr(flags), st))
            else:
                check(L.subgacc_walk_spg_list(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(bufs.roots), n, rp, rs,
                                              ptr(bufs.sorted_list), ptr(bufs.n_all), ptr(bufs.table), 0 if kr else bufs.capacity,
                                              ptr(bufs.ids), ptr(bufs.slot), ptr(bufs.nsize), ptr(flags), st))
        own, partner = _arange_segments(B, dev, PB)
    else:
        check(L.subgacc_step_prologue(ptr(bufs.table), 0 if kr else bufs.capacity, ptr(bufs.status), 4, ptr(e), ptr(bufs.roots), n, st))
        if rr:
            check(L.subgacc_rng_positions(cfg, ptr(csr.indptr), csr.num_nodes, ptr(bufs.roots), n, 1, 0, ptr(bufs.rng_pos),
                                          ptr(bufs.rng_seed), ptr(bufs.rng_ws), bufs.rng_ws.numel(), st))
        with _timed("walk_sets"):
            if bufs.key64:
                check(L.subgacc_walk_keyrows64(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(bufs.roots), n, rp, rs,
                                               None, None, ptr(bufs.ids), ptr(bufs.slot), ptr(bufs.nsize), ptr(flags), st))
            else:
                check(L.subgacc_walk_spg(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(bufs.roots), n, 0, rp, rs,
                                         ptr(bufs.table), 0 if kr else bufs.capacity, ptr(bufs.ids), ptr(bufs.slot), ptr(bufs.nsize),
                                         ptr(flags), st))
        own, partner = _arange_segments(B, dev, PB)
    check(L.subgacc_sjoin_sizes_rows(ptr(bufs.nsize), n, ptr(own), ptr(partner), n, ptr(bufs.seg), ptr(flags), ptr(bufs.ws),
                                     bufs.ws.numel(), st))
    if not kr:
        keys = bufs.table[: bufs.capacity * 8].view(torch.int64)
        check(L.subgacc_unpack_lp(ptr(keys), bufs.capacity, None, M, m, None, None, ptr(bufs.feat), 1, st))
    res = out if out is not None else bufs.out
    if res.dtype != torch.float32 or not res.is_contiguous() or res.device != dev or res.numel() < n * bufs.Q * 2 * k:
        raise ValueError("out= must hold 2B * (M*m+1) * 2 * (m+1) float32 on the graph's device")
    rows = res.numel() // (2 * k)
    xz = res.view(-1)[: rows * 2 * k].view(rows, 2, k)
    with _timed("sjoin_fill"):
        if kr:
            join_fill(JOIN_ROWS, JOIN_KEY64 if bufs.key64 else JOIN_KEY32, row_len=bufs.nsize, n_rows=n, row_stride=bufs.stride, ids=bufs.ids,
                      payload=bufs.slot, own=own, partner=partner, S=n, seg=bufs.seg, pair_block=PB, num_walks=M, num_steps=m,
                      out_xz=xz, flags=flags)
        else:
            join_fill(JOIN_ROWS, JOIN_SFPTR, row_len=bufs.nsize, n_rows=n, row_stride=bufs.stride, ids=bufs.ids, payload=bufs.slot,
                      own=own, partner=partner, S=n, seg=bufs.seg, pair_block=PB, table=bufs.feat, table_rows=bufs.capacity + 1, k=k,
                      out_xz=xz, flags=flags)
    sets = SampledSets(bufs.nsize, None, bufs.ids, None, None, None, M, m, bufs.stride, None)
    sets.slot, sets.table, sets.capacity, sets.strided = bufs.slot, bufs.table, (0 if kr else bufs.capacity), True
    if kr:
        sets.keyrows, sets.key64 = True, bufs.key64
        # number() registers the keys of the rows as they stand in the buffers (root dedup: the rows of repeated endpoints are
        # empty, and a repeated endpoint never is the first to show an LP row, so the numbering is the one of the whole batch);
        # once the buffers have taken a later batch -- or a captured step has been replayed -- the answer would describe that
        # batch, so it is refused (stamp of the step that made these sets against the buffers' current one)
        sets._keyctx = {"csr": csr, "roots": bufs.roots, "cfg": cfg, "rng_pos": bufs.rng_pos if rr else None,
                        "rng_seed": bufs.rng_seed if rr else None, "capacity": bufs.capacity,
                        "fresh": lambda: getattr(bufs, "step_id", 0) == step_id}
    # every set of a buffered step -- key rows or table form -- is a view of buffers that the NEXT step overwrites: what is computed
    # from them on demand (the member count of a deduplicated step, X / nnz) is refused once they hold a later batch
    sets._fresh = lambda: getattr(bufs, "step_id", 0) == step_id
    sets.status, sets._tail = bufs.status, bufs.tail[n: n + (6 if bufs.dedup else 5)]
    return xz, bufs.seg, sets


def sample_and_gather(csr, edge, num_walks=200, num_steps=3, seed=111413, rng="philox", dedup_roots=False, out=None,
                      lazy=False, strided=None, buffers=None, **kw):
    """The on-demand form of the path in one call: sample the endpoints of `edge` [2, B] (node ids), build their SpG
    rows, join -> (xz, indptr, sets), the same (xz, indptr) as `gather(edge, subg_matrix(G, arange(N)))` would give for sets
    drawn with the same RNG (Philox keys every walk by (seed, root id, walk, step), so a root's set does not depend on
    where or how often it appears).  `num_steps` = walk hops.

    dedup_roots=True samples every distinct endpoint once (needs rng="philox"): evaluation batches repeat a source
    against a thousand candidates (utils.py:92-95), so half of the endpoints of such a batch are duplicates.
    strided=None picks the joined-in-place form where the fused walk kernel is the faster one (spg.prefers_fused).
    buffers=StepBuffers(...): the same step without a single allocation or helper kernel (six launches; the result is
    lazy: sets.prefetch() / sets.resolve() as with lazy=True, xz is a view of out= or of the buffers' own output)."""
    from .spg import sample_spg
    e = _as_rows(edge, csr.device)
    if buffers is None and e.dim() != 2:
        raise ValueError("sample_and_gather: edge must be [2, B] (many batches at once: sample_and_gather_many)")
    B = e.shape[-1]
    if buffers is not None:     # the allocation-free form of a serving loop: same rows, same (xz, indptr), lazily resolved
        if (dedup_roots and not buffers.dedup) or rng != buffers.rng or strided is False or kw.get("fused") is False or \
                kw.get("bucket", -1) > 0 or (num_walks, num_steps) != (buffers.M, buffers.m) or \
                kw.get("uniq_capacity", buffers.capacity) != buffers.capacity:
            raise ValueError("buffers= serves the on-demand step (the StepBuffers' rng, fused strided rows; root dedup if the StepBuffers "
                             "were made with dedup_roots=True) of the shape the StepBuffers were made for")
        return _buffered_step(csr, e, buffers, seed, out)
    if dedup_roots:
        if rng != "philox":
            raise ValueError("dedup_roots=True needs rng='philox' (rand_r sets depend on the position in the stream)")
        roots, inv = torch.unique(e.reshape(-1), return_inverse=True)
        rows = inv.view(2, B)
    else:
        roots = e.reshape(-1)
        rows = None                   # row i of the batch's SpG = endpoint i: the segment lists are constants of B
    if strided is None:
        strided = True        # a transient batch: joined in place, whichever walk kernel made the rows
    if strided:       # joined by table slot below: the distinct LP rows need no numbering (SampledSets.number() does it on demand)
        kw.setdefault("number_rows", False)
    z, sets = sample_spg(csr, roots.to(torch.int32), num_walks=num_walks, num_steps=num_steps, seed=seed, rng=rng, lazy=lazy,
                         strided=strided, **kw)
    table = z.slot_table() if sets.strided else sets.feature_table()
    if rows is not None:
        xz, ind = gather(rows, z, e.device, ptr=True, encode=table, out=out, lazy=lazy)
    else:
        own, partner = _arange_segments(B, e.device)
        xz, ind = _checked(*sjoin(_as_spg(z), own, partner, table, ptr_mode=True, pair_block=B, out=out, lazy=lazy), lazy=lazy)
    if lazy:      # the join's status word travels with the sets' own: resolve() raises IndexError for a row outside the store
        sets._join_flags = getattr(ind, "join_flags", None)
    return xz, ind, sets


def sample_and_gather_many(csr, edges, num_walks=200, num_steps=3, seed=111413, rng="philox", out=None, buffers=None, **kw):
    """sample_and_gather() for MANY reference-sized batches at once (main.py:32: 1,024 pairs -- 2,048 roots cannot fill the
    chip): `edges` [nb, 2, B]; all nb*2B endpoints are sampled by ONE walk launch and joined by ONE join launch, and the result
    is cut into nb reference-shaped pieces -- [(xz_b, indptr_b)] * nb, bit for bit what sample_and_gather(csr, edges[b], ...)
    returns for every b with rng="philox" (a root's set is a function of (seed, root id)), plus the sets of the whole call.
    buffers=StepBuffers(csr, nb*B, ..., batch=B): the allocation-free six-launch form; the sets are resolved here (one read)."""
    e = _as_rows(edges, csr.device)
    if e.dim() != 3 or e.shape[1] != 2:
        raise ValueError("sample_and_gather_many: edges must be [nb, 2, B]")
    nb, _, B = e.shape
    if rng != "philox":
        raise ValueError("sample_and_gather_many needs rng='philox' (a rand_r set depends on its root's place in the stream, i.e. on "
                         "which other batches are sampled with it)")
    if buffers is not None:
        if buffers.batch != B or buffers.B != nb * B or buffers.dedup:
            raise ValueError(f"buffers= must be StepBuffers(csr, {nb * B}, ..., batch={B})")
        xz, seg, sets = sample_and_gather(csr, e, num_walks=num_walks, num_steps=num_steps, seed=seed, rng=rng, out=out, buffers=buffers, **kw)
        sets.resolve()
        views = split_batches(xz, seg, B)
        views._resolve()      # (the pointers live in the step buffers: read now, before the buffers take another step)
        return views, sets
    from .spg import sample_spg
    kw.setdefault("number_rows", False)
    z, sets = sample_spg(csr, e.reshape(-1).to(torch.int32), num_walks=num_walks, num_steps=num_steps, seed=seed, rng=rng,
                         strided=True, **kw)
    table = z.slot_table() if sets.strided else sets.feature_table()
    own = torch.arange(nb * 2 * B, device=e.device, dtype=torch.int64)
    xz, seg = _checked(*sjoin(_as_spg(z), own, None, table, ptr_mode=True, pair_block=B, out=out))
    return split_batches(xz, seg, B), sets


# the buffered step walks its rows in ascending order of root id (csrc/worklist.hip: one radix pass, two small launches): cit2-like
# step +5.7 % pairs/s, twitter-like +5.6 %, collab +1.8 %, ppa +2.1 %; StepBuffers(sort_roots=False): batch order.  Nothing observable changes.
SORT_ROOTS_MIN = 16384      # rows from which the two extra launches pay (a 1,024-pair step is 2,048 roots: one workgroup per resident slot)
_ARANGE_SEGMENTS = {}
_CACHE_LOCK = threading.Lock()     # the reference's pgather calls the join from 4 Python threads (train.py:88-99)


def _arange_segments(B, device, block=None):
    """gather()'s segment lists for edge = [[0..B), [B..2B)] -- the rows of a batch sampled endpoint by endpoint -- kept per
    (B, device): a serving loop does not rebuild them (three small kernels) for every batch.  Shared between threads and
    streams: built under a lock and COMPLETE (the building stream is synchronised, once) before anybody else can see them.
    block = b < B: the B pairs are B/b batches laid out [u_0 | v_0 | u_1 | v_1 | ...], every row mirrored inside its batch."""
    block = int(B) if block is None else int(block)
    key = (int(B), str(device), block)
    hit = _ARANGE_SEGMENTS.get(key)
    if hit is None:
        with _CACHE_LOCK:
            hit = _ARANGE_SEGMENTS.get(key)
            if hit is None:
                own = torch.arange(2 * B, device=device, dtype=torch.int64)
                hit = (own, own.view(-1, 2, block).flip(1).contiguous().view(-1))
                if not torch.cuda.is_current_stream_capturing():
                    torch.cuda.current_stream(own.device).synchronize()
                _ARANGE_SEGMENTS[key] = hit
    return hit


def gather_counts(edge, x, table_rows, device=None):
    """Count form of gather() for mean aggregation (reference consumer model.py:78-83).

    Returns (C float32 [2B, table_rows], sizes int64 [2B]) with C[j, p] = occurrences of LP row p (0 = partner
    absent) in either feature slot of segment j -- left blocks then right blocks, as gather().  For any row-wise
    embedding f:  segment_sum_j(f(xz).sum(-2)) == C[j] @ f(Z_SF), so `x = f(xz).sum(-2); aggr(x, ptr)` of the
    reference's Net.forward becomes `(C @ f(Z_SF)) / sizes[:, None]` and the [R,2,k] tensor never exists."""
    spg = _as_spg(x)
    if spg.data.dtype != torch.int32 or getattr(spg, "keyrows", False):
        raise TypeError("gather_counts needs an SFptr (integer) SpG (not a keyed() one)")
    if table_rows <= spg.max_data:
        raise IndexError(f"index {spg.max_data} is out of bounds for a table with {table_rows} rows")
    e = _as_rows(edge, spg.device)
    B = e.shape[1]
    if B and bool(((e < 0) | (e >= spg.n_rows)).any()):          # this form has no size pass to carry the check
        raise IndexError(f"row index out of range for an SpG with {spg.n_rows} rows")
    own = torch.cat([e[0], e[1]]).contiguous()
    partner = torch.cat([e[1], e[0]]).contiguous()
    dev = spg.device
    out = torch.empty((2 * B, int(table_rows)), dtype=torch.float32, device=dev)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    with _timed("sjoin_counts"):
        join_fill(JOIN_COUNTS, JOIN_SFPTR, row_off=spg.indptr, n_rows=spg.n_rows, ids=spg.indices, payload=spg.data, max_len=spg.max_len,
                  own=own, partner=partner, S=2 * B, pair_block=B, table_rows=int(table_rows), out_counts=out, flags=flags)
    sizes = spg.indptr[own + 1] - spg.indptr[own]
    _checked(out, sizes, flags)
    return out, sizes


def mean_stage(edge, x, encode, embed):
    """The reference's first model stage for mean aggregation, fused:  model.py:78-83
        x = pe_embedding(xz).sum(dim=-2);  xl, xr = aggr.MeanAggregation()(x, ptr=ptr).view(2, -1, H)
    as  (C @ embed(encode)) / sizes  with C = gather_counts(edge, x).  `embed` is any row-wise module (the reference's
    pe_embedding MLP); autograd reaches its parameters through the small [c+1, H] activation -- C is a constant of
    the batch -- so the stage trains like the original while xz [R,2,k] and the [R,2,H] activations never exist.
    Returns float32 [2, B, H] (left endpoints, right endpoints); empty segments give zero rows."""
    table = encode if torch.is_tensor(encode) else torch.as_tensor(encode)
    spg = _as_spg(x)
    table = table.to(device=spg.device, dtype=torch.float32)
    C, sizes = gather_counts(edge, spg, table.shape[0])
    out = (C @ embed(table)) / sizes.clamp(min=1).to(torch.float32)[:, None]
    return out.view(2, -1, out.shape[-1])


def gather_pairs(edge, x, device=None):
    """Pair form of gather() (include/subgacc.h: subgacc_sjoin_pairs): every segment as its DISTINCT index pairs with
    multiplicities -- (pairs int32 [R', 2], mult int32 [R'], indptr int64 [2B+1]), segments in gather()'s order (left
    blocks, then right blocks).  pairs[r] = (SFptr+1 of a member in its own row, in the partner row or 0): the row
    gather() would have emitted is encode[pairs[r]] and it occurs mult[r] times in its segment, so for a first model
    stage of the form f(xz).sum(-2) = f(encode)[pa] + f(encode)[pb] (model.py:78) any aggregation that is a weighted
    function of the rows -- mean, the attention gate of model.py:59-62 -- is exact over these rows with `mult` as
    weights, at ~1/10 of the rows (a set of ~400 members carries a few dozen distinct LP rows)."""
    spg = _as_spg(x)
    if isinstance(spg, StridedSpG):
        spg = spg.to_csr()
    if spg.data.dtype != torch.int32 or getattr(spg, "keyrows", False):
        raise TypeError("gather_pairs needs an SFptr (integer) SpG (not a keyed() one)")
    L, dev, st = lib(), spg.device, stream_ptr()
    e = _as_rows(edge, dev)
    B = e.shape[1]
    own = torch.cat([e[0], e[1]]).contiguous()
    partner = torch.cat([e[1], e[0]]).contiguous()
    S = 2 * B
    seg, flags = _seg_and_flags(S, dev)
    ws = torch.empty(L.subgacc_sjoin_workspace_bytes(S), dtype=torch.uint8, device=dev)
    check(L.subgacc_sjoin_sizes(ptr(spg.indptr), spg.n_rows, ptr(own), ptr(partner), S, ptr(seg), ptr(flags), ptr(ws),
                                ws.numel(), st))
    R = _size_and_row_check(seg, S, flags, spg.n_rows)
    pairs = torch.empty((R, 2), dtype=torch.int32, device=dev)
    mult = torch.empty(R, dtype=torch.int32, device=dev)
    cnt = torch.zeros(S, dtype=torch.int32, device=dev)
    with _timed("sjoin_pairs"):
        join_fill(JOIN_PAIRS, JOIN_SFPTR, row_off=spg.indptr, n_rows=spg.n_rows, ids=spg.indices, payload=spg.data, max_len=spg.max_len,
                  own=own, partner=partner, S=S, seg=seg, pair_block=B, out_pairs=pairs, out_mult=mult, out_cnt=cnt, flags=flags)
    # rows of segment j sit at [seg[j], seg[j] + cnt[j]): close the gaps (R' ~ R/10 elements from here on)
    indptr = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    torch.cumsum(cnt, 0, out=indptr[1:])
    Rc = int(indptr[-1].item())
    segid = torch.repeat_interleave(torch.arange(S, device=dev), cnt.long(), output_size=Rc)
    src = seg[:-1][segid] + (torch.arange(Rc, device=dev) - indptr[:-1][segid])
    _checked(pairs, indptr, flags)
    return pairs[src], mult[src], indptr


def attn_stage(edge, x, encode, embed, gate_nn, value_nn=None):
    """The reference's first model stage for --aggr attn, fused over the pair form of the join:  model.py:59-62,78-81
        x = pe_embedding(xz).sum(dim=-2)
        xl, xr = AttentionalAggregation(gate_nn, nn)(x, ptr=ptr).view(2, -1, H)
              = segment_sum(softmax_segment(gate_nn(x)) * nn(x))
    A row's x is e[pa] + e[pb] with e = embed(encode) ([c+1, H], tiny): gate and value are evaluated once per DISTINCT
    pair of a segment and the softmax / weighted sum carry the multiplicities -- exact (up to fp32 summation order),
    ~10x fewer rows than xz, and autograd reaches embed / gate_nn / value_nn through ordinary torch ops.
    Returns float32 [2, B, H'] (left endpoints, right endpoints); empty segments give zero rows (as PyG's do)."""
    table = encode if torch.is_tensor(encode) else torch.as_tensor(encode)
    spg = _as_spg(x)
    table = table.to(device=spg.device, dtype=torch.float32)
    pairs, mult, indptr = gather_pairs(edge, spg)
    if pairs.numel() and int(pairs.max().item()) >= table.shape[0]:
        raise IndexError(f"index {int(pairs.max().item())} is out of bounds for the encode table with {table.shape[0]} rows")
    S = indptr.numel() - 1
    e = embed(table)
    xr = e[pairs[:, 0].long()] + e[pairs[:, 1].long()]
    g = gate_nn(xr).reshape(-1)
    v = value_nn(xr) if value_nn is not None else xr
    seg = torch.repeat_interleave(torch.arange(S, device=spg.device), indptr[1:] - indptr[:-1], output_size=pairs.shape[0])
    gmax = torch.full((S,), float("-inf"), device=g.device, dtype=g.dtype).scatter_reduce(0, seg, g.detach(), "amax")
    w = mult.to(g.dtype) * torch.exp(g - gmax[seg])                 # softmax numerators, with multiplicity
    den = torch.zeros(S, device=g.device, dtype=g.dtype).index_add_(0, seg, w)
    num = torch.zeros((S, v.shape[-1]), device=g.device, dtype=v.dtype).index_add_(0, seg, w[:, None] * v)
    out = num / (den + 1e-16)[:, None]                              # torch_geometric.utils.softmax adds the same 1e-16
    return out.view(2, -1, out.shape[-1])


def gather_index(edge, x, device=None):
    """Index form of gather() (the variant that writes 8 instead of 8k bytes per row): (pairs int32 [R, 2],
    indptr int64 [2B+1]) with pairs[r] = (SFptr+1 of the member in its own row, in the partner row or 0) -- the row gather()
    would have emitted is encode[pairs[r]], in gather()'s row order (members sorted by node id inside a segment)."""
    spg = _as_spg(x)
    if isinstance(spg, StridedSpG):     # index pairs need the numbering of the distinct LP rows (transient batches skip it)
        if spg.keyrows:
            spg = spg.to_csr()
        else:
            spg.sets.number()
    e = _as_rows(edge, spg.device)
    own = torch.cat([e[0], e[1]])
    partner = torch.cat([e[1], e[0]])
    return _checked(*sjoin(spg, own, partner, None, ptr_mode=True, return_index=True, pair_block=e.shape[1]))


def lstm_stage(edge, x, encode, embed, lstm):
    """The reference's first model stage for --aggr lstm, over the index form of the join:  model.py:63-65,78-83
        x = pe_embedding(xz).sum(dim=-2);  xl, xr = LSTMAggregation(H, H)(x, index=ptr).view(2, -1, H)
    LSTMAggregation packs the rows of every segment, in row order, into a dense [S, L, H] batch padded with zero rows
    (L = the longest segment), runs `lstm` (batch_first) over it and returns the output at the LAST position L-1 -- the
    padding is part of its arithmetic, so it is reproduced.  A row's x is e[pa] + e[pb] with e = embed(encode) ([c+1, H]):
    the stage reads 8 bytes per row instead of xz's 8k and never evaluates the MLP on [R, 2, k].
    Returns float32 [2, B, H'] (left endpoints, right endpoints)."""
    table = encode if torch.is_tensor(encode) else torch.as_tensor(encode)
    spg = _as_spg(x)
    table = table.to(device=spg.device, dtype=torch.float32)
    pairs, indptr = gather_index(edge, spg)
    if pairs.numel() and int(pairs.max().item()) >= table.shape[0]:
        raise IndexError(f"index {int(pairs.max().item())} is out of bounds for the encode table with {table.shape[0]} rows")
    S = indptr.numel() - 1
    lens = indptr[1:] - indptr[:-1]
    L = int(lens.max().item()) if S else 0
    e = embed(table)
    rows = e[pairs[:, 0].long()] + e[pairs[:, 1].long()]
    seg = torch.repeat_interleave(torch.arange(S, device=spg.device), lens, output_size=pairs.shape[0])
    posn = torch.arange(pairs.shape[0], device=spg.device) - indptr[:-1][seg]
    dense = rows.new_zeros((S, max(L, 1), rows.shape[-1]))
    dense[seg, posn] = rows
    out = lstm(dense)[0][:, -1]
    return out.view(2, -1, out.shape[-1])
