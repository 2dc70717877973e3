"""One step of the on-demand path (sample both endpoints of B pairs -> SpG rows -> SpJoin) captured as ONE HIP graph.

The reference's online loop runs batches of 1,024 pairs (main.py:32, train.py:120-127).  At that size the GPU work of
a step (tens of microseconds) is smaller than the cost of launching its ~25 kernels one by one, so the step is
captured once for a fixed (B, M, m) -- the C ABI allocates nothing and never synchronises (include/subgacc.h), torch's
allocator serves a capture from a private pool -- and replayed with new pairs copied into a static input buffer.
Sizes and status flags stay on the device during the replay (the lazy forms of sampler / spjoin); finish() reads them
back in one small copy.  Results are the ones the eager path gives (tests/test_gpu_join.py).

Philox streams are keyed by (seed, root id, walk, step) and the seed is baked into the captured launches: a root's set
is the same in every replay -- the semantics of the reference's offline stage, where every node's set is sampled once
and joined in every epoch (main.py:172-178).
"""
import ctypes as _ctypes
import os

import torch

from . import _lib
from .sampler import check_walk_flags, unpack_status
from .spjoin import StepBuffers, _dedup_tick, sample_and_gather


_DEBUG = os.environ.get("SUBGACC_DEBUG", "0") == "1"


class CapturedStep:
    def __init__(self, csr, pairs, num_walks=200, num_steps=3, seed=111413, rng="philox", uniq_capacity=1 << 17,
                 strided=None, fused=None, warmup=2, dedup_roots=False, batch=None):
        """pairs = B, the fixed number of query pairs per step; num_steps = walk hops.  dedup_roots=True: every distinct
        endpoint of a batch is sampled once (StepBuffers(dedup_roots=True): its per-step stamp lives on the device, so the
        replayed graph works like a launched step); `distinct_roots` after finish()."""
        self.csr, self.B, self.M, self.m = csr, int(pairs), int(num_walks), int(num_steps)
        dev = csr.device
        # batch=b: the step takes pairs/b reference-sized batches as [nb, 2, b] (StepBuffers(batch=b)); finish_batches() cuts the result
        self.batch = self.B if batch is None else int(batch)
        if self.batch != self.B and (rng != "philox" or strided is False or fused is False or dedup_roots):
            raise ValueError("CapturedStep(batch=) needs the buffered form of the step (rng='philox', fused strided rows, no root dedup)")
        self.edge = torch.zeros((2, self.B) if self.batch == self.B else (self.B // self.batch, 2, self.batch), dtype=torch.int64, device=dev)
        self.out = torch.empty(2 * self.B * (self.M * self.m + 1) * 2 * (self.m + 1), dtype=torch.float32, device=dev)
        self._kw = dict(num_walks=self.M, num_steps=self.m, seed=seed, rng=rng, out=self.out, lazy=True, strided=strided,
                        fused=fused, uniq_capacity=uniq_capacity)
        if rng == "philox" and strided is not False and fused is not False:
            try:      # the allocation-free six-launch form of the step (spjoin.StepBuffers) where it applies
                self._kw["buffers"] = StepBuffers(csr, self.B, self.M, self.m, uniq_capacity=uniq_capacity, out=self.out,
                                                  dedup_roots=dedup_roots, batch=batch)
                self._kw["dedup_roots"] = bool(dedup_roots)
            except ValueError:
                if dedup_roots or self.batch != self.B:
                    raise
        elif dedup_roots:
            raise ValueError("dedup_roots=True needs the buffered form of the step (rng='philox', fused strided rows)")
        self._dedup_bufs = self._kw["buffers"] if dedup_roots else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                   # allocator steady state + lazy code-object loads, uncaptured
            for _ in range(max(int(warmup), 1)):
                self._queue()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # what finish() needs: the packed status of the sets (flags, distinct LP rows, members) and the join's row count --
        # with StepBuffers they sit next to each other ([rows, status x4(, distinct roots)]) and their copy to pinned host
        # memory is a node of the graph: a replay costs the host one launch, not one launch and one copy
        bufs = self._kw.get("buffers")
        tail_n = (6 if dedup_roots else 5) if bufs is not None else 0
        self._host = torch.empty(tail_n if tail_n else 5, dtype=torch.int64, pin_memory=True)
        self._copy_in_graph = False
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.xz, self.ind, self.sets = self._queue()
            if self.sets._tail is not None and self.sets._tail.numel() == self._host.numel():
                _lib.publish(self.sets._tail, self._host)
                self._copy_in_graph = True
        self.status, self._rows = self.sets.status, self.ind[-1:]
        self._tail = self.sets._tail          # StepBuffers: [rows, status x4] contiguous -> one copy
        if self._tail is None and self._host.numel() != self.status.numel() + 1:
            self._host = torch.empty(self.status.numel() + 1, dtype=torch.int64, pin_memory=True)
        self.distinct_roots = None
        self._event = torch.cuda.Event()

    def _queue(self):
        return sample_and_gather(self.csr, self.edge, **self._kw)

    def __call__(self, edge, stream=None):
        """queue one step for `edge` [2, B] (node ids, on the device): copy-in, graph replay, status on its way to
        pinned host memory.  Nothing waits for the GPU here.  stream: the current stream, if the caller has it at hand
        (spares torch's look-up of it, ~8 us)."""
        if tuple(edge.shape) != tuple(self.edge.shape):
            raise ValueError(f"this step was captured for {list(self.edge.shape)} pairs")
        if edge is not self.edge:
            self.edge.copy_(edge, non_blocking=True)
        if self._dedup_bufs is not None:
            _dedup_tick(self._dedup_bufs)
        bufs = self._kw.get("buffers")
        if bufs is not None and (self.sets._keyctx is not None or self.sets._fresh is not None):
            # the buffers take a new batch: what number() / X said about the previous one is void, and self.sets now stands for this one
            bufs.step_id = sid = getattr(bufs, "step_id", 0) + 1
            self.sets._fresh = lambda: getattr(bufs, "step_id", 0) == sid
            if self.sets._keyctx is not None:
                self.sets._keyctx["fresh"] = self.sets._fresh
            self.sets.ukeys = self.sets._ktable = None
        self.graph.replay()
        if self._copy_in_graph:
            pass
        elif self._tail is not None:
            _lib.publish(self._tail, self._host)
        else:
            self._host[:-1].copy_(self.status, non_blocking=True)
            self._host[-1:].copy_(self._rows, non_blocking=True)
        if stream is None:
            self._event.record()
        else:
            self._event.record(stream)
        return self

    def finish(self):
        """wait for the queued step, raise on its errors -> (xz float32 [R,2,k] view of the static buffer, indptr)"""
        self._event.synchronize()
        words = self._host.tolist()
        if self._tail is not None:
            if len(words) > 5:                # root dedup: [rows, status x4, distinct roots]
                self.distinct_roots = int(words[5])
                words = words[:5]
            words = words[1:] + words[:1]
            words[3] = words[4]               # members = rows of the join (every row belongs to an own set; with root dedup: an upper bound)
        st = unpack_status(words[:-1]) + words[-1:]
        check_walk_flags(self.sets, st[:4])
        if st[2] or (self.sets.ukeys is not None and st[4] > self.sets.ukeys.numel()):
            raise _lib.SubgAccError("the table of distinct LP rows overflowed in a captured step: capture it again with a "
                                    "larger uniq_capacity")
        self.distinct_rows, self.members = (st[4] if self.sets.ukeys is not None else None), st[5]   # None: not numbered
        return self.xz[: st[6]], self.ind


    def finish_batches(self):
        """finish() for a step captured with batch=b: [(xz_i, indptr_i)] for its pairs/b reference-sized batches (spjoin.split_batches)"""
        from .spjoin import split_batches
        xz, ind = self.finish()
        views = split_batches(xz, ind, self.batch)
        views._resolve()      # (the pointers live in the step's static buffers: read them now, before the next replay rewrites them)
        return views


class CapturedStepPool:
    """`lanes` captured steps, each replayed on its own HIP stream: a serving loop for small batches.

    A step of 1,024 pairs (the reference's default, main.py:32) is 2,048 roots -- one workgroup per resident slot for the
    length of one root's walk -- plus a dozen kernels of a few microseconds: it cannot fill the chip, and one stream
    runs it in ~100 us.  Consecutive batches are independent, so several of them in flight overlap almost perfectly
    (B = 1,024 on the cit2-like graph: 10 M pairs/s with one captured step in flight, 13 M with two, 19 M with four).

        pool = CapturedStepPool(csr, 1024, lanes=4, num_walks=200, num_steps=3)
        t = pool.submit(edge)             # queues the batch on the next lane (which must be free); nothing waits
        xz, indptr = pool.finish(t)       # waits for that batch, raises on its errors, frees the lane

    (xz, indptr) are views of the lane's static buffers: valid until that lane is submitted again, `lanes` submits later."""

    def __init__(self, csr, pairs, lanes=4, **kw):
        self._lanes([CapturedStep(csr, pairs, **kw) for _ in range(int(lanes))], csr.device)

    def _lanes(self, steps, device):
        self.steps, self.device = steps, device
        self.streams = [torch.cuda.Stream(device=device) for _ in self.steps]
        self._busy = [False] * len(self.steps)
        self._next = 0
        self._caller = torch.cuda.current_stream(device)     # looked up once: torch's stream bookkeeping costs ~6 us a call

    def submit(self, edge, stream=None, sync=True):
        """stream: the stream `edge` was produced on -- MANDATORY when the caller's current stream is not the one this pool was
        created on (the pool does not look the current stream up: ~6 us of the ~35 us a step costs the host; afterwards the
        current stream is `stream`, or the creation stream; SUBGACC_DEBUG=1 checks it).  sync=False: `edge` is
        known to be complete (produced and synchronised earlier): the lane does not wait for the caller's stream (~10 us of
        host time).  edge may be `pool.steps[lane].edge` itself (filled in place by the caller): no copy then."""
        i = self._next
        if self._busy[i]:
            raise RuntimeError("every lane is in flight: finish() the oldest batch before submitting another")
        self._next = (i + 1) % len(self.steps)
        st = self.streams[i]
        caller = self._caller if stream is None else stream
        if _DEBUG and torch.cuda.current_stream(self.device) != caller:
            raise RuntimeError("CapturedStepPool.submit: the current stream is not the pool's creation stream -- pass stream=")
        if sync:
            st.wait_stream(caller)                     # `edge` may have been produced on the caller's stream just now
        torch.cuda.set_stream(st)                      # (not `with torch.cuda.stream(st)`: it asks for the current stream twice)
        try:
            self.steps[i](edge, stream=st)
        finally:
            torch.cuda.set_stream(caller)
        self._busy[i] = True
        return i

    def finish(self, ticket):
        self._busy[ticket] = False
        return self.steps[ticket].finish()


class CapturedJoin:
    """gather(edge, z, encode) over a RESIDENT store for a fixed number of pairs, with everything a call needs built once: the
    reference's online loop joins one batch after the other from the store it sampled once (train.py:120-127), and for short rows
    -- the top-100 PPR store: 65,536 pairs are 64 us of fill -- what surrounds the fill decides the rate.  One call is ONE entry
    into the library (subgacc_sjoin_fill_v2 with SUBGACC_JOIN_OPT_SIZES: the size pass as a single launch, the fill behind it) on a
    descriptor and buffers made here; the row count and the status word arrive in pinned host memory by themselves -- no memset,
    no read-back copy, no allocation, ~10 us of host time.  Same results as gather().

        cj = CapturedJoin(z, 65536)                    # float payload; or CapturedJoin(z, B, encode=table) / (zk, B, encode=zk.slot_table())
        cj(edge); xz, indptr = cj.finish()             # views of the object's static buffers, valid until the next call

    triplets=True: the same for hgather(hedge [3, B], z, encode) (train.py:48-72, main_horder.py's batches of 2,048 triplets): four
    blocks [U|w ; W|u ; V|w ; W|v], finish() -> (xz, segment ids) as hgather returns them; one small kernel lays the four blocks out.

    graph=True replays the same launches as ONE HIP graph (rounds 3-4's form; the name is from there): less host time still for
    batches of ~1,024 pairs, but every replay costs the GPU ~20 us between graphs (`profiles/r24_join_call_probe.log`)."""

    def __init__(self, z, pairs, encode=None, warmup=2, graph=False, triplets=False):
        import ctypes as C
        from .spg import KEY_ROWS_ENCODE
        self.z, self.B, self.encode, self.triplets = z, int(pairs), encode, bool(triplets)
        dev = z.device
        B, S = self.B, (4 if triplets else 2) * self.B
        if triplets and encode is None:
            raise NotImplementedError          # (train.py:69-70)
        if triplets and graph:
            raise ValueError("triplets=True is the one-call form (graph=False)")
        d = _lib.JoinDesc()
        d.struct_bytes, d.form, d.options = C.sizeof(_lib.JoinDesc), _lib.JOIN_ROWS, _lib.JOIN_OPT_SIZES
        if getattr(z, "keyrows", False):
            if encode is not KEY_ROWS_ENCODE:
                raise ValueError("a keyed() store is joined with encode=zk.slot_table()")
            k, d.payload_kind, d.num_walks, d.num_steps = z.key_m + 1, _lib.JOIN_KEY32, z.key_M, z.key_m
        elif z.data.dtype == torch.float64:
            if encode is not None:
                raise TypeError("a float-payload SpG is joined without an encode table (train.py:39-43)")
            k, d.payload_kind = 1, _lib.JOIN_F64
        else:
            if encode is None:
                raise NotImplementedError("an integer SpG needs the encode table")
            self._table = encode.to(device=dev, dtype=torch.float32).contiguous()
            if self._table.shape[0] <= z.max_data:
                raise IndexError(f"index {z.max_data} is out of bounds for the encode table with {self._table.shape[0]} rows")
            k, d.payload_kind = int(self._table.shape[1]), _lib.JOIN_SFPTR
            d.table, d.table_rows, d.k = self._table.data_ptr(), self._table.shape[0], k
        self.edge = torch.zeros((3 if triplets else 2, B), dtype=torch.int64, device=dev)
        self.out = torch.empty(S * z.max_len * 2 * k, dtype=torch.float32, device=dev)
        self.ind = torch.zeros(S + 1, dtype=torch.int64, device=dev)
        self.flags = torch.zeros(4, dtype=torch.int32, device=dev)
        self._state = torch.zeros(_lib.lib().subgacc_sjoin_workspace_bytes(S), dtype=torch.uint8, device=dev)     # zeroed ONCE
        self._host = torch.zeros(2, dtype=torch.int64, pin_memory=True)
        if hasattr(z, "pitch"):       # HeadedSpG (SpG.aligned()): rows on whole lines, their lengths in their first slots, no row pointers
            d.row_stride, d.n_rows, d.ids, d.payload = z.pitch, z.n_rows, z.ids.data_ptr(), z.data.data_ptr()
        else:
            d.row_off, d.n_rows, d.ids, d.payload, d.max_len = z.indptr.data_ptr(), z.n_rows, z.indices.data_ptr(), z.data.data_ptr(), z.max_len
        d.own, d.S, d.pair_block = self.edge.data_ptr(), S, B
        self.segid = None
        if triplets:       # own = [u | w | v | w] (hedge rows 0, 2, 1, 2), the mirrored partner blocks are derived by the kernels
            self._blocks = torch.empty((4, B), dtype=torch.int64, device=dev)
            self._sel = torch.tensor([0, 2, 1, 2], dtype=torch.int64, device=dev)
            self.segid = torch.empty(S * z.max_len, dtype=torch.int64, device=dev)
            d.own, d.out_segid = self._blocks.data_ptr(), self.segid.data_ptr()
        d.out_xz, d.flags, d.out_seg = self.out.data_ptr(), self.flags.data_ptr(), self.ind.data_ptr()
        d.size_state, d.size_state_bytes, d.host_tail = self._state.data_ptr(), self._state.numel(), self._host.data_ptr()
        self._d, self._ref = d, C.byref(d)
        self._fill = _lib.lib().subgacc_sjoin_fill_v2
        self._own = None
        self.xz = self.out.view(S * z.max_len, 2, k)
        self._event = torch.cuda.Event()
        self.graph = None
        if graph:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(int(warmup), 1)):
                    self._launch()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._launch()            # (the capturing stream is the current one)

    def _launch(self, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream(self.edge.device)
        _lib.check(self._fill(self._ref, _ctypes.c_void_p(st.cuda_stream)))
        return st

    def _call_triplets(self, hedge, stream):
        if stream is not None and torch.cuda.current_stream(self.edge.device) != stream:
            with torch.cuda.stream(stream):
                return self._call_triplets(hedge, stream)
        if not (torch.is_tensor(hedge) and hedge.dtype == torch.int64 and hedge.device == self.edge.device):
            self.edge.copy_(torch.as_tensor(hedge), non_blocking=True)
            hedge = self.edge
        torch.index_select(hedge, 0, self._sel, out=self._blocks)
        self._event.record(self._launch(stream))
        return self

    def __call__(self, edge, stream=None):
        """queue the join of `edge` [2, B] on `stream` if given (the calling convention of CapturedStep.__call__ /
        CapturedStepPool.submit), else on the current one; a contiguous int64 device tensor is joined where it lies (and kept
        alive until the next call), anything else goes through the object's static buffer"""
        if tuple(edge.shape) != tuple(self.edge.shape):
            raise ValueError(f"this join was made for {list(self.edge.shape)} endpoints")
        if self.triplets:
            return self._call_triplets(edge, stream)
        if self.graph is None and torch.is_tensor(edge) and edge.dtype == torch.int64 and edge.device == self.edge.device and edge.is_contiguous():
            self._own = edge
            self._d.own = edge.data_ptr()
            self._event.record(self._launch(stream))
            return self
        if stream is not None and torch.cuda.current_stream(self.edge.device) != stream:
            with torch.cuda.stream(stream):
                return self(edge, stream)
        if edge is not self.edge:
            self.edge.copy_(torch.as_tensor(edge), non_blocking=True)
        self._own, self._d.own = None, self.edge.data_ptr()
        if self.graph is not None:
            self.graph.replay()
            self._event.record()
        else:
            self._event.record(self._launch(stream))
        return self

    def finish(self):
        """wait for the queued join, raise on its errors -> (xz float32 [R,2,k] view of the static buffer, indptr)"""
        self._event.synchronize()
        rows, word = self._host.tolist()
        # `word` is the SIZE pass's status (bits 16 and 64: every row index is looked at there).  What a fill could add -- a row longer
        # than max_len (1), an SFptr outside the encode table (2) -- is ruled out when this object is built (max_len and max_data are
        # the store's own; the table is checked against them), so the fills' word, flags[3], is read back with SUBGACC_DEBUG=1 only;
        # it is zeroed after a report either way, so that one bad batch does not fail every batch after it.
        if _DEBUG:
            word |= int(self.flags[3].item())
        if word & 64:
            # the one-pass scan's state was not zero when the launch began (an aborted launch before it): nothing it wrote means
            # anything, and it may not have left the state clean either -- zeroed here, on the host's side, before the next call
            self._state.zero_()
            self.flags.zero_()          # (bit 64 in the device flags keeps every fill from running)
            raise _lib.SubgAccError("the join's size state was not clean (an aborted launch?): zeroed, call again")
        if word & (16 | 1 | 2):
            self.flags.zero_()
        if word & 16:
            raise IndexError(f"row index out of range for an SpG with {self.z.n_rows} rows")
        if word & 1:
            raise _lib.SubgAccError("SpG row longer than SpG.max_len")
        if word & 2:
            raise IndexError("SFptr outside the encode table")
        return self.xz[:rows], (self.segid[:rows] if self.triplets else self.ind)


class CapturedJoinPool(CapturedStepPool):
    """`lanes` CapturedJoins over one resident store, each on its own HIP stream, taken in turn -- the serving loop for joins of
    short rows: the size pass and the first waves of one batch's fill run under the last waves of the batch before it (the top-100
    PPR store, 65,536 pairs: 1.02 G pairs/s with two lanes against 0.83 G one call after the other on one stream,
    `profiles/r24_join_call_probe.log`).  submit() / finish() as CapturedStepPool's:

        pool = CapturedJoinPool(z, 65536)              # or (z, B, encode=table) / (zk, B, encode=zk.slot_table())
        t = pool.submit(edge)                          # queues the batch on the next lane (which must be free); nothing waits
        xz, indptr = pool.finish(t)                    # views of the lane's buffers: valid until that lane is submitted again"""

    def __init__(self, z, pairs, lanes=2, encode=None, **kw):
        self._lanes([CapturedJoin(z, pairs, encode=encode, **kw) for _ in range(int(lanes))], z.device)

