"""Device-resident node-set sampler: walks -> per-root dedup + LP counts -> global unique LP rows.

Host side of the kernels in csrc/walk.hip and csrc/uniq.hip.  Everything stays in HBM.  Mirrors the stages of
set_sampler (reference subg_acc/subg_acc.c:736-1005).

Host round trips.  The eager form reads one 8-byte total per chunk of roots (to size the packed arrays exactly)
and one status word at the end.  The `lazy` form (single chunk) sizes its arrays by their upper bound
n*(M*m+1), leaves every count on the device and reads nothing back: sizes, flags and the distinct-row count
are fetched in one copy when the caller first asks for them (SampledSets.resolve()), so a whole
sample -> SpG -> SpJoin step queues up asynchronously.
"""
import ctypes
import functools
import types
import os
import threading
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import WalkCfg, check, lib, ptr, stream_ptr

# strided staging budget per chunk of roots (bytes); stride*12 B per root
STAGING_BYTES = 6 << 30
UNIQ_CAPACITY = 1 << 20
FUSED_MAX_Q = 818       # subgacc_walk_spg keeps 4 table slots per lane in registers (per-root table <= 1024 slots)
RANK_LIMIT = 16384      # distinct LP rows that the table-only numbering ranks directly
FINISH_MAX_STRIDE = 1024   # subgacc_finish_rows sorts a row from registers (16 members per lane of one wave)

# Packed hop records (csrc/walk_rows.hip, REC form): 8 bytes per CSR entry holding the neighbour, its row begin and its degree,
# so that a hop of the fused-row kernel is ONE dependent random read instead of two (-8 % kernel time on the cit2-like graph,
# -4 % on ppa; a graph that lives in L2 gains nothing from an array twice the size of its adjacency).  "auto": built (once
# per DeviceCSR, on first use) for int32-offset graphs whose adjacency is beyond the L2s and whose degrees fit the field;
# "1" / "0" force / forbid.  Results are identical either way.
HOP_RECORDS = os.environ.get("SUBGACC_HOP_RECORDS", "auto")
HOP_RECORDS_MIN_BYTES = 64 << 20      # adjacency bytes from which records are built in "auto" mode
HOP_RECORDS_MIN_DEG_BITS = 12
HOP_RECORDS_MAX_FREE_FRACTION = 0.25    # of the free device memory, in "auto" mode
# a single-chunk batch of SORT_ROOTS_MIN .. SORT_ROOTS_MAX roots is walked in ascending order of root id (sample_sets(sort_roots=True);
# the buffered step: StepBuffers(sort_roots=True)).  Nothing observable changes; cit2-like step +5.7 % pairs/s, twitter-like +5.6 %.
SORT_ROOTS_MIN, SORT_ROOTS_MAX = 16384, 1 << 20

# Key rows (csrc/walk_rows.hip KR form + subgacc_sjoin_fill_keyrows): a strided batch that will not be numbered carries its
# members' 32-bit LP keys instead of slots of a table of distinct rows; the join unpacks a key into its feature row itself.
# No table, no registration, no unpack pass: -14 % walk-kernel time on the cit2-like batch, -23 % on collab.  Asking such a
# batch for its numbering afterwards (number(), c, enc_int16(), to_csr()) samples it again with the table form.
# sample_sets(key_rows=False) / StepBuffers(key_rows=False) keep the table form (tests cross both).


# Batched registration (csrc/keyrows.hip): a store that is KEPT (subg_matrix: packed rows + numbered LP rows) is sampled with the
# key-rows kernel too; its keys are registered in the table of distinct rows by one pass over the rows, and only the few
# thousand roots that may be the first to show a row are walked again for their first-visit order (subgacc_walk_tags).
# sample_sets(batched_registration=False): the table form of the walk kernel registers inside every root's epilogue, as before
# round 3 (tests cross both; the shapes key rows do not serve -- keys beyond 31 bits -- always take it).  Results are identical.


def _table_slots(num_walks, num_steps):
    q = num_walks * num_steps + 1
    t = 64
    while t < q + q // 4 + 1:
        t <<= 1
    return t


def rows_kernel_takes(num_walks, num_steps, bucket=-1):
    """does the specialised fused-row kernel (csrc/walk_rows.hip: launch_walk_rows) take this shape?  set_sampler order, first hop
    without replacement and no raw walks are the caller's to check; a truncating `bucket` leaves it to walk_sets_kernel<SPG>."""
    return (bucket <= 0 and 2 <= num_steps <= 4 and num_walks <= 256 and _table_slots(num_walks, num_steps) in (512, 1024))


def key_rows_form(num_walks, num_steps):
    """the key-rows form of the fused-row kernel for this shape: 32 (the LP key fits 31 bits: 2 to 4 hops, a 512 / 1,024-slot
    table -- every reference configuration up to 3 hops, and 4 hops up to M = 127), 64 (4 hops with a longer key: M = 128 .. 204,
    e.g. the paper's citation2 sampler setting m = 4, M = 200 -- 33 bits; 1,024-slot table), or 0 (none: the table form)"""
    bits = num_steps * int(num_walks).bit_length() + 1
    t = _table_slots(num_walks, num_steps)
    if num_walks > 256 or num_steps not in (2, 3, 4):
        return 0
    if bits <= 31 and t in (512, 1024):
        return 32
    if num_steps == 4 and bits <= 63 and t == 1024:
        return 64
    return 0


def key_rows_ok(num_walks, num_steps):
    """32-bit key rows for this shape? (what the batched registration of a kept store, csrc/keyrows.hip, works on)"""
    return key_rows_form(num_walks, num_steps) == 32


# bench.py sets this to a callable(name) -> context manager that brackets one kernel launch with HIP events
# on the launch stream (roofline.achieved is measured live, not taken from a profile)
KERNEL_TIMER = None
_RECS_LOCK = threading.Lock()


class _NoTimer:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _timed(name):
    return KERNEL_TIMER(name) if KERNEL_TIMER is not None else _NoTimer()


class DeviceCSR:
    """Graph CSR resident in HBM: int32 node ids, int32 or int64 row offsets (`indptr64`)."""

    def __init__(self, indptr, indices, device=None, validate=True):
        """validate=True checks the CSR once, on the device, after the upload (row offsets monotone within
        [0, len(indices)], neighbour ids within [0, N)): the reference reads whatever the arrays say (a bad graph is a
        segfault there); on a GPU a stray address can take the device down, so nothing reaches a kernel unchecked.
        A few reductions over arrays that are resident anyway (3 ms for the 12 GB twitter-like graph)."""
        device = device or _lib.require_device()
        ip = indptr if torch.is_tensor(indptr) else torch.from_numpy(np.ascontiguousarray(indptr))
        ix = indices if torch.is_tensor(indices) else torch.from_numpy(np.ascontiguousarray(indices))
        if ip.dtype not in (torch.int32, torch.int64):
            raise TypeError("Input parsing error. (indptr must be int32 or int64)")
        if ix.dtype != torch.int32:
            raise TypeError("Input parsing error. (indices must be int32; the reference casts safely, "
                            "subg_acc.c:668)")
        self.indptr = ip.to(device).contiguous()
        self.indices = ix.to(device).contiguous()
        self.indptr64 = self.indptr.dtype == torch.int64
        self.num_nodes = self.indptr.numel() - 1
        self.device = self.indptr.device
        if self.indptr.dim() != 1 or self.indices.dim() != 1 or self.num_nodes < 0:
            raise TypeError("Input parsing error. (indptr / indices must be 1-D, indptr non-empty)")
        if validate:
            self._validate()

    def _validate(self):
        ip, ix, N = self.indptr, self.indices, self.num_nodes
        z = torch.zeros((), dtype=torch.bool, device=self.device)
        bad = torch.stack([ip[0] != 0, ip[-1] > ix.numel(), (ip[1:] < ip[:-1]).any() if N else z,
                           (ix.min() < 0) if ix.numel() else z, (ix.max() >= N) if ix.numel() else z]).tolist()
        if bad[0] or bad[1] or bad[2]:
            raise IndexError("CSR row offsets are not monotone within [0, len(indices)]")
        if bad[3] or bad[4]:
            raise IndexError("CSR neighbour ids outside [0, num_nodes)")

    @property
    def nnz(self):
        return self.indices.numel()

    def hop_records(self, force=None, bits=None):
        """(recs int64 [nnz], id_bits, beg_bits) -- the packed hop records of this graph (include/subgacc.h:
        subgacc_hop_records_build), built once and kept -- or None where they do not apply (int64 row offsets, degrees
        that do not fit, a graph that lives in the caches, SUBGACC_HOP_RECORDS=0).  int32 row offsets: 8 bytes per entry
        (id_bits, beg_bits > 0); int64 row offsets: 16 bytes per entry (id_bits = beg_bits = 0).  `bits=(id_bits, beg_bits)`
        overrides the field widths of the 8-byte form (tests: a narrow degree field exercises the escape path)."""
        cached = getattr(self, "_recs", None)
        if cached is not None and force is None and bits is None:
            return cached[1]                       # decided before (by the policy, or by an explicit call): it sticks
        with _RECS_LOCK:       # a DeviceCSR is shared between threads and streams: built once, complete before it is published
            cached = getattr(self, "_recs", None)
            if cached is not None and force is None and bits is None:
                return cached[1]
            out = self._build_hop_records(force, bits)
            if out is not None and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(self.device).synchronize()
            self._recs = (tuple(bits) if bits is not None else None, out)
            return out

    def _build_hop_records(self, force, bits):
        mode = HOP_RECORDS if force is None else ("1" if force else "0")
        if mode == "0" or self.nnz == 0:
            return None
        L = lib()
        out = None
        big = mode == "1" or bits is not None or 4 * self.nnz >= HOP_RECORDS_MIN_BYTES
        rec_bytes = (16 if self.indptr64 else 8) * self.nnz
        if big and mode == "auto" and bits is None:
            # "auto" never spends more than a quarter of the free memory on an array that buys ~8 % of the walk kernel (the
            # twitter-like graph: 47 GB of records beside a 12 GB CSR), and a failed allocation means "walk the plain CSR"
            big = rec_bytes <= HOP_RECORDS_MAX_FREE_FRACTION * torch.cuda.mem_get_info(self.device)[0]
        try:
            if self.indptr64:                           # 16-byte form: {id | degree, row begin}
                if big:
                    recs = torch.empty(2 * self.nnz, dtype=torch.int64, device=self.device)
                    check(L.subgacc_hop_records_build(ptr(self.indptr), 1, ptr(self.indices), self.num_nodes, self.nnz, 0, 0, ptr(recs),
                                                      stream_ptr()))
                    out = (recs, 0, 0)
            else:
                ib, bb = ctypes.c_int32(0), ctypes.c_int32(0)
                deg_bits = L.subgacc_hop_records_format(self.num_nodes, self.nnz, ctypes.byref(ib), ctypes.byref(bb))
                if bits is not None:
                    ib, bb = ctypes.c_int32(int(bits[0])), ctypes.c_int32(int(bits[1]))
                    deg_bits = 64 - ib.value - bb.value
                if big and deg_bits >= (HOP_RECORDS_MIN_DEG_BITS if (mode == "auto" and bits is None) else 1):
                    recs = torch.empty(self.nnz, dtype=torch.int64, device=self.device)
                    check(L.subgacc_hop_records_build(ptr(self.indptr), 0, ptr(self.indices), self.num_nodes, self.nnz, ib.value,
                                                      bb.value, ptr(recs), stream_ptr()))
                    out = (recs, ib.value, bb.value)
        except torch.cuda.OutOfMemoryError:
            if mode != "auto":
                raise
            out = None
        return out


@dataclass
class SampledSets:
    """Output of the sampler, all on device.  remap = (ids, sf) and enc in the reference's terms.

    In lazy form `ids` / `slot` / `ukeys` are capacity-sized until resolve() trims them; `X` and `c` resolve."""
    nsize: torch.Tensor      # int32 [n]
    row_off: torch.Tensor    # int64 [n+1]
    ids: torch.Tensor        # int32 [X]   members, first-visit order per root (sorted by id in the fused SpG form)
    keys: torch.Tensor       # int64 [X]   packed LP row (bit pattern of the uint64 key); None when only slots are kept
    sf: torch.Tensor         # int32 [X]   index of the member's LP row in `ukeys` (None until asked for: get_sf())
    ukeys: torch.Tensor      # int64 [c]   distinct LP rows in first-occurrence order
    num_walks: int
    num_steps: int
    stride: int
    walks: torch.Tensor = None   # int32 [n, M*(m+1)] when requested
    n_overflow: int = 0
    slot: torch.Tensor = None    # int32 [X]   slot of the member's key in `table` (fused compaction + insert)
    table: torch.Tensor = None   # the HBM table of distinct LP rows (uint8 blob, layout of csrc/uniq_table.hpp)
    capacity: int = 0
    status: torch.Tensor = None  # lazy form: device int64 [flags (4 x int32 in 2 words), distinct rows, members], see unpack_status
    data: torch.Tensor = None    # fused SpG form: SFptr+1 per member (capacity-sized while lazy)
    strided: bool = False        # ids / slot are the strided staging arrays (row i at i*stride), see sample_sets
    _members: int = 0
    _pending: tuple = None       # prefetch(): (pinned host copy of status, event, device source)
    extra: list = None           # values of prefetch(extra=...) once resolved
    _tail: torch.Tensor = None   # StepBuffers form: int64 [5] = [rows of the join (= members), status words x4], contiguous
    #                              ([6] with root dedup: + the number of distinct roots = rows that were sampled)
    n_distinct: int = None       # StepBuffers(dedup_roots=True): so many rows (the first occurrences of the endpoints) hold sets
    keyrows: bool = False        # strided rows whose payload (`slot`) is the member's LP key: no table, no numbering
    key64: bool = False          # ... a 64-bit key (`slot` is int64: 4-hop walks with M >= 128), else 32 bits
    _table_form: object = None   # key64: the same batch sampled again with the table form, once number() / to_csr() needed it
    _fresh: object = None        # sets of a buffered step: () -> do the step buffers still hold THIS batch? (spjoin._buffered_step)
    _keyctx: dict = None         # keyrows: what number() needs to register the rows' keys (csr, roots, cfg, rng positions, capacity, fresh())
    _ktable: torch.Tensor = None  # keyrows: the table of distinct LP rows once number() has built it (capacity _kcap)
    _kcap: int = 0
    _kcount: torch.Tensor = None  # ... and the number of distinct rows, on the device
    _join_flags: torch.Tensor = None   # sample_and_gather(lazy=True): the join's int32[4] status words (resolve() checks them)

    # ------------------------------------------------------------------ lazy bookkeeping
    @property
    def pending(self):
        return self.status is not None

    def prefetch(self, extra=None):
        """Queue the read-back of sizes and status flags (plus `extra`, a small int64 device tensor) into pinned host
        memory, behind everything queued so far.  resolve() then waits for exactly this copy instead of for whatever
        the stream holds by the time it is called -- a serving loop queues the next batch in between."""
        if self.status is None or self._pending is not None:
            return self
        if self._tail is not None:      # the step's buffers keep the join's row count next to the status words: no cat
            src = self._tail
        else:
            parts = [self.status] + ([extra.reshape(-1).to(torch.int64)] if extra is not None else [])
            if self._join_flags is not None:      # a lazy join's status word rides along (last): resolve() raises for it
                parts.append(self._join_flags[3:4].to(torch.int64))
            src = parts[0] if len(parts) == 1 else torch.cat(parts)
        host = torch.empty(src.numel(), dtype=torch.int64, pin_memory=True)
        _lib.publish(src, host)
        ev = torch.cuda.Event()
        ev.record()
        _lib.keep_until(ev, host)
        self._pending = (host, ev, src)          # src stays alive until the copy has run
        return self

    def resolve(self):
        """Read sizes and status flags back (one small copy), raise on errors, trim capacity-sized arrays."""
        if self.status is None:
            return self
        if self._pending is not None:
            host, ev, _ = self._pending
            ev.synchronize()
            st = host.tolist()
            self._pending = None
        else:
            src = self._tail if self._tail is not None else self.status
            if self._tail is None and self._join_flags is not None:
                src = torch.cat([src, self._join_flags[3:4].to(torch.int64)])
            st = src.tolist()
        if self._tail is None and self._join_flags is not None:
            join_word, st = int(st[-1]), st[:-1]
            self._join_flags = None
            if join_word & 16:
                self.status = None
                raise IndexError("row index out of range for the SpG (lazy join: the row was joined as an empty row)")
        if self._tail is not None:      # [rows, w0, w1, w2, w3]: every row of the join is a member of an own set
            self.extra = st[:1]
            if len(st) > 5:             # root dedup: the sets of the distinct endpoints are fewer than the join's rows
                self.n_distinct = int(st[5])
            st = unpack_status(st[1:4] + st[:1])
        else:
            self.extra = st[self.status.numel():]
            st = unpack_status(st[: self.status.numel()])
        self.status = None
        check_walk_flags(self, st[:4])
        if st[2]:
            raise _lib.SubgAccError("the table of distinct LP rows overflowed: sample again with a larger "
                                    "uniq_capacity (or lazy=False, which retries by itself)")
        c, X = st[4], st[5]
        if self.ukeys is not None:              # (None: strided rows sampled with number_rows=False, see number())
            if c > self.ukeys.numel():
                raise _lib.SubgAccError(f"{c} distinct LP rows exceed the direct-ranking limit: sample with lazy=False")
            self.ukeys = self.ukeys[:c]
        if self.strided:
            # root dedup: the rows of repeated endpoints are empty, so the members are fewer than the join's rows -- counted when
            # somebody asks (X): a reduction + read-back here would stall a serving loop once per step behind the NEXT step's launches
            self._members = X if self.n_distinct is None else None
            return self
        self.ids = self.ids[:X]
        for name in ("slot", "keys", "data", "sf"):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t[:X])
        return self

    @property
    def X(self):
        self.resolve()
        if self.strided and self._members is None:
            # (root dedup: counted from the buffers' sizes on first use -- which by then may belong to a later batch)
            fresh = self._fresh if self._fresh is not None else (self._keyctx["fresh"] if self._keyctx is not None else None)
            if fresh is not None and not fresh():
                raise _lib.SubgAccError("the member count of this batch was not asked for before its step buffers took a later "
                                        "batch (X / nnz must be read before the buffers are re-used or the captured step replayed)")
            self._members = int(self.nsize.sum().item())
        return self._members if self.strided else self.ids.numel()

    @property
    def c(self):
        return self.resolve().number().ukeys.numel()

    def number(self):
        """Number the distinct LP rows (first-occurrence order, subg_acc.c:957-978) if that has not happened yet: strided
        rows sampled with number_rows=False skip it -- a join by table slot (StridedSpG.slot_table()) consults no
        numbering, and a transient batch is dropped after its join -- until ukeys / c / enc / to_csr() are asked for."""
        if self.ukeys is not None:
            return self
        self.resolve()
        if self.keyrows and self.key64:
            self.ukeys = self.table_form().ukeys
            return self
        if self.keyrows:        # the keys are all there: register them now, and ask the few candidate roots for their order
            return self._number_keyrows()
        L, dev, st = lib(), self.ids.device, stream_ptr()
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        max_unique = min(self.capacity, RANK_LIMIT)
        ukeys = torch.empty(max_unique, dtype=torch.int64, device=dev)
        ws = torch.empty(L.subgacc_uniq_number_workspace_bytes(self.capacity, 0), dtype=torch.uint8, device=dev)
        check(L.subgacc_uniq_number(ptr(self.table), self.capacity, None, 0, ptr(ukeys), max_unique, ptr(count), RANK_LIMIT,
                                    ptr(ws), ws.numel(), st))
        c = int(count.item())
        if c > max_unique:
            raise _lib.SubgAccError(f"{c} distinct LP rows exceed the direct-ranking limit: sample with lazy=False")
        self.ukeys = ukeys[:c]
        return self

    def table_form(self):
        """64-bit key rows: csrc/keyrows.hip registers 32-bit keys, so what such a batch does not carry -- the numbering of its
        distinct LP rows, enc, the packed CSR rows -- comes from sampling the batch AGAIN with the table form of the walk kernel
        (a set is a function of (seed, root) under Philox, of the root's place in the stream under rand_r: the same rows)."""
        if self._table_form is None:
            ctx = self._keyctx
            if ctx is None or not ctx["fresh"]():
                raise _lib.SubgAccError("this key-rows batch cannot be numbered any more: its step buffers hold a later batch "
                                        "(number() / c / enc_int16() / to_csr() must be asked for before the buffers are re-used)")
            cfg = ctx["cfg"]
            rng = "rand_r" if cfg.rng_mode == _lib.RNG_RAND_R else "philox"
            # the same walks: the configuration of the first sampling (first-hop rule, root-degree cap, seed) and, under rand_r,
            # the stream positions it entered at (they carry rng_streams / calls_before) -- not this function's defaults
            state = (ctx["rng_pos"], ctx["rng_seed"]) if (rng == "rand_r" and ctx.get("rng_pos") is not None) else None
            self._table_form = sample_sets(ctx["csr"], ctx["roots"], num_walks=self.num_walks, num_steps=self.num_steps, seed=cfg.seed,
                                           rng=rng, first_hop_wo=bool(cfg.first_hop_wo), cap_root_degree=bool(cfg.cap_root_degree),
                                           fused_rows=True, strided=True, number_rows=True, key_rows=False,
                                           uniq_capacity=int(ctx["capacity"]), sort_roots=False, rng_state=state)
        return self._table_form

    def _number_keyrows(self):
        """Number the distinct LP rows of a key-rows batch after the fact (csrc/keyrows.hip): one pass over the rows registers
        their keys (coarse tags), the candidate roots are walked again for their first-visit tags, the table is ranked.
        The rows themselves are not touched; to_csr() then copies them with SFptr+1 looked up in this table."""
        ctx = self._keyctx
        if ctx is None or not ctx["fresh"]():
            raise _lib.SubgAccError("this key-rows batch cannot be numbered any more: its step buffers hold a later batch "
                                    "(number() / c / enc_int16() / to_csr() must be asked for before the buffers are re-used)")
        L, dev, st = lib(), self.ids.device, stream_ptr()
        csr, q, cfg, n = ctx["csr"], ctx["roots"], ctx["cfg"], self.nsize.numel()
        cap = int(ctx["capacity"])
        while True:
            table = torch.empty(L.subgacc_uniq_table_bytes(cap), dtype=torch.uint8, device=dev)
            check(L.subgacc_uniq_reset(ptr(table), cap, st))
            words = torch.zeros(4, dtype=torch.int64, device=dev)      # [flags x4 (int32) | candidates | distinct rows]
            flags = words.view(torch.int32)[:4]
            ccap = L.subgacc_keyrows_cand_capacity(n)
            cand = torch.empty(ccap, dtype=torch.int32, device=dev)
            check(L.subgacc_keyrows_register(ptr(self.slot), ptr(self.nsize), n, self.stride, 0, ptr(table), cap, ptr(cand),
                                             ccap, ptr(words[2:3]), ptr(flags), st))
            nc = int(words[2].item())
            check(L.subgacc_walk_tags(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(q), n, 0, ptr(ctx["rng_pos"]),
                                      ptr(ctx["rng_seed"]), ptr(cand), ptr(words[2:3]), max(min(nc, n), 1), ptr(table), cap, ptr(flags), st))
            max_unique = min(cap, RANK_LIMIT)
            ukeys = torch.empty(max_unique, dtype=torch.int64, device=dev)
            ws = torch.empty(L.subgacc_uniq_number_workspace_bytes(cap, 0), dtype=torch.uint8, device=dev)
            check(L.subgacc_uniq_number(ptr(table), cap, None, 0, ptr(ukeys), max_unique, ptr(words[3:4]), RANK_LIMIT, ptr(ws),
                                        ws.numel(), st))
            host = unpack_status(words[:2].tolist() + [int(words[3].item())])
            if host[3] & 32:
                raise _lib.SubgAccError("internal: the candidate list outgrew its launch")
            if host[2]:                   # the table of distinct LP rows was too small: again with a larger one
                cap *= 4
                continue
            break
        c = host[4]
        if c > max_unique:
            raise _lib.SubgAccError(f"{c} distinct LP rows exceed the direct-ranking limit: sample with lazy=False")
        self.ukeys, self._ktable, self._kcap, self._kcount = ukeys[:c], table, cap, words[3:4]
        return self

    def _count_dev(self):
        """device views of (distinct-row count, member count) while lazy, else (None, None)"""
        if self.status is None:
            return None, None
        return self.status[2:3], self.status[3:4]

    # ------------------------------------------------------------------ views of the result
    def get_sf(self):
        """int32 [X]: remap[1] of the reference.  Materialised on demand from the table slots."""
        if self.sf is None:
            if self.slot is None:
                raise ValueError("the sets were sampled with dedup=False")
            self.resolve()
            sf = self.slot.clone()
            check(lib().subgacc_uniq_translate(ptr(self.table), self.capacity, ptr(sf), self.X, None, 0, stream_ptr()))
            self.sf = sf
        return self.sf

    def enc_int16(self):
        """int16 [c, m+1]: the reference's `enc` (subg_acc.c:982-1000)."""
        self.number()
        out = torch.empty((self.c, self.num_steps + 1), dtype=torch.int16, device=self.ids.device)
        check(lib().subgacc_unpack_lp(ptr(self.ukeys), self.c, None, self.num_walks, self.num_steps, ptr(out), None, None,
                                      0, stream_ptr()))
        return out

    def counts_int32(self):
        """int32 [X, m+1]: per-member landing counts (rpe_encoder's second output, subg_acc.c:281-303)."""
        if self.keys is None:
            raise ValueError("the packed keys were not kept (sample_sets(..., keep_keys=True))")
        out = torch.empty((self.X, self.num_steps + 1), dtype=torch.int32, device=self.ids.device)
        check(lib().subgacc_unpack_lp(ptr(self.keys), self.X, None, self.num_walks, self.num_steps, None, ptr(out), None,
                                      0, stream_ptr()))
        return out

    def feature_table(self):
        """float32 [c+1, m+1] = [0-row ; enc / M]: Z_SF as main.py:174 + random_walks.py:81 build it.
        While lazy the table has one row per ukeys slot (rows past the real count are zero and never indexed)."""
        self.number()
        rows = self.ukeys.numel()
        cdev, _ = self._count_dev()
        out = torch.empty((rows + 1, self.num_steps + 1), dtype=torch.float32, device=self.ids.device)
        check(lib().subgacc_unpack_lp(ptr(self.ukeys), rows, ptr(cdev), self.num_walks, self.num_steps, None, None,
                                      ptr(out), 1, stream_ptr()))
        return out


    def feature_table_by_slot(self):
        """float32 [capacity+1, m+1]: row s+1 = enc/M of the LP row held in slot s of the table of distinct rows, row 0
        (and the rows of free slots, never indexed) zero -- Z_SF without the numbering, for joins over table slots."""
        if self.keyrows:
            raise ValueError("key rows are joined by subgacc_sjoin_fill_keyrows: there is no table to index")
        if self.table is None:
            raise ValueError("no table of distinct LP rows (sample_sets(..., dedup=True))")
        keys = self.table[: self.capacity * 8].view(torch.int64)
        out = torch.empty((self.capacity + 1, self.num_steps + 1), dtype=torch.float32, device=self.ids.device)
        check(lib().subgacc_unpack_lp(ptr(keys), self.capacity, None, self.num_walks, self.num_steps, None, None, ptr(out), 1,
                                      stream_ptr()))
        return out


def make_cfg(csr, num_walks, num_steps, bucket=-1, seed=111413, rng="rand_r", first_hop_wo=True,
             order=_lib.ORDER_WALK_MAJOR, cap_root_degree=True, emit_walks=False, records=True, row_pitch=0):
    """records=True: hand the graph's hop records (DeviceCSR.hop_records(), built on first use) to the kernel -- only the
    fused-row kernel reads them, so the callers that launch something else pass False and nothing is built for them"""
    rng_mode = {"rand_r": _lib.RNG_RAND_R, "philox": _lib.RNG_PHILOX}[rng]
    if num_walks <= 0 or num_steps <= 0:
        raise TypeError("Input parsing error. (num_walks and num_steps must be positive)")
    recs = csr.hop_records() if (records and first_hop_wo and not emit_walks and order == _lib.ORDER_WALK_MAJOR) else None
    return WalkCfg(int(num_walks), int(num_steps), int(bucket), rng_mode, int(seed) & 0xFFFFFFFF,
                   1 if first_hop_wo else 0, int(order), 1 if cap_root_degree else 0,
                   1 if csr.indptr64 else 0, 1 if emit_walks else 0,
                   recs[0].data_ptr() if recs else None, recs[1] if recs else 0, recs[2] if recs else 0, None, int(row_pitch))


def _as_query(query, device):
    """`query` is force-cast to int32 like the reference does (NPY_ARRAY_FORCECAST, subg_acc.c:673)."""
    if torch.is_tensor(query):
        return query.to(device=device, dtype=torch.int32).contiguous().view(-1)
    return torch.from_numpy(np.ascontiguousarray(np.asarray(query).astype(np.int32)).ravel()).to(device)


def unpack_status(words):
    """host list of the packed device status [flags0|flags1<<32, flags2|flags3<<32, distinct rows, members, ...] ->
    [flags0, flags1, flags2, flags3, distinct rows, members, ...].  The four int32 flag words the kernels write share
    one int64[.] buffer with the counts, so that a step zeroes, fills and reads back ONE small tensor."""
    w = [int(v) for v in words]
    return [w[0] & 0xFFFFFFFF, (w[0] >> 32) & 0xFFFFFFFF, w[1] & 0xFFFFFFFF, (w[1] >> 32) & 0xFFFFFFFF] + w[2:]


class RandRDeadEnd(_lib.SubgAccError):
    """rng='rand_r': a walk stood on a node without out-edges, so the stream positions computed from the degrees do not hold.
    sample_sets catches it and samples again with replayed positions (subgacc_rng_replay); a lazy batch, which reports it only
    at resolve(), has to be sampled again by its caller (lazy=False)."""


def check_walk_flags(sets, fl):
    if fl[3] & 32:
        raise _lib.SubgAccError("internal: the candidate list of the batched registration outgrew its launch")
    if fl[3] & 16:
        raise IndexError("query node ids outside [0, num_nodes) (such a root is never looked up on the device; the "
                         "reference reads out of bounds for it) -- or, for a step whose join shares these status words "
                         "(StepBuffers / CapturedStep), a join row number outside the store")
    if fl[0]:
        raise RandRDeadEnd(
            "rng='rand_r': a walk reached a node without out-edges, so the number of draws is data dependent (the "
            "reference's graphs are symmetrised, dataloader.py:122-135) and the stream has to be replayed.  sample_sets / "
            "sample_spg / sample_and_gather WITHOUT buffers= and with lazy=False do so by themselves; a lazy batch or "
            "StepBuffers(rng='rand_r') cannot (nothing is read back before the join): use those forms, or rng='philox'.")
    sets.n_overflow = fl[1]
    if fl[1] and _lib.VERBOSE:
        print(f"#SubGAcc: {fl[1]} keys exceed the buffer, try a larger bucket size > {sets.stride}.")


def _rng_positions(L, cfg, csr, q, n, rng_streams, calls_before, st):
    if cfg.rng_mode != _lib.RNG_RAND_R or n == 0:
        return None, None
    dev = csr.device
    rng_pos = torch.empty(n, dtype=torch.int32, device=dev)
    rng_seed = torch.empty(n, dtype=torch.int32, device=dev)
    ws = torch.empty(L.subgacc_rng_positions_workspace_bytes(n), dtype=torch.uint8, device=dev)
    check(L.subgacc_rng_positions(cfg, ptr(csr.indptr), csr.num_nodes, ptr(q), n, int(rng_streams), int(calls_before), ptr(rng_pos),
                                  ptr(rng_seed), ptr(ws), ws.numel(), st))
    return rng_pos, rng_seed


def walk_kernel_name(csr, num_walks, hops, fused_rows, bucket=-1):
    """which kernel a sample_sets(...) launch of this shape runs (set_sampler form; csrc/walk.hip:launch_walk decides):
    bench.py labels its roofline block with it, and the callers that hand the kernel a work list ask it first (only
    walk_rows_kernel reads one)"""
    if fused_rows:
        return "walk_rows_kernel" if rows_kernel_takes(num_walks, hops, bucket) else "walk_sets_kernel<SPG>"
    return "walk_pipe_kernel" if (num_walks <= 256 and 1 <= hops <= 6) else "walk_sets_kernel"


def _cat(parts, dtype, dev):
    parts = [p_ for p_ in parts if p_ is not None]
    if not parts:
        return torch.empty(0, dtype=dtype, device=dev)
    return parts[0] if len(parts) == 1 else torch.cat(parts)


REPLAY_WARN_WALKS = 1 << 26      # rand_r replay (csrc/replay.hip) is sequential per stream: warn from this many walks on


def _replays_dead_ends(fn):
    """rand_r on a graph with dead ends: the first batch that meets one raises RandRDeadEnd and is sampled again with the replayed
    stream; the discovery is remembered on the DeviceCSR (`_rand_r_dead_ends`), so later calls replay straight away instead of
    walking every batch twice (subg_matrix over all N of a directed graph used to)."""
    @functools.wraps(fn)
    def wrapper(*a, **kw):
        csr = a[0] if a else kw.get("csr")
        rng = kw.get("rng", a[6] if len(a) > 6 else "rand_r")
        if kw.get("walk_replay"):
            return fn(*a, **kw)
        if rng == "rand_r" and getattr(csr, "_rand_r_dead_ends", False):
            return fn(*a, **dict(kw, walk_replay=True))
        try:
            return fn(*a, **kw)
        except RandRDeadEnd:
            try:
                csr._rand_r_dead_ends = True
            except AttributeError:
                pass
            return fn(*a, **dict(kw, walk_replay=True))
    return wrapper


@_replays_dead_ends
def sample_sets(csr, query, num_walks=100, num_steps=3, bucket=-1, seed=111413, rng="rand_r", first_hop_wo=True,
                order=_lib.ORDER_WALK_MAJOR, cap_root_degree=True, emit_walks=False, rng_streams=1,
                calls_before=0, dedup=True, keep_keys=None, staging_bytes=None, uniq_capacity=UNIQ_CAPACITY,
                uniq_small_limit=0, fused_rows=False, lazy=False, strided=False, number_rows=True, key_rows=True,
                walk_replay=False, batched_registration=True, sort_roots=True, rng_state=None):
    """Run the sampler for `query` (roots) on the GPU.  See SampledSets.

    rng_state=(rng_pos, rng_seed): the rand_r stream positions of the roots, already computed (a batch that is sampled a second
    time -- SampledSets.table_form() -- enters the stream where its first sampling did, whatever rng_streams / calls_before were).

    rng="rand_r" on a graph with dead ends (directed graphs: a reached node without out-edges draws nothing in the reference,
    subg_acc.c:804-808): the first attempt notices (RandRDeadEnd) and the batch is sampled again with walk_replay=True -- the
    stream is replayed by subgacc_rng_replay for the position of every walk, then the general walk kernel samples the sets.

    dedup=True numbers the distinct LP rows (ukeys, slot / get_sf()); the packed keys are then only kept when
    keep_keys is set.  fused_rows=True uses subgacc_walk_spg: `ids` come out sorted by node id per root and `data`
    holds SFptr+1 -- finished SpG rows (needs M*m+1 <= 818 and set_sampler order; returns None when the
    configuration does not fit, the caller then takes the general pipeline).  lazy=True: see the module docstring.
    strided=True (one chunk): the finished rows stay in the walk kernel's staging layout -- row i at
    ids / slot [i*stride, +nsize[i]), ids ascending, slot = table slot -- for a join straight from there
    (spg.StridedSpG); no packed copy, no row offsets.  With fused_rows the walk kernel emits them itself
    (subgacc_walk_spg); without, the general walk kernel is followed by subgacc_finish_rows (the faster pair for short
    walks over a cache-resident graph, spg.prefers_fused).  number_rows=False (strided only): the distinct LP rows are not numbered either until somebody
    asks (SampledSets.number()); the join by table slot needs no numbering."""
    p = types.SimpleNamespace(**locals())
    if not _plan_sets(p):              # shapes, kernel form, chunking, the rand_r stream positions, the table of distinct rows
        return None
    res = _launch_sets(p)              # per chunk: walk -> sizes -> (finish rows | register | number) -> packed copy
    return _finish_sets(p) if res is _PACKED else res       # packed forms: global row offsets, numbering of the distinct LP rows


_PACKED = object()      # _launch_sets: the chunks were packed, _finish_sets takes over (the strided forms return their sets themselves)


def _plan_sets(p):
    """sample_sets, part 1: what will be launched.  Fills `p` (the call's arguments as attributes) with the kernel configuration,
    the form of the rows (key rows / table slots / packed keys), the chunking of the roots, the rand_r stream positions and the
    table of distinct LP rows; False when the requested form (fused / strided rows) does not apply to this shape."""
    p.L = lib()
    p.dev = p.csr.device
    p.q = _as_query(p.query, p.dev)
    p.n = p.q.numel()
    p.walk_replay = bool(p.walk_replay and p.rng == "rand_r")
    p.records = bool(p.fused_rows) and p.bucket <= 0 and p.num_walks * p.num_steps + 1 <= FUSED_MAX_Q and 2 <= p.num_steps <= 4 and not p.walk_replay
    p.cfg = make_cfg(p.csr, p.num_walks, p.num_steps, p.bucket, p.seed, p.rng, p.first_hop_wo, p.order, p.cap_root_degree, p.emit_walks, p.records)
    check(p.L.subgacc_key_shift(p.cfg.num_walks, p.cfg.num_steps))   # AssertionError like subg_acc.c:911-915
    p.M, p.m = p.cfg.num_walks, p.cfg.num_steps
    p.stride = p.bucket if p.bucket > 0 else p.M * p.m + 1
    if p.fused_rows and (p.M * p.m + 1 > FUSED_MAX_Q or not p.dedup or p.emit_walks or not p.first_hop_wo
                       or p.order != _lib.ORDER_WALK_MAJOR):
        return False
    p.st = stream_ptr()
    p.status = torch.zeros(4, dtype=torch.int64, device=p.dev)     # [flags x4 (int32) | distinct rows | members]: one memset,
    p.flags = p.status.view(torch.int32)[:4]                        # one read-back (unpack_status)
    if p.keep_keys is None:
        p.keep_keys = not p.dedup
    if p.fused_rows:
        p.keep_keys = False
    p.limit = p.uniq_small_limit if p.uniq_small_limit > 0 else RANK_LIMIT
    p.key_rows_arg = p.key_rows          # (what the caller asked for: a retry with a larger table asks for the same)
    p.kform = key_rows_form(p.M, p.m) if (p.key_rows and p.strided and p.fused_rows and not p.number_rows and p.bucket <= 0 and not p.walk_replay) else 0
    p.per_member = (12 if p.kform == 64 else 8) if p.fused_rows else 12
    if p.staging_bytes is None:     # 288 GB of HBM: one chunk of roots wherever a third of the free memory holds its staging rows
        p.staging_bytes = max(STAGING_BYTES, int(0.35 * torch.cuda.mem_get_info(p.dev)[0])) if p.n * p.stride * p.per_member > STAGING_BYTES \
            else STAGING_BYTES
    p.chunk = max(1, min(p.n, int(p.staging_bytes // (p.stride * p.per_member)), (1 << 31) - 16)) if p.n else 0
    p.lazy = bool(p.lazy and p.dedup and p.n > 0 and p.chunk == p.n)
    # strided rows come from the fused-row walk kernel, or (finish=True) from the general walk kernel + finish_rows
    p.finish = bool(p.strided and not p.fused_rows and p.dedup and not p.emit_walks and p.order == _lib.ORDER_WALK_MAJOR
                  and p.stride <= FINISH_MAX_STRIDE)
    if p.strided and not ((p.fused_rows or p.finish) and p.n > 0 and p.chunk == p.n):
        return False

    p.walk_pos = None
    if p.walk_replay and p.n and p.n * p.M >= REPLAY_WARN_WALKS:
        # csrc/replay.hip: ONE wavefront per rand_r stream replays its walks 64 at a time, m dependent loads per round (the stream
        # is sequential by definition, and set_sampler has a single stream): ~0.1 us per walk, i.e. tens of seconds from 10^8
        # walks on -- the price of bit-exactness with the reference on a graph it was not written for.  rng="philox" has no such cost.
        import warnings
        warnings.warn(f"rng='rand_r' on a graph with dead ends: replaying the sequential stream of {p.n * p.M:,} walks on one "
                      f"wavefront per stream (~{p.n * p.M * 1e-7:.0f} s); rng='philox' samples the same distribution in parallel",
                      RuntimeWarning, stacklevel=4)      # (the caller of sample_sets: past _plan_sets, sample_sets and its wrapper)
    if p.walk_replay and p.n:      # the stream replayed: the position of every root and of every walk (subgacc_rng_replay)
        p.rng_pos = torch.empty(p.n, dtype=torch.int32, device=p.dev)
        p.rng_seed = torch.empty(p.n, dtype=torch.int32, device=p.dev)
        p.walk_pos = torch.empty(p.n * p.M, dtype=torch.int32, device=p.dev)
        check(p.L.subgacc_rng_replay(p.cfg, ptr(p.csr.indptr), ptr(p.csr.indices), p.csr.num_nodes, ptr(p.q), p.n, int(p.rng_streams), int(p.calls_before),
                                   ptr(p.rng_pos), ptr(p.rng_seed), ptr(p.walk_pos), p.st))
    elif p.rng_state is not None and p.rng == "rand_r":
        p.rng_pos, p.rng_seed = p.rng_state
    else:
        p.rng_pos, p.rng_seed = _rng_positions(p.L, p.cfg, p.csr, p.q, p.n, p.rng_streams, p.calls_before, p.st)
    # key rows: strided fused rows that nobody asked to number carry LP keys instead of table slots (module header)
    p.key_rows = bool(p.kform and p.n > 0 and p.chunk == p.n)
    p.kform = p.kform if p.key_rows else 0
    # a store that is kept: key rows as well, registered by one pass over the rows (csrc/keyrows.hip, module header)
    p.batched = bool(p.batched_registration and p.fused_rows and p.dedup and not p.strided and p.bucket <= 0 and p.n > 0 and key_rows_ok(p.M, p.m)
                   and not p.walk_replay)
    p.table = None
    if p.dedup and not p.key_rows:
        p.table = torch.empty(p.L.subgacc_uniq_table_bytes(p.uniq_capacity), dtype=torch.uint8, device=p.dev)
        check(p.L.subgacc_uniq_reset(ptr(p.table), p.uniq_capacity, p.st))

    return True


def _launch_sets(p):
    """sample_sets, part 2: the launches, chunk by chunk -- work list (one chunk), walk kernel, row offsets, then per form: rows
    finished in place, keys registered, distinct rows numbered, packed copy.  Strided forms end here (returns the sets, None when
    the direct ranking cannot number them, or the retry with a larger table); packed forms return _PACKED."""
    p.nsize = torch.empty(p.n, dtype=torch.int32, device=p.dev)
    p.walks = torch.empty((p.n, p.M * (p.m + 1)), dtype=torch.int32, device=p.dev) if p.emit_walks else None
    p.ids_parts, p.key_parts, p.slot_parts = [], [], []
    p.off_chunk = None
    if p.n:
        p.st_ids = torch.empty(p.chunk * p.stride, dtype=torch.int32, device=p.dev)
        p.st_aux = torch.empty(p.chunk * p.stride, dtype=torch.int32 if (p.fused_rows and p.kform != 64) else torch.int64, device=p.dev)
        p.scan_ws = torch.empty(p.L.subgacc_scan_workspace_bytes(p.chunk), dtype=torch.uint8, device=p.dev)
        p.off_chunk = torch.empty(p.chunk + 1, dtype=torch.int64, device=p.dev)
    p.X = 0
    for p.lo in range(0, p.n, p.chunk if p.chunk else 1):
        p.cn = min(p.chunk, p.n - p.lo)
        if p.walk_pos is not None:
            p.cfg.walk_pos = p.walk_pos[p.lo * p.M:].data_ptr()
        # a BATCH sampled in one chunk walks its rows in ascending order of root id, like the buffered step (csrc/worklist.hip:
        # repeated and neighbouring roots share their lines in L2; the rows stay where they are).  Beyond a million roots the call
        # is the offline stage over a whole graph, whose nodes come in order already (listing them again cost it 1.5 %).
        p.by_root = (p.fused_rows and p.chunk == p.n and SORT_ROOTS_MIN <= p.cn <= SORT_ROOTS_MAX and p.sort_roots and p.walk_pos is None and
                   rows_kernel_takes(p.M, p.m, p.bucket))
        if p.by_root:
            p.wl = torch.empty(p.cn, dtype=torch.int32, device=p.dev)
            p.nwl = torch.zeros(1, dtype=torch.int64, device=p.dev)
            p.wws = torch.zeros(p.L.subgacc_worklist_workspace_bytes(p.cn), dtype=torch.uint8, device=p.dev)
            p.nsize.zero_()           # (a row that is not listed -- a root equal to SUBGACC_NO_ROOT -- reads as an empty set)
            check(p.L.subgacc_worklist_by_root(ptr(p.q), p.cn, p.csr.num_nodes, ptr(p.wl), ptr(p.nwl), ptr(p.wws), p.wws.numel(), p.st))
        with _timed("walk_sets"):
            if p.kform == 64:       # rows of 64-bit LP keys (4 hops, M >= 128): one chunk, optionally in work-list order
                check(p.L.subgacc_walk_keyrows64(p.cfg, ptr(p.csr.indptr), ptr(p.csr.indices), p.csr.num_nodes, ptr(p.q), p.cn,
                                               ptr(p.rng_pos) if p.rng_pos is not None else None,
                                               ptr(p.rng_seed) if p.rng_seed is not None else None,
                                               ptr(p.wl) if p.by_root else None, ptr(p.nwl) if p.by_root else None,
                                               ptr(p.st_ids), ptr(p.st_aux), ptr(p.nsize), ptr(p.flags), p.st))
            elif p.by_root:
                check(p.L.subgacc_walk_spg_list(p.cfg, ptr(p.csr.indptr), ptr(p.csr.indices), p.csr.num_nodes, ptr(p.q), p.cn,
                                              ptr(p.rng_pos) if p.rng_pos is not None else None,
                                              ptr(p.rng_seed) if p.rng_seed is not None else None, ptr(p.wl), ptr(p.nwl),
                                              None if p.batched else ptr(p.table), 0 if (p.key_rows or p.batched) else p.uniq_capacity,
                                              ptr(p.st_ids), ptr(p.st_aux), ptr(p.nsize), ptr(p.flags), p.st))
            elif p.fused_rows:
                check(p.L.subgacc_walk_spg(p.cfg, ptr(p.csr.indptr), ptr(p.csr.indices), p.csr.num_nodes, ptr(p.q[p.lo:]), p.cn, p.lo,
                                         ptr(p.rng_pos[p.lo:]) if p.rng_pos is not None else None,
                                         ptr(p.rng_seed[p.lo:]) if p.rng_seed is not None else None,
                                         None if p.batched else ptr(p.table), 0 if (p.key_rows or p.batched) else p.uniq_capacity,
                                         ptr(p.st_ids), ptr(p.st_aux), ptr(p.nsize[p.lo:]), ptr(p.flags), p.st))
            else:
                check(p.L.subgacc_walk_sets(p.cfg, ptr(p.csr.indptr), ptr(p.csr.indices), p.csr.num_nodes, ptr(p.q[p.lo:]), p.cn,
                                          ptr(p.rng_pos[p.lo:]) if p.rng_pos is not None else None,
                                          ptr(p.rng_seed[p.lo:]) if p.rng_seed is not None else None,
                                          ptr(p.st_ids), ptr(p.st_aux), ptr(p.nsize[p.lo:]),
                                          ptr(p.walks[p.lo:]) if p.walks is not None else None, ptr(p.flags), p.st))
        if not p.strided:
            check(p.L.subgacc_exclusive_scan_i32(ptr(p.nsize[p.lo:]), p.cn, ptr(p.off_chunk), ptr(p.scan_ws), p.scan_ws.numel(), p.st))
        if p.finish:            # (ids in first-visit order, keys) -> (ids sorted, table slots), in place: finished rows
            p.st_slot = torch.empty(p.chunk * p.stride, dtype=torch.int32, device=p.dev)
            with _timed("spg_build"):
                check(p.L.subgacc_finish_rows(ptr(p.st_ids), ptr(p.st_aux), ptr(p.nsize), p.cn, p.stride, 0, ptr(p.table), p.uniq_capacity,
                                            ptr(p.st_slot), ptr(p.flags), p.st))
            p.st_aux = p.st_slot
        p.numbered_early = (p.fused_rows or p.finish) and p.chunk == p.n and (p.number_rows or not p.strided)
        p.count = p.status[2:3]
        p.total = None
        if p.batched:
            p.ccap = p.L.subgacc_keyrows_cand_capacity(p.cn)
            p.cand = torch.empty(p.ccap, dtype=torch.int32, device=p.dev)
            p.ncand = torch.zeros(1, dtype=torch.int64, device=p.dev)
            p.rp, p.rs = (ptr(p.rng_pos[p.lo:]), ptr(p.rng_seed[p.lo:])) if p.rng_pos is not None else (None, None)
            if p.numbered_early:      # one chunk: register -> exact tags for the candidates -> number (below) -> copy with SFptr+1
                with _timed("register_rows"):
                    check(p.L.subgacc_keyrows_register(ptr(p.st_aux), ptr(p.nsize[p.lo:]), p.cn, p.stride, p.lo, ptr(p.table), p.uniq_capacity,
                                                     ptr(p.cand), p.ccap, ptr(p.ncand), ptr(p.flags), p.st))
                    if p.lazy:
                        p.total, p.wcap = p.cn * p.stride, 0
                    else:           # the one host read of the chunk carries the candidate count along
                        p.total, p.wcap = (int(p.v) for p.v in torch.cat([p.off_chunk[p.cn:p.cn + 1], p.ncand]).tolist())
                        p.wcap = max(min(p.wcap, p.cn), 1)
                    check(p.L.subgacc_walk_tags(p.cfg, ptr(p.csr.indptr), ptr(p.csr.indices), p.csr.num_nodes, ptr(p.q[p.lo:]), p.cn, p.lo, p.rp, p.rs,
                                              ptr(p.cand), ptr(p.ncand), p.wcap, ptr(p.table), p.uniq_capacity, ptr(p.flags), p.st))
        if p.strided and not p.numbered_early:
            p.ukeys, p.max_unique = None, p.uniq_capacity
        if p.numbered_early:    # one chunk: the table is complete -> number it now and let the copy emit SFptr+1
            p.max_unique = min(p.uniq_capacity, p.limit)
            p.ukeys = torch.empty(p.max_unique, dtype=torch.int64, device=p.dev)
            p.nws = torch.empty(p.L.subgacc_uniq_number_workspace_bytes(p.uniq_capacity, 0), dtype=torch.uint8, device=p.dev)
            with _timed("uniq_rows"):
                check(p.L.subgacc_uniq_number(ptr(p.table), p.uniq_capacity, None, 0, ptr(p.ukeys), p.max_unique, ptr(p.count), p.limit,
                                            ptr(p.nws), p.nws.numel(), p.st))
        if p.strided:           # rows are joined from the staging arrays themselves
            p.sets = SampledSets(p.nsize, None, p.st_ids, None, None, p.ukeys, p.M, p.m, p.stride, None)
            p.sets.slot, p.sets.table, p.sets.capacity, p.sets.strided = p.st_aux, p.table, (0 if p.key_rows else p.uniq_capacity), True
            if p.key_rows:
                p.sets.keyrows, p.sets.key64 = True, p.kform == 64
                p.sets._keyctx = {"csr": p.csr, "roots": p.q, "cfg": p.cfg, "rng_pos": p.rng_pos, "rng_seed": p.rng_seed,
                                "capacity": p.uniq_capacity, "fresh": lambda: True}
            torch.sum(p.nsize, dim=(0,), dtype=torch.int64, out=p.status[3])
            p.sets.status = p.status
            if not p.lazy:      # eager: same recovery as the packed forms below
                p.st_host = unpack_status(p.status.tolist())
                if p.st_host[2]:                    # the table of distinct LP rows overflowed: walk again with a larger one
                    return sample_sets(p.csr, p.q, p.num_walks, p.num_steps, p.bucket, p.seed, p.rng, p.first_hop_wo, p.order,
                                       p.cap_root_degree, p.emit_walks, p.rng_streams, p.calls_before, p.dedup, p.keep_keys,
                                       p.staging_bytes, p.uniq_capacity * 4, p.uniq_small_limit, p.fused_rows, p.lazy, p.strided,
                                       p.number_rows, key_rows=p.key_rows_arg, walk_replay=p.walk_replay,
                                       batched_registration=p.batched_registration, sort_roots=p.sort_roots, rng_state=p.rng_state)
                if p.st_host[4] > p.max_unique:       # more distinct rows than the direct ranking numbers: the caller
                    return None                   # (sample_spg) falls through to the packed forms
                p.sets.resolve()
            return p.sets
        # packed arrays: exact size (one 8-byte host read) or, lazily, the upper bound n*stride
        if p.total is None:
            p.total = p.cn * p.stride if p.lazy else int(p.off_chunk[p.cn].item())
        p.ids_c = torch.empty(p.total, dtype=torch.int32, device=p.dev)
        p.keys_c = torch.empty(p.total, dtype=torch.int64, device=p.dev) if p.keep_keys else None
        p.slot_c = torch.empty(p.total, dtype=torch.int32, device=p.dev) if p.dedup else None
        with _timed("compact_sets"):
            if p.batched:     # key rows -> packed rows: SFptr+1 looked up on the way, or (several chunks) the key kept as payload
                check(p.L.subgacc_keyrows_compact(ptr(p.st_ids), ptr(p.st_aux), ptr(p.nsize[p.lo:]), ptr(p.off_chunk), p.cn, p.stride, p.lo, ptr(p.table),
                                                p.uniq_capacity, ptr(p.ukeys) if p.numbered_early else None,
                                                ptr(p.count) if p.numbered_early else None, p.max_unique if p.numbered_early else 0,
                                                ptr(p.ids_c), ptr(p.slot_c), ptr(p.cand), p.ccap, ptr(p.ncand), ptr(p.flags), p.st))
                if not p.numbered_early:    # the pass registered this chunk's keys itself: now the candidates' exact tags
                    check(p.L.subgacc_walk_tags(p.cfg, ptr(p.csr.indptr), ptr(p.csr.indices), p.csr.num_nodes, ptr(p.q[p.lo:]), p.cn, p.lo, p.rp, p.rs,
                                              ptr(p.cand), ptr(p.ncand), 0, ptr(p.table), p.uniq_capacity, ptr(p.flags), p.st))
            elif p.fused_rows:
                check(p.L.subgacc_compact_rows(ptr(p.st_ids), ptr(p.st_aux), ptr(p.nsize[p.lo:]), ptr(p.off_chunk), p.cn, p.stride,
                                             ptr(p.ids_c), ptr(p.slot_c), ptr(p.table) if p.numbered_early else None,
                                             p.uniq_capacity if p.numbered_early else 0, p.st))
            else:
                check(p.L.subgacc_compact_sets(ptr(p.st_ids), ptr(p.st_aux), ptr(p.nsize[p.lo:]), ptr(p.off_chunk), p.cn, p.stride,
                                             ptr(p.ids_c), ptr(p.keys_c), ptr(p.table), p.uniq_capacity if p.dedup else 0, p.X,
                                             ptr(p.slot_c), ptr(p.flags), p.st))
        p.ids_parts.append(p.ids_c)
        p.key_parts.append(p.keys_c)
        p.slot_parts.append(p.slot_c)
        p.X += p.total
    return _PACKED


def _finish_sets(p):
    """sample_sets, part 3 (packed forms): the chunks joined, global row offsets, the distinct LP rows numbered by first occurrence
    (subg_acc.c:957-1000), payloads translated to SFptr(+1), status read (eager) -- with the retry when the table overflowed."""
    p.ids = _cat(p.ids_parts, torch.int32, p.dev)
    p.keys = _cat(p.key_parts, torch.int64, p.dev) if p.keep_keys else None
    p.slot = _cat(p.slot_parts, torch.int32, p.dev) if p.dedup else None
    del p.ids_parts, p.key_parts, p.slot_parts

    if p.lazy:
        p.row_off = p.off_chunk               # a single chunk: its offsets are the global ones
    else:
        p.row_off = torch.empty(p.n + 1, dtype=torch.int64, device=p.dev)
        p.ws = torch.empty(p.L.subgacc_scan_workspace_bytes(p.n), dtype=torch.uint8, device=p.dev)
        check(p.L.subgacc_exclusive_scan_i32(ptr(p.nsize), p.n, ptr(p.row_off), ptr(p.ws), p.ws.numel(), p.st))

    p.sets = SampledSets(p.nsize, p.row_off, p.ids, p.keys, None, None, p.M, p.m, p.stride, p.walks)
    if not p.dedup:
        check_walk_flags(p.sets, p.flags.tolist())
        return p.sets

    # number the distinct LP rows by first occurrence (subg_acc.c:957-1000)
    p.x_dev = p.row_off[p.n:p.n + 1]
    if p.n and p.fused_rows and p.chunk == p.n:
        pass                          # numbered before the copy, which already wrote SFptr+1
    elif p.fused_rows or p.lazy:          # table-only direct ranking (tags need not be element positions)
        p.count = p.status[2:3]
        p.max_unique = min(p.uniq_capacity, p.limit)
        p.ukeys = torch.empty(p.max_unique, dtype=torch.int64, device=p.dev)
        p.ws = torch.empty(p.L.subgacc_uniq_number_workspace_bytes(p.uniq_capacity, 0), dtype=torch.uint8, device=p.dev)
        with _timed("uniq_rows"):
            check(p.L.subgacc_uniq_number(ptr(p.table), p.uniq_capacity, None, 0, ptr(p.ukeys), p.max_unique, ptr(p.count), p.limit,
                                        ptr(p.ws), p.ws.numel(), p.st))
            if p.batched:               # LP key -> SFptr+1 in place (the chunks kept the keys as payload)
                check(p.L.subgacc_keyrows_translate(ptr(p.slot), p.slot.numel(), ptr(p.x_dev), ptr(p.table), p.uniq_capacity, ptr(p.ukeys),
                                                  ptr(p.count), p.max_unique, p.st))
            elif p.fused_rows:          # slot -> SFptr+1 in place: the rows are finished SpG rows
                check(p.L.subgacc_uniq_translate(ptr(p.table), p.uniq_capacity, ptr(p.slot), p.slot.numel(), ptr(p.x_dev), 1, p.st))
    else:
        p.count = p.status[2:3]
        p.max_unique = min(max(p.X, 1), p.uniq_capacity)
        p.ukeys = torch.empty(p.max_unique, dtype=torch.int64, device=p.dev)
        p.ws = torch.empty(p.L.subgacc_uniq_number_workspace_bytes(p.uniq_capacity, p.X), dtype=torch.uint8, device=p.dev)
        with _timed("uniq_rows"):
            check(p.L.subgacc_uniq_number(ptr(p.table), p.uniq_capacity, ptr(p.slot), p.X, ptr(p.ukeys), p.max_unique, ptr(p.count),
                                        p.uniq_small_limit, ptr(p.ws), p.ws.numel(), p.st))
    p.sets.ukeys, p.sets.table, p.sets.capacity = p.ukeys, p.table, p.uniq_capacity
    if p.fused_rows:
        p.sets.data = p.slot
    else:
        p.sets.slot = p.slot
    p.status[3:4].copy_(p.x_dev)
    p.sets.status = p.status
    if p.lazy:
        return p.sets
    # eager: read the status now; grow the table and walk again if it overflowed
    p.st_host = unpack_status(p.status.tolist())
    if p.fused_rows and not p.st_host[2] and p.st_host[4] > p.max_unique:
        return None           # more distinct rows than the direct ranking handles: the caller takes the general path
    if p.st_host[2]:
        return sample_sets(p.csr, p.q, p.num_walks, p.num_steps, p.bucket, p.seed, p.rng, p.first_hop_wo, p.order, p.cap_root_degree,
                           p.emit_walks, p.rng_streams, p.calls_before, p.dedup, p.keep_keys, p.staging_bytes, p.uniq_capacity * 4,
                           p.uniq_small_limit, p.fused_rows, p.lazy, p.strided, p.number_rows, key_rows=p.key_rows_arg, walk_replay=p.walk_replay,
                           batched_registration=p.batched_registration, sort_roots=p.sort_roots, rng_state=p.rng_state)
    p.sets.resolve()
    p.sets.ukeys = p.sets.ukeys.clone()
    return p.sets


def dedup_lp_rows(sets, capacity=UNIQ_CAPACITY, small_limit=0):
    """Global first-occurrence dedup of already packed LP keys (subg_acc.c:957-1000) -> sets.sf, sets.ukeys.
    The stand-alone form (subgacc_uniq_insert over sets.keys) for sets sampled with dedup=False; sample_sets itself
    fuses the insert into the compaction pass."""
    L = lib()
    dev = sets.ids.device
    st = stream_ptr()
    X = sets.X
    sf = torch.empty(X, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    while True:
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        table = torch.empty(L.subgacc_uniq_table_bytes(capacity), dtype=torch.uint8, device=dev)
        ws = torch.empty(L.subgacc_uniq_number_workspace_bytes(capacity, X), dtype=torch.uint8, device=dev)
        max_unique = min(max(X, 1), capacity)
        ukeys = torch.empty(max_unique, dtype=torch.int64, device=dev)
        check(L.subgacc_uniq_reset(ptr(table), capacity, st))
        check(L.subgacc_uniq_insert(ptr(table), capacity, ptr(sets.keys), X, 0, ptr(sf), ptr(flags), st))
        # probe chains are bounded in the kernels, so numbering an over-full table is harmless (and discarded)
        check(L.subgacc_uniq_number(ptr(table), capacity, ptr(sf), X, ptr(ukeys), max_unique, ptr(count),
                                    small_limit, ptr(ws), ws.numel(), st))
        check(L.subgacc_uniq_translate(ptr(table), capacity, ptr(sf), X, None, 0, st))
        status = torch.cat([flags.long(), count]).tolist()
        if status[2]:
            capacity *= 4           # table (nearly) full: the distinct-row count exceeded the guess, retry larger
            continue
        c = status[4]
        break
    sets.sf = sf
    sets.ukeys = ukeys[:c].clone()
    return sets
