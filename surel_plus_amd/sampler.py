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
        host.copy_(src, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
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
             order=_lib.ORDER_WALK_MAJOR, cap_root_degree=True, emit_walks=False, records=True):
    """records=True: hand the graph's hop records (DeviceCSR.hop_records(), built on first use) to the kernel -- only the
    fused-row kernel reads them, so the callers that launch something else pass False and nothing is built for them"""
    rng_mode = {"rand_r": _lib.RNG_RAND_R, "philox": _lib.RNG_PHILOX}[rng]
    if num_walks <= 0 or num_steps <= 0:
        raise TypeError("Input parsing error. (num_walks and num_steps must be positive)")
    recs = csr.hop_records() if (records and first_hop_wo and not emit_walks and order == _lib.ORDER_WALK_MAJOR) else None
    return WalkCfg(int(num_walks), int(num_steps), int(bucket), rng_mode, int(seed) & 0xFFFFFFFF,
                   1 if first_hop_wo else 0, int(order), 1 if cap_root_degree else 0,
                   1 if csr.indptr64 else 0, 1 if emit_walks else 0,
                   recs[0].data_ptr() if recs else None, recs[1] if recs else 0, recs[2] if recs else 0)


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
    L = lib()
    dev = csr.device
    q = _as_query(query, dev)
    n = q.numel()
    walk_replay = bool(walk_replay and rng == "rand_r")
    records = bool(fused_rows) and bucket <= 0 and num_walks * num_steps + 1 <= FUSED_MAX_Q and 2 <= num_steps <= 4 and not walk_replay
    cfg = make_cfg(csr, num_walks, num_steps, bucket, seed, rng, first_hop_wo, order, cap_root_degree, emit_walks, records)
    check(L.subgacc_key_shift(cfg.num_walks, cfg.num_steps))   # AssertionError like subg_acc.c:911-915
    M, m = cfg.num_walks, cfg.num_steps
    stride = bucket if bucket > 0 else M * m + 1
    if fused_rows and (M * m + 1 > FUSED_MAX_Q or not dedup or emit_walks or not first_hop_wo
                       or order != _lib.ORDER_WALK_MAJOR):
        return None
    st = stream_ptr()
    status = torch.zeros(4, dtype=torch.int64, device=dev)     # [flags x4 (int32) | distinct rows | members]: one memset,
    flags = status.view(torch.int32)[:4]                        # one read-back (unpack_status)
    if keep_keys is None:
        keep_keys = not dedup
    if fused_rows:
        keep_keys = False
    limit = uniq_small_limit if uniq_small_limit > 0 else RANK_LIMIT
    key_rows_arg = key_rows          # (what the caller asked for: a retry with a larger table asks for the same)
    kform = key_rows_form(M, m) if (key_rows and strided and fused_rows and not number_rows and bucket <= 0 and not walk_replay) else 0
    per_member = (12 if kform == 64 else 8) if fused_rows else 12
    if staging_bytes is None:     # 288 GB of HBM: one chunk of roots wherever a third of the free memory holds its staging rows
        staging_bytes = max(STAGING_BYTES, int(0.35 * torch.cuda.mem_get_info(dev)[0])) if n * stride * per_member > STAGING_BYTES \
            else STAGING_BYTES
    chunk = max(1, min(n, int(staging_bytes // (stride * per_member)), (1 << 31) - 16)) if n else 0
    lazy = bool(lazy and dedup and n > 0 and chunk == n)
    # strided rows come from the fused-row walk kernel, or (finish=True) from the general walk kernel + finish_rows
    finish = bool(strided and not fused_rows and dedup and not emit_walks and order == _lib.ORDER_WALK_MAJOR
                  and stride <= FINISH_MAX_STRIDE)
    if strided and not ((fused_rows or finish) and n > 0 and chunk == n):
        return None

    walk_pos = None
    if walk_replay and n and n * M >= REPLAY_WARN_WALKS:
        # csrc/replay.hip: ONE wavefront per rand_r stream replays its walks 64 at a time, m dependent loads per round (the stream
        # is sequential by definition, and set_sampler has a single stream): ~0.1 us per walk, i.e. tens of seconds from 10^8
        # walks on -- the price of bit-exactness with the reference on a graph it was not written for.  rng="philox" has no such cost.
        import warnings
        warnings.warn(f"rng='rand_r' on a graph with dead ends: replaying the sequential stream of {n * M:,} walks on one "
                      f"wavefront per stream (~{n * M * 1e-7:.0f} s); rng='philox' samples the same distribution in parallel",
                      RuntimeWarning, stacklevel=3)
    if walk_replay and n:      # the stream replayed: the position of every root and of every walk (subgacc_rng_replay)
        rng_pos = torch.empty(n, dtype=torch.int32, device=dev)
        rng_seed = torch.empty(n, dtype=torch.int32, device=dev)
        walk_pos = torch.empty(n * M, dtype=torch.int32, device=dev)
        check(L.subgacc_rng_replay(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(q), n, int(rng_streams), int(calls_before),
                                   ptr(rng_pos), ptr(rng_seed), ptr(walk_pos), st))
    elif rng_state is not None and rng == "rand_r":
        rng_pos, rng_seed = rng_state
    else:
        rng_pos, rng_seed = _rng_positions(L, cfg, csr, q, n, rng_streams, calls_before, st)
    # key rows: strided fused rows that nobody asked to number carry LP keys instead of table slots (module header)
    key_rows = bool(kform and n > 0 and chunk == n)
    kform = kform if key_rows else 0
    # a store that is kept: key rows as well, registered by one pass over the rows (csrc/keyrows.hip, module header)
    batched = bool(batched_registration and fused_rows and dedup and not strided and bucket <= 0 and n > 0 and key_rows_ok(M, m)
                   and not walk_replay)
    table = None
    if dedup and not key_rows:
        table = torch.empty(L.subgacc_uniq_table_bytes(uniq_capacity), dtype=torch.uint8, device=dev)
        check(L.subgacc_uniq_reset(ptr(table), uniq_capacity, st))

    nsize = torch.empty(n, dtype=torch.int32, device=dev)
    walks = torch.empty((n, M * (m + 1)), dtype=torch.int32, device=dev) if emit_walks else None
    ids_parts, key_parts, slot_parts = [], [], []
    off_chunk = None
    if n:
        st_ids = torch.empty(chunk * stride, dtype=torch.int32, device=dev)
        st_aux = torch.empty(chunk * stride, dtype=torch.int32 if (fused_rows and kform != 64) else torch.int64, device=dev)
        scan_ws = torch.empty(L.subgacc_scan_workspace_bytes(chunk), dtype=torch.uint8, device=dev)
        off_chunk = torch.empty(chunk + 1, dtype=torch.int64, device=dev)
    X = 0
    for lo in range(0, n, chunk if chunk else 1):
        cn = min(chunk, n - lo)
        if walk_pos is not None:
            cfg.walk_pos = walk_pos[lo * M:].data_ptr()
        # a BATCH sampled in one chunk walks its rows in ascending order of root id, like the buffered step (csrc/worklist.hip:
        # repeated and neighbouring roots share their lines in L2; the rows stay where they are).  Beyond a million roots the call
        # is the offline stage over a whole graph, whose nodes come in order already (listing them again cost it 1.5 %).
        by_root = (fused_rows and chunk == n and SORT_ROOTS_MIN <= cn <= SORT_ROOTS_MAX and sort_roots and walk_pos is None and
                   rows_kernel_takes(M, m, bucket))
        if by_root:
            wl = torch.empty(cn, dtype=torch.int32, device=dev)
            nwl = torch.zeros(1, dtype=torch.int64, device=dev)
            wws = torch.zeros(L.subgacc_worklist_workspace_bytes(cn), dtype=torch.uint8, device=dev)
            nsize.zero_()           # (a row that is not listed -- a root equal to SUBGACC_NO_ROOT -- reads as an empty set)
            check(L.subgacc_worklist_by_root(ptr(q), cn, csr.num_nodes, ptr(wl), ptr(nwl), ptr(wws), wws.numel(), st))
        with _timed("walk_sets"):
            if kform == 64:       # rows of 64-bit LP keys (4 hops, M >= 128): one chunk, optionally in work-list order
                check(L.subgacc_walk_keyrows64(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(q), cn,
                                               ptr(rng_pos) if rng_pos is not None else None,
                                               ptr(rng_seed) if rng_seed is not None else None,
                                               ptr(wl) if by_root else None, ptr(nwl) if by_root else None,
                                               ptr(st_ids), ptr(st_aux), ptr(nsize), ptr(flags), st))
            elif by_root:
                check(L.subgacc_walk_spg_list(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(q), cn,
                                              ptr(rng_pos) if rng_pos is not None else None,
                                              ptr(rng_seed) if rng_seed is not None else None, ptr(wl), ptr(nwl),
                                              None if batched else ptr(table), 0 if (key_rows or batched) else uniq_capacity,
                                              ptr(st_ids), ptr(st_aux), ptr(nsize), ptr(flags), st))
            elif fused_rows:
                check(L.subgacc_walk_spg(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(q[lo:]), cn, lo,
                                         ptr(rng_pos[lo:]) if rng_pos is not None else None,
                                         ptr(rng_seed[lo:]) if rng_seed is not None else None,
                                         None if batched else ptr(table), 0 if (key_rows or batched) else uniq_capacity,
                                         ptr(st_ids), ptr(st_aux), ptr(nsize[lo:]), ptr(flags), st))
            else:
                check(L.subgacc_walk_sets(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(q[lo:]), cn,
                                          ptr(rng_pos[lo:]) if rng_pos is not None else None,
                                          ptr(rng_seed[lo:]) if rng_seed is not None else None,
                                          ptr(st_ids), ptr(st_aux), ptr(nsize[lo:]),
                                          ptr(walks[lo:]) if walks is not None else None, ptr(flags), st))
        if not strided:
            check(L.subgacc_exclusive_scan_i32(ptr(nsize[lo:]), cn, ptr(off_chunk), ptr(scan_ws), scan_ws.numel(), st))
        if finish:            # (ids in first-visit order, keys) -> (ids sorted, table slots), in place: finished rows
            st_slot = torch.empty(chunk * stride, dtype=torch.int32, device=dev)
            with _timed("spg_build"):
                check(L.subgacc_finish_rows(ptr(st_ids), ptr(st_aux), ptr(nsize), cn, stride, 0, ptr(table), uniq_capacity,
                                            ptr(st_slot), ptr(flags), st))
            st_aux = st_slot
        numbered_early = (fused_rows or finish) and chunk == n and (number_rows or not strided)
        count = status[2:3]
        total = None
        if batched:
            ccap = L.subgacc_keyrows_cand_capacity(cn)
            cand = torch.empty(ccap, dtype=torch.int32, device=dev)
            ncand = torch.zeros(1, dtype=torch.int64, device=dev)
            rp, rs = (ptr(rng_pos[lo:]), ptr(rng_seed[lo:])) if rng_pos is not None else (None, None)
            if numbered_early:      # one chunk: register -> exact tags for the candidates -> number (below) -> copy with SFptr+1
                with _timed("register_rows"):
                    check(L.subgacc_keyrows_register(ptr(st_aux), ptr(nsize[lo:]), cn, stride, lo, ptr(table), uniq_capacity,
                                                     ptr(cand), ccap, ptr(ncand), ptr(flags), st))
                    if lazy:
                        total, wcap = cn * stride, 0
                    else:           # the one host read of the chunk carries the candidate count along
                        total, wcap = (int(v) for v in torch.cat([off_chunk[cn:cn + 1], ncand]).tolist())
                        wcap = max(min(wcap, cn), 1)
                    check(L.subgacc_walk_tags(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(q[lo:]), cn, lo, rp, rs,
                                              ptr(cand), ptr(ncand), wcap, ptr(table), uniq_capacity, ptr(flags), st))
        if strided and not numbered_early:
            ukeys, max_unique = None, uniq_capacity
        if numbered_early:    # one chunk: the table is complete -> number it now and let the copy emit SFptr+1
            max_unique = min(uniq_capacity, limit)
            ukeys = torch.empty(max_unique, dtype=torch.int64, device=dev)
            nws = torch.empty(L.subgacc_uniq_number_workspace_bytes(uniq_capacity, 0), dtype=torch.uint8, device=dev)
            with _timed("uniq_rows"):
                check(L.subgacc_uniq_number(ptr(table), uniq_capacity, None, 0, ptr(ukeys), max_unique, ptr(count), limit,
                                            ptr(nws), nws.numel(), st))
        if strided:           # rows are joined from the staging arrays themselves
            sets = SampledSets(nsize, None, st_ids, None, None, ukeys, M, m, stride, None)
            sets.slot, sets.table, sets.capacity, sets.strided = st_aux, table, (0 if key_rows else uniq_capacity), True
            if key_rows:
                sets.keyrows, sets.key64 = True, kform == 64
                sets._keyctx = {"csr": csr, "roots": q, "cfg": cfg, "rng_pos": rng_pos, "rng_seed": rng_seed,
                                "capacity": uniq_capacity, "fresh": lambda: True}
            torch.sum(nsize, dim=(0,), dtype=torch.int64, out=status[3])
            sets.status = status
            if not lazy:      # eager: same recovery as the packed forms below
                st_host = unpack_status(status.tolist())
                if st_host[2]:                    # the table of distinct LP rows overflowed: walk again with a larger one
                    return sample_sets(csr, q, num_walks, num_steps, bucket, seed, rng, first_hop_wo, order,
                                       cap_root_degree, emit_walks, rng_streams, calls_before, dedup, keep_keys,
                                       staging_bytes, uniq_capacity * 4, uniq_small_limit, fused_rows, lazy, strided,
                                       number_rows, key_rows=key_rows_arg, walk_replay=walk_replay,
                                       batched_registration=batched_registration, sort_roots=sort_roots, rng_state=rng_state)
                if st_host[4] > max_unique:       # more distinct rows than the direct ranking numbers: the caller
                    return None                   # (sample_spg) falls through to the packed forms
                sets.resolve()
            return sets
        # packed arrays: exact size (one 8-byte host read) or, lazily, the upper bound n*stride
        if total is None:
            total = cn * stride if lazy else int(off_chunk[cn].item())
        ids_c = torch.empty(total, dtype=torch.int32, device=dev)
        keys_c = torch.empty(total, dtype=torch.int64, device=dev) if keep_keys else None
        slot_c = torch.empty(total, dtype=torch.int32, device=dev) if dedup else None
        with _timed("compact_sets"):
            if batched:     # key rows -> packed rows: SFptr+1 looked up on the way, or (several chunks) the key kept as payload
                check(L.subgacc_keyrows_compact(ptr(st_ids), ptr(st_aux), ptr(nsize[lo:]), ptr(off_chunk), cn, stride, lo, ptr(table),
                                                uniq_capacity, ptr(ukeys) if numbered_early else None,
                                                ptr(count) if numbered_early else None, max_unique if numbered_early else 0,
                                                ptr(ids_c), ptr(slot_c), ptr(cand), ccap, ptr(ncand), ptr(flags), st))
                if not numbered_early:    # the pass registered this chunk's keys itself: now the candidates' exact tags
                    check(L.subgacc_walk_tags(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(q[lo:]), cn, lo, rp, rs,
                                              ptr(cand), ptr(ncand), 0, ptr(table), uniq_capacity, ptr(flags), st))
            elif fused_rows:
                check(L.subgacc_compact_rows(ptr(st_ids), ptr(st_aux), ptr(nsize[lo:]), ptr(off_chunk), cn, stride,
                                             ptr(ids_c), ptr(slot_c), ptr(table) if numbered_early else None,
                                             uniq_capacity if numbered_early else 0, st))
            else:
                check(L.subgacc_compact_sets(ptr(st_ids), ptr(st_aux), ptr(nsize[lo:]), ptr(off_chunk), cn, stride,
                                             ptr(ids_c), ptr(keys_c), ptr(table), uniq_capacity if dedup else 0, X,
                                             ptr(slot_c), ptr(flags), st))
        ids_parts.append(ids_c)
        key_parts.append(keys_c)
        slot_parts.append(slot_c)
        X += total
    ids = _cat(ids_parts, torch.int32, dev)
    keys = _cat(key_parts, torch.int64, dev) if keep_keys else None
    slot = _cat(slot_parts, torch.int32, dev) if dedup else None
    del ids_parts, key_parts, slot_parts

    if lazy:
        row_off = off_chunk               # a single chunk: its offsets are the global ones
    else:
        row_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        ws = torch.empty(L.subgacc_scan_workspace_bytes(n), dtype=torch.uint8, device=dev)
        check(L.subgacc_exclusive_scan_i32(ptr(nsize), n, ptr(row_off), ptr(ws), ws.numel(), st))

    sets = SampledSets(nsize, row_off, ids, keys, None, None, M, m, stride, walks)
    if not dedup:
        check_walk_flags(sets, flags.tolist())
        return sets

    # number the distinct LP rows by first occurrence (subg_acc.c:957-1000)
    x_dev = row_off[n:n + 1]
    if n and fused_rows and chunk == n:
        pass                          # numbered before the copy, which already wrote SFptr+1
    elif fused_rows or lazy:          # table-only direct ranking (tags need not be element positions)
        count = status[2:3]
        max_unique = min(uniq_capacity, limit)
        ukeys = torch.empty(max_unique, dtype=torch.int64, device=dev)
        ws = torch.empty(L.subgacc_uniq_number_workspace_bytes(uniq_capacity, 0), dtype=torch.uint8, device=dev)
        with _timed("uniq_rows"):
            check(L.subgacc_uniq_number(ptr(table), uniq_capacity, None, 0, ptr(ukeys), max_unique, ptr(count), limit,
                                        ptr(ws), ws.numel(), st))
            if batched:               # LP key -> SFptr+1 in place (the chunks kept the keys as payload)
                check(L.subgacc_keyrows_translate(ptr(slot), slot.numel(), ptr(x_dev), ptr(table), uniq_capacity, ptr(ukeys),
                                                  ptr(count), max_unique, st))
            elif fused_rows:          # slot -> SFptr+1 in place: the rows are finished SpG rows
                check(L.subgacc_uniq_translate(ptr(table), uniq_capacity, ptr(slot), slot.numel(), ptr(x_dev), 1, st))
    else:
        count = status[2:3]
        max_unique = min(max(X, 1), uniq_capacity)
        ukeys = torch.empty(max_unique, dtype=torch.int64, device=dev)
        ws = torch.empty(L.subgacc_uniq_number_workspace_bytes(uniq_capacity, X), dtype=torch.uint8, device=dev)
        with _timed("uniq_rows"):
            check(L.subgacc_uniq_number(ptr(table), uniq_capacity, ptr(slot), X, ptr(ukeys), max_unique, ptr(count),
                                        uniq_small_limit, ptr(ws), ws.numel(), st))
    sets.ukeys, sets.table, sets.capacity = ukeys, table, uniq_capacity
    if fused_rows:
        sets.data = slot
    else:
        sets.slot = slot
    status[3:4].copy_(x_dev)
    sets.status = status
    if lazy:
        return sets
    # eager: read the status now; grow the table and walk again if it overflowed
    st_host = unpack_status(status.tolist())
    if fused_rows and not st_host[2] and st_host[4] > max_unique:
        return None           # more distinct rows than the direct ranking handles: the caller takes the general path
    if st_host[2]:
        return sample_sets(csr, q, num_walks, num_steps, bucket, seed, rng, first_hop_wo, order, cap_root_degree,
                           emit_walks, rng_streams, calls_before, dedup, keep_keys, staging_bytes, uniq_capacity * 4,
                           uniq_small_limit, fused_rows, lazy, strided, number_rows, key_rows=key_rows_arg, walk_replay=walk_replay,
                           batched_registration=batched_registration, sort_roots=sort_roots, rng_state=rng_state)
    sets.resolve()
    sets.ukeys = sets.ukeys.clone()
    return sets


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
