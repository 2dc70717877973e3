"""SpG: the sparse node-set store of SUREL+, resident in HBM.

Reference: sampler/random_walks.py:74-82 (`subg_matrix`) builds a scipy csr_matrix whose row u lists the
sampled set of u (sorted node ids) with data = SFptr+1, plus the LP table `enc` with a zero row in front.
Here the same CSR lives on the GPU (int64 row offsets, int32 ids, int32 or float64 payload) and is what
`sjoin` / `gather` consume; it is built by the segmented sort in csrc/spg.hip.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .sampler import DeviceCSR, SampledSets, _timed, dedup_lp_rows, sample_sets


class SpG:
    """Device CSR of node sets.  `data` is int32 (SFptr+1, LP encoder) or float64 (PPR scores)."""

    def __init__(self, indptr, indices, data, max_len=None, shape=None, max_data=None):
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int32
        assert data.dtype in (torch.int32, torch.float64)
        self.indptr, self.indices, self.data = indptr.contiguous(), indices.contiguous(), data.contiguous()
        self.n_rows = indptr.numel() - 1
        if max_len is None:
            max_len = int((self.indptr[1:] - self.indptr[:-1]).max().item()) if self.n_rows else 0
        self.max_len = int(max_len)              # upper bound on the row length (sizes the LDS staging of sjoin)
        if max_data is None and data.dtype == torch.int32:
            max_data = int(data.max().item()) if data.numel() else 0
        self.max_data = max_data                 # largest SFptr+1: the encode table must have more rows than this
        self.shape = shape or (self.n_rows, self.n_rows)
        self.device = indptr.device

    @property
    def nnz(self):
        """number of stored members (one 8-byte read of indptr[-1]; the arrays may be capacity-sized)"""
        return int(self.indptr[-1].item()) if self.n_rows else 0

    @classmethod
    def from_sets(cls, sets, n_cols=None):
        """Segmented sort of the sampled sets by node id (random_walks.py:79-80).  Row i = root query[i].
        Sets in fused-row form are already finished rows; lazy sets give a capacity-sized SpG (row offsets rule)."""
        dev = sets.ids.device
        n = sets.nsize.numel()
        max_data = sets.ukeys.numel()          # while lazy: the capacity of the distinct-row table view (an upper bound)
        if sets.data is not None:
            return cls(sets.row_off, sets.ids, sets.data, max_len=sets.stride, shape=(n, n_cols or n), max_data=max_data)
        if sets.sf is None and sets.slot is None:
            raise ValueError("SpG.from_sets needs de-duplicated sets (sample_sets(..., dedup=True))")
        indices = torch.empty_like(sets.ids)
        data = torch.empty_like(sets.ids)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        with _timed("spg_build"):
            if sets.sf is not None:
                check(lib().subgacc_spg_build(ptr(sets.row_off), n, ptr(sets.ids), ptr(sets.sf), None, 0, sets.stride,
                                              ptr(indices), ptr(data), ptr(flags), stream_ptr()))
            else:   # table slots are translated to SFptr inside the sort kernel: no pass over `sf` at all
                check(lib().subgacc_spg_build(ptr(sets.row_off), n, ptr(sets.ids), ptr(sets.slot), ptr(sets.table),
                                              sets.capacity, sets.stride, ptr(indices), ptr(data), ptr(flags),
                                              stream_ptr()))
        return cls(sets.row_off, indices, data, max_len=sets.stride, shape=(n, n_cols or n), max_data=max_data)

    @classmethod
    def from_scipy(cls, z, device=None):
        """Upload a scipy CSR (e.g. the PPR matrix of sampler/pprgo.py) as an SpG; float payloads stay float64."""
        device = device or _lib.require_device()
        z = z.tocsr()
        z.sort_indices()
        data = z.data
        if np.issubdtype(data.dtype, np.floating):
            data = data.astype(np.float64)
        else:
            data = data.astype(np.int32)
        return cls(torch.from_numpy(z.indptr.astype(np.int64)).to(device),
                   torch.from_numpy(z.indices.astype(np.int32)).to(device),
                   torch.from_numpy(data).to(device), shape=z.shape)

    def save(self, path, encode=None):
        """Persist the store (and optionally its Z_SF table) -- the reference never does; the
        sampling stage then does not have to be repeated between runs.  Capacity-sized arrays are trimmed."""
        nnz = self.nnz
        blob = {"indptr": self.indptr.cpu(), "indices": self.indices[:nnz].cpu(), "data": self.data[:nnz].cpu(),
                "shape": tuple(self.shape), "max_len": self.max_len, "max_data": self.max_data}
        if encode is not None:
            blob["encode"] = torch.as_tensor(encode).cpu()
        torch.save(blob, path)

    @classmethod
    def load(cls, path, device=None):
        """-> (SpG on `device`, encode tensor or None)"""
        device = device or _lib.require_device()
        blob = torch.load(path, map_location="cpu")
        z = cls(blob["indptr"].to(device), blob["indices"].to(device), blob["data"].to(device), max_len=blob["max_len"],
                shape=tuple(blob["shape"]), max_data=blob["max_data"])
        enc = blob.get("encode")
        return z, (enc.to(device) if enc is not None else None)

    keyrows = False       # keyed(): the payload is the member's LP key, joined by subgacc_sjoin_fill_keys

    def keyed(self, enc, num_walks):
        """The same store with every member's 32-bit LP KEY as payload instead of SFptr+1 (shares indptr / indices): the join
        then unpacks a feature row from the key (counts / num_walks, the IEEE division of main.py:174) instead of gathering
        16 bytes per output slot from the Z_SF table -- bit-identical `xz`, ~1.4x the join rate on the cit2-like store.  For
        the reference's flow: `z, enc = subg_matrix(...)` once, `zk = z.keyed(enc, num_walks)` once, then per batch
        `gather(edge, zk, encode=zk.slot_table())`.
        enc: the LP rows as subg_matrix returns them -- integer counts [c+1, k], zero row in front, row p belongs to SFptr+1 = p,
        column 0 = num_walks on a root's own row (subg_acc.c:900-955).  Needs (k-1) * bits(num_walks) + 1 <= 31."""
        if self.data.dtype != torch.int32:
            raise TypeError("keyed() re-keys an SFptr (integer) SpG")
        tab = torch.as_tensor(enc).to(device=self.device, dtype=torch.int64)
        if tab.ndim != 2 or tab.shape[1] < 2 or tab.shape[0] <= self.max_data:
            raise IndexError(f"the LP table must be [c+1, k] with more than {self.max_data} rows")
        m = tab.shape[1] - 1
        shift = check(lib().subgacc_key_shift(int(num_walks), m))
        if m * shift + 1 > 31:
            raise AssertionError(f"LP keys of {m} steps x {shift} bits do not fit 32 bits")
        if int(tab[:, 1:].max().item()) > int(num_walks) or int(tab.min().item()) < 0:
            raise ValueError("LP counts outside [0, num_walks]: not the table of this num_walks")
        key = (tab[:, 0] != 0).to(torch.int64) << (m * shift)
        for j in range(1, m + 1):
            key |= tab[:, j] << ((m - j) * shift)
        keytab = key.to(torch.int32)
        nnz = self.nnz
        keys = torch.empty_like(self.data)
        keys[:nnz] = torch.index_select(keytab, 0, self.data[:nnz])
        z = SpG(self.indptr, self.indices, keys, max_len=self.max_len, shape=self.shape, max_data=0)
        z.keyrows, z.key_M, z.key_m = True, int(num_walks), m
        return z

    def slot_table(self):
        """what gather(..., encode=) takes for a keyed() store: the marker that the join unpacks the keys itself"""
        if not self.keyrows:
            raise ValueError("slot_table() belongs to a keyed() store; a plain SpG is joined with its Z_SF table")
        return KEY_ROWS_ENCODE

    def aligned(self, pitch=None):
        """The same store laid out for a SERVING loop (round 6): every row on whole 128-byte lines at a fixed pitch -- row r's ids
        begin at r*pitch with the row's LENGTH in their first slot, its payload at the same pitch (include/subgacc.h: headed rows).
        A join then needs no row pointer (a packed store reads a 128-byte line for 16 bytes of them, per row), and no row begins or
        ends inside a line it shares with its neighbours.  -> HeadedSpG: gather / hgather / CapturedJoin(Pool) take it like the
        store it was made from (same encode argument, bit-identical results); the count / pair / index forms and to_scipy() stay
        with the packed store.  pitch: words between two rows, default the longest row + 1 rounded up to a multiple of 32; the store
        grows by pitch / (mean row length) -- HBM is what an MI355X has plenty of (288 GB), lines per second it has not."""
        return HeadedSpG.from_spg(self, pitch)

    def to_scipy(self):
        import scipy.sparse as sp
        nnz = self.nnz
        return sp.csr_matrix((self.data[:nnz].cpu().numpy(), self.indices[:nnz].cpu().numpy(), self.indptr.cpu().numpy()),
                             shape=self.shape)


KEY_ROWS_ENCODE = "key rows"      # StridedSpG.slot_table() of a key-rows batch: the join needs no table


class HeadedSpG:
    """A resident store in the HEADED row layout of include/subgacc.h (ABI 7): `ids` int32 [n_rows * pitch] -- ids[r*pitch] = the
    length of row r, its members (ascending node ids) behind it --, `data` [n_rows * pitch] with member t's payload at r*pitch + t
    (int32 SFptr+1, int32 LP keys of a keyed() store, float64 PPR scores).  Made by SpG.aligned(); joined by gather / hgather /
    gather_many / CapturedJoin / CapturedJoinPool exactly like the SpG it came from."""

    def __init__(self, ids, data, pitch, n_rows, max_len, shape, max_data=None):
        assert ids.dtype == torch.int32 and data.dtype in (torch.int32, torch.float64) and pitch > 1
        self.ids, self.data, self.pitch, self.n_rows = ids, data, int(pitch), int(n_rows)
        self.max_len = int(max_len)              # the longest row (<= pitch - 1): sizes the worst-case output of a one-call join
        self.max_data, self.shape, self.device = max_data, shape, ids.device
        self.keyrows, self.key_M, self.key_m = False, 0, 0

    @classmethod
    def from_spg(cls, z, pitch=None):
        if not isinstance(z, SpG):
            raise TypeError("aligned() lays a packed SpG out again")
        need = int(z.max_len) + 1
        pitch = -(-need // 32) * 32 if pitch is None else int(pitch)
        if pitch < need:
            raise ValueError(f"pitch {pitch} < longest row + 1 = {need}")
        dev = z.device
        ids = torch.empty(max(z.n_rows, 1) * pitch, dtype=torch.int32, device=dev)
        data = torch.empty(max(z.n_rows, 1) * pitch, dtype=z.data.dtype, device=dev)
        flags = torch.zeros(4, dtype=torch.int32, device=dev)
        check(lib().subgacc_rows_to_headed(ptr(z.indptr), z.n_rows, ptr(z.indices), ptr(z.data), z.data.element_size(), pitch, ptr(ids),
                                           ptr(data), ptr(flags), stream_ptr()))
        if int(flags[3].item()) & 1:
            raise _lib.SubgAccError("aligned(): a row is longer than SpG.max_len says")
        h = cls(ids, data, pitch, z.n_rows, z.max_len, z.shape, max_data=z.max_data)
        if z.keyrows:
            h.keyrows, h.key_M, h.key_m = True, z.key_M, z.key_m
        return h

    @property
    def nbytes(self):
        return self.ids.numel() * 4 + self.data.numel() * self.data.element_size()

    def slot_table(self):
        if not self.keyrows:
            raise ValueError("slot_table() belongs to a keyed() store; a plain SpG is joined with its Z_SF table")
        return KEY_ROWS_ENCODE

    def row_lengths(self):
        """int32 [n_rows]: what the rows' first slots hold"""
        return self.ids.view(-1, self.pitch)[: self.n_rows, 0].contiguous()

    def to_spg(self):
        """back to packed rows (tests; the layout is for serving)"""
        lens = self.row_lengths().long()
        indptr = torch.zeros(self.n_rows + 1, dtype=torch.int64, device=self.device)
        torch.cumsum(lens, 0, out=indptr[1:])
        col = torch.arange(self.pitch - 1, device=self.device)[None, :]
        keep = col < lens[:, None]
        idx = self.ids.view(-1, self.pitch)[: self.n_rows, 1:][keep]
        dat = self.data.view(-1, self.pitch)[: self.n_rows, : self.pitch - 1][keep]
        z = SpG(indptr, idx.contiguous(), dat.contiguous(), max_len=self.max_len, shape=self.shape, max_data=self.max_data)
        if self.keyrows:
            z.keyrows, z.key_M, z.key_m = True, self.key_M, self.key_m
        return z


class StridedSpG:
    """The SpG of a transient batch in the layout the fused walk kernel writes: row i = indices / slot
    [i*stride, +nsize[i]), node ids ascending, payload = slot in the numbered table of distinct LP rows.
    `gather` / `hgather` join straight from it (subgacc_sjoin_*_rows); to_csr() makes the packed SpG."""

    def __init__(self, sets, n_cols):
        assert sets.strided
        self.sets = sets
        self.indices, self.slot, self.nsize = sets.ids, sets.slot, sets.nsize
        self.stride = self.max_len = int(sets.stride)
        self.table, self.capacity = sets.table, sets.capacity
        self.keyrows = bool(getattr(sets, "keyrows", False))      # payload = LP keys (subgacc_sjoin_fill_keyrows joins them)
        self._slot_table = None
        self.n_rows = sets.nsize.numel()
        self.shape = (self.n_rows, n_cols)
        self.device = sets.ids.device

    def slot_table(self):
        """The feature table indexed by table slot (SampledSets.feature_table_by_slot): pass it as `encode` and the
        join skips the slot -> SFptr indirection altogether."""
        if self.keyrows:          # nothing to index: the join unpacks the keys; the marker tells it so
            self._slot_table = KEY_ROWS_ENCODE
            return self._slot_table
        self._slot_table = self.sets.feature_table_by_slot()
        return self._slot_table

    @property
    def max_data(self):
        """largest SFptr+1 a member can carry: the distinct-row count (its capacity bound while the sets are lazy)"""
        return self.sets.number().ukeys.numel()

    @property
    def nnz(self):
        return self.sets.X

    def to_csr(self):
        if self.keyrows and self.sets.key64:      # 64-bit key rows: the packed rows come from the table form of the same batch
            return StridedSpG(self.sets.table_form(), self.shape[1]).to_csr()
        self.sets.number()           # the packed rows carry SFptr+1: the table must be numbered by now (key rows: registered now)
        n, dev = self.n_rows, self.device
        row_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        ws = torch.empty(max(lib().subgacc_scan_workspace_bytes(n), 8), dtype=torch.uint8, device=dev)
        check(lib().subgacc_exclusive_scan_i32(ptr(self.nsize), n, ptr(row_off), ptr(ws), ws.numel(), stream_ptr()))
        X = int(row_off[-1].item()) if n else 0
        ids = torch.empty(X, dtype=torch.int32, device=dev)
        data = torch.empty(X, dtype=torch.int32, device=dev)
        if X and self.keyrows:       # rows of LP keys: SFptr+1 is looked up in the table number() has just built
            flags = torch.zeros(4, dtype=torch.int32, device=dev)
            uk = self.sets.ukeys
            check(lib().subgacc_keyrows_compact(ptr(self.indices), ptr(self.slot), ptr(self.nsize), ptr(row_off), n, self.stride, 0,
                                                ptr(self.sets._ktable), self.sets._kcap, ptr(uk), ptr(self.sets._kcount), uk.numel(),
                                                ptr(ids), ptr(data), None, 0, None, ptr(flags), stream_ptr()))
        elif X:
            check(lib().subgacc_compact_rows(ptr(self.indices), ptr(self.slot), ptr(self.nsize), ptr(row_off), n, self.stride,
                                             ptr(ids), ptr(data), ptr(self.table), self.capacity, stream_ptr()))
        return SpG(row_off, ids, data, max_len=self.stride, shape=self.shape, max_data=self.max_data)


FUSED_MIN_GRAPH_BYTES = 64 << 20     # adjacency beyond this no longer lives in the 8 x 4 MiB L2s


def prefers_fused(csr, hops):
    """Is the fused-row walk kernel the faster way to the finished rows of a batch?  From 2 hops on: yes -- the specialised
    kernel (csrc/walk_rows.hip) beats the pipelined general kernel + finish_rows on the twitter-like graph (1.89 vs 1.99 ms
    per step, 2 hops), ties on the L2-resident collab-like one (1.05 vs 1.06 ms) and wins clearly at 3 hops (cit2-like
    1.73 vs 2.21 ms).  1-hop walks keep the general pair (walk_rows is not instantiated for them)."""
    return hops >= 2


def sample_spg(csr, query, num_walks=200, num_steps=3, seed=111413, rng="rand_r", bucket=-1, fused=None, lazy=False,
               strided=False, **kw):
    """sample -> SpG on the GPU: (SpG, SampledSets) -- the sets carry ukeys / nsize / feature_table().

    `num_steps` = walk hops (gset_sampler's meaning).  fused=True lets the walk kernel emit finished SpG rows
    (csrc/walk.hip SPG mode; falls back to the general pipeline when it does not apply); fused=None asks
    prefers_fused(): walks of >= 2 hops (measured: cit2-like, 3 hops +28 % pairs/s; twitter-like, 2 hops +5 %;
    collab-like, 2 hops, an 8 MB graph that lives in L2: +1 %).  lazy=True leaves every size on the
    device (no host round trip until SampledSets.resolve() / SpG.nnz); arrays are capacity-sized.  A lazy batch cannot
    recover by itself from a table of distinct LP rows that is too small (`uniq_capacity`, or more than
    sampler.RANK_LIMIT distinct rows): resolve() raises SubgAccError then and the batch is sampled again with
    lazy=False, which regrows the table / takes the packed path on its own (a serving loop: catch, re-run that batch).
    strided=True: for a batch that is sampled, joined and dropped -- returns a StridedSpG (no packed copy of the rows):
    the fused-row walk kernel's output, or the general walk kernel's sets finished in place (subgacc_finish_rows)."""
    sets = None
    if fused is None:
        fused = prefers_fused(csr, num_steps)
    if strided:     # transient batch: rows stay in the walk kernel's staging layout (falls through when it does not apply)
        for fr in ((True, False) if fused else (False, True)):      # the faster walk kernel first, the other as fallback
            sets = sample_sets(csr, query, num_walks=num_walks, num_steps=num_steps, bucket=bucket, seed=seed, rng=rng,
                               fused_rows=fr, lazy=lazy, strided=True, **kw)
            if sets is not None:
                return StridedSpG(sets, csr.num_nodes), sets
    if fused:
        sets = sample_sets(csr, query, num_walks=num_walks, num_steps=num_steps, bucket=bucket, seed=seed, rng=rng,
                           fused_rows=True, lazy=lazy, **kw)
    if sets is None:
        sets = sample_sets(csr, query, num_walks=num_walks, num_steps=num_steps, bucket=bucket, seed=seed, rng=rng,
                           lazy=lazy, **kw)
    return SpG.from_sets(sets, n_cols=csr.num_nodes), sets


def subg_matrix(G, train_idx, num_walks=200, num_steps=4, seed=111413, rng="rand_r", device=None, fused=None):
    """Drop-in for sampler/random_walks.py:74-82: returns (z, enc).

    z   -- SpG on the GPU (row i = sampled set of train_idx[i]); the reference indexes rows by node id and
           always passes train_idx = arange(N) (main.py:168-178), for which both conventions coincide.
    enc -- numpy int16 [c+1, num_steps] with the all-zero row 0, exactly what the reference returns, so
           `torch.from_numpy(xpe).to(device).float() / num_walks` (main.py:174) keeps working.
    `num_steps` is the CLI value: the walks have num_steps-1 hops (random_walks.py:78).
    """
    if _lib.VERBOSE:
        print(f'Start sampling for #{len(train_idx)} nodes with {num_walks} {num_steps}-step walks')
    csr = G if isinstance(G, DeviceCSR) else DeviceCSR(G.indptr, G.indices, device)
    z, sets = sample_spg(csr, train_idx, num_walks=num_walks, num_steps=num_steps - 1, seed=seed, rng=rng, fused=fused)
    enc = sets.enc_int16().cpu().numpy()
    enc = np.insert(enc, 0, np.zeros((1, num_steps), dtype=enc.dtype), axis=0)
    z.sets = sets
    return z, enc


# ------------------------------------------------------------------------------------------------------------------
# The walk_sampler route to the SpG (sampler/random_walks.py:35-71: np_sampling / rw_matrix) -- SUREL's original
# offline stage, kept by the reference next to subg_matrix.  Same kernels as above in walk_sampler's form: step-major
# first-visit order, first hop without replacement, one rand_r stream per OpenMP thread and per BATCH (every call of
# walk_sampler starts again from seed + thread id, so the result depends on batch size and thread count).
def _walk_batches(csr, target, bsize, num_walks, hops, nthread, seed, rng):
    dev = csr.device
    tgt = torch.as_tensor(target).to(device=dev, dtype=torch.int32).contiguous()
    parts = []
    for lo in range(0, tgt.numel(), int(bsize)):          # gen_batch(target, bsize, keep=True), random_walks.py:25-29
        parts.append(sample_sets(csr, tgt[lo:lo + int(bsize)], num_walks=num_walks, num_steps=hops, seed=seed, rng=rng,
                                 first_hop_wo=True, order=_lib.ORDER_STEP_MAJOR, cap_root_degree=False,
                                 rng_streams=max(int(nthread), 1), dedup=False))
    if not parts:
        z64 = torch.zeros(0, dtype=torch.int64, device=dev)
        return SampledSets(nsize=torch.zeros(0, dtype=torch.int32, device=dev), row_off=torch.zeros(1, dtype=torch.int64, device=dev),
                           ids=torch.zeros(0, dtype=torch.int32, device=dev), keys=z64, sf=None, ukeys=z64,
                           num_walks=num_walks, num_steps=hops, stride=num_walks * hops + 1)
    nsize = torch.cat([p.nsize for p in parts])
    row_off = torch.zeros(nsize.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(nsize, 0, out=row_off[1:])
    return SampledSets(nsize=nsize, row_off=row_off, ids=torch.cat([p.ids for p in parts]),
                       keys=torch.cat([p.keys for p in parts]), sf=None, ukeys=None, num_walks=num_walks,
                       num_steps=hops, stride=parts[0].stride)


def np_sampling(ptr, neighs, bsize, target, num_walks=200, num_steps=4, nthread=1, seed=111413, rng="rand_r"):
    """Drop-in for sampler/random_walks.py:35-47: (node ids int32[X], landing counts int32[X, num_steps+1]) of the
    sets of `target`, batch by batch.  `num_steps` = walk hops here, as in the reference's call (:61-62).
    nthread = the OpenMP team size whose streams are reproduced (the reference takes the machine's default)."""
    csr = ptr if isinstance(ptr, DeviceCSR) else DeviceCSR(ptr, neighs)
    sets = _walk_batches(csr, target, bsize, num_walks, num_steps, nthread, seed, rng)
    if sets.ids.numel() == 0:
        return np.zeros(0, np.int32), np.zeros((0, num_steps + 1), np.int32)
    return sets.ids.cpu().numpy(), sets.counts_int32().cpu().numpy()


def rw_matrix(G, train_idx, num_walks=200, num_steps=4, batch_size=2000, reduced=True, nthread=1, seed=111413,
              rng="rand_r", device=None):
    """Drop-in for sampler/random_walks.py:58-71: (z, freqs).

    z     -- SpG, row i = set of train_idx[i] (the reference assumes train_idx = arange, :51), data = LP row number + 1
    freqs -- numpy int32 [c+1, num_steps] with the all-zero row 0.
    reduced=True numbers the distinct LP rows in ASCENDING order of their base-(M+1) projection (:64-68:
    fastremap.unique sorts) -- not in first-occurrence order as subg_matrix does; the packed 64-bit key orders the
    rows the same way (step 0 = LEAD bit, then steps 1..m, most significant first)."""
    csr = G if isinstance(G, DeviceCSR) else DeviceCSR(G.indptr, G.indices, device=device)
    hops = num_steps - 1
    sets = _walk_batches(csr, train_idx, batch_size, num_walks, hops, nthread, seed, rng)
    X = sets.ids.numel()
    dev = csr.device
    if hops * check(lib().subgacc_key_shift(num_walks, hops)) >= 63:
        raise AssertionError("rw_matrix: LP key uses bit 63; its signed order would not be the row order")
    if reduced and X:
        sets = dedup_lp_rows(sets)
        ukeys, order = torch.sort(sets.ukeys)                       # c keys: a few hundred to a few thousand
        rank = torch.empty_like(order)
        rank[order] = torch.arange(order.numel(), device=dev)
        sets.sf = rank.to(torch.int32)[sets.sf.long()]
        sets.ukeys = ukeys
        rows = SampledSets(nsize=None, row_off=None, ids=torch.empty(ukeys.numel(), dtype=torch.int32, device=dev), keys=ukeys,
                           sf=None, ukeys=None, num_walks=num_walks, num_steps=hops, stride=sets.stride)
        freqs = rows.counts_int32().cpu().numpy()
    else:
        sets.sf = torch.arange(X, dtype=torch.int32, device=dev)   # every member its own row (:69-70)
        sets.ukeys = sets.keys
        freqs = sets.counts_int32().cpu().numpy() if X else np.zeros((0, num_steps), np.int32)
    z = SpG.from_sets(sets, n_cols=csr.num_nodes)
    return z, np.insert(freqs, 0, np.zeros((1, num_steps), freqs.dtype), axis=0)
