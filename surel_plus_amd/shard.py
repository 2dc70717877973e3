"""Multi-GPU sharding of the hot path: one process per GPU, graph CSR replicated in every GPU's HBM,
roots / query pairs split into contiguous ranges.  The reference has no distributed code at all; roots are
independent in set_sampler (subg_acc/subg_acc.c:745) and pairs are independent in gather (train.py:13-45), so
the data path needs NO collective.  The single exchange step is the numbering of distinct LP rows, which the
reference does with one global hash pass (subg_acc.c:957-978): every rank numbers its own rows, the (tiny)
per-rank tables are all-gathered once (RCCL over xGMI on GPUs, gloo in the CPU tests) and merged in rank
order, which reproduces the single-process first-occurrence numbering exactly.
"""
import torch


def shard_range(n, rank, world):
    """Contiguous range [lo, hi) of rank `rank` out of `world`; the first n % world ranks take one more."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def merge_unique_tables(tables):
    """tables: list (rank order) of int64 tensors, each the distinct LP keys of one rank in that rank's
    first-occurrence order.  Returns (global_keys, [local->global index tensor per rank]) where global_keys
    is the first-occurrence order over the concatenation == the order a single process would produce."""
    dev = tables[0].device
    sizes = [int(t.numel()) for t in tables]
    cat = torch.cat(tables) if sum(sizes) else torch.zeros(0, dtype=torch.int64, device=dev)
    if cat.numel() == 0:
        return cat, [torch.zeros(0, dtype=torch.int64, device=dev) for _ in tables]
    uniq, inv = torch.unique(cat, return_inverse=True)
    pos = torch.arange(cat.numel(), device=dev)
    first = torch.full((uniq.numel(),), cat.numel(), dtype=torch.int64, device=dev).scatter_reduce_(0, inv, pos, "amin")
    order = torch.argsort(first)                       # distinct keys by first appearance
    gid_of_uniq = torch.empty_like(order)
    gid_of_uniq[order] = torch.arange(order.numel(), device=dev)
    gids = gid_of_uniq[inv]
    return cat[first[order]], list(torch.split(gids, sizes))


def all_gather_varlen(t, group=None):
    """all_gather of 1-D tensors (any dtype, the same on every rank) of different lengths: pad to the max length, one
    collective each for the sizes and the payload.  gloo has no all_gather for device tensors: they are staged
    through the host there (the 1-GPU test set-up: several ranks share cuda:0 over gloo); RCCL gathers in HBM."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return [p.to(t.device) for p in all_gather_varlen(t.cpu(), group)]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:s] for b, s in zip(bufs, sizes)]


def rand_r_calls(indptr, roots, num_walks, num_steps, first_hop_wo=True, cap_root_degree=True):
    """Number of rand_r draws the reference's sequential loop makes for `roots` (subg_acc.c:763-776: M shuffle draws when
    deg > M; :790-808: one draw per later step of every walk; nothing for an isolated root) -- the stream position
    at which the NEXT root starts.  This is what a rank needs to know about the roots in front of its range
    (`calls_before` of subgacc_rng_positions); same arithmetic as rng_calls_kernel (csrc/walk.hip).  Works on
    host or device tensors (a gather of the roots' degrees and a sum)."""
    M, m = int(num_walks), int(num_steps)
    r = torch.as_tensor(roots).long().reshape(-1)
    if r.numel() == 0:
        return 0
    ip = indptr if torch.is_tensor(indptr) else torch.as_tensor(indptr)
    r = r.to(ip.device)
    ok = (r >= 0) & (r < ip.numel() - 1)             # a root out of range draws nothing (and the walk kernel flags it)
    r = r.clamp(0, max(ip.numel() - 2, 0))
    deg = (ip[r + 1] - ip[r]).long() * ok.long()
    if cap_root_degree:
        deg = deg.clamp(max=1000000)                      # NEBMAX, subg_acc.c:13
    if first_hop_wo:
        calls = (deg > M).long() * M + M * (m - 1)
    else:
        calls = torch.full_like(deg, M * m)
    return int((calls * (deg > 0).long()).sum().item())


def hip_sampler(csr, **cfg):
    """The product sampler (sampler.sample_sets over a DeviceCSR) in the form sample_sets_sharded drives:
    sampler(query, lo, hi) samples roots [lo, hi) of `query` exactly as a single process sampling all of `query`
    would have -- Philox streams are keyed by root id; the sequential rand_r stream is entered at the position the
    roots in front of `lo` leave it at (rand_r_calls)."""
    from . import sampler as S

    def run(query, lo, hi, **kw):
        q = S._as_query(query, csr.device)
        args = dict(cfg, **kw)
        if args.get("rng", "rand_r") == "rand_r" and lo > 0:
            args["calls_before"] = rand_r_calls(csr.indptr, q[:lo], args.get("num_walks", 100), args.get("num_steps", 3),
                                                args.get("first_hop_wo", True), args.get("cap_root_degree", True))
        return S.sample_sets(csr, q[lo:hi], **args)
    return run


def sample_sets_sharded(sampler, query, rank, world, group=None, **kw):
    """Sample the contiguous share of `query` owned by this rank and give its LP rows GLOBAL numbers.

    sampler(query, lo, hi, **kw) -> object with .sf / get_sf() (int32/int64 [X_local]) and .ukeys (int64 [c_local]):
    hip_sampler(csr, ...) on GPUs; an oracle-backed stand-in in the CPU tests.
    Returns (sets, global_ukeys, (lo, hi)); sets.sf is relabelled to index global_ukeys, sets.ukeys replaced."""
    lo, hi = shard_range(len(query), rank, world)
    sets = sampler(query, lo, hi, **kw)
    if hasattr(sets, "resolve"):
        sets.resolve()
    if world == 1:
        return sets, sets.ukeys, (lo, hi)
    tables = all_gather_varlen(sets.ukeys, group)
    gkeys, maps = merge_unique_tables(tables)
    sf = sets.get_sf() if hasattr(sets, "get_sf") else sets.sf
    sets.sf = maps[rank].to(sf.dtype)[sf.long()]
    sets.ukeys = gkeys
    return sets, gkeys, (lo, hi)


def sample_spg_sharded(csr, roots, rank, world, group=None, num_walks=200, num_steps=3, seed=111413, rng="rand_r",
                       bucket=-1, fused=None, replicate=True):
    """The offline stage (subg_matrix, sampler/random_walks.py:74-82) split over `world` GPUs: every rank samples the
    contiguous range of `roots` it owns with the product kernels (csr replicated), the per-rank tables of distinct LP
    rows are all-gathered and merged in rank order -- the numbering a single process gives (subg_acc.c:957-978) --
    and, with replicate=True, the finished rows are all-gathered so that every GPU holds the whole SpG (what the
    join shards its pairs over).  Returns (SpG, global ukeys int64 [c], (lo, hi)); with replicate=False the SpG holds
    rows [lo, hi) only.  `num_steps` = walk hops.  Bit-identical to the single-process sample_spg() in both RNG modes."""
    from . import sampler as S
    from .spg import SpG, sample_spg
    q = S._as_query(roots, csr.device)
    lo, hi = shard_range(q.numel(), rank, world)
    kw = {}
    if rng == "rand_r" and lo > 0:
        kw["calls_before"] = rand_r_calls(csr.indptr, q[:lo], num_walks, num_steps)
    z, sets = sample_spg(csr, q[lo:hi], num_walks=num_walks, num_steps=num_steps, seed=seed, rng=rng, bucket=bucket,
                         fused=fused, **kw)
    sets.resolve()
    nnz = z.nnz
    ids, data = z.indices[:nnz], z.data[:nnz]
    gkeys = sets.ukeys
    if world > 1:
        gkeys, maps = merge_unique_tables(all_gather_varlen(sets.ukeys, group))
        if nnz:
            data = (maps[rank].to(torch.int32) + 1)[(data - 1).long()]      # SFptr+1 of the global numbering
    n_cols = csr.num_nodes
    if world > 1 and replicate:
        row_off, ids, data = replicate_rows(sets.nsize, ids, data, group)
        return SpG(row_off, ids, data, max_len=sets.stride, shape=(q.numel(), n_cols), max_data=gkeys.numel()), gkeys, (lo, hi)
    return SpG(z.indptr, ids.contiguous(), data.contiguous(), max_len=sets.stride, shape=(hi - lo, n_cols),
               max_data=gkeys.numel()), gkeys, (lo, hi)


def lp_table(gkeys, num_walks, num_steps):
    """Z_SF of a numbering held as packed keys: float32 [c+1, num_steps+1] = [0-row ; enc / M] (main.py:174 +
    random_walks.py:81), on the device of `gkeys`."""
    from ._lib import check, lib, ptr, stream_ptr
    c = gkeys.numel()
    out = torch.empty((c + 1, num_steps + 1), dtype=torch.float32, device=gkeys.device)
    check(lib().subgacc_unpack_lp(ptr(gkeys.contiguous()), c, None, int(num_walks), int(num_steps), None, None, ptr(out), 1,
                                  stream_ptr()))
    return out


def shard_pairs(edge, rank, world):
    """Contiguous share of the query pairs [2, B] of this rank (SpG replicated, no collective)."""
    lo, hi = shard_range(edge.shape[1], rank, world)
    return edge[:, lo:hi], (lo, hi)


def replicate_rows(nsize, ids, data, group=None, alloc=torch.empty):
    """The one-off exchange that makes a row-sharded SpG resident on every GPU: every rank holds the rows
    of its contiguous root range (nsize [n_r], ids / data [X_r], rank order = row order); returns the full
    (row_off int64 [n+1], ids, data) on every rank.  Not part of the steady state.

    Memory: the full arrays are allocated ONCE (through `alloc`, so that a test can count) and every rank's slice travels
    straight into its place -- one small all_gather of the (n_r, X_r) pairs, then per rank one broadcast of each array into the
    slice [off_r, off_r + X_r) of the output (RCCL over xGMI on GPUs).  Peak transient = the store itself (+ this rank's own
    slices, which the caller holds anyway): no padded per-rank buffers, no list of `world` copies, no concatenation -- the
    twitter-scale store SURVEY 8(e) sizes at 35-50 GB stays at 35-50 GB per GPU while it is being replicated.
    gloo has no collectives for device tensors (the 1-GPU rehearsal, several ranks on cuda:0): staged through the host there,
    one slice at a time."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = nsize.device
    staged = nsize.is_cuda and dist.get_backend(group) == "gloo"
    mine = torch.tensor([nsize.numel(), ids.numel()], dtype=torch.int64, device="cpu" if staged else dev)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    sizes = torch.stack(sizes).cpu().tolist()             # [[n_r, X_r]] in rank order
    n_tot, x_tot = sum(s_[0] for s_ in sizes), sum(s_[1] for s_ in sizes)
    if data.numel() != ids.numel():
        raise ValueError("replicate_rows: ids and data of a rank must have the same length")
    out_ns = alloc(n_tot, dtype=nsize.dtype, device=dev)
    out_ids = alloc(x_tot, dtype=ids.dtype, device=dev)
    out_data = alloc(x_tot, dtype=data.dtype, device=dev)
    row_off = alloc(n_tot + 1, dtype=torch.int64, device=dev)
    n_off = x_off = 0
    for r, (n_r, x_r) in enumerate(sizes):
        src = dist.get_global_rank(group, r) if group is not None else r
        for full, local, off, cnt in ((out_ns, nsize, n_off, n_r), (out_ids, ids, x_off, x_r), (out_data, data, x_off, x_r)):
            if cnt == 0:
                continue
            piece = full[off: off + cnt]                   # a view: the broadcast writes the output in place
            if r == rank:
                piece.copy_(local)
            if staged:                                     # (rehearsal only: one slice at a time through the host)
                host = piece.cpu() if r == rank else torch.empty(cnt, dtype=full.dtype)
                dist.broadcast(host, src=src, group=group)
                if r != rank:
                    piece.copy_(host)
            else:
                dist.broadcast(piece, src=src, group=group)
        n_off, x_off = n_off + n_r, x_off + x_r
    row_off[0] = 0
    torch.cumsum(out_ns, 0, out=row_off[1:])
    return row_off, out_ids, out_data
