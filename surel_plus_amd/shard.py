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
    collective each for the sizes and the payload."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
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


def sample_sets_sharded(sampler, query, rank, world, group=None, **kw):
    """Sample the contiguous share of `query` owned by this rank and give its LP rows GLOBAL numbers.

    sampler(query_slice, lo=..., **kw) -> object with .sf (int32/int64 [X_local]) and .ukeys (int64 [c_local])
    (surel_plus_amd.sampler.sample_sets bound to a DeviceCSR on GPUs; an oracle-backed stand-in in the CPU tests).
    Returns (sets, global_ukeys, (lo, hi)); sets.sf is relabelled to index global_ukeys, sets.ukeys replaced.
    With rng="philox" the sets do not depend on the sharding; with rng="rand_r" the caller passes
    calls_before for its range (subgacc_rng_positions)."""
    lo, hi = shard_range(len(query), rank, world)
    sets = sampler(query[lo:hi], lo=lo, **kw)
    if world == 1:
        return sets, sets.ukeys, (lo, hi)
    tables = all_gather_varlen(sets.ukeys, group)
    gkeys, maps = merge_unique_tables(tables)
    sf = sets.get_sf() if hasattr(sets, "get_sf") else sets.sf
    sets.sf = maps[rank].to(sf.dtype)[sf.long()]
    sets.ukeys = gkeys
    return sets, gkeys, (lo, hi)


def shard_pairs(edge, rank, world):
    """Contiguous share of the query pairs [2, B] of this rank (SpG replicated, no collective)."""
    lo, hi = shard_range(edge.shape[1], rank, world)
    return edge[:, lo:hi], (lo, hi)


def replicate_rows(nsize, ids, data, group=None):
    """The one-off exchange that makes a row-sharded SpG resident on every GPU: every rank holds the rows
    of its contiguous root range (nsize [n_r], ids / data [X_r], rank order = row order); returns the full
    (row_off int64 [n+1], ids, data) on every rank.  Three all-gathers (RCCL ring over xGMI on GPUs; sized by the
    store, e.g. 8 B per member), not part of the steady state."""
    ns = torch.cat(all_gather_varlen(nsize, group))
    row_off = torch.zeros(ns.numel() + 1, dtype=torch.int64, device=ns.device)
    torch.cumsum(ns, 0, out=row_off[1:])
    return row_off, torch.cat(all_gather_varlen(ids, group)), torch.cat(all_gather_varlen(data, group))
