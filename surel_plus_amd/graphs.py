"""Synthetic graphs and SpG stores in the shapes BASELINE.json's configs name (there is no network: the
OGB datasets cannot be downloaded).  Generated on the GPU with torch, seeded, and symmetrised the way the
reference's loader does (G + G.T, dataloader.py:122-135): undirected, simple, sorted CSR, int32 ids.

  collab-like   N=235,868   avg deg 8.2     (paper Table 7)
  ppa-like      N=576,289   avg deg 73.7
  cit2-like     N=2,927,963 avg deg 20.7
"""
import torch

from .sampler import DeviceCSR

PRESETS = {
    "collab": dict(N=235_868, avg_deg=8.2, seed=0),
    "ppa": dict(N=576_289, avg_deg=73.7, seed=1),
    "cit2": dict(N=2_927_963, avg_deg=20.7, seed=2),
}


def powerlaw_graph(N, avg_deg, seed=0, exponent=2.5, device="cuda", max_weight_frac=0.002):
    """Chung-Lu style heavy-tailed undirected graph with ~N*avg_deg/2 distinct edges -> DeviceCSR."""
    gen = torch.Generator(device=device).manual_seed(int(seed))
    # expected-degree weights w_i ~ i^(-1/(exponent-1)), capped so the largest hub stays a small fraction of N
    ranks = torch.arange(1, N + 1, device=device, dtype=torch.float64)
    w = ranks.pow(-1.0 / (exponent - 1.0))
    w = torch.minimum(w, w.sum() * max_weight_frac / avg_deg)
    w = w[torch.randperm(N, device=device, generator=gen)]        # hubs are not the low ids
    cdf = torch.cumsum(w, 0)
    cdf = cdf / cdf[-1]
    E = int(N * avg_deg / 2 * 1.04)                                 # a few draws are lost to loops / duplicates
    u = torch.searchsorted(cdf, torch.rand(E, device=device, generator=gen, dtype=torch.float64)).clamp_(max=N - 1)
    v = torch.searchsorted(cdf, torch.rand(E, device=device, generator=gen, dtype=torch.float64)).clamp_(max=N - 1)
    keep = u != v
    u, v = u[keep], v[keep]
    key = torch.cat([u * N + v, v * N + u])                         # symmetrise
    key = torch.unique(key)                                         # sorted, duplicate edges removed
    row = torch.div(key, N, rounding_mode="floor")
    col = (key - row * N).to(torch.int32)
    counts = torch.bincount(row, minlength=N)
    nnz = int(key.numel())
    ptr_dtype = torch.int32 if nnz < 2**31 - 1 else torch.int64
    indptr = torch.zeros(N + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    return DeviceCSR(indptr.to(ptr_dtype), col, torch.device(device))


def community_graph(N, avg_deg, seed=4, exponent=2.5, block=2048, p_in=0.75, device="cuda", max_weight_frac=0.002):
    """A heavy-tailed undirected graph WITH id locality: the nodes form blocks of `block` consecutive ids (communities laid out
    contiguously, as the ids of a citation / co-author graph are by venue and year), every edge draws its first endpoint by
    expected-degree weight and its second from the SAME block with probability p_in (by weight inside the block), else from the
    whole graph.  Same degree law and density as powerlaw_graph; what differs is that a walk started in a block mostly stays
    near it -- the structure the Chung-Lu graphs lack and real graphs have (does a sorted work list cut L2 misses?)."""
    gen = torch.Generator(device=device).manual_seed(int(seed))
    ranks = torch.arange(1, N + 1, device=device, dtype=torch.float64)
    w = ranks.pow(-1.0 / (exponent - 1.0))
    w = torch.minimum(w, w.sum() * max_weight_frac / avg_deg)
    w = w[torch.randperm(N, device=device, generator=gen)]        # hubs are spread over the blocks
    cdf = torch.cumsum(w, 0)
    tot = cdf[-1].clone()
    E = int(N * avg_deg / 2 * 1.06)
    u = torch.searchsorted(cdf, torch.rand(E, device=device, generator=gen, dtype=torch.float64) * tot).clamp_(max=N - 1)
    blk_lo = (u // block) * block
    blk_hi = torch.clamp(blk_lo + block, max=N)
    c_lo = torch.where(blk_lo > 0, cdf[(blk_lo - 1).clamp(min=0)], torch.zeros_like(tot))
    c_hi = cdf[blk_hi - 1]
    local = torch.rand(E, device=device, generator=gen) < p_in
    r = torch.rand(E, device=device, generator=gen, dtype=torch.float64)
    target = torch.where(local, c_lo + r * (c_hi - c_lo), r * tot)
    v = torch.searchsorted(cdf, target).clamp_(max=N - 1)
    keep = u != v
    u, v = u[keep], v[keep]
    key = torch.unique(torch.cat([u * N + v, v * N + u]))
    row = torch.div(key, N, rounding_mode="floor")
    col = (key - row * N).to(torch.int32)
    counts = torch.bincount(row, minlength=N)
    indptr = torch.zeros(N + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    return DeviceCSR(indptr.to(torch.int32 if key.numel() < 2**31 - 1 else torch.int64), col, torch.device(device))


def _cumsum_f64_in_order(w):
    """cumsum of 4*10^7 doubles with ONE summation order.  The device's scan (decoupled look-back) combines its partial sums in an
    order that varies from run to run; in floating point that moves a few last bits of the CDF, and a handful of the billion
    edges drawn through it land on a neighbouring node: two processes built slightly different twitter-like graphs (found in round 6
    when two builds of the library disagreed on a digest that was the GRAPH's).  Sequentially on the host: 0.2 s."""
    return torch.cumsum(w.cpu(), 0).to(w.device)


def directed_powerlaw_graph(N, avg_deg, seed=3, exponent=2.2, device="cuda", chunk=1 << 28, max_weight_frac=0.0005):
    """Billion-edge stand-in for twitter-follower (41.65 M nodes, ~2.9 B adjacency entries after symmetrisation):
    every node gets max(1, d_i) out-neighbours drawn from a heavy-tailed popularity distribution, generated in
    chunks without any global sort (symmetrising 3 B edges would need one).  Not symmetric, not simple, rows
    unsorted -- none of which the walk kernel needs; there are no dead ends, use rng="philox".
    Row offsets are int64 whenever nnz >= 2^31."""
    gen = torch.Generator(device=device).manual_seed(int(seed))
    ranks = torch.arange(1, N + 1, device=device, dtype=torch.float64)
    w = ranks.pow(-1.0 / (exponent - 1.0))
    w = torch.minimum(w, w.sum() * max_weight_frac / avg_deg)
    w = w[torch.randperm(N, device=device, generator=gen)]
    deg = torch.clamp((w * (N * avg_deg / w.sum())).round().to(torch.int64), min=1)
    indptr = torch.zeros(N + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(deg, 0)
    nnz = int(indptr[-1].item())
    cdf = _cumsum_f64_in_order(w)
    cdf = (cdf / cdf[-1]).to(torch.float32)
    del ranks, w, deg
    indices = torch.empty(nnz, dtype=torch.int32, device=device)
    for lo in range(0, nnz, chunk):
        hi = min(lo + chunk, nnz)
        r = torch.rand(hi - lo, device=device, generator=gen, dtype=torch.float32)
        indices[lo:hi] = torch.searchsorted(cdf, r).clamp_(max=N - 1).to(torch.int32)
        del r
    if nnz < 2**31 - 1:
        indptr = indptr.to(torch.int32)
    return DeviceCSR(indptr, indices, torch.device(device))


def symmetric_powerlaw_graph_big(N, avg_deg, seed=3, exponent=2.2, device="cuda", row_chunks=16, max_weight_frac=0.0005):
    """The billion-edge graph as the reference's loader would hand it over (dataloader.py:122-135: G + G.T): undirected, SIMPLE,
    rows sorted, ~N*avg_deg adjacency entries (twitter-follower: 41.65 M nodes, ~1.47 B follows -> up to 2.94 B entries, int64 row
    offsets).  The directed edges are those of directed_powerlaw_graph at half the out-degree; symmetrising them is a sort of
    ~3 B 64-bit keys, done here as `row_chunks` independent sorts in HBM: chunk c owns the rows [r0, r1), takes the forward edges
    of its rows (a contiguous slice: the generator emits them row by row) and the reverse of every edge that POINTS into its
    rows (one masked pass over the edge list), and sorts + de-duplicates row*N + col -- ~2 * nnz / row_chunks keys at a time.
    Self loops are dropped.  Peak memory ~ 4 B (edge list) + 16 B * nnz / row_chunks (keys + sort) + the output."""
    gen = torch.Generator(device=device).manual_seed(int(seed))
    ranks = torch.arange(1, N + 1, device=device, dtype=torch.float64)
    w = ranks.pow(-1.0 / (exponent - 1.0))
    w = torch.minimum(w, w.sum() * max_weight_frac / avg_deg)
    w = w[torch.randperm(N, device=device, generator=gen)]
    odeg = torch.clamp((w * (N * avg_deg / 2 / w.sum())).round().to(torch.int64), min=1)       # out-degree: half the final degree
    optr = torch.zeros(N + 1, dtype=torch.int64, device=device)
    optr[1:] = torch.cumsum(odeg, 0)
    E = int(optr[-1].item())
    cdf = _cumsum_f64_in_order(w)
    cdf = (cdf / cdf[-1]).to(torch.float32)
    del ranks, w, odeg
    dst = torch.empty(E, dtype=torch.int32, device=device)
    step = 1 << 28
    for lo in range(0, E, step):
        hi = min(lo + step, E)
        r = torch.rand(hi - lo, device=device, generator=gen, dtype=torch.float32)
        dst[lo:hi] = torch.searchsorted(cdf, r).clamp_(max=N - 1).to(torch.int32)
        del r
    del cdf
    counts = torch.zeros(N, dtype=torch.int64, device=device)
    parts = []
    bounds = [int(round(c * N / row_chunks)) for c in range(row_chunks + 1)]
    for c in range(row_chunks):
        r0, r1 = bounds[c], bounds[c + 1]
        if r1 <= r0:
            continue
        e0, e1 = int(optr[r0].item()), int(optr[r1].item())
        src_f = torch.repeat_interleave(torch.arange(r0, r1, device=device, dtype=torch.int64), optr[r0 + 1:r1 + 1] - optr[r0:r1])
        key_f = src_f * N + dst[e0:e1].long()                        # forward: (u, v) with u in the chunk
        del src_f
        rev = []
        for lo in range(0, E, step):                                 # reverse: (v, u) for every edge u -> v with v in the chunk
            hi = min(lo + step, E)
            d = dst[lo:hi]
            idx = torch.nonzero((d >= r0) & (d < r1)).view(-1)
            if idx.numel():
                u = torch.searchsorted(optr, idx + lo, right=True) - 1
                rev.append(d[idx].long() * N + u)
            del d, idx
        key = torch.cat([key_f] + rev)
        del key_f, rev
        key = torch.unique(key)                                       # sorted by (row, col); duplicate edges fall together
        row = torch.div(key, N, rounding_mode="floor")
        col = key - row * N
        keep = row != col
        row, col = row[keep], col[keep].to(torch.int32)
        del key, keep
        counts[r0:r1] = torch.bincount(row - r0, minlength=r1 - r0)
        parts.append(col)
        del row
    indices = torch.cat(parts) if len(parts) > 1 else parts[0]
    del parts, dst
    indptr = torch.zeros(N + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    if indices.numel() < 2**31 - 1:
        indptr = indptr.to(torch.int32)
    return DeviceCSR(indptr, indices, torch.device(device))


def degree_ordered(csr):
    """The same graph with its nodes renumbered by descending degree (dev experiment: does a hub-first placement of the
    row pointers raise the walk's L2 hit rate?).  Returns (DeviceCSR, perm) with perm[new id] = old id; rows keep their
    neighbour order (renumbered, not re-sorted)."""
    ip = csr.indptr.long()
    N = csr.num_nodes
    deg = ip[1:] - ip[:-1]
    perm = torch.argsort(deg, descending=True, stable=True)
    rank = torch.empty_like(perm)
    rank[perm] = torch.arange(N, device=perm.device)
    ndeg = deg[perm]
    nip = torch.zeros(N + 1, dtype=torch.int64, device=perm.device)
    nip[1:] = torch.cumsum(ndeg, 0)
    row = torch.repeat_interleave(torch.arange(N, device=perm.device), ndeg)
    old_e = ip[perm][row] + (torch.arange(csr.nnz, device=perm.device) - nip[row])
    nidx = rank[csr.indices[old_e].long()].to(torch.int32)
    return DeviceCSR(nip.to(csr.indptr.dtype), nidx, csr.device), perm


def preset_graph(name, device="cuda", scale=1.0):
    if name == "cit2loc":     # the cit2-like graph with communities of consecutive ids (community_graph)
        return community_graph(max(int(2_927_963 * scale), 64), 20.7, seed=4, device=device)
    if name == "twitter":              # G + G.T of ~1.47 B follows, as dataloader.py:122-135 would hand it over (int64 row offsets)
        return symmetric_powerlaw_graph_big(max(int(41_652_230 * scale), 1000), 70.5, seed=3, device=device)
    if name == "twitter_directed":     # rounds 1-3's stand-in: directed, a multigraph, rows unsorted (no global sort needed)
        return directed_powerlaw_graph(max(int(41_652_230 * scale), 1000), 70.5, seed=3, device=device)
    p = PRESETS[name]
    return powerlaw_graph(max(int(p["N"] * scale), 16), p["avg_deg"], seed=p["seed"], device=device)


def query_pairs(csr, B, seed=7, device="cuda", pos_frac=0.5):
    """B query pairs -> int64 [2, B]: a fraction `pos_frac` 'positive-like' (drawn from the edges), the rest uniform
    random pairs -- the training mix of main.py:212-214 / dataloader.py:77-79 (positives + k uniform negatives per
    positive, shuffled into batches): pos_frac = 1/(k+1), e.g. 1/21 for ogbl-ppa's --k 20; SURVEY 8(d) uses 1/2."""
    gen = torch.Generator(device=device).manual_seed(int(seed))
    half = int(round(B * pos_frac))
    e = torch.randint(0, csr.nnz, (half,), device=device, generator=gen)
    src = torch.searchsorted(csr.indptr.long(), e, right=True) - 1
    dst = csr.indices[e].long()
    rnd = torch.randint(0, csr.num_nodes, (2, B - half), device=device, generator=gen)
    return torch.cat([torch.stack([src, dst]), rnd], dim=1).contiguous()


def ppr_like_spg(N, topk=100, seed=3, device="cuda"):
    """A float-payload SpG shaped like the citation2 PPR configuration: exactly min(topk, N) sorted distinct
    ids per row with scores in (0,1] (utils.py:36 maps PPR scores by (x+0.1)/(max+0.1))."""
    from .spg import SpG
    gen = torch.Generator(device=device).manual_seed(int(seed))
    k = min(topk, N)
    # distinct ids per row: a random start plus strictly increasing random gaps, wrapped and re-sorted
    gaps = torch.randint(1, max(N // k, 2), (N, k), device=device, generator=gen)
    start = torch.randint(0, N, (N, 1), device=device, generator=gen)
    ids = (start + torch.cumsum(gaps, 1) - gaps[:, :1]) % N
    ids, _ = torch.sort(ids, dim=1)
    data = (torch.rand((N, k), device=device, generator=gen, dtype=torch.float64) + 0.1) / 1.1
    indptr = torch.arange(0, (N + 1) * k, k, device=device, dtype=torch.int64)
    return SpG(indptr, ids.reshape(-1).to(torch.int32), data.reshape(-1), max_len=k, shape=(N, N))
