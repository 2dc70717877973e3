"""Top-K approximate-PPR node sets on the GPU: the offline stage of the PPR / SPD / DEG encoders.

Mirrors sampler/pprgo.py:85-111 (`topk_ppr_matrix`) and utils.py:35-36 (`encoding(x, adj, 'PPR')`).  The result is
an SpG with a float64 payload, which `gather` / `sjoin` consume as they consume the reference's scipy matrix
(train.py:39-43).  The graph is taken as unweighted (every stored entry counts 1), which is what the reference
feeds it (main.py:181-182: the training / inference adjacency).
"""
import math

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .sampler import DeviceCSR, _timed
from .spg import SpG

_MODES = {"row": 0, "sym": 1, "col": 2}
SLAB_BUDGET = 8 << 30        # bytes of HBM given to the per-wavefront tables of one launch
MAX_WAVES = 4096             # resident single-wave workgroups; the kernel saturates the memory system from ~4096 on (a module constant: set ppr.MAX_WAVES to experiment)
PILOT_ROOTS = 2048           # roots sampled to size the per-wavefront tables
LAST_STATS = None            # counters of the last ppr_topk call (dev / profiling)


def _as_csr(adj, device):
    if isinstance(adj, DeviceCSR):
        return adj
    if hasattr(adj, "indptr") and hasattr(adj, "indices"):
        return DeviceCSR(adj.indptr, adj.indices, device=device)
    raise TypeError("Input parsing error. (adjacency must be a DeviceCSR or expose indptr / indices)")


def _launch(csr, roots, alpha, eps, topk, table_log2, waves, cnt, ids, vals, flags, pushes):
    n = roots.numel()
    waves = max(1, min(waves, n))
    slab = torch.empty(lib().subgacc_ppr_slab_bytes(table_log2, waves), dtype=torch.uint8, device=roots.device)
    check(lib().subgacc_ppr_slab_reset(ptr(slab), table_log2, waves, stream_ptr()))
    with _timed("ppr_push"):
        check(lib().subgacc_ppr_topk(ptr(csr.indptr), int(csr.indptr64), ptr(csr.indices), csr.num_nodes, ptr(roots), n,
                                     float(alpha), float(eps), int(topk), ptr(slab), table_log2, waves, ptr(cnt), ptr(ids),
                                     ptr(vals), ptr(flags), ptr(pushes) if pushes is not None else None, stream_ptr()))


def _pilot_table_log2(csr, roots, alpha, epsilon, topk, log2_max):
    """smallest table (log2 slots) that holds >= 97 % of a sample of the roots"""
    n = roots.numel()
    m = min(n, PILOT_ROOTS)
    if m == 0:
        return 10
    sample = roots[:: max(1, n // m)][:m].contiguous()
    dev = roots.device
    cnt = torch.zeros(m, dtype=torch.int32, device=dev)
    ids = torch.empty(m * topk, dtype=torch.int32, device=dev)
    vals = torch.empty(m * topk, dtype=torch.float32, device=dev)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    for log2 in range(10, log2_max):
        _launch(csr, sample, alpha, epsilon, topk, log2, MAX_WAVES, cnt, ids, vals, flags, None)
        if int((cnt >= 0).sum()) >= 0.97 * m:
            return log2
    return log2_max


def ppr_topk(adj, alpha, epsilon, nodes, topk, table_log2=None, device=None):
    """pprgo.py:53-82 -> (row_off int64[n+1], ids int32[X] ascending per row, vals float32[X], pushes)."""
    device = device or _lib.require_device()
    csr = _as_csr(adj, device)
    roots = torch.as_tensor(nodes).to(device=device, dtype=torch.int32).contiguous()
    if not (0.0 < float(alpha) <= 1.0) or not float(epsilon) > 0.0 or int(topk) < 1:
        raise TypeError("Input parsing error. (alpha in (0,1], epsilon > 0, topk >= 1)")
    if roots.numel() and (int(roots.min()) < 0 or int(roots.max()) >= csr.num_nodes):
        raise IndexError("ppr_topk: node id out of range")
    n = roots.numel()
    cnt = torch.zeros(n, dtype=torch.int32, device=device)
    ids = torch.empty(n * topk, dtype=torch.int32, device=device)
    vals = torch.empty(n * topk, dtype=torch.float32, device=device)
    flags = torch.zeros(4, dtype=torch.int32, device=device)
    pushes = torch.zeros(2, dtype=torch.int64, device=device)      # pushes, touched nodes
    # a push of u moves >= alpha*eps*deg(u) into p and sum(p) <= 1: at most 1/(alpha*eps) nodes are ever touched.
    # Most roots need far less and dense tables are faster (more of them stay in L2), so a pilot over a sample of
    # the roots picks the smallest table that holds ~all of them; the few roots that overflow are run again, larger.
    worst = int(math.ceil(1.0 / (float(alpha) * float(epsilon)))) + 512
    log2_max = min(26, max(10, int(math.ceil(math.log2(2 * worst)))))
    if table_log2 is None:
        log2 = _pilot_table_log2(csr, roots, alpha, epsilon, topk, log2_max)
    else:
        log2 = min(log2_max, max(10, int(table_log2)))
    todo, dst = roots, None
    while n:
        waves = int(min(MAX_WAVES, max(64, SLAB_BUDGET // (24 << log2))))
        if dst is None:
            _launch(csr, todo, alpha, epsilon, topk, log2, waves, cnt, ids, vals, flags, pushes)
        else:   # later rounds: the overflowed roots only, scattered back into their rows
            c2 = torch.zeros(todo.numel(), dtype=torch.int32, device=device)
            i2 = torch.empty(todo.numel() * topk, dtype=torch.int32, device=device)
            v2 = torch.empty(todo.numel() * topk, dtype=torch.float32, device=device)
            _launch(csr, todo, alpha, epsilon, topk, log2, waves, c2, i2, v2, flags, pushes)
            cnt[dst] = c2
            ids.view(n, topk)[dst] = i2.view(-1, topk)
            vals.view(n, topk)[dst] = v2.view(-1, topk)
        bad = (cnt < 0).nonzero().flatten()
        if bad.numel() == 0:
            break
        if log2 >= log2_max:
            raise MemoryError("ppr_topk: a root touched more nodes than 1/(alpha*eps) allows -- repeated entries in a CSR row?")
        log2 = min(log2 + 2, log2_max)
        todo, dst = roots[bad].contiguous(), bad
        flags.zero_()
    # strided rows -> packed rows (the copy kernel of the fused walk form; the float32 payload travels as its bits)
    row_off = torch.empty(n + 1, dtype=torch.int64, device=device)
    ws = torch.empty(max(lib().subgacc_scan_workspace_bytes(n), 8), dtype=torch.uint8, device=device)
    check(lib().subgacc_exclusive_scan_i32(ptr(cnt), n, ptr(row_off), ptr(ws), ws.numel(), stream_ptr()))
    X = int(row_off[-1].item()) if n else 0
    out_ids = torch.empty(X, dtype=torch.int32, device=device)
    out_vals = torch.empty(X, dtype=torch.float32, device=device)
    if X:
        check(lib().subgacc_compact_rows(ptr(ids), ptr(vals.view(torch.int32)), ptr(cnt), ptr(row_off), n, int(topk),
                                         ptr(out_ids), ptr(out_vals.view(torch.int32)), None, 0, stream_ptr()))
    global LAST_STATS
    st = pushes.tolist()
    LAST_STATS = {"pushes": st[0], "touched": st[1], "roots": n}
    return row_off, out_ids, out_vals, st[0]


def topk_ppr_matrix(adj_matrix, alpha, eps, idx, topk, normalization="row", device=None, encode=False):
    """Drop-in for sampler/pprgo.py:85-111: SpG (float64 payload) whose row i holds the top-`topk` PPR entries of idx[i].
    encode=True also applies utils.py:35-36 (`encoding(x, adj, 'PPR')`) in the same pass over the payload."""
    if normalization not in _MODES:
        raise ValueError(f"Unknown PPR normalization: {normalization}")
    device = device or _lib.require_device()
    csr = _as_csr(adj_matrix, device)
    roots = torch.as_tensor(idx).to(device=device, dtype=torch.int32).contiguous()
    row_off, ids, vals, _ = ppr_topk(csr, alpha, eps, roots, topk, device=device)
    n, X = roots.numel(), ids.numel()
    data = torch.empty(X, dtype=torch.float64, device=device)
    mx = torch.zeros(1, dtype=torch.int64, device=device)
    check(lib().subgacc_ppr_normalize(ptr(csr.indptr), int(csr.indptr64), ptr(roots), n, ptr(row_off), X, ptr(ids),
                                      ptr(vals), _MODES[normalization], ptr(data), ptr(mx) if encode else None,
                                      stream_ptr()))
    if encode and X:
        check(lib().subgacc_ppr_encode(ptr(data), X, ptr(row_off[n:]), ptr(mx), stream_ptr()))
    return SpG(row_off, ids, data, max_len=int(topk), shape=(n, csr.num_nodes))


def encoding(x, adj=None, encoding="PPR", device=None):
    """Drop-in for utils.py:22-38: (x', agg).

    'PPR'  x.data = (x.data + 0.1) / (x.data.max() + 0.1), in place; agg None.
    'DEG'  pattern x u adj, value(i,j) = log(nnz(row j of x u adj) + 1); agg = x + l1-normalised adj (an SpG, where the
           reference returns the scipy matrix).
    'SPD'  pattern adj u x u diagonal, value = 1[adj] + 0.5[x and a 2-path] + 0.3[x], diagonal 2.3; agg None.
    x: SpG with one row per node (float64 payload, e.g. topk_ppr_matrix over all nodes); adj: the unweighted,
    symmetric adjacency (scipy CSR or DeviceCSR)."""
    if encoding == "PPR":
        X = x.nnz
        if X:
            mx = x.data[:X].max().view(1).view(torch.int64)     # bit pattern of a non-negative double
            nnz = torch.tensor([X], dtype=torch.int64, device=x.device)
            check(lib().subgacc_ppr_encode(ptr(x.data), X, ptr(nnz), ptr(mx), stream_ptr()))
        return x, None
    if encoding not in ("DEG", "SPD"):
        raise NotImplementedError
    import numpy as np
    dev = x.device
    csr = _as_csr(adj, dev)
    n = x.n_rows
    if n != csr.num_nodes or x.data.dtype != torch.float64:
        raise ValueError("encoding: x must hold one float64 row per node of adj")
    mode = 1 if encoding == "DEG" else 2
    st = stream_ptr()
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    row_len = torch.empty(n, dtype=torch.int32, device=dev)
    check(lib().subgacc_encode_sizes(ptr(x.indptr), ptr(x.indices), n, ptr(csr.indptr), int(csr.indptr64), ptr(csr.indices),
                                     mode, ptr(row_len), ptr(flags), st))
    out_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    ws = torch.empty(max(lib().subgacc_scan_workspace_bytes(n), 8), dtype=torch.uint8, device=dev)
    check(lib().subgacc_exclusive_scan_i32(ptr(row_len), n, ptr(out_off), ptr(ws), ws.numel(), st))
    total = int(out_off[-1].item()) if n else 0
    max_len = int(row_len.max().item()) if n else 0
    ids = torch.empty(total, dtype=torch.int32, device=dev)
    val = torch.empty(total, dtype=torch.float64, device=dev)
    agg = torch.empty(total, dtype=torch.float64, device=dev) if mode == 1 else None
    # np.log on the host: the same libm the reference's NumPy calls, so the values match it bit for bit
    logt = torch.from_numpy(np.log(np.arange(max_len + 1, dtype=np.int64) + 1)).to(dev) if mode == 1 else None
    check(lib().subgacc_encode_fill(ptr(x.indptr), ptr(x.indices), ptr(x.data), n, int(x.max_len), ptr(csr.indptr),
                                    int(csr.indptr64), ptr(csr.indices), mode, ptr(row_len) if mode == 1 else None,
                                    ptr(logt), max_len + 1 if mode == 1 else 0, ptr(out_off), ptr(ids), ptr(val), ptr(agg),
                                    ptr(flags), st))
    shape = (n, csr.num_nodes)
    z = SpG(out_off, ids, val, max_len=max_len, shape=shape)
    return z, (SpG(out_off, ids, agg, max_len=max_len, shape=shape) if mode == 1 else None)
