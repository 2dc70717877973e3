import sys, torch
sys.path.insert(0, "/root/repo")
from surel_plus_amd.graphs import preset_graph, query_pairs
for name in ("cit2", "cit2loc"):
    csr = preset_graph(name)
    deg = (csr.indptr[1:] - csr.indptr[:-1]).long()
    e = query_pairs(csr, 65536, seed=1000)
    rd = deg[e.reshape(-1)]
    print(name, "nnz", csr.nnz, "max deg", int(deg.max()), "deg>200 nodes", int((deg > 200).sum()), "batch roots deg>200", int((rd > 200).sum()),
          "mean root deg", float(rd.float().mean()), "distinct roots", int(torch.unique(e).numel()),
          "top-5 degs", deg.topk(5).values.tolist(), "p99 deg", int(deg.float().quantile(0.99)) if deg.numel() < 16e6 else None)
