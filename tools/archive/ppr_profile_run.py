"""Dev-only: one PPR launch for rocprofv3 (cit2-like graph, first 400k roots)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import torch
from surel_plus_amd import ppr
from surel_plus_amd.graphs import preset_graph
name = sys.argv[1] if len(sys.argv) > 1 else "cit2"
alpha = {"collab": 0.7, "ppa": 0.5, "cit2": 0.1}[name]
csr = preset_graph(name)
n = min(csr.num_nodes, 400000)
roots = torch.arange(n, dtype=torch.int32, device="cuda")
ppr.ppr_topk(csr, alpha, 1e-4, roots, 100)
torch.cuda.synchronize()
print(ppr.LAST_STATS)
