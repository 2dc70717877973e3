"""Dev-only: replays of a captured B-pair step for `rocprofv3 --kernel-trace --stats` (which kernels make up a small batch):
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/small_batch_trace.py [B] [replays]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import torch
import surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph, query_pairs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 50
csr = preset_graph("cit2")
step = sp.CapturedStep(csr, B, num_walks=200, num_steps=3, seed=1, rng="philox")
edges = [query_pairs(csr, B, seed=s) for s in range(8)]
torch.cuda.synchronize()
for s in range(R):
    step(edges[s % 8]).finish()
torch.cuda.synchronize()
print("done", B, R)
