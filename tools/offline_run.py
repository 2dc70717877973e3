"""Dev-only: the offline stage alone (subg_matrix over all N nodes, main.py:172-178), a few times, for a kernel trace:
    rocprofv3 --kernel-trace --stats -d gpurun_out/x -o off -- python3 tools/offline_run.py cit2 4"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import torch
import surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph

name, k = (sys.argv[1] if len(sys.argv) > 1 else "cit2"), (int(sys.argv[2]) if len(sys.argv) > 2 else 4)
rng = sys.argv[3] if len(sys.argv) > 3 else "philox"
csr = preset_graph(name)
idx = torch.arange(csr.num_nodes, dtype=torch.int32, device="cuda")
z = enc = None
for it in range(4):
    del z, enc
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z, enc = sp.subg_matrix(csr, idx, 200, k, rng=rng)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name} all-N subg_matrix ({rng}) pass {it}: {dt * 1e3:.2f} ms, {csr.num_nodes / dt / 1e6:.1f} M roots/s, members {z.nnz}, rows {enc.shape[0] - 1}", flush=True)
