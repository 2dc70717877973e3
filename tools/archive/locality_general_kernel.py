"""Dev-only: the fused-row launch of a shape walk_rows_kernel does NOT take (M = 300 walks, 2 hops -> walk_sets_kernel<SPG>) over the
graph with id locality and over the structureless one: what the second sort level of its epilogue is worth (VERDICT r3 item 9;
A/B against an older build with tools/ab_lib.sh-style SUBGACC_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.environ["SUBGACC_QUIET"] = "1"
import torch, surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph
from surel_plus_amd.spg import sample_spg
for name in ("cit2loc", "cit2"):
    csr = preset_graph(name)
    idx = torch.arange(400000, dtype=torch.int32, device="cuda")
    for M, m in ((300, 2), (260, 3)):
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            z, sets = sample_spg(csr, idx, num_walks=M, num_steps=m, rng="philox", fused=True)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{name} 400k roots M={M} m={m} lib={os.path.basename(os.environ.get('SUBGACC_LIB', 'shipped'))}: {dt * 1e3:.2f} ms  ({sets.X / 400000:.0f} members/root)", flush=True)
    del csr
