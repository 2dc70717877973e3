"""Dev-only: the PPR offline stage (main.py:181-182: topk_ppr_matrix over all nodes, 'sym', + encoding 'PPR') on the
GPU versus the sequential C port of the oracle on a sample of roots (the reference itself is numba code; numba is
not in the image).  usage: ppr_stage.py [collab|ppa|cit2] [n_roots|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import numpy as np, torch
from oracle import oracle as orc
from surel_plus_amd import ppr, sampler
from surel_plus_amd.graphs import preset_graph

name = sys.argv[1] if len(sys.argv) > 1 else "collab"
alpha = {"collab": 0.7, "ppa": 0.5, "cit2": 0.1}[name]          # main.py:101-111
eps, topk = 1e-4, 100                                            # main.py:44-47
csr = preset_graph(name)
N = csr.num_nodes
n = N if len(sys.argv) < 3 or sys.argv[2] == "all" else min(N, int(sys.argv[2]))
roots = torch.arange(n, dtype=torch.int32, device="cuda")

class Timer:
    def __init__(self): self.t = {}
    def __call__(self, name):
        t = self
        class _C:
            def __enter__(s): s.a, s.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); s.a.record()
            def __exit__(s, *e): s.b.record(); t.t.setdefault(name, []).append((s.a, s.b))
        return _C()
    def ms(self): torch.cuda.synchronize(); return {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.t.items()}

ppr.ppr_topk(csr, alpha, eps, roots[:4096], topk)                # warm-up
for rep in range(2):
    sampler.KERNEL_TIMER = tm = Timer()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z = ppr.topk_ppr_matrix(csr, alpha, eps, roots, topk, normalization="sym", encode=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sampler.KERNEL_TIMER = None
_, _, _, pushes = ppr.ppr_topk(csr, alpha, eps, roots, topk)
print(f"{name}: N={N} nnz={csr.nnz} alpha={alpha} eps={eps} topk={topk}; {n} roots -> {z.nnz} entries, {pushes} pushes "
      f"({pushes / n:.1f} per root, {ppr.LAST_STATS['touched'] / n:.0f} touched nodes per root)")
print(f"  GPU stage: {dt:.3f} s = {n / dt / 1e3:.1f} k roots/s, {pushes / dt / 1e6:.1f} M pushes/s; kernels: {tm.ms()}")
ns = min(n, 2000)
ptr_h, idx_h = csr.indptr[: ].cpu().numpy(), csr.indices.cpu().numpy()
t0 = time.perf_counter()
off, ids, data = orc.topk_ppr_matrix(ptr_h, idx_h, alpha, eps, np.arange(ns, dtype=np.int32), topk, "sym", table_log2=18)
dc = time.perf_counter() - t0
print(f"  oracle C port, 1 thread, first {ns} roots: {dc:.2f} s = {ns / dc:.0f} roots/s -> GPU/CPU-thread = {n / dt / (ns / dc):.0f}x")
