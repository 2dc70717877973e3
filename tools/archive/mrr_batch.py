"""Dev-only: an evaluation-style batch (utils.py:92-95: every source against ~1,000 candidates) through the buffered on-demand
step, with and without root dedup.   python tools/mrr_batch.py [workload] [sources] [targets]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import torch
import surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph
from bench import WORKLOADS

name = sys.argv[1] if len(sys.argv) > 1 else "cit2"
S, T = (int(sys.argv[2]) if len(sys.argv) > 2 else 64), (int(sys.argv[3]) if len(sys.argv) > 3 else 1001)
csr = preset_graph(name)
M, hops = 200, (2 if name in ("collab", "twitter") else 3)
B = S * T
gen = torch.Generator(device="cuda").manual_seed(5)
batches = []
for s in range(6):
    src = torch.randint(0, csr.num_nodes, (S,), device="cuda", generator=gen).repeat_interleave(T)
    dst = torch.randint(0, csr.num_nodes, (B,), device="cuda", generator=gen)
    batches.append(torch.stack([src, dst]))
res = {}
for dd in (False, True):
    bufs = [sp.StepBuffers(csr, B, num_walks=M, num_steps=hops, dedup_roots=dd) for _ in range(2)]
    outs = []
    def run(n):
        for i in range(n):
            xz, ind, sets = sp.sample_and_gather(csr, batches[i % len(batches)], num_walks=M, num_steps=hops, seed=1, rng="philox",
                                                 buffers=bufs[i & 1], dedup_roots=dd)
            sets.prefetch()
            outs.append(sets)
            if len(outs) > 1:
                outs.pop(0).resolve()
        outs.pop(0).resolve()
    run(4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(12)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 12
    res[dd] = (B / dt / 1e6, dt * 1e3)
    del bufs
print(f"{name}: {S} sources x {T} targets = {B} pairs/step, M={M}, {hops} hops: every endpoint sampled {res[False][0]:.1f} M pairs/s "
      f"({res[False][1]:.3f} ms), distinct endpoints only {res[True][0]:.1f} M pairs/s ({res[True][1]:.3f} ms)")
