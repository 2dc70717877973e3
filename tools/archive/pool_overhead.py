"""Dev-only: host time of the small-batch serving loop (CapturedStepPool.submit / finish), with a cProfile of it."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import torch
import surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph, query_pairs

csr = preset_graph("cit2")
SYNC = not (len(sys.argv) > 2 and sys.argv[2] == "nosync")
B, lanes, steps = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 3000
pool = sp.CapturedStepPool(csr, B, lanes=lanes, num_walks=200, num_steps=3)
edges = [query_pairs(csr, B, seed=s, device="cuda") for s in range(16)]


def loop(n):
    q = []
    for s in range(n):
        if len(q) == lanes:
            pool.finish(q.pop(0))
        q.append(pool.submit(edges[s % 16], sync=SYNC))
    while q:
        pool.finish(q.pop(0))


loop(200)
torch.cuda.synchronize(); t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{lanes} lanes, sync={SYNC}: {dt / steps * 1e6:.1f} us per step = {B * steps / dt / 1e6:.2f} M pairs/s")
pr = cProfile.Profile(); pr.enable(); loop(steps); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
