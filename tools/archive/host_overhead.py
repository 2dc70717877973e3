"""Dev-only: where does the step's host time go? (allocator traffic, syncs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import torch
import surel_plus_amd as sp
import bench
from surel_plus_amd.graphs import preset_graph, query_pairs
csr = preset_graph(sys.argv[1] if len(sys.argv) > 1 else "cit2")
B = 65536
edges = [query_pairs(csr, B, seed=s) for s in range(12)]
import cProfile, pstats
for s in range(12):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bench.hot_path_step(sp, csr, edges[s], 200, 4, s, "philox")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st1 = torch.cuda.memory_stats()
    print(f"step {s}: {1e3*(t1-t0):7.2f} ms  device_allocs +{st1['num_device_alloc']-st0['num_device_alloc']} frees +{st1['num_device_free']-st0['num_device_free']} "
          f"reserved {st1['reserved_bytes.all.current']/2**30:.2f} GiB retries {st1['num_alloc_retries']}")

pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for s in range(4):
    bench.hot_path_step(sp, csr, edges[s], 200, 4, 100 + s, "philox")
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
