"""Dev-only: per-phase cycle shares of the fused walk kernel (needs a -DSG_EXPERIMENT=7 build of walk.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import torch
from surel_plus_amd import _lib
from surel_plus_amd._lib import check, lib, ptr, stream_ptr
from surel_plus_amd.graphs import preset_graph, query_pairs
from surel_plus_amd.sampler import make_cfg
L = lib()
csr = preset_graph(sys.argv[1] if len(sys.argv) > 1 else "cit2")
M, m = 200, int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = 65536
roots = query_pairs(csr, B, seed=1).reshape(-1).to(torch.int32)
n = roots.numel(); stride = M * m + 1
cfg = make_cfg(csr, M, m, rng="philox")
flags = torch.zeros(64, dtype=torch.int32, device="cuda")
table = torch.empty(L.subgacc_uniq_table_bytes(1 << 20), dtype=torch.uint8, device="cuda")
check(L.subgacc_uniq_reset(ptr(table), 1 << 20, stream_ptr()))
ids = torch.empty(n * stride, dtype=torch.int32, device="cuda"); slot = torch.empty_like(ids)
nsize = torch.empty(n, dtype=torch.int32, device="cuda")
for it in range(3):
    flags.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    check(L.subgacc_walk_spg(cfg, ptr(csr.indptr), ptr(csr.indices), csr.num_nodes, ptr(roots), n, 0, None, None, ptr(table),
                             1 << 20, ptr(ids), ptr(slot), ptr(nsize), ptr(flags), stream_ptr()))
    b.record()
torch.cuda.synchronize()
print(f"kernel {a.elapsed_time(b):.3f} ms (stamps on 1 workgroup in 64), LDS pad {os.environ.get('LDS_PAD', '0')} (a build flag of tools/walk_phases.sh)")
c = flags[8:8 + 18].view(torch.int64).tolist()
n = (n + 63) // 64          # sampled workgroups
names = ["init tables + fisher-yates draws", "root insert", "WALK + visits", "member load + fold (LDS)", "flush fold to HBM + zero buckets",
         "histogram", "bucket scan (wave 0)", "scatter", "in-bucket rank + global write"]
tot = sum(c)
for nm, v in zip(names, c):
    print(f"{nm:36s} {v / n:9.0f} cycles/root  {100 * v / tot:5.1f} %")
print("total", tot / n, "cycles/root")
