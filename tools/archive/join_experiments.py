"""Dev-only: time SpJoin variants on one cit2-like batch (events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import torch
import surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph, query_pairs
from surel_plus_amd.sampler import sample_sets
from surel_plus_amd.spjoin import sjoin

csr = preset_graph(sys.argv[1] if len(sys.argv) > 1 else "cit2")
B = 65536
edge = query_pairs(csr, B, seed=1)
sets = sample_sets(csr, edge.reshape(-1).to(torch.int32), num_walks=200, num_steps=3, rng="philox")
z = sp.SpG.from_sets(sets)
table = sets.feature_table()
rows = torch.arange(2 * B, device="cuda").view(2, B)
own, partner = torch.cat([rows[0], rows[1]]), torch.cat([rows[1], rows[0]])


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

R = z.nnz
print("rows", R, "k", table.shape[1])
print("paired xz      ms", timeit(lambda: sjoin(z, own, partner, table, pair_block=B)))
print("generic xz     ms", timeit(lambda: sjoin(z, own, partner, table, pair_block=0)))
print("paired index   ms", timeit(lambda: sjoin(z, own, partner, None, return_index=True, pair_block=B)))
print("paired k=1 tab ms", timeit(lambda: sjoin(z, own, partner, table[:, :1].contiguous(), pair_block=B)))
print("paired k=8 tab ms", timeit(lambda: sjoin(z, own, partner, torch.cat([table, table], 1).contiguous(), pair_block=B)))
x = torch.empty(R * 8, device="cuda"); 
print("memset 1.6GB   ms", timeit(lambda: x.zero_()))
y = torch.empty(R * 8, device="cuda")
print("copy 1.6GB     ms", timeit(lambda: y.copy_(x)))

# ---- count form: first model stage of the reference vs counts @ MLP(table)
H = 256
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(table.shape[1], H), torch.nn.ReLU(), torch.nn.Linear(H, H)).cuda()
print("gather_counts  ms", timeit(lambda: sp.gather_counts(rows, z, table.shape[0])))
C, sizes = sp.gather_counts(rows, z, table.shape[0])
print("C shape", tuple(C.shape), "MB", C.numel() * 4 / 1e6)
with torch.no_grad():
    print("counts@MLP(tab) ms", timeit(lambda: (C @ mlp(table)) / sizes[:, None].clamp(min=1)))
    Bs = 4096                                   # the reference-style stage does not fit memory at B = 65536
    rs = torch.arange(2 * Bs, device="cuda").view(2, Bs)
    xz, ind = sp.gather(rs, z, "cuda", ptr=True, encode=table)
    segid = torch.repeat_interleave(torch.arange(2 * Bs, device="cuda"), ind[1:] - ind[:-1])
    def ref_stage():
        x = mlp(xz).sum(dim=-2)
        return torch.zeros(2 * Bs, H, device="cuda").index_add_(0, segid, x) / (ind[1:] - ind[:-1])[:, None].clamp(min=1)
    t = timeit(ref_stage, n=3)
    print(f"reference-style stage at B={Bs}: {t:.3f} ms  (x{B // Bs} for B={B}: {t * B / Bs:.1f} ms), xz rows {xz.shape[0]}")
    Cs, ss = sp.gather_counts(rs, z, table.shape[0])
    a = (Cs @ mlp(table)) / ss[:, None].clamp(min=1)
    b = ref_stage()
    print("max rel err", float(((a - b).abs() / (b.abs() + 1e-3)).max()))
