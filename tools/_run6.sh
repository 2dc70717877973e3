cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "pairs or attn or counts or mean_stage or gather" 2>&1 | tail -8
python - <<'PY'
import os, sys, time
os.environ["SUBGACC_QUIET"]="1"
sys.path.insert(0, os.getcwd())
import torch, surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph, query_pairs
csr = preset_graph("cit2")
B = 65536
e = query_pairs(csr, B, seed=1)
roots = e.reshape(-1).to(torch.int32)
z, sets = sp.sample_spg(csr, roots, num_walks=200, num_steps=3, seed=1, rng="philox")
rows = torch.arange(2*B, device="cuda").view(2, B)
table = sets.feature_table()
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): r=f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3, r
ms, (xz, ind) = t(lambda: sp.gather(rows, z, None, ptr=True, encode=table)); print("gather xz", ms, xz.shape)
ms, (pairs, mult, ptr_) = t(lambda: sp.gather_pairs(rows, z)); print("gather_pairs", ms, pairs.shape, "ratio", xz.shape[0]/pairs.shape[0])
ms, (C, sz) = t(lambda: sp.gather_counts(rows, z, table.shape[0])); print("gather_counts", ms, C.shape)
mlp = torch.nn.Sequential(torch.nn.Linear(4, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256)).cuda()
gate = torch.nn.Linear(256, 1).cuda(); val = torch.nn.Sequential(torch.nn.Linear(256,256), torch.nn.ReLU()).cuda()
with torch.no_grad():
    ms, out = t(lambda: sp.attn_stage(rows, z, table, mlp, gate, val)); print("attn_stage H=256 fwd", ms, out.shape)
    ms, out = t(lambda: sp.mean_stage(rows, z, table, mlp)); print("mean_stage H=256 fwd", ms, out.shape)
PY
