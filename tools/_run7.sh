cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys
os.environ["SUBGACC_QUIET"]="1"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch, surel_plus_amd as sp
from test_gpu_parity import sym_graph, _reference_style_attn
ptr_, idx = sym_graph(3000, 15000, seed=6, hubs=1)
z, sets = sp.sample_spg(sp.DeviceCSR(ptr_, idx), np.arange(3000), num_walks=64, num_steps=3, seed=2, rng="philox")
table = sets.feature_table()
edge = torch.from_numpy(np.random.default_rng(4).integers(0, 3000, (2, 512))).cuda()
def nets(dt):
    torch.manual_seed(7)
    return [m.to(dt) for m in (torch.nn.Sequential(torch.nn.Linear(4, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16)).cuda(),
            torch.nn.Sequential(torch.nn.Linear(16, 1)).cuda(),
            torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.ReLU()).cuda())]
torch.manual_seed(1)
w = torch.randn(2, 512, 16, device="cuda")
a = nets(torch.float32); b = nets(torch.float32); c = nets(torch.float64)
fused = sp.attn_stage(edge, z, table, *a); (fused * w).sum().backward()
xz, ind = sp.gather(edge, z, "cuda", ptr=True, encode=table)
ref = _reference_style_attn(xz, ind, *b).view(2, -1, 16); (ref * w).sum().backward()
class D:  # float64 truth
    pass
def ref64(xz, ind, mlp, gate, val):
    S = ind.numel() - 1
    x = mlp(xz).sum(dim=-2)
    seg = torch.repeat_interleave(torch.arange(S, device=xz.device), ind[1:] - ind[:-1])
    g = gate(x).reshape(-1)
    gmax = torch.full((S,), float("-inf"), device=g.device, dtype=g.dtype).scatter_reduce(0, seg, g.detach(), "amax")
    wgt = torch.exp(g - gmax[seg])
    den = torch.zeros(S, device=g.device, dtype=g.dtype).index_add_(0, seg, wgt)
    alpha = wgt / (den[seg] + 1e-16)
    return torch.zeros((S, x.shape[-1]), device=g.device, dtype=g.dtype).index_add_(0, seg, alpha[:, None] * val(x))
t = ref64(xz.double(), ind, *c).view(2, -1, 16); (t * w.double()).sum().backward()
print("fwd  fused-vs-64", float((fused.double()-t).abs().max()), "ref32-vs-64", float((ref.double()-t).abs().max()), "scale", float(t.abs().max()))
for (na, pa), (nb, pb), (nc, pc) in zip(*[[(n, p) for m in ms for n, p in m.named_parameters()] for ms in (a, b, c)]):
    s = float(pc.grad.abs().max())
    print(na, tuple(pa.shape), "fused", float((pa.grad.double()-pc.grad).abs().max())/s, "ref32", float((pb.grad.double()-pc.grad).abs().max())/s)
PY
