"""Dev-only: the reference's offline stage (main.py:172-178: subg_matrix over all N nodes) -- reference C/OpenMP
sampler + scipy COO->CSR on the host versus the drop-ins of this repo, same graph, same parameters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SUBGACC_QUIET"] = "1"
import numpy as np, scipy.sparse as sps, torch
import oracle, surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph
from bench import quiet_stdout

name, M, k = (sys.argv[1] if len(sys.argv) > 1 else "collab"), 200, (3 if len(sys.argv) < 3 else int(sys.argv[2]))
csr = preset_graph(name)
N = csr.num_nodes
ptr_h, idx_h = csr.indptr.cpu().numpy(), csr.indices.cpu().numpy()
idx = np.arange(N)

class G:
    indptr, indices = ptr_h, idx_h

def sync_time(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); return time.perf_counter() - t0, out

z = enc = None
for _ in range(3):
    del z, enc          # the previous store goes back to the allocator first: steady state, no fresh GB-sized hipMalloc
    t_dev, (z, enc) = sync_time(lambda: sp.subg_matrix(csr, torch.arange(N, dtype=torch.int32, device="cuda"), M, k, rng="philox"))
t_host_in, (z2, enc2) = sync_time(lambda: sp.subg_matrix(G, idx, M, k))            # numpy CSR in (uploaded), rand_r mode
t_api, out = sync_time(lambda: sp.gset_sampler(ptr_h, idx_h, idx, num_walks=M, num_steps=k - 1))   # numpy in / numpy out
# join throughput from the resident SpG (J of SURVEY 8(d)): 20 batches of 65,536 pairs, output buffer re-used
from surel_plus_amd.graphs import query_pairs
tab = torch.from_numpy(enc).to("cuda").float() / M
edges = [query_pairs(csr, 65536, seed=s, device="cuda") for s in range(22)]
buf = torch.empty(2 * 65536 * z.max_len * 2 * tab.shape[1], dtype=torch.float32, device="cuda")
for e in edges[:2]:
    sp.gather(e, z, "cuda", ptr=True, encode=tab, out=buf, lazy=True)
t_join, _ = sync_time(lambda: [sp.gather(e, z, "cuda", ptr=True, encode=tab, out=buf, lazy=True) for e in edges[2:]])
J = 20 * 65536 / t_join
print(f"  join from the resident SpG: {J / 1e6:.1f} M pairs/s; amortised (sample all N once + 1e8 pairs): "
      f"{1e8 / (t_dev + 1e8 / J) / 1e6:.1f} M pairs/s")
# ... and from the same store re-keyed once (SpG.keyed: the payload is the LP key, no Z_SF gather per output row)
zk = None
for _ in range(3):
    del zk              # steady-state allocator, as for the sampler above
    t_key, zk = sync_time(lambda: z.keyed(enc, M))
for e in edges[:2]:
    sp.gather(e, zk, "cuda", ptr=True, encode=zk.slot_table(), out=buf, lazy=True)
t_kjoin, _ = sync_time(lambda: [sp.gather(e, zk, "cuda", ptr=True, encode=zk.slot_table(), out=buf, lazy=True) for e in edges[2:]])
JK = 20 * 65536 / t_kjoin
print(f"  join from the keyed store:  {JK / 1e6:.1f} M pairs/s (re-keying once: {t_key * 1e3:.1f} ms); amortised: "
      f"{1e8 / (t_dev + t_key + 1e8 / JK) / 1e6:.1f} M pairs/s")
print(f"{name}: N={N} nnz={csr.nnz} M={M} --num_steps {k}: set members {z.nnz}, distinct LP rows {enc.shape[0] - 1}")
print(f"  subg_matrix, graph resident (philox):        {t_dev:8.3f} s  = {N / t_dev / 1e6:7.2f} M roots/s")
print(f"  subg_matrix, numpy CSR in (rand_r, exact):   {t_host_in:8.3f} s")
print(f"  gset_sampler drop-in (numpy in, numpy out):  {t_api:8.3f} s")
ref = oracle.ref_module()
if ref is not None and ptr_h.dtype == np.int32 and os.environ.get('SKIP_REF', '0') != '1':
    for nt in (8, 16, -1):
        t0 = time.perf_counter()
        with quiet_stdout():
            nsize, remap, enc_r = ref.gset_sampler(ptr_h, idx_h, idx, num_walks=M, num_steps=k - 1, nthread=nt)
        t1 = time.perf_counter()
        zs = sps.csr_matrix((remap[1] + 1, (np.repeat(idx, nsize), remap[0])), (N, N))
        t2 = time.perf_counter()
        print(f"  reference gset_sampler nthread={nt:3d}: {t1 - t0:8.2f} s + scipy csr_matrix {t2 - t1:6.2f} s = {t2 - t0:8.2f} s")
