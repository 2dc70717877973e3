import os, sys, time
sys.path.insert(0, "/root/repo"); os.environ["SUBGACC_QUIET"] = "1"
import torch, surel_plus_amd as sp
from surel_plus_amd.graphs import preset_graph
from surel_plus_amd.spg import sample_spg
csr = preset_graph("cit2loc")
idx = torch.arange(600000, dtype=torch.int32, device="cuda")
for k in (4, 5):
    for batched in (True, False):
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            z, sets = sample_spg(csr, idx, num_walks=200, num_steps=k - 1, rng="philox", batched_registration=batched)
            enc = sets.enc_int16()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"cit2loc 600k roots num_steps={k} batched_reg={batched}: {dt*1e3:.2f} ms", flush=True)
