#!/usr/bin/env python3
"""Dev-only: a digest of one on-demand step (rows sampled by the fused walk kernel, joined) on a preset graph -- two builds of the
library (SUBGACC_LIB) must print the same line:   rows_digest.py WORKLOAD [PAIRS]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import surel_plus_amd as sp  # noqa: E402
from surel_plus_amd.graphs import preset_graph, query_pairs  # noqa: E402

wl = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
preset, M, k, _, pos = bench.WORKLOADS[wl]
csr = preset_graph(preset, device="cuda")
e = query_pairs(csr, B, seed=5, device="cuda", pos_frac=pos)
xz, ind, sets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=k - 1, seed=3, rng="philox")
torch.cuda.synchronize()
h = hashlib.sha256()
h.update(ind.cpu().numpy().tobytes())
h.update(xz.cpu().numpy().tobytes())
print(wl, "rows", int(ind[-1]), "digest", h.hexdigest()[:16], "lib", os.path.basename(os.environ.get("SUBGACC_LIB", "shipped")))
