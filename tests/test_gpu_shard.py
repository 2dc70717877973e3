"""GPU: the multi-rank path on the PRODUCT kernels.  Two ranks are started as fresh processes (each initialises the
GPU itself; both use cuda:0, gloo carries the one exchange step) and run shard.sample_sets_sharded over
shard.hip_sampler, shard.sample_spg_sharded and the pair-sharded join; their merged results must be bit-identical
to the single-process HIP result and to the oracle (subg_acc.c:745: independent roots; :957-978: one global
numbering pass), in both RNG modes -- the sequential rand_r stream is entered mid-way by the second rank."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "workers", "shard_rank.py")


def _run_ranks(tmp_path, world, rng, fused):
    port = str(23000 + os.getpid() % 4000 + (7 if fused else 0) + (13 if rng == "philox" else 0))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), port, str(tmp_path), rng, "1" if fused else "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    return [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]


@pytest.mark.parametrize("rng", ["rand_r", "philox"])
@pytest.mark.parametrize("fused", [False, True])
def test_two_ranks_equal_one_process_and_the_oracle(tmp_path, rng, fused):
    import surel_plus_amd as sp
    from surel_plus_amd import shard
    sys.path.insert(0, os.path.join(ROOT, "tests", "workers"))
    from shard_rank import problem
    indptr, indices, roots, edge, M, m = problem()
    parts = _run_ranks(tmp_path, 2, rng, fused)

    # single process, product kernels
    csr = sp.DeviceCSR(indptr, indices)
    z1, sets1 = sp.sample_spg(csr, roots, num_walks=M, num_steps=m, seed=77, rng=rng, fused=fused)
    plain = sp.sample_sets(csr, roots, num_walks=M, num_steps=m, seed=77, rng=rng)
    # the oracle on the whole query
    nsize, remap, enc = oracle.gset_sampler(indptr, indices, roots, num_walks=M, num_steps=m, seed=77, rng=rng)
    o_indptr, o_ids, o_data = oracle.spg_build(nsize, remap)

    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == len(roots)
    cat = {k: np.concatenate([p[k] for p in parts]) for k in ("nsize", "ids", "sf")}
    assert np.array_equal(cat["nsize"], nsize) and np.array_equal(cat["nsize"], plain.nsize.cpu().numpy())
    assert np.array_equal(cat["ids"], remap[0]) and np.array_equal(cat["ids"], plain.ids.cpu().numpy())
    assert np.array_equal(cat["sf"], remap[1]) and np.array_equal(cat["sf"], plain.get_sf().cpu().numpy())
    ukeys1 = sets1.ukeys.cpu().numpy()
    nnz = z1.nnz
    for p in parts:
        assert np.array_equal(p["gkeys"], ukeys1) and np.array_equal(p["gk2"], ukeys1)        # one numbering everywhere
        assert np.array_equal(p["z_indptr"], o_indptr) and np.array_equal(p["z_indptr"], z1.indptr.cpu().numpy())
        assert np.array_equal(p["z_indices"], o_ids) and np.array_equal(p["z_indices"], z1.indices[:nnz].cpu().numpy())
        assert np.array_equal(p["z_data"], o_data) and np.array_equal(p["z_data"], z1.data[:nnz].cpu().numpy())
        # the rank's pairs, joined from its replica == the same pairs joined by one process
        e = torch.from_numpy(edge[:, int(p["plo"]):int(p["phi"])])
        xz, ind = sp.gather(e, z1, None, ptr=True, encode=sets1.feature_table())
        assert np.array_equal(p["ind"], ind.cpu().numpy()) and np.array_equal(p["xz"], xz.cpu().numpy())
    assert int(parts[0]["phi"]) == int(parts[1]["plo"]) and int(parts[1]["phi"]) == edge.shape[1]
    assert shard.rand_r_calls(csr.indptr, torch.from_numpy(roots), M, m) > 0
