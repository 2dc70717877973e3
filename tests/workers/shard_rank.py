"""One rank of the 2-rank GPU sharding test (tests/test_gpu_shard.py starts it as a FRESH process, so the rank
initialises the GPU itself; every rank uses cuda:0 and gloo carries the exchange step -- what a 1-GPU box can run;
on a node with several GPUs the same code runs over RCCL with one device per rank).

    python shard_rank.py RANK WORLD PORT OUT_DIR RNG FUSED
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SUBGACC_QUIET", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def problem():
    """the graph, roots and pairs of the test: rows shorter and longer than M, isolated nodes, repeated roots"""
    from gpu_helpers import sym_graph
    indptr, indices = sym_graph(3000, 9000, seed=11, hubs=3)
    rng = np.random.default_rng(4)
    roots = rng.integers(0, 3000, 2001).astype(np.int32)
    edge = rng.integers(0, 2001, (2, 777)).astype(np.int64)      # pairs of ROW numbers of the SpG
    return indptr, indices, roots, edge, 16, 3


def main():
    rank, world, port, out_dir, rng, fused = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6] == "1"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import surel_plus_amd as sp
    from surel_plus_amd import shard
    indptr, indices, roots, edge, M, m = problem()
    csr = sp.DeviceCSR(indptr, indices)
    # (1) the sampler alone, global LP-row numbers
    sets, gkeys, (lo, hi) = shard.sample_sets_sharded(shard.hip_sampler(csr, num_walks=M, num_steps=m, seed=77, rng=rng),
                                                      roots, rank, world)
    # (2) the sharded offline stage: every rank ends up with the whole SpG
    z, gk2, _ = shard.sample_spg_sharded(csr, roots, rank, world, num_walks=M, num_steps=m, seed=77, rng=rng, fused=fused)
    # (3) the rank's share of the pairs joined from the replicated store (no collective)
    e, (plo, phi) = shard.shard_pairs(torch.from_numpy(edge), rank, world)
    xz, ind = sp.gather(e, z, None, ptr=True, encode=shard.lp_table(gk2, M, m))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, plo=plo, phi=phi, nsize=sets.nsize.cpu().numpy(),
             ids=sets.ids.cpu().numpy(), sf=sets.sf.cpu().numpy(), gkeys=gkeys.cpu().numpy(), gk2=gk2.cpu().numpy(),
             z_indptr=z.indptr.cpu().numpy(), z_indices=z.indices.cpu().numpy(), z_data=z.data.cpu().numpy(),
             xz=xz.cpu().numpy(), ind=ind.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
