"""CPU, world_size 2 over gloo: the multi-GPU sharding logic (contiguous root / pair ranges, merge of the
per-rank LP-row tables) reproduces the single-process result exactly.  The per-rank compute is an
oracle-backed stand-in here (no GPU in this container); on GPUs the same code runs over RCCL."""
import os
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import GOLDEN
from surel_plus_amd import shard


def pack_keys(rows, M):
    """int16 LP rows [c, m+1] -> the packed 64-bit keys of subg_acc.c:936-955 (as int64 bit patterns)."""
    shift = int(M).bit_length()
    m = rows.shape[1] - 1
    key = np.zeros(rows.shape[0], np.uint64)
    for j in range(1, m + 1):
        key = (key << np.uint64(shift)) | rows[:, j].astype(np.uint64)
    key |= (rows[:, 0] != 0).astype(np.uint64) << np.uint64(m * shift)
    return key.view(np.int64)


def oracle_sampler(g, M, m, seed):
    def run(q, lo=0):
        nsize, remap, enc = oracle.gset_sampler(g["indptr"], g["indices"], q, num_walks=M, num_steps=m, seed=seed,
                                                rng="philox")
        return types.SimpleNamespace(nsize=torch.from_numpy(nsize), ids=torch.from_numpy(remap[0]),
                                     sf=torch.from_numpy(remap[1]).long(), ukeys=torch.from_numpy(pack_keys(enc, M)))
    return run


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(GOLDEN, "gset_mixeddeg_s1.npz"))
    M, m = int(g["M"]), int(g["m"])
    sets, gkeys, (lo, hi) = shard.sample_sets_sharded(oracle_sampler(g, M, m, 9), g["query"], rank, world)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), nsize=sets.nsize.numpy(), ids=sets.ids.numpy(), sf=sets.sf.numpy(),
             gkeys=gkeys.numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def _spg_worker(rank, world, port, out_dir):
    """sharded offline stage: sample the rank's roots, number LP rows globally, sort the rank's rows, replicate"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(GOLDEN, "gset_mixeddeg_s1.npz"))
    M, m = int(g["M"]), int(g["m"])
    sets, gkeys, _ = shard.sample_sets_sharded(oracle_sampler(g, M, m, 9), g["query"], rank, world)
    _, ids, data = oracle.spg_build(sets.nsize.numpy(), np.stack([sets.ids.numpy(), sets.sf.numpy().astype(np.int32)]))
    row_off, ids, data = shard.replicate_rows(sets.nsize, torch.from_numpy(ids), torch.from_numpy(data))
    np.savez(os.path.join(out_dir, f"spg{rank}.npz"), row_off=row_off.numpy(), ids=ids.numpy(), data=data.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_spg_is_replicated_identically(tmp_path):
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_spg_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, "gset_mixeddeg_s1.npz"))
    M, m = int(g["M"]), int(g["m"])
    nsize, remap, enc = oracle.gset_sampler(g["indptr"], g["indices"], g["query"], num_walks=M, num_steps=m, seed=9, rng="philox")
    indptr, ids, data = oracle.spg_build(nsize, remap)
    for r in range(world):
        p = np.load(os.path.join(str(tmp_path), f"spg{r}.npz"))
        assert np.array_equal(p["row_off"], indptr) and np.array_equal(p["ids"], ids) and np.array_equal(p["data"], data)


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 1000):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_merge_unique_tables_is_first_occurrence_order():
    a = torch.tensor([5, 3, 9], dtype=torch.int64)
    b = torch.tensor([3, 7, 5, 1], dtype=torch.int64)
    c = torch.zeros(0, dtype=torch.int64)
    gk, maps = shard.merge_unique_tables([a, c, b])
    assert gk.tolist() == [5, 3, 9, 7, 1]
    assert maps[0].tolist() == [0, 1, 2] and maps[1].tolist() == [] and maps[2].tolist() == [1, 3, 0, 4]


def test_two_rank_sharded_sampling_equals_single_process(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, "gset_mixeddeg_s1.npz"))
    M, m = int(g["M"]), int(g["m"])
    nsize, remap, enc = oracle.gset_sampler(g["indptr"], g["indices"], g["query"], num_walks=M, num_steps=m, seed=9,
                                            rng="philox")
    parts = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == len(g["query"])
    assert np.array_equal(np.concatenate([p["nsize"] for p in parts]), nsize)
    assert np.array_equal(np.concatenate([p["ids"] for p in parts]), remap[0])
    assert np.array_equal(np.concatenate([p["sf"] for p in parts]), remap[1])      # global LP-row numbers
    for p in parts:
        assert np.array_equal(p["gkeys"], pack_keys(enc, M))                        # identical table on every rank


def test_shard_pairs():
    e = torch.arange(20).view(2, 10)
    a, (lo, hi) = shard.shard_pairs(e, 1, 3)
    assert (lo, hi) == (4, 7) and torch.equal(a, e[:, 4:7])
