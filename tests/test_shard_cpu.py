"""CPU, world_size 2 over gloo: the multi-GPU sharding logic (contiguous root / pair ranges, merge of the
per-rank LP-row tables) reproduces the single-process result exactly.  The per-rank compute is an
oracle-backed stand-in here (no GPU in this container); on GPUs the same code runs over RCCL."""
import os
import types

import numpy as np
import pytest
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
    def run(query, lo, hi):
        q = query[lo:hi]
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
    # a counting allocator + no concatenation allowed: the replication must not hold more than the store it returns
    # (padded per-rank buffers + a list of `world` copies + torch.cat were >= 3x the store)
    allocated = []

    def counting_alloc(*shape, **kw):
        t = torch.empty(*shape, **kw)
        allocated.append(t.numel() * t.element_size())
        return t

    def no_cat(*a, **k):
        raise AssertionError("replicate_rows must not concatenate per-rank copies")
    real_cat, shard.torch.cat = shard.torch.cat, no_cat
    try:
        row_off, ids, data = shard.replicate_rows(sets.nsize, torch.from_numpy(ids), torch.from_numpy(data), alloc=counting_alloc)
    finally:
        shard.torch.cat = real_cat
    store = sum(t.numel() * t.element_size() for t in (row_off, ids, data)) + (row_off.numel() - 1) * sets.nsize.element_size()
    assert sum(allocated) == store, (sum(allocated), store)          # exactly the returned arrays (+ the gathered nsize): 1x
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


def test_eight_rank_sharded_sampling_and_replication_keep_the_single_process_order(tmp_path):
    """world size 8 (the node the driver scales to), gloo on the CPU: the eight per-rank LP-row tables merge in rank order to the
    single-process numbering (merge_unique_tables), and the eight row slices are replicated into the single-process SpG
    (replicate_rows: slices of very different sizes -- the golden query of 37 roots leaves some ranks 4 roots and others 5)."""
    world, port = 8, 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mp.spawn(_spg_worker, args=(world, port + 1, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, "gset_mixeddeg_s1.npz"))
    M, m = int(g["M"]), int(g["m"])
    nsize, remap, enc = oracle.gset_sampler(g["indptr"], g["indices"], g["query"], num_walks=M, num_steps=m, seed=9, rng="philox")
    indptr, ids, data = oracle.spg_build(nsize, remap)
    parts = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == len(g["query"])
    assert all(int(a["hi"]) == int(b["lo"]) for a, b in zip(parts, parts[1:]))
    assert np.array_equal(np.concatenate([p["nsize"] for p in parts]), nsize)
    assert np.array_equal(np.concatenate([p["ids"] for p in parts]), remap[0])
    assert np.array_equal(np.concatenate([p["sf"] for p in parts]), remap[1])      # global LP-row numbers, first-occurrence order
    for r in range(world):
        assert np.array_equal(parts[r]["gkeys"], pack_keys(enc, M))
        p = np.load(os.path.join(str(tmp_path), f"spg{r}.npz"))
        assert np.array_equal(p["row_off"], indptr) and np.array_equal(p["ids"], ids) and np.array_equal(p["data"], data)


def test_merge_unique_tables_over_eight_tables_is_rank_order_first_occurrence():
    rs = np.random.default_rng(0)
    tables = [torch.from_numpy(rs.permutation(40)[: rs.integers(0, 25)].astype(np.int64)) for _ in range(8)]
    gk, maps = shard.merge_unique_tables(tables)
    want = []
    for t in tables:
        for v in t.tolist():
            if v not in want:
                want.append(v)
    assert gk.tolist() == want
    for t, mp_ in zip(tables, maps):
        assert [want[i] for i in mp_.tolist()] == t.tolist()


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


def _mixed_graph(seed=5, N=400, M=8):
    """symmetric graph with isolated nodes, rows shorter and longer than M"""
    rng = np.random.default_rng(seed)
    src = rng.integers(10, N, 3000)
    dst = (rng.integers(10, N, 3000) ** 2 // N).clip(10, N - 1)        # skewed: some long rows
    keep = src != dst
    e = np.unique(np.stack([np.r_[src[keep], dst[keep]], np.r_[dst[keep], src[keep]]]), axis=1)
    indptr = np.zeros(N + 1, np.int64)
    np.add.at(indptr, e[0] + 1, 1)
    return np.cumsum(indptr).astype(np.int32), e[1].astype(np.int32)     # np.unique sorted by (row, col)


def test_rand_r_calls_is_the_stream_position_of_the_next_root():
    """shard.rand_r_calls(prefix) must be exactly what decides where the roots behind the prefix enter the sequential
    rand_r stream (subg_acc.c:771,807): two different prefixes with the same count leave the suffix's sets unchanged,
    a prefix with a different count does not -- checked with the oracle's sequential stream."""
    M, m = 8, 3
    indptr, indices = _mixed_graph(M=M)
    deg = np.diff(indptr)
    big, small, iso = np.flatnonzero(deg > M), np.flatnonzero((deg > 0) & (deg <= M)), np.flatnonzero(deg == 0)
    assert len(big) >= 8 and len(small) >= 8 and len(iso) >= 4
    suffix = np.r_[big[:3], small[:3], iso[:1], big[5:7]].astype(np.int32)
    pa = np.r_[big[:4], small[:4], iso[:2]].astype(np.int32)
    pb = np.r_[iso[2:4], small[4:8], big[4:8]].astype(np.int32)          # same class counts, other nodes, other order
    pc = np.r_[big[:5], small[:3], iso[:2]].astype(np.int32)             # one more shuffled root: M more draws
    ca, cb, cc = (shard.rand_r_calls(torch.from_numpy(indptr), p, M, m) for p in (pa, pb, pc))
    assert ca == cb == 4 * (M + M * (m - 1)) + 4 * M * (m - 1) and cc == ca + M

    def suffix_sets(prefix):
        nsize, remap, enc, raw = oracle.gset_sampler(indptr, indices, np.r_[prefix, suffix], num_walks=M, num_steps=m,
                                                     seed=3, rng="rand_r", debug=True)
        off = int(nsize[: len(prefix)].sum())
        return nsize[len(prefix):], remap[0][off:], raw[off:]
    a, b, c = suffix_sets(pa), suffix_sets(pb), suffix_sets(pc)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not all(np.array_equal(x, y) for x, y in zip(a, c))
    assert shard.rand_r_calls(indptr, np.zeros(0, np.int32), M, m) == 0
    assert shard.rand_r_calls(torch.from_numpy(indptr), pa, M, m, first_hop_wo=False) == 8 * M * m


def _device_worker(rank, world, port, same_device, backend_label, out_dir):
    """one rank of bench.py's device check over gloo: the records are all-gathered as in a real run, the device identity is faked"""
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    bench.device_identity = lambda dev: "AMD Instinct MI355X|uuid-0|idx0" if same_device else f"AMD Instinct MI355X|uuid-{rank}|idx{rank}"
    recs = bench.gather_rank_records(dist, world, rank, None, 0.02, 0.66, backend_label)
    code = 0
    try:
        bench.require_distinct_devices(recs, world, backend_label)
    except SystemExit as ex:
        code = ex.code
    with open(os.path.join(out_dir, f"code_{rank}"), "w") as f:
        f.write(str(code))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("same_device,backend,want", [(True, "nccl", 3), (False, "nccl", 0), (True, "gloo", 0)])
def test_bench_refuses_ranks_that_share_a_device_under_rccl(tmp_path, same_device, backend, want):
    """the first 8-GPU run must not be able to lie.  Two ranks over gloo, the all-gathered rank records as bench.py makes
    them with the device identity faked: a repeated device under "nccl" ends EVERY rank with exit code 3; distinct devices pass; the
    one-GPU gloo rehearsal (SUBGACC_DIST_BACKEND=gloo SUBGACC_SHARE_GPU=1) may share its device."""
    world, port = 2, 29640 + (os.getpid() + 7 * int(same_device) + 3 * (backend == "gloo")) % 300
    mp.spawn(_device_worker, args=(world, port, same_device, backend, str(tmp_path)), nprocs=world, join=True)
    assert [int(open(tmp_path / f"code_{r}").read()) for r in range(world)] == [want] * world
