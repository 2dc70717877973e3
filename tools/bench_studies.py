"""bench.py --full: the side studies (outside every clock of the driver's line; their numbers go to bench_detail.json only).
  batch_size_and_graph  the on-demand step at the reference's batch size (main.py:32: 1,024 pairs): eager, captured, 4 / 8 lanes, 64 batches per launch sequence
  offline_flow          the reference's own flow: all-N store once (S), resident-store joins (J, table / keyed), the amortised Q
  bench_hgather         main_horder.py's 2,048-triplet batches
  bench_mean_stage      the count form of the join + GEMM against the reference's first model stage (model.py:78-83)
  bench_walk_sampler    walk_sampler over all collab roots
bench.py's helpers (timer, step, medians, peak) are used as they are."""
import os
import sys
import time

import numpy as np
import torch

import bench
from bench import HBM_PEAK_GBS, UNIQ_CAPACITY, KernelTimer, _STEP_BUFS, _XZ_BUF, cpu_baseline, finish_step, hot_path_step, median, note  # noqa: F401


def batch_size_and_graph(sp, csr, M, k, rng, K):
    """Outside the clock (rank 0, 1 GPU): the reference's default batch of 1,024 pairs (main.py:32) and the headline's
    65,536, each as eager launches and as ONE captured HIP graph per step (stepgraph.CapturedStep), same double-buffered
    loop.  Pairs/s per variant."""
    from surel_plus_amd.graphs import query_pairs
    out = {}
    REPS = 3
    for B in (1024, 65536):
        steps = 300 if B == 1024 else max(K, 50)       # >= 20 ms per timed loop at either size; REPS loops, the median is reported
        edges = [query_pairs(csr, B, seed=7000 + s, device=csr.device) for s in range(min(steps, 100) + 3)]
        for mode in ("eager", "graph", "graph, 2 lanes", "graph, 4 lanes", "graph, 8 lanes", "graph, 8 lanes, inputs ready",
                     "many: 64 batches per launch sequence"):
            if mode.startswith("many") and B != 1024:
                continue
            try:
                _XZ_BUF.clear()
                _STEP_BUFS.clear()
                torch.cuda.empty_cache()
                lanes = int(mode.split()[1]) if "lanes" in mode else 0
                ready = "inputs ready" in mode        # the batches were produced (and synchronised) long ago: submit(sync=False)
                kw = dict(num_walks=M, num_steps=k - 1, seed=1, rng=rng, uniq_capacity=UNIQ_CAPACITY)
                # stepgraph.CapturedStepPool: `lanes` captured steps replayed on their own streams -- the dozen
                # few-microsecond kernels of one step run under the walk kernels of the others (a 1,024-pair step does
                # not fill the chip)
                pool = sp.CapturedStepPool(csr, B, lanes=lanes, **kw) if lanes else None
                caps = [sp.CapturedStep(csr, B, **kw) for _ in (0, 1)] if mode == "graph" else None
                many = None
                if mode.startswith("many"):
                    # the reference's batch size at the chip's batch size: 64 batches of 1,024 pairs as ONE captured step
                    # (StepBuffers(batch=1024)), cut into 64 reference-shaped (xz, indptr) on finish (spjoin.split_batches)
                    NB = 64
                    many = [sp.CapturedStep(csr, NB * B, batch=B, **kw) for _ in (0, 1)]
                    stacks = [torch.stack([edges[(i * NB + j) % len(edges)] for j in range(NB)]) for i in range(4)]

                def loop_many(n_groups):
                    pend = None
                    for g in range(n_groups):
                        q = many[g & 1](stacks[g % len(stacks)])
                        if pend is not None:
                            parts = pend.finish_batches()
                            assert len(parts) == NB
                        pend = q
                    pend.finish_batches()

                def loop(ids):
                    pending = []          # steps in flight
                    for s in ids:
                        e = edges[s % len(edges)]
                        if pool is not None:
                            if len(pending) == lanes:
                                pool.finish(pending.pop(0))
                            pending.append(pool.submit(e, sync=not ready))
                            continue
                        q = caps[s & 1](e) if caps is not None else hot_path_step(sp, csr, e, M, k, seed=1, rng=rng, slot=s & 1)
                        pending.append(q)
                        if len(pending) == 2:
                            d = pending.pop(0)
                            d.finish() if caps is not None else finish_step(*d)
                    for d in pending:
                        if pool is not None:
                            pool.finish(d)
                        else:
                            d.finish() if caps is not None else finish_step(*d)
                rates = []
                if many is not None:
                    groups = max(steps // NB, 8)
                    loop_many(3)
                    for _ in range(REPS):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        loop_many(groups)
                        torch.cuda.synchronize()
                        rates.append(B * NB * groups / (time.perf_counter() - t0))
                    n_steps = NB * groups
                else:
                    loop(range(8))
                    for _ in range(REPS):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        loop(range(8, 8 + steps))
                        torch.cuda.synchronize()
                        rates.append(B * steps / (time.perf_counter() - t0))
                    n_steps = steps
                out[f"B={B} {mode}"] = {"pairs_per_s": median(rates), "pairs_per_s_min": min(rates), "pairs_per_s_max": max(rates),
                                        "ms_per_step": B / median(rates) * 1e3, "steps": n_steps, "repeats": REPS}
                del caps, pool, many
            except Exception as ex:
                out[f"B={B} {mode}"] = {"failed": f"{type(ex).__name__}: {ex}"}
    return out



def offline_flow(sp, csr, M, k, B, K, cpu):
    """The reference's OWN flow, outside the clock (rank 0, 1 GPU): main.py:172-178 samples every node once (subg_matrix over
    all N: walk -> register -> number -> packed SpG resident in HBM = S of SURVEY 8(d)), train.py:120-127 then joins every
    batch from the resident store (J), from the Z_SF-table store and from the store re-keyed once (SpG.keyed).  Amortised Q
    for 1e8 pairs = 1e8 / (N / S + 1e8 / J); the same arithmetic with the CPU baseline's two halves beside it."""
    from surel_plus_amd.graphs import query_pairs
    from surel_plus_amd.spg import sample_spg
    dev, N = csr.device, csr.num_nodes
    # S / J / Q themselves are part of the default run since round 6 (bench.reference_flow); what --full adds follows
    base, z, sets, enc, table, zk = bench.reference_flow(sp, csr, M, k, B)
    idx = torch.arange(N, dtype=torch.int32, device=dev)
    # the paper's own sampler figure (Fig. 6a: citation2, m = 4, M = 200 -- 143..302 s on 16..1 CPU threads "incl. encoding + SpG
    # conversion") read with m as the hop count: all N roots, 4 hops, store resident + enc numbered
    t_m4 = None
    try:
        tm = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            z4, s4 = sample_spg(csr, idx, num_walks=M, num_steps=4, seed=111413, rng="philox", fused=True)
            e4 = s4.enc_int16()
            torch.cuda.synchronize()
            tm.append(time.perf_counter() - t0)
            m4_members, m4_rows = z4.nnz, int(e4.shape[0])
            del z4, s4, e4
        t_m4 = min(tm)
    except Exception as ex:
        t_m4 = None
        m4_members = m4_rows = f"{type(ex).__name__}: {ex}"
    # the reference's own loop AT the reference's batch size (main.py:32: 1,024 pairs; train.py:120-127), from the keyed store:
    # one eager gather per batch, one library call per batch (CapturedJoin; four lanes of them: CapturedJoinPool), and 64 batches per launch sequence
    # (gather_many: the permutation of an epoch is known when it starts) -- 3 repeats each, the median is reported
    b1024 = {}
    try:
        Bs, NB = 1024, 64
        es = [query_pairs(csr, Bs, seed=9500 + s_, device=dev) for s_ in range(128)]
        sbuf = torch.empty(NB * 2 * Bs * zk.max_len * 2 * k, dtype=torch.float32, device=dev)
        stacks = [torch.stack(es[i * NB:(i + 1) * NB]) for i in range(2)]

        def rate(fn, n_pairs):
            fn()
            rs = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                rs.append(n_pairs / (time.perf_counter() - t1))
            return {"pairs_per_s": median(rs), "pairs_per_s_min": min(rs), "pairs_per_s_max": max(rs)}

        def eager():
            for e in es * 2:
                sp.gather(e, zk, dev, ptr=True, encode=zk.slot_table(), out=sbuf, lazy=True)
        b1024["eager"] = rate(eager, 2 * len(es) * Bs)
        cjs = [sp.CapturedJoin(zk, Bs, encode=zk.slot_table()) for _ in range(4)]

        def graph():      # four captured joins in turn, each resolved (rows + status read back) three batches behind the queue
            pend = []
            for i, e in enumerate(es * 2):
                pend.append(cjs[i & 3](e))
                if len(pend) == 4:
                    pend.pop(0).finish()
            for q in pend:
                q.finish()
        b1024["graph"] = rate(graph, 2 * len(es) * Bs)          # (key kept from rounds 3-4: a CapturedJoin is ONE library call now, no graph)
        del cjs
        pool = sp.CapturedJoinPool(zk, Bs, lanes=4, encode=zk.slot_table())

        def lanes():      # the same on four streams (CapturedJoinPool), the batches known to be complete
            pend = []
            for e in es * 2:
                pend.append(pool.submit(e, sync=False))
                if len(pend) == 4:
                    pool.finish(pend.pop(0))
            for t in pend:
                pool.finish(t)
        b1024["pool_4lanes"] = rate(lanes, 2 * len(es) * Bs)
        del pool

        sbuf2 = torch.empty_like(sbuf)

        def many():     # a group of 64 batches is queued (no host read: lazy) before the previous group is consumed batch by batch
            prev = None
            for i in range(8):
                cur = sp.gather_many(stacks[i & 1], zk, dev, ptr=True, encode=zk.slot_table(), out=(sbuf, sbuf2)[i & 1], lazy=True)
                if prev is not None:
                    n_b = sum(1 for xz_b, ind_b in prev)          # boundaries read, 64 (xz, indptr) views made
                    assert n_b == NB
                prev = cur
            assert sum(1 for xz_b, ind_b in prev) == NB
        b1024["many_64"] = rate(many, 8 * NB * Bs)
        del sbuf, sbuf2, stacks
    except Exception as ex:
        b1024["failed"] = f"{type(ex).__name__}: {ex}"
    out = dict(base)
    out.update({"J_b1024_keyed": b1024, "all_N_4hop_sample_to_resident_spg_ms": t_m4 * 1e3 if t_m4 else None,
                "all_N_4hop_roots_per_s": N / t_m4 if t_m4 else None, "all_N_4hop_set_members": m4_members, "all_N_4hop_distinct_lp_rows": m4_rows})
    if cpu and cpu.get("sampler_roots_per_s"):
        Sc, Jc = cpu["sampler_roots_per_s"], cpu["join_pairs_per_s"]
        out["cpu_baseline"] = {"S_roots_per_s": Sc, "J_pairs_per_s": Jc, "Q_amortised_at_1e8_pairs": 1e8 / (N / Sc + 1e8 / Jc),
                               "cores": cpu["cores"], "kind": cpu["kind"],
                               "sample": "the two halves of cpu_baseline's bounded sample: the reference's gset_sampler (roots/s) and the "
                                         "oracle's SpG build + merge join (pairs/s), extrapolated to all N roots and 1e8 pairs"}
    return out, (z, table)


def bench_hgather(sp, z, table, k, K):
    """train.py:48-72 / main_horder.py:33: B = 2,048 triplets (u, v, w) -> xz [R4, 2, k] + ids, four blocks [U|w, W|u, V|w, W|v],
    joined from the resident store.  Outside the clock; >= 5 steps timed with HIP events."""
    dev, B = z.device, 2048
    gens = [torch.Generator(device=dev).manual_seed(400 + s_) for s_ in range(K + 2)]
    hedges = [torch.randint(0, z.n_rows, (3, B), device=dev, generator=g) for g in gens]
    for h in hedges[:2]:
        xz, ids = sp.hgather(h, z, dev, encode=table)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for h in hedges[2:]:
        xz, ids = sp.hgather(h, z, dev, encode=table)
    b.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = a.elapsed_time(b) / K
    rows = int(xz.shape[0])
    # per output row: id + SFptr read (8), xz written (8k), segment id written (8); per triplet: 3 ids (24) + 8 row offsets (64)
    abytes = rows * (8 + 8 * k + 8) + B * (24 + 64)
    # ... and 32 such batches per launch sequence (spjoin.hgather_many: bit for bit the per-batch results), 3 repeats, median
    many = None
    try:
        NB = 32
        stack = torch.stack([hedges[i % len(hedges)] for i in range(NB)])
        sp.hgather_many(stack, z, dev, encode=table)
        rs = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _r in range(4):
                parts = sp.hgather_many(stack, z, dev, encode=table)
            torch.cuda.synchronize()
            rs.append(4 * NB * B / (time.perf_counter() - t1))
        many = median(rs)
    except Exception as ex:
        many = f"{type(ex).__name__}: {ex}"
    # ... and one batch per call with everything built once (CapturedJoin(triplets=True): one small kernel lays the four blocks out,
    # ONE library call sizes and fills), alone and four lanes on their own streams; 3 repeats of 64 batches, median
    one_call = pool4 = None
    try:
        cj = sp.CapturedJoin(z, B, encode=table, triplets=True)
        pool = sp.CapturedJoinPool(z, B, lanes=4, encode=table, triplets=True)

        def loop_one():
            for i in range(64):
                cj(hedges[i % len(hedges)]).finish()

        def loop_pool():
            pend = []
            for i in range(64):
                pend.append(pool.submit(hedges[i % len(hedges)], sync=False))
                if len(pend) == 4:
                    pool.finish(pend.pop(0))
            for t in pend:
                pool.finish(t)
        res = []
        for fn in (loop_one, loop_pool):
            fn()
            rs = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                rs.append(64 * B / (time.perf_counter() - t1))
            res.append(median(rs))
        one_call, pool4 = res
        del cj, pool
    except Exception as ex:
        one_call = f"{type(ex).__name__}: {ex}"
    return {"metric": "triplets/sec (hgather from the resident store)", "value": B * K / wall, "unit": "triplets/s", "steps": K,
            "many32_triplets_per_s": many, "one_call_triplets_per_s": one_call, "pool4_triplets_per_s": pool4,
            "ms_per_step": wall / K * 1e3, "triplets_per_step": B, "xz_rows_last_step": rows,
            "roofline": {"bound": "hbm", "kernel": "sjoin_pair_kernel (4 blocks per triplet) + sizes + scan", "achieved": abytes / (ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": ms, "algorithmic_bytes_per_launch": abytes,
                         "note": "2,048 triplets are ~8k workgroups of one pair each: launch- and latency-bound, as the reference's batch size is"}}


def bench_mean_stage(sp, sampler_mod, z, table, csr, k, B, K):
    """SURVEY 8(f).1 / model.py:78-83: the first model stage for mean aggregation, `pe_embedding(xz).sum(-2)` + segment mean, three ways
    from the resident store, forward only, B pairs per step: (a) the reference's form over the materialised xz [R,2,k] (gather + the
    MLP on every row + a segment reduce), (b) spjoin.mean_stage: the count form C [2B, c+1] @ pe_embedding(Z_SF) (xz never exists),
    (c) the sparse form of C: the DISTINCT index pairs of every segment with their multiplicities (gather_pairs) -> gather-add of the
    embedded table rows -> segment sum.  H = 96 (main.py:30) and 256.  Outside the clock; 3 repeats of K steps, medians."""
    from surel_plus_amd.graphs import query_pairs
    dev = z.device
    edges = [query_pairs(csr, B, seed=9700 + s_, device=dev) for s_ in range(4)]
    # (a) materialises [R, 2, H] activations -- 100 GB at B = 65,536, H = 256 -- so it runs on batches of 4,096 pairs
    Bref = min(B, 4096)
    out = {"pairs_per_step": B, "pairs_per_step_ref_style": Bref, "table_rows": int(table.shape[0])}
    timer = KernelTimer()
    for H in (96, 256):
        torch.manual_seed(0)
        embed = torch.nn.Sequential(torch.nn.Linear(k, H), torch.nn.ReLU(), torch.nn.Linear(H, H)).to(dev)
        buf = torch.empty(2 * B * z.max_len * 2 * k, dtype=torch.float32, device=dev)

        def ref_style(e):
            xz, ind = sp.gather(e, z, dev, ptr=True, encode=table, out=buf)
            x = embed(xz).sum(dim=-2)
            return torch.segment_reduce(x, "mean", offsets=ind, axis=0).view(2, -1, H)

        def fused(e):
            return sp.mean_stage(e, z, table, embed)

        def sparse(e):
            pairs, mult, indptr = sp.gather_pairs(e, z)
            t = embed(table)
            rows = (t[pairs[:, 0].long()] + t[pairs[:, 1].long()]) * mult.to(torch.float32)[:, None]
            S = indptr.numel() - 1
            seg = torch.repeat_interleave(torch.arange(S, device=dev), indptr[1:] - indptr[:-1], output_size=pairs.shape[0])
            num = torch.zeros((S, H), device=dev, dtype=torch.float32).index_add_(0, seg, rows)
            own = torch.cat([e[0], e[1]])
            sizes = (z.indptr[own + 1] - z.indptr[own]).clamp(min=1).to(torch.float32)
            return (num / sizes[:, None]).view(2, -1, H)

        with torch.no_grad():
            e0 = edges[0][:, :Bref].contiguous()
            a, b_, c_ = ref_style(e0), fused(e0), sparse(e0)
            out[f"H{H}_max_abs_diff_fused_vs_ref_style"] = float((a - b_).abs().max().item())
            out[f"H{H}_max_abs_diff_sparse_vs_ref_style"] = float((a - c_).abs().max().item())
            del a, b_, c_
            for nm, fn in (("ref_style", ref_style), ("fused", fused), ("sparse", sparse)):
                rs = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    Bn = Bref if nm == "ref_style" else B
                    for s_ in range(K):
                        fn(edges[s_ % len(edges)][:, :Bn])
                    torch.cuda.synchronize()
                    rs.append(Bn * K / (time.perf_counter() - t0))
                out[f"H{H}_{nm}_pairs_per_s"] = median(rs)
            # the count kernel alone (HIP events) and the GEMM behind it
            sampler_mod.KERNEL_TIMER = timer
            timer.enabled = True
            C, sizes = sp.gather_counts(edges[1], z, table.shape[0])
            for s_ in range(5):
                C, sizes = sp.gather_counts(edges[s_ % len(edges)], z, table.shape[0])
            timer.enabled = False
            sampler_mod.KERNEL_TIMER = None
            t = embed(table)
            a_, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            C @ t
            a_.record()
            for _ in range(5):
                C @ t
            b2.record()
            torch.cuda.synchronize()
            out[f"H{H}_gemm_ms"] = a_.elapsed_time(b2) / 5
        rows = int(z.indptr[torch.cat([edges[1][0], edges[1][1]]) + 1].sum().item() - z.indptr[torch.cat([edges[1][0], edges[1][1]])].sum().item())
        del buf, embed
    ms, _ = timer.mean_ms("sjoin_counts")
    # algorithmic bytes of the count kernel: per pair 16 + 32, per row of the two sets 8 read (id + SFptr), C written 2B * (c+1) * 4
    abytes = B * 48 + rows * 8 + 2 * B * int(table.shape[0]) * 4
    out.update({"counts_kernel_ms": ms, "counts_algorithmic_bytes_per_launch": abytes, "rows_last_step": rows,
                "counts_frac_of_hbm_peak": (abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms else None,
                "dense_C_bytes": 2 * B * int(table.shape[0]) * 4})
    return out


def bench_walk_sampler(sp, sampler_mod, dev, K):
    """subg_acc.c:316-389 on the collab-like graph (configs[0]/[1] parameters: M = 200, 2 hops, first hop without replacement):
    all N roots -> raw walks int32 [n, M*(m+1)] + per-root (ids, counts) in step-major first-visit order, device part only
    (the drop-in's numpy hand-over is PCIe).  Outside the clock; >= 5 steps."""
    from surel_plus_amd import _lib
    from surel_plus_amd.graphs import preset_graph
    csr = preset_graph("collab", device=dev)
    M, m, n = 200, 2, csr.num_nodes
    q = torch.arange(n, dtype=torch.int32, device=dev)

    def run():
        s_ = sampler_mod.sample_sets(csr, q, num_walks=M, num_steps=m, seed=5, rng="rand_r", first_hop_wo=True,
                                     order=_lib.ORDER_STEP_MAJOR, cap_root_degree=False, emit_walks=True, rng_streams=4, dedup=False)
        return s_, s_.counts_int32()
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        sets, counts = run()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    deg = (csr.indptr[1:] - csr.indptr[:-1]).long()
    X = int(sets.X)
    # reads as SURVEY 8(d)'s sampler formula; writes: the raw walks, the ids and the [count, m+1] landing counts (the reference's outputs)
    abytes = int((4 + 8 + 4 * torch.clamp(deg, max=M) + 12 * M * (m - 1) * (deg > 0).long()).sum().item()) + n * 4 * M * (m + 1) + X * (4 + 4 * (m + 1))
    ms = wall / K * 1e3
    return {"metric": "roots/sec (walk_sampler, device part)", "value": n * K / wall, "unit": "roots/s", "steps": K, "ms_per_step": ms,
            "roots_per_step": n, "set_members": X, "rng": "rand_r (4 streams)",
            "roofline": {"bound": "hbm", "kernel": "walk_sets_kernel / walk_pipe_kernel (+ rng positions, scan, compaction, LP unpack)",
                         "achieved": abytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel_ms": ms,
                         "algorithmic_bytes_per_launch": abytes, "note": "whole call (several kernels + allocations), wall clock"}}
