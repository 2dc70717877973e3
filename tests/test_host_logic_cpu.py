"""CPU: host-side logic of the mirror layer that needs no GPU -- argument casting rules of the reference's
C module (subg_acc/subg_acc.c:663-676), synthetic inputs, signatures of the drop-in functions."""
import inspect

import numpy as np
import pytest
import torch

import surel_plus_amd as sp
from surel_plus_amd import graphs, subg_acc as mirror


def test_csr_casting_rules_match_the_reference():
    ok_ptr, ok_idx = np.array([0, 1, 2], np.int32), np.array([1, 0], np.int32)
    # PyArray_FROM_OTF(..., NPY_INT, NPY_ARRAY_IN_ARRAY) is a safe cast: floats / int64 indices are rejected
    with pytest.raises(TypeError, match="Input parsing error"):
        mirror._csr_from_host(ok_ptr.astype(np.float64), ok_idx)
    with pytest.raises(TypeError, match="Input parsing error"):
        mirror._csr_from_host(ok_ptr, ok_idx.astype(np.int64))
    with pytest.raises(TypeError, match="Input parsing error"):
        mirror._csr_from_host(ok_ptr, ok_idx.astype(np.uint32))
    # a malformed graph would be a segfault in the reference and a device fault here: DeviceCSR refuses it where the
    # arrays live (a few reductions on the device after the upload; the same torch code runs on host tensors here)
    with pytest.raises(IndexError):
        sp.DeviceCSR(np.array([0, 3, 2], np.int32), ok_idx, device="cpu")     # offsets not monotone / past the end
    with pytest.raises(IndexError):
        sp.DeviceCSR(np.array([1, 1, 2], np.int32), ok_idx, device="cpu")     # does not start at 0
    with pytest.raises(IndexError):
        sp.DeviceCSR(ok_ptr, np.array([1, 7], np.int32), device="cpu")        # neighbour id out of range
    with pytest.raises(IndexError):
        sp.DeviceCSR(ok_ptr, np.array([1, -1], np.int32), device="cpu")
    assert sp.DeviceCSR(ok_ptr, ok_idx, device="cpu").num_nodes == 2
    assert sp.DeviceCSR(np.array([0], np.int64), np.zeros(0, np.int32), device="cpu").num_nodes == 0
    with pytest.raises(IndexError):
        mirror._checked_query(np.array([0, 5]), 2)
    # smaller integer types are safe casts; they fail later only because there is no GPU here
    if not torch.cuda.is_available():
        with pytest.raises(sp.SubgAccError, match="no HIP device"):
            mirror._csr_from_host(ok_ptr.astype(np.int16), ok_idx.astype(np.int8))


def test_signatures_mirror_the_reference_module():
    """kwlist of subg_acc.c:655 / :322 and the defaults of :653 / :320."""
    g = inspect.signature(mirror.gset_sampler).parameters
    assert list(g)[:9] == ["indptr", "indices", "query", "num_walks", "num_steps", "bucket", "nthread", "seed", "debug"]
    assert (g["num_walks"].default, g["num_steps"].default, g["bucket"].default, g["nthread"].default,
            g["seed"].default, g["debug"].default) == (100, 3, -1, -1, 111413, -1)
    w = inspect.signature(mirror.walk_sampler).parameters
    assert list(w)[:8] == ["ptr", "neighs", "query", "num_walks", "num_steps", "nthread", "seed", "replacement"]
    assert (w["num_walks"].default, w["num_steps"].default, w["seed"].default, w["replacement"].default) == \
        (100, 3, 111413, False)
    b = inspect.signature(mirror.batch_sampler).parameters      # kwlist of subg_acc.c:397, defaults of :395
    assert list(b)[:8] == ["ptr", "neighs", "query", "num_walks", "num_steps", "thld", "nthread", "seed"]
    assert (b["num_walks"].default, b["num_steps"].default, b["thld"].default, b["nthread"].default, b["seed"].default) == \
        (200, 8, 1000, -1, 111413)
    j = inspect.signature(mirror.walk_join).parameters          # kwlist of subg_acc.c:515
    assert list(j) == ["walk", "key", "query", "nthread", "return_idx"]
    assert mirror.add(3, 4) == 3 * 2 + 4 * 7                    # subg_acc.c:116
    assert not hasattr(mirror, "run")                            # the system() wrapper is deliberately absent
    # train.py:13,48,75,88
    assert list(inspect.signature(sp.gather).parameters)[:5] == ["edge", "x", "device", "ptr", "encode"]
    assert list(inspect.signature(sp.hgather).parameters) == ["hedge", "x", "device", "encode"]
    assert list(inspect.signature(sp.bgather).parameters) == ["edge", "x", "out"]
    assert list(inspect.signature(sp.pgather).parameters) == ["edge", "M", "device", "encode", "gather_func", "ptr", "njobs"]
    # sampler/random_walks.py:74
    assert list(inspect.signature(sp.subg_matrix).parameters)[:4] == ["G", "train_idx", "num_walks", "num_steps"]
    import subg_acc as top                                       # the drop-in module name
    assert top.gset_sampler is mirror.gset_sampler and top.walk_sampler is mirror.walk_sampler
    assert top.batch_sampler is mirror.batch_sampler and top.walk_join is mirror.walk_join


def test_synthetic_graph_is_symmetric_simple_and_sorted():
    g = graphs.powerlaw_graph(20000, 8.2, seed=0, device="cpu")
    ip, ix = g.indptr.long(), g.indices.long()
    deg = ip[1:] - ip[:-1]
    rows = torch.repeat_interleave(torch.arange(g.num_nodes), deg)
    assert g.indices.dtype == torch.int32 and 6.5 < g.nnz / g.num_nodes < 10
    assert not bool((rows == ix).any())                                     # no self loops
    fwd, bwd = rows * g.num_nodes + ix, ix * g.num_nodes + rows
    assert torch.equal(torch.sort(fwd)[0], torch.sort(bwd)[0])              # G == G.T (dataloader.py:122-135)
    assert torch.unique(fwd).numel() == fwd.numel()                         # simple
    assert bool((fwd[1:] > fwd[:-1]).all())                                 # rows sorted by neighbour id
    again = graphs.powerlaw_graph(20000, 8.2, seed=0, device="cpu")
    assert torch.equal(again.indices, g.indices)                            # seeded
    e = graphs.query_pairs(g, 1000, seed=7, device="cpu")
    assert e.shape == (2, 1000) and e.dtype == torch.int64 and int(e.max()) < g.num_nodes
    # the first half of the pairs are edges of the graph
    key = e[0, :500] * g.num_nodes + e[1, :500]
    assert bool(torch.isin(key, fwd).all())


def test_ppr_like_spg_and_directed_graph_shapes():
    z = graphs.ppr_like_spg(2000, 100, seed=3, device="cpu")
    assert z.data.dtype == torch.float64 and z.max_len == 100 and z.indices.numel() == 200000
    rows = z.indices.view(2000, 100)
    assert bool((rows[:, 1:] > rows[:, :-1]).all()) and float(z.data.min()) > 0 and float(z.data.max()) <= 1
    d = graphs.directed_powerlaw_graph(50000, 20.0, seed=3, device="cpu", chunk=1 << 20)
    deg = d.indptr[1:] - d.indptr[:-1]
    assert int(deg.min()) >= 1 and 15 < d.nnz / d.num_nodes < 25            # no dead ends


def test_make_cfg_rejects_bad_parameters():
    from surel_plus_amd.sampler import make_cfg
    g = graphs.powerlaw_graph(100, 4.0, device="cpu")
    with pytest.raises(TypeError):
        make_cfg(g, 0, 3)
    with pytest.raises(KeyError):
        make_cfg(g, 10, 3, rng="mt19937")
    cfg = make_cfg(g, 200, 3, seed=-1)
    assert cfg.seed == 0xFFFFFFFF and cfg.indptr64 == 0 and cfg.first_hop_wo == 1


def test_kernel_shape_rules_of_the_fused_row_path():
    """which launches get key rows / the specialised kernel (host-side rules that mirror csrc/walk_rows.hip:launch_walk_rows)"""
    from surel_plus_amd.sampler import key_rows_ok, walk_kernel_name
    # every reference configuration up to 3 hops: 32-bit LP keys, a 512- or 1,024-slot table
    assert key_rows_ok(200, 2) and key_rows_ok(200, 3) and key_rows_ok(100, 3) and key_rows_ok(120, 2)
    assert not key_rows_ok(200, 4)        # 4*8+1 = 33 bits
    assert not key_rows_ok(300, 2)        # more walks than lanes
    assert not key_rows_ok(16, 2)         # a 64-slot table: the general kernel
    assert not key_rows_ok(200, 1)
    assert walk_kernel_name(None, 200, 3, True) == "walk_rows_kernel"
    assert walk_kernel_name(None, 200, 5, True) == "walk_sets_kernel<SPG>"
    assert walk_kernel_name(None, 200, 2, False) == "walk_pipe_kernel"
    assert walk_kernel_name(None, 300, 2, False) == "walk_sets_kernel"


def test_key_row_forms_and_the_launcher_mirror_of_round_4():
    """4 hops: 32-bit key rows up to M = 127 (512- and 1,024-slot tables), 64-bit key rows for M = 128 .. 204, the table form beyond;
    a truncating bucket or more than 256 walks leave the launch to the general fused kernel (sampler.rows_kernel_takes)"""
    from surel_plus_amd.sampler import key_rows_form, key_rows_ok, rows_kernel_takes, walk_kernel_name
    assert key_rows_form(100, 4) == 32 and key_rows_form(127, 4) == 32          # 4*7+1 = 29 bits
    assert key_rows_form(128, 4) == 64 and key_rows_form(200, 4) == 64 and key_rows_form(204, 4) == 64      # 33 bits, 1,024 slots
    assert key_rows_form(205, 4) == 0                                            # 821 members: beyond the 1,024-slot table
    assert key_rows_form(50, 4) == 0                                             # a 256-slot table: the general kernel
    assert key_rows_form(200, 3) == 32 and key_rows_form(200, 5) == 0 and key_rows_form(300, 2) == 0
    assert key_rows_ok(100, 4) and not key_rows_ok(200, 4)                       # batched registration works on 32-bit keys
    assert rows_kernel_takes(200, 3) and not rows_kernel_takes(200, 3, bucket=50) and not rows_kernel_takes(300, 2)
    assert walk_kernel_name(None, 200, 2, True, 50) == "walk_sets_kernel<SPG>"


def test_batch_views_cut_a_many_batch_join_into_reference_shaped_pieces():
    """spjoin.BatchViews (what gather_many / sample_and_gather_many / hgather_many return) on host tensors: per-batch rows, pointers
    from 0 (train.py:21-22) or segment ids, laziness, the status word, the error cases -- no GPU needed for the bookkeeping"""
    from surel_plus_amd.spjoin import BatchViews, split_batches
    B, nb = 3, 4
    lens = torch.arange(1, 2 * B * nb + 1) % 5                     # segment lengths, some empty
    seg = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)])
    R = int(seg[-1])
    xz = torch.arange(R * 2 * 2, dtype=torch.float32).view(R, 2, 2)
    v = split_batches(xz, seg, B)
    assert len(v) == nb and v._bounds is None                      # nothing read yet
    for b in range(nb):
        xb, pb = v[b]
        lo, hi = int(seg[2 * B * b]), int(seg[2 * B * (b + 1)])
        assert torch.equal(xb, xz[lo:hi]) and pb.dtype == torch.int64 and pb.numel() == 2 * B + 1
        assert int(pb[0]) == 0 and torch.equal(pb[1:] - pb[:-1], lens[2 * B * b: 2 * B * (b + 1)])
    assert torch.equal(v[-1][0], v[nb - 1][0]) and len(v[1:3]) == 2 and [x.shape[0] for x, _ in v] == [int(seg[6 * (b + 1)] - seg[6 * b]) for b in range(nb)]
    with pytest.raises(IndexError):
        v[nb]
    # segment ids instead of pointers (ptr=False / hgather): ids restart at 0 in every batch
    ids = torch.repeat_interleave(torch.arange(2 * B * nb) % (2 * B), lens)
    w = BatchViews(xz, seg, 2 * B, ids=ids)
    for b in range(nb):
        xb, ib = w[b]
        assert ib.numel() == xb.shape[0] and (ib.numel() == 0 or (int(ib.min()) >= 0 and int(ib.max()) < 2 * B))
    # the lazy form carries the join's status word: a row number outside the store raises when the first batch is taken
    flags = torch.zeros(4, dtype=torch.int32)
    flags[3] = 16
    bad = BatchViews(xz, seg, 2 * B, flags=flags, n_rows=7)
    with pytest.raises(IndexError, match="7 rows"):
        bad[0]
    with pytest.raises(ValueError):
        BatchViews(xz, seg, 5)                                      # 24 segments are not a whole number of batches of 5
    assert BatchViews(xz[:0], torch.zeros(1, dtype=torch.int64), 2 * B) == []


def test_the_join_kernels_quotient_formula_is_the_ieee_division():
    """csrc/sjoin.hip:lp_quotient computes count / num_walks as q0 = c * (1/M), r = fma(-M, q0, c), q = fma(r, 1/M, q0) for
    num_walks < 4,096 (and divides beyond).  main.py:174 divides (`float32(enc) / M`): the formula must give the correctly rounded
    quotient for EVERY count a key field of that num_walks can hold (c < 2^bits(M)), not just for most.  Exact emulation: the fma's
    exact arguments fit the 64-bit significand of x87 long double (c - M*q0: < 2^36 in units of ulp(q0); q0 + r*(1/M): < 2^60)."""
    if np.finfo(np.longdouble).nmant < 63:
        pytest.skip("needs the 64-bit significand of x87 long double")
    cases = 0
    for M in range(1, 4096):
        c = np.arange(0, 1 << int(M).bit_length(), dtype=np.int64)
        a, fm = c.astype(np.float32), np.float32(M)
        y = np.float32(1.0) / fm
        q0 = a * y
        r_exact = a.astype(np.longdouble) - np.longdouble(fm) * q0.astype(np.longdouble)
        r = r_exact.astype(np.float32)
        assert np.array_equal(r.astype(np.longdouble), r_exact)            # fma(-M, q0, c) is exact: nothing is rounded away
        q = (q0.astype(np.longdouble) + r.astype(np.longdouble) * y.astype(np.longdouble)).astype(np.float32)
        assert np.array_equal(q, a / fm), M
        cases += c.size
    assert cases == 11184810


def test_keep_until_never_drops_an_unfinished_entry_under_four_threads():
    """VERDICT r5 #8a / SURVEY 8(b) ("safe to call from the 4 pgather-style Python threads"): _lib.keep_until holds the pinned block a
    publish kernel is still going to write until its event has passed.  Four threads queue 4,000 entries each while their events
    finish in any order; an entry may only leave the list once ITS event reports finished (round 5's look-then-pop could pop the
    unfinished entry behind a finished one when two threads interleaved)."""
    import random
    import threading

    from surel_plus_amd import _lib

    class FakeEvent:
        def __init__(self):
            self.done = False

        def query(self):
            return self.done

    dropped_early = []

    class Block:
        def __init__(self, ev):
            self.ev = ev

        def __del__(self):
            if not self.ev.done:
                dropped_early.append(1)

    saved = list(_lib._KEPT)
    _lib._KEPT.clear()
    try:
        def worker(seed):
            rng = random.Random(seed)
            mine = []
            for _ in range(4000):
                ev = FakeEvent()
                _lib.keep_until(ev, Block(ev))
                mine.append(ev)
                if len(mine) > 3:
                    mine.pop(rng.randrange(len(mine))).done = True       # the kernels of different streams finish in any order
            return mine
        left = []
        threads = [threading.Thread(target=lambda s=s: left.extend(worker(s))) for s in range(4)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not dropped_early
        pending = {id(ev) for ev in left}
        assert pending <= {id(ev) for ev, _ in _lib._KEPT}                # every unfinished entry is still held
        for ev in left:
            ev.done = True
        done_ev = FakeEvent()
        done_ev.done = True
        _lib.keep_until(done_ev, object())
        assert len(_lib._KEPT) == 1 and not dropped_early                  # ... and leaves once its event has passed
    finally:
        for ev, _ in _lib._KEPT:
            ev.done = True
        _lib._KEPT[:] = saved
