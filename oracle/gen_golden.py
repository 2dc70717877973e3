#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE.  Needs /root/reference (read-only) and `make -C oracle ref`; it can NOT run on the
GPU box.  The fixtures it writes are data only: seeded inputs and the outputs the reference produced.

  G1  gset_*.npz    subg_acc.gset_sampler(nthread=1, debug=1)          subg_acc/subg_acc.c:649-1034
  G2  walk_*.npz    subg_acc.walk_sampler(both first-hop modes)        subg_acc/subg_acc.c:316-389
  G3  spg_*.npz     scipy COO->CSR exactly as sampler/random_walks.py:79-81 calls it
  G4  sjoin_*.npz   train.gather / train.pgather(bgather) (int + float payload, ptr True/False)  train.py:13-111
  G5  hjoin_*.npz   train.hgather                                      train.py:48-72
  G6  rand_r.npz    glibc rand_r streams (libc.so.6 via ctypes), the RNG the reference calls
  G7  walkjoin_*.npz subg_acc.walk_join over walk_sampler's output (legacy SUREL join)  subg_acc/subg_acc.c:509-647
  G8  batch_*.npz   subg_acc.batch_sampler (legacy SUREL mini-batch former); the reference seeds it with seed + getpid()
                    (:421), so every fixture records the effective seed of the run that made it   subg_acc/subg_acc.c:391-507

  G9  gset_dir*.npz / walk_dir*.npz   the same two samplers on DIRECTED graphs: walks reach nodes without out-edges, draw
                    nothing there and stay (subg_acc.c:804-808, :168-172, :236-240) -- the rand_r stream position of every
                    later walk then depends on how the earlier ones ended

`python oracle/gen_golden.py [rand_r gset walk sjoin walkjoin batch directed]` regenerates the named groups (default: all).
"""
import ctypes
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

ref = oracle.ref_module()
assert ref is not None, "run `make -C oracle ref` first (needs /root/reference)"
sys.path.insert(0, "/root/reference")
import torch  # noqa: E402
import train as ref_train  # noqa: E402  (the reference's train.py: gather/pgather/hgather)


def sym_graph(N, E, seed, isolated=0, star=0, k2=False):
    """Small undirected simple graph, symmetrised like dataloader.py:122-135 does (G + G.T)."""
    rng = np.random.default_rng(seed)
    r = rng.integers(0, N, E)
    c = rng.integers(0, N, E)
    if star:
        r = np.concatenate([r, np.zeros(star, int)])
        c = np.concatenate([c, rng.choice(np.arange(1, N), star, replace=False)])
    tot = N + isolated + (2 if k2 else 0)
    if k2:  # an isolated edge component, as subg_acc/test/test.py:45 needs
        r = np.concatenate([r, [tot - 2]])
        c = np.concatenate([c, [tot - 1]])
    A = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(tot, tot))
    A = sp.csr_matrix(A + A.T)
    A.setdiag(0)
    A.eliminate_zeros()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


def mask_isolated(ptr, query, nsize, remap):
    """The reference never writes the id of an isolated root (subg_acc.c:753-761): overwrite that
    uninitialised word with the root id so the fixture is well defined."""
    remap = remap.copy()
    deg = np.diff(ptr)[query]
    off = np.concatenate([[0], np.cumsum(nsize)])[:-1]
    remap[0, off[deg == 0]] = np.asarray(query)[deg == 0]
    return remap


GSET_CASES = {
    # name: (N, E, gseed, isolated, star, k2, M, m, bucket, query kind)
    "tiny":      (50, 120, 0, 0, 0, False, 5, 2, -1, "all"),
    "mixeddeg":  (300, 900, 1, 3, 100, True, 20, 3, -1, "all"),       # deg <M, =M.., >M (star hub), isolated, K2
    "dense":     (400, 16000, 3, 0, 0, False, 30, 2, -1, "all"),      # most roots have deg > M
    "bucket":    (300, 900, 1, 0, 100, False, 20, 3, 12, "all"),      # bucket overflow (visits dropped)
    "dupquery":  (200, 700, 5, 1, 0, False, 16, 4, -1, "dup"),        # repeated / unordered roots
    "collablike": (1500, 6000, 7, 2, 300, True, 200, 2, -1, "sub"),   # the collab parameters M=200, m=2
}


def gen_gset():
    for name, (N, E, gs, iso, star, k2, M, m, bucket, qk) in GSET_CASES.items():
        ptr, idx = sym_graph(N, E, gs, iso, star, k2)
        n_nodes = len(ptr) - 1
        rng = np.random.default_rng(100 + gs)
        if qk == "all":
            q = np.arange(n_nodes)
        elif qk == "dup":
            q = rng.integers(0, n_nodes, 150)
        else:
            q = np.sort(rng.choice(n_nodes, 200, replace=False))
        for seed in (1, 111413):
            nsize, remap, enc, raw = ref.gset_sampler(ptr, idx, q, num_walks=M, num_steps=m, bucket=bucket,
                                                      nthread=1, seed=seed, debug=1)
            remap = mask_isolated(ptr, q, nsize, remap)
            np.savez_compressed(os.path.join(OUT, f"gset_{name}_s{seed}.npz"), indptr=ptr, indices=idx, query=q,
                                M=M, m=m, bucket=bucket, seed=seed, nsize=nsize, remap=remap, enc=enc, raw=raw)
            # G3: the SpG that subg_matrix builds from these outputs (random_walks.py:79-81); rows = query
            # positions so that repeated roots stay separate rows.
            if qk != "dup":
                z = sp.csr_matrix((remap[1] + 1, (np.repeat(np.arange(len(q)), nsize), remap[0])),
                                  (len(q), n_nodes))
                assert z.has_sorted_indices
                encz = np.insert(enc, 0, np.zeros((1, m + 1)), axis=0)
                np.savez_compressed(os.path.join(OUT, f"spg_{name}_s{seed}.npz"), nsize=nsize, remap=remap, enc=enc,
                                    z_indptr=z.indptr.astype(np.int64), z_indices=z.indices.astype(np.int32),
                                    z_data=z.data.astype(np.int32), encz=encz)


WALK_CASES = {
    "tiny":     (50, 120, 0, 0, 0, False, 5, 2),
    "mixeddeg": (300, 900, 1, 3, 100, True, 20, 3),
    "dense":    (400, 16000, 3, 0, 0, False, 30, 2),
}


def gen_walk():
    for name, (N, E, gs, iso, star, k2, M, m) in WALK_CASES.items():
        ptr, idx = sym_graph(N, E, gs, iso, star, k2)
        q = np.arange(len(ptr) - 1)
        for rep in (False, True):
            for T in (1, 4):
                walks, obj = ref.walk_sampler(ptr, idx, q, num_walks=M, num_steps=m, nthread=T, seed=7,
                                              replacement=rep)
                nsize = np.array([len(obj[i, 0]) for i in range(len(q))], np.int32)
                ids = np.concatenate([obj[i, 0] for i in range(len(q))]).astype(np.int32)
                counts = np.concatenate([obj[i, 1] for i in range(len(q))]).astype(np.int32)
                np.savez_compressed(os.path.join(OUT, f"walk_{name}_r{int(rep)}_t{T}.npz"), indptr=ptr, indices=idx,
                                    query=q, M=M, m=m, seed=7, nthread=T, replacement=rep, walks=walks,
                                    nsize=nsize, ids=ids, counts=counts)


def dir_graph(N, E, seed, star=0):
    """Small DIRECTED simple graph (no G + G.T): about a third of the nodes end up without out-edges."""
    rng = np.random.default_rng(seed)
    r = rng.integers(0, (2 * N) // 3, E)          # only the first two thirds of the nodes have out-edges at all
    c = rng.integers(0, N, E)
    if star:
        r = np.concatenate([r, np.zeros(star, int)])
        c = np.concatenate([c, rng.choice(np.arange(1, N), star, replace=False)])
    A = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(N, N))
    A.sum_duplicates()
    A.setdiag(0)
    A.eliminate_zeros()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


DIRECTED_CASES = {
    # name: (N, E, gseed, star, M, m)
    "dirsparse": (300, 500, 11, 0, 20, 3),       # most walks end early
    "dirhub":    (400, 2400, 12, 120, 16, 4),    # a hub root with deg > M (shuffled first hop) in front of the dead ends
    "dircollab": (1200, 5000, 13, 260, 200, 2),  # the collab parameters
}


def gen_directed():
    for name, (N, E, gs, star, M, m) in DIRECTED_CASES.items():
        ptr, idx = dir_graph(N, E, gs, star)
        assert int((np.diff(ptr) == 0).sum()) > N // 4
        q = np.arange(N) if N <= 400 else np.sort(np.random.default_rng(gs).choice(N, 300, replace=False))
        nsize, remap, enc, raw = ref.gset_sampler(ptr, idx, q, num_walks=M, num_steps=m, bucket=-1, nthread=1, seed=111413, debug=1)
        remap = mask_isolated(ptr, q, nsize, remap)
        np.savez_compressed(os.path.join(OUT, f"gset_{name}_s111413.npz"), indptr=ptr, indices=idx, query=q, M=M, m=m, bucket=-1,
                            seed=111413, nsize=nsize, remap=remap, enc=enc, raw=raw)
        for rep in (False, True):
            for T in (1, 3):
                walks, obj = ref.walk_sampler(ptr, idx, q, num_walks=M, num_steps=m, nthread=T, seed=7, replacement=rep)
                nsz = np.array([len(obj[i, 0]) for i in range(len(q))], np.int32)
                ids = np.concatenate([obj[i, 0] for i in range(len(q))]).astype(np.int32)
                counts = np.concatenate([obj[i, 1] for i in range(len(q))]).astype(np.int32)
                np.savez_compressed(os.path.join(OUT, f"walk_{name}_r{int(rep)}_t{T}.npz"), indptr=ptr, indices=idx, query=q, M=M, m=m,
                                    seed=7, nthread=T, replacement=rep, walks=walks, nsize=nsz, ids=ids, counts=counts)


def random_spg(N, maxlen, c, seed, float_payload=False, empty_rows=()):
    """A synthetic SpG in the shape subg_matrix produces: sorted unique ids per row, data >= 1."""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for u in range(N):
        if u in empty_rows:
            continue
        ln = int(rng.integers(1, maxlen + 1))
        ids = np.sort(rng.choice(N, ln, replace=False))
        rows.append(np.full(ln, u))
        cols.append(ids)
        vals.append(rng.random(ln) * 0.9 + 0.1 if float_payload else rng.integers(1, c + 1, ln))
    z = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), (N, N))
    z.sort_indices()
    if not float_payload:
        z = z.astype(np.int32)
    return z


def gen_sjoin():
    N, c, k = 120, 40, 3
    rng = np.random.default_rng(11)
    enc_f = torch.from_numpy((rng.integers(0, 200, (c + 1, k)).astype(np.float32) / 200.0))
    enc_f[0] = 0
    for name, fl, empties in (("int", False, ()), ("int_emptyrows", False, (3, 17, 60)), ("float", True, ())):
        z = random_spg(N, 25, c, 21, fl, empties)
        e = rng.integers(0, N, (2, 64))
        e[:, 0] = (5, 5)          # identical endpoints
        e[:, 1] = (3, 17) if empties else (7, 9)
        e[:, 2] = (3, 8) if empties else (8, 7)
        enc = None if fl else enc_f
        out = {}
        for ptr in (True, False):
            xz, ind = ref_train.gather(e, z, "cpu", ptr=ptr, encode=enc)
            out[f"xz_ptr{int(ptr)}"] = xz.numpy()
            out[f"ind_ptr{int(ptr)}"] = ind.numpy()
            pxz, pind = ref_train.pgather(e, z, "cpu", enc, ref_train.bgather, ptr=ptr, njobs=4)
            assert torch.equal(pxz, xz) and torch.equal(pind, ind)   # pgather == gather
        np.savez_compressed(os.path.join(OUT, f"sjoin_{name}.npz"), z_indptr=z.indptr.astype(np.int64),
                            z_indices=z.indices.astype(np.int32), z_data=z.data, edge=e,
                            encode=(enc_f.numpy() if enc is not None else np.zeros((0, 0), np.float32)), **out)
    # G5: hgather (int payload only, train.py:69-70)
    z = random_spg(N, 25, c, 23, False, (4,))
    h = rng.integers(0, N, (3, 40))
    h[:, 0] = (4, 6, 6)
    xz, ind = ref_train.hgather(h, z, "cpu", encode=enc_f)
    np.savez_compressed(os.path.join(OUT, "hjoin_int.npz"), z_indptr=z.indptr.astype(np.int64),
                        z_indices=z.indices.astype(np.int32), z_data=z.data, hedge=h, encode=enc_f.numpy(),
                        xz=xz.numpy(), ind=ind.numpy())


def gen_walkjoin():
    """G7  walkjoin_*.npz  subg_acc.walk_join(return_idx=True) over walk_sampler's own output   subg_acc/subg_acc.c:509-647"""
    cases = {"small": (300, 900, 3, 120, 10, 3, 50), "wide": (800, 6000, 5, 400, 32, 2, 300), "dup": (200, 700, 9, 60, 6, 4, 40)}
    for name, (N, E, gs, n, M, m, Q) in cases.items():
        ptr, idx = sym_graph(N, E, gs)
        rng = np.random.default_rng(gs)
        roots = rng.permutation(N)[:n].astype(np.int32)
        if name == "dup":                       # repeated roots: the query resolves to the LAST row of a root
            roots = np.concatenate([roots, roots[:20]]).astype(np.int32)
        walks, obj = ref.walk_sampler(ptr, idx, roots, num_walks=M, num_steps=m, nthread=1, seed=7, replacement=True)
        q = roots[rng.integers(0, len(roots), (Q, 2))]
        q[0] = (roots[0], roots[0])             # (u, u)
        out, xrow = ref.walk_join(walks, list(obj[:, 0]), q, nthread=1, return_idx=True)
        key_len = np.array([len(obj[i, 0]) for i in range(len(roots))], np.int32)
        np.savez_compressed(os.path.join(OUT, f"walkjoin_{name}.npz"), walks=walks, key_len=key_len,
                            key_ids=np.concatenate(list(obj[:, 0])).astype(np.int32), query=q.astype(np.int32), out=out, xrow=xrow)


def gen_batch():
    """G8  batch_*.npz  subg_acc.batch_sampler   subg_acc/subg_acc.c:391-507"""
    cases = {  # name: (N, E, graph seed, star, n roots, M, S, thld, seed)
        "small": (300, 900, 3, 0, 24, 20, 4, 150, 7),
        "hub": (500, 1500, 5, 120, 16, 30, 5, 400, 11),        # root 0 has degree > M: the Fisher-Yates first hop
        "early": (400, 2400, 8, 0, 40, 50, 8, 60, 111413),      # a small threshold: most roots stop after one walk
        "all": (250, 1000, 2, 0, 10, 8, 3, 100000, 5),          # a threshold nobody reaches: every walk is taken
        "dup": (200, 800, 9, 0, 30, 12, 6, 120, 3),             # repeated roots
    }
    for name, (N, E, gs, star, n, M, S, thld, seed) in cases.items():
        ptr, idx = sym_graph(N, E, gs, star=star, isolated=2)
        rng = np.random.default_rng(gs)
        roots = rng.permutation(N)[:n].astype(np.int32)
        if name == "hub":
            roots[3] = 0
        if name == "dup":
            roots[10:20] = roots[:10]
        if name == "small":
            roots[5] = N          # an isolated node (degree 0): only the root is added (:457-460)
        out = ref.batch_sampler(ptr, idx, roots, num_walks=M, num_steps=S, thld=thld, seed=seed)
        np.savez_compressed(os.path.join(OUT, f"batch_{name}.npz"), indptr=ptr, indices=idx, query=roots, M=M, S=S, thld=thld,
                            seed_eff=np.uint32((seed + os.getpid()) & 0xFFFFFFFF), out=out.astype(np.int32))


def gen_rand_r():
    libc = ctypes.CDLL("libc.so.6")
    libc.rand_r.restype = ctypes.c_int
    out = {}
    for seed in (0, 1, 111413, 0xFFFFFFFF):
        st = ctypes.c_uint(seed)
        out[f"seed_{seed}"] = np.array([libc.rand_r(ctypes.byref(st)) for _ in range(1000)], np.uint32)
    np.savez_compressed(os.path.join(OUT, "rand_r.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    groups = {"rand_r": gen_rand_r, "gset": gen_gset, "walk": gen_walk, "sjoin": gen_sjoin, "walkjoin": gen_walkjoin,
              "batch": gen_batch, "directed": gen_directed}
    for g in (sys.argv[1:] or list(groups)):
        groups[g]()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {len(os.listdir(OUT))} fixtures, {tot / 1024:.0f} KiB -> {OUT}")
