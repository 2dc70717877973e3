#!/usr/bin/env python3
"""bench.py -- query-pairs/sec of the SubGAcc hot path (sample + SpJoin) on MI355X.

One "step" = one pass of the whole hot path over one batch of B synthetic query pairs that is already
resident in HBM:   2B endpoint roots -> walk + per-root dedup + LP counts (walk_sets) -> compaction ->
global unique LP rows -> SpG (segmented sort) -> Z_SF table -> SpJoin producing xz float32 [R,2,k] + indptr.
Nothing is cached between steps; every step gets fresh pairs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cit2|collab|ppa] [--pairs B]

N>1 is launched by the driver through torch.distributed.run (one rank per GPU); the graph is replicated, the
pair batches are sharded (each rank its own: "weak"; --scaling strong splits one fixed batch), there is no data-path
collective.  Rank 0 prints ONE JSON line, the last line of stdout, < 6 KB (compact_line(); the full record with every nested
block goes to bench_detail.json).  `roofline` is measured live with HIP events around the walk kernel on its
launch stream; `cpu_baseline` times the reference's own OpenMP sampler (oracle/_ref, compiled from the
reference sources in the build container) plus the oracle's C merge join on a bounded sample of the workload.
After the timed region (rank 0, 1 GPU): the random-line roof of this box (tools/line_roof_lib.hip), three more regions of 100
steps, and BASELINE.json's other configurations as passes of 3 x 100 steps (their medians are scalars of `config`).  `--full` adds the
side studies (the reference's offline-sample + resident-store-join flow, B = 1,024 loops, hgather, the mean stage, walk_sampler,
two-stream and root-dedup loops): minutes, bench_detail.json only.
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("SUBGACC_QUIET", "1")
os.environ.setdefault("OMP_PROC_BIND", "close")      # BASELINE.md section 3 (the CPU baseline's OpenMP teams); read when libgomp loads
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (graph preset, num_walks M, CLI num_steps k (walk hops m = k-1), description, fraction of positive pairs)
    "cit2": ("cit2", 200, 4, "cit2-like LP: N=2,927,963 avg-deg 20.7 power-law graph, M=200, --num_steps 4 (m=3 hops)", 0.5),
    # not a BASELINE config: the cit2 parameters on a graph WITH id locality (graphs.community_graph) -- does the walk's L2 miss
    # count respond to structure / to a work list sorted by root id?  (bench-only switch SUBGACC_SORT_ROOTS=0: batch order)
    "cit2loc": ("cit2loc", 200, 4, "cit2-like LP on a community-structured graph: N=2,927,963 avg-deg 20.7, blocks of 2,048 consecutive ids, "
                                   "75 % of the edges inside a block, M=200, --num_steps 4 (m=3 hops)", 0.5),
    # the paper's sampler figure (Fig. 6a: citation2, m = 4, M = 200 -- BASELINE.md section 1) read with m as the HOP count: CLI
    # --num_steps 5; the LP key is 4 x 8 + 1 = 33 bits, i.e. beyond the 32-bit key rows of the 2- and 3-hop configurations
    "cit2m4": ("cit2", 200, 5, "cit2-like LP, 4-hop walks: N=2,927,963 avg-deg 20.7 power-law graph, M=200, --num_steps 5 (m=4 hops)", 0.5),
    "collab": ("collab", 200, 3, "collab-like LP: N=235,868 avg-deg 8.2 power-law graph, M=200, --num_steps 3 (m=2 hops)", 0.5),
    # configs[2]: --k 20 negatives per positive (README.md:86) -> 1 pair in 21 is an edge of the graph, 20 are uniform pairs
    "ppa": ("ppa", 200, 4, "ppa-like LP: N=576,289 avg-deg 73.7 power-law graph, M=200, --num_steps 4 (m=3 hops), "
                           "1:20 positive:negative pairs (k=20)", 1.0 / 21.0),
    # configs[4]: twitter-follower scale (41.65 M nodes, ~2.9 B adjacency entries, int64 row offsets, 12 GB CSR
    # resident in HBM); the reference gives no walk parameters for it -- collab's are used
    "twitter": ("twitter", 200, 3, "twitter-like LP: N=41,652,230, G + G.T of ~1.47e9 follows as dataloader.py:122-135 hands it over (undirected, "
                                   "simple, rows sorted, ~2.9e9 adjacency entries, int64 indptr), M=200, --num_steps 3", 0.5),
    # rounds 1-3's stand-in for the same scale: directed, a multigraph, rows unsorted (generated without any global sort)
    "twitter_directed": ("twitter_directed", 200, 3, "twitter-like LP, DIRECTED multigraph stand-in: N=41,652,230, ~2.9e9 adjacency entries "
                                                     "(int64 indptr), M=200, --num_steps 3", 0.5),
    # configs[3]: SpJoin over the float SpG of the PPR encoder (the store is built by the GPU PPR sampler in set-up)
    "cit2ppr": (None, 0, 1, "cit2-like PPR encoder: float64 SpG = topk_ppr_matrix(alpha=0.1, eps=1e-4, top-100, 'sym') + encoding 'PPR' "
                            "over all N=2,927,963 nodes (built on the GPU in set-up), SpJoin only (train.py:39-43)", 0.5),
}
# line_probe.hip (tools/archive.tar.gz; profiles/r02_line_probe_pmc.csv) established that an L2 miss moves one whole 128-byte line whatever
# the access width and that the chip sustains ~55 G random lines/s from tables up to 1 GiB (Infinity Cache or HBM alike),
# ~48-50 G lines/s from 4-12 GiB.  This -- not bytes of useful data -- is what bounds a random walk.  The rate is measured
# again in every run, on the table the walk kernel reads (the hop records, else the adjacency array), right after the timed
# region (subgacc_line_probe: the study's `gather4` shape; box-to-box spread is 2-4 %).
LINE_BYTES = 128


def measure_line_roof(csr):
    """-> (random reads = 128-byte lines per second, table bytes, what the table is); best of three ~1-2 ms launches"""
    from surel_plus_amd._lib import ptr, stream_ptr
    probe = line_probe_lib()
    if probe is None:
        return None, None, None
    recs = csr.hop_records()
    table, what = (recs[0], "hop records") if recs else (csr.indices, "adjacency array")
    nbytes = table.numel() * table.element_size()
    sink = torch.zeros(1, dtype=torch.int32, device=table.device)
    reads = ctypes.c_int64(0)
    best = None
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if probe.subgacc_line_probe(ptr(table), nbytes, 32, 12345 + rep, ptr(sink), ctypes.byref(reads), stream_ptr()) != 0:
            return None, None, None
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b)
        if rep and (best is None or ms < best):      # (the first launch warms the TLBs)
            best = ms
    return reads.value / (best * 1e-3), nbytes, what


_PROBE = []


def line_probe_lib():
    """tools/build/libsubgacc_probe.so (tools/line_roof_lib.hip: a measurement aid OUTSIDE the product library and its ABI), built
    by __graft_entry__.build(); None when it is not there (the roof is then not reported)"""
    if not _PROBE:
        path = os.path.join(ROOT, "tools", "build", "libsubgacc_probe.so")
        try:
            L = ctypes.CDLL(path)
            vp = ctypes.c_void_p
            L.subgacc_line_probe.restype = ctypes.c_int
            L.subgacc_line_probe.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint32, vp, ctypes.POINTER(ctypes.c_int64), vp]
            _PROBE.append(L)
        except OSError:
            _PROBE.append(None)
    return _PROBE[0]


def kernel_source_sha():
    """identifies the kernels a PMC traffic figure belongs to (the GPU box has no .git): sha256 over the CODE of csrc + the header
    -- comments and white space are taken out first, so that a reworded comment does not orphan a measurement"""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "surel_plus_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "subgacc.h")]):
        text = open(f, "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # block comments
        text = re.sub(r"//[^\n]*", " ", text)                       # line comments (no string literal of these sources holds "//")
        h.update(os.path.basename(f).encode() + b"\0" + " ".join(text.split()).encode())
    return h.hexdigest()[:16]


HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# SUBGACC_FUSED: 1 = the walk kernel also emits finished SpG rows (walk_spg), 0 = general pipeline, unset = the
# library's choice (fused for walks of >= 3 hops)
FUSED = {"1": True, "0": False}.get(os.environ.get("SUBGACC_FUSED", ""), None)
# join the batch from its strided rows (no packed CSR copy of a batch that is joined once and dropped): unset = yes, rows
# from the fused-row walk kernel or from the general one + finish_rows (spg.prefers_fused); SUBGACC_STRIDED=0 builds the CSR SpG per step
STRIDED = {"1": True, "0": False}.get(os.environ.get("SUBGACC_STRIDED", ""), None)
UNIQ_CAPACITY = 1 << int(os.environ.get("SUBGACC_UNIQ_LOG2", "17"))   # slots of the table of distinct LP rows (a batch holds ~10^3)
LAZY = os.environ.get("SUBGACC_LAZY", "1") == "1"     # sizes stay on the device: one host round trip per step


@contextlib.contextmanager
def quiet_stdout():
    """The reference extension printf()s statistics; keep them off the JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)      # the C stdio buffer of the extension's printf
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)


class KernelTimer:
    """HIP events on the launch stream around one kernel (torch's current stream IS the launch stream: the
    C ABI receives torch.cuda.current_stream().cuda_stream)."""

    def __init__(self):
        self.pairs = {}
        self.enabled = False
        self.every = 1          # n: only every n-th launch of a name is bracketed (an event pair costs the GPU ~5 us around the kernel)
        self._seen = {}

    @contextlib.contextmanager
    def __call__(self, name):
        if not self.enabled:
            yield
            return
        if self.every > 1:
            k_ = self._seen.get(name, 0)
            self._seen[name] = k_ + 1
            if k_ % self.every:
                yield
                return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        yield
        b.record()
        self.pairs.setdefault(name, []).append((a, b))

    def mean_ms(self, name):
        ev = self.pairs.get(name, [])
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None, len(ev)


_XZ_BUF = {}
_STEP_BUFS = {}
BUFFERED = os.environ.get("SUBGACC_STEP_BUFFERS", "1") == "1"     # 0: the general (allocating) form of the step, for A/B
DEDUP = os.environ.get("SUBGACC_DEDUP_ROOTS", "0") == "1"        # 1: every distinct endpoint of a batch sampled once (Philox)


def hot_path_step(sp, csr, edge, M, k, seed, rng, slot=0):
    """Queue one pass of the hot path: sample both endpoints of every pair, build the SpG, join.  Nothing here waits
    for the GPU when LAZY (every size stays on the device); finish_step() reads the sizes / status back.
    Returns (xz buffer view, indptr, sets)."""
    B = edge.shape[1]
    strided = STRIDED     # None: the library's choice (a transient batch is joined in place from its strided rows)
    # the join output is written into re-used buffers sized for the worst case (every set full), two of them in
    # turn so that step s+1 never overwrites what step s handed out: a serving loop would do the same, and it keeps
    # GB-sized device allocations -- tens of ms on some hosts -- out of the steps
    cap = 2 * B * (M * (k - 1) + 1) * 2 * k
    buf = _XZ_BUF.get((edge.device, cap, slot))
    if buf is None:
        buf = _XZ_BUF[(edge.device, cap, slot)] = torch.empty(cap, dtype=torch.float32, device=edge.device)
    # the library's on-demand entry point: sample both endpoints of every pair -> SpG rows -> join.  A serving loop
    # hands it preallocated StepBuffers (two sets in turn, like the output buffers): six launches per step, no allocation
    bufs = None
    if LAZY and BUFFERED and strided is not False and FUSED is not False and not (DEDUP and rng != "philox"):
        key = (edge.device, B, M, k, slot, DEDUP, rng)
        bufs = _STEP_BUFS.get(key)
        if bufs is None:
            try:
                bufs = _STEP_BUFS[key] = sp.StepBuffers(csr, B, M, k - 1, uniq_capacity=UNIQ_CAPACITY, out=buf, dedup_roots=DEDUP, rng=rng,
                                                        sort_roots=os.environ.get("SUBGACC_SORT_ROOTS", "1") == "1",
                                                        align_rows=os.environ.get("SUBGACC_ALIGN_ROWS", "1") == "1")     # (bench-only switches)
            except ValueError:
                bufs = _STEP_BUFS[key] = False
    xz, ind, sets = sp.sample_and_gather(csr, edge, num_walks=M, num_steps=k - 1, seed=seed, rng=rng, out=buf if LAZY else None,
                                         lazy=LAZY, strided=strided, fused=FUSED, uniq_capacity=UNIQ_CAPACITY,
                                         **({"buffers": bufs} if bufs else {}))
    if LAZY:    # sizes + status + the join's row count start their way to pinned host memory now, behind this step
        sets.prefetch(extra=ind[-1:])
    return xz, ind, sets


def finish_step(xz, ind, sets):
    """sizes, status flags and the distinct-row count of a queued step: two small reads, errors raised here"""
    sets.resolve()
    if LAZY:
        xz = xz[: int(sets.extra[0])]
    return xz, ind, sets


def algorithmic_walk_bytes(csr, roots, sets, M, m):
    """SURVEY.md 8(d), per root: 4 (query) + 8 (indptr pair) + 4*min(deg,M) (first hop) + 12*M*(m-1) (later
    steps: 8 B indptr pair + 4 B neighbour) + 8*|S_r| (id + SFptr written) + 4 (nsize); int64 offsets: 8->16, 12->20."""
    ip = csr.indptr
    deg = (ip[roots.long() + 1] - ip[roots.long()]).long()
    w = 16 if csr.indptr64 else 8
    live = deg > 0
    per = 4 + w + 4 * torch.clamp(deg, max=M) + (w + 4) * M * (m - 1) * live.long() + 4
    return int(per.sum().item()) + 8 * int(sets.X)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(csr, edge_all, M, k, target_s=2.5, repeats=2, teams=(-1, 64, 32, 16, 8, 1), with_t1=True):
    """BASELINE.md section 3's protocol, on this host: the reference (oracle/_ref: the real subg_acc.gset_sampler, OpenMP) + the
    oracle's C merge join on a bounded number of pairs of the same workload, at 1 thread, 16 threads and the fastest team size
    of a probe (all cores, 64, 32, 16, 8 -- the reference shares one rand_r state between its threads, subg_acc.c:731-732, and
    slows down with many of them), OMP_PROC_BIND=close (set before libgomp loads, top of this file), best of `repeats` per setting
    (the 1-thread setting runs once), each run sized to ~target_s seconds: ~15-20 s of CPU work in all.  Rank 0, N=1 only.
    `value` = the fastest full-size run; `cores` = the team size of THAT run."""
    import oracle
    t_begin = time.perf_counter()
    ref = oracle.ref_module()
    ptr_h = csr.indptr.cpu().numpy()
    idx_h = csr.indices.cpu().numpy()
    cores = os.cpu_count() or 1
    threads = oracle.num_threads()
    use_ref = ref is not None and ptr_h.dtype == np.int32

    def run(B, nthread):
        e = edge_all[:, :B].cpu().numpy()
        roots = e.reshape(-1).astype(np.int32)
        t0 = time.perf_counter()
        with quiet_stdout():
            if use_ref:
                nsize, remap, enc = ref.gset_sampler(ptr_h, idx_h, roots, num_walks=M, num_steps=k - 1, nthread=nthread)
            else:
                nsize, remap, enc = oracle.gset_sampler(ptr_h, idx_h, roots, num_walks=M, num_steps=k - 1, rng="philox",
                                                        nthreads=threads if nthread < 0 else nthread)
        t1 = time.perf_counter()
        jt = threads if nthread != 1 else 1
        spg = oracle.spg_build(nsize, remap, nthreads=jt)
        table = oracle.enc_table(enc).astype(np.float32) / np.float32(M)
        rows = np.arange(2 * B, dtype=np.int64).reshape(2, B)
        xz, ind = oracle.gather(rows, spg, ptr=True, encode=table, nthreads=jt)
        t2 = time.perf_counter()
        return t2 - t0, t1 - t0

    Bmax = edge_all.shape[1]
    B0 = min(2048, Bmax)          # (512 pairs were too few: the call's fixed costs decided the probe, not the team size)
    probe = {}
    for nt in (teams if use_ref else [-1, 1]):
        if nt > cores:
            continue
        probe[nt] = run(B0 if nt != 1 else min(512, Bmax), nt)[0] * (1 if nt != 1 else B0 / min(512, Bmax))
    probe_best = min((nt for nt in probe if nt != 1), key=lambda nt: probe[nt])
    # the full-size runs: 1 thread, 16 threads and -- when it is another one -- the probe's pick; `value`
    # is the fastest of THESE runs and `cores` / `best_nthread` name the team that produced it (the probe only chooses what to run)
    settings = {}
    for label, nt in (("t1", 1), ("t16", 16 if (16 in probe or not use_ref) else None), ("probe_best", probe_best)):
        if nt is None or nt not in probe or (label == "probe_best" and any(v["nthread_arg"] == nt for v in settings.values())):
            continue
        if label == "t1" and not with_t1:
            continue
        Bn = int(min(Bmax, max(B0, B0 * target_s / max(probe[nt], 1e-6))))
        runs = [run(Bn, nt) for _ in range(1 if nt == 1 else repeats)]
        t, ts = min(runs)
        settings[label] = {"nthread": cores if nt < 0 else nt, "nthread_arg": nt, "pairs": Bn, "runs": len(runs), "pairs_per_s": Bn / t,
                           "sampler_roots_per_s": 2 * Bn / ts, "join_pairs_per_s": Bn / max(t - ts, 1e-9), "seconds_best": t, "sampler_seconds": ts}
    b = max(settings.values(), key=lambda v: v["pairs_per_s"])      # `value`: the fastest of the full-size settings
    kind = "reference" if use_ref else "port"
    sampler_name = (f"the reference's subg_acc.gset_sampler (oracle/_ref), nthread={b['nthread']}" if use_ref else f"oracle C port, {threads} threads")
    out = {"value": b["pairs_per_s"], "unit": "query-pairs/s", "cores": b["nthread"] if use_ref else threads, "kind": kind,
           "cpu_model": cpu_model(), "host_threads": cores, "omp_proc_bind": os.environ.get("OMP_PROC_BIND"),
           # the two halves, for the reference's own flow (offline_flow): sampler roots/s, and pairs/s of SpG build + join
           "sampler_roots_per_s": b["sampler_roots_per_s"], "join_pairs_per_s": b["join_pairs_per_s"],
           "t1_pairs_per_s": settings["t1"]["pairs_per_s"] if "t1" in settings else None,
           "t16_pairs_per_s": settings["t16"]["pairs_per_s"] if "t16" in settings else None,
           "best_nthread": b["nthread"], "probe_best_nthread": cores if probe_best < 0 else probe_best,
           "cpu_seconds": time.perf_counter() - t_begin, "settings": settings,
           "probe_seconds_2048_pairs": {str(cores if nt < 0 else nt): round(v, 4) for nt, v in probe.items()},
           "sample": f"{b['pairs']} pairs ({2 * b['pairs']} roots) of the workload, best of {b['runs']}: {sampler_name} + oracle C SpG build "
                     f"and merge join; fastest of the 1 / 16 / probe-best team sizes"}
    return out


def literal_dropin(csr, M, k, n_cpu=32768, nthread_cpu=-1):
    """What seam A alone buys: the body of the reference's subg_matrix, random_walks.py:77-81, run
    literally -- NumPy arrays in, `gset_sampler`, scipy csr_matrix of the returned remap, np.insert of the zero row -- once over
    this repo's drop-in module `subg_acc` (GPU inside, PCIe + NumPy hand-over included) for ALL N roots, and once over the compiled
    reference (oracle/_ref) for the first n_cpu roots.  roots/s each."""
    import scipy.sparse as sps
    import oracle
    import subg_acc as shim
    ptr_h, idx_h = csr.indptr.cpu().numpy(), csr.indices.cpu().numpy()
    N = csr.num_nodes

    def body(mod, idx, **kw):
        t0 = time.perf_counter()
        with quiet_stdout():
            nsize, remap, enc = mod.gset_sampler(ptr_h, idx_h, idx, num_walks=M, num_steps=k - 1, **kw)
        t1 = time.perf_counter()
        z = sps.csr_matrix((remap[1] + 1, (np.repeat(idx, nsize), remap[0])), shape=(N, N))
        assert z.has_sorted_indices
        enc = np.insert(enc, 0, np.zeros((1, k), dtype=enc.dtype), axis=0)
        t2 = time.perf_counter()
        return t2 - t0, t1 - t0, int(remap.shape[1])
    idx = np.arange(N, dtype=np.int32)
    body(shim, idx[:4096])
    t_gpu, t_shim, members = min(body(shim, idx) for _ in range(3))
    out = {"roots": N, "members": members, "drop_in_seconds": t_gpu, "drop_in_roots_per_s": N / t_gpu,
           # the two halves of the seam: inside gset_sampler (ours: upload, kernels, hand-over of 8 bytes per member) and the
           # caller's own lines (scipy's COO -> CSR sort of X members, np.insert) that no drop-in can touch
           "shim_seconds": t_shim, "shim_roots_per_s": N / t_shim, "caller_seconds": t_gpu - t_shim,
           "what": "random_walks.py:77-81 verbatim over `import subg_acc` of this repo: numpy in / numpy out, scipy csr_matrix, np.insert"}
    ref = oracle.ref_module()
    if ref is not None and ptr_h.dtype == np.int32:
        n = min(n_cpu, N)
        t_cpu, t_cpu_call, _ = min(body(ref, idx[:n], nthread=nthread_cpu) for _ in range(2))
        out.update({"reference_roots": n, "reference_seconds": t_cpu, "reference_roots_per_s": n / t_cpu,
                    "reference_call_roots_per_s": n / t_cpu_call, "reference_nthread": nthread_cpu})
    return out


_PPR_STORE = {}


def bench_ppr(args, sp, sampler_mod, dev, rank, world, dist, desc, B, K, W, extra_regions=0, layout="aligned"):
    """SpJoin over a resident float-payload SpG (the citation2 PPR configuration): one step = B pairs -> xz [R,2,1].
    layout: "aligned" (round 6, the serving layout: SpG.aligned() -- rows on whole 128-byte lines at a fixed pitch, their lengths in
    their first slots, no row pointers) or "packed" (the CSR the offline stage leaves, as rounds 1-5 joined it)."""
    from surel_plus_amd.graphs import ppr_like_spg, preset_graph
    key = (os.environ.get("SUBGACC_PPR_SYNTH", "0"), args.scale, str(dev))
    if key in _PPR_STORE:                                    # (the store of the pass before: the offline stage is not what is timed)
        zp, prep_s = _PPR_STORE[key]
        N = zp.n_rows
    elif os.environ.get("SUBGACC_PPR_SYNTH", "0") == "1":    # stand-in payload: exactly 100 random ids per row
        N = max(int(2_927_963 * args.scale), 1000)
        zp, prep_s = ppr_like_spg(N, 100, seed=3, device=dev), None
    else:                                                    # the real offline stage (main.py:181-183), not timed
        from surel_plus_amd.ppr import topk_ppr_matrix
        csr = preset_graph("cit2", device=dev, scale=args.scale)
        N = csr.num_nodes
        torch.cuda.synchronize()
        t_prep = time.perf_counter()
        zp = topk_ppr_matrix(csr, 0.1, 1e-4, torch.arange(N, dtype=torch.int32, device=dev), 100, normalization="sym", encode=True)
        torch.cuda.synchronize()
        prep_s = time.perf_counter() - t_prep
        del csr
    _PPR_STORE.clear()
    _PPR_STORE[key] = (zp, prep_s)
    members = zp.nnz
    z = zp.aligned() if layout == "aligned" else zp
    store_bytes = z.nbytes if layout == "aligned" else (zp.indptr.numel() * 8 + members * 12)
    import gc
    gc.collect()          # (the previous pass's garbage: collected here, not by a pause inside this pass's regions -- see bench_lp)
    torch.cuda.synchronize()
    gens = [torch.Generator(device=dev).manual_seed(1000 * rank + s) for s in range(min(K + W, 103))]
    all_edges = [torch.randint(0, N, (2, B), device=dev, generator=g) for g in gens]

    class _Cyclic:       # (a long pass re-uses its ~100 resident batches in turn)
        def __getitem__(self, i):
            return all_edges[i % len(all_edges)]
    edges = _Cyclic()
    timer = KernelTimer()
    sampler_mod.KERNEL_TIMER = timer         # spjoin brackets the fill kernel ("sjoin_fill"); "join" below = the whole call
    # a serving loop's form of the join: stepgraph.CapturedJoin (two in turn) -- buffers and descriptor built once, a step is ONE call
    # into the library (the size pass as a single launch, the fill behind it), the row count and the status word arrive in pinned
    # host memory by themselves; gather(lazy=True), launch by launch, is host-bound (0.17 ms of Python for ~0.09 ms of kernels)
    eager = os.environ.get("SUBGACC_PPR_EAGER", "0") == "1"
    bufs = [torch.empty(2 * B * z.max_len * 2, dtype=torch.float32, device=dev) for _ in (0, 1)] if eager else None
    # ... two of them in turn, each on its own stream (CapturedJoinPool): the size pass and the first waves of a batch's fill run
    # under the last waves of the batch before it
    pool = None if eager else sp.CapturedJoinPool(z, B, lanes=2)

    def step(s):
        if eager:
            with timer("join"):
                return sp.gather(edges[s], z, dev, ptr=True, encode=None, out=bufs[s & 1], lazy=True)
        return pool.submit(edges[s], sync=False)          # (the batches were made before the region and synchronised)

    def resolve(q):
        if eager:
            import surel_plus_amd.spjoin as sj
            sj.lazy_join_status(q[1])
            return int(q[1][-1].item())
        return int(pool.finish(q)[0].shape[0])
    for s in range(W):
        resolve(step(s))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    pending = None
    for s in range(W, W + K):
        cur = step(s)
        if pending is not None:
            rows_out = resolve(pending)       # (the previous step is resolved while this one runs)
        pending = cur
    rows_out = resolve(pending)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    region_values = [world * B * K / elapsed]
    for r_ in range(extra_regions if (world == 1 and rank == 0) else 0):      # more regions of K steps, outside the clock
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pending = None
        for s in range(W + r_, W + r_ + K):
            cur = step(s)
            if pending is not None:
                resolve(pending)
            pending = cur
        resolve(pending)
        torch.cuda.synchronize()
        region_values.append(B * K / (time.perf_counter() - t1))
    if not eager:      # one call at a time, alone on the GPU, between HIP events: what a whole join call takes (size pass + fill)
        timer.enabled = True
        for s in range(W, W + 10):
            with timer("join"):
                pool.steps[0](edges[s])
            pool.steps[0].finish()
        timer.enabled = False
    sampler_mod.KERNEL_TIMER = None
    elapsed_local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rank_records = gather_rank_records(dist, world, rank, dev, elapsed_local, timer.mean_ms("join")[0],
                                       os.environ.get("SUBGACC_DIST_BACKEND", "nccl") if dist is not None else None)
    require_distinct_devices(rank_records, world, os.environ.get("SUBGACC_DIST_BACKEND", "nccl") if dist is not None else None)
    if rank != 0:
        return None
    if extra_regions:
        elapsed = world * B * K / median(region_values)
    call_ms, launches = timer.mean_ms("join")       # HIP events around a whole join call alone on the GPU: the size pass and the fill
    ms, _ = timer.mean_ms("sjoin_fill")             # the fill kernel alone (eager mode only: events do not time inside a replayed graph)
    from surel_plus_amd import _lib as lib_
    xz_e, ind_e = sp.gather(edges[W], z, dev, ptr=True, encode=None, out=torch.empty(2 * B * z.max_len * 2, dtype=torch.float32, device=dev), lazy=True)
    own_e = edges[W].contiguous().view(-1)

    def fill_alone():       # the fill kernel of one batch, sizes known: one launch
        lib_.join_fill(lib_.JOIN_ROWS, lib_.JOIN_F64, n_rows=z.n_rows, payload=z.data, own=own_e, S=own_e.numel(), seg=ind_e, pair_block=B,
                       out_xz=xz_e, flags=ind_e.join_flags,
                       **(dict(row_stride=z.pitch, ids=z.ids) if layout == "aligned" else dict(row_off=z.indptr, ids=z.indices, max_len=z.max_len)))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if ms is None and layout == "aligned":
        # a lazy gather from a headed store is ONE library call (size pass + fill, what join_graph_ms times): the fill alone is
        # launched here, on the segments of one of the timed batches, one launch per event pair
        one = []
        for _ in range(6):
            ev0.record()
            fill_alone()
            ev1.record()
            torch.cuda.synchronize()
            one.append(ev0.elapsed_time(ev1))
        fill_ms = sum(one[1:]) / len(one[1:])
        ms, ms_source = fill_ms, "HIP events around single launches of the fill kernel on one batch of the timed region, sizes known (after the region)"
    elif ms is None:                                # ... so a few eager joins after the timed region time it, for the record
        ebuf = torch.empty(2 * B * z.max_len * 2, dtype=torch.float32, device=dev)
        timer.pairs.pop("sjoin_fill", None)
        sampler_mod.KERNEL_TIMER = timer
        timer.enabled = True
        for s in range(W, W + min(K, 5)):
            sp.gather(edges[s], z, dev, ptr=True, encode=None, out=ebuf, lazy=True)
        torch.cuda.synchronize()
        timer.enabled = False
        sampler_mod.KERNEL_TIMER = None
        fill_ms, _ = timer.mean_ms("sjoin_fill")
        ms, ms_source = fill_ms, ("HIP events around the fill kernel of gather(lazy=True) launches of the same joins right after the timed region "
                                  "(the timed region's call holds the size pass and the fill: join_graph_ms is that whole call)")
    else:
        fill_ms, ms_source = ms, "HIP events around the fill kernel in the timed region"
    # an event pair around ONE launch of a ~60 us kernel reads 4-6 us more than the kernel runs (rocprofv3's kernel trace of the
    # same launches is the arbiter: profiles/rNN_cit2ppr_kernel_stats.csv): the same fill ten times back to back between one pair
    b2b = []
    for _ in range(3):
        ev0.record()
        for _ in range(10):
            fill_alone()
        ev1.record()
        torch.cuda.synchronize()
        b2b.append(ev0.elapsed_time(ev1) / 10)
    b2b_ms = median(b2b)
    steady_ms = world * B / median(region_values) * 1e3          # the loop's period per call (per rank)
    abytes = B * 64 + rows_out * (12 + 8)      # 64 + (|S_u|+|S_v|) * (12 read: id + f64 payload, 8 written: f32 [.,2,1])
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath)).get(f"cit2ppr:{B}:join" + (":aligned" if layout == "aligned" else ""), {})
        if tj.get("kernel_source_sha") == kernel_source_sha():
            traffic = tj.get("join_hbm_bytes_per_launch")
    return {"metric": "query-pairs/sec (SpJoin, PPR payload)", "value": world * B * K / elapsed, "unit": "query-pairs/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "pairs_per_step_per_gpu": B, "pairs_per_step_all_gpus": world * B, "xz_rows_last_step": rows_out,
                       # one record per rank (all-gathered), as for the LP workloads: which devices really took part
                       "ranks_seen": len(rank_records), "distinct_devices": len({r["device"] for r in rank_records}),
                       "per_rank_ms_min": min(r["elapsed_ms"] for r in rank_records) / max(K, 1),
                       "per_rank_ms_max": max(r["elapsed_ms"] for r in rank_records) / max(K, 1),
                       "dist_backend": (os.environ.get("SUBGACC_DIST_BACKEND", "nccl") if dist is not None else "none"),
                       "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if dist is not None else None,
                       "rank_records": rank_records, "rng": "none (join only)", "parallelism": f"query-shard x{world}",
                       "region_pairs_per_s": [round(v) for v in region_values], "pairs_per_s_min": min(region_values),
                       "pairs_per_s_median": median(region_values), "pairs_per_s_max": max(region_values),
                       "spg_members": members, "offline_ppr_stage_s": prep_s, "store_layout": layout, "store_bytes": store_bytes,
                       "store_bytes_packed": zp.indptr.numel() * 8 + members * 12,
                       "join_call_ms": call_ms,          # (two launches: the single-pass size kernel, the fill; + the event pair)
                       # ... and what a call costs in the loop (two joins in turn on two streams): the period of the timed regions
                       "join_call_ms_steady": steady_ms,
                       "frac_of_hbm_peak_whole_join_call": (abytes / ((steady_ms or call_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS) if (steady_ms or call_ms) else None},
            # `frac` / `achieved`: the fill's AVERAGE launch duration = ten launches between one pair of HIP events / 10.  An event pair
            # around ONE launch of this ~58 us kernel reads 5-7 us more than the kernel runs: rocprofv3's kernel trace of the same
            # fill (profiles/r70_cit2ppr_kernel_stats.csv: 58.8 us without the first, cold launch) sides with the average of ten
            # (57.8 us in that call), not with the single-launch reading (64.8 us), which stays beside it as *_single_launch
            "roofline": {"bound": "hbm", "kernel": "sjoin_f64pair_kernel<64> (one wave per pair)",
                         "achieved": abytes / (b2b_ms * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": abytes / (b2b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": b2b_ms,
                         "kernel_ms_source": "ten launches of the fill kernel back to back between one pair of HIP events, / 10; median of 3",
                         "kernel_ms_single_launch": ms, "kernel_ms_single_launch_source": ms_source,
                         "frac_single_launch": abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_x10_back_to_back": b2b_ms, "frac_x10_back_to_back": abytes / (b2b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "join_graph_ms": call_ms, "frac_whole_join": (abytes / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if call_ms else None,
                         "launches_timed": launches, "algorithmic_bytes_per_launch": abytes}}


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def bench_lp(args, name, rng, B, K, W, sp, sampler_mod, dev, rank, world, dist, with_cpu_baseline, csr_variant=True,
             small_batches=False, two_stream_extra=True, dedup_extra=None, offline=False, extra_regions=(0, 0), value_is_median=False, captured=False,
             cpu_kw={}, wake_s=0.0, host_fed=False, ref_flow=False):
    """sample + SpJoin over one LP workload: W warm-up steps, K timed steps bracketed by barrier + synchronize, max over
    ranks.  Returns the JSON object (rank 0) or None.
    extra_regions = (R, Kx): after the timed region, R more regions of Kx steps each in the same loop (rank 0, 1 GPU): their
    pairs/s as min / median / max -- the timed region of the driver's K steps is ~25 ms, one host hiccup away from any number;
    value_is_median: `value` is the median over the main region and the extras (the short passes over the other workloads).
    captured: the timed loop replays the whole step as ONE HIP graph (stepgraph.CapturedStep, three in turn: the read-back of step s
    is waited for after step s+2 has been queued) -- what a serving loop does for steps whose kernels (~0.4 ms: 2-hop walks,
    small batches) are shorter than the host's six launches and one read-back on a busy box; the per-kernel times then come from a
    few eager steps after the regions (HIP events cannot bracket a kernel inside a replayed graph)."""
    global STRIDED, DEDUP
    from surel_plus_amd.graphs import preset_graph, query_pairs
    preset, M, k, desc, pos_frac = WORKLOADS[name]
    # what the previous pass left behind (hundreds of HIP events, step buffers, its graph) is collected NOW, not by a collector
    # pause somewhere inside this pass's timed region (r19b: the first pass after the headline lost ~35 ms that way, all of it
    # between an event and the launch it brackets -- its walk kernel "took" 1.5x its time)
    import gc
    gc.collect()
    torch.cuda.synchronize()
    if name.startswith("twitter"):      # 12 GB of CSR (+ transients of its generation) + ~8 GB of step buffers per rank: look before building
        free, total = torch.cuda.mem_get_info(dev)     # (the 47 GB of hop records are the library's call: DeviceCSR.hop_records
        need = int(24e9 * args.scale) + int(8e9)       #  builds them only within a quarter of the free memory)
        if free < need:
            raise RuntimeError(f"twitter workload: {free >> 30} GiB free of {total >> 30} GiB on {dev}, the graph and the step buffers need {need >> 30} GiB")
    csr = preset_graph(preset, device=dev, scale=args.scale)
    if os.environ.get("SUBGACC_DEGREE_ORDER") == "1":     # dev experiment: nodes renumbered hub-first
        from surel_plus_amd.graphs import degree_ordered
        csr = degree_ordered(csr)[0]
    # every step's pairs are resident in HBM before the clock starts; ranks and steps get different pairs
    edges = [query_pairs(csr, B, seed=1000 * rank + s, device=dev, pos_frac=pos_frac) for s in range(max(K + W, min(extra_regions[1], 100)))]
    _XZ_BUF.clear()
    _STEP_BUFS.clear()
    timer = KernelTimer()
    sampler_mod.KERNEL_TIMER = timer
    last = None
    step_marks = []
    # consecutive steps on alternating HIP streams (a serving loop may do that: the batches are independent).  Not what
    # the timed region does -- its kernels run back to back on one stream, so that the HIP events around the walk kernel
    # measure that kernel alone -- but reported next to it (config.two_stream_loop), outside the clock.
    STREAMS = [torch.cuda.Stream(device=dev) for _ in range(2)] if os.environ.get("SUBGACC_STREAMS") == "2" else None

    def run_steps(step_ids):
        """Double-buffered serving loop: step s is queued before the sizes of step s-1 are read back (they travel to
        pinned host memory behind step s-1's kernels), so the GPU never waits for the host; every step is complete
        (kernels done, sizes on the host, errors raised) when this returns."""
        nonlocal last
        pending = None
        for s in step_ids:
            e = edges[s % len(edges)]
            if STREAMS is not None:      # SUBGACC_STREAMS=2 (experiment): consecutive steps on alternating streams
                with torch.cuda.stream(STREAMS[s & 1]):
                    queued = hot_path_step(sp, csr, e, M, k, seed=s, rng=rng, slot=s & 1)
            else:
                queued = hot_path_step(sp, csr, e, M, k, seed=s, rng=rng, slot=s & 1)
            if pending is not None:
                xz, ind, sets = finish_step(*pending[1])
                last = (pending[0], sets, xz)
            pending = (e, queued)
            step_marks.append(time.perf_counter())
        if pending is not None:
            xz, ind, sets = finish_step(*pending[1])
            last = (pending[0], sets, xz)

    caps = None
    if captured:
        caps = [sp.CapturedStep(csr, B, num_walks=M, num_steps=k - 1, seed=1, rng=rng, uniq_capacity=UNIQ_CAPACITY) for _ in range(3)]
    eager_run_steps = run_steps

    def run_steps_captured(step_ids):
        pend = []
        for s in step_ids:
            pend.append(caps[s % 3](edges[s % len(edges)]))
            if len(pend) == 3:        # the sizes / status of step s-2 are read now, two steps behind the queue
                pend.pop(0).finish()
            step_marks.append(time.perf_counter())
        for q in pend:
            q.finish()
    if captured:
        run_steps = run_steps_captured

    # Priming (part of set-up, like building the graph) + the W warm-up steps, in the same loop shape as the timed
    # region so that torch's caching allocator reaches its steady state (two steps in flight) here: a fresh GB-sized
    # hipMalloc inside the timed region costs ~10 ms on some hosts of the pool and is not part of the path.
    PRIME = 3
    for slot in (0, 1):       # the two output buffers first, so that everything allocated per step settles around them
        cap = 2 * B * (M * (k - 1) + 1) * 2 * k
        _XZ_BUF[(dev, cap, slot)] = torch.empty(cap, dtype=torch.float32, device=dev)
    run_steps(list(range(PRIME)) + list(range(W)))
    torch.cuda.synchronize()
    # a pass that follows seconds of host-only work (the CPU baseline) finds the GPU clocked down: its first region read 36 % low and
    # the HIP events of its first launches doubled the kernel's mean.  Part of set-up, outside every clock: steps until wake_s is over
    # (the clocks come back over a second or more: steps in chunks of 20 until three chunks in a row agree within 2 %, 3 s at most)
    t_wake, chunks = time.perf_counter(), []
    while wake_s > 0 and time.perf_counter() - t_wake < 3.0:
        t_c = time.perf_counter()
        run_steps(range(W, W + 20))
        torch.cuda.synchronize()
        chunks.append(time.perf_counter() - t_c)
        if time.perf_counter() - t_wake >= wake_s and len(chunks) >= 3 and max(chunks[-3:]) <= 1.02 * min(chunks[-3:]):
            break
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    # the kernels of every 4th step are bracketed by HIP events (at least 5 steps of the region): the four event records of a step
    # cost the GPU ~22 us, 2 % of the step they measure (the kernel trace, profiles/r31_graph_gaps_cit2.txt)
    timer.every = 4 if K >= 20 else 1
    allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    t0 = time.perf_counter()
    step_marks[:] = [t0]
    run_steps(range(W, W + K))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    timer.every = 1
    allocs_timed = torch.cuda.memory_stats().get("num_device_alloc", 0) - allocs0
    host_steps = [b - a for a, b in zip(step_marks, step_marks[1:])]
    elapsed_local = elapsed
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    region_values = [world * B * K / elapsed]
    if rank == 0 and world == 1 and extra_regions[0] > 0 and extra_regions[1] > 0:
        timer.enabled = False
        for r_ in range(extra_regions[0]):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(range(W + r_, W + r_ + extra_regions[1]))
            torch.cuda.synchronize()
            region_values.append(B * extra_regions[1] / (time.perf_counter() - t1))
    if captured:       # the stage times and the last step's sizes: eager launches of the same steps, outside the clock
        run_steps = eager_run_steps
        timer.enabled = False
        run_steps(range(3))
        torch.cuda.synchronize()
        timer.enabled = True
        run_steps(range(W, W + min(K, 10)))
        torch.cuda.synchronize()
        timer.enabled = False
    rank_records = gather_rank_records(dist, world, rank, dev, elapsed_local, timer.mean_ms("walk_sets")[0],
                                       os.environ.get("SUBGACC_DIST_BACKEND", "nccl") if dist is not None else None)
    require_distinct_devices(rank_records, world, os.environ.get("SUBGACC_DIST_BACKEND", "nccl") if dist is not None else None)

    # what the JSON line says about the last timed step, taken NOW: the extra passes below re-use the step buffers
    last_members = last_rows = last_distinct = last_abytes = None
    if rank == 0 and last is not None:
        last_members, last_rows = int(last[1].X), int(last[2].shape[0])
        last_distinct = int(last[1].c)
        last_abytes = algorithmic_walk_bytes(csr, last[0].reshape(-1), last[1], M, k - 1)   # the last timed launch
    host_fed_rate = None
    if host_fed and rank == 0 and world == 1 and not captured and STREAMS is None:
        # the same loop fed from the host (train.py:123), outside the driver's clock: 100 steps, [2,B] int64 from pinned memory per step
        # (after the last timed step's sizes were taken, above: the loop re-uses the step buffers)
        timer.enabled = False
        keep_last = last
        pinned = [e.cpu().pin_memory() for e in edges[:min(len(edges), 16)]]
        host_fed_rate = host_fed_loop(lambda e, s: hot_path_step(sp, csr, e, M, k, seed=s, rng=rng, slot=s & 1), finish_step, pinned, B, dev,
                                      max(extra_regions[1], 100))
        last = keep_last
    # for the record, outside the clock: the same steps with a packed CSR SpG built for every batch (rank 0, 1 GPU)
    csr_ms = None
    if csr_variant and rank == 0 and world == 1 and last is not None and last[1].strided:
        keep_last, keep_strided, STRIDED = last, STRIDED, False
        run_steps(range(3))                                   # allocator steady state for this variant
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(range(W, W + K))
        torch.cuda.synchronize()
        csr_ms = (time.perf_counter() - t1) / max(K, 1) * 1e3
        last, STRIDED = keep_last, keep_strided
    two_streams = None
    if rank == 0 and world == 1 and STREAMS is None and last is not None and two_stream_extra:
        keep_last = last
        timer.enabled = False
        STREAMS = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for st_ in STREAMS:
            st_.wait_stream(torch.cuda.current_stream(dev))
        run_steps(range(4))
        torch.cuda.synchronize()
        K2 = max(K, 100)
        t1 = time.perf_counter()
        run_steps(range(W, W + K2))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        two_streams = {"pairs_per_s": B * K2 / dt, "ms_per_step": dt / K2 * 1e3, "steps": K2}
        STREAMS, last = None, keep_last
    # ... and, outside the clock as well: the same steps with every DISTINCT endpoint of a batch sampled once (Philox keys a walk by
    # its root's id, so (xz, indptr) are the same -- tested); what a serving loop may do, not what the timed region does
    dedup_loop = None
    if rank == 0 and world == 1 and last is not None and (two_stream_extra if dedup_extra is None else dedup_extra) and BUFFERED and not DEDUP and rng == "philox":
        keep_last, DEDUP = last, True
        timer.enabled = False
        try:
            run_steps(range(4))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(range(W, W + K))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            nd = getattr(last[1], "n_distinct", None)
            dedup_loop = {"pairs_per_s": B * K / dt, "ms_per_step": dt / max(K, 1) * 1e3, "steps": K,
                          "distinct_roots_last_step": nd, "endpoints_last_step": 2 * B}
        except ValueError as ex:
            dedup_loop = {"skipped": str(ex)}
        DEDUP, last = False, keep_last
    sampler_mod.KERNEL_TIMER = None
    if rank != 0:
        return None

    edge, sets, xz = last
    walk_ms, launches = timer.mean_ms("walk_sets")
    join_ms, _ = timer.mean_ms("sjoin_fill")
    abytes = last_abytes
    achieved = abytes / (walk_ms * 1e-3) / 1e9 if walk_ms else None
    fused_rows = sets.data is not None or sets.strided
    finished = timer.mean_ms("spg_build")[0] is not None and sets.strided    # rows came from the general kernel + finish_rows
    # HBM-side traffic / missed lines of the walk kernel: PMC passes (tools/pmc_traffic.py) of exactly these kernels
    # (matched by a hash of the kernel sources) and this configuration -- or null
    traffic = lines = join_traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath)).get(f"{name}:{B}:{M}:{k}:{'spg' if fused_rows else 'sets'}:{rng}", {})
        if tj.get("kernel_source_sha") == kernel_source_sha():
            traffic, lines = tj.get("walk_sets_hbm_bytes_per_launch"), tj.get("walk_sets_l2_miss_lines_per_launch")
            join_traffic = tj.get("join_hbm_bytes_per_launch")
    roof, roof_bytes, roof_table = measure_line_roof(csr) if rank == 0 else (None, None, None)
    # SURVEY 8(d), SpJoin: per pair 16 (query ids) + 32 (4 row offsets) + 16 (two segment pointers) and per output row 8 read
    # (id + payload) + 8k written (float32 [.,2,k])
    join_abytes = (B * 64 + last_rows * (8 + 8 * k)) if last_rows is not None else None
    value = world * B * K / elapsed
    ms_per_step = elapsed / K * 1e3
    if value_is_median and len(region_values) > 1:
        value = median(region_values)
        ms_per_step = B / value * 1e3
    out = {
        "metric": "query-pairs/sec (sample+SpJoin)", "value": value, "unit": "query-pairs/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": desc, "pairs_per_step_per_gpu": B, "roots_per_step_per_gpu": 2 * B,
                   "pairs_per_step_all_gpus": world * B,
                   # one record per rank (all-gathered): the first multi-GPU run must show which devices really took part
                   "ranks_seen": len(rank_records), "distinct_devices": len({r["device"] for r in rank_records}),
                   "per_rank_ms_min": min(r["elapsed_ms"] for r in rank_records) / max(K, 1),
                   "per_rank_ms_max": max(r["elapsed_ms"] for r in rank_records) / max(K, 1),
                   "per_rank_walk_kernel_ms_min": min((r["walk_kernel_ms"] or 0.0) for r in rank_records),
                   "per_rank_walk_kernel_ms_max": max((r["walk_kernel_ms"] or 0.0) for r in rank_records),
                   "dist_backend": (os.environ.get("SUBGACC_DIST_BACKEND", "nccl") if dist is not None else "none"),
                   "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if dist is not None else None,
                   "rank_records": rank_records,
                   "num_walks": M, "num_steps_cli": k, "rng": rng, "parallelism": f"query-shard x{world}",
                   "set_members_last_step": last_members, "distinct_lp_rows_last_step": last_distinct,
                   "xz_rows_last_step": last_rows, "graph_nnz": csr.nnz,
                   # the on-demand step's sampler stages (key rows: the walk kernel alone -- nothing is numbered or packed);
                   # SURVEY 8(d)'s S (walk -> register -> number -> packed SpG resident) is offline_S_roots_per_s below
                   "S_on_demand_rows_roots_per_s": 2 * B / (1e-3 * sum(v for v in (timer.mean_ms(n_)[0] for n_ in
                                                   ("walk_sets", "register_rows", "compact_sets", "uniq_rows", "spg_build")) if v)),
                   "J_pairs_per_s": (B / (1e-3 * join_ms)) if join_ms else None,
                   "fused_spg_rows": fused_rows, "spg_layout": "strided rows joined in place (no CSR copy of the batch)" if sets.strided else "csr",
                   "packed_csr_ms_per_step": csr_ms,
                   "two_stream_loop": two_streams, "dedup_roots_loop": dedup_loop,
                   "device_allocs_in_timed_region": allocs_timed,
                   "timed_loop": ("one HIP graph per step (CapturedStep x3, read-back two steps behind)" if captured else
                                  "eager launches (six per step), read-back one step behind"),
                   # the main region followed by the extra regions (outside the driver's clock): pairs/s each
                   "region_pairs_per_s": [round(v) for v in region_values], "extra_region_steps": extra_regions[1],
                   "pairs_per_s_min": min(region_values), "pairs_per_s_median": median(region_values),
                   "pairs_per_s_max": max(region_values),
                   "host_step_ms_min_median_max": [round(1e3 * v, 3) for v in
                                                   (min(host_steps), sorted(host_steps)[len(host_steps) // 2], max(host_steps))]
                   if host_steps else None,
                   "stage_ms": {n_: timer.mean_ms(n_)[0] for n_ in
                                ("walk_sets", "register_rows", "compact_sets", "uniq_rows", "spg_build", "sjoin_fill")}},
        "roofline": {"bound": "hbm", "kernel": sampler_mod.walk_kernel_name(csr, M, k - 1, fused_rows and not finished),
                     "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                     "kernel_ms": walk_ms, "launches_timed": launches, "algorithmic_bytes_per_launch": abytes,
                     # the step's second kernel (sjoin_keypair_kernel), same contract: algorithmic bytes / its own HIP-event time
                     "join_kernel_ms": join_ms, "join_algorithmic_bytes_per_launch": join_abytes,
                     "join_achieved": (join_abytes / (join_ms * 1e-3) / 1e9) if (join_ms and join_abytes) else None,
                     "join_frac": (join_abytes / (join_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (join_ms and join_abytes) else None,
                     "join_traffic": join_traffic,
                     # the step as a whole at the memory side of L2 (walk + join, PMC): what the two kernels queue on together --
                     # plain streams reach 5.1-5.6 TB/s on this pool's boxes (profiles/r17_stream_probe.csv), the guide's float4 copy 6.29 TB/s
                     "step_traffic": (traffic + join_traffic) if (traffic and join_traffic) else None,
                     "step_traffic_TBps": ((traffic + join_traffic) / (ms_per_step * 1e-3) / 1e12) if (traffic and join_traffic) else None,
                     # the roof this kernel actually sits under: random 128-byte lines per second
                     "random_line_roof": {"lines_per_s": roof, "line_bytes": LINE_BYTES, "table_bytes": roof_bytes,
                                          "source": (f"tools/line_roof_lib.hip in this run: independent random 4-byte reads over the {roof_table} "
                                                     f"({roof_bytes >> 20} MiB), 2048 x 256 lanes, best of 3 (study: profiles/r02_line_probe_pmc.csv)")
                                          if roof else "tools/build/libsubgacc_probe.so not built",
                                          "l2_miss_lines_per_launch": lines,
                                          "achieved_lines_per_s": (lines / (walk_ms * 1e-3)) if (lines and walk_ms) else None,
                                          "frac": (lines / (walk_ms * 1e-3) / roof) if (lines and walk_ms and roof) else None}},
    }
    if small_batches:
        out["config"]["batch_size_and_hip_graph"] = bench_studies().batch_size_and_graph(sp, csr, M, k, rng, K)
    if two_stream_extra and two_streams:
        out["config"]["two_stream_pairs_per_s"] = two_streams["pairs_per_s"]
    if host_fed_rate:
        out["config"]["host_fed_pairs_per_s"] = host_fed_rate
        out["config"]["host_fed_over_resident"] = host_fed_rate / median(region_values[1:] or region_values)
    if dedup_loop and "pairs_per_s" in dedup_loop:
        out["config"]["dedup_roots_pairs_per_s"] = dedup_loop["pairs_per_s"]
    if with_cpu_baseline and not name.startswith("twitter"):   # 12 GB CSR: no host copy
        try:
            out["cpu_baseline"] = cpu_baseline(csr, edges[W], M, k, **cpu_kw)
        except Exception as ex:  # the baseline is a report, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "query-pairs/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {type(ex).__name__}: {ex}"}
    if ref_flow and not offline:     # SURVEY 8(d)'s S / J / Q on the reference's own flow: part of the default run since round 6 (~1.5 s)
        try:
            flow = reference_flow(sp, csr, M, k, B)[0]
            out["config"]["offline_flow"] = flow
        except Exception as ex:     # an extra must never cost the headline line
            out["config"]["offline_flow"] = {"failed": f"{type(ex).__name__}: {ex}"}
    if offline:      # the reference's own flow + the rest of the surface, outside the clock (rank 0, 1 GPU, the headline workload)
        try:
            torch.cuda.empty_cache()
            flow, (z, table) = bench_studies().offline_flow(sp, csr, M, k, B, K, out.get("cpu_baseline"))
            out["config"]["offline_flow"] = flow
            try:
                out["config"]["hgather"] = bench_studies().bench_hgather(sp, z, table, k, max(K, 5))
            except Exception as ex:
                out["config"]["hgather"] = {"failed": f"{type(ex).__name__}: {ex}"}
            try:
                out["config"]["mean_stage"] = bench_studies().bench_mean_stage(sp, sampler_mod, z, table, csr, k, B, 10)
            except Exception as ex:
                out["config"]["mean_stage"] = {"failed": f"{type(ex).__name__}: {ex}"}
            del z, table
        except Exception as ex:     # an extra must never cost the headline line
            out["config"]["offline_flow"] = {"failed": f"{type(ex).__name__}: {ex}"}
    return out


def reference_flow(sp, csr, M, k, B, reps=3):
    """SURVEY 8(d)'s S / J / Q on the reference's OWN flow, outside the driver's clock (rank 0, 1 GPU, the headline graph):
    main.py:172-178 samples every node once -- subg_matrix over all N: walk -> register -> number -> packed SpG resident in HBM
    with its enc table (S, roots/s) --, train.py:120-127 then joins every batch from that resident store (J, pairs/s: gather(edge, z,
    encode=Z_SF), B pairs per call; J_keyed: the store re-keyed once, SpG.keyed).  Q_formula = 1 / (2/S + 1/J), the survey's
    on-demand figure made of the two; Q_amortised = 1e8 pairs / (N/S + 1e8/J), the reference-style number.
    -> (dict, z, sets, enc, table, zk)"""
    from surel_plus_amd.graphs import query_pairs
    from surel_plus_amd.spg import sample_spg
    dev, N = csr.device, csr.num_nodes
    idx = torch.arange(N, dtype=torch.int32, device=dev)
    z = sets = None
    times = []
    for _ in range(reps):       # the previous store goes back to the allocator first: steady state, no fresh GB-sized hipMalloc
        del z, sets
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        z, sets = sample_spg(csr, idx, num_walks=M, num_steps=k - 1, seed=111413, rng="philox", fused=True)
        enc = sets.enc_int16()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t_off = min(times[1:])
    table = sets.feature_table()
    edges = [query_pairs(csr, B, seed=9000 + s_, device=dev) for s_ in range(7)]
    cap = 2 * B * z.max_len * 2 * k
    buf = _XZ_BUF.get((dev, cap, 0))
    if buf is None:
        buf = torch.empty(cap, dtype=torch.float32, device=dev)

    def join_rate(store, tab):
        for e in edges[:2]:
            sp.gather(e, store, dev, ptr=True, encode=tab, out=buf, lazy=True)
        rs = []
        for _ in range(3):          # three loops of 50 joins (>= 20 ms each); the median is reported
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for j in range(50):
                sp.gather(edges[2 + j % 5], store, dev, ptr=True, encode=tab, out=buf, lazy=True)
            torch.cuda.synchronize()
            rs.append(50 * B / (time.perf_counter() - t1))
        return median(rs)
    J = join_rate(z, table)
    encz = torch.cat([torch.zeros((1, enc.shape[1]), dtype=enc.dtype, device=dev), enc])
    zk = None
    for _ in range(2):          # (steady-state allocator again: the second re-keying is the one that counts)
        del zk
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        zk = z.keyed(encz, M)
        torch.cuda.synchronize()
        t_key = time.perf_counter() - t2
    JK = join_rate(zk, zk.slot_table())
    # ... and the keyed store laid out for serving (SpG.aligned(): rows on whole lines at a fixed pitch, no row pointers; round 6)
    JKA = aligned_ms = aligned_bytes = None
    try:
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        za = zk.aligned()
        torch.cuda.synchronize()
        aligned_ms, aligned_bytes = (time.perf_counter() - t3) * 1e3, za.nbytes
        JKA = join_rate(za, za.slot_table())
        del za
    except Exception as ex:      # (an extra: never a reason to lose S / J / Q)
        aligned_ms = f"{type(ex).__name__}: {ex}"
    S = N / t_off
    flow = {"all_N_sample_to_resident_spg_ms": t_off * 1e3, "S_roots_per_s": S, "set_members": z.nnz,
            "distinct_lp_rows": int(enc.shape[0]), "J_pairs_per_s_table_store": J, "J_pairs_per_s_keyed_store": JK,
            "rekey_once_ms": t_key * 1e3, "pairs_per_batch": B,
            "J_pairs_per_s_keyed_aligned_store": JKA, "aligned_once_ms": aligned_ms, "aligned_store_bytes": aligned_bytes,
            "packed_store_bytes": (N + 1) * 8 + z.nnz * 8,
            "Q_formula_pairs_per_s": 1.0 / (2.0 / S + 1.0 / J), "Q_formula_keyed_pairs_per_s": 1.0 / (2.0 / S + 1.0 / JK),
            "Q_amortised_at_1e8_pairs_table": 1e8 / (t_off + 1e8 / J), "Q_amortised_at_1e8_pairs_keyed": 1e8 / (t_off + t_key + 1e8 / JK),
            "reference": "main.py:172-178 (subg_matrix over all N once) + train.py:120-127 (one join per batch from the resident store)"}
    return flow, z, sets, enc, table, zk


def host_fed_loop(run_step, finish, host_edges, B, dev, steps):
    """The timed loop's shape with the [2,B] query batch coming from the HOST every step, as the reference feeds edges[:, perm]
    (train.py:123): the batches lie in pinned host memory; the upload of step s+1 is queued on a copy stream while step s's kernels
    run (two device slots in turn: the copy into a slot waits for the step that last read it), the step waits for its own upload only.
    -> pairs/s (outside the driver's clock; SURVEY 7: "nothing on the steady-state path touches the host except the [2,B] query upload")"""
    cur = torch.cuda.current_stream(dev)
    copy_stream = torch.cuda.Stream(device=dev)
    slots = [torch.empty_like(host_edges[0], device=dev) for _ in range(2)]
    uploaded = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    for ev in consumed:
        ev.record(cur)

    def upload(s):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[s & 1])
            slots[s & 1].copy_(host_edges[s % len(host_edges)], non_blocking=True)
            uploaded[s & 1].record(copy_stream)

    def loop(n):
        pending = None
        upload(0)
        for s in range(n):
            if s + 1 < n:
                upload(s + 1)
            cur.wait_event(uploaded[s & 1])
            queued = run_step(slots[s & 1], s)
            consumed[s & 1].record(cur)
            if pending is not None:
                finish(*pending)
            pending = queued
        finish(*pending)
    loop(6)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t1)


def bench_studies():
    """the side studies of --full (B = 1,024 loops, the reference's offline flow, hgather, the mean stage, walk_sampler) live in
    tools/bench_studies.py: minutes of work whose numbers go to bench_detail.json only -- not part of the default run"""
    if __name__ == "__main__":
        sys.modules.setdefault("bench", sys.modules["__main__"])       # (the studies `import bench`: this module, not a second copy)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_studies as mod
    return mod


def device_identity(dev):
    """what tells two ranks' GPUs apart: PCI bus id / uuid where torch exposes them"""
    p = torch.cuda.get_device_properties(dev)
    parts = [str(getattr(p, a)) for a in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id") if hasattr(p, a)]
    return f"{p.name}|{'|'.join(parts)}|idx{torch.cuda.current_device()}"


def gather_rank_records(dist, world, rank, dev, elapsed_local, kernel_ms, backend):
    """every rank's (rank, device identity, own elapsed, own walk-kernel ms) on rank 0 -- so that the first 8-GPU run is a record,
    not a debugging session: ranks_seen, distinct devices, per-rank spread"""
    rec = {"rank": rank, "device": device_identity(dev), "elapsed_ms": elapsed_local * 1e3, "walk_kernel_ms": kernel_ms,
           "host": os.uname().nodename, "pid": os.getpid()}
    if dist is None:
        return [rec]
    recs = [None] * world
    dist.all_gather_object(recs, rec)
    return recs


def require_distinct_devices(rank_records, world, backend):
    """The first multi-GPU run must not be able to lie: under RCCL ("nccl": one rank per GPU is the whole point) every rank has to
    report a device of its own.  Ranks that share one -- a launcher that did not set LOCAL_RANK, HIP_VISIBLE_DEVICES pinning every rank
    to device 0 -- would still print a line, with a `value` N times too good to be true for N GPUs' worth of hardware: every rank
    (they all hold the all-gathered records) ends with exit code 3 instead.  The gloo rehearsal on ONE GPU (SUBGACC_DIST_BACKEND=gloo
    SUBGACC_SHARE_GPU=1) is the stated exception."""
    if world <= 1 or backend != "nccl":
        return
    devs = [r["device"] for r in rank_records]
    if len(rank_records) != world or len(set(devs)) != world:
        dup = sorted({d for d in devs if devs.count(d) > 1})
        print(f"bench.py: {world} ranks over nccl but {len(set(devs))} distinct device(s) in {len(rank_records)} rank record(s); shared: "
              f"{dup[:2]} -- one rank per GPU is required (LOCAL_RANK / visible devices?)", file=sys.stderr, flush=True)
        sys.exit(3)


def flatten(out):
    """The driver keeps the SCALAR keys of `config` (names cut at 40 characters) and drops everything nested: every number
    DESIGN.md quotes is therefore promoted to a scalar key here; the nested blocks stay for whoever reads the full line."""
    c = out["config"]

    def put(key, val):
        assert len(key) <= 40, key
        if val is not None:
            c[key] = val
    # the headline: the driver's region is `value`; the three 100-step regions behind it as min / median / max
    rv = c.get("region_pairs_per_s") or []
    if len(rv) > 1:
        put("headline_median_of_3_x100", median(rv[1:]))
        put("headline_min_of_3_x100", min(rv[1:]))
        put("headline_max_of_3_x100", max(rv[1:]))
    for name, short in (("cit2 (rng=rand_r: the reference's own stream, bit-exact mode)", "rand_r"), ("cit2m4", "cit2m4"), ("collab", "collab"),
                        ("ppa", "ppa"), ("twitter", "twitter"), ("cit2ppr", "cit2ppr"), ("cit2ppr_packed", "pprpk"), ("cit2loc", "cit2loc")):
        o = (c.get("other_workloads") or {}).get(name) or {}
        put(f"{short}_pairs_per_s", o.get("value"))             # the MEDIAN of three regions of >= 100 steps
        put(f"{short}_pairs_per_s_min", (o.get("config") or {}).get("pairs_per_s_min"))
        put(f"{short}_pairs_per_s_max", (o.get("config") or {}).get("pairs_per_s_max"))
        put(f"{short}_ms_per_step", o.get("ms_per_step"))
        if "failed" in o:
            put(f"{short}_failed", str(o["failed"])[:200])
        r = o.get("roofline") or {}
        put(f"{short}_join_frac", r.get("join_frac"))
        put(f"{short}_frac", r.get("frac"))
        put(f"{short}_kernel_ms", r.get("kernel_ms"))
        put(f"{short}_line_roof_frac", (r.get("random_line_roof") or {}).get("frac"))
        put(f"{short}_traffic_bytes", r.get("traffic"))
        put(f"{short}_join_ms", ((o.get("config") or {}).get("stage_ms") or {}).get("sjoin_fill"))
        put(f"{short}_dedup_pairs_per_s", ((o.get("config") or {}).get("dedup_roots_loop") or {}).get("pairs_per_s"))
        put(f"{short}_join_call_ms", (o.get("config") or {}).get("join_call_ms"))
        put(f"{short}_join_call_ms_steady", (o.get("config") or {}).get("join_call_ms_steady"))
        put(f"{short}_frac_whole_join_call", (o.get("config") or {}).get("frac_of_hbm_peak_whole_join_call"))
        put(f"{short}_frac_x10_back_to_back", r.get("frac_x10_back_to_back"))
        put(f"{short}_frac_single_launch", r.get("frac_single_launch"))
        if "cpu_baseline" in o:
            put(f"{short}_cpu_pairs_per_s", o["cpu_baseline"].get("value"))
            put(f"{short}_cpu_cores", o["cpu_baseline"].get("cores"))
            put(f"{short}_cpu_t1_pairs_per_s", o["cpu_baseline"].get("t1_pairs_per_s"))
            put(f"{short}_cpu_t16_pairs_per_s", o["cpu_baseline"].get("t16_pairs_per_s"))
        ld = o.get("literal_dropin") or {}
        put(f"{short}_literal_dropin_roots_per_s", ld.get("drop_in_roots_per_s"))
        put(f"{short}_dropin_shim_roots_per_s", ld.get("shim_roots_per_s"))
        put(f"{short}_dropin_shim_ms", None if ld.get("shim_seconds") is None else 1e3 * ld["shim_seconds"])
        put(f"{short}_dropin_caller_ms", None if ld.get("caller_seconds") is None else 1e3 * ld["caller_seconds"])
        put(f"{short}_literal_ref_roots_per_s", ld.get("reference_roots_per_s"))
    ws = (c.get("other_workloads") or {}).get("walk_sampler (collab)") or {}
    put("walk_sampler_collab_roots_per_s", ws.get("value"))
    put("walk_sampler_collab_frac", (ws.get("roofline") or {}).get("frac"))
    hg = c.get("hgather") or {}
    put("hgather_b2048_triplets_per_s", hg.get("value"))
    put("hgather_b2048_ms_per_step", hg.get("ms_per_step"))
    put("hgather_b2048_frac", (hg.get("roofline") or {}).get("frac"))
    put("hgather_b2048_one_call_triplets_per_s", hg.get("one_call_triplets_per_s") if not isinstance(hg.get("one_call_triplets_per_s"), str) else None)
    put("hgather_b2048_pool4_triplets_per_s", hg.get("pool4_triplets_per_s"))
    if hg.get("pool4_triplets_per_s") and (hg.get("roofline") or {}).get("algorithmic_bytes_per_launch"):
        # ... as a fraction of the HBM peak: the algorithmic bytes of one batch x batches per second (four lanes in flight)
        put("hgather_b2048_pool4_frac", hg["roofline"]["algorithmic_bytes_per_launch"] * hg["pool4_triplets_per_s"] / hg["triplets_per_step"] / 1e9 / HBM_PEAK_GBS)
    put("hgather_b2048_many32_triplets_per_s", hg.get("many32_triplets_per_s"))
    cb = out.get("cpu_baseline") or {}
    put("cpu_model", cb.get("cpu_model"))
    put("cpu_t1_pairs_per_s", cb.get("t1_pairs_per_s"))
    put("cpu_t16_pairs_per_s", cb.get("t16_pairs_per_s"))
    put("cpu_best_nthread", cb.get("best_nthread"))
    put("cpu_probe_best_nthread", cb.get("probe_best_nthread"))
    ms_ = c.get("mean_stage") or {}
    for H in (96, 256):
        put(f"mean_stage_H{H}_pairs_per_s", ms_.get(f"H{H}_fused_pairs_per_s"))
        put(f"mean_stage_H{H}_ref_style_pairs_per_s", ms_.get(f"H{H}_ref_style_pairs_per_s"))
        put(f"mean_stage_H{H}_sparse_pairs_per_s", ms_.get(f"H{H}_sparse_pairs_per_s"))
        put(f"mean_stage_H{H}_gemm_ms", ms_.get(f"H{H}_gemm_ms"))
    put("mean_stage_counts_kernel_ms", ms_.get("counts_kernel_ms"))
    put("mean_stage_counts_frac", ms_.get("counts_frac_of_hbm_peak"))
    put("mean_stage_counts_alg_bytes", ms_.get("counts_algorithmic_bytes_per_launch"))
    if "failed" in ms_:
        put("mean_stage_failed", str(ms_["failed"])[:200])
    f = c.get("offline_flow") or {}
    jb = f.get("J_b1024_keyed") or {}
    put("offline_J_b1024_eager_pairs_per_s", (jb.get("eager") or {}).get("pairs_per_s"))
    put("offline_J_b1024_graph_pairs_per_s", (jb.get("graph") or {}).get("pairs_per_s"))
    put("offline_J_b1024_pool4_pairs_per_s", (jb.get("pool_4lanes") or {}).get("pairs_per_s"))
    put("offline_J_b1024_many64_pairs_per_s", (jb.get("many_64") or {}).get("pairs_per_s"))
    if "failed" in jb:
        put("offline_J_b1024_failed", str(jb["failed"])[:200])
    put("S_offline_roots_per_s", f.get("S_roots_per_s"))              # SURVEY 8(d)'s triple under the names the round-5 review asked for
    put("J_resident_pairs_per_s", f.get("J_pairs_per_s_table_store"))
    put("Q_formula_pairs_per_s", f.get("Q_formula_pairs_per_s"))
    put("Q_amortised_1e8_pairs_per_s", f.get("Q_amortised_at_1e8_pairs_table"))
    put("offline_all_N_ms", f.get("all_N_sample_to_resident_spg_ms"))
    put("offline_all_N_4hop_ms", f.get("all_N_4hop_sample_to_resident_spg_ms"))
    put("offline_4hop_roots_per_s", f.get("all_N_4hop_roots_per_s"))
    put("offline_S_roots_per_s", f.get("S_roots_per_s"))
    put("offline_J_table_pairs_per_s", f.get("J_pairs_per_s_table_store"))
    put("offline_J_keyed_pairs_per_s", f.get("J_pairs_per_s_keyed_store"))
    put("offline_J_keyed_aligned_pairs_per_s", f.get("J_pairs_per_s_keyed_aligned_store"))
    put("offline_Q_1e8_table_pairs_per_s", f.get("Q_amortised_at_1e8_pairs_table"))
    put("offline_Q_1e8_keyed_pairs_per_s", f.get("Q_amortised_at_1e8_pairs_keyed"))
    fc = f.get("cpu_baseline") or {}
    put("offline_cpu_S_roots_per_s", fc.get("S_roots_per_s"))
    put("offline_cpu_J_pairs_per_s", fc.get("J_pairs_per_s"))
    put("offline_cpu_Q_1e8_pairs_per_s", fc.get("Q_amortised_at_1e8_pairs"))
    if "failed" in f:
        put("offline_flow_failed", f["failed"])
    bg = c.get("batch_size_and_hip_graph") or {}
    for key, short in (("B=1024 eager", "b1024_pairs_per_s_eager"), ("B=1024 graph", "b1024_pairs_per_s_graph"),
                       ("B=1024 graph, 4 lanes", "b1024_pairs_per_s_graph_4lanes"), ("B=1024 graph, 8 lanes", "b1024_pairs_per_s_graph_8lanes"),
                       ("B=1024 graph, 8 lanes, inputs ready", "b1024_pairs_per_s_8lanes_ready"),
                       ("B=1024 many: 64 batches per launch sequence", "b1024_pairs_per_s_many64"),
                       ("B=65536 graph, 8 lanes", "b65536_pairs_per_s_graph_8lanes")):
        put(short, (bg.get(key) or {}).get("pairs_per_s"))
    rl = out.get("roofline") or {}
    lr = rl.get("random_line_roof") or {}
    rl["line_roof_lines_per_s"] = lr.get("lines_per_s")         # (the driver keeps the roofline block's scalars too)
    rl["line_roof_frac"] = lr.get("frac")
    rl["l2_miss_lines_per_launch"] = lr.get("l2_miss_lines_per_launch")
    put("join_frac", rl.get("join_frac"))
    put("join_traffic_bytes", rl.get("join_traffic"))
    put("step_traffic_TBps", rl.get("step_traffic_TBps"))
    sm = c.get("stage_ms") or {}
    put("walk_kernel_ms", sm.get("walk_sets"))
    put("join_kernel_ms", sm.get("sjoin_fill"))


# ---- the driver's line ------------------------------------------------------------------------------------------------------
# The driver parses the LAST stdout line; round 4's 31 KB line (every block nested + ~170 promoted scalars) did not make it into
# BENCH_r04.json.  So: ONE compact line (< 6 KB, guarded by tests/test_bench_line_cpu.py) with the contract's keys, <= 40 scalars
# of `config`, the `roofline` and `cpu_baseline` blocks without prose -- and the full record (every nested block and every
# promoted scalar, as before) in bench_detail.json next to this script (and under gpurun_out/ when that directory exists).
LINE_LIMIT = 6000
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
CONFIG_KEYS = ("workload", "pairs_per_step_per_gpu", "ranks_seen", "distinct_devices", "rng",
               # the driver keeps the FIRST 24 keys of `config` (BENCH_r05: three BASELINE configurations fell off the end).  So, in this order:
               # BASELINE.json's configs[0..4] -- collab on the CPU (the compiled reference), collab / ppa / cit2-PPR / twitter on the GPU, each
               # with its walk and join fractions of the HBM peak (medians of 3 x 100 steps) --, then SURVEY 8(d)'s S / J / Q on the
               # reference's own flow, the host-fed loop, the graph with id locality.  tests/test_bench_line_cpu.py asserts the POSITIONS.
               "collab_cpu_pairs_per_s", "collab_pairs_per_s", "collab_frac", "collab_join_frac",
               "ppa_pairs_per_s", "ppa_frac", "ppa_join_frac",
               "cit2ppr_pairs_per_s", "cit2ppr_frac", "cit2ppr_frac_whole_join_call",
               "twitter_pairs_per_s", "twitter_frac", "twitter_join_frac",
               "S_offline_roots_per_s", "J_resident_pairs_per_s", "Q_formula_pairs_per_s", "Q_amortised_1e8_pairs_per_s",
               "host_fed_pairs_per_s", "cit2loc_pairs_per_s",
               # ---- beyond the driver's 24: the line still carries them (and bench_detail.json everything)
               "cit2loc_frac", "pprpk_pairs_per_s", "pprpk_frac", "offline_J_keyed_aligned_pairs_per_s", "cit2m4_pairs_per_s", "cit2m4_frac", "cit2m4_join_frac", "headline_median_of_3_x100",
               "rand_r_pairs_per_s", "rand_r_frac", "two_stream_pairs_per_s", "rccl_version",
               # seam A alone: random_walks.py:77-81 verbatim over the drop-in module, all N collab roots (shim = inside gset_sampler)
               "collab_literal_dropin_roots_per_s", "collab_dropin_shim_roots_per_s", "collab_literal_ref_roots_per_s",
               "detail")
DRIVER_KEEPS = 24       # config keys the driver's record keeps
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "launches_timed",
                 "algorithmic_bytes_per_launch", "join_kernel_ms", "join_algorithmic_bytes_per_launch", "join_frac", "join_traffic",
                 "line_roof_lines_per_s", "line_roof_frac", "l2_miss_lines_per_launch")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "cpu_model", "host_threads", "t1_pairs_per_s", "t16_pairs_per_s",
            "best_nthread", "sampler_roots_per_s", "join_pairs_per_s", "cpu_seconds")


def _short(v, digits=6):
    """floats to 6 significant digits (the line is a record, not a checkpoint), long strings cut"""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    if isinstance(v, str) and len(v) > 200:
        return v[:197] + "..."
    return v


def compact_line(out):
    """the one line the driver parses: contract keys + a bounded pick of `config` scalars + roofline + cpu_baseline"""
    line = {k_: _short(out.get(k_)) for k_ in TOP_KEYS}
    c = out.get("config") or {}
    line["config"] = {k_: _short(c[k_]) for k_ in CONFIG_KEYS if c.get(k_) is not None and not isinstance(c[k_], (dict, list))}
    assert len(line["config"]) <= 40, len(line["config"])
    r = out.get("roofline") or {}
    line["roofline"] = {k_: _short(r.get(k_)) for k_ in ROOFLINE_KEYS if k_ in r}
    if out.get("cpu_baseline") is not None:
        line["cpu_baseline"] = {k_: _short(out["cpu_baseline"].get(k_)) for k_ in CPU_KEYS if k_ in out["cpu_baseline"]}
    text = json.dumps(line)
    if len(text) > LINE_LIMIT:         # cannot happen with the key lists above; if it ever does, keep the contract and say so
        line["config"] = {k_: line["config"][k_] for k_ in CONFIG_KEYS[:DRIVER_KEEPS] if k_ in line["config"]}
        line["config"]["truncated"] = True
        text = json.dumps(line)
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def emit(out):
    """full record -> bench_detail.json (+ gpurun_out/), compact record -> the last line of stdout"""
    flatten(out)
    paths = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    written = None
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f)
            written = written or os.path.relpath(p, ROOT)
        except OSError:
            pass
    out["config"]["detail"] = written
    print(compact_line(out), flush=True)


def note(msg):
    """progress to stderr: short, so that the driver's tail still ends in the JSON line"""
    if os.environ.get("SUBGACC_BENCH_QUIET", "0") != "1":
        print(f"[bench {time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def summary(o):
    """what an `other_workloads` entry keeps of a full line"""
    keep = {k_: o[k_] for k_ in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype")}
    keep["config"] = {k_: o["config"].get(k_) for k_ in ("workload", "pairs_per_step_per_gpu", "rng", "store_layout", "store_bytes", "store_bytes_packed", "set_members_last_step", "device_allocs_in_timed_region", "host_step_ms_min_median_max",
                                                         "distinct_lp_rows_last_step", "xz_rows_last_step", "graph_nnz", "spg_layout", "stage_ms", "two_stream_loop", "dedup_roots_loop", "spg_members",
                                                         "timed_loop", "region_pairs_per_s", "pairs_per_s_min", "pairs_per_s_median", "pairs_per_s_max",
                                                         "offline_ppr_stage_s", "join_call_ms", "join_call_ms_steady",
                                                         "frac_of_hbm_peak_whole_join_call") if k_ in o["config"]}
    keep["roofline"] = o["roofline"]
    if "cpu_baseline" in o:
        keep["cpu_baseline"] = o["cpu_baseline"]
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cit2", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=65536, help="query pairs per step per GPU")
    ap.add_argument("--rng", default="philox", choices=["philox", "rand_r"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the short passes over BASELINE.json's other configs")
    ap.add_argument("--full", action="store_true",
                    help="also run the side studies (two-stream and root-dedup loops, packed-CSR variant, B=1,024 sweeps, the reference's "
                         "offline flow, hgather, mean stage, walk_sampler): minutes, all of it into bench_detail.json")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; invalidates the number)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --pairs per step PER GPU (the driver's scaling runs); strong: --pairs per step in all, split over "
                         "the ranks as shard.shard_pairs splits a batch (contiguous shares)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # `python bench.py --gpus N` without a launcher: this process has not touched the GPU yet, so it may start the
        # N ranks itself (one process per GPU through torch.distributed.run, as the driver does) and hand their exit
        # code on.  A launcher that set WORLD_SIZE to something else than --gpus is a usage error.
        if "WORLD_SIZE" in os.environ:
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; launch with "
                     f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
                     f"--master-port 29511 bench.py --gpus {args.gpus} ...`")
        import subprocess
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 400))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") is the backend; SUBGACC_DIST_BACKEND=gloo + SUBGACC_SHARE_GPU=1 lets a 1-GPU box exercise the
        # multi-rank control flow (barriers, max-over-ranks) with every rank on cuda:0
        backend = os.environ.get("SUBGACC_DIST_BACKEND", "nccl")
        if os.environ.get("SUBGACC_SHARE_GPU", "0") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import surel_plus_amd as sp
    from surel_plus_amd import sampler as sampler_mod

    B, K, W = args.pairs, args.steps, args.warmup
    if args.scaling == "strong":      # one fixed batch per step, every rank its contiguous share (shard.shard_range)
        from surel_plus_amd.shard import shard_range
        lo, hi = shard_range(args.pairs, rank, world)
        B = hi - lo
        if args.pairs % world:
            sys.exit(f"bench.py --scaling strong: --pairs {args.pairs} must be a multiple of the {world} ranks (equal shares keep "
                     f"value = pairs / max-over-ranks time honest)")
    t_start = time.perf_counter()
    if WORKLOADS[args.workload][0] is None:
        out = bench_ppr(args, sp, sampler_mod, dev, rank, world, dist, WORKLOADS[args.workload][3], B, K, W,
                        layout=os.environ.get("SUBGACC_PPR_LAYOUT", "aligned"))
    else:
        out = bench_lp(args, args.workload, args.rng, B, K, W, sp, sampler_mod, dev, rank, world, dist,
                       with_cpu_baseline=(world == 1 and not args.no_cpu_baseline),
                       small_batches=(world == 1 and args.full),
                       # outside the clock, K more steps with two step buffers in turn on two streams (what a serving loop gains by it);
                       # the pass with every distinct endpoint sampled once: --full only
                       # (not with --no-others: that is the command the profiles are taken with, and kernels that overlap on two
                       #  streams would spoil its per-kernel averages)
                       two_stream_extra=not args.no_others, dedup_extra=args.full,
                       csr_variant=args.full,                    # ... and no pass with the packed-CSR (table rows) variant
                       offline=(rank == 0 and world == 1 and args.full and args.scale == 1.0),
                       ref_flow=(rank == 0 and world == 1 and not args.no_others and args.scale == 1.0 and not args.workload.startswith("twitter")),
                       host_fed=not args.no_others,
                       # the driver's K steps are the headline region; three regions of 100 steps follow, outside its clock
                       extra_regions=((3, 100) if (world == 1 and not args.no_others) else (0, 0)))
    if rank == 0 and out is not None:
        note(f"{args.workload}: {out['value'] / 1e6:.2f} M pairs/s in the driver's region of {K} steps")
    # BASELINE.json's other single-GPU configurations (and the reference-bit-exact rand_r stream on the headline one),
    # as short passes after the timed region: same code path, >= 5 timed steps each, their own roofline blocks
    # (configs[0], the reference's CPU-runnable collab case, rides on the collab entry as its cpu_baseline).
    if rank == 0 and world == 1 and not args.no_others and args.workload == "cit2" and args.rng == "philox" and args.scale == 1.0:
        others = {}
        budget_s = float(os.environ.get("SUBGACC_OTHERS_BUDGET_S", "900"))
        for key, (wl, rng_o) in {"cit2 (rng=rand_r: the reference's own stream, bit-exact mode)": ("cit2", "rand_r"),
                                 "cit2m4": ("cit2m4", "philox"),
                                 "collab": ("collab", "philox"), "ppa": ("ppa", "philox"),
                                 "cit2ppr": ("cit2ppr", "philox"), "cit2ppr_packed": ("cit2ppr", "packed"),
                                 "cit2loc": ("cit2loc", "philox"), "twitter": ("twitter", "philox")}.items():
            if time.perf_counter() - t_start > budget_s:
                others[key] = {"skipped": f"time budget of {budget_s:.0f} s for the whole run reached"}
                continue
            try:
                # Blocks of the earlier passes are handed back to the driver only where the next pass needs the room (twitter: 12 GB of
                # CSR + 47 GB of hop records).  The driver wipes freed VRAM in the background: right after torch.cuda.empty_cache()
                # released ~10 GB, the first 100-step region of whatever pass came next ran 25-30 % slow with its walk kernel at 1.5x
                # its time, whichever workload that was (r19b logs) -- so: no release where none is needed, and a pause after one
                if wl.startswith("twitter"):
                    torch.cuda.empty_cache()
                    time.sleep(1.0)
                # every pass: three regions of 100 steps (>= 40 ms each; the join-only PPR pass 300), the MEDIAN is its value
                Ko, Wo = 100, 3
                if WORKLOADS[wl][0] is None:     # the PPR store: the serving layout (rows on whole lines), then the packed CSR as rounds 1-5 joined it
                    o = bench_ppr(args, sp, sampler_mod, dev, 0, 1, None, WORKLOADS[wl][3], B, 300, Wo, extra_regions=2,
                                  layout="packed" if rng_o == "packed" else "aligned")
                else:
                    o = bench_lp(args, wl, rng_o, B, Ko, Wo, sp, sampler_mod, dev, 0, 1, None,
                                 with_cpu_baseline=(wl == "collab" and not args.no_cpu_baseline), csr_variant=False,
                                 two_stream_extra=args.full, extra_regions=(2, Ko), value_is_median=True, wake_s=0.25,
                                 # configs[0] (the reference's CPU-runnable case): 16 threads and the probe's pick only, ~1.5 s each
                                 cpu_kw=({} if args.full else {"target_s": 1.5, "teams": (-1, 16, 8), "with_t1": False}),
                                 # a 2-hop step is ~0.4 ms of kernels: replayed as one HIP graph, or the host is what gets measured
                                 captured=(WORKLOADS[wl][2] <= 3))
                others[key] = summary(o)
                note(f"{wl} ({rng_o}): {o['value'] / 1e6:.1f} M pairs/s  regions {[round(v / 1e6, 1) for v in o['config'].get('region_pairs_per_s', [])]}"
                     f"  walk {o['roofline'].get('kernel_ms') or 0:.3f} ms")
                if wl == "collab" and not args.no_cpu_baseline:
                    from surel_plus_amd.graphs import preset_graph
                    try:
                        others[key]["literal_dropin"] = literal_dropin(preset_graph("collab", device=dev), WORKLOADS[wl][1], WORKLOADS[wl][2])
                    except Exception as ex:
                        others[key]["literal_dropin"] = {"failed": f"{type(ex).__name__}: {ex}"}
            except Exception as ex:   # an extra must never cost the headline line
                others[key] = {"failed": f"{type(ex).__name__}: {ex}"}
        if args.full:
            try:
                torch.cuda.empty_cache()
                others["walk_sampler (collab)"] = bench_studies().bench_walk_sampler(sp, sampler_mod, dev, 5)
            except Exception as ex:
                others["walk_sampler (collab)"] = {"failed": f"{type(ex).__name__}: {ex}"}
        out["config"]["other_workloads"] = others
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
