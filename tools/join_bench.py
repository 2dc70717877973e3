#!/usr/bin/env python3
"""Dev-only: the join kernel of the on-demand step ALONE, on the real rows of one batch, across library builds.

    python tools/join_bench.py [--wl=collab,twitter,cit2,cit2m4,ppa] [--libs=-,tools/build/libsubgacc_x.so,...] [--n=50] [--reps=2] [--store=1]
(--store=1: the resident-store joins of the reference's flow -- table and keyed -- instead of the on-demand step's)

One process per (workload, library): builds the preset graph, runs one buffered step (so that the step buffers hold the walk
kernel's rows and the segment pointers), then times `n` launches of subgacc_sjoin_fill_keyrows(64) over those buffers with HIP
events on the launch stream; prints ms, algorithmic bytes and the fraction of the 8 TB/s peak.  `-` = the shipped
library.  JB_PITCH=n: the same rows laid out again n words apart first (how the 128-byte-aligned pitch was measured before it became
subgacc_walk_cfg::row_pitch; StepBuffers have it now).  Variant libraries are built HERE into tools/build/ (they travel to the GPU box): tools/join_bench.py --build "-DX=1" name."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SUBGACC_QUIET", "1")
CSRC = os.path.join(ROOT, "surel_plus_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off".split()


def build(flags, name, files=("sjoin.hip",)):
    out = os.path.join(ROOT, "tools", "build", f"libsubgacc_{name}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs = [os.path.join(CSRC, "build", f) for f in os.listdir(os.path.join(CSRC, "build")) if f.endswith(".o") and f[:-2] + ".hip" not in files]
    hooks = ["-include", os.path.join(ROOT, "tools", "dev_hooks.hpp")] if "EXPERIMENT" in flags else []
    for f in files:
        o = f"/tmp/jb_{name}_{f[:-4]}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + hooks + flags.split() + ["-c", os.path.join(CSRC, f), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    print("built", out)


def one(wl, n):
    import torch
    import bench
    import surel_plus_amd as sp
    from surel_plus_amd._lib import check, lib, ptr, stream_ptr
    from surel_plus_amd.graphs import preset_graph, query_pairs
    from surel_plus_amd.spjoin import _arange_segments
    preset, M, k, _, pos = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    csr = preset_graph(preset, device=dev)
    B = 65536
    bufs = sp.StepBuffers(csr, B, M, k - 1)
    e = query_pairs(csr, B, seed=1, device=dev, pos_frac=pos)
    xz, ind, sets = sp.sample_and_gather(csr, e, num_walks=M, num_steps=k - 1, seed=1, rng="philox", buffers=bufs, lazy=True)
    sets.prefetch(extra=ind[-1:]).resolve()
    rows = int(sets.extra[0])
    L, st = lib(), stream_ptr()
    own, partner = _arange_segments(B, dev, B)
    flags = bufs.status.view(torch.int32)[:4]
    from surel_plus_amd import _lib
    out = bufs.out.view(-1)
    # JB_OUT=fresh2m: the output in a buffer of its own that begins on a 2 MB boundary and was written once (round 6: does WHERE xz lies
    # explain the run-to-run spread of the 4-hop join?); JB_OUT_SHIFT=bytes: the same, shifted by that many bytes
    mode = os.environ.get("JB_OUT", "")
    if mode in ("fresh2m", "fresh_empty"):      # a buffer of its own on a 2 MB boundary: written once before the clock / never written
        big = (torch.zeros if mode == "fresh2m" else torch.empty)(out.numel() + (1 << 20), dtype=torch.float32, device=dev)
        off = (-big.data_ptr()) % (2 << 20) + int(os.environ.get("JB_OUT_SHIFT", "0"))
        out = big[off // 4: off // 4 + out.numel()]
    elif mode == "pretouch":                     # the step buffers' own output, every page of it written once before the clock
        out.zero_()
    elif mode == "free_first":                   # the step buffers' own output, after the caching allocator gave its free blocks back
        torch.cuda.empty_cache()
    # JB_PITCH=n (experiment): the same rows laid out again n words apart (n a multiple of 32: every row begins on a 128-byte line)
    stride, ids, slot = bufs.stride, bufs.ids, bufs.slot
    pitch = int(os.environ.get("JB_PITCH", "0"))
    if pitch:
        assert pitch >= stride
        ids2 = torch.zeros(2 * B * pitch, dtype=ids.dtype, device=dev)
        slot2 = torch.zeros(2 * B * pitch, dtype=slot.dtype, device=dev)
        ids2.view(2 * B, pitch)[:, :stride] = ids[: 2 * B * stride].view(2 * B, stride)
        slot2.view(2 * B, pitch)[:, :stride] = slot[: 2 * B * stride].view(2 * B, stride)
        stride, ids, slot = pitch, ids2, slot2
        wl = f"{wl}@{pitch}"

    def launch():        # (ABI 7: through the descriptor; builds older than round 6 need the round-5 version of this script)
        _lib.join_fill(_lib.JOIN_ROWS, _lib.JOIN_KEY64 if bufs.key64 else _lib.JOIN_KEY32, row_len=bufs.nsize, n_rows=2 * B, row_stride=stride,
                       ids=ids, payload=slot, own=own, partner=partner, S=2 * B, seg=bufs.seg, pair_block=B, num_walks=M, num_steps=k - 1,
                       out_xz=out, flags=flags)
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            launch()
        b.record()
        b.synchronize()
        best.append(a.elapsed_time(b) / n)
    ms = sorted(best)[1]
    abytes = B * 64 + rows * (8 + 8 * k)
    where = "  ".join(f"{nm} %2M={t.data_ptr() % (2 << 20):>8d} %1G={t.data_ptr() % (1 << 30) >> 20:>4d}M" for nm, t in (("xz", out), ("ids", ids), ("keys", slot)))
    print(f"{wl:8s} lib={os.environ.get('SUBGACC_LIB', '-'):40s} rows/pair {rows / B:6.1f}  join {ms:.4f} ms (min {min(best):.4f})  "
          f"{abytes / ms / 1e9:.2f} TB/s  frac {abytes / (ms * 1e-3) / 8e12:.3f}   {where}", flush=True)


def one_store(wl, n):
    """the reference's own flow: subg_matrix over all N nodes once, then gather(edge, z, encode=Z_SF) per batch -- the fill kernel of
    the TABLE join (payload SFptr+1, features from the Z_SF table) and of the keyed store, HIP events around the fill launch"""
    import numpy as np
    import torch
    import bench
    import surel_plus_amd as sp
    from surel_plus_amd import sampler as sampler_mod
    from surel_plus_amd.graphs import preset_graph, query_pairs
    preset, M, k, _, pos = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    csr = preset_graph(preset, device=dev)
    z, enc = sp.subg_matrix(csr, np.arange(csr.num_nodes), num_walks=M, num_steps=k, rng="philox", seed=3)
    table = torch.from_numpy(enc.astype(np.float32) / np.float32(M)).to(dev)
    zk = z.keyed(enc, M)
    B = 65536
    e = query_pairs(csr, B, seed=1, device=dev, pos_frac=pos)
    buf = torch.empty(2 * B * z.max_len * 2 * k, dtype=torch.float32, device=dev)
    za = zk.aligned()
    for name, store, encode in (("table", z, table), ("keyed", zk, zk.slot_table()), ("keyed-aligned", za, za.slot_table())):
        timer = bench.KernelTimer()
        sampler_mod.KERNEL_TIMER = timer
        rows = None
        for it in range(n + 5):
            timer.enabled = it >= 5
            xz, ind = sp.gather(e, store, dev, ptr=True, encode=encode, out=buf, lazy=True)
        torch.cuda.synchronize()
        rows = int(ind[-1].item())
        ms = timer.mean_ms("sjoin_fill")[0]
        abytes = B * 64 + rows * (8 + 8 * k)
        print(f"{wl:8s} {name:13s} store lib={os.environ.get('SUBGACC_LIB', '-'):40s} rows/pair {rows / B:6.1f}  fill {ms:.4f} ms  "
              f"{abytes / ms / 1e9:.2f} TB/s  frac {abytes / (ms * 1e-3) / 8e12:.3f}", flush=True)
    sampler_mod.KERNEL_TIMER = None


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--one-store":
        return one_store(sys.argv[2], int(sys.argv[3]))
    if len(sys.argv) > 1 and sys.argv[1] == "--build":
        return build(sys.argv[2], sys.argv[3], tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else ("sjoin.hip",))
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        return one(sys.argv[2], int(sys.argv[3]))
    opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    for wl in opts.get("wl", "collab,twitter").split(","):
        for rep in range(int(opts.get("reps", "1"))):       # (processes of one build differ by a few per cent: alternate, repeat)
            for libp in opts.get("libs", "-").split(","):
                env = dict(os.environ)
                env.pop("SUBGACC_LIB", None)
                if libp != "-":
                    env["SUBGACC_LIB"] = os.path.join(ROOT, libp)
                subprocess.call([sys.executable, os.path.abspath(__file__), "--one-store" if "store" in opts else "--one", wl, opts.get("n", "50")], env=env)


if __name__ == "__main__":
    main()
