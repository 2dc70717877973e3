#!/usr/bin/env python3
"""Dev-only: the float join's fill kernel ALONE over a resident PPR-shaped store (100 members per row, 65,536 pairs), 20 launches
back to back between one pair of events, across library builds (timing builds of tools/dev_hooks.hpp give WRONG rows on purpose).
    python tools/ppr_join_probe.py [--libs=-,tools/build/libsubgacc_x.so,...] [--one]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SUBGACC_QUIET", "1")


def one():
    import torch
    import surel_plus_amd as sp
    from surel_plus_amd import _lib as L
    from surel_plus_amd.graphs import ppr_like_spg
    dev = torch.device("cuda:0")
    N, B = 2_927_963, 65536
    z = ppr_like_spg(N, 100, seed=3, device=dev)
    e = torch.randint(0, N, (2, B), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    buf = torch.empty(2 * B * z.max_len * 2, dtype=torch.float32, device=dev)
    xz, ind = sp.gather(e, z, dev, ptr=True, encode=None, out=buf, lazy=True)
    own = e.contiguous().view(-1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = []
    for _ in range(5):
        ev0.record()
        for _ in range(20):
            L.join_fill(L.JOIN_ROWS, L.JOIN_F64, row_off=z.indptr, n_rows=z.n_rows, ids=z.indices, payload=z.data, max_len=z.max_len,
                        own=own, S=own.numel(), seg=ind, pair_block=B, out_xz=xz, flags=ind.join_flags)
        ev1.record()
        torch.cuda.synchronize()
        res.append(ev0.elapsed_time(ev1) / 20)
    res.sort()
    rows = int(ind[-1].item())
    ab = B * 64 + rows * 20
    print(f"lib={os.environ.get('SUBGACC_LIB', '-'):60s} fill {res[2] * 1e3:7.2f} us (min {res[0] * 1e3:7.2f})  {ab / res[2] / 1e9:6.2f} TB/s  frac {ab / res[2] / 1e9 / 8:5.3f}", flush=True)


if __name__ == "__main__":
    if "--one" in sys.argv:
        one()
        sys.exit(0)
    opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    for lib in opts.get("libs", "-").split(","):
        env = dict(os.environ)
        env.pop("SUBGACC_LIB", None)
        if lib != "-":
            env["SUBGACC_LIB"] = os.path.join(ROOT, lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env)
