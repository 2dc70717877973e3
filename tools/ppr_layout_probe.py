#!/usr/bin/env python3
"""Dev-only (round 6): the float join of the cit2-PPR store by row layout -- packed CSR rows against headed rows (SpG.aligned()) at
several pitches and speculation lengths: the fill kernel alone, ten launches back to back between one event pair, 3 repeats.
    ppr_layout_probe.py [synth]        (synth: exactly 100 random members per row instead of the real top-100 store)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import surel_plus_amd as sp  # noqa: E402
from surel_plus_amd import _lib  # noqa: E402
from surel_plus_amd.graphs import ppr_like_spg, preset_graph  # noqa: E402

dev = torch.device("cuda", 0)
B = 65536
if len(sys.argv) > 1 and sys.argv[1] == "synth":
    zp = ppr_like_spg(2_927_963, 100, seed=3, device=dev)
else:
    from surel_plus_amd.ppr import topk_ppr_matrix
    csr = preset_graph("cit2", device=dev)
    zp = topk_ppr_matrix(csr, 0.1, 1e-4, torch.arange(csr.num_nodes, dtype=torch.int32, device=dev), 100, normalization="sym", encode=True)
    del csr
N = zp.n_rows
lens = (zp.indptr[1:] - zp.indptr[:-1]).float()
print(f"rows {N}, members {zp.nnz}, mean length {lens.mean().item():.1f}, median {lens.median().item():.0f}, "
      f"share of rows with >= 96 members {(lens >= 96).float().mean().item():.3f}, == 100: {(lens == 100).float().mean().item():.3f}", flush=True)
e = torch.randint(0, N, (2, B), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
own = e.contiguous().view(-1)
xz, ind = sp.gather(e, zp, dev, ptr=True, encode=None)
R = int(xz.shape[0])
abytes = B * 64 + R * 20
out = torch.empty(R * 2, dtype=torch.float32, device=dev)
flags = torch.zeros(4, dtype=torch.int32, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def time_fill(label, **layout):
    def launch():
        _lib.join_fill(_lib.JOIN_ROWS, _lib.JOIN_F64, n_rows=N, own=own, S=own.numel(), seg=ind, pair_block=B, out_xz=out, flags=flags, **layout)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    assert torch.equal(out.view(R, 2, 1), xz), label
    best = []
    for _ in range(3):
        ev0.record()
        for _ in range(10):
            launch()
        ev1.record()
        torch.cuda.synchronize()
        best.append(ev0.elapsed_time(ev1) / 10)
    ms = sorted(best)[1]
    print(f"{label:34s} {ms * 1e3:7.1f} us   {abytes / (ms * 1e-3) / 8e12:.3f} of the 8 TB/s peak", flush=True)


time_fill("packed", row_off=zp.indptr, ids=zp.indices, payload=zp.data, max_len=zp.max_len)
for pitch in (128, 160, 224):
    za = zp.aligned(pitch=pitch)
    for spec in (1, 64, 96, 128):
        time_fill(f"headed pitch {pitch} spec {spec}", row_stride=za.pitch, ids=za.ids, payload=za.data, max_len=spec)
    del za
