#!/usr/bin/env python3
"""Dev-only (round 6): what the key-row epilogue's distribution sort sees on a preset graph -- rows of one batch sampled on the GPU,
the bucketing replayed on the host: how many sets take the finer levels (a level-1 bucket > 12 members), how crowded the parts of
level 2 are, and what a level 1 cut over a window around the root (fine buckets inside, coarse outside) would leave.
    sort_sim.py WORKLOAD [ROOTS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import surel_plus_amd as sp  # noqa: E402
from surel_plus_amd.graphs import preset_graph  # noqa: E402

wl = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
preset, M, k, _, pos = bench.WORKLOADS[wl]
csr = preset_graph(preset, device="cuda")
N = csr.num_nodes if hasattr(csr, "num_nodes") else csr.indptr.numel() - 1
roots = torch.randint(0, N, (R,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)).to(torch.int32)
z, sets = sp.sample_spg(csr, roots, num_walks=M, num_steps=k - 1, seed=3, rng="philox")
indptr = z.indptr.cpu().numpy()
ids = z.indices.cpu().numpy()
rt = roots.cpu().numpy()
NT, FINE_TOTAL, ABOVE = 128, 1023, 12
stat = {"sets": 0, "finer_now": 0, "again_now": 0, "finer_window": 0, "members": 0}
maxpart_now, maxpart_win, inblock = [], [], []
FRAC = float(os.environ.get('SIM_FRAC', '0.7'))
for r in range(R):
    s = ids[indptr[r]:indptr[r + 1]].astype(np.int64)
    ns = len(s)
    if ns < 2:
        continue
    stat["sets"] += 1
    stat["members"] += ns
    mn, mx = s.min(), s.max()
    rng_ = mx - mn + 1
    bshift = 0
    while (rng_ >> bshift) > NT:           # B <= NT equal-width buckets
        bshift += 1
    b1 = (s - mn) >> bshift
    c1 = np.bincount(b1, minlength=NT)
    if c1.max() > ABOVE:
        stat["finer_now"] += 1
        FINE = FINE_TOTAL // (ns + 1)
        lo1 = np.concatenate([[0], np.cumsum(c1)])[:-1]
        kb = c1[b1] * FINE
        off = (s - mn) - (b1 << bshift)
        idx = lo1[b1] * FINE + (off * kb >> bshift)
        c2 = np.bincount(idx)
        maxpart_now.append(c2.max())
        if c2.max() > 16:
            stat["again_now"] += 1
    # a level 1 over a window around the root: the smallest radius 2^j that holds >= SIM_FRAC of the set, 96 fine buckets over
    # [root - 2^j, root + 2^j), 16 coarse buckets on either side
    d = np.abs(s - int(rt[r]))
    j = None
    for jj in range(8, 18):
        if (d < (1 << jj)).sum() >= FRAC * ns:
            j = jj
            break
    inblock.append((d < 2048).sum() / ns)
    if j is None or (2 << j) >= rng_:
        cw = c1
    else:
        lo_w, hi_w = int(rt[r]) - (1 << j), int(rt[r]) + (1 << j)
        fw = max(1, (2 << j) // 96)
        inside = (s >= lo_w) & (s < hi_w)
        bf = ((s[inside] - lo_w) // fw).clip(0, 95)
        cf = np.bincount(bf, minlength=96)
        left, right = s[s < lo_w], s[s >= hi_w]
        cl = np.bincount(((left - mn) * 16 // max(1, lo_w - mn)).clip(0, 15), minlength=16) if len(left) else np.zeros(16, int)
        cr = np.bincount(((right - hi_w) * 16 // max(1, mx - hi_w + 1)).clip(0, 15), minlength=16) if len(right) else np.zeros(16, int)
        cw = np.concatenate([cl, cf, cr])
    maxpart_win.append(cw.max())
    if cw.max() > ABOVE:
        stat["finer_window"] += 1
print(wl, "window threshold", FRAC, stat, "mean set", round(stat["members"] / max(1, stat["sets"]), 1), "share within 2,048 ids of the root: mean %.2f p10 %.2f p90 %.2f" % (np.mean(inblock), *np.percentile(inblock, [10, 90])))
if maxpart_now:
    print("  level-2 parts now: max per set  mean %.1f  p50 %d  p90 %d  p99 %d" % (np.mean(maxpart_now), *np.percentile(maxpart_now, [50, 90, 99])))
print("  windowed level 1: max bucket per set  mean %.1f  p50 %d  p90 %d  p99 %d; sets above 12: %d, above 16: %d" % (
    np.mean(maxpart_win), *np.percentile(maxpart_win, [50, 90, 99]), (np.array(maxpart_win) > 12).sum(), (np.array(maxpart_win) > 16).sum()))
