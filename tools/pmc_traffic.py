"""Dev-only: per-kernel means of the PMC passes of tools/pmc_collect.sh (with the gfx950 byte correction: FETCH_SIZE
tallies a 128-byte line request at 64 bytes -- MI355X_MICROARCH.md, HBM section; confirmed for random 4-byte reads by
tools/line_probe.hip: one request per missed 128-byte line) and the matching entry of profiles/traffic.json.

    python tools/pmc_traffic.py OUTDIR TAG [bench args that were used]

The per-kernel table is WRITTEN to profiles/TAG_pmc_per_launch.csv (and printed, and copied into OUTDIR so that gpurun brings
it back) BEFORE profiles/traffic.json is stamped with it: an entry's `source` always names a file that exists under profiles/
(tests/test_bench_line_cpu.py checks the tracked tree the same way)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out, tag, bargs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "subgacc" in k or "compact_rows" in k:
            acc[k.split("(")[0].replace("void subgacc::", "").replace("subgacc::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
table = ["kernel,launches,FETCH_SIZE,WRITE_SIZE,TCC_REQ_sum,TCC_HIT_sum,TCC_MISS_sum,hbm_bytes_per_launch=(2*FETCH_SIZE+WRITE_SIZE)*1024"]
rows = []
JOIN_KERNELS = ("sjoin_pair_kernel", "sjoin_keypair_kernel", "sjoin_f64pair_kernel")      # the fill kernel of a step, whichever form
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        rows.append((-(2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]), k, len(c["FETCH_SIZE"]), m))
walk = None
join_k = None
for _, k, n, m in sorted(rows)[:12]:
    hbm = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    table.append(f"\"{k}\",{n},{m['FETCH_SIZE']:.0f},{m['WRITE_SIZE']:.0f},{m.get('TCC_REQ_sum', 0):.0f},{m.get('TCC_HIT_sum', 0):.0f},{m.get('TCC_MISS_sum', 0):.0f},{hbm:.0f}")
    if (k.startswith("walk_sets_kernel") or k.startswith("walk_rows_kernel") or k.startswith("walk_pipe_kernel")) and \
            (walk is None or n > walk[4]):      # the walk kernel of the timed steps: the one with the most launches
        walk = (k, hbm, m.get("TCC_MISS_sum", 0), m.get("TCC_REQ_sum", 0), n)
    if k.startswith(JOIN_KERNELS) and (join_k is None or n > join_k[3]):       # the join of the timed steps
        join_k = (k, hbm, m.get("TCC_MISS_sum", 0), n)
csv_name = f"{tag}_pmc_per_launch.csv"
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
for dest in (os.path.join(ROOT, "profiles", csv_name), os.path.join(out, csv_name)):
    with open(dest, "w") as fh:
        fh.write("\n".join(table) + "\n")
print("\n".join(table))
assert os.path.exists(os.path.join(ROOT, "profiles", csv_name))        # what the entries below cite
# the bench line of one of the passes tells the configuration (workload, B, M, k, layout, rng)
line = None
for f in glob.glob(os.path.join(out, "*.json")):
    for ln in open(f):
        if ln.startswith('{"metric"'):
            line = json.loads(ln)
if line and "PPR" in line.get("metric", ""):      # the float join: its fill kernel's traffic, keyed like bench_ppr looks it up
    import bench
    join = [(k, m) for _, k, n, m in sorted(rows) if k.startswith(JOIN_KERNELS)]
    if join:
        k, m = join[0]
        B = line["config"]["pairs_per_step_per_gpu"]
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        tj = {kk: v for kk, v in tj.items() if isinstance(v, dict) and "kernel_source_sha" in v}
        layout = os.environ.get("SUBGACC_PPR_LAYOUT", "aligned")       # (bench_ppr's default store layout since round 6: headed rows)
        tj[f"cit2ppr:{B}:join" + (":aligned" if layout == "aligned" else "")] = {"join_hbm_bytes_per_launch": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024,
                                   "join_l2_miss_lines_per_launch": m.get("TCC_MISS_sum", 0), "kernel": k,
                                   "kernel_source_sha": bench.kernel_source_sha(), "source": f"profiles/{tag}_pmc_per_launch.csv",
                                   "how": "tools/pmc_collect.sh ... --workload cit2ppr (SUBGACC_PPR_EAGER=1: eager launches); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"}
        json.dump(tj, open(tpath, "w"), indent=1)
        json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    walk = None
if walk and line:
    import bench
    cfg = line["config"]
    name = "cit2"
    for i, a in enumerate(bargs):
        if a == "--workload":
            name = bargs[i + 1]
    # (M and k from the workload table: the compact line no longer carries them; the step's rows are fused rows for every preset)
    key = f"{name}:{cfg['pairs_per_step_per_gpu']}:{bench.WORKLOADS[name][1]}:{bench.WORKLOADS[name][2]}:spg:{cfg['rng']}"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    tj = {k: v for k, v in tj.items() if isinstance(v, dict) and "kernel_source_sha" in v}     # entries without a hash are stale
    tj[key] = {"walk_sets_hbm_bytes_per_launch": walk[1], "walk_sets_l2_miss_lines_per_launch": walk[2],
               "walk_sets_l2_requests_per_launch": walk[3], "kernel": walk[0], "launches_averaged": walk[4],
               "join_hbm_bytes_per_launch": join_k[1] if join_k else None, "join_l2_miss_lines_per_launch": join_k[2] if join_k else None,
               "join_kernel": join_k[0] if join_k else None,
               "kernel_source_sha": bench.kernel_source_sha(), "source": f"profiles/{tag}_pmc_per_launch.csv",
               "how": "tools/pmc_collect.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum "
                      "in separate passes over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-others`; mean over the "
                      "walk kernel's dispatches; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (KB units; gfx950 tallies a 128-B line "
                      "request at 64 B: MI355X_MICROARCH.md HBM section, and tools/line_probe.hip for random 4-B reads)"}
    json.dump(tj, open(tpath, "w"), indent=1)
    json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)      # gpurun only brings gpurun_out/ back
    print(f"# profiles/traffic.json[{key}] <- {walk[1]:.0f} B, {walk[2]:.0f} missed lines per launch (kernel sources {tj[key]['kernel_source_sha']})")
