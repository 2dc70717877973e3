"""The driver parses bench.py's LAST stdout line; round 4's 31 KB line was lost (BENCH_r04.parsed = null).  This guards the
emitter: whatever the run collected, the line stays under 6 KB, carries the contract's keys, `roofline` and `cpu_baseline`,
and at most 40 scalars of `config`; everything else lands in bench_detail.json."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fat_record(bench):
    """a record shaped like a --full run's, with every optional block present and prose in the places prose used to be"""
    wl = {"metric": "query-pairs/sec (sample+SpJoin)", "value": 1.23456789e8, "unit": "query-pairs/s", "steps": 100, "warmup": 3,
          "ms_per_step": 1.123456789, "dtype": "int32",
          "config": {"workload": "w" * 300, "pairs_per_s_min": 1.0, "pairs_per_s_max": 2.0, "stage_ms": {"sjoin_fill": 0.4},
                     "dedup_roots_loop": {"pairs_per_s": 3.0}, "join_call_ms": 0.1,
                     "frac_of_hbm_peak_whole_join_call": 0.4321, "region_pairs_per_s": list(range(50))},
          "roofline": {"frac": 0.2, "join_frac": 0.55, "kernel_ms": 0.6, "traffic": 4.5e9,
                       "random_line_roof": {"frac": 1.0, "source": "s" * 400}},
          "cpu_baseline": {"value": 7.6e4, "cores": 16, "t1_pairs_per_s": 2.6e4, "t16_pairs_per_s": 7.6e4},
          "literal_dropin": {"drop_in_roots_per_s": 5.2e5, "shim_roots_per_s": 6e6, "shim_seconds": 0.04, "caller_seconds": 0.4,
                             "reference_roots_per_s": 1.8e5}}
    others = {name: json.loads(json.dumps(wl)) for name in
              ("cit2 (rng=rand_r: the reference's own stream, bit-exact mode)", "cit2m4", "collab", "ppa", "twitter", "cit2ppr", "cit2loc")}
    others["walk_sampler (collab)"] = {"value": 2.2e8, "roofline": {"frac": 0.1}}
    return {
        "metric": "query-pairs/sec (sample+SpJoin)", "value": 58934567.123456, "unit": "query-pairs/s", "n_gpus": 8, "steps": 20,
        "warmup": 5, "ms_per_step": 1.1123456, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": "cit2-like LP: " + "x" * 400, "pairs_per_step_per_gpu": 65536, "roots_per_step_per_gpu": 131072,
                   "pairs_per_step_all_gpus": 524288, "ranks_seen": 8, "distinct_devices": 8, "per_rank_ms_min": 1.1, "per_rank_ms_max": 1.2,
                   "dist_backend": "nccl", "rccl_version": "2.26.6", "num_walks": 200, "num_steps_cli": 4, "rng": "philox",
                   "parallelism": "query-shard x8",
                   "rank_records": [{"rank": r, "device": "AMD Instinct MI355X|" + "u" * 64, "elapsed_ms": 22.0, "walk_kernel_ms": 0.66,
                                     "host": "h" * 40, "pid": 12345} for r in range(8)],
                   "region_pairs_per_s": [58e6, 59e6, 59.1e6, 59.2e6], "stage_ms": {"walk_sets": 0.66, "sjoin_fill": 0.42},
                   "other_workloads": others, "host_fed_pairs_per_s": 5.9e7,
                   "offline_flow": {"S_roots_per_s": 1.6e8, "J_pairs_per_s_table_store": 1.45e8, "Q_formula_pairs_per_s": 5.2e7,
                                    "Q_amortised_at_1e8_pairs_table": 1.41e8, "junk": ["j" * 100] * 50},
                   "hgather": {"value": 2.7e7, "roofline": {"frac": 0.21}}, "mean_stage": {"H96_fused_pairs_per_s": 7.2e7},
                   "batch_size_and_hip_graph": {"B=1024 eager": {"pairs_per_s": 1.5e7}}},
        "roofline": {"bound": "hbm", "kernel": "walk_rows_kernel<false,1,3,8,128,8,true,true>", "achieved": 1601.23456, "peak": 8000.0,
                     "unit": "GB/s", "frac": 0.2001543, "traffic": 4578123456.7, "kernel_ms": 0.6581234, "launches_timed": 20,
                     "algorithmic_bytes_per_launch": 1052413276, "join_kernel_ms": 0.4241, "join_algorithmic_bytes_per_launch": 1997000000,
                     "join_achieved": 4710.0, "join_frac": 0.589, "join_traffic": 2.1e9, "step_traffic_TBps": 5.9,
                     "random_line_roof": {"lines_per_s": 5.48e10, "frac": 1.01, "l2_miss_lines_per_launch": 36.06e6, "source": "p" * 500}},
        "cpu_baseline": {"value": 23100.5, "unit": "query-pairs/s", "cores": 16, "kind": "reference", "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                         "host_threads": 256, "t1_pairs_per_s": 4800.1, "t16_pairs_per_s": 23100.5, "best_nthread": 16,
                         "probe_best_nthread": 8, "sampler_roots_per_s": 5e4, "join_pairs_per_s": 7.5e4, "cpu_seconds": 17.0,
                         "settings": {"t1": {"x": ["y" * 50] * 40}}, "probe_seconds_2048_pairs": {str(i): 0.1 for i in range(8)},
                         "sample": "z" * 900},
    }


def test_compact_line_is_small_and_complete():
    import bench
    out = fat_record(bench)
    assert len(json.dumps(out)) > 12000          # the kind of record that was lost in round 4
    bench.flatten(out)
    out["config"]["detail"] = "bench_detail.json"
    text = bench.compact_line(out)
    assert len(text) < 6000 and "\n" not in text
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["value"] == pytest.approx(out["value"], rel=1e-5) and line["n_gpus"] == 8 and line["vs_baseline"] is None
    cfg = line["config"]
    assert len(cfg) <= 40 and all(not isinstance(v, (dict, list)) for v in cfg.values())
    for key in ("workload", "pairs_per_step_per_gpu", "ranks_seen", "distinct_devices", "rccl_version", "rng",
                "rand_r_pairs_per_s", "cit2m4_pairs_per_s", "collab_pairs_per_s", "ppa_pairs_per_s", "twitter_pairs_per_s",
                "cit2ppr_pairs_per_s", "collab_frac", "collab_join_frac", "twitter_join_frac", "cit2m4_join_frac",
                "cit2ppr_frac_whole_join_call", "collab_dropin_shim_roots_per_s", "headline_median_of_3_x100", "detail"):
        assert key in cfg, key
    # the driver's record keeps the FIRST 24 keys of `config` (BENCH_r05 lost ppa / twitter / cit2-PPR off the end): every BASELINE.json
    # configuration, SURVEY 8(d)'s S / J / Q and the host-fed loop must sit in front of that cut -- position, not just presence
    head = list(cfg)[:bench.DRIVER_KEEPS]
    assert bench.DRIVER_KEEPS == 24
    for key in ("workload", "pairs_per_step_per_gpu", "ranks_seen", "distinct_devices",
                "collab_cpu_pairs_per_s",                                                    # configs[0]
                "collab_pairs_per_s", "collab_frac", "collab_join_frac",                     # configs[1]
                "ppa_pairs_per_s", "ppa_frac", "ppa_join_frac",                              # configs[2]
                "cit2ppr_pairs_per_s", "cit2ppr_frac", "cit2ppr_frac_whole_join_call",       # configs[3]
                "twitter_pairs_per_s", "twitter_frac", "twitter_join_frac",                  # configs[4]
                "S_offline_roots_per_s", "J_resident_pairs_per_s", "Q_formula_pairs_per_s", "host_fed_pairs_per_s"):
        assert key in head, (key, head)
    rl = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rl)
    assert rl["frac"] == pytest.approx(rl["achieved"] / rl["peak"], rel=1e-4)
    assert all(not isinstance(v, (dict, list)) for v in rl.values()) and "line_roof_frac" in rl
    cb = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and len(cb["sample"]) <= 200
    assert cb["cores"] == cb["best_nthread"]          # the team size that produced `value`, not the probe's pick
    assert "settings" not in cb and "probe_seconds_2048_pairs" not in cb


def test_emit_prints_one_line_and_writes_the_detail_file(tmp_path, monkeypatch, capsys):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.mkdir(tmp_path / "gpurun_out")
    out = fat_record(bench)
    bench.emit(out)
    printed = capsys.readouterr().out
    assert printed.endswith("\n") and printed.count("\n") == 1 and len(printed) < 6000
    line = json.loads(printed)
    assert line["config"]["detail"] == "bench_detail.json"
    for p in (tmp_path / "bench_detail.json", tmp_path / "gpurun_out" / "bench_detail.json"):
        full = json.load(open(p))
        assert "other_workloads" in full["config"] and "rank_records" in full["config"]          # nothing is dropped, only moved
        assert full["config"]["collab_pairs_per_s"] == pytest.approx(1.23456789e8)


def test_minimal_record_without_optional_blocks():
    import bench
    out = {k: v for k, v in fat_record(bench).items() if k != "cpu_baseline"}
    out["config"] = {k: v for k, v in out["config"].items() if k in ("workload", "pairs_per_step_per_gpu", "ranks_seen")}
    bench.flatten(out)
    line = json.loads(bench.compact_line(out))
    assert "cpu_baseline" not in line and line["config"]["ranks_seen"] == 8


def test_traffic_json_cites_tracked_files():
    """VERDICT r4 (weak 4): `roofline.traffic` comes from profiles/traffic.json; every entry names the counter table it was computed
    from, and that table has to be in the tree a reader gets -- not in a git-ignored scratch directory."""
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert tj
    for key, entry in tj.items():
        src = entry.get("source", "")
        assert src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src)), (key, src)
        assert entry.get("kernel_source_sha")
