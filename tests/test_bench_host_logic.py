"""Host-side logic of bench.py that needs no GPU: the ef selection rule (selection set / confidence bound) and the
usable-core count."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_recall_stats_mean_stderr_lower_bound():
    import bench
    hits = np.array([10, 9, 10, 8, 10, 10, 9, 10])
    mean, se, lower = bench.recall_stats(hits, 10)
    assert abs(mean - 0.95) < 1e-12
    assert abs(se - (hits / 10).std(ddof=1) / np.sqrt(8)) < 1e-12
    assert abs(lower - (mean - 1.645 * se)) < 1e-12
    assert bench.recall_stats([10], 10) == (1.0, 0.0, 1.0)


def test_select_ef_takes_the_smallest_ef_whose_lower_bound_clears_the_target():
    import bench
    table = {32: (0.93, 0.001, 0.928), 48: (0.9505, 0.001, 0.9489), 64: (0.955, 0.001, 0.9534), 96: (0.97, 0.001, 0.968)}
    asked = []

    def measure(ef):
        asked.append(ef)
        return table[ef]
    ef, rows = bench.select_ef([32, 48, 64, 96], measure, 0.95)
    assert ef == 64 and asked == [32, 48, 64]          # 48 has mean >= 0.95 but not with 95 % confidence; 96 is never tried
    assert [r["ef_search"] for r in rows] == [32, 48, 64] and rows[-1]["lower95"] == 0.9534
    ef, rows = bench.select_ef([32, 48], measure, 0.95)  # nothing qualifies: the largest candidate is used
    assert ef == 48 and len(rows) == 2


def test_effective_cores_is_bounded_by_affinity():
    import bench
    n = bench.effective_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_resolve_configs_auto_all_none_and_world_rules():
    import bench
    main = bench.MAIN_WORKLOAD
    assert main == "c4shard" and (bench.MAIN_M0, bench.MAIN_M) == (256, 64)             # `value` = the metric's own configuration
    assert bench.resolve_configs("auto", 1, main) == bench.ALL_CONFIGS                  # every other BASELINE config at N = 1
    assert bench.resolve_configs("auto", 8, main) == []                                 # N > 1: the main workload IS configs[3]
    assert bench.resolve_configs("auto", 1, "smoke") == [] and bench.resolve_configs("none", 1, main) == []
    assert bench.resolve_configs("auto", 1, "c2") == []                                 # a non-standard main workload runs alone
    assert bench.resolve_configs("c3,c5", 1, main) == ["c3", "c5"]
    assert bench.resolve_configs("all", 2, main) == []                  # single-GPU configs (the one-device 8-shard proxy too) are dropped at N > 1
    assert bench.resolve_configs("c4shard_ref,c4shard_ref_m0_256_m_64", 2, main) == ["c4shard_ref", "c4shard_ref_m0_256_m_64"]
    assert {"c4_8shards_one_device", "c2", "c2_sigma01", "c2_uniform", "c3", "c5"} == set(bench.ALL_CONFIGS) and "c4shard_ref" in bench.OPTIONAL_CONFIGS
    assert bench.resolve_configs("c4shard_exact,c4shard_ref_m0_128", 1, main) == ["c4shard_exact", "c4shard_ref_m0_128"]   # optional records
    import pytest
    with pytest.raises(SystemExit):
        bench.resolve_configs("c9", 1, main)


def test_value_is_quoted_on_the_metrics_configuration_at_every_n():
    """`value` at --gpus N = N id-range shards of 12.5M x 1024 (BASELINE configs[3]; the metric says 1024-dim, 1/2/4/8 GPUs): the name the
    real line carries in config.workload, for N = 1 (BENCH) and N = 8 (SCALE) alike"""
    import bench
    for n in (1, 2, 4, 8):
        w = bench.value_workload(n)
        assert "configs[3]" in w and "12500000 x 1024" in w and f"{n} id-range shard" in w
        assert ("all-gather" in w) == (n > 1)
    assert "1024-dim" in bench.METRIC and bench.WORKLOADS[bench.MAIN_WORKLOAD][:2] == (12_500_000, 1024)


def test_launcher_command_is_the_drivers_own():
    import bench
    cmd = bench.launcher_command(4, 29999, ["--gpus", "4", "--steps", "3"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert cmd[cmd.index("--master-port") + 2].endswith("bench.py")


def test_stdout_line_keeps_every_result_and_drops_the_prose():
    """bench.py prints a slim line (a few KB) and writes the complete record to a file: every config's qps / recall / roofline /
    cpu_baseline / parity block must survive the slimming, floats are rounded to 9 significant digits"""
    import json
    import bench
    full = bench.rounded(json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_record_sample.json"))))
    line = bench.slim_line(full)
    assert len(json.dumps(line)) < 16_000 < len(json.dumps(full))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity_vs_oracle", "recall_at_10", "single_batch_qps", "host_api_pcie_inclusive", "configs"):
        assert k in line, k
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert set(line["configs"]) == set(full["configs"]) == {"c2_uniform", "c5", "c3", "c4shard_ref_m0_256_m_64", "c4_8shards_one_device"}
    assert line["configs"]["c3"]["hnsw_walk_quaternary"]["parity_vs_oracle"]["id_mismatch_queries"] == 0        # c3 (ii): the quaternary walk survives the slimming
    assert line["value_at_query_batch_256"]["min_over_caller_counts"] >= 3.0e6                                  # the reference's calling pattern, top level
    assert line["roofline"]["step_frac"] == full["roofline"]["step_frac"] and line["configs"]["c5"]["roofline"]["parts"]["bm25_score"]["frac"] > 0.5
    for name, c in line["configs"].items():
        f = full["configs"][name]
        if name == "c4_8shards_one_device":       # the 8-shard proxy: merged recall, exchange + merge cost, the merge property — no kernel of its own
            assert c["merged_recall_at_10"] == f["merged_recall_at_10"] >= 0.95 and c["merged_equals_merge_of_shard_answers"] is True
            assert c["shards_in_merged_answers"] == list(range(8)) and c["ms_exchange_plus_merge"] < 1.0 and c["shards"] == 8
            continue
        assert c["qps"] == f["qps"] and c["roofline"]["frac"] == f["roofline"]["frac"] and c["roofline"]["bound"] in ("hbm", "mfma")
        assert c["cpu_baseline"]["value"] == f["cpu_baseline"]["value"] and c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1
        p = c["parity_vs_oracle"]
        assert p["queries"] >= 256 and sum(v for k, v in p.items() if k != "queries") == 0
    assert bench.rounded(1.23456789012345) == 1.23456789 and bench.rounded({"a": [float("inf"), 2]}) == {"a": [float("inf"), 2]}


def test_value_at_query_batch_256_takes_the_best_setting_per_caller_count_and_skips_bad_runs():
    import bench
    host = {"concurrent_256_query_callers": [
        {"callers": 64, "coalescing_max_queries": 8192, "qps": 3.2e6, "failed_calls": 0, "identical_to_uncoalesced_call": True},
        {"callers": 64, "coalescing_max_queries": 16384, "qps": 2.7e6, "failed_calls": 0, "identical_to_uncoalesced_call": True},
        {"callers": 128, "coalescing_max_queries": 16384, "qps": 3.9e6, "failed_calls": 0, "identical_to_uncoalesced_call": True},
        {"callers": 128, "coalescing_max_queries": 32768, "qps": 9.9e6, "failed_calls": 3, "identical_to_uncoalesced_call": True},     # failed calls: ignored
        {"callers": 256, "coalescing_max_queries": 32768, "qps": 8.8e6, "failed_calls": 0, "identical_to_uncoalesced_call": False}]}   # wrong answers: ignored
    v = bench.value_at_batch_256(host)
    assert v["pcie_inclusive"] and set(v["by_callers"]) == {"64", "128"}
    assert v["by_callers"]["64"] == {"qps": 3.2e6, "coalescing_max_queries": 8192} and v["by_callers"]["128"]["qps"] == 3.9e6
    assert v["min_over_caller_counts"] == 3.2e6
    assert bench.value_at_batch_256(None) is None and bench.value_at_batch_256({"concurrent_256_query_callers": []}) is None
