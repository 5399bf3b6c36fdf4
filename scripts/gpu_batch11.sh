#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_walk_table.py tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -q -x > $OUT/r04_b11_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r04_b11_pytest.log
COS_BENCH_FULL_RECORD=r04_b11_bench_c2_full.json timeout 400 python bench.py --configs none --ef-sweep "" --no-cpu-baseline --no-hbm-probe > $OUT/r04_b11_bench_c2.json 2> $OUT/r04_b11_bench_c2.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b11_bench_c2.json")); print(r["value"], r["single_batch_qps"], r["single_batch_qps_one_wave_latency_kernel"], r["single_batch_qps_throughput_kernel"], r["single_batch_latency_walk_identical_to_throughput_walk"]); print(json.dumps(r["host_api_pcie_inclusive"]["concurrent_256_query_callers"]))
except Exception as e: print("parse", e)
PY
COS_FORCE_DIST=1 COS_BENCH_FULL_RECORD=r04_b11_bench_c2_forced_dist_full.json timeout 400 python bench.py --configs none --ef-sweep "" --no-cpu-baseline --no-hbm-probe > $OUT/r04_b11_bench_c2_forced_dist.json 2> $OUT/r04_b11_bench_c2_forced_dist.err; echo "dist rc=$?"; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b11_bench_c2_forced_dist.json")); print("forced dist", r["value"], r["config"].get("launches_in_flight"))
except Exception as e: print("parse", e)
PY
tail -2 $OUT/r04_b11_bench_c2_forced_dist.err
