#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_walk_table.py -m gpu -q -x > $OUT/r04_b8_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r04_b8_pytest.log
timeout 200 python scripts/single_batch_probe.py > $OUT/r04_b8_single_batch.jsonl 2> $OUT/r04_b8_single_batch.err; cat $OUT/r04_b8_single_batch.jsonl
COS_BENCH_FULL_RECORD=r04_b8_bench_c2_full.json timeout 400 python bench.py --configs none --ef-sweep "" --no-cpu-baseline > $OUT/r04_b8_bench_c2.json 2> $OUT/r04_b8_bench_c2.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b8_bench_c2.json")); print(r["value"], r["single_batch_qps"]); print(json.dumps(r["host_api_pcie_inclusive"])[:1500]); print(r["roofline"]["per_launch"])
except Exception as e: print("parse", e)
PY
tail -3 $OUT/r04_b8_bench_c2.err
