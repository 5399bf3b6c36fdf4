#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 700 python -m pytest tests -m gpu -q -x > $OUT/r04_b7_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r04_b7_pytest.log
COS_BENCH_FULL_RECORD=r04_b7_bench_c2_full.json timeout 400 python bench.py --configs none --ef-sweep "" > $OUT/r04_b7_bench_c2.json 2> $OUT/r04_b7_bench_c2.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b7_bench_c2.json")); print(r["value"], r["single_batch_qps"], r["parity_vs_oracle"]); print(json.dumps(r["host_api_pcie_inclusive"])[:1500]); print(json.dumps(r["roofline"]["parts"])[:1800]); print(r["roofline"]["per_launch"])
except Exception as e: print("parse", e)
PY
tail -3 $OUT/r04_b7_bench_c2.err
cd /tmp; export TMPDIR=/tmp
PROBE_COLS=8192 PROBE_REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/scripts/table_probe.py > $OUT/r04_b7_probe.jsonl 2> $OUT/r04_b7_probe.err
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/r04_b7_kernel_trace.txt; head -20 $OUT/r04_b7_kernel_trace.txt; tail -1 $OUT/r04_b7_probe.jsonl | cut -c1-400
