#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_walk_table.py tests/test_gpu_hybrid.py tests/test_gpu_parity.py tests/test_gpu_concurrency.py -m gpu -q -x > $OUT/r04_b9_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r04_b9_pytest.log
timeout 200 python scripts/single_batch_probe.py > $OUT/r04_b9_single_batch.jsonl 2> $OUT/r04_b9_single_batch.err; cat $OUT/r04_b9_single_batch.jsonl
timeout 300 python scripts/bench_c5.py --cpu-seconds 0 > $OUT/r04_b9_c5.json 2> $OUT/r04_b9_c5.err; python - <<'PY'
import json
try:
    r=json.load(open("gpurun_out/r04_b9_c5.json")); print({k:r.get(k) for k in ("qps","ms_per_step","parity_vs_oracle")}); print(json.dumps(r.get("roofline"))[:400]); print(json.dumps(r.get("roofline_dense_half"))[:400]); print(json.dumps(r.get("timing", r.get("breakdown")))[:600])
except Exception as e: print("parse", e)
PY
tail -2 $OUT/r04_b9_c5.err
