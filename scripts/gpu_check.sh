#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_sparse.py -m gpu -q > $OUT/r3_pytest_sparse.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3_pytest_sparse.log
timeout 300 python scripts/bench_sparse.py > $OUT/r3_sparse_400k_branch_free.json 2> $OUT/r3_sparse.err; echo "sparse rc=$?"
python - <<'P'
import json
j=json.loads(open("gpurun_out/r3_sparse_400k_branch_free.json").read()); print("sparse host ms", j["ms_per_batch_host_api"], "kernel ms", j["roofline"]["per_launch"]["avg_ms"], "frac", j["roofline"]["frac"], j["parity_vs_oracle"])
P
