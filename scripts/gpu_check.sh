#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sparse.py -m gpu -q -x > $OUT/r3_pytest_lat4.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3_pytest_lat4.log
SWEEP_BS=256 timeout 600 python scripts/latency_sweep.py > $OUT/r3_latency_sweep_scalar.jsonl 2> $OUT/r3_sweep.err; echo "sweep rc=$?"
timeout 300 python scripts/bench_sparse.py > $OUT/r3_sparse_400k_final.json 2> $OUT/r3_sparse.err; echo "sparse rc=$?"
python - <<'P'
import json
for l in open("gpurun_out/r3_latency_sweep_scalar.jsonl"):
    j=json.loads(l)
    if "variant" in j: print(j["ef"], j["B"], j["variant"], "ms %.3f qps %.0f" % (j["ms"], j["qps"]))
j=json.loads(open("gpurun_out/r3_sparse_400k_final.json").read()); print("sparse host ms", j["ms_per_batch_host_api"], "kernel ms", j["roofline"]["per_launch"]["avg_ms"], "frac", j["roofline"]["frac"], j["parity_vs_oracle"])
P
