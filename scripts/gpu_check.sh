#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_meta.py -m gpu -q > $OUT/r3_pytest_part.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3_pytest_part.log
timeout 300 python scripts/host_api_sweep.py > $OUT/r3_host_api_sweep.json 2> $OUT/r3_host.err; echo "host sweep rc=$?"; cat $OUT/r3_host_api_sweep.json
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/r3_bench_full.json 2> $OUT/r3_bench_full.err; echo "bench rc=$?"; tail -3 $OUT/r3_bench_full.err
