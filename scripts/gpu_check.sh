#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
PROBE_SPLITS="default" timeout 500 python $R/scripts/locality_probe.py > $OUT/r3_locality_probe_fin.jsonl 2> $OUT/r3_locality_probe_fin.err; echo "probe rc=$?"; cat $OUT/r3_locality_probe_fin.jsonl; tail -3 $OUT/r3_locality_probe_fin.err
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_concurrency.py -m gpu -q -x > $OUT/r3_pytest_fin.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3_pytest_fin.log
