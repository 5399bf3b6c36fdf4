#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_meta.py tests/test_gpu_builder.py -m gpu -q -x > $OUT/r3_pytest_meta.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/r3_pytest_meta.log
