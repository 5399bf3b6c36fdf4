#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_sparse.py tests/test_abi.py -m gpu -q > $OUT/r3_pytest_sparse.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r3_pytest_sparse.log
timeout 600 python scripts/bench_sparse.py > $OUT/r3_sparse_400k.json 2> $OUT/r3_sparse.err; echo "sparse rc=$?"; cat $OUT/r3_sparse_400k.json; tail -3 $OUT/r3_sparse.err
