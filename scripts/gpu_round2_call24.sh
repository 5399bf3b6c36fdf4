#!/bin/bash
# GPU call 24 of round 2: learned-sparse index after the key-pointer / byte-flag change: parity tests + a first timing
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_sparse.py -m gpu -q --timeout 300 --tb=short > $O/r2_c24_pytest.log 2>&1; tail -3 $O/r2_c24_pytest.log
timeout 200 python scripts/bench_sparse.py > $O/r2_c24_sparse.json 2> $O/r2_c24_sparse.err; tail -2 $O/r2_c24_sparse.err; cat $O/r2_c24_sparse.json
