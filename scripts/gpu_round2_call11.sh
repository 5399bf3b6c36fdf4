#!/bin/bash
# GPU call 11 of round 2: software-pipelined BM25 scoring kernel (parity + c5 timing), single-batch numbers with the latency walk
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hybrid.py -m gpu -q --timeout 600 --tb=short > $O/r2_c11_pytest.log 2>&1; tail -5 $O/r2_c11_pytest.log
timeout 400 python scripts/bench_c5.py > $O/r2_c11_c5.json 2> $O/r2_c11_c5.err; tail -2 $O/r2_c11_c5.err; cat $O/r2_c11_c5.json
