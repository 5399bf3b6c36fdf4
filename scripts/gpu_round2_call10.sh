#!/bin/bash
# GPU call 10 of round 2: latency variant of the walk — full GPU suite (both variants through the shared parity helpers),
# launch-size sweep against the throughput kernel, default bench
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --timeout 600 --tb=short > $O/r2_c10_pytest.log 2>&1; tail -25 $O/r2_c10_pytest.log
timeout 300 python scripts/latency_sweep.py > $O/r2_c10_latency_sweep.jsonl 2> $O/r2_c10_latency_sweep.err; tail -3 $O/r2_c10_latency_sweep.err; cat $O/r2_c10_latency_sweep.jsonl
timeout 300 python bench.py --steps 20 --warmup 5 > $O/r2_c10_bench.json 2> $O/r2_c10_bench.err; tail -2 $O/r2_c10_bench.err; head -c 1500 $O/r2_c10_bench.json
