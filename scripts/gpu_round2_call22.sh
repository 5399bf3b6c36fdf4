#!/bin/bash
# GPU call 22 of round 2: winner pre-screen in both walk kernels — full GPU suite + the driver's bench command
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --tb=short > $O/r2_c22_pytest.log 2>&1; tail -6 $O/r2_c22_pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2_c22_bench.json 2> $O/r2_c22_bench.err; tail -1 $O/r2_c22_bench.err
python -c "
import json;d=json.load(open('$O/r2_c22_bench.json'));print({k:d[k] for k in ('value','ms_per_step','recall_at_10','single_batch_qps','single_batch_qps_throughput_kernel','single_batch_latency_walk_identical_to_throughput_walk')}, d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['parity_vs_oracle'], d['ef_sweep'])"
