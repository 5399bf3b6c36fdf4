#!/bin/bash
# GPU call 21 of round 2: smoke() and the N > 1 code path of bench.py forced on one GPU (shard set, RCCL all-gather with a world of one)
O=gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r2_c21_smoke.log 2>&1; tail -2 $O/r2_c21_smoke.log
COS_FORCE_DIST=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-hbm-probe --ef-sweep 256 > $O/r2_c21_bench_forced_dist.json 2> $O/r2_c21_bench_forced_dist.err; tail -2 $O/r2_c21_bench_forced_dist.err
python -c "
import json;d=json.load(open('$O/r2_c21_bench_forced_dist.json'));print({k:d[k] for k in ('value','n_gpus','recall_at_10','single_batch_qps')}, d['config']['exchange'], d['roofline']['frac'])"
