#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
COS_BENCH_FULL_RECORD=final2_bench_all_configs_full_record.json timeout 280 python bench.py > $OUT/final2_bench_all_configs.json 2> $OUT/final2_bench_all_configs.err; echo "bench rc=$?"; head -c 400 $OUT/final2_bench_all_configs.json; tail -2 $OUT/final2_bench_all_configs.err
