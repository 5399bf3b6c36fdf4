#!/bin/bash
# the last GPU call of the round: profiler passes of the final code, the default bench (reads the json the passes wrote), the GPU tests
R=$PWD; OUT=$R/gpurun_out
bash scripts/gpu_profile_only.sh > $OUT/final_profile_only.log 2>&1; tail -3 $OUT/final_profile_only.log
cd $R
COS_BENCH_FULL_RECORD=final_bench_all_configs_full_record.json timeout 1200 python bench.py > $OUT/final_bench_all_configs.json 2> $OUT/final_bench_all_configs.err; echo "bench rc=$?"
timeout 600 python -m pytest tests -m gpu -q > $OUT/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/final_pytest.log
head -c 300 $OUT/final_bench_all_configs.json
