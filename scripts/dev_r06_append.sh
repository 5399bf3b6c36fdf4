#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_append}
timeout 900 python -m pytest tests/test_gpu_append.py tests/test_gpu_builder.py tests/test_gpu_walk_table.py -m gpu -q -x > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/${TAG}_pytest.log
COS_BENCH_FULL_RECORD=${TAG}_bench_full_record.json timeout 1700 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"; tail -5 $OUT/${TAG}_bench.err; head -c 3000 $OUT/${TAG}_bench.json
