#!/bin/bash
# GPU-box check used while developing (run from the repo root through gpurun)
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_builder.py tests/test_gpu_meta.py tests/test_gpu_dim1024.py -m gpu -q -x > $OUT/r3_pytest_link.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r3_pytest_link.log
COS_BUILD_PROFILE=1 timeout 600 python bench.py --steps 10 --warmup 3 --configs none --no-cpu-baseline --ef-sweep "" --no-hbm-probe > $OUT/r3_bench_c2_link_dpp.json 2> $OUT/r3_a.err; echo "A rc=$?"; grep -i "cos_index_build" $OUT/r3_a.err | head -3
