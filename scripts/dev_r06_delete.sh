#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_delete}
timeout 1200 python -m pytest tests/test_gpu_append.py tests/test_gpu_builder.py tests/test_gpu_parity.py tests/test_gpu_walk_order.py tests/test_gpu_meta.py -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/${TAG}_pytest.log
