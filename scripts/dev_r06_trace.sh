#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r06_trace}
timeout 600 python -m pytest tests/test_gpu_meta.py -m gpu -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/${TAG}_pytest.log
cd /tmp; export TMPDIR=/tmp
COS_BENCH_FULL_RECORD=${TAG}_full.json timeout 900 rocprofv3 --kernel-trace -d /tmp/p_tl -o tl -- python $R/bench.py --ef 112 --configs none --no-cpu-baseline --no-hbm-probe --steps 12 --warmup 3 --recall-queries 2048 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
cd $R/scripts; python trace_timeline.py /tmp/p_tl/tl_results.db 3 > $OUT/${TAG}_timeline.txt 2>&1; tail -3 $OUT/${TAG}_timeline.txt; python rocprof_summary.py /tmp/p_tl/tl_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/${TAG}_summary.txt; head -30 $OUT/${TAG}_summary.txt
