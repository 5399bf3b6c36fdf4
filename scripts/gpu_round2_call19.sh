#!/bin/bash
# GPU call 19 of round 2: round-end evidence — full GPU suite, the driver's bench command, the same command under a rocprofv3 kernel
# trace, c5 under a kernel trace
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --tb=short > $O/r2_c19_pytest.log 2>&1; tail -6 $O/r2_c19_pytest.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2_c19_bench.json 2> $O/r2_c19_bench.err; tail -1 $O/r2_c19_bench.err
python -c "
import json;d=json.load(open('$O/r2_c19_bench.json'));print({k:d[k] for k in ('value','ms_per_step','recall_at_10','single_batch_qps')}, d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['parity_vs_oracle'])"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r2_c19_bench_under_rocprof.json 2> $O/r2_c19_rocprof.err
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $O/r2_c19_bench_kernel_trace.txt; head -8 $O/r2_c19_bench_kernel_trace.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o c5 -- python $R/scripts/bench_c5.py > $O/r2_c19_c5.json 2> $O/r2_c19_c5.err
python $R/scripts/rocprof_summary.py /tmp/p_c5/c5_results.db > $O/r2_c19_c5_kernel_trace.txt; grep "bm25\|walk_lat\|finalize\|rrf" $O/r2_c19_c5_kernel_trace.txt | head -8
cd $R
