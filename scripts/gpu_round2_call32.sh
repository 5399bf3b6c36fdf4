#!/bin/bash
# f32-storage walk at c2: sweep + rocprofv3 kernel trace
O=/root/repo/gpurun_out; mkdir -p $O; R=/root/repo
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_f32 -o f32 -- python $R/scripts/bench_c2_sweep.py --cases clustered:f32 --efs 64,256 > $O/r2_c32_f32_sweep.json 2> $O/r2_c32_f32.err
python $R/scripts/rocprof_summary.py /tmp/p_f32/f32_results.db > $O/r2_c32_f32_kt.txt; head -14 $O/r2_c32_f32_kt.txt; cat $O/r2_c32_f32_sweep.json | head -c 1500
