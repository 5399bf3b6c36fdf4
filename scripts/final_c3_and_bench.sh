#!/bin/bash
# Round 6, after the FP4 scan moved to two waves per SIMD: the GPU suite, c3's kernel trace + FETCH / WRITE passes (-> profiles/pmc_traffic.json
# entries c3_scan / c3_scan_i8 / c3_walk on the box, copied to gpurun_out/), then the default bench + forced-dist run + smoke
# (scripts/final_bench_only.sh).  Everything else of scripts/final_profile.sh is unchanged by that kernel and keeps its committed summaries.
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/final_pytest.log
cd /tmp; export TMPDIR=/tmp
S="python $R/scripts/rocprof_summary.py"
rocprofv3 --kernel-trace --stats -d /tmp/p_c3 -o c3 -- python $R/scripts/bench_c3.py --walk-n 1000000 --cpu-seconds 0 > $OUT/final_c3.json 2> $OUT/final_c3.err
$S /tmp/p_c3/c3_results.db > $OUT/final_kernel_trace_c3.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_c3f -o c3f -- python $R/scripts/bench_c3.py --walk-n 1000000 --cpu-seconds 0 > $OUT/pmc_fetch_c3.json 2> $OUT/pmc_fetch_c3.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_c3w -o c3w -- python $R/scripts/bench_c3.py --walk-n 1000000 --cpu-seconds 0 > $OUT/pmc_write_c3.json 2> $OUT/pmc_write_c3.err
cd $R
python scripts/pmc_traffic_kernel.py /tmp/p_c3f/c3f_results.db /tmp/p_c3w/c3w_results.db c3_scan "flat_scan_q2_fp4" > $OUT/pmc_traffic_c3_scan.json
python scripts/pmc_traffic_kernel.py /tmp/p_c3f/c3f_results.db /tmp/p_c3w/c3w_results.db c3_scan_i8 "flat_scan_q2_areg" > $OUT/pmc_traffic_c3_scan_i8.json
python scripts/pmc_traffic_kernel.py /tmp/p_c3f/c3f_results.db /tmp/p_c3w/c3w_results.db c3_walk "walk_kernel<1, " 8192 > $OUT/pmc_traffic_c3_walk.json
$S /tmp/p_c3f/c3f_results.db > $OUT/final_pmc_fetch_size_c3.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
head -c 600 $OUT/pmc_traffic_c3_scan.json; echo; grep -i "flat_scan" $OUT/final_kernel_trace_c3.txt | head -8
bash scripts/final_bench_only.sh
