#!/bin/bash
# Round-end evidence on the GPU box (run from the repo root as the LAST gpurun call of a round): summaries land in gpurun_out/ and
# are copied to profiles/r03_final_* afterwards.
#   1. the GPU test suite;
#   2. the default bench (every BASELINE config, driver flags) — the line the driver will reproduce;
#   3. rocprofv3 kernel trace of the main workload (c2 + its ef sweep), then the two PMC traffic passes (FETCH_SIZE / WRITE_SIZE, each in
#      its own run, kernel trace only) that feed profiles/pmc_traffic.json, and two SQ passes for the walk kernel;
#   4. kernel traces of c5, c3 and the learned-sparse index (small scripts: an index build under PMC counters is slow, so counters are
#      only collected where the script does little else).
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/final_pytest.log
COS_BENCH_FULL_RECORD=final_bench_all_configs_full_record.json timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/final_bench_all_configs.json 2> $OUT/final_bench_all_configs.err; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
export COS_BENCH_FULL_RECORD=bench_full_record_of_a_profiler_pass.json
MAIN="--steps 20 --warmup 5 --configs none --no-cpu-baseline --no-hbm-probe"
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py $MAIN > $OUT/final_bench_c2_under_rocprofv3.json 2> $OUT/final_kt.err
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $OUT/final_kernel_trace_c2.txt
PM="--steps 8 --warmup 2 --configs none --no-cpu-baseline --no-hbm-probe --ef-sweep 256 --recall-queries 2048"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/bench.py $PM > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/bench.py $PM > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/p_s1 -o s1 -- python $R/bench.py $PM > $OUT/pmc_sq1_bench.json 2> $OUT/pmc_sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d /tmp/p_s2 -o s2 -- python $R/bench.py $PM > $OUT/pmc_sq2_bench.json 2> $OUT/pmc_sq2.err
cd $R
# the walk of a 32768-query step is two dispatches of the kernel (cut after level 2, cos_index_walk_order_cuts): bytes per step = 2 x the average
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c2 64 "walk_kernel<0, 1, 1, true, false, 8>" /tmp/p_w/w_results.db 2 > $OUT/pmc_traffic_ef64.json
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c2 256 "walk_kernel<0, 1, 4, true, false, 8>" /tmp/p_w/w_results.db 2 > $OUT/pmc_traffic_ef256.json
python scripts/rocprof_summary.py /tmp/p_f/f_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_fetch_size.txt
python scripts/rocprof_summary.py /tmp/p_w/w_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_write_size.txt
python scripts/rocprof_summary.py /tmp/p_s1/s1_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_sq_wave_cycles.txt 2>> $OUT/pmc_sq1.err
python scripts/rocprof_summary.py /tmp/p_s2/s2_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_sq_instruction_mix.txt 2>> $OUT/pmc_sq2.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o c5 -- python $R/scripts/bench_c5.py --cpu-seconds 0 > $OUT/final_c5.json 2> $OUT/final_c5.err
python $R/scripts/rocprof_summary.py /tmp/p_c5/c5_results.db > $OUT/final_kernel_trace_c5.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_c3 -o c3 -- python $R/scripts/bench_c3.py --walk-n 0 --cpu-seconds 0 > $OUT/final_c3.json 2> $OUT/final_c3.err
python $R/scripts/rocprof_summary.py /tmp/p_c3/c3_results.db > $OUT/final_kernel_trace_c3.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_sp -o sp -- python $R/scripts/bench_sparse.py > $OUT/final_sparse.json 2> $OUT/final_sparse.err
python $R/scripts/rocprof_summary.py /tmp/p_sp/sp_results.db > $OUT/final_kernel_trace_sparse.txt
cd $R
head -c 300 $OUT/final_bench_all_configs.json; echo; head -6 $OUT/final_kernel_trace_c2.txt; cat $OUT/pmc_traffic_ef64.json; grep "walk_kernel" $OUT/final_pmc_sq_wave_cycles.txt | head -8
