#!/bin/bash
# A shorter round-end pass when the walk kernels have not changed since scripts/final_profile.sh ran (its SQ pass and its c3 / c5 traces
# stay valid): GPU tests, the c2 kernel trace, the two PMC traffic passes, then the default bench, which reads profiles/pmc_traffic.json.
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
# the walk kernel of the c2 launches at ef 64 / 256 as rocprofv3 names it (another kernel became the default? export these)
K64=${WALK_KERNEL_EF64:-"walk_kernel<0, 1, 1, true, false, 8>"}; K256=${WALK_KERNEL_EF256:-"walk_kernel<0, 1, 4, true, false, 8>"}
timeout 900 python -m pytest tests -m gpu -q > $OUT/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/final_pytest.log
cd /tmp; export TMPDIR=/tmp
export COS_BENCH_FULL_RECORD=bench_full_record_of_a_profiler_pass.json
MAIN="--steps 20 --warmup 5 --configs none --no-cpu-baseline --no-hbm-probe --no-host-api --ef-sweep 256"
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py $MAIN > $OUT/final_bench_c2_under_rocprofv3.json 2> $OUT/final_kt.err
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $OUT/final_kernel_trace_c2.txt
PM="--steps 8 --warmup 2 --configs none --no-cpu-baseline --no-hbm-probe --no-host-api --ef-sweep 256 --recall-queries 2048"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/bench.py $PM > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/bench.py $PM > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
cd $R
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c2 64 "$K64" /tmp/p_w/w_results.db ref 20992 > $OUT/pmc_traffic_ef64.json
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c2 256 "$K256" /tmp/p_w/w_results.db ref 20992 > $OUT/pmc_traffic_ef256.json
python scripts/rocprof_summary.py /tmp/p_f/f_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_fetch_size.txt
python scripts/rocprof_summary.py /tmp/p_w/w_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_write_size.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
unset COS_BENCH_FULL_RECORD
COS_BENCH_FULL_RECORD=final_bench_all_configs_full_record.json timeout 1500 python bench.py > $OUT/final_bench_all_configs.json 2> $OUT/final_bench_all_configs.err; echo "bench rc=$?"
head -c 400 $OUT/final_bench_all_configs.json; echo; cat $OUT/pmc_traffic_ef64.json | head -c 1200
