#!/bin/bash
# Round-end evidence on the GPU box (run from the repo root as the LAST gpurun call of a round): summaries land in gpurun_out/ and
# are copied to profiles/r05_final_* afterwards.
#   1. the GPU test suite;
#   2. rocprofv3 kernel trace of the main workload (c2), then the PMC passes, each in its own run with the kernel trace only:
#      FETCH_SIZE / WRITE_SIZE (-> profiles/pmc_traffic.json through scripts/pmc_traffic.py, split by dispatch, factor from
#      profiles/pmc_calibration.json) and one SQ instruction pass (-> profiles/pmc_issue.json through scripts/pmc_issue.py);
#   3. the default bench (every BASELINE config, driver flags) — the line the driver will reproduce; it reads the two json files of
#      step 2, so the traffic / issue figures in it belong to the same build;
#   4. kernel traces of c5 and c3 (small scripts), FETCH_SIZE pass of the c3 scan;
#   5. the metric's own config (c4shard, level_0_neighbors_count 256, neighbors_count 64, ef 128) as the main workload: kernel trace,
#      FETCH_SIZE and WRITE_SIZE passes -> its entry of profiles/pmc_traffic.json (round 4 had no trace and no counters for it).
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
# the walk kernel of the c2 launches at ef 64 / 256 as rocprofv3 names it (another kernel became the default? export these)
K64=${WALK_KERNEL_EF64:-"walk_kernel<0, 1, 1, true, false, 8>"}; K256=${WALK_KERNEL_EF256:-"walk_kernel<0, 1, 4, true, false"}   # (ef 256: four row buffers above the cut, eight below it)
timeout 900 python -m pytest tests -m gpu -q > $OUT/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/final_pytest.log
cd /tmp; export TMPDIR=/tmp
export COS_BENCH_FULL_RECORD=bench_full_record_of_a_profiler_pass.json
MAIN="--steps 20 --warmup 5 --configs none --no-cpu-baseline --no-hbm-probe --no-host-api --ef-sweep 256"
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py $MAIN > $OUT/final_bench_c2_under_rocprofv3.json 2> $OUT/final_kt.err
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $OUT/final_kernel_trace_c2.txt
PM="--steps 8 --warmup 2 --configs none --no-cpu-baseline --no-hbm-probe --no-host-api --ef-sweep 256 --recall-queries 2048"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/bench.py $PM > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/bench.py $PM > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d /tmp/p_s2 -o s2 -- python $R/bench.py $PM > $OUT/pmc_sq2_bench.json 2> $OUT/pmc_sq2.err
cd $R
EVALS=$(python - <<'PY'
import json
r = json.load(open("gpurun_out/pmc_fetch_bench.json")); p = r["roofline"]["parts"]; print(int(p["walk_upper"]["evals"] + p["walk_lower"]["evals"]))
PY
)
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c2 64 "$K64" /tmp/p_w/w_results.db ref 256 > $OUT/pmc_traffic_ef64.json
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c2 256 "$K256" /tmp/p_w/w_results.db ref 256 > $OUT/pmc_traffic_ef256.json
python scripts/pmc_issue.py /tmp/p_s2/s2_results.db 32768 c2 64 "$K64" $EVALS ref > $OUT/pmc_issue_ef64.json
python scripts/rocprof_summary.py /tmp/p_f/f_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_fetch_size.txt
python scripts/rocprof_summary.py /tmp/p_w/w_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_write_size.txt
python scripts/rocprof_summary.py /tmp/p_s2/s2_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_sq_instruction_mix.txt 2>> $OUT/pmc_sq2.err
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp profiles/pmc_issue.json $OUT/pmc_issue.json
unset COS_BENCH_FULL_RECORD
COS_BENCH_FULL_RECORD=final_bench_all_configs_full_record.json timeout 1500 python bench.py > $OUT/final_bench_all_configs.json 2> $OUT/final_bench_all_configs.err; echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o c5 -- python $R/scripts/bench_c5.py --cpu-seconds 0 > $OUT/final_c5.json 2> $OUT/final_c5.err
python $R/scripts/rocprof_summary.py /tmp/p_c5/c5_results.db > $OUT/final_kernel_trace_c5.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_c3 -o c3 -- python $R/scripts/bench_c3.py --walk-n 1000000 --cpu-seconds 0 > $OUT/final_c3.json 2> $OUT/final_c3.err
python $R/scripts/rocprof_summary.py /tmp/p_c3/c3_results.db > $OUT/final_kernel_trace_c3.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_c3f -o c3f -- python $R/scripts/bench_c3.py --walk-n 0 --cpu-seconds 0 > $OUT/pmc_fetch_c3.json 2> $OUT/pmc_fetch_c3.err
python $R/scripts/rocprof_summary.py /tmp/p_c3f/c3f_results.db > $OUT/final_pmc_fetch_size_c3.txt
# 5. the metric's own shard: 12.5M x 1024, M0 256 / M 64, ef 128 (the graph build runs under the profiler too: ~50 s per pass)
C4="--workload c4shard --m0 256 --m 64 --ef 128 --steps 8 --warmup 2 --configs none --no-cpu-baseline --no-hbm-probe --no-host-api --ef-sweep= --recall-queries 2048"
K128="walk_kernel<0, 1, 4, true, false"
COS_BENCH_FULL_RECORD=bench_full_record_of_a_profiler_pass.json rocprofv3 --kernel-trace --stats -d /tmp/p_c4 -o c4 -- python $R/bench.py $C4 > $OUT/final_bench_c4shard_under_rocprofv3.json 2> $OUT/final_c4_kt.err
python $R/scripts/rocprof_summary.py /tmp/p_c4/c4_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_kernel_trace_c4shard.txt
COS_BENCH_FULL_RECORD=bench_full_record_of_a_profiler_pass.json rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_c4f -o c4f -- python $R/bench.py $C4 > $OUT/pmc_fetch_c4shard.json 2> $OUT/pmc_fetch_c4.err
COS_BENCH_FULL_RECORD=bench_full_record_of_a_profiler_pass.json rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_c4w -o c4w -- python $R/bench.py $C4 > $OUT/pmc_write_c4shard.json 2> $OUT/pmc_write_c4.err
cd $R
python scripts/pmc_traffic.py /tmp/p_c4f/c4f_results.db 32768 c4shard 128 "$K128" /tmp/p_c4w/c4w_results.db ref 256 > $OUT/pmc_traffic_c4shard.json
python scripts/rocprof_summary.py /tmp/p_c4f/c4f_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_fetch_size_c4shard.txt
python scripts/rocprof_summary.py /tmp/p_c4w/c4w_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/final_pmc_write_size_c4shard.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
head -c 600 $OUT/final_bench_all_configs.json; echo; head -8 $OUT/final_kernel_trace_c2.txt; cat $OUT/pmc_traffic_ef64.json | head -c 900; echo; cat $OUT/pmc_issue_ef64.json | head -c 600; echo; head -c 700 $OUT/pmc_traffic_c4shard.json
