#!/bin/bash
# Round-end evidence on the GPU box (run from the repo root as the LAST gpurun call of a round): summaries land in gpurun_out/ and are copied
# to profiles/r06_final_* afterwards (scripts/collect_final_profile.py r06).  Round 6: the main workload is the metric's own shard.
#   1. the GPU test suite;
#   2. the main workload (one 12.5M x 1024 shard, M0 256 / M 64, reference filter, ef auto-selected) under rocprofv3: kernel trace, then the
#      PMC passes, each in its own run with the kernel trace only: FETCH_SIZE / WRITE_SIZE (-> profiles/pmc_traffic.json through
#      scripts/pmc_traffic.py, split by dispatch) and one SQ instruction pass (-> profiles/pmc_issue.json);
#   3. the same four passes for c2 (1M x 768) as a main workload, FETCH / WRITE passes for c2_uniform and c2_sigma01;
#   4. kernel traces + FETCH / WRITE passes of c5 and c3 (their own scripts) -> per-dispatch traffic of their dominant kernels;
#   5. the default bench (every BASELINE config, driver flags) — the line the driver will reproduce; it reads the json files of 2-4, so the
#      traffic / issue figures in it belong to the same build.
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/final_pytest.log
cd /tmp; export TMPDIR=/tmp
export COS_BENCH_FULL_RECORD=bench_full_record_of_a_profiler_pass.json
S="python $R/scripts/rocprof_summary.py"; NOB="link_kernel\|evict_kernel\|claim_kernel"
# ---- 2. the main workload -------------------------------------------------------------------------------------------------------
MAIN="--steps 16 --warmup 4 --configs none --no-cpu-baseline --no-hbm-probe --no-append-probe"
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py $MAIN > $OUT/final_bench_main_under_rocprofv3.json 2> $OUT/final_kt.err
$S /tmp/p_kt/kt_results.db | grep -v "$NOB" > $OUT/final_kernel_trace_main.txt
( cd $R/scripts; python trace_timeline.py /tmp/p_kt/kt_results.db 3 > $OUT/final_step_timeline_main.txt 2>&1 )
EF=$(python -c "import json; print(json.load(open('$OUT/final_bench_main_under_rocprofv3.json'))['config']['ef_search'])")
R_=$(python -c "ef=$EF; print(1 if ef<=64 else (2 if ef<=128 else (4 if ef<=256 else 8)))")
KMAIN="walk_kernel<0, 1, $R_, true, false"     # both level ranges (four / eight row buffers above ef 64)
echo "main: ef $EF kernel $KMAIN"
PM="--ef $EF --steps 8 --warmup 2 --configs none --no-cpu-baseline --no-hbm-probe --no-append-probe --recall-queries 2048"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/bench.py $PM > $OUT/pmc_fetch_bench_main.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/bench.py $PM > $OUT/pmc_write_bench_main.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d /tmp/p_s2 -o s2 -- python $R/bench.py $PM > $OUT/pmc_sq2_bench_main.json 2> $OUT/pmc_sq2.err
cd $R
EVALS=$(python -c "import json; p=json.load(open('gpurun_out/pmc_fetch_bench_main.json'))['roofline']['parts']; print(int(p['walk_upper']['evals']+p['walk_lower']['evals']))")
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c4shard $EF "$KMAIN" /tmp/p_w/w_results.db ref > $OUT/pmc_traffic_main.json
python scripts/pmc_issue.py /tmp/p_s2/s2_results.db 32768 c4shard $EF "$KMAIN" $EVALS ref > $OUT/pmc_issue_main.json
$S /tmp/p_f/f_results.db | grep -v "$NOB" > $OUT/final_pmc_fetch_size_main.txt
$S /tmp/p_w/w_results.db | grep -v "$NOB" > $OUT/final_pmc_write_size_main.txt
$S /tmp/p_s2/s2_results.db | grep -v "$NOB" > $OUT/final_pmc_sq_instruction_mix_main.txt 2>> $OUT/pmc_sq2.err
# ---- 3. c2 as a main workload (ef 64 selected + the ef 256 sweep entry), c2_uniform, c2_sigma01 --------------------------------------
cd /tmp
C2="--workload c2 --steps 16 --warmup 4 --configs none --no-cpu-baseline --no-hbm-probe --no-append-probe --no-host-api --ef-sweep 256"
rocprofv3 --kernel-trace --stats -d /tmp/p_c2 -o c2 -- python $R/bench.py $C2 > $OUT/final_bench_c2_under_rocprofv3.json 2> $OUT/final_c2_kt.err
$S /tmp/p_c2/c2_results.db | grep -v "$NOB" > $OUT/final_kernel_trace_c2.txt
C2P="--workload c2 --steps 8 --warmup 2 --configs none --no-cpu-baseline --no-hbm-probe --no-append-probe --no-host-api --ef-sweep 256 --recall-queries 2048"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_c2f -o c2f -- python $R/bench.py $C2P > $OUT/pmc_fetch_bench_c2.json 2> $OUT/pmc_fetch_c2.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_c2w -o c2w -- python $R/bench.py $C2P > $OUT/pmc_write_bench_c2.json 2> $OUT/pmc_write_c2.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d /tmp/p_c2s -o c2s -- python $R/bench.py $C2P > $OUT/pmc_sq2_bench_c2.json 2> $OUT/pmc_sq2_c2.err
cd $R
EF2=$(python -c "import json; print(json.load(open('gpurun_out/pmc_fetch_bench_c2.json'))['config']['ef_search'])")
EV2=$(python -c "import json; p=json.load(open('gpurun_out/pmc_fetch_bench_c2.json'))['roofline']['parts']; print(int(p['walk_upper']['evals']+p['walk_lower']['evals']))")
python scripts/pmc_traffic.py /tmp/p_c2f/c2f_results.db 32768 c2 $EF2 "walk_kernel<0, 1, 1, true, false, 8>" /tmp/p_c2w/c2w_results.db ref 256 > $OUT/pmc_traffic_c2_ef64.json
python scripts/pmc_traffic.py /tmp/p_c2f/c2f_results.db 32768 c2 256 "walk_kernel<0, 1, 4, true, false" /tmp/p_c2w/c2w_results.db ref 256 > $OUT/pmc_traffic_c2_ef256.json
python scripts/pmc_issue.py /tmp/p_c2s/c2s_results.db 32768 c2 $EF2 "walk_kernel<0, 1, 1, true, false, 8>" $EV2 ref > $OUT/pmc_issue_c2.json
$S /tmp/p_c2f/c2f_results.db | grep -v "$NOB" > $OUT/final_pmc_fetch_size_c2.txt
$S /tmp/p_c2w/c2w_results.db | grep -v "$NOB" > $OUT/final_pmc_write_size_c2.txt
$S /tmp/p_c2s/c2s_results.db | grep -v "$NOB" > $OUT/final_pmc_sq_instruction_mix_c2.txt 2>> $OUT/pmc_sq2_c2.err
cd /tmp
for W in c2_uniform c2_sigma01; do
  WP="--workload $W --steps 6 --warmup 2 --configs none --no-cpu-baseline --no-hbm-probe --no-append-probe --no-host-api --ef-sweep= --recall-queries 2048"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_${W}f -o f -- python $R/bench.py $WP > $OUT/pmc_fetch_bench_$W.json 2> $OUT/pmc_fetch_$W.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_${W}w -o w -- python $R/bench.py $WP > $OUT/pmc_write_bench_$W.json 2> $OUT/pmc_write_$W.err
  EFW=$(python -c "import json; print(json.load(open('$OUT/pmc_fetch_bench_$W.json'))['config']['ef_search'])")
  RW=$(python -c "ef=$EFW; print(1 if ef<=64 else (2 if ef<=128 else (4 if ef<=256 else 8)))")
  ( cd $R; python scripts/pmc_traffic.py /tmp/p_${W}f/f_results.db 32768 $W $EFW "walk_kernel<0, 1, $RW, true, false" /tmp/p_${W}w/w_results.db ref > $OUT/pmc_traffic_$W.json )
done
# ---- 4. c5 and c3 through their own scripts ----------------------------------------------------------------------------------------
rocprofv3 --kernel-trace --stats -d /tmp/p_c5 -o c5 -- python $R/scripts/bench_c5.py --cpu-seconds 0 > $OUT/final_c5.json 2> $OUT/final_c5.err
$S /tmp/p_c5/c5_results.db > $OUT/final_kernel_trace_c5.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_c5f -o c5f -- python $R/scripts/bench_c5.py --cpu-seconds 0 > $OUT/pmc_fetch_c5.json 2> $OUT/pmc_fetch_c5.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_c5w -o c5w -- python $R/scripts/bench_c5.py --cpu-seconds 0 > $OUT/pmc_write_c5.json 2> $OUT/pmc_write_c5.err
rocprofv3 --kernel-trace --stats -d /tmp/p_c3 -o c3 -- python $R/scripts/bench_c3.py --walk-n 1000000 --cpu-seconds 0 > $OUT/final_c3.json 2> $OUT/final_c3.err
$S /tmp/p_c3/c3_results.db > $OUT/final_kernel_trace_c3.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_c3f -o c3f -- python $R/scripts/bench_c3.py --walk-n 1000000 --cpu-seconds 0 > $OUT/pmc_fetch_c3.json 2> $OUT/pmc_fetch_c3.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_c3w -o c3w -- python $R/scripts/bench_c3.py --walk-n 1000000 --cpu-seconds 0 > $OUT/pmc_write_c3.json 2> $OUT/pmc_write_c3.err
cd $R
python scripts/pmc_traffic_kernel.py /tmp/p_c5f/c5f_results.db /tmp/p_c5w/c5w_results.db c5_walk "walk_lat4_kernel" 256 > $OUT/pmc_traffic_c5_walk.json
python scripts/pmc_traffic_kernel.py /tmp/p_c3f/c3f_results.db /tmp/p_c3w/c3w_results.db c3_scan "flat_scan_q2_fp4" > $OUT/pmc_traffic_c3_scan.json
python scripts/pmc_traffic_kernel.py /tmp/p_c3f/c3f_results.db /tmp/p_c3w/c3w_results.db c3_scan_i8 "flat_scan_q2_areg" > $OUT/pmc_traffic_c3_scan_i8.json
python scripts/pmc_traffic_kernel.py /tmp/p_c3f/c3f_results.db /tmp/p_c3w/c3w_results.db c3_walk "walk_kernel<1, " 8192 > $OUT/pmc_traffic_c3_walk.json
$S /tmp/p_c3f/c3f_results.db > $OUT/final_pmc_fetch_size_c3.txt
$S /tmp/p_c5f/c5f_results.db > $OUT/final_pmc_fetch_size_c5.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json; cp profiles/pmc_issue.json $OUT/pmc_issue.json
# ---- 5. the default bench ------------------------------------------------------------------------------------------------------------
unset COS_BENCH_FULL_RECORD
COS_BENCH_FULL_RECORD=final_bench_all_configs_full_record.json timeout 1700 python bench.py > $OUT/final_bench_all_configs.json 2> $OUT/final_bench_all_configs.err; echo "bench rc=$?"
head -c 700 $OUT/final_bench_all_configs.json; echo; head -12 $OUT/final_kernel_trace_main.txt; head -c 900 $OUT/pmc_traffic_main.json; echo; head -c 500 $OUT/pmc_issue_main.json; echo; tail -2 $OUT/final_step_timeline_main.txt
