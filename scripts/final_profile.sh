#!/bin/bash
# Round-end evidence on the GPU box (run from the repo root): the default bench under rocprofv3 kernel trace, the two PMC
# traffic passes (FETCH_SIZE / WRITE_SIZE, each in its own run, kernel-trace only) that feed profiles/pmc_traffic.json, and two SQ
# passes (wave-cycle split; instruction mix) for the walk kernel.  Summaries land in gpurun_out/ (copy them to profiles/).
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $OUT/final_kt.txt
PM="--no-cpu-baseline --no-hbm-probe --steps 8 --warmup 2 --ef-sweep 256 --recall-queries 2048"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/bench.py $PM > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/bench.py $PM > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/p_s1 -o s1 -- python $R/bench.py $PM > $OUT/pmc_sq1_bench.json 2> $OUT/pmc_sq1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace -d /tmp/p_s2 -o s2 -- python $R/bench.py $PM > $OUT/pmc_sq2_bench.json 2> $OUT/pmc_sq2.err
cd $R
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c2 64 "walk_kernel<0, 1, 1, true, false, 8>" /tmp/p_w/w_results.db > $OUT/pmc_traffic_ef64.json
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 32768 c2 256 "walk_kernel<0, 1, 4, true, false, 8>" /tmp/p_w/w_results.db > $OUT/pmc_traffic_ef256.json
python scripts/rocprof_summary.py /tmp/p_f/f_results.db > $OUT/final_pmc_fetch.txt
python scripts/rocprof_summary.py /tmp/p_w/w_results.db > $OUT/final_pmc_write.txt
python scripts/rocprof_summary.py /tmp/p_s1/s1_results.db > $OUT/final_pmc_sq_cycles.txt 2>> $OUT/pmc_sq1.err
python scripts/rocprof_summary.py /tmp/p_s2/s2_results.db > $OUT/final_pmc_sq_insts.txt 2>> $OUT/pmc_sq2.err
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
head -c 400 $OUT/final_bench.json; echo; head -8 $OUT/final_kt.txt; cat $OUT/pmc_traffic_ef64.json; grep walk_kernel $OUT/final_pmc_sq_cycles.txt | head -12
