#!/bin/bash
# Round-end evidence on the GPU box (run from the repo root): the default bench under rocprofv3 kernel trace, then the two
# PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run, kernel-trace only) that feed profiles/pmc_traffic.json.
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py > $OUT/final_bench.json 2> $OUT/final_bench.err
python $R/scripts/rocprof_summary.py /tmp/p_kt/kt_results.db > $OUT/final_kt.txt
PM="--no-cpu-baseline --no-hbm-probe --steps 256 --ef-sweep 256"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/bench.py $PM > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/bench.py $PM > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
cd $R
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 8192 c2 64 "walk_kernel<0, 1, 1, true, false>" /tmp/p_w/w_results.db > $OUT/pmc_traffic_ef64.json
python scripts/pmc_traffic.py /tmp/p_f/f_results.db 8192 c2 256 "walk_kernel<0, 1, 4, true, false>" /tmp/p_w/w_results.db > $OUT/pmc_traffic_ef256.json
python scripts/rocprof_summary.py /tmp/p_f/f_results.db > $OUT/final_pmc_fetch.txt
python scripts/rocprof_summary.py /tmp/p_w/w_results.db > $OUT/final_pmc_write.txt
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
head -c 600 $OUT/final_bench.json; echo; head -8 $OUT/final_kt.txt; cat $OUT/pmc_traffic_ef64.json
