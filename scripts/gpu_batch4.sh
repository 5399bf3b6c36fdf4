#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_cal -o cal -- python $R/scripts/pmc_calibrate.py run > $OUT/r04_pmc_calibrate_run.jsonl 2> $OUT/r04_pmc_calibrate.err
cat $OUT/r04_pmc_calibrate_run.jsonl
python $R/scripts/pmc_calibrate.py read /tmp/p_cal/cal_results.db $OUT/r04_pmc_calibration.json
export PROBE_COLS=8192
PROBE_REPS=6 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/scripts/table_probe.py > $OUT/r04_b4_fetch.jsonl 2> $OUT/r04_b4_fetch.err
python $R/scripts/rocprof_summary.py /tmp/p_f/f_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/r04_b4_pmc_fetch_size.txt
grep -A8 "by level range" $OUT/r04_b4_pmc_fetch_size.txt
