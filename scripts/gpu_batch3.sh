#!/bin/bash
# round-4 development batch: launch-shape variants of the walk on c2 (each its own process: the knobs are read once) + PMC traffic passes
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export PROBE_COLS=8192
PROBE_3=1 timeout 200 python scripts/table_probe.py 2>/dev/null | tail -1 > $OUT/r04_b3_default.json; cut -c1-330 $OUT/r04_b3_default.json
COS_WALK_PB=4 timeout 200 python scripts/table_probe.py 2>/dev/null | tail -1 > $OUT/r04_b3_pb4.json; cut -c1-330 $OUT/r04_b3_pb4.json
for sp in 4,2 3 4 3,1; do COS_WALK_SPLIT=$sp timeout 200 python scripts/table_probe.py 2>/dev/null | tail -1 > $OUT/r04_b3_split_$sp.json; echo "split $sp"; cut -c1-330 $OUT/r04_b3_split_$sp.json; done
cd /tmp; export TMPDIR=/tmp
PROBE_REPS=6 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/scripts/table_probe.py > $OUT/r04_b3_fetch.jsonl 2> $OUT/r04_b3_fetch.err
PROBE_REPS=6 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/p_w -o w -- python $R/scripts/table_probe.py > $OUT/r04_b3_write.jsonl 2> $OUT/r04_b3_write.err
python $R/scripts/rocprof_summary.py /tmp/p_f/f_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/r04_b3_pmc_fetch_size.txt
python $R/scripts/rocprof_summary.py /tmp/p_w/w_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/r04_b3_pmc_write_size.txt
grep "walk_kernel<0, 1, 1\|flat_codes\|finalize" $OUT/r04_b3_pmc_fetch_size.txt | head; grep "walk_kernel<0, 1, 1\|flat_codes\|finalize" $OUT/r04_b3_pmc_write_size.txt | head -6
