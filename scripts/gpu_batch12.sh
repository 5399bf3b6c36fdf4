#!/bin/bash
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_walk_table.py -m gpu -q -x > $OUT/r04_b12_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/r04_b12_pytest.log
export PROBE_COLS=8192
for o in 1 0; do COS_WALK_TABLE_ORDER=$o timeout 200 python scripts/table_probe.py 2>/dev/null | tail -1 > $OUT/r04_b12_order$o.json; echo "table order $o"; cut -c1-420 $OUT/r04_b12_order$o.json; done
cd /tmp; export TMPDIR=/tmp
PROBE_REPS=6 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p_f -o f -- python $R/scripts/table_probe.py > $OUT/r04_b12_fetch.jsonl 2> $OUT/r04_b12_fetch.err
python $R/scripts/rocprof_summary.py /tmp/p_f/f_results.db | grep -v "link_kernel\|evict_kernel\|claim_kernel" > $OUT/r04_b12_pmc_fetch_size.txt
grep -A4 "by level range" $OUT/r04_b12_pmc_fetch_size.txt | grep "32768"
